"""Thin tensor-level wrappers over the C ABI (device pointers + current HIP stream).

PyTorch owns memory and streams only; all arithmetic happens in libmsocr.so.
Activations are NHWC tensors [N, H, W, C] (possibly channel-slice views of a wider
concat buffer: the kernels take explicit strides / leading dimensions).
"""
import ctypes
import os

import torch

from . import _native as nat


# When set to a list, the wrappers below append one record per kernel launch: (start_event, end_event, kind, work, tag), with
# the events recorded on the launch stream (torch's current stream).  bench.py turns them into the live HIP-event rooflines.
#   kind "conv_gemm": work = (algorithmic direct-convolution FLOP, FLOP the MFMAs execute, direct-form layer bytes: input + output
#                     (+ residual) + weights at the tensor dtype), tag = (M, N, K, "direct"|"winograd")
#   kind "wino_in" / "wino_out" / "se_residual" / "maxpool" / ...: work = algorithmic HBM bytes (bytes in + bytes out)
#   kind "bilstm" / "attn_beam": work = (algorithmic bytes per SURVEY.md 8d, recurrent steps of the launch)
PROFILE = None


def _prof_begin():
    if PROFILE is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0


def _prof_end(e0, kind, work, tag=None):
    if e0 is None or PROFILE is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    PROFILE.append((e0, e1, kind, work, tag))


def _dt(t):
    if t.dtype == torch.float32:
        return nat.F32
    if t.dtype == torch.bfloat16:
        return nat.BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise nat.NativeError("native ops need device tensors (no CPU fallback)")


def _pixel_dense_ld(t):
    """Leading dimension (elements per pixel) of an NHWC tensor/view whose pixels are dense."""
    N, H, W, C = t.shape
    ld = t.stride(2)
    if t.stride(3) != 1 or t.stride(1) != W * ld or (N > 1 and t.stride(0) != H * W * ld) or ld < C:
        raise ValueError(f"tensor is not pixel-dense NHWC: shape {tuple(t.shape)} strides {t.stride()}")
    return ld


# Winograd F(2x2,3x3) for f32 3x3/1/1 convolutions with at least this many input channels (0 disables it).  Below 128
# channels the 16 transform-domain GEMMs have K < 128 and become HBM-bound themselves.
WINOGRAD_MIN_CIN = int(os.environ.get("MSOCR_WINOGRAD_MIN_CIN", "128"))


# Winograd workspace (V and Mw, 16 * tiles * (Cin + Cout) f32): ONE reusable arena per launch stream, grown lazily to the
# largest layer seen — calls on a stream are stream-ordered, so consecutive layers reuse the same bytes.  (A fresh torch.empty
# per call left every layer's high-water block cached in every one of the 32 stream pools: 279 GB reserved at 3072x4096.)
# Calls whose workspace would exceed WINO_WS_LIMIT are split over the batch dimension.
WINO_WS_LIMIT = int(os.environ.get("MSOCR_WINO_WS_LIMIT", str(1 << 30)))
_WINO_ARENA = {}


def _wino_workspace(nbytes, device):
    if torch.cuda.is_current_stream_capturing():  # a hipGraph capture owns its allocations (private pool)
        return torch.empty((nbytes,), dtype=torch.uint8, device=device)
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    buf = _WINO_ARENA.get(key)
    if buf is None or buf.numel() < nbytes:
        # the old block returns to the caching allocator on the stream it was allocated on: kernels already queued there
        # finish with it before any later allocation of that stream can reuse it
        buf = None
        _WINO_ARENA.pop(key, None)
        buf = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        _WINO_ARENA[key] = buf
    return buf


# Tall Winograd form F(4,3) x F(2,3) (csrc/winograd.hip, wino42_*): 24 transform points per 4x2 outputs = 3 multiplies per output
# instead of 4, V / Mw 3x instead of 4x.  Taken when it is cheaper for the map height: 24 * ceil(H/4) < 16 * ceil(H/2)
# (H = 4, 7, 8, 11, 12, >= 15 ...).  MSOCR_WINO_TALL=0 keeps every layer on F(2x2,3x3).
WINOGRAD_TALL = int(os.environ.get("MSOCR_WINO_TALL", "1"))


def _tall_pays(H):
    return 24 * (-(-H // 4)) < 16 * (-(-H // 2))


# Square form F(4,3) x F(4,3) on the points {0, +-3/2, +-2/3, inf} (csrc/winograd.hip, wino44_*; round 4): 36 points per 4x4 outputs =
# 2.25 multiplies and workspace words per output instead of 3.  Split-operand GEMMs only (precision="fp32" default); taken where it is
# cheaper than the tall form for the map width: 36 * ceil(W/4) < 24 * ceil(W/2).  MSOCR_WINO_SQUARE=0 keeps the tall form.
# Which layers: per layer the square form's rounding error is 1.1-1.3x the tall form's on the textbook points and 2.2-2.5x the tall
# form's on the new points; end to end (tests/test_gpu_f64.py, device error against f64 relative to the reference's own f32 error,
# bound 2.0, three weight sets) the recogniser reads 1.47-1.60 all tall, 1.79-1.91 with the square form on its Cin = 512 layers
# (+4.0 % pages/s on one box: 81.6 -> 84.9) and 1.71-2.24 with it on every eligible layer (+5.9 %: 86.4) — the default is the
# largest set that holds the bound on all three: Cin >= 512 (MSOCR_WINO_SQUARE_MIN_CIN=256 takes the rest).
WINOGRAD_SQUARE = int(os.environ.get("MSOCR_WINO_SQUARE", "1"))
WINOGRAD_SQUARE_MIN_CIN = int(os.environ.get("MSOCR_WINO_SQUARE_MIN_CIN", "512"))


def _square_pays(W):
    return 36 * (-(-W // 4)) < 24 * (-(-W // 2))


# Cin == 64 layers (TRBA conv0b, the 3x3s of ResNet-50 layer1): tall Winograd with the 24 GEMMs (K = 64) and the output transform
# fused in one kernel (csrc/winograd.hip, wino42_fused64_kernel) — unfused, such a layer is HBM-bound on Mw.  conv2d(pool2=True)
# folds the following 2x2/2 max-pool into the same kernel.  MSOCR_WINO_FUSED64=0 keeps these layers on the direct kernel.
WINOGRAD_FUSED64 = int(os.environ.get("MSOCR_WINO_FUSED64", "1"))


# Split-operand f32 ("bf16x3", csrc/conv_split.hip): the f32 1x1 convolutions and the Winograd-domain GEMMs of the tall form run on
# the bf16 matrix pipes with each operand split EXACTLY into three bf16 terms (six products per f32 product, f32 accumulate; the
# dropped terms are <= 2^-25 of a product).  The weight operand is split once, here, at load time.  0 = exact-f32 MFMA everywhere
# (precision="fp32-exact" of EAST / TRBA).
SPLIT_BF16X3 = int(os.environ.get("MSOCR_SPLIT", "1"))


def split_planes(t):
    """f32 tensor (any device) -> bf16 tensor [3, *t.shape] on t's device with t == p0 + p1 + p2 exactly (msocr_split_bf16x3_host)."""
    th = t.detach().float().cpu().contiguous()
    planes = torch.empty((3,) + tuple(th.shape), dtype=torch.int16)
    nat.check(nat.lib().msocr_split_bf16x3_host(th.data_ptr(), th.numel(), planes.data_ptr()), "split_bf16x3_host")
    return planes.view(torch.bfloat16).to(t.device)


def split_planes_ktile(t, nbatch, rows):
    """f32 tensor viewed as [nbatch][rows][k] -> bf16 planes [3, nbatch, k/32, rows, 32] on t's device: the K-tile-major layout the
    split-operand GEMM kernels read (msocr_split_bf16x3_ktile_host; include/msocr.h says why)."""
    th = t.detach().float().cpu().contiguous()
    k = th.numel() // (nbatch * rows)
    assert k % 32 == 0 and nbatch * rows * k == th.numel()
    planes = torch.empty((3, nbatch, k // 32, rows, 32), dtype=torch.int16)
    nat.check(nat.lib().msocr_split_bf16x3_ktile_host(th.data_ptr(), nbatch, rows, k, planes.data_ptr()), "split_bf16x3_ktile_host")
    return planes.view(torch.bfloat16).to(t.device)


def unsplit_planes_ktile(planes):
    """Inverse of split_planes_ktile for tests: planes [3, nb, k/32, rows, 32] -> f32 [nb, rows, k] = p0 + p1 + p2."""
    nb, kt, rows = planes.shape[1:4]
    return planes.float().sum(0).permute(0, 2, 1, 3).reshape(nb, rows, kt * 32)


# Below this reduction length (KH * KW * Cin) a convolution stays on the exact-f32 kernel: the split kernel's K-tiles of 32 with two
# barriers each lose to the exact lean kernel's K-tiles of 16 at K = 64 (55 against 64 TFLOP/s, profiles/r03_conv_layers_split_all.txt;
# these layers sit at 0.78 of their HBM roofline anyway).  Whole pipeline, same box, after the split kernel lost its spill:
# 70.4 / 70.9 / 71.2 pages/s with the threshold at 256 / 192 / 128.
SPLIT_MIN_K = int(os.environ.get("MSOCR_SPLIT_MIN_K", "128"))


def _split_eligible(w):
    Cout, KH, KW, Cin = w.shape
    return w.dtype == torch.float32 and Cin % 32 == 0 and Cout % 64 == 0 and KH * KW * Cin >= SPLIT_MIN_K


def attach_split(w, split=None):
    """Load-time: give a [Cout,1,1,Cin] f32 device weight its three bf16 planes (conv2d() then takes the split-operand kernels);
    split=False marks the weight as exact-f32 only (precision="fp32-exact").  Weights of other kernel sizes get their planes on
    first use as a strided / non-Winograd convolution (conv2d), so a 3x3 weight that only ever runs as Winograd holds none."""
    if split is False:
        w._msocr_nosplit = True
        return w
    Cout, KH, KW, Cin = w.shape
    if (SPLIT_BF16X3 if split is None else split) and KH == 1 and KW == 1 and _split_eligible(w):
        w._msocr_split = split_planes_ktile(w, 1, Cout)
    return w


def attach_winograd(w, split=None, square=True):
    """Load-time: give a [Cout,3,3,Cin] f32 device weight its transform-domain twins U = G g G^T ([16,Cout,Cin] f32 for F(2x2,3x3),
    [24,Cout,Cin] for the tall form F(4,3) x F(2,3); computed on the host in f64 by msocr_winograd[42]_weights_host).  conv2d() then
    takes the Winograd path for 3x3/1/1 calls."""
    Cout, KH, KW, Cin = w.shape
    if (WINOGRAD_MIN_CIN and WINOGRAD_FUSED64 and WINOGRAD_TALL and w.dtype == torch.float32 and KH == 3 and KW == 3 and Cin == 64
            and Cin < WINOGRAD_MIN_CIN and Cout % 32 == 0):
        wh = w.detach().cpu().contiguous()
        u42 = torch.empty((24, Cout, Cin), dtype=torch.float32)
        nat.check(nat.lib().msocr_winograd42_weights_host(wh.data_ptr(), Cout, Cin, u42.data_ptr()), "winograd42_weights_host")
        w._msocr_wino42_fused = u42.to(w.device)
        # the K = 64 GEMMs on the bf16 pipes pay only for the wide layer (TRBA conv0b, 64 -> 128 + pool: 2.43 -> 2.21 ms per 960 crops);
        # with 64 output channels the kernel is bound by staging and barriers either way (0.84 -> 0.86 ms) and stays exact
        v2 = os.environ.get("MSOCR_WINO_FUSED_V2", "1") != "0"  # wino42_fused64_v2_kernel (Cout % 64 == 0)
        if (SPLIT_BF16X3 if split is None else split) and ((v2 and Cout % 64 == 0) or Cout >= 128):
            w._msocr_wino42_fused_split = split_planes(u42).to(w.device)  # [3][24][Cout][64] bf16
        return w
    if not (WINOGRAD_MIN_CIN and w.dtype == torch.float32 and KH == 3 and KW == 3 and Cin >= WINOGRAD_MIN_CIN and Cin % 16 == 0
            and Cout % 32 == 0):
        return w
    wh = w.detach().cpu().contiguous()
    u = torch.empty((16, Cout, Cin), dtype=torch.float32)
    nat.check(nat.lib().msocr_winograd_weights_host(wh.data_ptr(), Cout, Cin, u.data_ptr()), "winograd_weights_host")
    w._msocr_wino = u.to(w.device)
    u42 = torch.empty((24, Cout, Cin), dtype=torch.float32)
    nat.check(nat.lib().msocr_winograd42_weights_host(wh.data_ptr(), Cout, Cin, u42.data_ptr()), "winograd42_weights_host")
    w._msocr_wino42 = u42.to(w.device)
    if (SPLIT_BF16X3 if split is None else split) and Cin % 32 == 0 and Cout % 64 == 0:
        w._msocr_wino42_split = split_planes_ktile(u42, 24, Cout).to(w.device)  # [3][24][Cin/32][Cout][32] bf16
        if WINOGRAD_SQUARE and square and Cin >= WINOGRAD_SQUARE_MIN_CIN:  # square=False: the caller keeps this layer on the tall form (half the rounding error)
            u44 = torch.empty((36, Cout, Cin), dtype=torch.float32)
            nat.check(nat.lib().msocr_winograd44_weights_host(wh.data_ptr(), Cout, Cin, u44.data_ptr()), "winograd44_weights_host")
            w._msocr_wino44_split = split_planes_ktile(u44, 36, Cout).to(w.device)  # [3][36][Cin/32][Cout][32] bf16
    return w


def _conv3x3_fused64(x, w, u42, bias, relu, residual, pool2, out):
    """Cin == 64, 3x3/1/1, f32: tall Winograd, GEMMs + output transform (+ 2x2 max-pool) in one kernel."""
    N, H, W, Cin = x.shape
    Cout = w.shape[0]
    oh, ow = (H // 2, W // 2) if pool2 else (H, W)
    if out is None:
        out = torch.empty((N, oh, ow, Cout), dtype=x.dtype, device=x.device)
    assert out.shape == (N, oh, ow, Cout) and out.dtype == x.dtype
    d = nat.ConvDesc()
    d.dtype = _dt(x)
    d.N, d.H, d.W, d.Cin = N, H, W, Cin
    d.in_sN, d.in_sH, d.in_sW = x.stride(0), x.stride(1), x.stride(2)
    d.KH, d.KW, d.stride_h, d.stride_w, d.pad_h, d.pad_w = 3, 3, 1, 1, 1, 1
    d.Ho, d.Wo, d.Cout = H, W, Cout
    d.out_ld = _pixel_dense_ld(out)
    flags = (nat.CONV_RELU if relu else 0) | (nat.CONV_POOL2 if pool2 else 0)
    if residual is not None:
        assert residual.shape == out.shape and residual.dtype == x.dtype and not pool2
        d.res_ld = _pixel_dense_ld(residual)
        flags |= nat.CONV_RESIDUAL
    d.flags = flags
    L = nat.lib()
    nbytes = L.msocr_conv3x3_winograd42_fused_workspace_bytes(ctypes.byref(d))
    if nbytes < 0:
        raise nat.NativeError(f"winograd42_fused: unsupported shape {tuple(x.shape)} * {tuple(w.shape)} pool2={pool2}")
    parts = min(N, -(-nbytes // WINO_WS_LIMIT))
    per = -(-N // parts)
    ws = _wino_workspace(nbytes if parts == 1 else (nbytes // N) * per, x.device)
    bp = bias.data_ptr() if bias is not None else None
    up = getattr(w, "_msocr_wino42_fused_split", None) if SPLIT_BF16X3 else None
    f_whole, f_gemm, name = L.msocr_conv3x3_winograd42_fused, L.msocr_winograd42_fused_gemm_output, "winograd42_fused"
    if up is not None:  # the 24 K = 64 GEMMs on the bf16 pipes with exactly split operands
        u42, f_whole, f_gemm, name = up, L.msocr_conv3x3_winograd42_fused_split, L.msocr_winograd42_fused_gemm_output_split, "winograd42_fused_split"
    what = f"msocr_conv3x3_{name} {tuple(x.shape)} * {tuple(w.shape)}"
    TH, TW = (H + 3) // 4, (W + 1) // 2
    alg = 2.0 * N * H * W * Cout * 9 * Cin
    for n0 in range(0, N, per):
        n1 = min(N, n0 + per)
        d.N = n1 - n0
        xp, rp_, op = x[n0:n1].data_ptr(), (residual[n0:n1].data_ptr() if residual is not None else None), out[n0:n1].data_ptr()
        if PROFILE is None:
            nat.check(f_whole(ctypes.byref(d), xp, u42.data_ptr(), bp, rp_, op, ws.data_ptr(), _stream()), what)
        else:
            nn, mt = n1 - n0, (n1 - n0) * TH * TW
            e = _prof_begin()
            nat.check(L.msocr_winograd42_input_transform(ctypes.byref(d), xp, ws.data_ptr(), _stream()), what)
            _prof_end(e, "wino_in", 4.0 * (nn * H * W * Cin + 24 * mt * Cin), (mt, Cin))
            e = _prof_begin()
            nat.check(f_gemm(ctypes.byref(d), u42.data_ptr(), ws.data_ptr(), bp, rp_, op, _stream()), what)
            io = x.element_size() * (nn * H * W * Cin + nn * (H * W // (4 if pool2 else 1)) * Cout * (2 if residual is not None else 1) + Cout * 9 * Cin)
            _prof_end(e, "conv_gemm", (alg * nn / N, 2.0 * 24 * mt * Cin * Cout, io), (nn * H * W, Cout, 9 * Cin, name))
    return out


def conv2d(x, w, bias, stride=(1, 1), pad=(0, 0), relu=False, residual=None, out=None, out_hw=None, alg_k=None, pool2=False):
    """x [N,H,W,Cin] (any N/H/W strides, channel stride 1), w [Cout,KH,KW,Cin], bias f32 [Cout] or None.
    alg_k: algorithmic reduction length when the packed K is padded (stem), for FLOP accounting only.
    pool2: return maxpool2x2/2 of the result (fused into the convolution where the fused Cin = 64 kernel applies, else a second
    kernel)."""
    _need_cuda(x, w, bias, residual, out)
    N, H, W, Cin = x.shape
    Cout, KH, KW, Cw = w.shape
    assert Cw == Cin and w.is_contiguous() and w.dtype == x.dtype and x.stride(3) == 1
    sh, sw = stride
    ph, pw = pad
    Ho, Wo = out_hw if out_hw else ((H + 2 * ph - KH) // sh + 1, (W + 2 * pw - KW) // sw + 1)
    u42f = getattr(w, "_msocr_wino42_fused", None)
    if (u42f is not None and WINOGRAD_FUSED64 and WINOGRAD_TALL and (sh, sw, ph, pw) == (1, 1, 1, 1) and (Ho, Wo) == (H, W)
            and _tall_pays(H) and (not pool2 or (H % 2 == 0 and W % 2 == 0 and residual is None))):
        return _conv3x3_fused64(x, w, u42f, bias, relu, residual, pool2, out)
    if pool2:
        return maxpool2d(conv2d(x, w, bias, stride, pad, relu, residual, None, out_hw, alg_k), 2, 2, 0, out=out)
    u = getattr(w, "_msocr_wino", None)
    use_wino = u is not None and (sh, sw, ph, pw) == (1, 1, 1, 1) and (Ho, Wo) == (H, W)
    if out is None:
        out = torch.empty((N, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
    assert out.shape == (N, Ho, Wo, Cout) and out.dtype == x.dtype
    d = nat.ConvDesc()
    d.dtype = _dt(x)
    d.N, d.H, d.W, d.Cin = N, H, W, Cin
    d.in_sN, d.in_sH, d.in_sW = x.stride(0), x.stride(1), x.stride(2)
    d.KH, d.KW, d.stride_h, d.stride_w, d.pad_h, d.pad_w = KH, KW, sh, sw, ph, pw
    d.Ho, d.Wo, d.Cout = Ho, Wo, Cout
    d.out_ld = _pixel_dense_ld(out)
    flags = nat.CONV_RELU if relu else 0
    rp = None
    if residual is not None:
        assert residual.shape == out.shape and residual.dtype == x.dtype
        d.res_ld = _pixel_dense_ld(residual)
        flags |= nat.CONV_RESIDUAL
        rp = residual.data_ptr()
    d.flags = flags
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == Cout and bias.is_contiguous()
    prof = PROFILE
    bp = bias.data_ptr() if bias is not None else None
    alg = 2.0 * N * Ho * Wo * Cout * (alg_k if alg_k else KH * KW * Cin)  # ALGORITHMIC direct-convolution FLOP (2 * MACs)
    if use_wino:
        L = nat.lib()
        tall = bool(WINOGRAD_TALL) and _tall_pays(H) and getattr(w, "_msocr_wino42", None) is not None
        if tall:
            u = w._msocr_wino42
            name, f_ws, whole = "winograd42", L.msocr_conv3x3_winograd42_workspace_bytes, L.msocr_conv3x3_winograd42
            st_in, st_gemm, st_out = L.msocr_winograd42_input_transform, L.msocr_winograd42_gemm, L.msocr_winograd42_output_transform
            up = getattr(w, "_msocr_wino42_split", None)
            if up is not None and SPLIT_BF16X3:  # the 24 GEMMs on the bf16 pipes with exactly split operands
                u, name, whole, st_gemm = up, "winograd42_split", L.msocr_conv3x3_winograd42_split, L.msocr_winograd42_gemm_split
            TH, TW, npts = (Ho + 3) // 4, (Wo + 1) // 2, 24
            up44 = getattr(w, "_msocr_wino44_split", None)
            if up44 is not None and SPLIT_BF16X3 and WINOGRAD_SQUARE and _square_pays(W):
                u, name, f_ws, whole = up44, "winograd44_split", L.msocr_conv3x3_winograd44_workspace_bytes, L.msocr_conv3x3_winograd44_split
                st_in, st_gemm, st_out = L.msocr_winograd44_input_transform, L.msocr_winograd44_gemm_split, L.msocr_winograd44_output_transform
                TH, TW, npts = (Ho + 3) // 4, (Wo + 3) // 4, 36
        else:
            name, f_ws, whole = "winograd", L.msocr_conv3x3_winograd_workspace_bytes, L.msocr_conv3x3_winograd
            st_in, st_gemm, st_out = L.msocr_winograd_input_transform, L.msocr_winograd_gemm, L.msocr_winograd_output_transform
            TH, TW, npts = (Ho + 1) // 2, (Wo + 1) // 2, 16
        nbytes = f_ws(ctypes.byref(d))
        if nbytes < 0:
            raise nat.NativeError(f"{name}: unsupported shape {tuple(x.shape)} * {tuple(w.shape)}")
        parts = min(N, -(-nbytes // WINO_WS_LIMIT))  # images per call such that the workspace stays under the limit
        per = -(-N // parts)
        ws = _wino_workspace(nbytes if parts == 1 else (nbytes // N) * per, x.device)
        what = f"msocr_conv3x3_{name} {tuple(x.shape)} * {tuple(w.shape)}"
        for n0 in range(0, N, per):
            n1 = min(N, n0 + per)
            d.N = n1 - n0
            xp, rp_, op = x[n0:n1].data_ptr(), (residual[n0:n1].data_ptr() if residual is not None else None), out[n0:n1].data_ptr()
            if prof is None:
                nat.check(whole(ctypes.byref(d), xp, u.data_ptr(), bp, rp_, op, ws.data_ptr(), _stream()), what)
            else:  # the same three kernels through the per-stage entry points, one event pair each
                nn, mt = n1 - n0, (n1 - n0) * TH * TW
                v_el = npts * mt * Cin   # transformed input array V, elements
                m_el = npts * mt * Cout  # transformed output array Mw
                e = _prof_begin()
                nat.check(st_in(ctypes.byref(d), xp, ws.data_ptr(), _stream()), what)
                _prof_end(e, "wino_in", 4.0 * (nn * H * W * Cin + v_el), (mt, Cin))
                e = _prof_begin()
                nat.check(st_gemm(ctypes.byref(d), u.data_ptr(), ws.data_ptr(), _stream()), what)
                io = x.element_size() * (nn * H * W * Cin + nn * Ho * Wo * Cout * (2 if residual is not None else 1) + Cout * KH * KW * Cin)
                _prof_end(e, "conv_gemm", (alg * nn / N, 2.0 * npts * mt * Cin * Cout, io), (nn * Ho * Wo, Cout, KH * KW * Cin, name))
                e = _prof_begin()
                nat.check(st_out(ctypes.byref(d), ws.data_ptr(), bp, rp_, op, _stream()), what)
                _prof_end(e, "wino_out", 4.0 * (m_el + nn * Ho * Wo * Cout * (2 if residual is not None else 1)), (mt, Cout))
        d.N = N
    else:
        wp = getattr(w, "_msocr_split", None)
        if (wp is None and SPLIT_BF16X3 and not getattr(w, "_msocr_nosplit", False) and _split_eligible(w)
                and not torch.cuda.is_current_stream_capturing()):
            wp = w._msocr_split = split_planes_ktile(w, 1, Cout)  # first use of this weight outside the Winograd path: split once, keep
        split = wp is not None and SPLIT_BF16X3 and KH * KW * Cin >= SPLIT_MIN_K and all(v % 4 == 0 for v in x.stride()[:3])
        lean = (split and (KH, KW, sh, sw, ph, pw) == (1, 1, 1, 1, 0, 0) and (Ho, Wo) == (H, W)
                and (H == 1 or x.stride(1) == W * x.stride(2)) and (N == 1 or x.stride(0) == H * W * x.stride(2)))
        e = _prof_begin()
        if split:
            fn = nat.lib().msocr_conv1x1_split if lean else nat.lib().msocr_conv2d_split
            rc = fn(ctypes.byref(d), x.data_ptr(), wp.data_ptr(), bp, rp, out.data_ptr(), _stream())
            nat.check(rc, f"msocr_conv{'1x1' if lean else '2d'}_split {tuple(x.shape)} * {tuple(w.shape)}")
        else:
            rc = nat.lib().msocr_conv2d(ctypes.byref(d), x.data_ptr(), w.data_ptr(), bp, rp, out.data_ptr(), _stream())
            nat.check(rc, f"msocr_conv2d {tuple(x.shape)} * {tuple(w.shape)}")
        io = x.element_size() * (N * H * W * Cin + N * Ho * Wo * Cout * (2 if residual is not None else 1) + Cout * KH * KW * Cin)
        _prof_end(e, "conv_gemm", (alg, 2.0 * N * Ho * Wo * Cout * KH * KW * Cin, io),
                  (N * Ho * Wo, Cout, KH * KW * Cin, "direct_split" if split else "direct"))
    return out


def normalize_u8(imgs_u8, pad_t, pad_l, Hp, Wp, mode, dtype, cpad=4):
    """imgs [N,H,W,3] u8 (device) -> [N,Hp,Wp,cpad] normalised, zero canvas (mode 0 EAST, 1 TRBA)."""
    _need_cuda(imgs_u8)
    N, H, W, C = imgs_u8.shape
    assert C == 3 and imgs_u8.dtype == torch.uint8 and imgs_u8.is_contiguous()
    out = torch.empty((N, Hp, Wp, cpad), dtype=dtype, device=imgs_u8.device)
    nat.check(nat.lib().msocr_normalize_u8(imgs_u8.data_ptr(), N, H, W, pad_t, pad_l, Hp, Wp, cpad, mode, _dt(out), out.data_ptr(),
                                           _stream()),
              "normalize_u8")
    return out


def resize_linear_u8(src, dh, dw):
    _need_cuda(src)
    N, sh, sw, C = src.shape
    assert C == 3 and src.dtype == torch.uint8 and src.is_contiguous()
    dst = torch.empty((N, dh, dw, 3), dtype=torch.uint8, device=src.device)
    nat.check(nat.lib().msocr_resize_linear_u8(src.data_ptr(), N, sh, sw, dst.data_ptr(), dh, dw, _stream()), "resize_linear_u8")
    return dst


def maxpool2d(x, k, s, p, out=None):
    _need_cuda(x, out)
    N, H, W, C = x.shape
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    if out is None:
        out = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
    e = _prof_begin()
    nat.check(nat.lib().msocr_maxpool2d(x.data_ptr(), N, H, W, C, _pixel_dense_ld(x), k, s, p, _dt(x), out.data_ptr(), Ho, Wo,
                                        _pixel_dense_ld(out), _stream()), "maxpool2d")
    _prof_end(e, "maxpool", float(x.element_size()) * N * C * (H * W + Ho * Wo))
    return out


def upsample2x_into(x, out):
    """Bilinear x2 of x [N,H,W,C] into out[..., :C] (out is [N,2H,2W,Ctot] or a channel-slice view)."""
    _need_cuda(x, out)
    N, H, W, C = x.shape
    assert out.shape[0] == N and out.shape[1] == 2 * H and out.shape[2] == 2 * W and out.dtype == x.dtype
    nat.check(nat.lib().msocr_upsample2x_bilinear(x.data_ptr(), N, H, W, C, _pixel_dense_ld(x), _dt(x), out.data_ptr(),
                                                  _pixel_dense_ld(out), _stream()), "upsample2x")
    return out


def east_head(h1, w9, b9, score=None, geo=None):
    _need_cuda(h1, w9, b9)
    N, H, W, C = h1.shape
    assert C == 32 and w9.shape == (9, 32) and w9.dtype == torch.float32
    if score is None:
        score = torch.empty((N, H, W), dtype=torch.float32, device=h1.device)
    if geo is None:
        geo = torch.empty((N, H, W, 8), dtype=torch.float32, device=h1.device)
    nat.check(nat.lib().msocr_east_head(h1.data_ptr(), N * H * W, _pixel_dense_ld(h1), _dt(h1), w9.data_ptr(), b9.data_ptr(),
                                        score.data_ptr(), geo.data_ptr(), _stream()), "east_head")
    return score, geo


def east_decode(score, geo, thresh, scale, quant, max_cand):
    """score [N,H,W] f32, geo [N,H,W,8] f32 -> cand [N,max_cand,9] f32, counts [N] int32."""
    _need_cuda(score, geo)
    N, H, W = score.shape
    assert score.dtype == torch.float32 and geo.shape == (N, H, W, 8) and score.is_contiguous() and geo.is_contiguous()
    cand = torch.empty((N, max_cand, 9), dtype=torch.float32, device=score.device)
    counts = torch.empty((N,), dtype=torch.int32, device=score.device)
    nat.check(nat.lib().msocr_east_decode(score.data_ptr(), geo.data_ptr(), N, H, W, float(thresh), float(scale), int(quant),
                                          cand.data_ptr(), counts.data_ptr(), max_cand, _stream()), "east_decode")
    return cand, counts


def east_lanms(cand, counts, iou_thr, workspace=None):
    _need_cuda(cand, counts)
    N, max_cand, _ = cand.shape
    if workspace is None:
        nbytes = nat.lib().msocr_lanms_workspace_bytes(N, max_cand)
        workspace = torch.empty((nbytes,), dtype=torch.uint8, device=cand.device)
    boxes = torch.empty_like(cand)
    nbox = torch.empty((N,), dtype=torch.int32, device=cand.device)
    nat.check(nat.lib().msocr_east_lanms(cand.data_ptr(), counts.data_ptr(), N, max_cand, float(iou_thr), boxes.data_ptr(),
                                         nbox.data_ptr(), workspace.data_ptr(), _stream()), "east_lanms")
    return boxes, nbox


def east_box_tail(boxes, nbox, expand_w, expand_h, scale_x, scale_y, axis_aligned, remove_anomalies, sigma, min_count):
    """boxes [N,max_cand,9] f32 + nbox [N] i32 (device, from east_lanms) -> (final boxes [N,max_cand,9], counts [N] i32; -1 = page
    with more than min(max_cand, 16384) boxes, to be finished by the host path).  expand / scale / contained / anomalies / axis-aligned."""
    _need_cuda(boxes, nbox)
    N, max_cand, _ = boxes.shape
    ws = torch.empty((nat.lib().msocr_east_box_tail_workspace_bytes(N, max_cand),), dtype=torch.uint8, device=boxes.device)
    out = torch.empty_like(boxes)
    n_out = torch.empty((N,), dtype=torch.int32, device=boxes.device)
    nat.check(nat.lib().msocr_east_box_tail(boxes.data_ptr(), nbox.data_ptr(), N, max_cand, float(expand_w), float(expand_h),
                                            float(scale_x), float(scale_y), int(bool(axis_aligned)), int(bool(remove_anomalies)),
                                            float(sigma), int(min_count), out.data_ptr(), n_out.data_ptr(), ws.data_ptr(), _stream()),
              "east_box_tail")
    return out, n_out


def reading_order_crops(boxes, nbox, page_hw, min_text_size, img_h, img_w, page_base=0, y_tol_ratio=0.6, x_gap_ratio=float("inf")):
    """Final boxes [N,max_cand,9] f32 + counts [N] i32 (device, from east_box_tail) -> per page, on the device: reading order of
    the words, which positions yield a crop, and the crop descriptors for crop_resize_pad (msocr_reading_order_crops).
    Returns (order [N,max_cand] i32, keep [N,max_cand] i32, desc [N,max_cand,8] i32, ncrop [N] i32; ncrop -1 = host path)."""
    _need_cuda(boxes, nbox)
    N, max_cand, _ = boxes.shape
    dev = boxes.device
    ws = torch.empty((nat.lib().msocr_reading_order_workspace_bytes(N, max_cand),), dtype=torch.uint8, device=dev)
    order = torch.empty((N, max_cand), dtype=torch.int32, device=dev)
    keep = torch.empty((N, max_cand), dtype=torch.int32, device=dev)
    desc = torch.empty((N, max_cand, 8), dtype=torch.int32, device=dev)
    ncrop = torch.empty((N,), dtype=torch.int32, device=dev)
    nat.check(nat.lib().msocr_reading_order_crops(boxes.data_ptr(), nbox.data_ptr(), N, max_cand, int(page_hw[0]), int(page_hw[1]),
                                                  int(min_text_size), int(img_h), int(img_w), float(y_tol_ratio), float(x_gap_ratio),
                                                  int(page_base), order.data_ptr(), keep.data_ptr(), desc.data_ptr(), ncrop.data_ptr(),
                                                  ws.data_ptr(), _stream()), "reading_order_crops")
    return order, keep, desc, ncrop


def nchw_to_nhwc(x_f32, dtype, out=None):
    _need_cuda(x_f32)
    N, C, H, W = x_f32.shape
    assert x_f32.dtype == torch.float32 and x_f32.is_contiguous()
    if out is None:
        out = torch.empty((N, H, W, C), dtype=dtype, device=x_f32.device)
    nat.check(nat.lib().msocr_nchw_f32_to_nhwc(x_f32.data_ptr(), N, C, H, W, _dt(out), out.data_ptr(), _pixel_dense_ld(out), _stream()),
              "nchw_to_nhwc")
    return out


def nhwc_to_nchw_f32(x):
    _need_cuda(x)
    N, H, W, C = x.shape
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
    nat.check(nat.lib().msocr_nhwc_to_nchw_f32(x.data_ptr(), N, C, H, W, _pixel_dense_ld(x), _dt(x), out.data_ptr(), _stream()),
              "nhwc_to_nchw")
    return out


# ------------------------------------------------------------------------------------------- TRBA
def se_residual(x, identity, w1, w2, out=None):
    """relu(x * sigmoid(W2 relu(W1 mean_hw(x))) + identity); x, identity [N,H,W,C] contiguous."""
    _need_cuda(x, identity, w1, w2)
    N, H, W, C = x.shape
    assert x.is_contiguous() and identity.is_contiguous() and identity.shape == x.shape and identity.dtype == x.dtype
    assert w1.shape == (C // 16, C) and w2.shape == (C, C // 16) and w1.dtype == torch.float32
    if out is None:
        out = torch.empty_like(x)
    gate = torch.empty((N, C), dtype=torch.float32, device=x.device)
    e = _prof_begin()
    nat.check(nat.lib().msocr_se_residual(x.data_ptr(), identity.data_ptr(), N, H * W, C, _dt(x), w1.data_ptr(), w2.data_ptr(),
                                          gate.data_ptr(), out.data_ptr(), _stream()), "se_residual")
    _prof_end(e, "se_residual", 3.0 * x.element_size() * N * H * W * C, (N, H * W, C))  # x + identity in, out (x re-read from cache)
    return out


def mean_over_h(x):
    """[N,H,W,C] (dtype) -> [N,W,C] f32."""
    _need_cuda(x)
    N, H, W, C = x.shape
    assert x.is_contiguous()
    out = torch.empty((N, W, C), dtype=torch.float32, device=x.device)
    nat.check(nat.lib().msocr_mean_over_h(x.data_ptr(), N, H, W, C, _dt(x), out.data_ptr(), _stream()), "mean_over_h")
    return out


def bilstm_recurrent(xproj, whh_t, B, T, H, whh_planes=None):
    """xproj [B*T, 2*4H] f32 (= [B][T][2][4H]), whh_t [2,H,H,4] f32 (gate-interleaved) -> hcat [B,T,2H] f32.
    whh_planes (H == 256): W_hh of both directions in the packed split form -> the matrix-core kernel (csrc/bilstm_mfma.hip)."""
    _need_cuda(xproj, whh_t)
    assert xproj.is_contiguous() and xproj.numel() == B * T * 8 * H and whh_t.shape == (2, H, H, 4) and whh_t.is_contiguous()
    out = torch.empty((B, T, 2 * H), dtype=torch.float32, device=xproj.device)
    e = _prof_begin()
    if (whh_planes is not None and H == 256 and SPLIT_BF16X3 and B * T * 8 * H < 2 ** 32
            and os.environ.get("MSOCR_BILSTM_MFMA", "1") != "0"):
        nat.check(nat.lib().msocr_bilstm_recurrent_split(xproj.data_ptr(), whh_planes.data_ptr(), B, T, H, out.data_ptr(), _stream()),
                  "bilstm_recurrent_split")
    else:
        nat.check(nat.lib().msocr_bilstm_recurrent(xproj.data_ptr(), whh_t.data_ptr(), B, T, H, out.data_ptr(), _stream()), "bilstm_recurrent")
    # per step per direction W_hh (4H x H f32) + B * (h + c + 4H pre-gates) * 4 B * 2
    _prof_end(e, "bilstm", (2.0 * T * (4.0 * H * H * 4 + B * (H + H + 4 * H) * 4 * 2), T), (B, T, H))
    return out


def crop_descriptors(boxes, page_ids, page_hw, img_h, img_w):
    """Host side of the device crop: clamped AABB (reference _pipeline.py:211-217) and ResizeAndPadA's size
    arithmetic (transforms.py:91-95,114-117; Python round = banker's = np.rint on the same doubles).  boxes: iterable of
    (x_min,y_min,x_max,y_max) ints.  Returns (int32 [M,8] descriptors, keep mask) — empty crops are dropped like the
    reference does.  Vectorised; `_crop_descriptors_loop` is the literal per-box form it is tested against."""
    import numpy as np
    H, W = page_hw
    b = np.asarray(boxes, dtype=np.int64).reshape(-1, 4)
    pg = np.asarray(page_ids, dtype=np.int64).reshape(-1)
    a_, b_ = np.maximum(0, b[:, 0]), np.maximum(0, b[:, 1])
    c_, d_ = np.minimum(W, b[:, 2]), np.minimum(H, b[:, 3])
    c_ = np.where(c_ < 0, np.maximum(W + c_, 0), c_)  # Python slice semantics of image[y1:y2, x1:x2] with a negative stop
    d_ = np.where(d_ < 0, np.maximum(H + d_, 0), d_)
    keep = (c_ > a_) & (d_ > b_)
    a_, b_, c_, d_, pg = a_[keep], b_[keep], c_[keep], d_[keep], pg[keep]
    h, w = d_ - b_, c_ - a_
    scale = np.minimum(img_h / np.maximum(h, 1), img_w / np.maximum(w, 1))  # float64, as Python's min of two true divisions
    nw = np.maximum(1, np.rint(w * scale).astype(np.int64))
    nh = np.maximum(1, np.rint(h * scale).astype(np.int64))
    yy = np.maximum(0, np.minimum((img_h - nh) // 2, img_h - nh))
    desc = np.stack([pg, a_, b_, c_, d_, nw, nh, yy], axis=1).astype(np.int32) if len(pg) else np.zeros((0, 8), dtype=np.int32)
    return desc, keep


def _crop_descriptors_loop(boxes, page_ids, page_hw, img_h, img_w):
    """DIAGNOSTIC, not on any product path: the literal per-box form of `crop_descriptors`, kept only as the second implementation
    of the differential CPU test (tests/test_host_cpu.py::test_vectorised_crop_descriptors_equal_the_loop)."""
    import numpy as np
    H, W = page_hw
    desc, keep = [], []
    for (x0, y0, x1, y1), pg in zip(boxes, page_ids):
        a, b = max(0, int(x0)), max(0, int(y0))
        c, d = min(W, int(x1)), min(H, int(y1))
        if c < 0:
            c = max(W + c, 0)
        if d < 0:
            d = max(H + d, 0)
        if c <= a or d <= b:
            keep.append(False)
            continue
        h, w = d - b, c - a
        scale = min(img_h / max(h, 1), img_w / max(w, 1))
        nw, nh = max(1, int(round(w * scale))), max(1, int(round(h * scale)))
        yy = max(0, min((img_h - nh) // 2, img_h - nh))
        desc.append((pg, a, b, c, d, nw, nh, yy))
        keep.append(True)
    return np.asarray(desc, dtype=np.int32).reshape(-1, 8), np.asarray(keep, dtype=bool)


def crop_resize_pad(pages_u8, desc_host, img_h, img_w, desc_dev=None):
    """pages [N,H,W,3] u8 device, desc_host int32 [M,8] (numpy) -> canvases [M,img_h,img_w,3] u8 device.
    desc_dev: the same descriptors already on the device (uploaded by the caller on another stream); desc_host may be None
    when they were produced on the device (reading_order_crops): the kernel then validates every descriptor itself."""
    _need_cuda(pages_u8, desc_dev)
    N, H, W, C = pages_u8.shape
    assert C == 3 and pages_u8.dtype == torch.uint8 and pages_u8.is_contiguous()
    hp = None
    if desc_host is not None:
        desc_host = desc_host.astype("int32", copy=False)
        hp = desc_host.ctypes.data
    M = len(desc_host) if desc_host is not None else int(desc_dev.shape[0])
    if desc_dev is None:
        desc_dev = torch.from_numpy(desc_host).to(pages_u8.device)
    else:
        assert desc_dev.dtype == torch.int32 and desc_dev.is_contiguous() and desc_dev.shape == (M, 8)
        desc_dev.record_stream(torch.cuda.current_stream())
    out = torch.empty((M, img_h, img_w, 3), dtype=torch.uint8, device=pages_u8.device)
    nat.check(nat.lib().msocr_crop_resize_pad(pages_u8.data_ptr(), N, H, W, desc_dev.data_ptr(), hp, M, img_h, img_w,
                                              out.data_ptr(), _stream()), "crop_resize_pad")
    return out
