"""GPU parity tests of the individual HIP ops, through the C ABI (libmsocr.so).

Floating-point kernels are compared with a plain PyTorch fp32 CPU reference of the same
op (tolerances stated per test); integer / index / fp64-geometry work (decode, LANMS) is
compared bit-for-bit with the oracle and the committed golden vectors.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from manuscript_ocr_amd import ops as _ops
    return _ops


def _to_nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()


def _w_khwc(w, dtype):
    return w.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, relu, residual
    (2, 17, 23, 64, 256, (1, 1), (1, 1), (0, 0), True, True),
    (1, 20, 28, 64, 64, (3, 3), (1, 1), (1, 1), True, False),
    (2, 19, 21, 128, 128, (3, 3), (2, 2), (1, 1), True, False),
    (1, 16, 24, 256, 512, (1, 1), (2, 2), (0, 0), False, False),
    (1, 24, 40, 64, 32, (3, 3), (1, 1), (1, 1), True, False),
    (3, 4, 13, 512, 512, (2, 2), (2, 1), (0, 1), True, False),
    (1, 9, 11, 2048, 512, (1, 1), (1, 1), (0, 0), True, False),
    (1, 33, 47, 32, 64, (3, 3), (1, 1), (1, 1), False, False),
    (2, 16, 50, 128, 256, (3, 3), (2, 2), (1, 1), True, False),   # TRBA layer1.0.conv1: 3x3 / 2
    (2, 8, 25, 256, 512, (1, 1), (2, 2), (0, 0), False, True),    # a strided downsample with a residual
]


@pytest.mark.parametrize("dtype,tol,split", [(torch.float32, 2e-5, 0), (torch.float32, 2e-5, 1), (torch.bfloat16, 2e-2, 0)])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_vs_torch(ops, case, dtype, tol, split, monkeypatch):
    """split = 0: the exact-f32 (or bf16) implicit-GEMM kernel; split = 1: every eligible f32 case (Cin % 32 == 0, Cout % 64 == 0)
    through the split-operand kernels — msocr_conv1x1_split for dense 1x1 / stride 1, msocr_conv2d_split (general loader: taps,
    stride, padding) for the rest — with the K threshold lifted so that short reductions are covered too."""
    monkeypatch.setattr(ops, "SPLIT_BF16X3", split)
    monkeypatch.setattr(ops, "SPLIT_MIN_K", 0)
    N, H, W, Cin, Cout, k, stride, pad, relu, use_res = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) * (2.0 / (Cin * k[0] * k[1])) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    if dtype == torch.bfloat16:  # compare on the bf16-rounded operands
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g)
        if dtype == torch.bfloat16:
            res = res.bfloat16().float()
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    ops.PROFILE = []
    out = ops.conv2d(_to_nhwc(x, dtype), _w_khwc(w, dtype), b.cuda(), stride, pad, relu,
                     _to_nhwc(res, dtype) if use_res else None)
    tag = ops.PROFILE[0][4][3]
    ops.PROFILE = None
    assert tag == ("direct_split" if (split and Cin % 32 == 0 and Cout % 64 == 0) else "direct"), tag
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= tol * max(scale, 1.0), (err, scale)


WINO_CASES = [
    # N, H, W, Cin, Cout, relu, residual     (odd H/W: partial 2x2 tiles at the bottom/right edge)
    (2, 8, 25, 256, 256, True, False),
    (3, 4, 13, 512, 512, True, True),
    (1, 17, 23, 128, 64, False, False),
    (2, 1, 7, 128, 128, True, True),
    (1, 32, 48, 256, 128, False, True),
    (5, 7, 9, 128, 384, True, True),   # odd x odd map, 3 cout blocks, tiles not a multiple of the 64-tile workgroup
]


@pytest.mark.parametrize("case", WINO_CASES)
@pytest.mark.parametrize("form", ["tall", "plain"])
def test_conv3x3_winograd_vs_direct_and_f64(ops, case, form, monkeypatch):
    """Winograd paths (ops.attach_winograd + conv2d dispatch) against an f64 convolution: same 2e-5 bound as the direct kernel, and
    the error stays within a small multiple of the direct kernel's own error against f64 (rounding order only).
    tall: F(4,3) x F(2,3) (csrc/winograd.hip wino42_*; forced on every height here, so H = 1, 7, 17 exercise its partial tiles;
    its 6-point H transform is allowed 8x the direct error, measured ~2-4x); plain: the V / Mw F(2x2) form (csrc/winograd.hip).
    Both with the exact-f32 GEMM (the split-operand GEMM of the tall form: test_winograd42_split_vs_exact_and_f64)."""
    monkeypatch.setattr(ops, "WINOGRAD_TALL", 1 if form == "tall" else 0)
    monkeypatch.setattr(ops, "_tall_pays", lambda H: True)
    monkeypatch.setattr(ops, "SPLIT_BF16X3", 0)
    err_mult = 8 if form == "tall" else 4
    N, H, W, Cin, Cout, relu, use_res = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(N, Cout, H, W, generator=g) if use_res else None
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if use_res:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    xd, rd = _to_nhwc(x, torch.float32), (_to_nhwc(res, torch.float32) if use_res else None)
    w_direct = ops.attach_split(_w_khwc(w, torch.float32), False)  # the exact-f32 direct kernel
    w_wino = ops.attach_winograd(_w_khwc(w, torch.float32))
    assert getattr(w_wino, "_msocr_wino", None) is not None and not hasattr(w_direct, "_msocr_wino")
    out_d = ops.conv2d(xd, w_direct, b.cuda(), (1, 1), (1, 1), relu, rd)
    big = torch.full((N, H, W, Cout + 32), 7.0, device="cuda")  # written into a channel slice, like the concat buffers
    ops.conv2d(xd, w_wino, b.cuda(), (1, 1), (1, 1), relu, rd, out=big[..., 32:])
    torch.cuda.synchronize()
    scale = max(ref.abs().max().item(), 1.0)
    e_d = (out_d.cpu().permute(0, 3, 1, 2).double() - ref).abs().max().item()
    e_w = (big[..., 32:].cpu().permute(0, 3, 1, 2).double() - ref).abs().max().item()
    assert e_w <= 2e-5 * scale and e_w <= err_mult * e_d + 1e-6 * scale, (e_w, e_d, scale)
    assert torch.all(big[..., :32] == 7.0)


SPLIT_1X1_CASES = [
    # M(pixels as N,H,W), Cin, Cout, relu, residual, in_ld_extra
    ((2, 9, 13), 64, 128, True, False, 0),
    ((1, 300, 1), 512, 1024, False, False, 0),      # the LSTM / linear GEMM form, M not a multiple of the tile
    ((3, 8, 25), 256, 64, True, True, 0),           # Cout = 64: the 128 x 64 tile
    ((1, 17, 5), 384, 256, True, False, 128),       # input read from a channel slice of a wider buffer (concat)
    ((2, 4, 13), 2048, 512, False, True, 0),
]


@pytest.mark.parametrize("case", SPLIT_1X1_CASES)
def test_conv1x1_split_vs_exact_and_f64(ops, case, monkeypatch):
    """Split-operand 1x1 convolution (csrc/conv_split.hip: each f32 operand = three bf16 terms, six products on the bf16 matrix
    pipes, f32 accumulate) against an f64 product: the same 2e-5 bound as the exact-f32 kernel and within 2x of that kernel's own
    error (+ 1e-6) — i.e. the three dropped cross terms (<= 2^-25 of a product) do not show."""
    monkeypatch.setattr(ops, "SPLIT_MIN_K", 0)
    (N, H, W), Cin, Cout, relu, use_res, extra = case
    g = torch.Generator().manual_seed(Cin + Cout + H)
    xw = torch.randn(N, H, W, Cin + extra, generator=g)
    x = xw[..., extra:]
    w = torch.randn(Cout, 1, 1, Cin, generator=g) * (2.0 / Cin) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(N, H, W, Cout, generator=g) if use_res else None
    ref = x.double().reshape(-1, Cin) @ w.view(Cout, Cin).double().t() + b.double()
    if use_res:
        ref = ref + res.double().reshape(-1, Cout)
    if relu:
        ref = torch.relu(ref)
    xd = xw.cuda()[..., extra:]
    rd = res.cuda() if use_res else None
    w_e = ops.attach_split(w.cuda(), False)   # exact-f32 only
    w_s = ops.attach_split(w.cuda(), True)
    assert w_s._msocr_split.shape == (3, 1, Cin // 32, Cout, 32) and w_s._msocr_split.dtype == torch.bfloat16  # K-tile-major
    assert torch.equal(ops.unsplit_planes_ktile(w_s._msocr_split).cpu().view(Cout, 1, 1, Cin), w)  # the planes add up to the weight exactly
    out_e = ops.conv2d(xd, w_e, b.cuda(), relu=relu, residual=rd)
    big = torch.full((N, H, W, Cout + 32), 7.0, device="cuda")
    ops.PROFILE = []
    ops.conv2d(xd, w_s, b.cuda(), relu=relu, residual=rd, out=big[..., 32:])
    tags = [t[4][3] for t in ops.PROFILE]
    ops.PROFILE = None
    assert tags == ["direct_split"], tags
    torch.cuda.synchronize()
    scale = max(ref.abs().max().item(), 1.0)
    e_e = (out_e.cpu().double().reshape(-1, Cout) - ref).abs().max().item()
    e_s = (big[..., 32:].cpu().double().reshape(-1, Cout) - ref).abs().max().item()
    print(f"conv1x1 split {case}: exact err {e_e / scale:.2e}, split err {e_s / scale:.2e}")
    assert e_s <= 2e-5 * scale and e_s <= 2 * e_e + 1e-6 * scale, (e_s, e_e, scale)
    assert torch.all(big[..., :32] == 7.0)


@pytest.mark.parametrize("use_res", [False, True])
def test_conv1x1_split_operand_above_4_gb_is_cut_into_row_ranges(ops, use_res):
    """the lean loaders address rows with a 32-bit byte offset from a uniform base.  A 1x1 launch whose input spans
    4 GB or more (here 4.4 M pixels x 256 channels f32 = 4.5 GB) is cut into row ranges on the host (bases advance in 64 bits,
    csrc/conv_split.hip::launch_split_any) instead of wrapping its offsets: rows sampled from the first, the middle and the LAST
    range — beyond 2^32 bytes — against an f64 product, through both kernels (no residual: conv_split_pp_kernel; with: conv_split_kernel)."""
    M, Cin, Cout = 4_400_000, 256, 128
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(1, M, 1, Cin, device="cuda", generator=g)
    assert x.numel() * 4 > 2 ** 32
    w = ops.attach_split((torch.randn(Cout, 1, 1, Cin, device="cuda", generator=g) * (2.0 / Cin) ** 0.5), True)
    b = torch.randn(Cout, device="cuda", generator=g) * 0.1
    res = torch.randn(1, M, 1, Cout, device="cuda", generator=g) if use_res else None
    out = ops.conv2d(x, w, b, relu=True, residual=res)
    rows = torch.cat([torch.arange(0, 64), torch.arange(M // 2 - 32, M // 2 + 32), torch.arange(M - 300, M)]).cuda()
    ref = x[0, rows, 0].double() @ w.view(Cout, Cin).double().t() + b.double()
    if use_res:
        ref = ref + res[0, rows, 0].double()
    ref = torch.relu(ref)
    err = (out[0, rows, 0].double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-5, err
    del x, out, res
    torch.cuda.empty_cache()


@pytest.mark.parametrize("case", [(3, 4, 13, 512, 512, False, False), (2, 8, 25, 256, 256, True, True), (1, 17, 9, 128, 192, True, False),
                                  (2, 12, 16, 160, 64, False, True)])
def test_winograd42_split_vs_exact_and_f64(ops, case, monkeypatch):
    """Tall Winograd with the 24 transform-domain GEMMs on the split-operand kernel against an f64 convolution: 2e-5 bound, within
    2x of the exact-f32 Winograd path's own error."""
    monkeypatch.setattr(ops, "_tall_pays", lambda H: True)
    monkeypatch.setattr(ops, "WINOGRAD_SQUARE", 0)   # the tall form itself; the square form has its own test below
    N, H, W, Cin, Cout, relu, use_res = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(N, Cout, H, W, generator=g) if use_res else None
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if use_res:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    xd, rd = _to_nhwc(x, torch.float32), (_to_nhwc(res, torch.float32) if use_res else None)
    w_e = ops.attach_winograd(_w_khwc(w, torch.float32), False)
    w_s = ops.attach_winograd(_w_khwc(w, torch.float32), True)
    assert not hasattr(w_e, "_msocr_wino42_split") and w_s._msocr_wino42_split.shape == (3, 24, Cin // 32, Cout, 32)
    out_e = ops.conv2d(xd, w_e, b.cuda(), (1, 1), (1, 1), relu, rd)
    ops.PROFILE = []
    out_s = ops.conv2d(xd, w_s, b.cuda(), (1, 1), (1, 1), relu, rd)
    tags = [t[4][3] for t in ops.PROFILE if t[2] == "conv_gemm"]
    ops.PROFILE = None
    assert tags == ["winograd42_split"], tags
    torch.cuda.synchronize()
    scale = max(ref.abs().max().item(), 1.0)
    e_e = (out_e.cpu().permute(0, 3, 1, 2).double() - ref).abs().max().item()
    e_s = (out_s.cpu().permute(0, 3, 1, 2).double() - ref).abs().max().item()
    print(f"winograd42 split {case}: exact err {e_e / scale:.2e}, split err {e_s / scale:.2e}")
    assert e_s <= 2e-5 * scale and e_s <= 2 * e_e + 1e-6 * scale, (e_s, e_e, scale)


@pytest.mark.parametrize("case", [(3, 4, 13, 512, 512, False, False), (2, 8, 25, 256, 256, True, True), (1, 17, 9, 128, 192, True, False),
                                  (2, 12, 16, 160, 64, False, True), (1, 96, 128, 128, 128, True, False), (5, 16, 50, 128, 128, True, True)])
def test_winograd44_square_form_vs_tall_and_f64(ops, case, monkeypatch):
    """Round 4: F(4,3) x F(4,3) on the interpolation points {0, +-3/2, +-2/3, inf} (36 points per 4 x 4 outputs, 2.25 multiplies
    and workspace words per output) against an f64 convolution: the 2e-5 bound of the tall form, and no more than 3x the tall
    form's own error on the same layer — the tall form runs its H axis on the same points since round 4, which halved ITS error;
    against the tall form on the textbook points {0, +-1, +-2} (rounds 2-3) the square form measures 1.1-1.3x, and on those points
    itself 4.7x (tools/winograd_points.py) — partial tiles on both axes, residual, ReLU, TRBA's 4 x 13 / 8 x 25 / 16 x 50 maps and an
    EAST-sized one."""
    monkeypatch.setattr(ops, "_tall_pays", lambda H: True)
    monkeypatch.setattr(ops, "WINOGRAD_SQUARE_MIN_CIN", 128)   # the kernels at every width (the product's default: Cin >= 512)
    N, H, W, Cin, Cout, relu, use_res = case
    g = torch.Generator().manual_seed(sum(case[:5]) + 1)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(N, Cout, H, W, generator=g) if use_res else None
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if use_res:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    xd, rd = _to_nhwc(x, torch.float32), (_to_nhwc(res, torch.float32) if use_res else None)
    w_s = ops.attach_winograd(_w_khwc(w, torch.float32), True)
    assert w_s._msocr_wino44_split.shape == (3, 36, Cin // 32, Cout, 32)
    outs = {}
    for square in (0, 1):
        monkeypatch.setattr(ops, "WINOGRAD_SQUARE", square)
        monkeypatch.setattr(ops, "_square_pays", lambda W_: True)
        ops.PROFILE = []
        big = torch.full((N, H, W, Cout + 32), 7.0, device="cuda")  # written into a channel slice
        ops.conv2d(xd, w_s, b.cuda(), (1, 1), (1, 1), relu, rd, out=big[..., 32:])
        tags = [t[4][3] for t in ops.PROFILE if t[2] == "conv_gemm"]
        ops.PROFILE = None
        assert tags == [("winograd44_split" if square else "winograd42_split")], tags
        assert torch.all(big[..., :32] == 7.0)
        plain = ops.conv2d(xd, w_s, b.cuda(), (1, 1), (1, 1), relu, rd)      # the one-call entry point (no profiling stages)
        assert torch.equal(plain, big[..., 32:])
        outs[square] = big[..., 32:].cpu().permute(0, 3, 1, 2).double()
    scale = max(ref.abs().max().item(), 1.0)
    e_t, e_q = (outs[0] - ref).abs().max().item(), (outs[1] - ref).abs().max().item()
    print(f"winograd44 {case}: tall err {e_t / scale:.2e}, square err {e_q / scale:.2e}, ratio {e_q / e_t:.2f}")
    assert e_q <= 2e-5 * scale and e_q <= 3.0 * e_t + 1e-7 * scale, (e_q, e_t, scale)


FUSED64_CASES = [
    # N, H, W, Cout, relu, residual, pool2     (Cin = 64)
    (3, 32, 100, 128, True, False, True),    # TRBA conv0b + MaxPool2d(2, 2)
    (41, 32, 100, 128, True, False, True),   # 16 400 tiles: 129 workgroup rows of 128 tiles, the last one partial
    (2, 30, 44, 64, False, True, False),     # residual without ReLU, partial tiles on the H axis
    (2, 16, 50, 128, True, False, True),
    (1, 4, 6, 96, True, False, True),        # one tile row, 3 cout blocks
    (2, 8, 12, 64, True, True, False),       # ResNet-50 layer1 style, residual
    (1, 7, 9, 32, False, False, False),      # odd sizes: partial tiles, tiles not a multiple of the 32-tile workgroup
    (5, 12, 11, 64, True, False, False),
]


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("case", FUSED64_CASES)
def test_conv3x3_fused64_winograd_vs_direct_and_f64(ops, case, split, monkeypatch):
    """Cin = 64 layers: tall Winograd with GEMMs + output transform (+ 2x2 max-pool) fused in one kernel (wino42_fused64_kernel)
    against an f64 convolution (+ residual, ReLU, max_pool2d): 2e-5 bound, within 8x of the direct kernel's own error; the
    pooled result also equals max-pooling the kernel's own unpooled result exactly."""
    monkeypatch.setattr(ops, "SPLIT_BF16X3", split)  # 1: the K = 64 GEMMs on the bf16 pipes (wino42_fused64_kernel<POOL, true>)
    N, H, W, Cout, relu, use_res, pool = case
    Cin = 64
    g = torch.Generator().manual_seed(sum(case[:4]) + 7)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(N, Cout, H, W, generator=g) if use_res else None
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if use_res:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    ref_full = ref
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    xd, rd = _to_nhwc(x, torch.float32), (_to_nhwc(res, torch.float32) if use_res else None)
    w_direct = ops.attach_split(_w_khwc(w, torch.float32), False)  # the exact-f32 direct kernel
    w_f = ops.attach_winograd(_w_khwc(w, torch.float32))
    assert getattr(w_f, "_msocr_wino42_fused", None) is not None and getattr(w_f, "_msocr_wino", None) is None
    if split and not hasattr(w_f, "_msocr_wino42_fused_split"):  # narrow layers keep the exact GEMMs by default: force the planes here
        w_f._msocr_wino42_fused_split = ops.split_planes(w_f._msocr_wino42_fused)
    out_d = ops.conv2d(xd, w_direct, b.cuda(), (1, 1), (1, 1), relu, rd)
    oh, ow = (H // 2, W // 2) if pool else (H, W)
    big = torch.full((N, oh, ow, Cout + 32), 7.0, device="cuda")  # written into a channel slice
    ops.conv2d(xd, w_f, b.cuda(), (1, 1), (1, 1), relu, rd, out=big[..., 32:], pool2=pool)
    full = ops.conv2d(xd, w_f, b.cuda(), (1, 1), (1, 1), relu, rd)  # same kernel, no pooling
    torch.cuda.synchronize()
    scale = max(ref_full.abs().max().item(), 1.0)
    e_d = (out_d.cpu().permute(0, 3, 1, 2).double() - ref_full).abs().max().item()
    got = big[..., 32:].cpu().permute(0, 3, 1, 2)
    e_w = (got.double() - ref).abs().max().item()
    assert e_w <= 2e-5 * scale and e_w <= 8 * e_d + 1e-6 * scale, (e_w, e_d, scale)
    assert torch.all(big[..., :32] == 7.0)
    if pool:
        assert torch.equal(got, F.max_pool2d(full.cpu().permute(0, 3, 1, 2), 2, 2))


def test_conv_pool2_unfused_paths(ops):
    """pool2=True where the fused kernel does not apply (Cin = 128 Winograd layer; Cin = 64 with an odd width): convolution followed
    by the max-pool kernel, same result as the two calls made by hand."""
    g = torch.Generator().manual_seed(3)
    for Cin, H, W in ((128, 8, 12), (64, 8, 11)):
        x = _to_nhwc(torch.randn(2, Cin, H, W, generator=g), torch.float32)
        w = ops.attach_winograd(_w_khwc(torch.randn(64, Cin, 3, 3, generator=g) * 0.05, torch.float32))
        b = torch.randn(64, generator=g).cuda()
        a = ops.conv2d(x, w, b, (1, 1), (1, 1), True, pool2=True)
        c = ops.maxpool2d(ops.conv2d(x, w, b, (1, 1), (1, 1), True), 2, 2, 0)
        torch.cuda.synchronize()
        assert a.shape == (2, H // 2, W // 2, 64) and torch.equal(a, c)


def test_winograd_weight_transform_matches_definition(ops, monkeypatch):
    """U = G g G^T evaluated in f64 (msocr_winograd_weights_host) for every (xi, nu)."""
    monkeypatch.setattr(ops, "WINOGRAD_SQUARE_MIN_CIN", 128)
    g = torch.Generator().manual_seed(9)
    w = torch.randn(64, 128, 3, 3, generator=g)
    wk = ops.attach_winograd(_w_khwc(w, torch.float32))
    G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
    exp = torch.einsum("xk,ockl,nl->xnoc", G, w.double(), G).reshape(16, 64, 128).float()
    assert torch.equal(wk._msocr_wino.cpu(), exp)
    # F(4,3) on the points {0, 3/2, -3/2, 2/3, -2/3, inf} (round 4): G[j] = [1, a, a^2] / prod_{l != j}(a_j - a_l), last row [0, 0, 1]
    pts = [0.0, 1.5, -1.5, 2 / 3, -2 / 3]
    G6 = torch.zeros(6, 3, dtype=torch.float64)
    for j, a in enumerate(pts):
        f = np.prod([a - b for l, b in enumerate(pts) if l != j])
        G6[j] = torch.tensor([1.0, a, a * a], dtype=torch.float64) / f
    G6[5, 2] = 1.0
    exp42 = torch.einsum("xk,ockl,nl->xnoc", G6, w.double(), G).reshape(24, 64, 128)
    got42 = wk._msocr_wino42.cpu()
    assert got42.shape == (24, 64, 128)
    # f64 evaluation in a different summation order than einsum's: equal after the single rounding to f32 up to 1 ulp
    assert (got42.double() - exp42).abs().max().item() <= 1.2e-7 * exp42.abs().max().item()
    u44 = torch.empty((36, 64, 128), dtype=torch.float32)
    wh = _w_khwc(w, torch.float32).cpu().contiguous()
    from manuscript_ocr_amd import _native as nat
    nat.check(nat.lib().msocr_winograd44_weights_host(wh.data_ptr(), 64, 128, u44.data_ptr()), "winograd44_weights_host")
    exp44 = torch.einsum("xk,ockl,nl->xnoc", G6, w.double(), G6).reshape(36, 64, 128)
    assert (u44.double() - exp44).abs().max().item() <= 1.2e-7 * exp44.abs().max().item()
    assert torch.equal(ops.unsplit_planes_ktile(wk._msocr_wino44_split.cpu()), u44)   # the planes the kernels read: the exact three-term split


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_concat_slices(ops, dtype):
    """Input read from / output written into channel slices of wider concat buffers."""
    g = torch.Generator().manual_seed(5)
    N, H, W = 1, 12, 20
    x = torch.randn(N, 64, H, W, generator=g)
    w = torch.randn(128, 64, 3, 3, generator=g) * 0.05
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv2d(x, w, None, padding=1)
    big_in = torch.zeros(N, H, W, 64 + 32, dtype=dtype, device="cuda")
    big_in[..., 32:] = _to_nhwc(x, dtype)
    big_out = torch.full((N, H, W, 128 + 64), 7.0, dtype=dtype, device="cuda")
    ops.conv2d(big_in[..., 32:], _w_khwc(w, dtype), None, (1, 1), (1, 1), False, None, out=big_out[..., 64:])
    torch.cuda.synchronize()
    got = big_out[..., 64:].float().cpu().permute(0, 3, 1, 2)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert (got - ref).abs().max().item() <= tol * max(ref.abs().max().item(), 1.0)
    assert torch.all(big_out[..., :64].float() == 7.0)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_stem_7x7_packed(ops, dtype, tol):
    """ResNet stem 7x7/2 p3 on C=3 via the padded-C4 canvas + packed [64][7][1][32] weights."""
    from manuscript_ocr_amd.detectors._east.net import pack_stem_weight, stem_view
    g = torch.Generator().manual_seed(9)
    N, H, W = 2, 64, 96
    img = torch.randint(0, 256, (N, H, W, 3), generator=g, dtype=torch.uint8)
    x = ((img.float() / 255.0) - 0.5) / 0.5
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.08
    b = torch.randn(64, generator=g) * 0.1
    if dtype == torch.bfloat16:
        w = w.bfloat16().float()
        xr = x.bfloat16().float()
    else:
        xr = x
    ref = F.relu(F.conv2d(xr.permute(0, 3, 1, 2), w, b, stride=2, padding=3))
    canvas = ops.normalize_u8(img.cuda(), 3, 3, H + 6, W + 6 + 2, 0, dtype)
    out = ops.conv2d(stem_view(canvas, 32, 7), pack_stem_weight(w, 32).to(dtype).cuda(), b.cuda(), (2, 2), (0, 0), True,
                     out_hw=(H // 2, W // 2))
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() <= tol * max(ref.abs().max().item(), 1.0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool_upsample_head(ops, dtype):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 18, 26, generator=g)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    xd = _to_nhwc(x, dtype)
    mp = ops.maxpool2d(xd, 3, 2, 1).float().cpu().permute(0, 3, 1, 2)
    assert torch.equal(mp, F.max_pool2d(x, 3, 2, 1))
    mp2 = ops.maxpool2d(xd, 2, 2, 0).float().cpu().permute(0, 3, 1, 2)
    assert torch.equal(mp2, F.max_pool2d(x, 2, 2))
    cat = torch.zeros(2, 36, 52, 64 + 32, dtype=dtype, device="cuda")
    ops.upsample2x_into(xd, cat)
    up = cat[..., :64].float().cpu().permute(0, 3, 1, 2)
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert (up - ref).abs().max().item() <= tol * ref.abs().max().item()
    assert torch.all(cat[..., 64:] == 0)
    h1 = torch.randn(2, 32, 10, 14, generator=g)
    if dtype == torch.bfloat16:
        h1 = h1.bfloat16().float()
    w9, b9 = torch.randn(9, 32, generator=g) * 0.3, torch.randn(9, generator=g)
    score, geo = ops.east_head(_to_nhwc(h1, dtype), w9.cuda(), b9.cuda())
    o = F.conv2d(h1, w9.view(9, 32, 1, 1), b9)
    np.testing.assert_allclose(score.cpu().numpy(), torch.sigmoid(o[:, 0]).numpy(), atol=2e-6)
    np.testing.assert_allclose(geo.cpu().numpy(), o[:, 1:].permute(0, 2, 3, 1).numpy(), atol=1e-5)


def test_normalize_and_resize_vs_oracle(ops):
    from oracle import imgproc
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(2, 37, 53, 3), dtype=np.uint8)
    can = ops.normalize_u8(torch.from_numpy(img).cuda(), 3, 3, 37 + 6, 53 + 8, 0, torch.float32).cpu().numpy()
    ref = imgproc.east_preprocess(img[0], 53, 37)[0].transpose(1, 2, 0)
    assert np.array_equal(can[0, 3:40, 3:56, :3], ref)
    assert np.all(can[:, :3] == 0) and np.all(can[:, :, :3] == 0) and np.all(can[..., 3] == 0) and np.all(can[:, :, 56:] == 0)
    can1 = ops.normalize_u8(torch.from_numpy(img).cuda(), 1, 1, 39, 60, 1, torch.float32).cpu().numpy()
    ref1 = (img[1].astype(np.float32) - np.float32(127.5)) * np.float32(1.0 / 127.5)
    assert np.array_equal(can1[1, 1:38, 1:54, :3], ref1)
    for (dh, dw) in ((64, 96), (20, 31), (37, 53), (74, 106)):
        got = ops.resize_linear_u8(torch.from_numpy(img).cuda(), dh, dw).cpu().numpy()
        for n in range(2):
            assert np.array_equal(got[n], imgproc.resize_linear_u8(img[n], dw, dh)), (dh, dw)
    big = rng.integers(0, 256, size=(1, 40, 60, 3), dtype=np.uint8)
    got = ops.resize_linear_u8(torch.from_numpy(big).cuda(), 20, 30).cpu().numpy()
    assert np.array_equal(got[0], imgproc.resize_linear_u8(big[0], 30, 20))


def _nan_eq(a, b):
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def test_decode_golden_bit_exact(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "east_post.npz"))
    score = torch.from_numpy(g["score"]).cuda()[None].contiguous()
    geo = torch.from_numpy(g["geo"]).cuda()[None].contiguous()
    for q, thr, key in ((1, 0.6, "decoded_q1"), (2, 0.6, "decoded_q2"), (4, 0.6, "decoded_q4"), (2, 0.9, "decoded_thr09_q2")):
        cand, cnt = ops.east_decode(score, geo, thr, 4.0, q, 16384)
        n = int(cnt.cpu()[0])
        exp = g[key]
        assert n == len(exp), (key, n, len(exp))
        assert np.array_equal(cand[0, :n].cpu().numpy().view(np.uint32), exp.view(np.uint32)), key
    cand, cnt = ops.east_decode(torch.zeros_like(score), geo, 0.6, 4.0, 2, 64)
    assert int(cnt.cpu()[0]) == 0
    cand, cnt = ops.east_decode(score, geo, 0.6, 4.0, 2, 100)  # overflow flagged, first rows intact
    c = int(cnt.cpu()[0])
    assert c < 0 and (c & 0x7FFFFFFF) == 100
    assert np.array_equal(cand[0].cpu().numpy().view(np.uint32), g["decoded_q2"][:100].view(np.uint32))


@pytest.mark.parametrize("name", ["page_small", "page_mid", "rot_1", "rot_50", "rot_600", "rot_rev_40"])
def test_lanms_golden_bit_exact(ops, golden_dir, name):
    g = np.load(os.path.join(golden_dir, "lanms.npz"))
    inp, exp = g[f"{name}_in"], g[f"{name}_out"]
    mc = 2048
    cand = torch.zeros(2, mc, 9, dtype=torch.float32, device="cuda")
    cand[0, :len(inp)] = torch.from_numpy(inp).cuda()
    cand[1, :len(inp)] = torch.from_numpy(inp).cuda()
    counts = torch.tensor([len(inp), 0], dtype=torch.int32, device="cuda")
    boxes, nbox = ops.east_lanms(cand, counts, 0.2)
    nb = nbox.cpu().numpy()
    assert nb[0] == len(exp) and nb[1] == 0
    got = boxes[0, :nb[0]].cpu().numpy()
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


def test_decode_lanms_pipeline_vs_oracle(ops):
    """Full-size injected maps (384x512, ~10k candidates): HIP decode+LANMS == oracle, bit for bit."""
    from manuscript_ocr_amd import synth
    from oracle import east_post as P
    from oracle import lanms as L
    N, H, W = 2, 1536, 2048
    scores, geos = [], []
    for n in range(N):
        rects = synth.synth_layout(300 + n, H, W)
        s, g_ = synth.synth_maps(rects, (H, W), (H // 4, W // 4), 300 + n)
        scores.append(s), geos.append(g_)
    score = torch.from_numpy(np.stack(scores)).cuda()
    geo = torch.from_numpy(np.stack(geos)).cuda()
    cand, cnt = ops.east_decode(score, geo, 0.6, 4.0, 2, 32768)
    boxes, nbox = ops.east_lanms(cand, cnt, 0.2)
    cnt, nbox = cnt.cpu().numpy(), nbox.cpu().numpy()
    for n in range(N):
        dec = P.decode_quads_from_maps(scores[n], geos[n], 0.6, 4.0, 2)
        assert cnt[n] == len(dec) and len(dec) > 5000
        assert np.array_equal(cand[n, :cnt[n]].cpu().numpy().view(np.uint32), dec.view(np.uint32))
        exp = L.locality_aware_nms(dec, 0.2)
        assert nbox[n] == len(exp)
        assert np.array_equal(boxes[n, :nbox[n]].cpu().numpy().view(np.uint32), exp.view(np.uint32))


@pytest.mark.parametrize("seed,n_base,max_run", [(0, 1, 3000), (1, 5, 900), (2, 60, 200), (3, 400, 12), (4, 3, 2), (5, 1, 1), (6, 2000, 3)])
def test_lanms_speculative_scan_long_and_short_runs(ops, seed, n_base, max_run):
    """Merge runs from 1 to 3000 candidates (far longer than one thread's segment, forcing multi-round carry
    fix-ups) and pages with almost no merging: HIP LANMS == oracle, bit for bit."""
    from oracle import lanms as L
    rng = np.random.default_rng(seed)
    rows = []
    x = 10.0
    for _ in range(n_base):
        w, h = rng.uniform(40, 200), rng.uniform(12, 40)
        y = rng.uniform(10, 1500)
        base = np.array([x, y, x + w, y, x + w, y + h, x, y + h])
        for _ in range(int(rng.integers(1, max_run + 1))):
            q = base + rng.normal(0, 0.4, 8)
            rows.append(np.concatenate([q, [rng.uniform(0.05, 1.0)]]))
        x += rng.uniform(0.3, 1.2) * w  # neighbours overlap in x, sometimes in y
    inp = np.asarray(rows, dtype=np.float32)
    inp = inp[rng.permutation(len(inp))]
    exp = L.locality_aware_nms(inp, 0.2)  # x0 ties (if any) resolve by index in the oracle and on the device alike
    mc = max(64, len(inp))
    cand = torch.zeros(1, mc, 9, dtype=torch.float32, device="cuda")
    cand[0, :len(inp)] = torch.from_numpy(inp).cuda()
    boxes, nbox = ops.east_lanms(cand, torch.tensor([len(inp)], dtype=torch.int32, device="cuda"), 0.2)
    nb = int(nbox.cpu()[0])
    assert nb == len(exp)
    assert np.array_equal(boxes[0, :nb].cpu().numpy().view(np.uint32), exp.view(np.uint32))


def test_decode_lanms_many_pages_stress(ops):
    """12 pages in one launch (grid-of-words maps with sprinkled noise, different layouts): every page's decode and
    LANMS output equals the oracle bit for bit; run twice to catch nondeterminism."""
    from manuscript_ocr_amd import synth
    from oracle import east_post as P
    from oracle import lanms as L
    H, W = 768, 1024
    rng = np.random.default_rng(123)
    scores, geos = [], []
    for n in range(12):
        rects = synth.synth_layout(900 + n, H, W, line_pitch=int(rng.integers(30, 60)), word_h=int(rng.integers(16, 28)))
        s, g_ = synth.synth_maps(rects, (H, W), (H // 4, W // 4), 900 + n)
        ys, xs = rng.integers(0, H // 4, 150), rng.integers(0, W // 4, 150)
        s[ys, xs] = rng.uniform(0.5, 0.99, 150).astype(np.float32)  # isolated junk candidates
        scores.append(s), geos.append(g_)
    score = torch.from_numpy(np.stack(scores)).cuda()
    geo = torch.from_numpy(np.stack(geos)).cuda()
    exp = []
    for n in range(12):
        dec = P.decode_quads_from_maps(scores[n], geos[n], 0.6, 4.0, 2)
        exp.append((dec, L.locality_aware_nms(dec, 0.2)))
    for _ in range(2):
        cand, cnt = ops.east_decode(score, geo, 0.6, 4.0, 2, 16384)
        boxes, nbox = ops.east_lanms(cand, cnt, 0.2)
        cnt_h, nbox_h = cnt.cpu().numpy(), nbox.cpu().numpy()
        for n in range(12):
            dec, out = exp[n]
            assert cnt_h[n] == len(dec) and nbox_h[n] == len(out), (n, cnt_h[n], len(dec), nbox_h[n], len(out))
            assert np.array_equal(cand[n, :cnt_h[n]].cpu().numpy().view(np.uint32), dec.view(np.uint32))
            assert np.array_equal(boxes[n, :nbox_h[n]].cpu().numpy().view(np.uint32), out.view(np.uint32)), n


def test_lanms_many_unmerged_polygons_general_nms_path(ops):
    """> 4096 merged polygons (nothing merges in phase 1): exercises the NMS path that keeps polygons in global
    memory instead of registers; equals the oracle bit for bit."""
    from oracle import lanms as L
    rng = np.random.default_rng(77)
    n = 4300
    cx, cy = rng.uniform(50, 6000, n), rng.uniform(50, 6000, n)
    w, h = rng.uniform(10, 60, n), rng.uniform(6, 20, n)
    inp = np.stack([cx - w, cy - h, cx + w, cy - h, cx + w, cy + h, cx - w, cy + h, rng.uniform(0.1, 1.0, n)], axis=1).astype(np.float32)
    exp, nm = L.locality_aware_nms(inp, 0.2, return_merged_count=True)
    assert nm > 4096
    cand = torch.from_numpy(inp).cuda()[None].contiguous()
    boxes, nbox = ops.east_lanms(cand, torch.tensor([n], dtype=torch.int32, device="cuda"), 0.2)
    nb = int(nbox.cpu()[0])
    assert nb == len(exp)
    assert np.array_equal(boxes[0, :nb].cpu().numpy().view(np.uint32), exp.view(np.uint32))


@pytest.mark.parametrize("n,bits", [(4300, "0"), (1500, "0"), (9000, "1")])
def test_lanms_nms_path_variants(ops, monkeypatch, n, bits):
    """The greedy pass of standard_nms has three implementations that must all equal the oracle bit for bit: the chip-wide IoU
    bit matrix + single-wave replay (default, up to 8192 merged polygons), and — MSOCR_LANMS_BITS=0, or above 8192 — the two
    in-kernel loops of the page kernel (polygons in registers up to 4096, in memory beyond)."""
    from oracle import lanms as L
    monkeypatch.setenv("MSOCR_LANMS_BITS", bits)
    rng = np.random.default_rng(n)
    span = 9000 if n > 5000 else 6000
    cx, cy = rng.uniform(50, span, n), rng.uniform(50, span, n)
    w, h = rng.uniform(10, 60, n), rng.uniform(6, 20, n)
    inp = np.stack([cx - w, cy - h, cx + w, cy - h, cx + w, cy + h, cx - w, cy + h, rng.uniform(0.1, 1.0, n)], axis=1).astype(np.float32)
    exp, nm = L.locality_aware_nms(inp, 0.2, return_merged_count=True)
    assert (nm > 8192) if n == 9000 else (nm > 4096) == (n == 4300)
    cand = torch.from_numpy(inp).cuda()[None].contiguous()
    boxes, nbox = ops.east_lanms(cand, torch.tensor([n], dtype=torch.int32, device="cuda"), 0.2)
    nb = int(nbox.cpu()[0])
    assert nb == len(exp) and nb < nm
    assert np.array_equal(boxes[0, :nb].cpu().numpy().view(np.uint32), exp.view(np.uint32))


def _random_quads(rng, M, trial):
    cx, cy = rng.random(M) * 1800, rng.random(M) * 1400
    w, h = rng.random(M) * 150 + 2, rng.random(M) * 40 + 2
    if trial % 3 == 0 and M > 4:  # nested boxes and a giant one
        cx[1], cy[1], w[1], h[1] = cx[0], cy[0], w[0] * 0.5, h[0] * 0.5
        w[2], h[2] = 1500, 900
    ang = (rng.random(M) - 0.5) * 0.4
    pts = np.stack([np.stack([-w / 2, -h / 2], 1), np.stack([w / 2, -h / 2], 1), np.stack([w / 2, h / 2], 1), np.stack([-w / 2, h / 2], 1)], 1)
    if trial % 5 == 0:
        pts = pts[:, ::-1]
    c, s_ = np.cos(ang), np.sin(ang)
    R = np.stack([np.stack([c, -s_], 1), np.stack([s_, c], 1)], 1)
    pts = np.einsum("mij,mkj->mki", R, pts) + np.stack([cx, cy], 1)[:, None, :]
    q = np.concatenate([pts.reshape(M, 8), rng.random((M, 1))], 1).astype(np.float32)
    if trial % 7 == 0:
        q[:, :8] = np.round(q[:, :8])
    return q


def test_east_box_tail_device_equals_oracle_tail(ops):
    """msocr_east_box_tail (expand, scale, contained-box removal, area anomalies, axis-aligned; one workgroup per page) against the
    ORACLE's restatement of infer.py:134-233 / utils.py:384-422 (oracle/east_post.py): bit-identical boxes, page by page, over
    random layouts with nested / giant / reversed / integer-coordinate quads and all parameter combinations."""
    from oracle import east_post as P
    rng = np.random.default_rng(5)
    for trial in range(14):
        counts = [int(c) for c in rng.choice([0, 1, 2, 5, 31, 32, 60, 200, 500, 1500], size=3)]
        max_cand = 2304
        kw = dict(ew=float(rng.choice([0.9, 0.0, 0.3])), eh=float(rng.choice([0.9, 0.0, 0.5])), aa=bool(trial % 2), anom=bool(trial % 4),
                  minc=int(rng.choice([30, 5])))
        ohw = (int(rng.choice([1536, 720, 4250])), int(rng.choice([2048, 1280, 5390])))
        twh = (int(rng.choice([1280, 2048])), int(rng.choice([1280, 1536])))
        boxes = np.zeros((3, max_cand, 9), dtype=np.float32)
        quads = []
        for pg, M in enumerate(counts):
            q = _random_quads(rng, M, trial + pg)
            boxes[pg, :M] = q
            quads.append(q)
        out, n_out = ops.east_box_tail(torch.from_numpy(boxes).cuda(), torch.tensor(counts, dtype=torch.int32).cuda(), kw["ew"], kw["eh"],
                                       ohw[1] / twh[0], ohw[0] / twh[1], kw["aa"], kw["anom"], 5.0, kw["minc"])
        out, n_out = out.cpu().numpy(), n_out.cpu().numpy()
        for pg, q in enumerate(quads):
            e = P.expand_boxes(q.copy(), kw["ew"], kw["eh"])
            e = P.scale_boxes_to_original(e, ohw, twh)
            e = P.remove_fully_contained_boxes(e)
            e = P.remove_area_anomalies(e, kw["anom"], 5.0, kw["minc"])
            e = P.convert_to_axis_aligned(e) if kw["aa"] else e
            assert n_out[pg] == len(e), (trial, pg, counts[pg], n_out[pg], len(e))
            assert np.array_equal(out[pg, : len(e)], e), (trial, pg)
    # a page above the 2048 boxes whose per-box arrays fit LDS (they move to the workspace), and a count above the capacity (-1)
    q = _random_quads(rng, 2300, 1)
    big = np.zeros((2, 2304, 9), dtype=np.float32)
    big[0, :2300] = q
    out, n_big = ops.east_box_tail(torch.from_numpy(big).cuda(), torch.tensor([2300, 2400], dtype=torch.int32).cuda(), 0.9, 0.9, 1.5, 1.25,
                                   True, True, 5.0, 30)
    e = P.convert_to_axis_aligned(P.remove_area_anomalies(P.remove_fully_contained_boxes(P.scale_boxes_to_original(
        P.expand_boxes(q.copy(), 0.9, 0.9), (1250, 1500), (1000, 1000))), True, 5.0, 30))
    assert int(n_big[1]) == -1 and int(n_big[0]) == len(e) and np.array_equal(out[0, :len(e)].cpu().numpy(), e)
