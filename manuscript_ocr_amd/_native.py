"""ctypes binding of libmsocr.so (include/msocr.h) — the only door to the HIP kernels.

There is no CPU fallback: if the shared library is missing or fails to load, every
native op raises.  PyTorch is used only to own device memory and streams; the
C ABI sees raw pointers and sizes.
"""
import ctypes
import os

import torch  # noqa: F401  -- MUST precede the dlopen below: libmsocr.so's NEEDED libamdhip64.so.7 then binds to the HIP
#                              runtime PyTorch already loaded (one runtime per process; two runtimes = launch failures)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmsocr.so")

F32, BF16 = 0, 1
CONV_RELU, CONV_RESIDUAL, CONV_POOL2 = 1, 2, 4

c_i32, c_i64, c_u32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32
c_f32, c_f64, c_vp = ctypes.c_float, ctypes.c_double, ctypes.c_void_p


class ConvDesc(ctypes.Structure):
    _fields_ = [
        ("dtype", c_i32), ("N", c_i32), ("H", c_i32), ("W", c_i32), ("Cin", c_i32),
        ("in_sN", c_i64), ("in_sH", c_i64), ("in_sW", c_i64),
        ("KH", c_i32), ("KW", c_i32), ("stride_h", c_i32), ("stride_w", c_i32), ("pad_h", c_i32), ("pad_w", c_i32),
        ("Ho", c_i32), ("Wo", c_i32), ("Cout", c_i32),
        ("out_ld", c_i64), ("res_ld", c_i64), ("flags", c_u32),
    ]


class JpegInfo(ctypes.Structure):
    _fields_ = [("width", c_i32), ("height", c_i32), ("ncomp", c_i32), ("hs", c_i32 * 3), ("vs", c_i32 * 3),
                ("blocks_w", c_i32 * 3), ("blocks_h", c_i32 * 3), ("supported", c_i32), ("coef_off", c_i64 * 3),
                ("coef_total", c_i64), ("quant", (ctypes.c_uint16 * 64) * 3)]


class AttnWeights(ctypes.Structure):
    _fields_ = [(n, c_vp) for n in ("h2h_wt", "h2h_b", "score_w", "wih_ctx_t", "wih_tok", "whh_t", "b_gates", "gen_wt", "gen_b")]


class AttnSplitWeights(ctypes.Structure):
    _fields_ = [(n, c_vp) for n in ("h2h_p", "whh_p", "gen_p")]


_SIGS = {
    "msocr_conv2d": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "msocr_conv3x3_winograd_workspace_bytes": (c_i64, [ctypes.POINTER(ConvDesc)]),
    "msocr_conv3x3_winograd": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "msocr_winograd_weights_host": (c_i32, [c_vp, c_i32, c_i32, c_vp]),
    "msocr_winograd_input_transform": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp]),
    "msocr_winograd_gemm": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp]),
    "msocr_winograd_output_transform": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp]),
    "msocr_conv3x3_winograd42_workspace_bytes": (c_i64, [ctypes.POINTER(ConvDesc)]),
    "msocr_conv3x3_winograd42": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "msocr_winograd42_weights_host": (c_i32, [c_vp, c_i32, c_i32, c_vp]),
    "msocr_winograd42_input_transform": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp]),
    "msocr_winograd42_gemm": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp]),
    "msocr_winograd42_output_transform": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp]),
    "msocr_split_bf16x3_host": (c_i32, [c_vp, c_i64, c_vp]),
    "msocr_split_bf16x3_ktile_host": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp]),
    "msocr_conv1x1_split": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "msocr_conv2d_split": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "msocr_winograd42_gemm_split": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp]),
    "msocr_conv3x3_winograd42_split": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "msocr_conv3x3_winograd44_workspace_bytes": (c_i64, [ctypes.POINTER(ConvDesc)]),
    "msocr_winograd44_weights_host": (c_i32, [c_vp, c_i32, c_i32, c_vp]),
    "msocr_winograd44_input_transform": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp]),
    "msocr_winograd44_gemm_split": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp]),
    "msocr_winograd44_output_transform": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp]),
    "msocr_conv3x3_winograd44_split": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "msocr_conv3x3_winograd42_fused_workspace_bytes": (c_i64, [ctypes.POINTER(ConvDesc)]),
    "msocr_conv3x3_winograd42_fused": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "msocr_winograd42_fused_gemm_output": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "msocr_conv3x3_winograd42_fused_split": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "msocr_winograd42_fused_gemm_output_split": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "msocr_normalize_u8": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "msocr_resize_linear_u8": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp]),
    "msocr_maxpool2d": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_i64, c_vp]),
    "msocr_upsample2x_bilinear": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_vp, c_i64, c_vp]),
    "msocr_east_head": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "msocr_east_decode": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_f32, c_i32, c_vp, c_vp, c_i32, c_vp]),
    "msocr_lanms_workspace_bytes": (c_i64, [c_i32, c_i32]),
    "msocr_east_lanms": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_f64, c_vp, c_vp, c_vp, c_vp]),
    "msocr_se_residual": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "msocr_mean_over_h": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "msocr_bilstm_recurrent": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "msocr_bilstm_recurrent_split": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "msocr_attn_greedy": (c_i32, [c_vp, c_vp, ctypes.POINTER(AttnWeights), c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                  c_vp, c_vp, c_vp]),
    "msocr_attn_greedy_hoisted": (c_i32, [c_vp, c_vp, c_vp, ctypes.POINTER(AttnWeights), ctypes.POINTER(AttnSplitWeights), c_i32, c_i32,
                                          c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "msocr_attn_beam": (c_i32, [c_vp, c_vp, ctypes.POINTER(AttnWeights), c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_f32,
                                c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "msocr_attn_pack_split_elems": (c_i64, [c_i32]),
    "msocr_attn_pack_split_host": (c_i32, [c_vp, c_i32, c_i32, c_vp]),
    "msocr_attn_beam_hoisted": (c_i32, [c_vp, c_vp, c_vp, ctypes.POINTER(AttnWeights), ctypes.POINTER(AttnSplitWeights), c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_f32,
                                        c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "msocr_attn_beam_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    "msocr_attn_beam_finalize": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "msocr_seq_confidence": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "msocr_crop_resize_pad": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "msocr_east_box_tail_workspace_bytes": (c_i64, [c_i32, c_i32]),
    "msocr_east_box_tail": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_f64, c_f64, c_f64, c_f64, c_i32, c_i32, c_f64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "msocr_east_box_tail_host": (c_i32, [c_vp, c_i32, c_f64, c_f64, c_f64, c_f64, c_i32, c_i32, c_f64, c_i32, c_vp, c_vp]),
    "msocr_reading_order_host": (c_i32, [c_vp, c_i32, c_f64, c_f64, c_vp]),
    "msocr_reading_order_workspace_bytes": (c_i64, [c_i32, c_i32]),
    "msocr_reading_order_crops": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f64, c_f64, c_i32,
                                          c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "msocr_jpeg_parse_host": (c_i32, [c_vp, c_i64, ctypes.POINTER(JpegInfo)]),
    "msocr_jpeg_entropy_decode_host": (c_i32, [c_vp, c_i64, ctypes.POINTER(JpegInfo), c_vp]),
    "msocr_jpeg_scan_desc_bytes": (c_i64, []),
    "msocr_jpeg_scan_prepare_host": (c_i64, [c_vp, c_i64, ctypes.POINTER(JpegInfo), c_i64, c_vp, c_vp, c_i64]),
    "msocr_jpeg_entropy_decode_device": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "msocr_jpeg_entropy_decode_intervals_host": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "msocr_jpeg_workspace_bytes": (c_i64, [ctypes.POINTER(JpegInfo)]),
    "msocr_jpeg_reconstruct": (c_i32, [ctypes.POINTER(JpegInfo), c_vp, c_vp, c_vp, c_vp]),
    "msocr_jpeg_reconstruct_host": (c_i32, [ctypes.POINTER(JpegInfo), c_vp, c_vp]),
    "msocr_nchw_f32_to_nhwc": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "msocr_nhwc_to_nchw_f32": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_vp, c_vp]),
    "msocr_version": (ctypes.c_char_p, []),
}

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """Load libmsocr.so (once).  Fails loudly: the product has no non-HIP path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) and os.path.exists("/opt/rocm/bin/hipcc"):
            # the in-tree build product is missing (fresh checkout): compile it once, in-tree, for gfx950.  One process per
            # GPU may get here at the same time (torchrun): serialise on a lock file, the others then find the library.
            import fcntl
            import subprocess
            try:
                with open(os.path.join(_HERE, "csrc", ".build.lock"), "w") as lock:
                    fcntl.flock(lock, fcntl.LOCK_EX)
                    if not os.path.exists(LIB_PATH):
                        subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(_HERE, "csrc"), "ARCH=gfx950"])
            except Exception as e:  # fall through to the loud failure below
                print(f"[manuscript_ocr_amd] building libmsocr.so failed: {e}")
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C manuscript_ocr_amd/csrc`). There is no CPU fallback."
            )
        L = ctypes.CDLL(LIB_PATH)
        missing = [name for name in _SIGS if not hasattr(L, name)]
        if missing:
            raise NativeError(f"{LIB_PATH} lacks symbols declared in include/msocr.h: {missing}")
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def exported_symbols():
    return list(_SIGS)


def check(rc, what):
    if rc != 0:
        raise NativeError(f"{what} failed with code {rc} (-1 bad argument/shape, -2 launch failure)")
