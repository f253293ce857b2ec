"""Steady-state vs short-K behaviour of the lean f32 GEMM (1x1 convolution form of msocr_conv2d), dev tool, GPU only.
For each (M, N, K): time, executed TFLOP/s.  A long K shows the K-loop's own rate; the short K of the Winograd GEMMs (128..512)
adds the per-tile prologue / epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manuscript_ocr_amd import ops

def run(M, N, K, iters=6):
    x = torch.randn(1, M, 1, K, device="cuda")
    w = torch.randn(N, 1, 1, K, device="cuda") * 0.05
    out = ops.conv2d(x, w, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv2d(x, w, None, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"M={M} N={N} K={K}: {ms:.3f} ms {2.0 * M * N * K / ms / 1e9:.1f} TF/s", flush=True)

if __name__ == "__main__":
    for K in (128, 256, 512, 1024, 4096):
        run(16 * 53248, 512, K)      # the 16 Winograd GEMMs of a 960-crop C=512 layer, as one GEMM
    for K in (128, 256, 512, 4096):
        run(16 * 49152, 256, K)
    run(98304, 128, 8192)
    run(768 * 128, 128, 8192)        # exactly one wave of workgroups (768 slots)
    run(768 * 128 * 4, 128, 2048)
