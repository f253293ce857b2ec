"""Does a power-of-two row stride of the activation operand cost the split GEMM? (L2 channel camping probe; dev tool, GPU only)
A [M][K] f32 with row stride K + pad, both kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manuscript_ocr_amd import ops

def run(M, N, K, pad, iters=8):
    x = torch.randn(1, M, 1, K + pad, device="cuda")[..., :K]
    w = ops.attach_split(torch.randn(N, 1, 1, K, device="cuda") * 0.05, True)
    out = ops.conv2d(x, w, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv2d(x, w, None, out=out)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * M * N * K / 1e9 / (e0.elapsed_time(e1) / iters)

if __name__ == "__main__":
    for (M, N, K) in [(161280, 512, 512), (161280, 512, 4096), (161280, 256, 256)]:
        print("PP", os.environ.get("MSOCR_SPLIT_PP", "1"), f"{M}x{N}x{K}:", " ".join(f"pad{pad}: {run(M, N, K, pad):.1f}" for pad in (0, 32, 64, 16, 8)), flush=True)
