"""Timing of one split-operand GEMM shape list under the current MSOCR_PP_DBG ablation (results are wrong by construction for DBG != 0).
Dev tool, GPU only:  MSOCR_PP_DBG=2 python tools/pp_ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manuscript_ocr_amd import ops

def run(M, N, K, iters=8):
    z = os.environ.get("MSOCR_ZERO", "0")   # 1: zero activations, 2: zero weights, 3: both (DVFS probe: zero operands draw less power)
    x = torch.randn(1, M, 1, K, device="cuda") * (0.0 if z in ("1", "3") else 1.0)
    w = ops.attach_split(torch.randn(N, 1, 1, K, device="cuda") * (0.0 if z in ("2", "3") else 0.05), True)
    out = ops.conv2d(x, w, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv2d(x, w, None, out=out)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters
    return 2.0 * M * N * K / 1e9 / t

if __name__ == "__main__":
    shapes = [(161280, 512, 512), (161280, 512, 4096), (161280, 256, 256), (393216, 128, 512)]
    print("ZERO", os.environ.get("MSOCR_ZERO", "0"), "PP", os.environ.get("MSOCR_SPLIT_PP", "1"),
          " ".join(f"{M}x{N}x{K}: {run(M, N, K):.1f}" for (M, N, K) in shapes), flush=True)
