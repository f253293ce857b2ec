"""Dev: se_residual_kernel on the TRBA shapes of one 1920-crop sub-batch: ms and effective HBM rate (x twice + identity + out)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manuscript_ocr_amd import ops
N = 1920
for (h, w, C) in ((16, 50, 128), (8, 25, 256), (4, 13, 512)):
    x = torch.randn(N, h, w, C, device="cuda"); idt = torch.randn(N, h, w, C, device="cuda")
    w1 = torch.randn(C // 16, C, device="cuda") * 0.1; w2 = torch.randn(C, C // 16, device="cuda") * 0.3
    for _ in range(3): ops.se_residual(x, idt, w1, w2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.se_residual(x, idt, w1, w2)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{N} x {h}x{w}x{C}: {ms:.3f} ms, {3 * x.numel() * 4 / ms / 1e9:.2f} TB/s on 3 passes (x, identity, out)")
