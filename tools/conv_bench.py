"""Micro-benchmark of msocr_conv2d on the dominant shapes of the pipeline (dev tool, GPU only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manuscript_ocr_amd import ops

SHAPES = [  # N, H, W, Cin, Cout, k, stride, pad
    (960, 4, 13, 512, 512, 3, 1, 1),      # TRBA layer3/4 3x3 at the pipeline's per-group crop count
    (960, 8, 25, 256, 256, 3, 1, 1),
    (2048, 4, 13, 512, 512, 3, 1, 1),     # TRBA layer3/4 3x3 (45 launches/step)
    (2048, 8, 25, 256, 256, 3, 1, 1),     # TRBA layer2
    (2048, 16, 50, 128, 128, 3, 1, 1),    # TRBA layer1
    (512, 32, 100, 64, 128, 3, 1, 1),     # TRBA conv0b (+ max-pool with POOL=1)
    (4, 384, 512, 64, 256, 1, 1, 0),      # EAST layer1 1x1 64->256 (K=64)
    (4, 384, 512, 64, 64, 3, 1, 1),       # EAST layer1 3x3
    (4, 96, 128, 256, 256, 3, 1, 1),      # EAST layer3 3x3
    (4, 48, 64, 512, 512, 3, 1, 1),       # EAST layer4 3x3
    (4, 192, 256, 128, 128, 3, 1, 1),     # EAST layer2 3x3
    (4, 96, 128, 256, 1024, 1, 1, 0),     # EAST layer3 1x1
]

def main():
    dt = torch.float32 if (len(sys.argv) < 2 or sys.argv[1] == "fp32") else torch.bfloat16
    for (N, H, W, Cin, Cout, k, s, p) in SHAPES:
        x = torch.randn(N, H, W, Cin, device="cuda").to(dt)
        w = (torch.randn(Cout, k, k, Cin, device="cuda") * 0.05).to(dt)
        if os.environ.get("WINO", "1") == "1":
            ops.attach_winograd(w)
        b = torch.randn(Cout, device="cuda")
        pool = os.environ.get("POOL", "0") == "1" and Cin == 64 and k == 3
        out = ops.conv2d(x, w, b, (s, s), (p, p), True, pool2=pool)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 10
        e0.record()
        for _ in range(iters):
            ops.conv2d(x, w, b, (s, s), (p, p), True, out=out, pool2=pool)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        fl = 2.0 * N * Ho * Wo * Cout * k * k * Cin
        print(f"N{N} {H}x{W} Cin{Cin} Cout{Cout} k{k} {'wino' if hasattr(w, '_msocr_wino') else ('fused64' + ('+pool' if pool else '') if hasattr(w, '_msocr_wino42_fused') else 'direct' + ('+pool' if pool else ''))}: {ms:.3f} ms  "
              f"{fl / ms / 1e9:.1f} TF/s (algorithmic)")

if __name__ == "__main__":
    main()
