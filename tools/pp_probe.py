"""Split-operand GEMM shapes of the pipeline through ops.conv2d (1x1 lean form) and the batched Winograd form: time, f32-equivalent
TFLOP/s, error against an f64 product on sampled rows.  Run once per kernel choice (MSOCR_SPLIT_PP=0 / 1); dev tool, GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manuscript_ocr_amd import ops


def timed(fn, iters=8):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def run(M, N, K, epi=False):
    x = torch.randn(1, M, 1, K, device="cuda")
    w = ops.attach_split(torch.randn(N, 1, 1, K, device="cuda") * 0.05, True)
    b = torch.randn(N, device="cuda") if epi else None
    res = torch.randn(1, M, 1, N, device="cuda") if epi else None
    out = ops.conv2d(x, w, b, relu=epi, residual=res)
    rows = torch.randint(0, M, (256,), device="cuda")
    rows[0], rows[1] = 0, M - 1
    ref = x[0, rows, 0].double() @ w.view(N, K).double().t()
    if epi:
        ref = torch.relu(ref + b.double() + res[0, rows, 0].double())
    err = (out[0, rows, 0].double() - ref).abs().max().item() / ref.abs().max().item()
    t = timed(lambda: ops.conv2d(x, w, b, relu=epi, residual=res, out=out))
    print(f"M={M} N={N} K={K}{' +bias+res+relu' if epi else ''}: {t:.3f} ms {2.0 * M * N * K / 1e9 / t:.1f} TF/s err {err:.2e}", flush=True)


def run_wino(N, H, W, C):
    x = torch.randn(N, H, W, C, device="cuda")
    w = ops.attach_winograd(torch.randn(C, 3, 3, C, device="cuda") * (2.0 / (9 * C)) ** 0.5, True)
    o = ops.conv2d(x, w, None, pad=(1, 1))
    ref = torch.nn.functional.conv2d(x[:4].permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), padding=1).permute(0, 2, 3, 1)
    err = (o[:4].double() - ref).abs().max().item() / ref.abs().max().item()
    t = timed(lambda: ops.conv2d(x, w, None, pad=(1, 1), out=o))
    fl = 2.0 * N * H * W * C * C * 9 / 1e9
    print(f"wino42 N={N} {H}x{W} C={C}: {t:.3f} ms ({fl / t:.0f} alg TF/s, {fl / 3 / t:.0f} executed-equivalent) err {err:.2e}", flush=True)


if __name__ == "__main__":
    print("MSOCR_SPLIT_PP =", os.environ.get("MSOCR_SPLIT_PP", "default"), "MSOCR_PP_PRIO =", os.environ.get("MSOCR_PP_PRIO", "default"))
    for K in (128, 256, 512, 1024, 4096):
        run(24 * 6720, 512, K)
    run(161280, 256, 256)
    run(98304, 1024, 256, True)
    run(98304, 256, 1024, True)
    run(393216, 128, 512, True)
    run(393216, 512, 128, True)
    run(24576, 2048, 512, True)
    run(24576, 512, 2048, True)
    run(26351, 512, 2048)
    run_wino(960, 4, 13, 512)
    run_wino(960, 8, 25, 256)
    run_wino(1920, 16, 50, 128)
    run_wino(8, 96, 128, 256)
    run_wino(8, 48, 64, 512)
