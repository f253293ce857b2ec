"""Split-operand ("bf16x3") GEMM against the exact-f32 lean GEMM, dev tool, GPU only: time, f32-equivalent TFLOP/s and the error of
both against an f64 product on a sampled block, for the 1x1 form and for the batched 24-point Winograd form."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manuscript_ocr_amd import _native as nat
from manuscript_ocr_amd import ops


def timed(fn, iters=6):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def run(M, N, K):
    x = torch.randn(1, M, 1, K, device="cuda")
    w = torch.randn(N, 1, 1, K, device="cuda") * 0.05
    ws = ops.attach_split(w.clone(), True)
    out_e = ops.conv2d(x, w, None)
    out_s = ops.conv2d(x, ws, None)
    rows = torch.randint(0, M, (256,), device="cuda")
    ref = x[0, rows, 0].double() @ w.view(N, K).double().t()
    sc = ref.abs().max().item()
    ee = (out_e[0, rows, 0].double() - ref).abs().max().item() / sc
    es = (out_s[0, rows, 0].double() - ref).abs().max().item() / sc
    te = timed(lambda: ops.conv2d(x, w, None, out=out_e))
    ts = timed(lambda: ops.conv2d(x, ws, None, out=out_s))
    fl = 2.0 * M * N * K / 1e9
    print(f"M={M} N={N} K={K}: exact {te:.3f} ms {fl / te:.1f} TF/s err {ee:.2e} | split {ts:.3f} ms {fl / ts:.1f} TF/s err {es:.2e} | x{te / ts:.2f}",
          flush=True)


def run_wino(N, H, W, C):
    x = torch.randn(N, H, W, C, device="cuda")
    w = torch.randn(C, 3, 3, C, device="cuda") * (2.0 / (9 * C)) ** 0.5
    we = ops.attach_winograd(w.clone(), False)
    wsp = ops.attach_winograd(w.clone(), True)
    oe = ops.conv2d(x, we, None, pad=(1, 1))
    os_ = ops.conv2d(x, wsp, None, pad=(1, 1))
    ref = torch.nn.functional.conv2d(x[:4].permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), padding=1).permute(0, 2, 3, 1)
    sc = ref.abs().max().item()
    ee, es = (oe[:4].double() - ref).abs().max().item() / sc, (os_[:4].double() - ref).abs().max().item() / sc
    te = timed(lambda: ops.conv2d(x, we, None, pad=(1, 1), out=oe))
    ts = timed(lambda: ops.conv2d(x, wsp, None, pad=(1, 1), out=os_))
    fl = 2.0 * N * H * W * C * C * 9 / 1e9
    print(f"wino42 N={N} {H}x{W} C={C}: exact {te:.3f} ms ({fl / te:.0f} alg TF/s) err {ee:.2e} | split {ts:.3f} ms ({fl / ts:.0f}) err {es:.2e} | x{te / ts:.2f}",
          flush=True)


if __name__ == "__main__":
    print("MSOCR_SPLIT_WPE =", os.environ.get("MSOCR_SPLIT_WPE", "3 (default)"))
    for K in (128, 256, 512, 1024, 4096):
        run(24 * 6720, 512, K)
    for K in (64, 128, 256, 512):
        run(1572864 // 2, 256, K)
    run(98304, 1024, 256)
    run(98304, 256, 1024)
    run(393216, 64, 256)
    run_wino(960, 4, 13, 512)
    run_wino(960, 8, 25, 256)
    run_wino(8, 96, 128, 256)
    run_wino(8, 48, 64, 512)
