"""Times the beam-8 decode launch (all 25 steps, no early exit) for B crops, split-operand form against exact-f32 MFMA (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manuscript_ocr_amd.recognizers._trba.net import TrbaNet
from manuscript_ocr_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
net = TrbaNet(synth.trba_state_dict(194, 256, seed=1), 194, 256, torch.float32)
torch.manual_seed(0)
bH = torch.randn(B, 13, 256, device="cuda")
pH = torch.randn(B, 13, 256, device="cuda")
res = {}
for mode in ("1", "0"):
    os.environ["MSOCR_BEAM_SPLIT"] = mode
    for _ in range(2):
        ws, fin, lp = net.beam(bH, pH, 25, 8, 0.9, 1.7, 1, 2, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ws, fin, lp = net.beam(bH, pH, 25, 8, 0.9, 1.7, 1, 2, None)
    e1.record()
    torch.cuda.synchronize()
    trun = torch.full((B,), 25, dtype=torch.int32, device="cuda")
    logits, ids = net.beam_finalize(ws, B, 25, 8, trun)
    res[mode] = (e0.elapsed_time(e1) / 5, logits.cpu(), ids.cpu())
    print(f"MSOCR_BEAM_SPLIT={mode}: {res[mode][0]:.3f} ms per call (GEMM + beam kernel), B={B}")
same = (res["1"][2] == res["0"][2]).all(dim=1)
d = (res["1"][1] - res["0"][1]).abs()
print(f"rows with identical ids: {int(same.sum())}/{B}; max |dlogit| on identical rows: {float(d[same].max()):.3e}")
for t in range(25):
    dt = d[:, t, :]
    print(f"step {t:2d}: max |dlogit| {float(dt.max()):.3e} (max |logit| {float(res['0'][1][:, t].abs().max()):.2f}), rows > 1e-3: {int((dt.amax(dim=1) > 1e-3).sum())}")
