"""Dev: full beam workspace of the split-operand decode against the exact-f32 one, per step / crop slot / beam."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manuscript_ocr_amd.recognizers._trba.net import TrbaNet
from manuscript_ocr_amd import synth
B, V, S, K = 512, 194, 6, 8
net = TrbaNet(synth.trba_state_dict(V, 256, seed=1), V, 256, torch.float32)
torch.manual_seed(0)
bH = torch.randn(B, 13, 256, device="cuda")
pH = torch.randn(B, 13, 256, device="cuda")
out = {}
for mode in ("1", "0"):
    os.environ["MSOCR_BEAM_SPLIT"] = mode
    ws, fin, lp = net.beam(bH, pH, S, K, 0.9, 1.7, 1, 2, None)
    torch.cuda.synchronize()
    n = B * S * K * V
    lg = ws[: 4 * n].view(torch.float32).view(B, S, K, V).cpu().clone()
    rest = ws[4 * n: 4 * n + 8 * B * S * K].view(torch.int32).view(2, B, S, K).cpu().clone()
    out[mode] = (lg, rest)
d = (out["1"][0] - out["0"][0]).abs()
for s in range(S):
    same_bt = (out["1"][1][:, :, s] == out["0"][1][:, :, s]).all(dim=0).all(dim=-1)   # back + tok equal for the crop
    bad = (d[:, s].amax(dim=-1) > 1e-4)   # [B][K]
    rows = bad.nonzero().tolist()
    print(f"step {s}: crops with equal back/tok {int(same_bt.sum())}/{B}; (crop, beam) with |dlogit| > 1e-4: {len(rows)}")
    for b, k in rows[:24]:
        dd = d[b, s, k]
        print(f"   crop {b} (slot {b % 4}) beam {k}: max {float(dd.max()):.3e} at v={int(dd.argmax())}, cols > 1e-4: {int((dd > 1e-4).sum())}, back {int(out['1'][1][0, b, s, k])}/{int(out['0'][1][0, b, s, k])} tok {int(out['1'][1][1, b, s, k])}/{int(out['0'][1][1, b, s, k])}")
