"""Dev: the BiLSTM recurrence of the TRBA encoder, VALU kernel against the matrix-core kernel (csrc/bilstm_mfma.hip), B crops x T = 13."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manuscript_ocr_amd import ops, _native as nat
T, H = 13, 256
torch.manual_seed(0)
whh_t = (torch.randn(2, H, H, 4) * 0.05).contiguous()
n = nat.lib().msocr_attn_pack_split_elems(4 * H)
packed = torch.empty((2, n), dtype=torch.int16)
for d in (0, 1):
    assert nat.lib().msocr_attn_pack_split_host(whh_t[d].data_ptr(), 4 * H, 1, packed[d].data_ptr()) == 0
whh_d, packed_d = whh_t.cuda(), packed.cuda()
for B in (960, 1920):
    xproj = torch.randn(B * T, 8 * H, device="cuda")
    res = {}
    for name, pl in (("valu", None), ("mfma", packed_d)):
        for _ in range(3):
            out = ops.bilstm_recurrent(xproj, whh_d, B, T, H, pl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out = ops.bilstm_recurrent(xproj, whh_d, B, T, H, pl)
        e1.record()
        torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / 10, out)
    print(f"B={B}: VALU kernel {res['valu'][0]:.3f} ms, matrix-core kernel {res['mfma'][0]:.3f} ms, max |diff| {float((res['valu'][1] - res['mfma'][1]).abs().max()):.2e}")
