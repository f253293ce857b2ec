"""Dev: the reference-side BiLSTM binding of INTEGRATION.md section 2, run as written against torch.nn.LSTM (needs the MI355X)."""
import ctypes, torch, os, sys
_lib = ctypes.CDLL(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "manuscript_ocr_amd", "libmsocr.so"))
_lib.msocr_attn_pack_split_elems.restype = ctypes.c_int64
def _pack_whh(lstm):
    H = lstm.hidden_size
    planes = torch.empty((2, _lib.msocr_attn_pack_split_elems(4 * H)), dtype=torch.int16)
    for d, name in enumerate(("weight_hh_l0", "weight_hh_l0_reverse")):
        wt = getattr(lstm, name).detach().float().t().reshape(H, 4, H).permute(0, 2, 1).contiguous()
        assert _lib.msocr_attn_pack_split_host(ctypes.c_void_p(wt.data_ptr()), 4 * H, 1, ctypes.c_void_p(planes[d].data_ptr())) == 0
    return planes.cuda()
def _bilstm_hip(lstm, planes, x):
    B, T, _ = x.shape; H = lstm.hidden_size
    w_ih = torch.cat([lstm.weight_ih_l0, lstm.weight_ih_l0_reverse])
    b = torch.cat([lstm.bias_ih_l0 + lstm.bias_hh_l0, lstm.bias_ih_l0_reverse + lstm.bias_hh_l0_reverse])
    xproj = torch.addmm(b, x.reshape(B * T, -1), w_ih.t()).contiguous()
    out = torch.empty((B, T, 2 * H), dtype=torch.float32, device=x.device)
    rc = _lib.msocr_bilstm_recurrent_split(ctypes.c_void_p(xproj.data_ptr()), ctypes.c_void_p(planes.data_ptr()), B, T, H,
                                           ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    return out
torch.manual_seed(0)
lstm = torch.nn.LSTM(64, 256, bidirectional=True, batch_first=True)
x = torch.randn(37, 13, 64)
with torch.no_grad():
    ref, _ = lstm(x)
    planes = _pack_whh(lstm)
    lstm = lstm.cuda()
    got = _bilstm_hip(lstm, planes, x.cuda())
torch.cuda.synchronize()
print("max diff", float((got.cpu() - ref).abs().max()))
