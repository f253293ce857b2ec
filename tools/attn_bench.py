"""Micro-benchmark of the attention decode kernels (dev tool, GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from manuscript_ocr_amd.recognizers._trba.net import TrbaNet
from manuscript_ocr_amd import synth

net = TrbaNet(synth.trba_state_dict(194, 256, seed=1), 194, 256, torch.float32)
for B in (960, 2048):
    bH = torch.randn(B, 13, 256, device="cuda")
    pH = torch.randn(B, 13, 256, device="cuda")
    for name, fn in (("beam8", lambda: net.beam(bH, pH, 25, 8, 0.9, 1.7, 1, 2, None)), ("greedy", lambda: net.greedy(bH, pH, 25, 1, 2, None))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"B={B} {name}: {e0.elapsed_time(e1)/3:.3f} ms  NB={os.environ.get('MSOCR_BEAM_NB','auto')}")

# BiLSTM recurrence
from manuscript_ocr_amd import ops
for B in (960,):
    H, T = 256, 13
    xproj = torch.randn(B * T, 8 * H, device="cuda")
    whh = torch.randn(2, H, H, 4, device="cuda") * 0.05
    ops.bilstm_recurrent(xproj, whh, B, T, H); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.bilstm_recurrent(xproj, whh, B, T, H)
    e1.record(); torch.cuda.synchronize()
    print(f"B={B} bilstm: {e0.elapsed_time(e1)/5:.3f} ms")
