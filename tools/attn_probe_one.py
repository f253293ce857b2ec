"""One beam-8 decode of B crops (25 steps, no early exit) repeated: target for rocprofv3 --pmc runs (dev tool, tools/pmc_attn.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manuscript_ocr_amd.recognizers._trba.net import TrbaNet
from manuscript_ocr_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
net = TrbaNet(synth.trba_state_dict(194, 256, seed=1), 194, 256, torch.float32)
bH = torch.randn(B, 13, 256, device="cuda")
pH = torch.randn(B, 13, 256, device="cuda")
for _ in range(3):
    net.beam(bH, pH, 25, 8, 0.9, 1.7, 1, 2, None)
torch.cuda.synchronize()
