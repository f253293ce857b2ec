"""One Winograd convolution (960 crops, 4x13, 512->512) repeated a few times: target for rocprofv3 --pmc runs (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manuscript_ocr_amd import ops
x = torch.randn(960, 4, 13, 512, device="cuda")
w = ops.attach_winograd((torch.randn(512, 3, 3, 512, device="cuda") * 0.05))
b = torch.randn(512, device="cuda")
out = ops.conv2d(x, w, b, (1, 1), (1, 1), True)
for _ in range(5):
    ops.conv2d(x, w, b, (1, 1), (1, 1), True, out=out)
torch.cuda.synchronize()
