"""One split-operand GEMM (M N K [iters]) repeated a few times: target for rocprofv3 --pmc runs (dev tool, tools/pmc_split.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manuscript_ocr_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 4
x = torch.randn(1, M, 1, K, device="cuda")
w = ops.attach_split(torch.randn(N, 1, 1, K, device="cuda") * 0.05, True)
out = ops.conv2d(x, w, None)
for _ in range(iters):
    ops.conv2d(x, w, None, out=out)
torch.cuda.synchronize()
