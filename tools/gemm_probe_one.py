"""One GEMM shape for counter runs (dev tool): python tools/gemm_probe_one.py M N K [iters]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_probe import run
M, N, K = (int(v) for v in sys.argv[1:4])
run(M, N, K, iters=int(sys.argv[4]) if len(sys.argv) > 4 else 4)
