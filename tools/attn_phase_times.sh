#!/bin/bash
# Per-phase wall-clock of attn_beam_mfma_kernel's step loop (workgroup 0), dev tool for the GPU box:
#   gpurun -- bash tools/attn_phase_times.sh [B]
# builds a private copy of the two recogniser translation units with -DMSOCR_ATTN_TIMING and runs one beam decode through it.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
B=${1:-1024}
cd $R/manuscript_ocr_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -DMSOCR_ATTN_TIMING -shared \
  attn_beam_mfma.hip trba_kernels.hip attn_general.hip -o /tmp/libattn_timing.so
cd $R
python3 - <<PY
import ctypes, sys, numpy as np, torch
sys.path.insert(0, "$R")
from manuscript_ocr_amd import _native as nat, synth
from manuscript_ocr_amd.recognizers._trba.net import TrbaNet
L = ctypes.CDLL("/tmp/libattn_timing.so")
net = TrbaNet(synth.trba_state_dict(194, 256, seed=1), 194, 256, torch.float32)
real = nat.lib()
# route msocr_attn_beam of the product library object to the timing build for this process
for name in ("msocr_attn_beam", "msocr_attn_beam_hoisted"):
    fn = getattr(L, name); fn.restype, fn.argtypes = nat._SIGS[name]
    setattr(real, name, fn)
B = $B
bH, pH = torch.randn(B, 13, 256, device="cuda"), torch.randn(B, 13, 256, device="cuda")
for _ in range(2):
    net.beam(bH, pH, 25, 8, 0.9, 1.7, 1, 2, None)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 16))()
assert L.msocr_attn_timing_read(buf) == 0
t = np.array(buf, dtype=np.float64).reshape(64, 16)[:25, :11]
names = ["a h2h", "b score/tanh", "c softmax", "d ctx", "e gates+cell", "f generator", "g lse", "h top-k", "i bookkeeping", "j permute", "exit check"]
d = np.diff(np.concatenate([t[:-1], t[1:, :1]], axis=1), axis=1)[2:]  # phase k = stamp k+1 - stamp k; last = next step's stamp 0
us = d.mean(axis=0) / 100.0  # wall_clock64: 100 MHz
for n, v in zip(names, us):
    print(f"{n:16s} {v:7.2f} us")
print(f"{'step':16s} {us.sum():7.2f} us")
PY
