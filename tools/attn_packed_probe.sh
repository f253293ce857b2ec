#!/bin/bash
# dev: the packed-f32 fault of the hoisted context sum (csrc/attn_beam_mfma.hip, note at add_np / fmac_np), reproduced on purpose:
# the kernel built as the product builds it (no packed-f32 instructions: csrc/Makefile NOPK), then WITH them — the compiler pairs the
# sum into v_pk_fma_f32 ... op_sel:[0,1,0] — with and without a workgroup barrier between the sum and the bf16 MFMA loop
# (-DMSOCR_ATTN_SUM_BARRIER), against the exact-f32 kernel of the product library: rows of every beam's step-1 logits off by more
# than 1e-4.  The instruction-level reproducer is tools/microbench/pk_fma_beside_mfma.hip.   gpurun -- bash tools/attn_packed_probe.sh
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/manuscript_ocr_amd/csrc
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -shared attn_beam_mfma.hip trba_kernels.hip attn_general.hip"
/opt/rocm/bin/hipcc $F -Xclang -target-feature -Xclang -packed-fp32-ops -o /tmp/libattn_v0.so 2> >(grep -v "not a recognized feature" >&2) &
/opt/rocm/bin/hipcc $F -o /tmp/libattn_v1.so &
/opt/rocm/bin/hipcc $F -DMSOCR_ATTN_SUM_BARRIER -o /tmp/libattn_v2.so &
wait
cd $R
python3 - <<PY
import ctypes, os, sys, torch
sys.path.insert(0, "$R")
from manuscript_ocr_amd import _native as nat, synth
from manuscript_ocr_amd.recognizers._trba.net import TrbaNet
B, V, S, K = 1920, 194, 4, 8
net = TrbaNet(synth.trba_state_dict(V, 256, seed=1), V, 256, torch.float32)
real = nat.lib()
torch.manual_seed(0)
bH = torch.randn(B, 13, 256, device="cuda"); pH = torch.randn(B, 13, 256, device="cuda")
def run(mode):
    os.environ["MSOCR_BEAM_SPLIT"] = mode
    ws, fin, lp = net.beam(bH, pH, S, K, 0.9, 1.7, 1, 2, None)
    torch.cuda.synchronize()
    return ws[: 4 * B * S * K * V].view(torch.float32).view(B, S, K, V).clone()
ref = run("0")
names = {0: "product build (no packed-f32 instructions)", 1: "packed sum", 2: "packed sum + barrier before the MFMA loop"}
for v in (0, 1, 2):
    L = ctypes.CDLL(f"/tmp/libattn_v{v}.so")
    for name in ("msocr_attn_beam", "msocr_attn_beam_hoisted"):
        fn = getattr(L, name); fn.restype, fn.argtypes = nat._SIGS[name]
        setattr(real, name, fn)
    tot = 0
    slots = set()
    for rep in range(6):
        d = (run("1") - ref).abs()
        bad = (d[:, 1].amax(dim=-1) > 1e-4)
        tot += int(bad.sum())
        slots |= set((int(b) % 4, int(k)) for b, k in bad.nonzero().tolist())
    print(f"{names[v]}: {tot} of {6 * B * K} state rows off at step 1; (crop slot, beam) seen: {sorted(slots)}")
PY
