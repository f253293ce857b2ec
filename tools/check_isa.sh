#!/bin/bash
# dev: counts the packed-f32 VALU instructions per translation unit as the Makefile compiles them, and the forms whose LOW result
# takes the HIGH dword of a source (op_sel:[..1..]) — the form that misbehaved beside bf16 MFMAs (csrc/Makefile).
# Expected: no such form anywhere.   bash tools/check_isa.sh   (CPU only, about two minutes)
R=$(cd "$(dirname "$0")/.." && pwd)
NOPK=$(sed -n 's/^NOPK_OBJS = //p' $R/manuscript_ocr_amd/csrc/Makefile)
bad=0
for src in $R/manuscript_ocr_amd/csrc/*.hip; do
  f=$(basename $src .hip); fl=""
  case " $NOPK " in *" $f.o "*) fl="-Xclang -target-feature -Xclang -packed-fp32-ops";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I$R/include $fl -S --cuda-device-only $src -o /tmp/isa_$f.s 2>/dev/null
  n=$(grep -c 'v_pk_[a-z]*_f32' /tmp/isa_$f.s); m=$(grep 'v_pk_[a-z]*_f32' /tmp/isa_$f.s | grep -c 'op_sel:')
  echo "$f: packed f32 instructions $n, cross-dword low-result forms $m"
  bad=$((bad + m))
done
[ $bad -eq 0 ] && echo "OK: no cross-dword low-result packed form" || { echo "FOUND $bad"; exit 1; }
