#!/bin/bash
# PMC counter groups of attn_beam_mfma_kernel (dev tool; gpurun: tools/pmc_attn.sh [B])
R=$GRAFT_REPO_ROOT
B=${1:-1024}
cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmca_$i
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmca_$i -- python3 $R/tools/attn_probe_one.py $B > /dev/null 2> $R/gpurun_out/pmca_$i.err
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
for i in range(1,6):
    agg=collections.defaultdict(float); n=collections.defaultdict(int); dur=[]
    for f in glob.glob(f"{R}/gpurun_out/pmca_{i}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_beam_mfma" in r["Kernel_Name"]:
                agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
                dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    for k in agg: print(k, agg[k]/max(n[k],1), n[k])
    if dur: print("  avg kernel ns", sum(dur)/len(dur))
PY
rm -rf $R/gpurun_out/pmca_[1-5] $R/gpurun_out/pmca_*.err
