#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA" "TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ TCC_EA0_WRREQ_64B SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmcp_$i
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcp_$i -- python3 $R/tools/gemm_probe_one.py 1572864 256 64 4 > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
for i in range(1,5):
    agg=collections.defaultdict(float); n=collections.defaultdict(int)
    for f in glob.glob(f"{R}/gpurun_out/pmcp_{i}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_igemm" in r["Kernel_Name"]:
                agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    for k in agg: print(k, agg[k]/max(n[k],1), n[k])
PY
