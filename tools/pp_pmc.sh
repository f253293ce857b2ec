#!/bin/bash
# dev: arbitrary PMC counter sets on one split GEMM per MSOCR_PP_DBG ablation.
#   gpurun: bash tools/pp_pmc.sh M N K "0 32" "CNT_A CNT_B;CNT_C CNT_D"
R=$GRAFT_REPO_ROOT
M=$1; N=$2; K=$3; DBGS=$4; SETS=$5
cd /tmp; export TMPDIR=/tmp
IFS=';' read -ra SETARR <<< "$SETS"
for d in $DBGS; do
  i=0
  for C in "${SETARR[@]}"; do
    i=$((i+1)); rm -rf $R/gpurun_out/pmcg_${d}_$i
    MSOCR_PP_DBG=$d MSOCR_SPLIT_PP=${PP:-1} rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcg_${d}_$i -- python3 $R/tools/split_probe_one.py $M $N $K 4 > /dev/null 2> $R/gpurun_out/pmcg_${d}_$i.err
  done
done
python3 - "$DBGS" ${#SETARR[@]} <<'PY'
import csv, glob, os, collections, sys
R=os.environ["GRAFT_REPO_ROOT"]
for d in sys.argv[1].split():
    out=[]
    for i in range(1, int(sys.argv[2])+1):
        agg=collections.defaultdict(float); n=collections.defaultdict(int)
        for f in glob.glob(f"{R}/gpurun_out/pmcg_{d}_{i}/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "conv_split" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
        out += [f"{k}={agg[k]/max(n[k],1):.4g}" for k in sorted(agg)]
    print("DBG", d, " ".join(out))
PY
rm -rf $R/gpurun_out/pmcg_*
