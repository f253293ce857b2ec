#!/bin/bash
# dev: matrix-pipe busy fraction and effective clock of the split GEMM per MSOCR_PP_DBG ablation (gpurun: bash tools/pp_clock.sh M N K "0 1 2 8 10")
R=$GRAFT_REPO_ROOT
M=${1:-161280}; N=${2:-512}; K=${3:-4096}; DBGS=${4:-"0 1 2 8 10"}
cd /tmp; export TMPDIR=/tmp
for d in $DBGS; do
  rm -rf $R/gpurun_out/pmcc_$d
  MSOCR_PP_DBG=$d MSOCR_SPLIT_PP=${PP:-1} rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmcc_$d -- python3 $R/tools/split_probe_one.py $M $N $K 4 > /dev/null 2> $R/gpurun_out/pmcc_$d.err
done
python3 - "$DBGS" <<'PY'
import csv, glob, os, collections, sys
R=os.environ["GRAFT_REPO_ROOT"]
for d in sys.argv[1].split():
    agg=collections.defaultdict(float); n=collections.defaultdict(int); dur=[]
    for f in glob.glob(f"{R}/gpurun_out/pmcc_{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_split" in r["Kernel_Name"]:
                agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
                if r["Counter_Name"]=="GRBM_GUI_ACTIVE": dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    if not dur: print("DBG", d, "no data"); continue
    k=n["GRBM_GUI_ACTIVE"]; ns=sum(dur)/len(dur); cyc=agg["GRBM_GUI_ACTIVE"]/k/8
    print(f"DBG {d}: {ns/1e3:.0f} us  clock {cyc/ns:.3f} GHz  mfma_busy {agg['SQ_VALU_MFMA_BUSY_CYCLES']/k/1024/cyc:.3f}  wait_any/wave_cycles {agg['SQ_WAIT_ANY']/max(agg['SQ_WAVE_CYCLES'],1):.3f}")
PY
for d in $DBGS; do rm -rf $R/gpurun_out/pmcc_$d; done
