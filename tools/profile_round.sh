#!/bin/bash
# rocprofv3 profile set of the default bench (dev tool, runs on the GPU box through gpurun):
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>_{conc,ser,fetch,write,mfma}/<host>/<pid>_*.csv
# then on the CPU side:  python tools/prof_digest.py <tag> gpurun_out/<tag>_conc gpurun_out/<tag>_ser gpurun_out/<tag>_fetch \
#                                                    gpurun_out/<tag>_write gpurun_out/<tag>_mfma
# PMC counters are collected in their own passes, with --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not
# fit one pass; gpurun refuses --pmc together with the system traces).
set -e
T=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp
export TMPDIR=/tmp
rm -rf $R/gpurun_out/${T}_conc $R/gpurun_out/${T}_ser $R/gpurun_out/${T}_fetch $R/gpurun_out/${T}_write $R/gpurun_out/${T}_mfma  # one <pid>_*.csv set per directory
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_conc -- python3 $R/bench.py --no-cpu-baseline --no-live-traffic --no-secondary > $R/gpurun_out/${T}_conc.json 2> $R/gpurun_out/${T}_conc.log
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_ser -- python3 $R/bench.py --no-cpu-baseline --no-secondary --serialize-streams > $R/gpurun_out/${T}_ser.json 2> $R/gpurun_out/${T}_ser.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${T}_fetch -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 1 --warmup 1 > /dev/null 2> $R/gpurun_out/${T}_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_write -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 1 --warmup 1 > /dev/null 2> $R/gpurun_out/${T}_write.log
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${T}_mfma -- python3 $R/bench.py --no-cpu-baseline --no-roofline --serialize-streams --steps 1 --warmup 1 > /dev/null 2> $R/gpurun_out/${T}_mfma.log
echo "profiles done: $T"
