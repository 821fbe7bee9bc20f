#!/bin/bash
# End-of-round measurement set (run on the GPU box from the repo root): PMC passes, the profiled bench and a plain bench.
#   TAG=r02d bash tools/final_round.sh
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${TAG:-r02d}
cd $R
TAG=$T bash tools/pmc_round.sh > gpurun_out/${T}_pmc.log 2>&1
cd /tmp
export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$T -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batch1 > $R/gpurun_out/${T}_prof_bench.log 2>&1
rm -f $R/gpurun_out/prof_$T/*/*kernel_trace.csv
cd $R
timeout -k 10 500 python3 bench.py > gpurun_out/${T}_bench.log 2>&1
SDMI_DUMP_SHAPES=gpurun_out/shapes_$T.txt timeout -k 10 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batch1 --no-secondary > gpurun_out/${T}_shapes.log 2>&1 || true
echo final round done
