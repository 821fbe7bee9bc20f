#!/bin/bash
# End-of-round measurement set (run on the GPU box from the repo root): PMC passes, the profiled bench and a plain bench.
#   TAG=r03 bash tools/final_round.sh
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${TAG:-r03}
cd $R
TAG=$T bash tools/pmc_round.sh > gpurun_out/${T}_pmc.log 2>&1
cd /tmp
export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$T -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batch1 --no-legs > $R/gpurun_out/${T}_prof_bench.log 2>&1
rm -f $R/gpurun_out/prof_$T/*/*kernel_trace.csv
cd $R
timeout -k 10 500 python3 bench.py > gpurun_out/${T}_bench.log 2>&1
SDMI_DUMP_SHAPES=gpurun_out/shapes_$T.txt timeout -k 10 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batch1 --no-secondary --no-legs > gpurun_out/${T}_shapes.log 2>&1 || true
python3 tools/summarize_pmc.py gpurun_out/pmc_${T}_fetch/*/*counter_collection.csv gpurun_out/pmc_${T}_write/*/*counter_collection.csv gpurun_out/pmc_${T}_stats/*/*kernel_stats.csv gpurun_out/${T}_traffic.json gpurun_out/pmc_${T}_shapes_fetch/*/*counter_collection.csv gpurun_out/pmc_${T}_shapes_write/*/*counter_collection.csv gpurun_out/pmc_conv_shapes.json > gpurun_out/${T}_traffic.log 2>&1 || true
cp gpurun_out/prof_$T/*/*kernel_stats.csv gpurun_out/${T}_bench_kernel_stats.csv 2>/dev/null || true
rm -rf gpurun_out/pmc_${T}_fetch gpurun_out/pmc_${T}_write gpurun_out/pmc_${T}_shapes_fetch gpurun_out/pmc_${T}_shapes_write      # (summarised above)
echo final round done
