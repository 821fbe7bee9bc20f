#!/bin/bash
# PMC passes behind profiles/rNN_traffic.json and profiles/rNN_attn_pmc.json (run on the GPU box from the repo root).
# TAG=<name> picks the output directory prefix (gpurun_out/pmc_<TAG>_*).
# Every pass is its own short rocprofv3 run of <= 8 SQ / <= 4 TCC counters; any failure ends the script.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
if [ -z "$SKIP_BENCH" ]; then
BENCH="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batch1 --no-secondary --no-legs"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${TAG:-r03}_fetch -- $BENCH > $O/pmc_${TAG:-r03}_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${TAG:-r03}_write -- $BENCH > $O/pmc_${TAG:-r03}_write.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pmc_${TAG:-r03}_stats -- $BENCH > $O/pmc_${TAG:-r03}_stats.log 2>&1
rm -f $O/pmc_${TAG:-r03}_stats/*/*kernel_trace.csv
fi
SH="python3 $R/tools/pmc_conv_shapes.py"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${TAG:-r03}_shapes_fetch -- $SH > $O/pmc_${TAG:-r03}_shapes_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${TAG:-r03}_shapes_write -- $SH > $O/pmc_${TAG:-r03}_shapes_write.log 2>&1
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY"
P2="SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
for N in 4096 1024; do
  A="python3 $R/tools/one_attn.py 16 5 $N $N 3"
  timeout -s KILL 90 rocprofv3 --pmc $P1 --output-format csv -d $O/pmc_${TAG:-r03}_attn${N}_1 -- $A > $O/pmc_${TAG:-r03}_attn${N}_1.log 2>&1
  timeout -s KILL 90 rocprofv3 --pmc $P2 --output-format csv -d $O/pmc_${TAG:-r03}_attn${N}_2 -- $A > $O/pmc_${TAG:-r03}_attn${N}_2.log 2>&1
done
echo pmc passes done
