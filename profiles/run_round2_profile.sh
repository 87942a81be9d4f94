#!/bin/bash
# Round-2 measurement recipe (run on the GPU box through gpurun, from the repo root):
#   bash profiles/run_round2_profile.sh <tag> <circuit> <batch>
# 1. rocprofv3 kernel stats of the default (pipelined) bench command, 2. the same with SPP_SERIAL=1 (every dispatch alone on
# the chip: the durations the roofline figure uses), 3./4. FETCH_SIZE / WRITE_SIZE passes (one counter per pass,
# MI355X_MICROARCH.md "HBM"), 5. SQ VALU counters on the serialised pipeline.
set -e
TAG=$1; CIRCUIT=$2; BATCH=$3
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
COMMON="--circuit $CIRCUIT --batch $BATCH --no-extras --no-single --no-cpu-baseline --no-host-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o stats -- python3 bench.py $COMMON --steps 10 --warmup 3 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.err
echo "stats done"
SPP_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_serial -o stats -- python3 bench.py $COMMON --steps 10 --warmup 3 > $OUT/${TAG}_bench_under_rocprof_serial.json 2> $OUT/${TAG}_stats_serial.err
echo "serial stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o pmc -- python3 bench.py $COMMON --steps 2 --warmup 1 > $OUT/${TAG}_pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o pmc -- python3 bench.py $COMMON --steps 2 --warmup 1 > $OUT/${TAG}_pmc_write.log 2>&1
echo "write done"
SPP_SERIAL=1 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_valu -o pmc -- python3 bench.py $COMMON --steps 2 --warmup 1 > $OUT/${TAG}_pmc_valu.log 2>&1
echo "valu done"
# keep only the small csv files (the merge-back limit is 64 MiB)
find $OUT/${TAG}_stats $OUT/${TAG}_stats_serial $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_valu -type f ! -name '*.csv' -delete 2>/dev/null || true
find $OUT -name '*kernel_trace.csv' -size +20M -delete 2>/dev/null || true
