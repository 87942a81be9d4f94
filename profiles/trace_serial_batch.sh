#!/bin/bash
# per-kernel durations of serialised batches (one stream): bash profiles/trace_serial_batch.sh <tag> <circuit> <batch>
set -e
TAG=$1; CIRCUIT=$2; BATCH=$3
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
COMMON="--circuit $CIRCUIT --batch $BATCH --no-extras --no-single --no-cpu-baseline --no-host-leg"
SPP_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_serial -o t -- python3 bench.py $COMMON --steps 6 --warmup 2 > $OUT/${TAG}_serial_bench.json 2> $OUT/${TAG}_serial.err
find $OUT/${TAG}_serial -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null || true
