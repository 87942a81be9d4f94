#!/bin/bash
# kernel trace of small pipelined batches (the per-rank shard of BASELINE.json configs[2]): bash profiles/trace_small_batch.sh <tag> <batch> [depth]
set -e
TAG=$1; BATCH=$2; DEPTH=${3:-4}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
COMMON="--circuit audit --batch $BATCH --no-extras --no-single --no-cpu-baseline --no-host-leg"
SPP_DEPTH=$DEPTH GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o t -- python3 bench.py $COMMON --steps 24 --warmup 4 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}.err
SPP_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_serial -o t -- python3 bench.py $COMMON --steps 8 --warmup 2 > $OUT/${TAG}_bench_serial.json 2> $OUT/${TAG}_serial.err
find $OUT/${TAG}_trace $OUT/${TAG}_trace_serial -type f ! -name '*.csv' -delete 2>/dev/null || true
