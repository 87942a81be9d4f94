#!/bin/bash
# Round-3 measurement recipe (run on the GPU box through gpurun, from the repo root):
#   bash profiles/run_round3_profile.sh <tag> <circuit> <batch>
# 1. rocprofv3 kernel stats of the default (pipelined) bench command, 2. the same with SPP_SERIAL=1 (every dispatch alone on
# the chip: the durations the roofline figure uses), 3./4. FETCH_SIZE / WRITE_SIZE passes (one counter per pass,
# MI355X_MICROARCH.md "HBM"), 5. SQ VALU counters on the serialised pipeline, 6. the summaries under profiles/ (and
# profiles/pmc_hbm_latest.json with the sha256 of the libspp.so they were taken on: bench.py takes roofline.traffic from it only
# while the hash matches).
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
mkdir -p $OUT/${TAG}_summary
python3 profiles/summarize_pmc_valu.py $OUT/${TAG}_pmc_valu $OUT/${TAG}_summary/round3_${CIRCUIT}_b${BATCH}_pmc_valu_serial.json
SPP_WRITE_LATEST=1 SPP_VALU_JSON=$OUT/${TAG}_summary/round3_${CIRCUIT}_b${BATCH}_pmc_valu_serial.json python3 profiles/summarize_pmc.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_summary/round3_${CIRCUIT}_b${BATCH}_pmc_hbm.json $CIRCUIT $BATCH 0
cp $(find $OUT/${TAG}_stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_summary/round3_kernel_stats_${CIRCUIT}_b${BATCH}.csv
cp $(find $OUT/${TAG}_stats_serial -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_summary/round3_kernel_stats_${CIRCUIT}_b${BATCH}_serial.csv
cp $OUT/${TAG}_bench_under_rocprof.json $OUT/${TAG}_summary/round3_${CIRCUIT}_b${BATCH}_bench_under_rocprof.json
cp $OUT/${TAG}_bench_under_rocprof_serial.json $OUT/${TAG}_summary/round3_${CIRCUIT}_b${BATCH}_bench_under_rocprof_serial.json
# keep only the summaries (the merge-back limit is 64 MiB)
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_stats_serial $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_valu
