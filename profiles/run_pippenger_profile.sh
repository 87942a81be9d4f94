#!/bin/bash
# Round-3 measurement recipe of the 2^24-point Pippenger MSM (run on the GPU box through gpurun, from the repo root):
#   bash profiles/run_pippenger_profile.sh <tag>
# kernel stats of the whole leg, then FETCH_SIZE / WRITE_SIZE (separate passes) and the SQ counters on the uniform MSM alone.
set -e
TAG=$1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o pip -- python3 profiles/pippenger_leg_only.py 5 > $OUT/${TAG}_leg.json 2> $OUT/${TAG}_leg.err
echo "stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_u -o pip -- python3 profiles/pippenger_leg_only.py 5 uniform > $OUT/${TAG}_leg_u.json 2> $OUT/${TAG}_leg_u.err
echo "uniform stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o pmc -- python3 profiles/pippenger_leg_only.py 2 uniform > $OUT/${TAG}_pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o pmc -- python3 profiles/pippenger_leg_only.py 2 uniform > $OUT/${TAG}_pmc_write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_valu -o pmc -- python3 profiles/pippenger_leg_only.py 2 uniform > $OUT/${TAG}_pmc_valu.log 2>&1
echo "valu done"
mkdir -p $OUT/${TAG}_summary
python3 profiles/summarize_pmc_valu.py $OUT/${TAG}_pmc_valu $OUT/${TAG}_summary/round3_pippenger_pmc_valu.json > /dev/null
python3 profiles/summarize_pmc.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_summary/round3_pippenger_pmc_hbm_kernels.json
cp $(find $OUT/${TAG}_stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_summary/round3_kernel_stats_pippenger.csv
cp $(find $OUT/${TAG}_stats_u -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_summary/round3_kernel_stats_pippenger_uniform.csv
cp $OUT/${TAG}_leg.json $OUT/${TAG}_summary/round3_pippenger_leg_under_rocprof.json
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_stats_u $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_valu
python3 - "$OUT/${TAG}_summary" <<'PY'
import hashlib, json, os, sys
d = sys.argv[1]
k = json.load(open(os.path.join(d, "round3_pippenger_pmc_hbm_kernels.json")))["kernels"]
per = {name.split("<")[0].replace("void ", "").replace("spp::", ""): v["hbm_bytes_per_launch_corrected"] for name, v in k.items() if "k_pip_" in name}
raw = {name.split("<")[0].replace("void ", "").replace("spp::", ""): {"FETCH_SIZE_KB": v["FETCH_SIZE"]["avg_kb"], "WRITE_SIZE_KB": v["WRITE_SIZE"]["avg_kb"]}
       for name, v in k.items() if "k_pip_" in name}
root = os.path.dirname(os.path.dirname(os.path.abspath(d)))
lib = os.path.join(root, "shielded-pool-pinocchio-solana_amd", "libspp.so")
doc = {"points": 1 << 24, "scalars": "uniform (the MSM bench.py quotes; the witness-like MSM is not in these passes)",
       "libspp_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(),
       "kernels_hbm_bytes_per_launch": per, "hbm_bytes_per_msm": sum(per.values()), "raw_counters_per_launch": raw,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 profiles/pippenger_leg_only.py 2 uniform; (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per dispatch (MI355X_MICROARCH.md, HBM)",
       "note": "k_pip_segments gathers every 64-byte base once per window: 2^28 gathers = 17.2 GB asked, FETCH_SIZE reports 64 B per gather and the prescribed doubling counts 128 B (a whole line per gather; tests/micro/gather64.hip: the counter reports the bytes asked for this pattern and cannot tell whether the other half of the line moves -- the doubled figure is the upper bound).  Algorithmic bytes of the MSM: 96 B x 2^24 = 1.61 GB"}
json.dump(doc, open(os.path.join(d, "round3_pippenger_pmc_hbm.json"), "w"), indent=1)
PY
