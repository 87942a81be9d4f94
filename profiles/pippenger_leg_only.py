"""2^24-point Pippenger leg of bench.py alone (BASELINE.json configs[4]); `python profiles/pippenger_leg_only.py [iters] [uniform]`.
With `uniform` only the uniform-scalar MSM runs (warm-up + iters launches): the form the counter passes of
run_pippenger_profile.sh are taken on, so that per-kernel averages are those of ONE distribution."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd"))
import spp, bench
ctx = spp.Context(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
if len(sys.argv) > 2 and sys.argv[2] == "uniform":
    _, ms, ms_bucket = ctx.msm_g1_pippenger_bench(1 << 24, seed=5, iters=iters)
    print(json.dumps({"points": 1 << 24, "ms_per_msm": round(ms, 3), "bucket_kernel_ms": round(ms_bucket, 3)}), flush=True)
else:
    print(json.dumps(bench.pippenger_leg(ctx, iters=iters)), flush=True)
