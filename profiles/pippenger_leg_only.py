"""2^24-point Pippenger leg of bench.py alone (BASELINE.json configs[4]); `python profiles/pippenger_leg_only.py [iters]`."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd"))
import spp, bench
ctx = spp.Context(0)
print(json.dumps(bench.pippenger_leg(ctx, iters=int(sys.argv[1]) if len(sys.argv) > 1 else 10)), flush=True)
