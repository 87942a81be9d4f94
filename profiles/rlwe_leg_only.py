"""RLWE witness leg of bench.py alone (BASELINE.json configs[3]); `python profiles/rlwe_leg_only.py [iters ...]`."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd"))
import torch, spp, bench
pk = json.load(open(os.path.join(ROOT, "tests", "golden", "rlwe_pk.json")))
ctx = spp.Context(0)
for it in ([int(a) for a in sys.argv[1:]] or [10]):
    print(json.dumps(bench.rlwe_leg(ctx, torch.device("cuda", 0), pk, iters=it)), flush=True)
