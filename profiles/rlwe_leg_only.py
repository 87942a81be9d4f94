import json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "shielded-pool-pinocchio-solana_amd"))
import torch, spp, bench
pk = json.load(open("tests/golden/rlwe_pk.json"))
ctx = spp.Context(0)
print(json.dumps(bench.rlwe_leg(ctx, torch.device("cuda", 0), pk)))
