"""One rank's share of BASELINE.json configs[2] (128 audit proofs per step, pipelined) alone, for experiments with the
environment switches of libspp (read at load): `ENV=... python profiles/rehearsal_probe.py [batch] [steps]`."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd"))
import torch
import spp
from spp import workload
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
ctx = spp.Context(0)
work = "/tmp/spp_rehearsal"
os.makedirs(work, exist_ok=True)
rlwe_pk = json.load(open(os.path.join(ROOT, "tests", "golden", "rlwe_pk.json")))
sppc, pk = os.path.join(work, "audit.sppc"), os.path.join(work, "audit.pk")
spp.build_circuit(2, sppc, aux=list(rlwe_pk["a"]) + list(rlwe_pk["b"]))
vk = os.path.join(work, "audit.vk")
ctx.setup(sppc, b"\x09" * 32, pk, vk)
h = ctx.load_circuit(sppc, pk, 0)
dev = torch.device("cuda", 0)
rows = workload.audit_rows(ctx, rlwe_pk["a"], rlwe_pk["b"], B)
inp = torch.frombuffer(bytearray(rows), dtype=torch.uint8).to(dev)
import random
rng = random.Random(1)
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
rs = torch.frombuffer(bytearray(b"".join(rng.randrange(1, R).to_bytes(32, "big") + rng.randrange(1, R).to_bytes(32, "big") for _ in range(B))), dtype=torch.uint8).to(dev)
outs = [(torch.zeros(B * 388, dtype=torch.uint8, device=dev), torch.zeros(B * h.pw_len, dtype=torch.uint8, device=dev),
         torch.zeros(B, dtype=torch.int32, device=dev)) for _ in range(8)]
def go(i):
    pr, pw, st_ = outs[i & 7]
    h.prove_batch_device(B, inp.data_ptr(), rs.data_ptr(), pr.data_ptr(), pw.data_ptr(), st_.data_ptr())
for i in range(8):
    go(i)
h.sync(); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    go(8 + i)
h.sync(); torch.cuda.synchronize()
el = time.perf_counter() - t0
assert sum(int(o[2].abs().sum().item()) for o in outs) == 0
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("SPP_")}, "batch": B, "proofs_per_s": round(B * steps / el, 1),
                  "ms_per_step": round(el / steps * 1e3, 3)}), flush=True)
h.close(); ctx.close()
