"""Serialised stage and per-set MSM kernel times of one batch (distinct rows, largest tables, one stream): the quick probe used
while tuning the MSM kernels.  argv: withdraw|audit [batch] [reps].  Prints one JSON line."""
import json, os, sys, tempfile, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd"))
import torch, spp
from spp import workload
name = sys.argv[1] if len(sys.argv) > 1 else "audit"
B = int(sys.argv[2]) if len(sys.argv) > 2 else (2048 if name == "audit" else 4096)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cid = 1 if name == "withdraw" else 2
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
pk = json.load(open(os.path.join(ROOT, "tests/golden/rlwe_pk.json")))
sppc, pkp, vkp = (os.path.join(tmp, name + e) for e in (".sppc", ".pk", ".vk"))
spp.build_circuit(cid, sppc, aux=(list(pk["a"]) + list(pk["b"])) if cid == 2 else None)
ctx = spp.Context(0); ctx.setup(sppc, b"\x2a" * 32, pkp, vkp)
t0 = time.perf_counter()
h = ctx.load_circuit(sppc, pkp, int(os.environ.get("SPP_WINDOW", "0")))
load_s = time.perf_counter() - t0
rows = workload.withdraw_rows(ctx, B) if cid == 1 else workload.audit_rows(ctx, pk["a"], pk["b"], B)
rng = random.Random(1)
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
rs = b"".join(rng.randrange(R).to_bytes(32, "big") + rng.randrange(R).to_bytes(32, "big") for _ in range(B))
inp = torch.frombuffer(bytearray(rows), dtype=torch.uint8).to(dev)
rst = torch.frombuffer(bytearray(rs), dtype=torch.uint8).to(dev)
pr = torch.zeros(388 * B, dtype=torch.uint8, device=dev); pw = torch.zeros(h.pw_len * B, dtype=torch.uint8, device=dev)
st = torch.zeros(B, dtype=torch.int32, device=dev)
h.set_serial(True)
out = []
for it in range(reps + 1):
    torch.cuda.synchronize(); t = time.perf_counter()
    h.prove_batch_device(B, inp.data_ptr(), rst.data_ptr(), pr.data_ptr(), pw.data_ptr(), st.data_ptr()); h.sync()
    ms = (time.perf_counter() - t) * 1e3
    if it:
        out.append({"wall_ms": round(ms, 2), "stages": [round(x, 2) for x in h.last_timings(0)[:7]],
                    "msm_kernel_ms": dict(zip(["CB", "A", "B1", "K", "Z", "CS", "B2"], [round(x, 2) for x in h.msm_kernel_ms(0)]))})
assert int(st.abs().sum().item()) == 0
names = ["A", "B1", "K", "Z", "CB", "CS", "B2"]
print(json.dumps({"circuit": name, "batch": B, "load_s": round(load_s, 1), "windows": dict(zip(names, h.msm_windows())),
                  "table_rows": dict(zip(names, h.msm_table_rows())), "sizes": dict(zip(names, h.msm_sizes())),
                  "table_GB": round(h.table_bytes / 1e9, 1), "runs": out}))
h.close(); ctx.close()
