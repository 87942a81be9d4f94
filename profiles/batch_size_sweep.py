"""Latency and throughput of spp_prove_batch_device against the batch size (distinct rows, full-size blinding, largest tables):
shows where the small-batch paths hand over to the batch paths.  argv[1] = withdraw|audit."""
import json, os, sys, tempfile, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd"))
import torch, spp
from spp import workload
name = sys.argv[1] if len(sys.argv) > 1 else "withdraw"
cid = 1 if name == "withdraw" else 2
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
pk = json.load(open(os.path.join(ROOT, "tests/golden/rlwe_pk.json")))
sppc, pkp, vkp = (os.path.join(tmp, name + e) for e in (".sppc", ".pk", ".vk"))
spp.build_circuit(cid, sppc, aux=(list(pk["a"]) + list(pk["b"])) if cid == 2 else None)
ctx = spp.Context(0); ctx.setup(sppc, b"\x2a" * 32, pkp, vkp)
h = ctx.load_circuit(sppc, pkp, int(os.environ.get("SPP_WINDOW", "0")))
sizes = [1, 2, 4, 8, 16, 17, 32, 64, 65, 128, 256, 512, 1024, 1025, 2048, 2049] + ([4096] if cid == 1 else [])
top = max(sizes)
rows = workload.withdraw_rows(ctx, top) if cid == 1 else workload.audit_rows(ctx, pk["a"], pk["b"], top)
rng = random.Random(1)
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
rs = b"".join(rng.randrange(R).to_bytes(32, "big") + rng.randrange(R).to_bytes(32, "big") for _ in range(top))
inp = torch.frombuffer(bytearray(rows), dtype=torch.uint8).to(dev)
rst = torch.frombuffer(bytearray(rs), dtype=torch.uint8).to(dev)
pr = torch.zeros(388 * top, dtype=torch.uint8, device=dev); pw = torch.zeros(h.pw_len * top, dtype=torch.uint8, device=dev)
st = torch.zeros(top, dtype=torch.int32, device=dev)
for n in sizes:
    lat = []
    for it in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        h.prove_batch_device(n, inp.data_ptr(), rst.data_ptr(), pr.data_ptr(), pw.data_ptr(), st.data_ptr()); h.sync()
        lat.append((time.perf_counter() - t) * 1e3)
    assert int(st[:n].abs().sum().item()) == 0
    ms = sorted(lat[1:])[len(lat[1:]) // 2]
    print("%s batch %5d  %9.2f ms  %9.1f proofs/s (one batch at a time, not pipelined)  stages %s" % (
        name, n, ms, n / ms * 1e3, [round(x, 2) for x in h.last_timings(0)[:6]]), flush=True)
h.close(); ctx.close()
