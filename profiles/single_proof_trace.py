"""One proof at a time (batch 1, 8-bit windows) -- run under `rocprofv3 --kernel-trace --output-format csv` to get the
per-dispatch timeline of the drop-in generateProof path. argv[1] = withdraw|audit."""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd"))
import torch, spp
# full-size blinding factors (what a real prover draws): the s*Ar / r*Bs1 part of the assembly depends on their length
R1 = 0x1f3a9c0de4b5a697887766554433221100ffeeddccbbaa998877665544332211 % (1 << 253)
R2 = 0x0e2d4c6b8a79685746352413021f0e0dccbbaa99887766554433221100fedcba
from spp import workload
name = sys.argv[1] if len(sys.argv) > 1 else "withdraw"
cid = 1 if name == "withdraw" else 2
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
pk = json.load(open(os.path.join(ROOT, "tests/golden/rlwe_pk.json")))
sppc, pkp, vkp = (os.path.join(tmp, name + e) for e in (".sppc", ".pk", ".vk"))
spp.build_circuit(cid, sppc, aux=(list(pk["a"]) + list(pk["b"])) if cid == 2 else None)
ctx = spp.Context(0); ctx.setup(sppc, b"\x2a" * 32, pkp, vkp)
h = ctx.load_circuit(sppc, pkp, 8)
rows = workload.withdraw_rows(ctx, 1) if cid == 1 else workload.audit_rows(ctx, pk["a"], pk["b"], 1)
inp = torch.frombuffer(bytearray(rows), dtype=torch.uint8).to(dev)
rs = torch.frombuffer(bytearray((R1).to_bytes(32, "big") + (R2).to_bytes(32, "big")), dtype=torch.uint8).to(dev)
pr = torch.zeros(388, dtype=torch.uint8, device=dev); pw = torch.zeros(h.pw_len, dtype=torch.uint8, device=dev)
st = torch.zeros(1, dtype=torch.int32, device=dev)
for _ in range(4):
    h.prove_batch_device(1, inp.data_ptr(), rs.data_ptr(), pr.data_ptr(), pw.data_ptr(), st.data_ptr()); h.sync()
print("status", int(st.cpu()[0]), "stages", [round(x, 2) for x in h.last_timings(0)[:7]])
