import sys, os, json, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd"))
log = open(os.path.join(ROOT, "gpurun_out", "vb.log"), "w")
def say(*a):
    print(*a, file=log, flush=True); print(*a, flush=True)
import spp
from oracle import native, circuit as C
os.makedirs('/tmp/vt', exist_ok=True)
spp.build_circuit(1, '/tmp/vt/w.sppc')
native.setup('/tmp/vt/w.sppc', b"\x07" * 32, '/tmp/vt/w.pk', '/tmp/vt/w.vk')
p = native.Prover('/tmp/vt/w.sppc', '/tmp/vt/w.pk')
row = C.withdraw_inputs(json.load(open(os.path.join(ROOT, 'tests/golden/withdraw_kat.json'))))
rc, proof, pw = p.prove(row, 3, 4)
vk = open('/tmp/vt/w.vk', 'rb').read()
say("oracle proof ready", rc)
ctx = spp.Context(0)
say("ctx ready")
for n in (1, 1, 8, 64, 256):
    t = time.time()
    bad = bytearray(proof); bad[100] ^= 1
    ps = [proof if i % 2 == 0 else bytes(bad) for i in range(n)]
    res, ms = ctx.verify_batch(vk, ps, [pw] * n, want_ms=True)
    say("n=%d wall %.3f s kernel %.1f ms ok=%s" % (n, time.time() - t, ms, res[:4]))
