"""The TypeScript-side boundary: N-API addon + CommonJS twin of client/proof.helper.ts and generate-proof-hex.ts.
Runs under node (v12 in this image); skipped when node or its headers are absent."""
import json
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

NODE_DIR = os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd", "node")
pytestmark = pytest.mark.skipif(shutil.which("node") is None or not os.path.exists("/usr/include/node/node_api.h"),
                                reason="node toolchain not present")


@pytest.fixture(scope="module")
def addon():
    subprocess.run(["make", "-C", NODE_DIR, "-s"], check=True)
    return os.path.join(NODE_DIR, "proof.helper.js")


def _node(script):
    return subprocess.run(["node", "-e", script], capture_output=True, text=True)


def test_addon_exports_and_prover_toml(addon, withdraw_kat, tmp_path):
    from spp.proof_helper import ShieldedPoolInputs, prover_toml
    k = withdraw_kat
    fields = ("root", "nullifier", "recipient", "amount", "wa_commitment", "secret_key", "owner_x", "owner_y", "randomness", "index", "siblings")
    script = """
      const h = require(%s);
      const inputs = %s;
      process.stdout.write(JSON.stringify({
        fns: ['init','buildCircuit','setup','loadCircuit','circuitInfo','proveBatch','verify','verifyBatch','version'].map(n => typeof h.addon[n]),
        version: h.addon.version(), toml: h.proverToml(inputs), n: h.addon.buildCircuit(1, %s),
        amount: h.toField32(inputs.amount).toString('hex') }));
    """ % (json.dumps(addon), json.dumps({f: k[f] for f in fields}), json.dumps(str(tmp_path / "w.sppc")))
    r = _node(script)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out["fns"] == ["function"] * 9 and out["version"].startswith("libspp")
    assert out["toml"] == prover_toml(ShieldedPoolInputs(**{f: k[f] for f in fields}))     # proof.helper.ts:32-50
    assert out["n"] > 5000 and int(out["amount"], 16) == k["amount"]


def test_addon_verify_on_host(addon, withdraw_artifacts, withdraw_kat):
    """addon.verify = `sunspot verify` (prove_linux.sh:86-87): accepts an oracle proof, rejects a flipped byte; no GPU."""
    from oracle import native, circuit as C
    p = native.Prover(withdraw_artifacts["sppc"], withdraw_artifacts["pk"])
    rc, proof, pw = p.prove(C.withdraw_inputs(withdraw_kat), 21, 22)
    assert rc == 0
    vk = open(withdraw_artifacts["vk"], "rb").read()
    bad = bytearray(proof); bad[5] ^= 1
    script = """
      const h = require(%s);
      const vk = Buffer.from(%s, 'hex'), pr = Buffer.from(%s, 'hex'), bad = Buffer.from(%s, 'hex'), pw = Buffer.from(%s, 'hex');
      let threw = false;
      try { h.addon.verify(vk, pr.slice(0, 100), pw); } catch (e) { threw = /libspp error -7/.test(e.message); }
      process.stdout.write(JSON.stringify({ok: h.addon.verify(vk, pr, pw), bad: h.addon.verify(vk, bad, pw), threw}));
    """ % (json.dumps(addon), json.dumps(vk.hex()), json.dumps(proof.hex()), json.dumps(bytes(bad).hex()), json.dumps(pw.hex()))
    r = _node(script)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout) == {"ok": True, "bad": False, "threw": True}


def test_generate_proof_hex_twins_agree(tmp_path):
    from spp import generate_proof_hex as G
    root = tmp_path
    js = os.path.join(NODE_DIR, "generate-proof-hex.js")
    r = subprocess.run(["node", js, str(root)], capture_output=True, text=True)
    code, out, err = G.render(str(root))
    assert r.returncode == 1 == code and r.stdout == out and r.stderr == err       # generate-proof-hex.ts:36-44
    assert "Error: Withdraw proof file not found at" in err and "  sunspot prove target/shielded_pool_verifier.json ..." in err
    for d, base in (("noir_circuit", "shielded_pool_verifier"), ("audit_circuit", "rlwe_audit")):
        os.makedirs(root / d / "target")
        (root / d / "target" / (base + ".proof")).write_bytes(bytes(range(97)) * 4)
        (root / d / "target" / (base + ".pw")).write_bytes(b"\x00\x00\x00\x05" + bytes(72))
    r = subprocess.run(["node", js, str(root)], capture_output=True, text=True)
    code, out, err = G.render(str(root))
    assert r.returncode == 0 == code and r.stdout == out
    assert "1. WITHDRAW PROOF (hex):\n" + "=" * 60 + "\n\n0x000102" in out                 # :82-87
    assert "Withdraw proof size: 388 bytes" in out and "4. Copy AUDIT PUBLIC WITNESS hex -> paste into 'Audit Public Witness (hex)' field" in out


@pytest.mark.gpu
def test_generate_proof_through_node_on_gpu(addon, withdraw_artifacts, withdraw_kat, tmp_path):
    """generateProof(config, inputs) as client/test-shielded-pool.ts:246 calls it; files as proof.helper.ts:68-69."""
    from oracle import groth16
    cdir = tmp_path / "noir_circuit"
    os.makedirs(cdir / "target")
    shutil.copy(withdraw_artifacts["sppc"], cdir / "target" / "shielded_pool_verifier.sppc")
    shutil.copy(withdraw_artifacts["pk"], cdir / "target" / "shielded_pool_verifier.pk")
    fields = ("root", "nullifier", "recipient", "amount", "wa_commitment", "secret_key", "owner_x", "owner_y", "randomness", "index", "siblings")
    script = """
      const h = require(%s);
      const r = h.generateProof({circuitDir: %s, circuitName: 'shielded_pool_verifier'}, %s);
      let threw = false;
      try { h.generateProof({circuitDir: %s, circuitName: 'shielded_pool_verifier'}, Object.assign({}, %s, {recipient: '0x0'})); } catch (e) { threw = /libspp error -4/.test(e.message); }
      const fs = require('fs');
      const vk = fs.readFileSync(%s);
      const badp = Buffer.from(r.proof); badp[9] ^= 1;
      const vb = h.addon.verifyBatch(vk, 2, Buffer.concat([r.proof, badp]), Buffer.concat([r.publicWitness, r.publicWitness]));
      process.stdout.write(JSON.stringify({proof: r.proof.toString('hex'), pw: r.publicWitness.toString('hex'), threw, vb}));
    """ % (json.dumps(addon), json.dumps(str(cdir)), json.dumps({f: withdraw_kat[f] for f in fields}), json.dumps(str(cdir)),
           json.dumps({f: withdraw_kat[f] for f in fields}), json.dumps(withdraw_artifacts["vk"]))
    r = _node(script)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    proof, pw = bytes.fromhex(out["proof"]), bytes.fromhex(out["pw"])
    assert len(proof) == 388 and len(pw) == 172 and out["threw"]
    assert out["vb"] == [True, False]                     # addon.verifyBatch on the GPU
    assert (cdir / "target" / "shielded_pool_verifier.proof").read_bytes() == proof
    assert (cdir / "Prover.toml").exists()
    assert groth16.verify(open(withdraw_artifacts["vk"], "rb").read(), proof, pw)


def test_node_helper_exports_audit_and_batch_and_toml_text(addon, rlwe_pk):
    """generateAuditProof(config, auditInputs) and generateProofBatch exist in the node helper; the audit
    Prover.toml it writes is the text of scripts/generate_audit.py:630-641 (byte-identical to the Python mirror's)."""
    import random
    from oracle import rlwe
    from spp.proof_helper import audit_prover_toml
    d = rlwe.audit_inputs(rlwe_pk["a"], rlwe_pk["b"], 12345, random.Random(999))
    inputs = dict(secret_key=d["secret_key"], wa_commitment=d["wa_commitment"], ct_commitment=d["ct_commitment"], c0_packed=d["c0_packed"],
                  c1_packed=d["c1_packed"], r=d["r"], e1_sparse=d["e1"], e2=d["e2"], k0=d["k0"], k1=d["k1"])
    as_js = {k: ([str(x) for x in v] if isinstance(v, list) else str(v)) for k, v in inputs.items()}
    script = """
      const h = require(%s);
      const inputs = %s;
      let threw = false;
      try { h.generateAuditProof({circuitDir: '/nonexistent', circuitName: 'x'}, Object.assign({}, inputs, {r: [1, 2]})); } catch (e) { threw = /r must hold 1024/.test(e.message); }
      process.stdout.write(JSON.stringify({fns: [typeof h.generateAuditProof, typeof h.generateProofBatch, typeof h.generateProof], toml: h.auditProverToml(inputs), threw,
                                           empty: h.generateProofBatch({circuitDir: '/nonexistent', circuitName: 'x'}, [])}));
    """ % (json.dumps(addon), json.dumps(as_js))
    r = _node(script)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out["fns"] == ["function"] * 3 and out["threw"] and out["empty"] == []
    assert out["toml"] == audit_prover_toml(inputs)
    assert out["toml"].startswith('secret_key = "0x%064x"\n' % 12345) and '"0"' in out["toml"]
    ts = open(os.path.join(NODE_DIR, "proof.helper.ts")).read()
    for name in ("export function generateProof(", "export function generateProofBatch(", "export function generateAuditProof(", "export interface AuditInputs",
                 "export interface ShieldedPoolInputs", "export interface CircuitConfig"):
        assert name in ts, name


@pytest.mark.gpu
def test_generate_proof_batch_and_audit_proof_through_node_on_gpu(addon, withdraw_artifacts, audit_artifacts, withdraw_kat, rlwe_pk, tmp_path):
    """generateProofBatch (payroll-demo.ts:326-352's three proofs in one call) and generateAuditProof through the N-API addon."""
    import random
    from oracle import groth16, rlwe, hashes as H
    cdir = tmp_path / "noir_circuit"
    adir = tmp_path / "audit_circuit"
    for d, art, name in ((cdir, withdraw_artifacts, "shielded_pool_verifier"), (adir, audit_artifacts, "rlwe_audit")):
        os.makedirs(d / "target")
        shutil.copy(art["sppc"], d / "target" / (name + ".sppc"))
        shutil.copy(art["pk"], d / "target" / (name + ".pk"))
    # three recipients, one tree (the payroll demo's shape)
    rng = random.Random(8)
    tree = H.MerkleTree()
    notes = []
    for _ in range(3):
        sk = rng.randrange(1, 1 << 128)
        owner = H.fixed_base_scalar_mul(sk)
        amount, rnd = rng.randrange(1, 1 << 40), rng.randrange(1 << 250)
        notes.append((sk, owner, amount, rnd, tree.insert(H.poseidon_hash4(owner[0], owner[1], amount, rnd))))
    hx = lambda v: "0x%064x" % v
    batch = [dict(root=hx(tree.root()), nullifier=hx(H.poseidon_hash2(sk, idx)), recipient=hx(rng.randrange(1, 1 << 240)), amount=amount,
                  wa_commitment=hx(H.poseidon_hash2(owner[0], owner[1])), secret_key=hx(sk), owner_x=hx(owner[0]), owner_y=hx(owner[1]),
                  randomness=hx(rnd), index=idx, siblings=[hx(s) for s in tree.proof(idx)]) for sk, owner, amount, rnd, idx in notes]
    d = rlwe.audit_inputs(rlwe_pk["a"], rlwe_pk["b"], 12345, random.Random(999))
    audit = dict(secret_key=d["secret_key"], wa_commitment=d["wa_commitment"], ct_commitment=d["ct_commitment"], c0_packed=d["c0_packed"],
                 c1_packed=d["c1_packed"], r=d["r"], e1_sparse=d["e1"], e2=d["e2"], k0=d["k0"], k1=d["k1"])
    audit_js = {k: ([str(x) for x in v] if isinstance(v, list) else str(v)) for k, v in audit.items()}
    script = """
      const h = require(%s);
      const rs = h.generateProofBatch({circuitDir: %s, circuitName: 'shielded_pool_verifier'}, %s);
      let threw = false;
      const bad = %s; bad[1] = Object.assign({}, bad[1], {amount: bad[1].amount + 1});
      try { h.generateProofBatch({circuitDir: %s, circuitName: 'shielded_pool_verifier'}, bad); } catch (e) { threw = /proof 1 do not satisfy/.test(e.message); }
      const a = h.generateAuditProof({circuitDir: %s, circuitName: 'rlwe_audit'}, %s);
      process.stdout.write(JSON.stringify({proofs: rs.map(r => r.proof.toString('hex')), pws: rs.map(r => r.publicWitness.toString('hex')), threw,
                                           ap: a.proof.toString('hex'), aw: a.publicWitness.toString('hex')}));
    """ % (json.dumps(addon), json.dumps(str(cdir)), json.dumps(batch), json.dumps(batch), json.dumps(str(cdir)), json.dumps(str(adir)), json.dumps(audit_js))
    r = _node(script)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out["threw"] and len(out["proofs"]) == 3
    vk = open(withdraw_artifacts["vk"], "rb").read()
    for p, w in zip(out["proofs"], out["pws"]):
        assert len(bytes.fromhex(p)) == 388 and len(bytes.fromhex(w)) == 172 and groth16.verify(vk, bytes.fromhex(p), bytes.fromhex(w))
    ap, aw = bytes.fromhex(out["ap"]), bytes.fromhex(out["aw"])
    assert len(ap) == 388 and aw == groth16.public_witness_bytes([d["wa_commitment"], d["ct_commitment"]])       # submit_audit.rs:18-21
    assert groth16.verify(open(audit_artifacts["vk"], "rb").read(), ap, aw)
    assert (adir / "target" / "rlwe_audit.proof").read_bytes() == ap and (adir / "Prover.toml").exists()
