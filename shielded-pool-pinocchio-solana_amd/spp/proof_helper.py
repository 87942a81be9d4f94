"""Python mirror of the reference's client/proof.helper.ts (same names, argument meaning, side effects).

    ShieldedPoolInputs   proof.helper.ts:6-21
    CircuitConfig        proof.helper.ts:23-26
    generateProof()      proof.helper.ts:28-72   (alias generate_proof)

The reference writes <circuitDir>/Prover.toml (:29-52), spawns `nargo execute` (:55) and `sunspot prove`
(:64) and reads target/<name>.proof and .pw back (:68-69).  Here the two child processes are replaced by
one call into libspp (HIP); Prover.toml and the two output files are still written so that
client/generate-proof-hex.ts-style consumers keep working.  Errors surface as exceptions, as execSync's do.
"""
import os
from dataclasses import dataclass, field
from typing import List, Union

from .lib import SppError, SPP_CIRCUIT_WITHDRAW


@dataclass
class ShieldedPoolInputs:
    # public inputs
    root: str
    nullifier: str
    recipient: str
    amount: Union[int, str]
    wa_commitment: str
    # private inputs
    secret_key: str
    owner_x: str
    owner_y: str
    randomness: str
    index: Union[int, str]
    siblings: List[str] = field(default_factory=list)


@dataclass
class CircuitConfig:
    circuitDir: str
    circuitName: str


def prover_toml(inputs: ShieldedPoolInputs) -> str:
    """Exact text of proof.helper.ts:32-50."""
    toml = ""
    toml += 'root = "%s"\n' % inputs.root
    toml += 'nullifier = "%s"\n' % inputs.nullifier
    toml += 'recipient = "%s"\n' % inputs.recipient
    toml += "amount = %s\n" % inputs.amount
    toml += 'wa_commitment = "%s"\n' % inputs.wa_commitment
    toml += 'secret_key = "%s"\n' % inputs.secret_key
    toml += 'owner_x = "%s"\n' % inputs.owner_x
    toml += 'owner_y = "%s"\n' % inputs.owner_y
    toml += 'randomness = "%s"\n' % inputs.randomness
    toml += "index = %s\n" % inputs.index
    toml += "siblings = [\n"
    for sib in inputs.siblings:
        toml += '  "%s",\n' % sib
    toml += "]\n"
    return toml


def _field(v):
    if isinstance(v, str):
        return int(v, 16) if v.lower().startswith("0x") else int(v)
    return int(v)


def input_vector(inputs: ShieldedPoolInputs):
    """Order of the circuit's input wires: 5 public then 21 private (proof.helper.ts:34-50)."""
    if len(inputs.siblings) != 16:
        raise ValueError("siblings must hold 16 elements (TREE_DEPTH, noir_circuit/src/main.nr:5)")
    vals = [_field(getattr(inputs, k)) for k in ("root", "nullifier", "recipient", "amount", "wa_commitment",
                                                 "secret_key", "owner_x", "owner_y", "randomness", "index")]
    return vals + [_field(s) for s in inputs.siblings]


_HANDLES = {}


def _handle(config: CircuitConfig, window_bits=None):
    """Circuit + proving key resident in HBM, loaded once per (dir, name) -- the reference re-reads
    .ccs/.pk from disk in every `sunspot prove`.  The helper proves one statement (or a handful) per call, so it builds
    8-bit window tables (~6 GB, under a second) rather than the 225 GB a batch server uses; SPP_WINDOW overrides (0 = auto)."""
    from .prover import Context
    if window_bits is None:
        window_bits = int(os.environ.get("SPP_WINDOW", "8"))
    key = (os.path.abspath(config.circuitDir), config.circuitName)
    if key not in _HANDLES:
        target = os.path.join(config.circuitDir, "target")
        ccs = os.path.join(target, config.circuitName + ".sppc")
        pk = os.path.join(target, config.circuitName + ".pk")
        for p in (ccs, pk):
            if not os.path.exists(p):
                raise FileNotFoundError("ENOENT: no such file or directory, open '%s' (run setup first)" % p)
        ctx = Context(int(os.environ.get("SPP_DEVICE", "0")))
        _HANDLES[key] = ctx.load_circuit(ccs, pk, window_bits)
    return _HANDLES[key]


def generateProof(config: CircuitConfig, inputs: ShieldedPoolInputs, rs=None):
    """Synchronous; returns {"proof": bytes(388), "publicWitness": bytes(172)} like proof.helper.ts:71.
    rs: optional (r, s) blinding for reproducible proofs (SPP_SEED-style parity runs)."""
    with open(os.path.join(config.circuitDir, "Prover.toml"), "w") as f:
        f.write(prover_toml(inputs))
    h = _handle(config)
    if h.circuit_id != SPP_CIRCUIT_WITHDRAW:
        raise ValueError("generateProof expects the withdraw circuit")
    proofs, pws, status = h.prove_batch([input_vector(inputs)], None if rs is None else [rs])
    if status[0] != 0:
        raise SppError(status[0], "inputs do not satisfy the circuit (Command failed: sunspot prove)")
    target = os.path.join(config.circuitDir, "target")
    with open(os.path.join(target, config.circuitName + ".proof"), "wb") as f:
        f.write(proofs[0])
    with open(os.path.join(target, config.circuitName + ".pw"), "wb") as f:
        f.write(pws[0])
    return {"proof": proofs[0], "publicWitness": pws[0]}


generate_proof = generateProof


def generateProofBatch(config: CircuitConfig, inputs_list, rs=None):
    """Many withdraw proofs in one call (node/proof.helper generateProofBatch; the shape client/payroll-demo.ts:326-352 wants
    from its Promise.all).  Returns a list of {"proof", "publicWitness"}; raises naming the first unsatisfied index."""
    if not inputs_list:
        return []
    h = _handle(config)
    if h.circuit_id != SPP_CIRCUIT_WITHDRAW:
        raise ValueError("generateProofBatch expects the withdraw circuit")
    proofs, pws, status = h.prove_batch([input_vector(i) for i in inputs_list], rs)
    for k, st in enumerate(status):
        if st != 0:
            raise SppError(st, "inputs of proof %d do not satisfy the circuit" % k)
    return [{"proof": p, "publicWitness": w} for p, w in zip(proofs, pws)]


generate_proof_batch = generateProofBatch

BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_AUDIT_ORDER = ("secret_key", "wa_commitment", "ct_commitment", "c0_packed", "c1_packed", "r", "e1_sparse", "e2", "k0", "k1")


def _format_field(v):
    """scripts/generate_audit.py:77-82."""
    v = _field(v) % BN254_R
    return '"0"' if v == 0 else '"0x%064x"' % v


def audit_prover_toml(inputs: dict) -> str:
    """Text of scripts/generate_audit.py:630-641 (same key order, single-line arrays)."""
    out = ""
    for k in _AUDIT_ORDER:
        v = inputs[k]
        if isinstance(v, (list, tuple)):
            out += "%s = [%s]\n" % (k, ", ".join(_format_field(x) for x in v))
        else:
            out += "%s = %s\n" % (k, _format_field(v))
    return out


def generateAuditProof(config: CircuitConfig, inputs: dict, rs=None):
    """Audit-circuit counterpart of generateProof (the reference proves it from a shell script,
    audit_circuit/prove_audit.sh:74-99 / scripts/generate_audit.py:668-685): inputs carry the Prover.toml keys
    secret_key, wa_commitment, ct_commitment, c0_packed[10], c1_packed[147], r[1024], e1_sparse[64], e2[1024],
    k0[64], k1[1024] (signed values allowed). Returns {"proof": 388 B, "publicWitness": 76 B}."""
    from .lib import SPP_CIRCUIT_AUDIT
    with open(os.path.join(config.circuitDir, "Prover.toml"), "w") as f:
        f.write(audit_prover_toml(inputs))
    h = _handle(config)
    if h.circuit_id != SPP_CIRCUIT_AUDIT:
        raise ValueError("generateAuditProof expects the audit circuit")
    row = [_field(inputs["wa_commitment"]), _field(inputs["ct_commitment"])] + [_field(v) for v in inputs["c0_packed"]] + \
          [_field(v) for v in inputs["c1_packed"]] + [_field(inputs["secret_key"])]
    for k in ("r", "e1_sparse", "e2", "k0", "k1"):
        row += [_field(v) for v in inputs[k]]
    row = [v % BN254_R for v in row]
    proofs, pws, status = h.prove_batch([row], None if rs is None else [rs])
    if status[0] != 0:
        raise SppError(status[0], "inputs do not satisfy the circuit (Command failed: sunspot prove)")
    target = os.path.join(config.circuitDir, "target")
    with open(os.path.join(target, config.circuitName + ".proof"), "wb") as f:
        f.write(proofs[0])
    with open(os.path.join(target, config.circuitName + ".pw"), "wb") as f:
        f.write(pws[0])
    return {"proof": proofs[0], "publicWitness": pws[0]}


generate_audit_proof = generateAuditProof
