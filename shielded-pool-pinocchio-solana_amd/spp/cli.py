"""`spp` command line: the four sunspot sub-commands the reference's scripts call, same positional arguments and
output-file naming (outputs land next to the circuit file, as sunspot writes them next to the .ccs).

    python -m spp.cli compile withdraw|audit [--rlwe-pk rlwe_pk.json] -o target/<name>.sppc     # prints nbConstraints=<n>
    python -m spp.cli compile target/<name>.json [-o target/<name>.sppc]                          # sunspot compile <acir>: the
                                                                                                  # nargo-compiled program itself
    python -m spp.cli setup   target/<name>.sppc [--seed HEX32] [--force]                         # -> <name>.pk, <name>.vk (+ .setup.json;
                                                                                                  #    skipped when the keys match the circuit)
    python -m spp.cli prove   target/<name>.sppc target/<name>.pk Prover.toml                    # -> <name>.proof, <name>.pw
    python -m spp.cli verify  target/<name>.vk target/<name>.proof target/<name>.pw              # exit 0 / 1
    python -m spp.cli execute target/<name>.json Prover.toml [-o target/<name>.gz]               # `nargo execute`: ACIR witness stack
    python -m spp.cli prove   target/<name>.json target/<name>.gz target/<name>.sppc target/<name>.pk
                              # sunspot's own argument order (acir, witness, constraint system, proving key;
                              # client/proof.helper.ts:58-64): the inputs are taken from the nargo witness file
    python -m spp.cli compile target/<name>.ccs                                                  # the reference's OWN gnark R1CS -> <name>.sppc
    python -m spp.cli setup   target/<name>.ccs [--seed HEX32]                                   # `sunspot setup <ccs>` -> <name>.pk, <name>.vk
    python -m spp.cli prove   target/<name>.json target/<name>.gz target/<name>.ccs target/<name>.pk
                              # `sunspot prove` on the files sunspot itself takes: gnark's solver loop (spp/ccs.py) completes the
                              # witness of the .ccs from the nargo witness, the GPU checks every row, commits and proves

Reference call sites: noir_circuit/prove_linux.sh:66-87, audit_circuit/prove_audit.sh:53-99,
scripts/generate_audit.py:659-691, scripts/benchmark_all.py:646-690 (parses the `nbConstraints=` line).
"""
import argparse
import json
import os
import re
import sys

from . import lib
from .prover import Context, build_circuit, verify

_AUDIT_KEYS = ("c0_packed", "c1_packed", "r", "e1_sparse", "e2", "k0", "k1")
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _num(tok):
    tok = tok.strip().strip('"')
    return int(tok, 16) if tok.lower().startswith("0x") else int(tok)


def parse_prover_toml(text):
    """Scalars and (possibly multi-line) arrays of quoted hex / bare integers, as written by
    client/proof.helper.ts:32-50 and scripts/generate_audit.py:630-641."""
    out = {}
    for m in re.finditer(r'^(\w+)\s*=\s*(\[[^\]]*\]|[^\n]+)$', text, re.M):
        k, v = m.group(1), m.group(2).strip()
        if v.startswith("["):
            out[k] = [_num(t) for t in re.findall(r'"[^"]*"|-?\w+', v[1:-1])]
        else:
            out[k] = _num(v)
    return out


def input_vector(vals):
    """Circuit input wires (public first) from a parsed Prover.toml; the circuit is recognised by its keys."""
    if "siblings" in vals:
        order = ("root", "nullifier", "recipient", "amount", "wa_commitment", "secret_key", "owner_x", "owner_y", "randomness", "index")
        return lib.SPP_CIRCUIT_WITHDRAW, [vals[k] % R for k in order] + [s % R for s in vals["siblings"]]
    row = [vals["wa_commitment"], vals["ct_commitment"]] + vals["c0_packed"] + vals["c1_packed"] + [vals["secret_key"]]
    for k in ("r", "e1_sparse", "e2", "k0", "k1"):
        row += vals[k]
    return lib.SPP_CIRCUIT_AUDIT, [v % R for v in row]


def main(argv=None):
    ap = argparse.ArgumentParser(prog="spp")
    sub = ap.add_subparsers(dest="cmd", required=True)
    c = sub.add_parser("compile"); c.add_argument("circuit", help="withdraw | audit | path of a nargo-compiled target/<name>.json (ACIR)")
    c.add_argument("--rlwe-pk"); c.add_argument("-o", "--out", default=None)
    s = sub.add_parser("setup"); s.add_argument("sppc"); s.add_argument("--seed", default=None); s.add_argument("--device", type=int, default=0)
    s.add_argument("--force", action="store_true", help="redo the setup even when matching keys exist")
    p = sub.add_parser("prove"); p.add_argument("files", nargs="+", help="<sppc> <pk> <Prover.toml>  |  <acir.json> <witness.gz> <sppc> <pk>")
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--window", type=int, default=0)
    p.add_argument("--rs", nargs=2, default=None, metavar=("R", "S"),
                   help="blinding factors (parity runs only: the default draws fresh ones from the OS, as every real proof must)")
    x = sub.add_parser("execute"); x.add_argument("acir"); x.add_argument("toml"); x.add_argument("-o", "--out", default=None)
    v = sub.add_parser("verify"); v.add_argument("vk"); v.add_argument("proof"); v.add_argument("pw")
    a = ap.parse_args(argv)
    if a.cmd == "compile" and a.circuit.endswith(".ccs"):
        from . import ccs
        c = ccs.load_ccs(a.circuit)
        n = ccs.to_sppc(ccs.decode_system(c), c, a.out or os.path.splitext(a.circuit)[0] + ".sppc")
        print("nbConstraints=%d" % n)
        return 0
    if a.cmd == "setup" and a.sppc.endswith(".ccs"):
        from . import ccs
        c = ccs.load_ccs(a.sppc)
        a.sppc = os.path.splitext(a.sppc)[0] + ".sppc"
        ccs.to_sppc(ccs.decode_system(c), c, a.sppc)
    if a.cmd == "compile":
        if a.circuit not in ("withdraw", "audit"):
            # `sunspot compile target/<name>.json` (prove_linux.sh:66-70): the reference's own compiled ACIR -> R1CS
            from . import acir
            out = a.out or os.path.splitext(a.circuit)[0] + ".sppc"
            try:
                n = acir.compile_to_sppc(a.circuit, out)
            except (acir.AcirFormatError, lib.SppError) as e:
                print("spp compile: %s" % e, file=sys.stderr)
                return 1
            print("nbConstraints=%d" % n)
            return 0
        if not a.out:
            ap.error("-o/--out is required")
        aux = None
        if a.circuit == "audit":
            if not a.rlwe_pk:
                ap.error("audit needs --rlwe-pk (demo-frontend/public/rlwe/rlwe_pk.json)")
            pk = json.load(open(a.rlwe_pk))
            aux = [_num(str(x)) for x in pk["a"]] + [_num(str(x)) for x in pk["b"]]
        n = build_circuit(lib.SPP_CIRCUIT_WITHDRAW if a.circuit == "withdraw" else lib.SPP_CIRCUIT_AUDIT, a.out, aux)
        print("nbConstraints=%d" % n)
        return 0
    if a.cmd == "setup":
        # the skip-if-exists step of noir_circuit/prove_linux.sh:72-79 ("if [ ! -f pk ] || [ ! -f vk ]; then sunspot setup"),
        # made safe: the keys are reused only when <name>.setup.json records the hash of THIS circuit file (and the same
        # seed, when one is given); a circuit rebuilt with other constraints gets a fresh setup.
        import hashlib
        base = os.path.splitext(a.sppc)[0]
        meta_path = base + ".setup.json"
        digest = hashlib.sha256(open(a.sppc, "rb").read()).hexdigest()
        if not a.force and all(os.path.exists(base + e) for e in (".pk", ".vk")) and os.path.exists(meta_path):
            try:
                meta = json.load(open(meta_path))
            except Exception:
                meta = {}
            if meta.get("circuit_sha256") == digest and (a.seed is None or meta.get("seed_sha256") == hashlib.sha256(bytes.fromhex(a.seed)).hexdigest()) \
                    and meta.get("pk_bytes") == os.path.getsize(base + ".pk"):
                print("setup: %s.pk / .vk are up to date for this circuit (use --force to redo)" % base)
                return 0
        seed = bytes.fromhex(a.seed) if a.seed else os.urandom(32)
        ctx = Context(a.device)
        ctx.setup(a.sppc, seed, base + ".pk", base + ".vk")
        ctx.close()
        json.dump({"circuit_sha256": digest, "seed_sha256": hashlib.sha256(seed).hexdigest(), "pk_bytes": os.path.getsize(base + ".pk")},
                  open(meta_path, "w"))
        return 0
    if a.cmd == "execute":
        from . import acir
        prog = acir.load_program(a.acir)
        _, row = input_vector(parse_prover_toml(open(a.toml).read()))
        try:
            w = acir.execute(prog, row)
        except acir.UnsatisfiedConstraint as e:
            print("spp execute: %s" % e, file=sys.stderr)
            return 1
        out = a.out or os.path.splitext(a.acir)[0] + ".gz"
        acir.write_witness_stack(out, w)
        print("[%s] Circuit witness successfully solved" % prog.main.name)
        print("[%s] Witness saved to %s" % (prog.main.name, out))
        return 0
    if a.cmd == "prove":
        if len(a.files) == 3:
            sppc, pk, toml = a.files
            _, row = input_vector(parse_prover_toml(open(toml).read()))
        elif len(a.files) == 4 and a.files[2].endswith(".ccs"):
            # the reference's own constraint system: every wire is an input of the container, the witness comes from gnark's
            # solver loop over the decoded .ccs, fed with the nargo witness and the commitment challenge of THIS proving key
            from . import acir, ccs
            acir_path, gz, ccs_path, pk = a.files
            c = ccs.load_ccs(ccs_path)
            system = ccs.decode_system(c)
            sppc = os.path.splitext(ccs_path)[0] + ".sppc"
            ccs.to_sppc(system, c, sppc)          # always from THIS .ccs: a stale container of another system must not be proved
            stack = acir.read_witness_stack(gz)
            public = acir.abi_input_row(acir.load_program(acir_path), stack)[:len(c.public) - 1]
            secret = {"__witness_%d" % k: v for k, v in stack.items()}
            ctx = Context(a.device)
            h = ctx.load_circuit(sppc, pk, a.window)
            try:
                row = ccs.reference_witness(system, c, public, secret, lambda partial: h.commitment_challenge([partial])[0])
                proofs, pws, status = h.prove_batch([row])
            except (ValueError, KeyError) as e:
                print("spp prove: %s" % e, file=sys.stderr)
                return 1
            finally:
                h.close()
                ctx.close()
            if status[0] != 0:
                print("spp prove: inputs do not satisfy the circuit", file=sys.stderr)
                return 1
            base = os.path.splitext(ccs_path)[0]
            open(base + ".proof", "wb").write(proofs[0])
            open(base + ".pw", "wb").write(pws[0])
            return 0
        elif len(a.files) == 4:
            from . import acir
            acir_path, gz, sppc, pk = a.files
            row = acir.abi_input_row(acir.load_program(acir_path), acir.read_witness_stack(gz))
        else:
            ap.error("prove takes <sppc> <pk> <Prover.toml> or <acir.json> <witness.gz> <sppc> <pk>")
        a.sppc, a.pk = sppc, pk
        base = os.path.splitext(a.sppc)[0]
        ctx = Context(a.device)
        h = ctx.load_circuit(a.sppc, a.pk, a.window)
        proofs, pws, status = h.prove_batch([row], None if a.rs is None else [(_num(a.rs[0]), _num(a.rs[1]))])
        h.close()
        ctx.close()
        if status[0] != 0:
            print("spp prove: inputs do not satisfy the circuit", file=sys.stderr)
            return 1
        open(base + ".proof", "wb").write(proofs[0])
        open(base + ".pw", "wb").write(pws[0])
        return 0
    ok = verify(open(a.vk, "rb").read(), open(a.proof, "rb").read(), open(a.pw, "rb").read())
    print("verification %s" % ("succeeded" if ok else "FAILED"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
