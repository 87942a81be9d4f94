"""CPU: host logic of the product (circuit builder, containers, host mirror) and the C ABI surface --
no compute call needs a GPU here."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT


def test_c_abi_exports_every_declared_symbol():
    import spp
    lib = spp.load_library()
    hdr = open(os.path.join(ROOT, "include", "spp.h")).read()
    names = sorted(set(re.findall(r"\b(spp_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "include/spp.h declares %s but libspp.so does not export it" % n
    assert lib.spp_version().startswith(b"libspp")


def test_no_device_fails_loudly():
    """There is no CPU fallback: without a HIP device spp_init must fail (skipped on the GPU box)."""
    import spp
    try:
        import torch
        if torch.cuda.device_count() > 0:
            pytest.skip("a GPU is visible")
    except ImportError:
        pass
    with pytest.raises(spp.SppError) as e:
        spp.Context(0)
    assert e.value.code == -2


def test_host_arithmetic_header_matches_oracle(tmp_path):
    """The header the HIP kernels compile (csrc/bn254.hpp), built for the host, against Python big ints."""
    import random
    from oracle import bn254 as B, hashes as H
    exe = str(tmp_path / "field_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host", "field_check.cpp"), "-o", exe], check=True)
    rng = random.Random(1)
    h = lambda v: "%064x" % v
    lines, exp = [], []
    for name, m in (("fr", B.R), ("fq", B.P)):
        cases = [(rng.randrange(m), rng.randrange(m)) for _ in range(60)] + [(0, 0), (m - 1, m - 1), (1, m - 1), (0, 5), (m - 1, 1)]
        for a, b in cases:
            lines.append("%s mul %s %s" % (name, h(a), h(b))); exp.append(h(a * b % m))
            lines.append("%s add %s %s" % (name, h(a), h(b))); exp.append(h((a + b) % m))
            lines.append("%s sub %s %s" % (name, h(a), h(b))); exp.append(h((a - b) % m))
            lines.append("%s neg %s" % (name, h(a))); exp.append(h((-a) % m))
        # inverses: the binary extended Euclid of Fp::inv() against Python, against a^(p-2), on the non-canonical word of the
        # same element, and on the values that drive its shift counts to the extremes (1, 2^k, p-1, (p+-1)/2, tiny, 0)
        inv_cases = [rng.randrange(1, m) for _ in range(120)] + [1, 2, 3, m - 1, m - 2, (m - 1) // 2, (m + 1) // 2, 1 << 128, 1 << 253,
                                                                  (1 << 253) + 1, 0xffffffff, 1 << 32, (1 << 64) - 1, m >> 1, 5]
        for a in inv_cases:
            for op in ("inv", "invf", "invnc"):
                lines.append("%s %s %s" % (name, op, h(a))); exp.append(h(pow(a, -1, m)))
        lines.append("%s inv %s" % (name, h(0))); exp.append(h(0))
        for _ in range(4):
            a = rng.randrange(1, m)
            u = rng.randrange(1 << 256)
            lines.append("%s u256 %s" % (name, h(u))); exp.append(h(u % m))
    for _ in range(6):
        a = (rng.randrange(B.P), rng.randrange(B.P)); b = (rng.randrange(B.P), rng.randrange(B.P))
        r = B.f2_mul(a, b)
        lines.append("fq2 mul %s %s %s %s" % (h(a[0]), h(a[1]), h(b[0]), h(b[1]))); exp.append(h(r[0]) + " " + h(r[1]))
        r = B.f2_inv(a)
        lines.append("fq2 inv %s %s" % (h(a[0]), h(a[1]))); exp.append(h(r[0]) + " " + h(r[1]))
    for _ in range(3):
        k1, k = rng.randrange(B.R), rng.randrange(B.R)
        p = B.g1_mul(B.G1_GEN, k1); r = B.g1_mul(p, k)
        lines.append("g1 mul %s %s %s" % (h(p[0]), h(p[1]), h(k))); exp.append(h(r[0]) + " " + h(r[1]))
        q = B.g1_mul(B.G1_GEN, rng.randrange(B.R))
        c = B.g1_add(B.g1_mul(p, 3), B.g1_mul(q, 2)); d = B.g1_add(B.g1_mul(p, 3), q)
        lines.append("g1 add %s %s %s %s" % (h(p[0]), h(p[1]), h(q[0]), h(q[1]))); exp.append(" ".join(h(v) for v in (c[0], c[1], d[0], d[1])))
        p2 = B.g2_mul(B.G2_GEN, k1); r2 = B.g2_mul(p2, k)
        lines.append("g2 mul " + " ".join(h(v) for v in (p2[0][0], p2[0][1], p2[1][0], p2[1][1])) + " " + h(k))
        exp.append(" ".join(h(v) for v in (r2[0][0], r2[0][1], r2[1][0], r2[1][1])))
        kk = rng.randrange(1 << 128); rg = H.grumpkin_mul(H.GRUMPKIN_G, kk)
        lines.append("gk mul %s %s %s" % (h(H.GRUMPKIN_G[0]), h(H.GRUMPKIN_G[1]), h(kk))); exp.append(h(rg[0]) + " " + h(rg[1]))
    out = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.strip().split("\n")
    assert out == exp


def test_withdraw_circuit_semantics(withdraw_artifacts, withdraw_kat):
    """The product's R1CS + solver program, interpreted by the Python oracle, accepts the reference's golden
    inputs and refuses exactly what noir_circuit/src/main.nr asserts against."""
    from oracle import circuit as C
    c = C.Circuit(withdraw_artifacts["sppc"])
    assert (c.n_public - 1, c.n_secret) == (5, 21) and c.n_constraints == withdraw_artifacts["n_constraints"]
    assert c.n_constraints <= 12452            # the reference's gnark R1CS (emulated-field Grumpkin) is larger
    good = C.withdraw_inputs(withdraw_kat)
    chal = lambda w: 0xabcdef
    assert C.first_unsatisfied(c, C.solve(c, good, chal)) == -1
    def bad(i, v):
        x = list(good); x[i] = v
        return C.first_unsatisfied(c, C.solve(c, x, chal)) >= 0
    assert bad(0, good[0] + 1)          # root            main.nr:78
    assert bad(1, good[1] + 1)          # nullifier       main.nr:74
    assert bad(2, 0)                    # recipient != 0  main.nr:81
    assert bad(3, 1 << 64)              # amount: u64     main.nr:43
    assert bad(4, good[4] + 1)          # wa_commitment   main.nr:67
    assert bad(5, good[5] + 1)          # secret_key -> public key mismatch main.nr:61-62
    assert bad(9, 1)                    # wrong leaf index
    assert bad(9, 1 << 16)              # index out of the 16-bit path range
    assert bad(12, good[12] + 1)        # a sibling


def test_audit_circuit_semantics(audit_artifacts, rlwe_pk):
    import random
    from oracle import circuit as C, rlwe
    c = C.Circuit(audit_artifacts["sppc"])
    assert (c.n_public - 1, c.n_inputs()) == (2, 3360) and c.domain_log == 15
    d = rlwe.audit_inputs(rlwe_pk["a"], rlwe_pk["b"], 12345, random.Random(999))
    good = rlwe.audit_input_vector(d)
    chal = lambda w: 0x1234567
    assert C.first_unsatisfied(c, C.solve(c, good, chal)) == -1
    def bad(i, v):
        x = list(good); x[i] = v
        return C.first_unsatisfied(c, C.solve(c, x, chal)) >= 0
    R0 = 2 + 157 + 1
    assert bad(0, good[0] + 1)                    # wa_commitment
    assert bad(1, good[1] + 1)                    # ct_commitment
    assert bad(2, good[2] + 1)                    # a packed ciphertext field
    assert bad(R0 + 5, (good[R0 + 5] + 1))        # r[5]: breaks the quotient equations
    assert bad(R0 + 7, 200)                       # r[7] outside [-128,127]
    assert bad(R0 + 1024 + 64 + 1024 + 3, good[R0 + 1024 + 64 + 1024 + 3] + 1)   # k0[3]


def test_cpu_oracle_end_to_end(withdraw_artifacts, withdraw_kat):
    """C oracle prover + Python pairing verifier: proof verifies, byte flips and wrong public inputs do not
    (client/test-shielded-pool.ts:386-417 failure modes), formats match withdraw.rs:13-16."""
    from oracle import native, groth16, circuit as C
    p = native.Prover(withdraw_artifacts["sppc"], withdraw_artifacts["pk"])
    row = C.withdraw_inputs(withdraw_kat)
    rc, proof, pw, wires = p.prove(row, 11111, 22222, want_wires=True)
    assert rc == 0 and len(proof) == 388 and len(pw) == 172
    assert pw == groth16.public_witness_bytes(row[:5])
    assert pw[12 + 32 * 3 + 24:12 + 32 * 4] == int(withdraw_kat["amount"]).to_bytes(8, "big")   # withdraw.rs:157-161
    vk = open(withdraw_artifacts["vk"], "rb").read()
    assert len(vk) == 1296
    assert groth16.verify(vk, proof, pw)
    bad = bytearray(proof); bad[0] ^= 1
    assert not groth16.verify(vk, bytes(bad), pw)
    pw2 = bytearray(pw); pw2[-1] ^= 1
    assert not groth16.verify(vk, proof, bytes(pw2))
    c = C.Circuit(withdraw_artifacts["sppc"])
    assert C.solve(c, row, lambda w: wires[c.challenge_wire], rs=(11111, 22222)) == wires      # two independent solvers agree
    rc2, proof2, _ = p.prove(row, 11111, 22222)
    assert proof2 == proof                                                   # deterministic under fixed (r, s)
    rc3, proof3, _ = p.prove(row, 5, 6)
    assert proof3 != proof and groth16.verify(vk, proof3, pw)                # different blinding, same statement
    # the commitment hides: one committed wire is a mask derived from (r, s), so the commitment point itself (bytes 260..324)
    # changes with the blinding although the committed VALUES are the same
    assert proof3[260:324] != proof[260:324] and proof2[260:324] == proof[260:324]
    from oracle.bn254 import hash_to_fr, DST_COMMITMENT, DST_MASK
    mask_wire = c.program[c.program.index(C.OP_MASK) + 1]       # (no other instruction of this program has the operand value 12)
    rs_bytes = (11111).to_bytes(32, "big") + (22222).to_bytes(32, "big")
    assert mask_wire in c.committed and wires[mask_wire] == hash_to_fr(rs_bytes, DST_MASK)[0]     # the mask's own domain ...
    assert wires[mask_wire] != hash_to_fr(rs_bytes, DST_COMMITMENT)[0]                            # ... not the challenge's
    rows_with_mask = [k for k in range(c.n_constraints) if any(mask_wire in m.row(k)[0] for m in (c.A, c.B, c.C))]
    assert len(rows_with_mask) == 1 and c.A.row(rows_with_mask[0])[0] == (mask_wire,) and c.B.row(rows_with_mask[0])[0] == (0,)   # mask * 1 = mask
    assert p.prove([row[0] + 1] + row[1:], 1, 2)[0] == 1                     # unsatisfied -> refused


def test_prover_toml_and_input_order(withdraw_kat):
    """Host mirror of client/proof.helper.ts:32-50."""
    from spp.proof_helper import ShieldedPoolInputs, prover_toml, input_vector
    from oracle import circuit as C
    k = withdraw_kat
    inp = ShieldedPoolInputs(**{f: k[f] for f in ("root", "nullifier", "recipient", "amount", "wa_commitment", "secret_key", "owner_x",
                                                   "owner_y", "randomness", "index", "siblings")})
    t = prover_toml(inp)
    assert t.startswith('root = "%s"\nnullifier = "%s"\nrecipient = "%s"\namount = %d\nwa_commitment = ' % (
        k["root"], k["nullifier"], k["recipient"], k["amount"]))
    assert 'index = 0\nsiblings = [\n  "%s",\n' % k["siblings"][0] in t and t.endswith('",\n]\n')
    assert input_vector(inp) == C.withdraw_inputs(k)
    with pytest.raises(ValueError):
        input_vector(ShieldedPoolInputs(**{**inp.__dict__, "siblings": k["siblings"][:3]}))


def test_shard_ranges():
    from spp.multi import shard_range
    for total in (0, 1, 7, 1024, 1025):
        for world in (1, 2, 3, 8):
            rs = [shard_range(total, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1


def test_product_verifier_agrees_with_oracle(withdraw_artifacts, audit_artifacts, withdraw_kat, rlwe_pk):
    """spp_verify (host pairing in the product, `sunspot verify` equivalent) vs the oracle's Python verifier."""
    import random
    import spp
    from oracle import native, groth16, circuit as C, rlwe
    cases = []
    p = native.Prover(withdraw_artifacts["sppc"], withdraw_artifacts["pk"])
    rc, proof, pw = p.prove(C.withdraw_inputs(withdraw_kat), 3, 4)
    cases.append((open(withdraw_artifacts["vk"], "rb").read(), proof, pw))
    pa = native.Prover(audit_artifacts["sppc"], audit_artifacts["pk"])
    row = rlwe.audit_input_vector(rlwe.audit_inputs(rlwe_pk["a"], rlwe_pk["b"], 12345, random.Random(999)))
    rc, proof, pw = pa.prove(row, 5, 6)
    cases.append((open(audit_artifacts["vk"], "rb").read(), proof, pw))
    for vk, proof, pw in cases:
        assert spp.verify(vk, proof, pw) and groth16.verify(vk, proof, pw)
        for pos in (0, 70, 200, 259, 300, 387):                      # Ar, Bs, Krs, count, commitment, PoK
            bad = bytearray(proof); bad[pos] ^= 1
            try:
                got = spp.verify(vk, bytes(bad), pw)
            except spp.SppError:
                got = False
            assert got is False
        pw2 = bytearray(pw); pw2[-1] ^= 1
        assert not spp.verify(vk, proof, bytes(pw2))
        # aliased public inputs: v + r encodes the same field element in other bytes (for a nullifier: a second spend of
        # the same note); gnark's witness reader refuses such words, and so must every verifier here -- for EVERY input
        from oracle import bn254 as B
        npub = (len(pw) - 12) // 32
        tried = 0
        for k in range(npub):
            v = int.from_bytes(pw[12 + 32 * k:44 + 32 * k], "big")
            for mult in (1, 2, 3, 4, 5):
                if v + mult * B.R < 1 << 256:
                    al = pw[:12 + 32 * k] + (v + mult * B.R).to_bytes(32, "big") + pw[44 + 32 * k:]
                    assert not spp.verify(vk, proof, al), (k, mult)
                    assert not groth16.verify(vk, proof, al), (k, mult)
                    tried += 1
        assert tried >= 5 * npub - 2
        # aliased point coordinates (x + q): refused, the proof is not malleable through its encoding
        for off in list(range(0, 256, 32)) + list(range(260, 388, 32)):
            v = int.from_bytes(proof[off:off + 32], "big") + B.P
            if v < 1 << 256:
                al = proof[:off] + v.to_bytes(32, "big") + proof[off + 32:]
                assert not spp.verify(vk, al, pw), off
                assert not groth16.verify(vk, al, pw), off
    with pytest.raises(spp.SppError):
        spp.verify(cases[0][0], cases[0][1][:100], cases[0][2])      # truncated proof
    assert not spp.verify(cases[0][0], cases[1][1], cases[0][2])     # proof of the other circuit


def test_cli_compile_verify_and_toml(tmp_path, withdraw_artifacts, withdraw_kat, capsys):
    from spp import cli
    from spp.proof_helper import ShieldedPoolInputs, prover_toml
    from oracle import native, circuit as C
    out = str(tmp_path / "w.sppc")
    assert cli.main(["compile", "withdraw", "-o", out]) == 0
    assert capsys.readouterr().out.strip() == "nbConstraints=%d" % withdraw_artifacts["n_constraints"]   # benchmark_all.py:646 parses this
    assert open(out, "rb").read() == open(withdraw_artifacts["sppc"], "rb").read()
    k = withdraw_kat
    fields = ("root", "nullifier", "recipient", "amount", "wa_commitment", "secret_key", "owner_x", "owner_y", "randomness", "index", "siblings")
    cid, row = cli.input_vector(cli.parse_prover_toml(prover_toml(ShieldedPoolInputs(**{f: k[f] for f in fields}))))
    assert cid == 1 and row == C.withdraw_inputs(k)
    # audit Prover.toml as scripts/generate_audit.py:630-641 writes it (single-line arrays, negative values mod p)
    from oracle import rlwe
    fmt = rlwe.format_field
    toml = "secret_key = %s\nwa_commitment = %s\nct_commitment = %s\n" % (fmt(12345), fmt(7), fmt(9))
    arrs = {"c0_packed": [3] * 10, "c1_packed": [4] * 147, "r": [-1] * 1024, "e1_sparse": [2] * 64, "e2": [0] * 1024, "k0": [-5] * 64, "k1": [6] * 1024}
    for name, vals in arrs.items():
        toml += "%s = [%s]\n" % (name, ", ".join(fmt(v) for v in vals))
    cid, row = cli.input_vector(cli.parse_prover_toml(toml))
    assert cid == 2 and len(row) == 3360 and row[0] == 7 and row[2 + 157] == 12345 and row[2 + 157 + 1] == rlwe.BN254_R - 1 and row[-1] == 6
    p = native.Prover(withdraw_artifacts["sppc"], withdraw_artifacts["pk"])
    rc, proof, pw = p.prove(C.withdraw_inputs(k), 1, 2)
    (tmp_path / "a.proof").write_bytes(proof); (tmp_path / "a.pw").write_bytes(pw)
    assert cli.main(["verify", withdraw_artifacts["vk"], str(tmp_path / "a.proof"), str(tmp_path / "a.pw")]) == 0
    (tmp_path / "b.proof").write_bytes(bytes([proof[0] ^ 1]) + proof[1:])
    assert cli.main(["verify", withdraw_artifacts["vk"], str(tmp_path / "b.proof"), str(tmp_path / "a.pw")]) == 1


def test_f29_unsaturated_arithmetic_host_check(tmp_path):
    """csrc/f29.hpp (the 9x29-bit limb form the G1 MSM accumulators run on) against Fp on random and extremal-limb
    inputs, and XYZZ29::madd chains (doubling / cancellation / negated entries) against XYZZ<Fq>::madd."""
    exe = str(tmp_path / "f29_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host", "f29_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.strip().splitlines()[-1].startswith("OK "), out


def test_rlwe_lds_ntt_phases_host_check(tmp_path):
    """csrc/rlwe_ntt.hpp: the per-lane phases of the RLWE witness kernel's 1024-point LDS NTT (two primes + CRT digit + wrap
    correction), run lane by lane on the host against the schoolbook definition of scripts/generate_audit.py:45-66,236-243 --
    random, extremal (a = q-1, r = +-128, e = +-128, m = 255), zero and sparse public keys -- plus the derived growth bounds
    of the lazy signed arithmetic."""
    exe = str(tmp_path / "rlwe_ntt_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host", "rlwe_ntt_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.strip().splitlines()[-1].startswith("OK rlwe_ntt"), out


def test_f29_limb_bounds_certificate():
    """Interval arithmetic over the generated constants: no 64-bit column can overflow, every lifted subtraction
    constant dominates its subtrahend, and the accumulator's value bounds are inductive."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("f29_bounds", os.path.join(ROOT, "tests", "host", "f29_bounds.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for params in ("FqParams", "FrParams"):
        log = m.check_madd(params)
        assert log["Y3"] < 1.2 and log["PP"] < 1.3
    g2 = m.check_madd_g2()
    assert g2["Y3.c0"] < 1.5 and g2["PP.c0"] < 2.3


def test_withdraw_reference_shape_circuit(tmp_path, withdraw_kat):
    """SPP_CIRCUIT_WITHDRAW_REFSHAPE: same statement, ballast up to the reference's gnark R1CS size (12 452 constraints,
    domain 2^14: SURVEY F6); the Python interpreter accepts the golden inputs and still refuses a wrong root."""
    import spp
    from oracle import circuit as C
    path = str(tmp_path / "wref.sppc")
    assert spp.build_circuit(spp.lib.SPP_CIRCUIT_WITHDRAW_REFSHAPE, path) == 12452
    c = C.Circuit(path)
    assert c.id == 1 and c.domain_log == 14 and (c.n_public - 1, c.n_secret) == (5, 21) and c.n_wires > 12000
    good = C.withdraw_inputs(withdraw_kat)
    chal = lambda w: 0xabcdef
    assert C.first_unsatisfied(c, C.solve(c, good, chal)) == -1
    bad = list(good); bad[0] += 1
    assert C.first_unsatisfied(c, C.solve(c, bad, chal)) >= 0


def test_batched_verifier_pairing_path_host_check(tmp_path):
    """csrc/pairing_fast.hpp (what the GPU verifier runs: shared Miller loop over per-key line tables, projective lines
    for the proof's G2 point, x-power final exponentiation) against the single-proof host pairing, compiled for the host."""
    exe = str(tmp_path / "pairing_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host", "pairing_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.strip().splitlines()[-1].startswith("OK "), out


def test_verifiers_reject_bs_outside_the_subgroup(withdraw_artifacts, withdraw_kat):
    """A point on the twist but outside the order-r subgroup in the Bs slot is refused by the oracle and by the product's
    host verifier before any pairing (the twist's cofactor is ~2^254)."""
    import spp
    from oracle import native, groth16, bn254 as B, circuit as C
    p = native.Prover(withdraw_artifacts["sppc"], withdraw_artifacts["pk"])
    row = C.withdraw_inputs(withdraw_kat)
    rc, proof, pw = p.prove(row, 3, 4)
    vk = open(withdraw_artifacts["vk"], "rb").read()
    assert rc == 0 and groth16.verify(vk, proof, pw) and spp.verify(vk, proof, pw)
    # find a twist point with small x: y^2 = x^3 + b'  (square root in Fq2, p = 3 mod 4)
    P = B.P

    def fq_sqrt(v):
        r = pow(v, (P + 1) // 4, P)
        return r if r * r % P == v % P else None

    def fq2_sqrt(a):
        alpha = fq_sqrt((a[0] * a[0] + a[1] * a[1]) % P)
        if alpha is None:
            return None
        for d in ((a[0] + alpha) * pow(2, -1, P) % P, (a[0] - alpha) * pow(2, -1, P) % P):
            x0 = fq_sqrt(d)
            if x0:
                y = (x0, a[1] * pow(2 * x0, -1, P) % P)
                if B.f2_mul(y, y) == (a[0] % P, a[1] % P):
                    return y
        return None
    x = 1
    pt = None
    while pt is None:
        x += 1
        rhs = B.f2_add(B.f2_mul(B.f2_mul((x, 0), (x, 0)), (x, 0)), B.G2_B)
        y = fq2_sqrt(rhs)
        if y is not None:
            pt = ((x, 0), y)
    assert B.g2_is_on_curve(pt) and groth16._g2_times_r(pt) is not None
    forged = proof[:64] + B.g2_to_bytes(pt) + proof[192:]
    assert not groth16.verify(vk, forged, pw)
    assert not spp.verify(vk, forged, pw)


def test_no_return_address_clobber_in_device_code(tmp_path):
    """Codegen hazard met on gfx950 / ROCm 7.2: in a large LEAF device function whose loop back-edges
    need long branches, the branch relaxation used s[30:31] -- the live return address -- for s_getpc/s_setpc, and the
    function never returned.  Out-of-line device functions are structured to avoid it; this scans the gfx950 assembly of
    EVERY translation unit of libspp (all .hip files, and the .cpp files the Makefile compiles with -x hip) for the pattern."""
    import glob
    import shutil
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    csrc = os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd", "csrc")
    units = sorted(glob.glob(os.path.join(csrc, "*.hip"))) + [os.path.join(csrc, f) for f in ("spp_api.cpp", "circuit.cpp", "circuit_audit.cpp")]
    assert len(units) >= 10

    def scan(src):
        asm = str(tmp_path / (os.path.basename(src) + ".s"))
        subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-x", "hip", "-S",
                        "--cuda-device-only", "-o", asm, src], check=True, stderr=subprocess.DEVNULL)
        txt = open(asm).read()
        kernels = set(re.findall(r"\.amdhsa_kernel (\S+)", txt))
        bad, n_funcs = [], 0
        for name in re.findall(r"^(_Z\w+):", txt, re.M):
            if name in kernels:
                continue
            n_funcs += 1
            body = txt.split(name + ":", 1)[1]
            body = body[:body.find(".Lfunc_end")]
            if re.search(r"s_getpc_b64 s\[30:31\]", body):
                bad.append(os.path.basename(src) + ":" + name)
        return bad, n_funcs

    with ThreadPoolExecutor(4) as ex:
        results = list(ex.map(scan, units))
    offenders = [o for bad, _ in results for o in bad]
    assert not offenders, offenders
    assert sum(n for _, n in results) > 0      # the scan saw out-of-line device functions at all


def test_committed_bench_line_follows_the_contract():
    """The bench line committed under profiles/ (the output of `python bench.py` on the MI355X box, this round) carries every
    field of the driver's contract, the roofline object and the CPU baseline, with consistent arithmetic; the headline is the
    audit circuit on a batch of distinct witnesses, and every extra leg ran at least 10 timed steps."""
    import json
    j = json.load(open(os.path.join(ROOT, "profiles", "round3_bench_default_run.json")))
    for k, t in (("metric", str), ("value", (int, float)), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", (int, float)), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(j[k], t), k
    assert j["vs_baseline"] is None and j["scaling"] == "weak" and j["higher_is_better"] is True and "workload" in j["config"]
    c = j["config"]
    assert c["circuit"] == "audit" and c["n_distinct_witnesses"] == c["batch_per_gpu"] == 2048 and j["steps"] >= 10
    assert abs(j["value"] - c["batch_per_gpu"] * j["n_gpus"] / (j["ms_per_step"] * 1e-3)) / j["value"] < 0.01
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 0.01
    assert r["launches_per_step"] == 4 and r["launches_per_step"] * r["avg_launch_ms"] <= j["ms_per_step"]   # A, B1, K, Z walks fit inside a step
    assert r["traffic"] is None or r["traffic"] > r["alg_bytes_per_launch"]
    cb = j["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == j["unit"] and cb["sample"]
    for leg in ("withdraw_circuit", "withdraw_at_reference_r1cs_size", "withdraw_depth20_variant"):
        assert j[leg]["steps"] >= 10 and j[leg]["config"]["n_distinct_witnesses"] == j[leg]["config"]["batch_per_gpu"], leg
    assert j["withdraw_at_reference_r1cs_size"]["config"]["n_constraints"] == 12452
    assert j["rlwe_witness_2p16"]["iters"] >= 10 and j["msm_g1_2p24"]["iters"] >= 10
    # round 3's legs: the per-rank shard of configs[2], the clock at the secrets, both circuits resident, the reference's own R1CS
    assert j["strong_scaling_rank_rehearsal"]["batch"] == 128 and j["strong_scaling_rank_rehearsal"]["steps"] >= 10
    assert j["audit_end_to_end_from_secrets"]["proof_bytes_equal_the_proofs_from_rows"] is True
    assert j["audit_plus_withdraw_coresident"]["steps"] >= 10 and j["withdraw_reference_gnark_r1cs"]["config"]["n_inputs"] == 26
    # the PMC summary the line's `traffic` comes from describes the same workload
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_hbm_latest.json")))
    assert pmc["circuit"] == "audit" and pmc["batch"] == 2048 and pmc["n_distinct_witnesses"] == 2048


def _depth20_rows(count):
    import random
    from oracle import hashes as H
    rng = random.Random(2020)
    tree = H.MerkleTree(20)
    notes = []
    for _ in range(count):
        sk = rng.randrange(1, 1 << 128)
        owner = H.fixed_base_scalar_mul(sk)
        amount = rng.randrange(1, 1 << 40)
        rnd = rng.randrange(1 << 250)
        idx = tree.insert(H.poseidon_hash4(owner[0], owner[1], amount, rnd))
        notes.append((sk, owner, amount, rnd, idx))
    root = tree.root()
    return [[root, H.poseidon_hash2(sk, idx), rng.randrange(1, 1 << 240), amount, H.poseidon_hash2(owner[0], owner[1]),
             sk, owner[0], owner[1], rnd, idx] + tree.proof(idx) for sk, owner, amount, rnd, idx in notes]


def test_withdraw_depth20_variant_semantics(tmp_path):
    """SPP_CIRCUIT_WITHDRAW_DEPTH20 (SURVEY 8d Config 2's synthetic variant): 20 siblings, accepted by the Python
    interpreter for notes of a depth-20 tree, refused for a wrong sibling at the new levels and for an index >= 2^20."""
    import spp
    from oracle import circuit as C
    path = str(tmp_path / "w20.sppc")
    n = spp.build_circuit(spp.lib.SPP_CIRCUIT_WITHDRAW_DEPTH20, path)
    c = C.Circuit(path)
    assert c.n_constraints == n and (c.n_public - 1, c.n_secret) == (5, 25) and c.domain_log == 14
    rows = _depth20_rows(3)
    chal = lambda w: 0x5eed
    for r in rows:
        assert C.first_unsatisfied(c, C.solve(c, r, chal)) == -1
    bad = list(rows[1]); bad[10 + 19] += 1                # the topmost sibling
    assert C.first_unsatisfied(c, C.solve(c, bad, chal)) >= 0
    bad = list(rows[1]); bad[9] = 1 << 20
    assert C.first_unsatisfied(c, C.solve(c, bad, chal)) >= 0
