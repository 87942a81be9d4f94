#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference's Python (build container only).

Run from the repo root:  python tests/golden/make_fixtures.py
Reads  /root/reference/scripts/generate_audit.py (imported, never copied),
       /root/reference/demo-frontend/public/rlwe/rlwe_pk.json, client/prover-params.toml (data).
Writes tests/golden/{rlwe_pk.json, rlwe_vectors.json, withdraw_kat.json, pack_kat.json} and copies the binary artefacts
reference_{withdraw,audit}.vk, reference_withdraw.ccs, reference_withdraw_acir.json (bytecode + abi, no source text).
Only inputs and expected outputs are stored -- no reference source text.
"""
import importlib.util, json, os, random, re, sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_ref(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, "scripts", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode = True
    spec.loader.exec_module(mod)
    return mod


def main():
    ga = load_ref("generate_audit")
    a, b = ga.load_rlwe_pk()
    with open(os.path.join(HERE, "rlwe_pk.json"), "w") as f:
        json.dump({"a": a, "b": b}, f)

    N, Q, DELTA, SLOTS = ga.N, ga.RLWE_Q, ga.DELTA, ga.MSG_SLOTS
    vectors = []
    # RLWE-1: the reference's own run (generate_audit.py:469-505): Random(999), sk=12345 message
    cases = [("rlwe1_seed999", 999, None)] + [("rlwe_seed%d" % s, s, s) for s in (1, 2, 3)]
    for name, seed, msg_seed in cases:
        rng = random.Random(seed)
        if msg_seed is None:
            # message bytes of Grumpkin(12345*G); owner computed by our oracle (pinned separately by
            # prover-params KATs); the reference would obtain it from nargo (generate_audit.py:482)
            sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
            from oracle import hashes
            ox, oy = hashes.fixed_base_scalar_mul(12345)
            msg = ga.encode_field_to_bytes(ox, 32) + ga.encode_field_to_bytes(oy, 32)
        else:
            mr = random.Random(10_000 + msg_seed)
            msg = [mr.randrange(256) for _ in range(SLOTS)]
        r_s = [rng.randint(-3, 3) for _ in range(N)]
        e1_s = [rng.randint(-3, 3) for _ in range(SLOTS)]
        e2_s = [rng.randint(-3, 3) for _ in range(N)]
        r_q = [v % Q for v in r_s]
        br = ga.negacyclic_mul_mod_q(b, r_q, N, Q)
        ar = ga.negacyclic_mul_mod_q(a, r_q, N, Q)
        c0 = [(br[i] + e1_s[i] % Q + DELTA * msg[i]) % Q for i in range(SLOTS)]
        c1 = [(ar[i] + e2_s[i] % Q) % Q for i in range(N)]
        rows_b = [ga.negacyclic_matrix_row_mod_q(b, k, N, Q) for k in range(SLOTS)]
        rows_a = [ga.negacyclic_matrix_row_mod_q(a, k, N, Q) for k in range(N)]
        k0, k1 = [], []
        for i in range(SLOTS):
            full = sum(rows_b[i][j] * r_s[j] for j in range(N)) + e1_s[i] + DELTA * msg[i]
            k, rem = ga.compute_quotient_and_remainder(full, Q)
            assert rem == c0[i]
            k0.append(k)
        for i in range(N):
            full = sum(rows_a[i][j] * r_s[j] for j in range(N)) + e2_s[i]
            k, rem = ga.compute_quotient_and_remainder(full, Q)
            assert rem == c1[i]
            k1.append(k)
        vectors.append(dict(name=name, seed=seed, msg=msg, r=r_s, e1=e1_s, e2=e2_s,
                            br=br, ar=ar, c0=c0, c1=c1, k0=k0, k1=k1,
                            c0_packed=[hex(v) for v in ga.pack_values(c0)],
                            c1_packed=[hex(v) for v in ga.pack_values(c1)],
                            k0_fmt=[ga.format_field(v) for v in k0[:8]],
                            row_b_5=rows_b[5][:16], row_a_1000=rows_a[1000][:16]))
        print(name, "c1[0..2] =", c1[:3], "k range", min(k0 + k1), max(k0 + k1))
    with open(os.path.join(HERE, "rlwe_vectors.json"), "w") as f:
        json.dump(vectors, f)

    # auditor side (scripts/rlwe_decrypt.py): Shamir shares 1+2 (data files of the reference) -> sk, decrypt every vector
    rd = load_ref("rlwe_decrypt")
    shares = []
    for idx in (1, 2, 3):
        d = json.load(open(os.path.join(REF, "demo-frontend", "public", "rlwe", "rlwe_sk_shares", "share_%d.json" % idx)))
        shares.append({"share_index": d["share_index"], "threshold": d["threshold"], "x": d["coefficients"][0]["x"],
                       "y": [c["y"] for c in d["coefficients"]]})
    sk_q = []
    for k in range(N):
        v = rd.shamir_reconstruct_field([(shares[0]["x"], int(shares[0]["y"][k], 16)), (shares[1]["x"], int(shares[1]["y"][k], 16))], 2)
        sk_q.append(rd.centered_mod(v, rd.BN254_P) % Q)
    dec = []
    for v in vectors:
        sk_c1 = rd.negacyclic_mul_mod_q(sk_q, v["c1"], N, Q)
        msg = []
        for i in range(SLOTS):
            noisy = rd.centered_mod((v["c0"][i] + sk_c1[i]) % Q, Q)
            msg.append(round(noisy / rd.DELTA) % 256)
        assert msg == v["msg"], v["name"]
        dec.append({"name": v["name"], "sk_c1_head": sk_c1[:8], "msg": msg})
    with open(os.path.join(HERE, "rlwe_decrypt.json"), "w") as f:
        json.dump({"shares": shares[:2], "share3_y_head": shares[2]["y"][:4], "share3_x": shares[2]["x"], "sk_mod_q": sk_q, "decrypt": dec}, f)

    with open(os.path.join(HERE, "pack_kat.json"), "w") as f:
        json.dump({"in": list(range(1, 9)), "out": [hex(v) for v in ga.pack_values(list(range(1, 9)))],
                   "bytes_in": hex(0x0102030405), "bytes_out": ga.encode_field_to_bytes(0x0102030405, 8),
                   "fmt": [[-3, ga.format_field(-3)], [0, ga.format_field(0)], [5, ga.format_field(5)]]}, f)

    # withdraw KAT: client/prover-params.toml verbatim (data)
    txt = open(os.path.join(REF, "client", "prover-params.toml")).read()
    kat = {}
    for m in re.finditer(r'^(\w+) = (.+)$', txt, re.M):
        k, v = m.group(1), m.group(2).strip()
        if v.startswith('"'):
            kat[k] = v.strip('"')
        elif v != '[':
            kat[k] = int(v)
    kat["siblings"] = re.findall(r'^\s+"(0x[0-9a-f]+)",', txt, re.M)
    assert len(kat["siblings"]) == 16
    with open(os.path.join(HERE, "withdraw_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    # binary / compiled artefacts the reference holds for the proving path, kept as DATA fixtures:
    #  * the two gnark-made verifying keys
    #  * the gnark R1CS container of the withdraw circuit (App. A.4)
    #  * the compiled ACIR program of the withdraw circuit: bytecode + abi only -- file_map / debug_symbols embed Noir source
    #    text and are dropped
    import shutil
    shutil.copyfile(os.path.join(REF, "noir_circuit", "target", "shielded_pool_verifier.vk"), os.path.join(HERE, "reference_withdraw.vk"))
    shutil.copyfile(os.path.join(REF, "audit_circuit", "target", "rlwe_audit.vk"), os.path.join(HERE, "reference_audit.vk"))
    shutil.copyfile(os.path.join(REF, "noir_circuit", "target", "shielded_pool_verifier.ccs"), os.path.join(HERE, "reference_withdraw.ccs"))
    j = json.load(open(os.path.join(REF, "noir_circuit", "target", "shielded_pool_verifier.json")))
    out = {k: j[k] for k in ("noir_version", "hash", "abi", "bytecode", "expression_width")}
    out["_note"] = ("data fixture: the compiled ACIR program of the reference's withdraw circuit (noir_circuit/target/"
                    "shielded_pool_verifier.json) WITHOUT its file_map / debug_symbols members (those embed Noir source text); "
                    "made by tests/golden/make_fixtures.py")
    with open(os.path.join(HERE, "reference_withdraw_acir.json"), "w") as f:
        json.dump(out, f)
    print("fixtures written")


if __name__ == "__main__":
    main()
