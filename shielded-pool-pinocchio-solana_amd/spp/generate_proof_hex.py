"""Python twin of the reference CLI client/generate-proof-hex.ts (:18-27 paths, :36-63 checks, :71-119 output):
prints the withdraw / audit proof and public-witness files as 0x-hex with the same banners.

    python -m spp.generate_proof_hex [repo_root]
"""
import os
import sys

_FILES = [
    ("Withdraw", "proof", "noir_circuit", "shielded_pool_verifier", "proof", "1. WITHDRAW PROOF (hex):"),
    ("Withdraw", "witness", "noir_circuit", "shielded_pool_verifier", "pw", "2. WITHDRAW PUBLIC WITNESS (hex):"),
    ("Audit", "proof", "audit_circuit", "rlwe_audit", "proof", "3. AUDIT PROOF (hex):"),
    ("Audit", "witness", "audit_circuit", "rlwe_audit", "pw", "4. AUDIT PUBLIC WITNESS (hex):"),
]
_FIELDS = ["'Proof (hex)'", "'Public Witness (hex)'", "'Audit Proof (hex)'", "'Audit Public Witness (hex)'"]


def render(root):
    """Returns (exit_code, stdout_text, stderr_text)."""
    bar = "=" * 60
    out = [bar, "Shielded Pool - Proof to Hex Converter", bar, ""]
    paths = [os.path.join(root, d, "target", "%s.%s" % (base, ext)) for (_, _, d, base, ext, _) in _FILES]
    for (label, kind, d, base, _, _), p in zip(_FILES, paths):
        if not os.path.exists(p):
            err = ["Error: %s %s file not found at %s" % (label, kind, p)]
            if kind == "proof":
                err += ["\nMake sure you have run:", "  cd %s" % d, "  nargo execute", "  sunspot prove target/%s.json ..." % base]
            return 1, "\n".join(out) + "\n", "\n".join(err) + "\n"
    blobs = [open(p, "rb").read() for p in paths]
    for i, ((label, kind, _, _, _, _), p, b) in enumerate(zip(_FILES, paths, blobs)):
        out += ["%s %s file: %s" % (label, kind, p), "%s %s size: %d bytes" % (label, kind, len(b))]
        if i % 2 == 1:
            out.append("")
    for (_, _, _, _, _, title), b in zip(_FILES, blobs):
        out += [bar, title, bar, "", "0x" + b.hex(), ""]
    out += [bar, "Instructions:", bar]
    for i, (_, _, _, _, _, title) in enumerate(_FILES):
        out.append("%d. Copy %s hex -> paste into %s field" % (i + 1, title[3:-7], _FIELDS[i]))
    out += ["5. Verify the recipient address matches the one used in Prover.toml", "6. Click 'Submit via Relayer'", ""]
    return 0, "\n".join(out) + "\n", ""


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    root = argv[0] if argv else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
    code, out, err = render(root)
    sys.stdout.write(out)
    sys.stderr.write(err)
    return code


if __name__ == "__main__":
    sys.exit(main())
