"""SURVEY section 5, sanitizer row: the reference has none (its known race is the shared Prover.toml of payroll-demo.ts:326).
GPU AddressSanitizer is not available on this pool, so the sanitizers run where they can: AddressSanitizer + UBSan builds of
 (1) every header the HIP kernels compile that also compiles for the host (field arithmetic, the 9x29-bit MSM arithmetic, both
     pairing implementations, the RLWE LDS-NTT phases) through the tests/host drivers,
 (2) the host-side circuit builders (csrc/circuit.cpp, csrc/circuit_audit.cpp),
 (3) the oracle's C restatement (setup, solver, prover, RLWE) driven through ctypes under LD_PRELOAD=libasan.
Any report (heap/stack overflow, use after free, signed overflow, misaligned or out-of-range shift ...) fails the test."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest
from conftest import ROOT, GOLDEN

CSRC = os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd", "csrc")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def _clean(proc, what):
    text = proc.stdout + proc.stderr
    assert proc.returncode == 0, "%s exit %d\n%s" % (what, proc.returncode, text[-3000:])
    assert "runtime error" not in text and "AddressSanitizer" not in text, "%s\n%s" % (what, text[-3000:])
    return proc.stdout


def test_device_headers_under_asan_ubsan(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")

    def run(name):
        exe = str(tmp_path / ("san_" + name))
        subprocess.run(["g++", "-O1", "-std=c++17"] + SAN + ["-I", CSRC, os.path.join(ROOT, "tests", "host", name + ".cpp"), "-o", exe], check=True)
        return name, subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=900)
    with ThreadPoolExecutor(3) as ex:
        results = list(ex.map(run, ["rlwe_ntt_check", "f29_check", "pairing_check"]))
    for name, proc in results:
        out = _clean(proc, name)
        assert out.strip().splitlines()[-1].startswith("OK"), out[-500:]


def test_circuit_builders_under_asan_ubsan(tmp_path, rlwe_pk):
    """The R1CS builders (gadgets, lookup argument, solver-program emitter, SPPC writer) are host code: run them instrumented
    for the withdraw circuit, its two variants and the audit circuit, and check the files equal the uninstrumented build's."""
    import spp
    exe = str(tmp_path / "san_build_circuit")
    subprocess.run(["g++", "-O1", "-std=c++17"] + SAN + ["-I", CSRC, "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host", "build_circuit.cpp"),
                    os.path.join(CSRC, "circuit.cpp"), os.path.join(CSRC, "circuit_audit.cpp"), "-o", exe], check=True)
    pk_txt = tmp_path / "pk.txt"
    pk_txt.write_text(" ".join(str(int(v)) for v in list(rlwe_pk["a"]) + list(rlwe_pk["b"])))
    for kind, cid, extra in (("withdraw", 1, []), ("withdraw-refshape", 3, []), ("withdraw-depth20", 4, []), ("audit", 2, [str(pk_txt)])):
        out = str(tmp_path / (kind + ".sppc"))
        _clean(subprocess.run([exe, kind, out] + extra, capture_output=True, text=True, env=ENV, timeout=900), kind)
        ref = str(tmp_path / (kind + ".ref.sppc"))
        spp.build_circuit(cid, ref, aux=(list(rlwe_pk["a"]) + list(rlwe_pk["b"])) if cid == 2 else None)
        assert open(out, "rb").read() == open(ref, "rb").read(), kind


def test_oracle_c_under_asan_ubsan(tmp_path, withdraw_kat, withdraw_artifacts):
    """oracle/c built with -fsanitize=address,undefined and driven like the parity tests drive it: setup, a proof, the batch
    checker, an RLWE instance."""
    lib_asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(lib_asan) or not os.path.exists(lib_asan):
        pytest.skip("libasan not installed")
    so = str(tmp_path / "liboracle_san.so")
    src = [os.path.join(ROOT, "oracle", "c", f) for f in ("field.c", "sha256.c", "groth16.c", "rlwe.c")]
    subprocess.run(["gcc", "-O1", "-fopenmp", "-fPIC", "-std=gnu11", "-shared"] + SAN + ["-o", so] + src, check=True)
    script = r'''
import ctypes, json, os, sys
sys.path.insert(0, %(root)r)
from oracle import native, circuit as C, groth16
native._LIB = None
import oracle.native as N
real = ctypes.CDLL
N._HERE = %(tmp)r
os.symlink(%(so)r, os.path.join(%(tmp)r, "liboracle.so"))
kat = json.load(open(%(kat)r))
sppc = %(sppc)r
pk, vk = os.path.join(%(tmp)r, "s.pk"), os.path.join(%(tmp)r, "s.vk")
N.set_threads(2)
N.setup(sppc, b"\x07" * 32, pk, vk)
assert open(pk, "rb").read() == open(%(refpk)r, "rb").read()
p = N.Prover(sppc, pk)
row = C.withdraw_inputs(kat)
rc, proof, pw = p.prove(row, 3, 4)
assert rc == 0 and groth16.verify(open(vk, "rb").read(), proof, pw)
bad = list(row); bad[1] += 1
assert N.check_many(p, [row, bad])[0] == -1 and N.check_many(p, [row, bad])[1] >= 0
import numpy as np
pkj = json.load(open(%(rlwe)r))
a = np.array(pkj["a"], dtype=np.uint32); b = np.array(pkj["b"], dtype=np.uint32)
r = np.array([(-1) ** i * (i %% 4) for i in range(1024)], dtype=np.int32); e1 = np.zeros(64, dtype=np.int32); e2 = np.ones(1024, dtype=np.int32)
m = np.arange(64, dtype=np.uint32)
c0 = np.zeros(64, dtype=np.uint32); c1 = np.zeros(1024, dtype=np.uint32); k0 = np.zeros(64, dtype=np.int64); k1 = np.zeros(1024, dtype=np.int64)
q = lambda x: x.ctypes.data_as(ctypes.c_void_p)
N.lib().orc_rlwe_witness(q(a), q(b), q(r), q(e1), q(e2), q(m), q(c0), q(c1), q(k0), q(k1))
assert int(c1.max()) < 167772161
print("SAN-OK")
''' % dict(root=ROOT, tmp=str(tmp_path / "lib"), so=so, kat=os.path.join(GOLDEN, "withdraw_kat.json"), sppc=withdraw_artifacts["sppc"],
           refpk=withdraw_artifacts["pk"], rlwe=os.path.join(GOLDEN, "rlwe_pk.json"))
    os.makedirs(str(tmp_path / "lib"))
    env = dict(ENV, LD_PRELOAD=lib_asan, OMP_NUM_THREADS="2")
    proc = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, timeout=1800)
    out = _clean(proc, "oracle/c under sanitizers")
    assert "SAN-OK" in out
