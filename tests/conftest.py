import os
import sys
import json
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "shielded-pool-pinocchio-solana_amd"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def withdraw_kat():
    return json.load(open(os.path.join(GOLDEN, "withdraw_kat.json")))


@pytest.fixture(scope="session")
def rlwe_pk():
    return json.load(open(os.path.join(GOLDEN, "rlwe_pk.json")))


@pytest.fixture(scope="session")
def rlwe_vectors():
    return json.load(open(os.path.join(GOLDEN, "rlwe_vectors.json")))


@pytest.fixture(scope="session")
def workdir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("spp"))


@pytest.fixture(scope="session")
def withdraw_artifacts(workdir):
    """SPPC built by the product's builder (host), pk/vk by the ORACLE's CPU setup (seed 7)."""
    import spp
    from oracle import native
    sppc = os.path.join(workdir, "withdraw.sppc")
    pk = os.path.join(workdir, "withdraw.pk")
    vk = os.path.join(workdir, "withdraw.vk")
    n = spp.build_circuit(1, sppc)
    native.setup(sppc, b"\x07" * 32, pk, vk)
    return dict(sppc=sppc, pk=pk, vk=vk, n_constraints=n)


@pytest.fixture(scope="session")
def audit_artifacts(workdir, rlwe_pk):
    """Audit SPPC from the product's builder over the reference's rlwe_pk.json; pk/vk from the oracle's CPU setup."""
    import spp
    from oracle import native
    sppc = os.path.join(workdir, "audit.sppc")
    pk = os.path.join(workdir, "audit.pk")
    vk = os.path.join(workdir, "audit.vk")
    n = spp.build_circuit(2, sppc, aux=list(rlwe_pk["a"]) + list(rlwe_pk["b"]))
    native.setup(sppc, b"\x09" * 32, pk, vk)
    return dict(sppc=sppc, pk=pk, vk=vk, n_constraints=n)
