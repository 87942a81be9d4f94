"""spp -- host-side Python over libspp.so (C ABI in include/spp.h).

Mirrors the reference's client/proof.helper.ts interface (ShieldedPoolInputs, CircuitConfig,
generateProof) on top of the HIP prover.  There is no CPU fallback: importing works anywhere (so the
ABI can be inspected), but every proving call needs the compiled library and a HIP device.
"""
from .lib import load_library, SppError, last_error  # noqa: F401
from .prover import Context, CircuitHandle, build_circuit, verify, pairing_check_host  # noqa: F401
from .proof_helper import (ShieldedPoolInputs, CircuitConfig, generate_proof, generateProof,  # noqa: F401
                           generateAuditProof, generate_audit_proof, generateProofBatch, generate_proof_batch)
