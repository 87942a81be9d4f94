"""The algebra behind libspp's four-transform computeH ("computeH in product form"), checked in plain integer
arithmetic mod r -- no GPU, no curve: the group elements Z_j are replaced by random field scalars z_j, which is enough for a
statement that is linear in them.

gnark's computeH (groth16/bn254/prove.go, the step `sunspot prove` runs at client/proof.helper.ts:64) gives the proof
sum_j h_j Z_j with h = (A*B - C) / (X^n - 1).  libspp computes the same element as
    sum_wire w_wire * X_wire  +  sum_i A(zeta w^i) B(zeta w^i) * W'_i
with X_wire = sum_i C[i][wire] W_i, W_i = (1/2n) sum_j w^(-ij) Z_j, W'_i = -(1/2n) sum_j zeta^(-j) w^(-ij) Z_j, zeta^2 = w
(spp_api.cpp load_circuit_impl, h_mode 2).  The GPU tests assert the proof BYTES against the oracle's seven-transform prover;
this test pins the identity itself, and what becomes of it when the witness does not satisfy the system."""
import random

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _root_of_unity(log_n):
    g = pow(5, (R - 1) >> 28, R)               # 5 generates Fr*: a primitive 2^28-th root of unity
    assert pow(g, 1 << 27, R) == R - 1
    return pow(g, 1 << (28 - log_n), R)


def _interpolate(vals, w):
    """coefficients of the polynomial of degree < n with p(w^i) = vals[i] (plain O(n^2) inverse DFT)"""
    n = len(vals)
    ninv, winv = pow(n, -1, R), pow(w, -1, R)
    return [ninv * sum(vals[i] * pow(winv, i * j, R) for i in range(n)) % R for j in range(n)]


def _evaluate(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


def _system(rng, n, n_wires, satisfied=True):
    """random rows A, B over the wires; C_i chosen as one term on a wire with a non-zero value so that a_i b_i = c_i"""
    wit = [1] + [rng.randrange(1, R) for _ in range(n_wires - 1)]
    A = [[rng.randrange(R) if rng.random() < 0.5 else 0 for _ in range(n_wires)] for _ in range(n)]
    B = [[rng.randrange(R) if rng.random() < 0.5 else 0 for _ in range(n_wires)] for _ in range(n)]
    dot = lambda row: sum(c * v for c, v in zip(row, wit)) % R
    C = []
    for i in range(n):
        k = rng.randrange(n_wires)
        row = [0] * n_wires
        row[k] = dot(A[i]) * dot(B[i]) % R * pow(wit[k], -1, R) % R
        C.append(row)
    if not satisfied:
        C[3][0] = (C[3][0] + 1) % R
    return wit, A, B, C, dot


def _both_sides(rng, log_n, n_wires, satisfied=True):
    n = 1 << log_n
    w, zeta = _root_of_unity(log_n), _root_of_unity(log_n + 1)
    assert zeta * zeta % R == w
    wit, A, B, C, dot = _system(rng, n, n_wires, satisfied)
    a, b, c = [dot(r) for r in A], [dot(r) for r in B], [dot(r) for r in C]
    z = [rng.randrange(R) for _ in range(n - 1)]                 # stand-ins for pk.G1.Z (n - 1 points)
    # ---- gnark: h = (A B - C) / (X^n - 1), coefficient by coefficient ----
    pa, pb, pc = _interpolate(a, w), _interpolate(b, w), _interpolate(c, w)
    prod = [0] * (2 * n - 1)
    for i, x in enumerate(pa):
        for j, y in enumerate(pb):
            prod[i + j] = (prod[i + j] + x * y) % R
    num = [(prod[k] - (pc[k] if k < n else 0)) % R for k in range(2 * n - 1)]
    # division by X^n - 1: num = h X^n - h  =>  h_j = num[n + j] and the low half must be -h
    h = [num[n + j] for j in range(n - 1)]
    exact = all((num[j] + (h[j] if j < n - 1 else 0)) % R == 0 for j in range(n))
    lhs = sum(hj * zj for hj, zj in zip(h, z)) % R
    # ---- libspp: the product form ----
    inv2n, winv, zinv = pow(2 * n, -1, R), pow(w, -1, R), pow(zeta, -1, R)
    WH = [inv2n * sum(pow(winv, i * j, R) * z[j] for j in range(n - 1)) % R for i in range(n)]
    WZ = [(R - inv2n) * sum(pow(zinv, j, R) * pow(winv, i * j, R) * z[j] for j in range(n - 1)) % R for i in range(n)]
    X = [sum(C[i][k] * WH[i] for i in range(n)) % R for k in range(n_wires)]
    coset = [_evaluate(pa, zeta * pow(w, i, R) % R) * _evaluate(pb, zeta * pow(w, i, R) % R) % R for i in range(n)]
    rhs = (sum(wv * xv for wv, xv in zip(wit, X)) + sum(p_ * w_ for p_, w_ in zip(coset, WZ))) % R
    return lhs, rhs, exact


def test_product_form_equals_gnarks_compute_h_on_satisfied_systems():
    rng = random.Random(2024)
    for log_n, n_wires in ((3, 5), (4, 9), (5, 12)):
        lhs, rhs, exact = _both_sides(rng, log_n, n_wires)
        assert exact and lhs == rhs, (log_n, n_wires)


def test_product_form_needs_the_satisfaction_check():
    """With a_i b_i != c_i on one row A B - C is no multiple of X^n - 1: gnark then emits a proof that cannot verify, and the
    product form (which takes P(w^i) = c_i from the witness) computes a different element -- which is why k_spmv_check's verdict is
    final in libspp: such a row never reaches the transforms with status 0 (test_audit_batch_refuses_bad_rows_in_place)."""
    rng = random.Random(7)
    lhs, rhs, exact = _both_sides(rng, 4, 9, satisfied=False)
    assert not exact and lhs != rhs
