"""ORACLE (test infrastructure only) -- Poseidon (t=3,5), Poseidon2 (t=4), Grumpkin, Poseidon-Merkle.

Restates, with plain Python ints:
  * poseidon_hash_2 / poseidon_hash_4      noir_circuit/src/main.nr:1-9  (noir-lang/poseidon v0.1.1 bn254
                                           hash_2/hash_4 = circomlib Poseidon: client/merkle.ts:22-38)
  * compute_merkle_root                    noir_circuit/src/main.nr:11-29
  * ShieldedPoolMerkleTree                 client/merkle.ts:146-222
  * generateIdentityKeypair (Grumpkin)     client/merkle.ts:47-74,98-113
  * Poseidon2 sponge ct_commitment         ct_helper/src/main.nr:15-34 (= scripts/generate_audit.py:355-374)
Round constants are regenerated with the Grain LFSR of the Poseidon reference generator (third-party,
noir-lang/poseidon v0.1.1 / circomlibjs 0.1.7 are not vendored in the reference); pinned by
client/prover-params.toml (tests/test_oracle_golden.py). Poseidon2 t=4: the reference pins no value
("parity unpinned" at the reference level); pinned here by the literature KAT perm([0,1,2,3]).
"""
from .bn254 import R, P, inv

# ----------------------------------------------------------------------------- Grain LFSR


class Grain:
    def __init__(self, t, rf, rp, n=254, field=1, sbox=0):
        bits = []

        def put(v, w):
            for i in range(w - 1, -1, -1):
                bits.append((v >> i) & 1)
        put(field, 2)
        put(sbox, 4)
        put(n, 12)
        put(t, 12)
        put(rf, 10)
        put(rp, 10)
        bits.extend([1] * 30)
        assert len(bits) == 80
        self.s = bits
        for _ in range(160):
            self._step()

    def _step(self):
        s = self.s
        nb = s[0] ^ s[13] ^ s[23] ^ s[38] ^ s[51] ^ s[62]
        s.pop(0)
        s.append(nb)
        return nb

    def bit(self):
        while True:
            a = self._step()
            b = self._step()
            if a:
                return b

    def sample(self, n=254):
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bit()
        return v

    def field_rejection(self):
        while True:
            v = self.sample()
            if v < R:
                return v

    def field_mod(self):
        return self.sample() % R


_POSEIDON_CACHE = {}


def poseidon_params(t):
    """(RF, RP, round constants [(RF+RP)*t], MDS t x t) for circomlib-compatible Poseidon."""
    if t in _POSEIDON_CACHE:
        return _POSEIDON_CACHE[t]
    rf = 8
    rp = {2: 56, 3: 57, 4: 56, 5: 60}[t]
    g = Grain(t, rf, rp)
    rc = [g.field_rejection() for _ in range((rf + rp) * t)]
    while True:
        xy = [g.field_mod() for _ in range(2 * t)]
        if len(set(xy)) == 2 * t:
            break
    xs, ys = xy[:t], xy[t:]
    mds = [[inv(xs[i] + ys[j], R) for j in range(t)] for i in range(t)]
    _POSEIDON_CACHE[t] = (rf, rp, rc, mds)
    return _POSEIDON_CACHE[t]


def poseidon_permute(state):
    t = len(state)
    rf, rp, rc, mds = poseidon_params(t)
    s = list(state)
    for rnd in range(rf + rp):
        s = [(s[i] + rc[rnd * t + i]) % R for i in range(t)]
        if rnd < rf // 2 or rnd >= rf // 2 + rp:
            s = [pow(x, 5, R) for x in s]
        else:
            s[0] = pow(s[0], 5, R)
        s = [sum(mds[i][j] * s[j] for j in range(t)) % R for i in range(t)]
    return s


def poseidon_hash(inputs):
    """state = [0, in_1..in_{t-1}] -> permute -> state[0]."""
    return poseidon_permute([0] + [x % R for x in inputs])[0]


def poseidon_hash2(a, b):
    return poseidon_hash([a, b])


def poseidon_hash4(a, b, c, d):
    return poseidon_hash([a, b, c, d])


# ----------------------------------------------------------------------------- Poseidon2 t=4
_P2_CACHE = None
P2_ME = ((5, 7, 1, 3), (4, 6, 1, 1), (1, 3, 5, 7), (1, 1, 4, 6))
P2_MU_EXPECTED = (
    0x10dc6e9c006ea38b04b1e03b4bd9490c0d03f98929ca1d7fb56821fd19d3b6e7,
    0x0c28145b6a44df3e0149b3d0a30b3bb599df9756d4dd9b84a86b38cfb45a740b,
    0x00544b8338791518b2c7645a50392798b21f75bb60e3596170067d00141cac15,
    0x222c01175718386f2e2e82eb122789e352e105a3b8fa852613bc534433ee428b,
)


def poseidon2_params():
    """(rc[88], mu[4]) for BN254 t=4, RF=8, RP=56, d=5."""
    global _P2_CACHE
    if _P2_CACHE is None:
        g = Grain(4, 8, 56)
        rc = [g.field_rejection() for _ in range(8 * 4 + 56)]
        mu = None
        for cand in range(5):
            d = [g.field_mod() for _ in range(4)]
            if cand == 4:
                mu = tuple((x - 1) % R for x in d)
        _P2_CACHE = (rc, mu)
    return _P2_CACHE


def _p2_external(s):
    return [sum(P2_ME[i][j] * s[j] for j in range(4)) % R for i in range(4)]


def poseidon2_permute(state):
    rc, mu = poseidon2_params()
    s = _p2_external([x % R for x in state])
    k = 0
    for _ in range(4):
        s = [pow((s[i] + rc[k + i]) % R, 5, R) for i in range(4)]
        k += 4
        s = _p2_external(s)
    for _ in range(56):
        s[0] = pow((s[0] + rc[k]) % R, 5, R)
        k += 1
        tot = sum(s) % R
        s = [(mu[i] * s[i] + tot) % R for i in range(4)]
    for _ in range(4):
        s = [pow((s[i] + rc[k + i]) % R, 5, R) for i in range(4)]
        k += 4
        s = _p2_external(s)
    return s


def poseidon2_sponge(elems):
    """ct_helper/src/main.nr:15-34: rate 3, capacity slot state[3]; absorb by addition, permute
    after each full block, absorb the remainder (1 or 2), permute once more, output state[0]."""
    state = [0, 0, 0, 0]
    n = len(elems)
    full = n // 3
    for i in range(full):
        for j in range(3):
            state[j] = (state[j] + elems[3 * i + j]) % R
        state = poseidon2_permute(state)
    rem = n - 3 * full
    if rem >= 1:
        state[0] = (state[0] + elems[3 * full]) % R
    if rem >= 2:
        state[1] = (state[1] + elems[3 * full + 1]) % R
    state = poseidon2_permute(state)
    return state[0]


# ----------------------------------------------------------------------------- Grumpkin: y^2 = x^3 - 17 over Fr, order P
GRUMPKIN_G = (1, 17631683881184975370165255887551781615748388533673675138860)
GRUMPKIN_B = (-17) % R


def grumpkin_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % R == 0:
            return None
        lam = 3 * x1 * x1 * inv(2 * y1, R) % R
    else:
        lam = (y2 - y1) * inv(x2 - x1, R) % R
    x3 = (lam * lam - x1 - x2) % R
    return (x3, (lam * (x1 - x3) - y1) % R)


def grumpkin_mul(pt, k):
    acc = None
    add = pt
    while k:
        if k & 1:
            acc = grumpkin_add(acc, add)
        add = grumpkin_add(add, add)
        k >>= 1
    return acc


def identity_keypair(secret_key):
    """client/merkle.ts:98-113: sk reduced to 128 bits, pk = sk * G."""
    sk = secret_key % (1 << 128)
    pk = grumpkin_mul(GRUMPKIN_G, sk)
    return sk, pk


def fixed_base_scalar_mul(secret_key):
    """noir_circuit/src/main.nr:54-59: scalar = lo + 2^128*hi of the canonical field element."""
    return grumpkin_mul(GRUMPKIN_G, secret_key % R)


# ----------------------------------------------------------------------------- Merkle
TREE_DEPTH = 16


def compute_merkle_root(leaf, index, siblings):
    """noir_circuit/src/main.nr:11-29."""
    cur = leaf
    for i, sib in enumerate(siblings):
        if (index >> i) & 1 == 0:
            cur = poseidon_hash2(cur, sib)
        else:
            cur = poseidon_hash2(sib, cur)
    return cur


def default_hashes(depth=TREE_DEPTH):
    """client/merkle.ts:150-156."""
    d = [0]
    for _ in range(depth):
        d.append(poseidon_hash2(d[-1], d[-1]))
    return d


class MerkleTree:
    """client/merkle.ts:146-222 (ShieldedPoolMerkleTree), sparse evaluation of the same values."""

    def __init__(self, depth=TREE_DEPTH):
        self.depth = depth
        self.leaves = []
        self.defaults = default_hashes(depth)

    def insert(self, commitment):
        self.leaves.append(commitment % R)
        return len(self.leaves) - 1

    def _levels(self):
        if getattr(self, "_cache", None) is not None and self._cache[0] == len(self.leaves):
            return self._cache[1]               # same recomputation as client/merkle.ts, memoised until the next insert
        levels = [list(self.leaves)]
        for i in range(self.depth):
            cur = levels[-1]
            nxt = []
            for j in range(0, len(cur), 2):
                left = cur[j]
                right = cur[j + 1] if j + 1 < len(cur) else self.defaults[i]
                nxt.append(poseidon_hash2(left, right))
            levels.append(nxt)
        self._cache = (len(self.leaves), levels)
        return levels

    def root(self):
        lv = self._levels()
        return lv[self.depth][0] if lv[self.depth] else self.defaults[self.depth]

    def proof(self, index):
        lv = self._levels()
        out = []
        idx = index
        for i in range(self.depth):
            sib = idx ^ 1
            out.append(lv[i][sib] if sib < len(lv[i]) else self.defaults[i])
            idx >>= 1
        return out


def withdraw_public_values(secret_key, amount, randomness, index, siblings):
    """All values main.nr:38-82 asserts, from the private inputs."""
    owner = fixed_base_scalar_mul(secret_key)
    wa = poseidon_hash2(owner[0], owner[1])
    commitment = poseidon_hash4(owner[0], owner[1], amount, randomness)
    nullifier = poseidon_hash2(secret_key, index)
    root = compute_merkle_root(commitment, index, siblings)
    return dict(owner_x=owner[0], owner_y=owner[1], wa_commitment=wa, commitment=commitment,
                nullifier=nullifier, root=root)
