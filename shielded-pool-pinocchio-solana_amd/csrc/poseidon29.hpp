// Poseidon permutation (circomlib parameters: main.nr:1-9, client/merkle.ts:22-38) on the unsaturated 9x29-bit form,
// shared by the solver (kernels_solve.hip: every S-box power is a witness wire) and the stand-alone hash / Merkle kernels
// (kernels_witness.hip: nothing is emitted).
#pragma once
#include "f29.hpp"

namespace spp {

struct PoseidonNoEmit {
  __device__ __forceinline__ void operator()(const F29<FrParams>&) const {}
};

// Poseidon permutation with the state in the unsaturated 9x29-bit form (f29.hpp).  State words stay "x * 2^256"
// integers (what W holds), so no domain conversion is ever needed:
//   * S-box: xs = 32 * x (a 5-bit limb shift), then x2 = mont29(xs, x), x3 = mont29(xs, x2), ... : mont29 divides by 2^261,
//     the factor 32 restores 2^256 -- the four power wires of the S-box come out as storable words;
//   * MDS row: T products against entries pre-scaled by 2^261 accumulate in the same 64-bit columns and are reduced ONCE
//     (column bound (T+1) * 9 * 2^58 < 2^64 for T <= 5): 384 instructions per row for t = 3 instead of ~970.
// EMIT = true: x^2, x^3, x^4, x^5 of every S-box go to `emit` (four products through xs = 32x);
// EMIT = false: x^5 only, as x2 = xs*x, x4 = (32*x2)*x2, x5 = xs*x4 (three products).
template <int T, bool EMIT, class Emit>
__device__ __forceinline__ void poseidon_permute29(Fr (&st)[T], const Fr* __restrict__ rc, const uint32_t* __restrict__ mds29, int rp, Emit emit) {
  using F = F29<FrParams>;
  const int rf = 8;
  // value bounds: inputs < 2p are first brought below 1.01 p (one product with 2^261 mod p); round constants are uploaded
  // canonical (< p), so every S-box input is < 2.05 p and every stored power < 1.8 p -- inside Fp's [0, 2p) contract
  F s[T];
  SPP_UNROLL for (int i = 0; i < T; i++) s[i] = F::from_words(st[i].l) * F::template konst<FrParams::K29_ONE>();
  auto times32 = [](const F& x) {   // 32 * x as an integer (x normalised, < 4p: the result fits the nine limbs)
    F xs;
    xs.l[0] = (x.l[0] << 5) & F::M;
    SPP_UNROLL for (int k = 1; k < 8; k++) xs.l[k] = ((x.l[k] << 5) | (x.l[k - 1] >> 24)) & F::M;
    xs.l[8] = (x.l[8] << 5) | (x.l[7] >> 24);
    return xs;
  };
  auto sbox = [&](const F& x) {   // x normalised, < 2.05 p
    const F xs = times32(x);
    const F x2 = xs * x;
    if (EMIT) {
      emit(x2);
      const F x3 = xs * x2;
      emit(x3);
      const F x4 = xs * x3;
      emit(x4);
      const F x5 = xs * x4;
      emit(x5);
      return x5;
    }
    const F x4 = times32(x2) * x2;
    return xs * x4;
  };
#pragma unroll 1
  for (int r = 0; r < rf + rp; r++) {
    SPP_UNROLL for (int i = 0; i < T; i++) s[i] = add_norm(s[i], F::from_words(rc[r * T + i].l));
    const bool full = r < rf / 2 || r >= rf / 2 + rp;
    if (full) {
      SPP_UNROLL for (int i = 0; i < T; i++) s[i] = sbox(s[i]);
    } else {
      s[0] = sbox(s[0]);
    }
    F nx[T];
    SPP_UNROLL for (int i = 0; i < T; i++) {
      uint64_t c[18];
      F::clear(c);
      SPP_UNROLL for (int j = 0; j < T; j++) {
        F m;
        SPP_UNROLL for (int k = 0; k < 9; k++) m.l[k] = mds29[(i * T + j) * 9 + k];
        F::mac(c, m, s[j]);
      }
      nx[i] = F::reduce(c);
    }
    SPP_UNROLL for (int i = 0; i < T; i++) s[i] = nx[i];
  }
  SPP_UNROLL for (int i = 0; i < T; i++) s[i].to_words(st[i].l);
}

}  // namespace spp
