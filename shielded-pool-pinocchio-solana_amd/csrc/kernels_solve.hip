// Witness generation on the GPU: input decoding, the solver-program interpreter (one lane per proof),
// and the constraint-matrix evaluation <A,w>, <B,w>, <C,w> with the satisfaction check.
//
// Replaces the two CPU stages the reference spawns per proof: `nargo execute` (ACVM witness solving,
// client/proof.helper.ts:55) and gnark's R1CS solver inside `sunspot prove` (client/proof.helper.ts:64),
// for the circuits of noir_circuit/src/main.nr:38-82 and scripts/generate_audit.py:405-463.
// Every lane of a wave executes the same instruction stream on its own proof (column p of W[.][P]), so
// there is no divergence and every witness access is a coalesced 2 KiB row segment.  Poseidon / Poseidon2
// permutations (main.nr:1-9, ct_helper/src/main.nr:15-34) run natively with the state in registers and
// emit the power wires of every S-box as they go (x^2, x^3, x^4, x^5).
#include "kernels.hpp"
#include "circuit.hpp"   // opcodes only
#include "poseidon29.hpp"
#include "lanes.hpp"
#include "gnark_hints.hpp"
#include "sha256.hpp"

namespace spp {

// terms first, first + step, ... of row k without its last k_end_skip terms (first = 0, step = 1: all of them)
__device__ __forceinline__ Fr dev_row_dot(const DevSparse& m, const Fr* __restrict__ coeffs, uint32_t k, uint32_t k_end_skip,
                                           const Fr* __restrict__ W, uint32_t P, uint32_t p, uint32_t first = 0, uint32_t step = 1) {
  Fr acc = Fr::zero();
  const uint32_t b = m.rowptr[k] + first, e = m.rowptr[k + 1] - k_end_skip;
  for (uint32_t t = b; t < e; t += step) {
    const uint32_t ci = m.coeff[t];
    Fr w = W[(size_t)m.wire[t] * P + p];
    if (ci & COEFF_ONE) acc = acc + w;
    else if (ci & COEFF_MINUS_ONE) acc = acc - w;
    else acc = acc + coeffs[ci & COEFF_MASK] * w;
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_load_inputs(const uint8_t* __restrict__ in_be, const uint8_t* __restrict__ rs_be,
                                                     Fr* __restrict__ W, uint32_t n_inputs, uint32_t n_wires, uint32_t P) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t total = (uint64_t)(n_inputs + 2) * P;
  if (g >= total) return;
  const uint32_t p = (uint32_t)(g % P), idx = (uint32_t)(g / P);
  uint8_t buf[32];
  if (idx < n_inputs) {
    const uint8_t* src = in_be + ((size_t)p * n_inputs + idx) * 32;
    for (int i = 0; i < 32; i++) buf[i] = src[i];
    W[(size_t)(1 + idx) * P + p] = Fr::from_bytes_be(buf);
    if (idx == 0) W[p] = Fr::one();
  } else if (idx == n_inputs) {
    // blinding r, s and their product: rows n_wires, n_wires+1, n_wires+2
    const uint8_t* src = rs_be + (size_t)p * 64;
    for (int i = 0; i < 32; i++) buf[i] = src[i];
    Fr r = Fr::from_bytes_be(buf);
    for (int i = 0; i < 32; i++) buf[i] = src[32 + i];
    Fr s = Fr::from_bytes_be(buf);
    W[(size_t)n_wires * P + p] = r;
    W[(size_t)(n_wires + 1) * P + p] = s;
    W[(size_t)(n_wires + 2) * P + p] = r * s;
  }
}
void launch_load_inputs(hipStream_t st, const uint8_t* d_inputs_be, const uint8_t* d_rs_be, Fr* W, uint32_t n_inputs, uint32_t n_wires,
                        uint32_t P) {
  uint64_t total = (uint64_t)(n_inputs + 2) * P;
  hipLaunchKernelGGL(k_load_inputs, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, d_inputs_be, d_rs_be, W, n_inputs, n_wires, P);
}

// ---------------------------------------------------------------------------------------------------
// native permutations
// ---------------------------------------------------------------------------------------------------
// Poseidon and Poseidon2 S-boxes: x^2, x^3, x^4, x^5 (csrc/circuit.cpp sbox5)
__device__ __forceinline__ Fr dev_sbox_emit(const Fr& x, Fr* __restrict__ W, uint32_t& out, uint32_t P, uint32_t p) {
  Fr x2 = x.sqr();
  Fr x3 = x2 * x;
  Fr x4 = x2.sqr();
  Fr x5 = x4 * x;
  W[(size_t)out * P + p] = x2;
  W[(size_t)(out + 1) * P + p] = x3;
  W[(size_t)(out + 2) * P + p] = x4;
  W[(size_t)(out + 3) * P + p] = x5;
  out += 4;
  return x5;
}

template <int T>
__device__ __noinline__ void dev_poseidon29(Fr (&st)[T], const Fr* __restrict__ rc, const uint32_t* __restrict__ mds29, int rp,
                                            Fr* __restrict__ W, uint32_t out, uint32_t P, uint32_t p) {
  poseidon_permute29<T, true>(st, rc, mds29, rp, [&](const F29<FrParams>& v) {
    Fr w;
    v.to_words(w.l);
    W[(size_t)out * P + p] = w;
    out++;
  });
}

template <int T>
__device__ __noinline__ void dev_poseidon(Fr (&s)[T], const Fr* __restrict__ rc, const Fr* __restrict__ mds, int rp, Fr* __restrict__ W,
                                          uint32_t out, uint32_t P, uint32_t p) {
  const int rf = 8;
#pragma unroll 1
  for (int r = 0; r < rf + rp; r++) {
    SPP_UNROLL for (int i = 0; i < T; i++) s[i] = s[i] + rc[r * T + i];
    const bool full = r < rf / 2 || r >= rf / 2 + rp;
    if (full) {
      SPP_UNROLL for (int i = 0; i < T; i++) s[i] = dev_sbox_emit(s[i], W, out, P, p);
    } else {
      s[0] = dev_sbox_emit(s[0], W, out, P, p);
    }
    Fr nx[T];
    SPP_UNROLL for (int i = 0; i < T; i++) {
      nx[i] = mds[i * T] * s[0];
      SPP_UNROLL for (int j = 1; j < T; j++) nx[i] = nx[i] + mds[i * T + j] * s[j];
    }
    SPP_UNROLL for (int i = 0; i < T; i++) s[i] = nx[i];
  }
}

__device__ __forceinline__ void dev_p2_external(Fr (&s)[4]) {
  // rows (5,7,1,3),(4,6,1,1),(1,3,5,7),(1,1,4,6)
  Fr t01 = s[0] + s[1], t23 = s[2] + s[3];
  Fr d0 = s[0].dbl(), d1 = s[1].dbl(), d2 = s[2].dbl(), d3 = s[3].dbl();
  Fr q0 = d0.dbl(), q1 = d1.dbl(), q2 = d2.dbl(), q3 = d3.dbl();
  Fr n0 = q0 + s[0] + q1 + d1 + s[1] + s[2] + d3 + s[3];          // 5a+7b+c+3d
  Fr n1 = q0 + q1 + d1 + t23;                                      // 4a+6b+c+d
  Fr n2 = s[0] + d1 + s[1] + q2 + s[2] + q3 + d3 + s[3];          // a+3b+5c+7d
  Fr n3 = t01 + q2 + q3 + d3;                                      // a+b+4c+6d
  s[0] = n0; s[1] = n1; s[2] = n2; s[3] = n3;
}

__device__ __noinline__ void dev_poseidon2(Fr (&s)[4], const Fr* __restrict__ rc, const Fr* __restrict__ mu, Fr* __restrict__ W,
                                           uint32_t out, uint32_t P, uint32_t p) {
  dev_p2_external(s);
  int k = 0;
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
    SPP_UNROLL for (int i = 0; i < 4; i++) s[i] = dev_sbox_emit(s[i] + rc[k + i], W, out, P, p);
    k += 4;
    dev_p2_external(s);
  }
#pragma unroll 1
  for (int r = 0; r < 56; r++) {
    s[0] = dev_sbox_emit(s[0] + rc[k], W, out, P, p);
    k++;
    Fr tot = s[0] + s[1] + s[2] + s[3];
    SPP_UNROLL for (int i = 0; i < 4; i++) s[i] = mu[i] * s[i] + tot;
  }
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
    SPP_UNROLL for (int i = 0; i < 4; i++) s[i] = dev_sbox_emit(s[i] + rc[k + i], W, out, P, p);
    k += 4;
    dev_p2_external(s);
  }
}

// ---------------------------------------------------------------------------------------------------
// solver interpreter
// ---------------------------------------------------------------------------------------------------
__device__ __noinline__ void dev_div_range(const DevCircuit& dc, Fr* __restrict__ W, Fr* __restrict__ scratch, uint32_t k0, uint32_t n,
                                           uint32_t P, uint32_t p) {
  Fr prod = Fr::one();
  for (uint32_t i = 0; i < n; i++) {
    Fr den = dev_row_dot(dc.B, dc.coeffs, k0 + i, 0, W, P, p);
    scratch[(size_t)i * P + p] = prod;
    if (!den.is_zero()) prod = prod * den;
  }
  Fr inv = prod.inv();
  for (uint32_t i = n; i-- > 0;) {
    const uint32_t k = k0 + i;
    const uint32_t out = dc.A.wire[dc.A.rowptr[k]];
    Fr den = dev_row_dot(dc.B, dc.coeffs, k, 0, W, P, p);
    if (den.is_zero()) {
      W[(size_t)out * P + p] = Fr::zero();
      continue;
    }
    Fr di = inv * scratch[(size_t)i * P + p];
    inv = inv * den;
    Fr num = dev_row_dot(dc.C, dc.coeffs, k, 0, W, P, p);
    W[(size_t)out * P + p] = num * di;
  }
}

// OP_GRUMPKIN: slopes of the affine ladder acc <- acc + T_j[digit_j] (acc_0 = O), then + N, over Grumpkin.
// The ladder is first walked in XYZZ coordinates (no inversions), every intermediate point is normalised
// with ONE shared inversion, and the 65 slope denominators with a second one.
__device__ __noinline__ void dev_grumpkin(const DevCircuit& dc, Fr* __restrict__ W, Fr* __restrict__ scratch, uint32_t bit0, uint32_t nbits,
                                          uint32_t aux_off, uint32_t nl, const uint32_t* __restrict__ lw, uint32_t P, uint32_t p) {
  const Fr* __restrict__ aux = dc.aux + aux_off;
  auto digit = [&](uint32_t j) {
    uint32_t d = 0;
    for (uint32_t k = 0; k < 4; k++) {
      const uint32_t bi = 4 * j + k;
      if (bi < nbits && !W[(size_t)(bit0 + bi) * P + p].is_zero()) d |= 1u << k;
    }
    return d;
  };
  auto sel = [&](uint32_t j) -> GkAffine {
    if (j >= 64) return {aux[2], aux[3]};
    const uint32_t o = 4 + (j * 16 + digit(j)) * 2;
    return {aux[o], aux[o + 1]};
  };
  auto row = [&](uint32_t r) -> Fr& { return scratch[(size_t)r * P + p]; };
  const GkAffine O{aux[0], aux[1]};
  GkXYZZ acc = GkXYZZ::from_affine(O);
  Fr prod = Fr::one();
  for (uint32_t j = 0; j < 64; j++) {
    acc.madd(sel(j));
    row(5 * j) = acc.X; row(5 * j + 1) = acc.Y; row(5 * j + 2) = acc.ZZ; row(5 * j + 3) = acc.ZZZ;
    row(5 * j + 4) = prod;
    prod = prod * (acc.ZZ * acc.ZZZ);
  }
  Fr inv = prod.inv();
  for (uint32_t j = 64; j-- > 0;) {
    Fr X = row(5 * j), Y = row(5 * j + 1), ZZ = row(5 * j + 2), ZZZ = row(5 * j + 3);
    Fr I = inv * row(5 * j + 4);
    inv = inv * (ZZ * ZZZ);
    row(5 * j) = X * (I * ZZZ);      // affine x of acc_j
    row(5 * j + 1) = Y * (I * ZZ);   // affine y of acc_j
  }
  prod = Fr::one();
  for (uint32_t j = 0; j < nl; j++) {
    Fr px = j == 0 ? O.x : row(5 * (j - 1));
    Fr den = sel(j).x - px;
    row(5 * j + 2) = prod;
    if (!den.is_zero()) prod = prod * den;
  }
  inv = prod.inv();
  for (uint32_t j = nl; j-- > 0;) {
    GkAffine s = sel(j);
    Fr px = j == 0 ? O.x : row(5 * (j - 1));
    Fr py = j == 0 ? O.y : row(5 * (j - 1) + 1);
    Fr den = s.x - px;
    Fr lam = Fr::zero();
    if (!den.is_zero()) {
      Fr di = inv * row(5 * j + 2);
      inv = inv * den;
      lam = (s.y - py) * di;
    }
    W[(size_t)lw[j] * P + p] = lam;
  }
}

// instructions [pc, pc_end) of the solver program for proof p, one lane
__device__ __forceinline__ void solve_range(const DevCircuit& dc, Fr* __restrict__ W, Fr* __restrict__ scratch, uint32_t pc, uint32_t pc_end,
                                            uint32_t P, uint32_t p) {
  const uint32_t* __restrict__ pr = dc.program;
  uint32_t last_k = 0xffffffffu;      // constraint whose B value is cached in last_b (dc.row_flags)
  Fr last_b = Fr::zero();
  while (pc < pc_end) {
    const uint32_t op = pr[pc];
    if (op == OP_END || op == OP_COMMIT) break;
    switch (op) {
      case OP_SOLVE_C: {
        const uint32_t k = pr[pc + 1];
        pc += 2;
        const uint32_t fl = dc.row_flags[k];
        Fr b = ((fl & 1) && last_k + 1 == k) ? last_b : dev_row_dot(dc.B, dc.coeffs, k, 0, W, P, p);
        Fr a = (fl & 2) ? b : dev_row_dot(dc.A, dc.coeffs, k, 0, W, P, p);
        last_k = k;
        last_b = b;
        Fr rest = dev_row_dot(dc.C, dc.coeffs, k, 1, W, P, p);
        const uint32_t out = dc.C.wire[dc.C.rowptr[k + 1] - 1];
        W[(size_t)out * P + p] = a * b - rest;
        break;
      }
      case OP_SOLVE_A: {
        dev_div_range(dc, W, scratch, pr[pc + 1], 1, P, p);
        pc += 2;
        break;
      }
      case OP_BATCH_DIV: {
        dev_div_range(dc, W, scratch, pr[pc + 1], pr[pc + 2], P, p);
        pc += 3;
        break;
      }
      case OP_BITS: {
        const uint32_t h = pr[pc + 1], nb = pr[pc + 2], out0 = pr[pc + 3];
        pc += 4;
        Fr v = dev_row_dot(dc.H, dc.coeffs, h, 0, W, P, p);
        uint32_t c[8];
        v.to_canonical(c);
        const Fr one = Fr::one(), zero = Fr::zero();
        uint32_t word = 0;
        for (uint32_t i = 0; i < nb; i++) {
          if ((i & 31) == 0) {
            // select limb i/32 without dynamic register indexing
            const uint32_t li = i >> 5;
            word = li == 0 ? c[0] : li == 1 ? c[1] : li == 2 ? c[2] : li == 3 ? c[3] : li == 4 ? c[4] : li == 5 ? c[5] : li == 6 ? c[6] : c[7];
          }
          W[(size_t)(out0 + i) * P + p] = (word & 1) ? one : zero;
          word >>= 1;
        }
        break;
      }
      case OP_INV_H: {       // unconstrained inverse hint (ACIR Brillig): 1 / <H_h,w>, 0 for 0
        const uint32_t h = pr[pc + 1], out = pr[pc + 2];
        pc += 3;
        Fr v = dev_row_dot(dc.H, dc.coeffs, h, 0, W, P, p);
        W[(size_t)out * P + p] = v.is_zero() ? Fr::zero() : v.inv();
        break;
      }
      case OP_LIMBS8: {
        const uint32_t h = pr[pc + 1], nl = pr[pc + 2], out0 = pr[pc + 3];
        pc += 4;
        Fr v = dev_row_dot(dc.H, dc.coeffs, h, 0, W, P, p);
        uint32_t c[8];
        v.to_canonical(c);
        uint32_t word = 0;
        for (uint32_t i = 0; i < nl; i++) {
          if ((i & 3) == 0) {
            const uint32_t li = i >> 2;
            word = li == 0 ? c[0] : li == 1 ? c[1] : li == 2 ? c[2] : li == 3 ? c[3] : li == 4 ? c[4] : li == 5 ? c[5] : li == 6 ? c[6] : c[7];
          }
          W[(size_t)(out0 + i) * P + p] = dc.byte_mont[word & 0xFF];
          word >>= 8;
        }
        break;
      }
      case OP_COUNT8: {
        const uint32_t h0 = pr[pc + 1], n = pr[pc + 2], out0 = pr[pc + 3];
        pc += 4;
        const Fr one = Fr::one();
        for (uint32_t j = 0; j < 256; j++) W[(size_t)(out0 + j) * P + p] = Fr::zero();
        for (uint32_t i = 0; i < n; i++) {
          Fr v = dev_row_dot(dc.H, dc.coeffs, h0 + i, 0, W, P, p);
          uint32_t c[8];
          v.to_canonical(c);
          if (c[0] < 256 && (c[1] | c[2] | c[3] | c[4] | c[5] | c[6] | c[7]) == 0) {
            Fr* slot = &W[(size_t)(out0 + c[0]) * P + p];
            *slot = *slot + one;
          }
        }
        break;
      }
      case OP_SOLVE_ROW: {   // one row of a decoded gnark system solved for its last term of side `side`
        const uint32_t k = pr[pc + 1], side = pr[pc + 2], inv_ci = pr[pc + 3], odiv_ci = pr[pc + 4];
        pc += 5;
        Fr v;
        uint32_t out;
        if (side == 2) {
          const Fr a = dev_row_dot(dc.A, dc.coeffs, k, 0, W, P, p), b = dev_row_dot(dc.B, dc.coeffs, k, 0, W, P, p);
          v = a * b - dev_row_dot(dc.C, dc.coeffs, k, 1, W, P, p);
          out = dc.C.wire[dc.C.rowptr[k + 1] - 1];
        } else {
          const DevSparse& mine = side == 0 ? dc.A : dc.B;
          const DevSparse& other = side == 0 ? dc.B : dc.A;
          const Fr c = dev_row_dot(dc.C, dc.coeffs, k, 0, W, P, p);
          Fr oi;
          if (odiv_ci != 0xffffffffu) oi = dc.coeffs[odiv_ci];
          else {
            const Fr o = dev_row_dot(other, dc.coeffs, k, 0, W, P, p);
            oi = o.is_zero() ? Fr::zero() : o.inv();
          }
          v = c * oi - dev_row_dot(mine, dc.coeffs, k, 1, W, P, p);
          out = mine.wire[mine.rowptr[k + 1] - 1];
        }
        if (inv_ci != 0xffffffffu) v = v * dc.coeffs[inv_ci];
        W[(size_t)out * P + p] = v;
        last_k = 0xffffffffu;
        break;
      }
      case OP_LIMBS: {
        const uint32_t h = pr[pc + 1], nl = pr[pc + 2], width = pr[pc + 3] & 0x7fffffffu, out0 = pr[pc + 4];
        const bool reversed = (pr[pc + 3] >> 31) != 0;      // limb i goes to out0 + n-1-i (a quotient stored in front of its remainder)
        pc += 5;
        uint32_t c[9];
        dev_row_dot(dc.H, dc.coeffs, h, 0, W, P, p).to_canonical(c);
        c[8] = 0;
        // the canonical words go through the lane's scratch row (dynamic word indexing; a handful of calls per proof)
        uint32_t* cw = reinterpret_cast<uint32_t*>(&scratch[(size_t)0 * P + p]);
        SPP_UNROLL for (int i = 0; i < 8; i++) cw[i] = c[i];
        for (uint32_t i = 0; i < nl; i++) {
          uint32_t o[8];
          SPP_UNROLL for (int j = 0; j < 8; j++) o[j] = 0;
          for (uint32_t j = 0; j < 4 && 32 * j < width; j++) {
            const uint32_t bit = i * width + 32 * j;
            uint32_t v = 0;
            if (bit < 256) {
              const uint32_t wi = bit >> 5, sh = bit & 31;
              v = cw[wi] >> sh;
              if (sh && wi + 1 < 8) v |= cw[wi + 1] << (32 - sh);
            }
            const uint32_t left = width - 32 * j;
            if (left < 32) v &= (1u << left) - 1u;
            o[j] = v;
          }
          W[(size_t)(out0 + (reversed ? nl - 1 - i : i)) * P + p] = Fr::from_canonical(o);
        }
        break;
      }
      case OP_COUNTN: {
        const uint32_t h0 = pr[pc + 1], n = pr[pc + 2], out0 = pr[pc + 3], size = pr[pc + 4];
        pc += 5;
        const Fr one = Fr::one();
        for (uint32_t j = 0; j < size; j++) W[(size_t)(out0 + j) * P + p] = Fr::zero();
        for (uint32_t i = 0; i < n; i++) {
          uint32_t c[8];
          dev_row_dot(dc.H, dc.coeffs, h0 + i, 0, W, P, p).to_canonical(c);
          if (c[0] < size && (c[1] | c[2] | c[3] | c[4] | c[5] | c[6] | c[7]) == 0) {
            Fr* slot = &W[(size_t)(out0 + c[0]) * P + p];
            *slot = *slot + one;
          }
        }
        break;
      }
      case OP_GK_MUL: {
        const uint32_t hl = pr[pc + 1], hh = pr[pc + 2], gy = pr[pc + 3], ox = pr[pc + 4], oy = pr[pc + 5], oi = pr[pc + 6];
        pc += 7;
        uint32_t lo[8], hi[8], k[8];
        dev_row_dot(dc.H, dc.coeffs, hl, 0, W, P, p).to_canonical(lo);
        dev_row_dot(dc.H, dc.coeffs, hh, 0, W, P, p).to_canonical(hi);
        SPP_UNROLL for (int i = 0; i < 4; i++) {
          k[i] = lo[i];
          k[4 + i] = hi[i];
        }
        Fr x = Fr::zero(), y = Fr::zero();
        const bool fin = dev_grumpkin_mul(k, dc.coeffs[gy], &x, &y);
        W[(size_t)ox * P + p] = x;
        W[(size_t)oy * P + p] = y;
        W[(size_t)oi * P + p] = fin ? Fr::zero() : Fr::one();
        break;
      }
      case OP_GLV: {
        const uint32_t h = pr[pc + 1], out0 = pr[pc + 2];
        const uint32_t* kc = pr + pc + 3;
        pc += 3 + 28;
        uint32_t c[8], s1[4], s2[4];
        dev_row_dot(dc.H, dc.coeffs, h, 0, W, P, p).to_canonical(c);
        dev_glv_split(kc, c, s1, s2);
        for (uint32_t i = 0; i < 2; i++) {
          uint32_t o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          o[0] = s1[2 * i]; o[1] = s1[2 * i + 1];
          W[(size_t)(out0 + i) * P + p] = Fr::from_canonical(o);
          o[0] = s2[2 * i]; o[1] = s2[2 * i + 1];
          W[(size_t)(out0 + 4 + i) * P + p] = Fr::from_canonical(o);
        }
        for (uint32_t i = 2; i < 4; i++) {   // limbs 2, 3 of a value below 2^127: zero
          W[(size_t)(out0 + i) * P + p] = Fr::zero();
          W[(size_t)(out0 + 4 + i) * P + p] = Fr::zero();
        }
        break;
      }
      case OP_EMUL: {
        const uint32_t h0 = pr[pc + 1], out0 = pr[pc + 2];
        const uint32_t* qc = pr + pc + 3;
        pc += 3 + 16;
        uint32_t a[6][8], kq[8], rem[8];
        for (uint32_t i = 0; i < 6; i++) dev_row_dot(dc.H, dc.coeffs, h0 + i, 0, W, P, p).to_canonical(a[i]);
        Big384 carry[6];
        dev_emulated_reduce<6, 6>(a, qc, kq, rem, carry);
        for (uint32_t i = 0; i < 4; i++) {
          uint32_t o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          o[0] = kq[2 * i]; o[1] = kq[2 * i + 1];
          W[(size_t)(out0 + i) * P + p] = Fr::from_canonical(o);
          o[0] = rem[2 * i]; o[1] = rem[2 * i + 1];
          W[(size_t)(out0 + 4 + i) * P + p] = Fr::from_canonical(o);
        }
        for (uint32_t i = 0; i < 6; i++) W[(size_t)(out0 + 8 + i) * P + p] = fr_from_bigs(carry[i]);
        break;
      }
      case OP_POSEIDON: {
        const uint32_t t = pr[pc + 1], h0 = pr[pc + 2], out0 = pr[pc + 3];
        pc += 4;
        if (t == 3) {
          Fr s[3];
          SPP_UNROLL for (int i = 0; i < 3; i++) s[i] = dev_row_dot(dc.H, dc.coeffs, h0 + i, 0, W, P, p);
          dev_poseidon29<3>(s, dc.pos3_rc, dc.pos3_mds29, 57, W, out0, P, p);
        } else {
          Fr s[5];
          SPP_UNROLL for (int i = 0; i < 5; i++) s[i] = dev_row_dot(dc.H, dc.coeffs, h0 + i, 0, W, P, p);
          dev_poseidon29<5>(s, dc.pos5_rc, dc.pos5_mds29, 60, W, out0, P, p);
        }
        break;
      }
      case OP_POSEIDON2: {
        const uint32_t h0 = pr[pc + 1], out0 = pr[pc + 2];
        pc += 3;
        Fr s[4];
        SPP_UNROLL for (int i = 0; i < 4; i++) s[i] = dev_row_dot(dc.H, dc.coeffs, h0 + i, 0, W, P, p);
        dev_poseidon2(s, dc.p2_rc, dc.p2_mu, W, out0, P, p);
        break;
      }
      case OP_MASK: {        // the commitment's random mask: fr.Hash(r || s, "spp-commit-mask1") of the proof's blinding factors (rows n_wires, n_wires + 1)
        const uint32_t out = pr[pc + 1];
        pc += 2;
        uint32_t m[16], cw[8];
        W[(size_t)dc.n_wires * P + p].to_canonical(cw);
        SPP_UNROLL for (int i = 0; i < 8; i++) m[i] = cw[7 - i];
        W[(size_t)(dc.n_wires + 1) * P + p].to_canonical(cw);
        SPP_UNROLL for (int i = 0; i < 8; i++) m[8 + i] = cw[7 - i];
        W[(size_t)out * P + p] = commitment_mask(m);
        break;
      }
      case OP_GRUMPKIN: {
        const uint32_t nl = pr[pc + 4];
        dev_grumpkin(dc, W, scratch, pr[pc + 1], pr[pc + 2], pr[pc + 3], nl, pr + pc + 5, P, p);
        pc += 5 + nl;
        break;
      }
      default:
        return;  // unknown opcode: leave the remaining wires zero -> unsatisfied
    }
  }
}
__global__ void __launch_bounds__(64) k_solve(DevCircuit dc, Fr* __restrict__ W, Fr* __restrict__ scratch, uint32_t pc, uint32_t pc_end,
                                              uint32_t P) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  solve_range(dc, W, scratch, pc, pc_end, P, p);
}
void launch_solve(hipStream_t st, DevCircuit dc, Fr* W, Fr* scratch, uint32_t pc_begin, uint32_t pc_end, uint32_t P) {
  if (pc_begin >= pc_end) return;
  hipLaunchKernelGGL(k_solve, dim3((P + 63) / 64), dim3(64), 0, st, dc, W, scratch, pc_begin, pc_end, P);
}

// ---------------------------------------------------------------------------------------------------
// cooperative solver for small batches: one 64-lane wave per proof
// ---------------------------------------------------------------------------------------------------
// With one lane per proof a single proof is one lane's serial work: 14 ms for the withdraw circuit, 29 ms for the audit
// circuit, three quarters of the drop-in generateProof latency, almost all of it inside the hash permutations (a lone wave
// issues one dependent multiplication after the other on one of the chip's 1024 SIMDs).  Here the 64 lanes of a wave belong
// to ONE proof and the host-built item list (spp_api.cpp, coop_plan) says how each stretch of the program uses them:
//   COOP_SEQ       [pc_a, pc_b) on lane 0 (whatever has no parallel form)
//   COOP_PAR       independent BITS / LIMBS8 / INV_H instructions, one per lane
//   COOP_LEVELS    a run of SOLVE_C rows in dependency levels: the rows of a level are solved by different lanes
//   COOP_POSEIDON  one permutation: state words, the two halves of every S-box (x^3 | x^4) and the MDS products on
//   COOP_POSEIDON2 different lanes -- three dependent multiplications per partial round instead of 13 (t = 3) / 8
//   COOP_GRUMPKIN  the fixed-base ladder as a 64-lane prefix sum, one window per lane
// The values written are the same field elements as the one-lane solver's (the words may be another representative < 2p).
__device__ __forceinline__ void emit4(Fr* __restrict__ W, uint32_t out, uint32_t P, uint32_t p, const Fr& x2, const Fr& x3, const Fr& x4,
                                      const Fr& x5) {
  W[(size_t)out * P + p] = x2;
  W[(size_t)(out + 1) * P + p] = x3;
  W[(size_t)(out + 2) * P + p] = x4;
  W[(size_t)(out + 3) * P + p] = x5;
}

// Poseidon2 (t = 4) in lane-parallel form: lanes.hpp, coop_p2_permute; the S-box powers go to the witness as they appear
__device__ __noinline__ void coop_poseidon2(const DevCircuit& dc, Fr* __restrict__ W, uint32_t h0, uint32_t out, uint32_t P, uint32_t p,
                                            uint32_t lane) {
  Fr s = Fr::zero();
  if (lane < 4) s = dev_row_dot(dc.H, dc.coeffs, h0 + lane, 0, W, P, p);
  coop_p2_permute(dc.p2_rc, dc.p2_mu, s, lane, [&](uint32_t o, const Fr& x2, const Fr& x3, const Fr& x4, const Fr& x5) {
    emit4(W, out + o, P, p, x2, x3, x4, x5);
  });
}

// Poseidon (t = 3, 5): lanes 0..T-1 hold the state; lanes 8+i the x^4 halves (full rounds) or M[i][0] * x (partial rounds);
// lanes 16 + i*T + j the MDS products M[i][j] * s_j.  In a partial round the products with j >= 1 do not wait for the S-box.
template <int T>
__device__ __noinline__ void coop_poseidon(const DevCircuit& dc, const Fr* __restrict__ rc, const Fr* __restrict__ mds, int rp,
                                           Fr* __restrict__ W, uint32_t h0, uint32_t out, uint32_t P, uint32_t p, uint32_t lane) {
  const uint32_t li = lane < T ? lane : 0;                       // state index of this lane (clamped)
  const bool is_y = lane >= 8 && lane < 8 + T;
  const uint32_t yi = is_y ? lane - 8 : 0;
  const bool is_q = lane >= 16 && lane < 16 + T * T;
  const uint32_t q = is_q ? lane - 16 : 0, qj = q % T;
  const Fr mq = mds[q];                                          // M[qi][qj]
  const Fr my = mds[yi * T];                                     // M[yi][0]
  Fr s = Fr::zero();
  if (lane < T) s = dev_row_dot(dc.H, dc.coeffs, h0 + lane, 0, W, P, p);
#pragma unroll 1
  for (int r = 0; r < 8 + rp; r++) {
    s = s + rc[r * T + li];
    const bool full = r < 4 || r >= 4 + rp;
    if (full) {
      const Fr x2 = s * s;
      const Fr t = lane_get(x2, is_y ? yi : li);
      const Fr R = t * lane_sel(lane < 8, s, t);                 // lanes < T: x^3 ; lanes 8+i: x^4
      const Fr x4 = lane_get(R, li + 8);
      const Fr x5 = x4 * s;
      if (lane < T) emit4(W, out + 4 * lane, P, p, x2, R, x4, x5);
      out += 4 * T;
      const Fr prod = mq * lane_get(x5, qj);
      Fr acc = lane_get(prod, 16 + li * T);
      SPP_UNROLL for (int j = 1; j < T; j++) acc = acc + lane_get(prod, 16 + li * T + j);
      s = acc;
    } else {
      const Fr x0 = lane_bcast<0>(s);
      const Fr sj = lane_get(s, qj);
      // lane 0: x * x | lanes 8+i: M[i][0] * x | lanes 16+i*T+j: M[i][j] * s_j
      const Fr R1 = lane_sel(lane == 0, x0, lane_sel(is_y, my, mq)) * lane_sel(lane == 0 || is_y, x0, sj);
      const Fr x2 = lane_bcast<0>(R1);
      const Fr R2 = x2 * lane_sel(lane == 0, x0, x2);            // lane 0: x^3 ; lane 1: x^4
      const Fr x4 = lane_bcast<1>(R2);
      const Fr R3 = x4 * lane_sel(lane == 0, x0, R1);            // lane 0: x^5 ; lanes 8+i: M[i][0] * x^5
      if (lane == 0) emit4(W, out, P, p, R1, R2, x4, R3);
      out += 4;
      Fr acc = lane_get(R3, 8 + li);
      SPP_UNROLL for (int j = 1; j < T; j++) acc = acc + lane_get(R1, 16 + li * T + j);
      s = acc;
    }
  }
}

// OP_GRUMPKIN on 64 lanes: lane j looks up the table point of window j, a shuffle scan (six XYZZ additions) leaves
// acc_j = O + T_0[d_0] + ... + T_j[d_j] in lane j, every lane normalises its own point and divides for its own slope
// (the one-lane form walks the 64 additions and two batch inversions in sequence: 1.7 ms of a proof).
__device__ __noinline__ void coop_grumpkin(const DevCircuit& dc, Fr* __restrict__ W, const uint32_t* __restrict__ op, uint32_t P, uint32_t p,
                                           uint32_t lane) {
  const uint32_t bit0 = op[1], nbits = op[2], nl = op[4];
  const uint32_t* __restrict__ lw = op + 5;
  const Fr* __restrict__ aux = dc.aux + op[3];
  uint32_t d = 0;
  for (uint32_t k = 0; k < 4; k++) {
    const uint32_t bi = 4 * lane + k;
    if (bi < nbits && !W[(size_t)(bit0 + bi) * P + p].is_zero()) d |= 1u << k;
  }
  const GkAffine mine{aux[4 + (lane * 16 + d) * 2], aux[4 + (lane * 16 + d) * 2 + 1]};
  const GkAffine O{aux[0], aux[1]}, N{aux[2], aux[3]};
  GkXYZZ acc = GkXYZZ::from_affine(lane == 0 ? O : mine);
  if (lane == 0) acc.madd(mine);
#pragma unroll 1
  for (uint32_t off = 1; off < 64; off <<= 1) {
    const uint32_t src = lane >= off ? lane - off : lane;
    GkXYZZ o;
    o.X = lane_get(acc.X, src);
    o.Y = lane_get(acc.Y, src);
    o.ZZ = lane_get(acc.ZZ, src);
    o.ZZZ = lane_get(acc.ZZZ, src);
    if (lane >= off) acc.add(o);
  }
  const GkAffine a = acc.to_affine();
  GkAffine prev{lane_get(a.x, lane ? lane - 1 : 0), lane_get(a.y, lane ? lane - 1 : 0)};
  if (lane == 0) prev = O;
  const Fr den = mine.x - prev.x;
  if (lane < nl) W[(size_t)lw[lane] * P + p] = (mine.y - prev.y) * den.inv();   // inv(0) = 0: slope 0, as the one-lane form
  if (lane == 63 && nl > 64) {
    const Fr den2 = N.x - a.x;
    W[(size_t)lw[64] * P + p] = (N.y - a.y) * den2.inv();
  }
}

__device__ __forceinline__ void solve_c_row(const DevCircuit& dc, Fr* __restrict__ W, uint32_t k, uint32_t P, uint32_t p) {
  const Fr b = dev_row_dot(dc.B, dc.coeffs, k, 0, W, P, p);
  const Fr a = (dc.row_flags[k] & 2) ? b : dev_row_dot(dc.A, dc.coeffs, k, 0, W, P, p);
  const Fr rest = dev_row_dot(dc.C, dc.coeffs, k, 1, W, P, p);
  const uint32_t out = dc.C.wire[dc.C.rowptr[k + 1] - 1];
  W[(size_t)out * P + p] = a * b - rest;
}

__global__ void __launch_bounds__(64) k_solve_coop(DevCircuit dc, DevCoop co, Fr* __restrict__ W, Fr* __restrict__ scratch, CoopTracks tracks,
                                                   uint32_t P) {
  __shared__ uint32_t lvl_buf[2][COOP_CHUNK];
  const uint32_t p = blockIdx.x, lane = threadIdx.x;
  uint32_t item = tracks.begin[blockIdx.y];
  const uint32_t item_end = tracks.end[blockIdx.y];
  const uint32_t* __restrict__ pr = dc.program;
  for (; item < item_end; item++) {
    const uint32_t kind = co.items[3 * item], a = co.items[3 * item + 1], b = co.items[3 * item + 2];
    switch (kind) {
      case COOP_SEQ:
      case COOP_PAR: {
        // SEQ: one range, lane 0.  PAR: ranges par[2i], par[2i+1] for i in [a, b), one per lane
        const uint32_t n = kind == COOP_SEQ ? 1 : b - a;
        for (uint32_t i = lane; i < n; i += 64) {
          const uint32_t pa = kind == COOP_SEQ ? a : co.par[2 * (a + i)], pb = kind == COOP_SEQ ? b : co.par[2 * (a + i) + 1];
          solve_range(dc, W, scratch, pa, pb, P, p);
        }
        break;
      }
      case COOP_LEVELS:
        for (uint32_t lv = a; lv < b; lv++) {
          const uint32_t r0 = co.lvl_ptr[lv], r1 = co.lvl_ptr[lv + 1], nr = r1 - r0;
          // G lanes per row (a power of two, <= 16): the terms of the three linear forms are dealt over them and the partial
          // sums joined by shuffles -- a level of a compiled (ACIR) program has two or three rows of 40-term forms
          uint32_t G = 1;
          while (G < 16 && nr * (G * 2) <= 64) G *= 2;
          const uint32_t per_pass = 64 / G, sub = lane % G;
          for (uint32_t r = r0; r < r1; r += per_pass) {           // uniform trip count: every lane reaches the shuffles
            const uint32_t mine = r + lane / G;
            const bool live = mine < r1;
            const uint32_t k = live ? co.lvl_rows[mine] : 0;
            Fr bv = Fr::zero(), av = Fr::zero(), rest = Fr::zero();
            const bool square = live && (dc.row_flags[k] & 2);
            if (live) {
              bv = dev_row_dot(dc.B, dc.coeffs, k, 0, W, P, p, sub, G);
              if (!square) av = dev_row_dot(dc.A, dc.coeffs, k, 0, W, P, p, sub, G);
              rest = dev_row_dot(dc.C, dc.coeffs, k, 1, W, P, p, sub, G);
            }
            for (uint32_t off = G >> 1; off >= 1; off >>= 1) {
              bv = bv + lane_get(bv, lane ^ off);
              av = av + lane_get(av, lane ^ off);
              rest = rest + lane_get(rest, lane ^ off);
            }
            if (live && sub == 0) {
              if (square) av = bv;
              W[(size_t)dc.C.wire[dc.C.rowptr[k + 1] - 1] * P + p] = av * bv - rest;
            }
          }
          __syncthreads();
        }
        break;
      case COOP_LEVEL_STREAM: {
        // A level of a compiled program is one to three short rows, and 4 800 levels follow one another: what a level costs is
        // the chain of dependent loads (level table -> row list -> row pointers -> term indices -> witness).  Here everything but
        // the witness comes out of LDS, staged one chunk ahead.
        const uint32_t* __restrict__ stream = co.lvl_stream + (size_t)a * COOP_CHUNK;
        for (uint32_t i = lane; i < COOP_CHUNK; i += 64) lvl_buf[0][i] = stream[i];
        __syncthreads();
        for (uint32_t ch = 0; ch < b; ch++) {
          uint32_t pre[COOP_CHUNK / 64];
          const bool more = ch + 1 < b;
          if (more) {
            SPP_UNROLL for (uint32_t j = 0; j < COOP_CHUNK / 64; j++) pre[j] = stream[(size_t)(ch + 1) * COOP_CHUNK + lane + 64 * j];
          }
          const uint32_t* sb = lvl_buf[ch & 1];
          uint32_t pos = 0;
          while (pos < COOP_CHUNK) {
            const uint32_t nr = sb[pos] & 0xffffffu;
            if (sb[pos] == 0xffffffffu) break;
            const uint32_t G = (sb[pos] >> 24) & 31u;   // lanes per row, chosen by the host from the longest form of the level
            const uint32_t per_pass = 64 / G, sub = lane % G;
            for (uint32_t r = 0; r < nr; r += per_pass) {
              const uint32_t mine = r + lane / G;
              const bool live = mine < nr;
              const uint32_t base = pos + (live ? sb[pos + 2 + mine] : 2 + nr);
              const uint32_t out = live ? sb[base] : 0, na_w = live ? sb[base + 1] : 0;
              const uint32_t nA = na_w & 0x7fffffffu, nB = live ? sb[base + 2] : 0, nC = live ? sb[base + 3] : 0;
              auto dot = [&](uint32_t start, uint32_t n) {
                Fr acc = Fr::zero();
                for (uint32_t t = sub; t < n; t += G) {
                  const uint32_t wi = sb[start + 2 * t], cw = sb[start + 2 * t + 1];
                  const Fr w = W[(size_t)wi * P + p];
                  if (cw & COEFF_ONE) acc = acc + w;
                  else if (cw & COEFF_MINUS_ONE) acc = acc - w;
                  else acc = acc + dc.coeffs[cw & COEFF_MASK] * w;
                }
                return acc;
              };
              Fr av = dot(base + 4, nA), bv = dot(base + 4 + 2 * nA, nB), rest = dot(base + 4 + 2 * (nA + nB), nC);
              for (uint32_t off = G >> 1; off >= 1; off >>= 1) {
                bv = bv + lane_get(bv, lane ^ off);
                av = av + lane_get(av, lane ^ off);
                rest = rest + lane_get(rest, lane ^ off);
              }
              if (live && sub == 0) {
                if (na_w >> 31) av = bv;
                W[(size_t)out * P + p] = av * bv - rest;
              }
            }
            pos += sb[pos + 1];
            __syncthreads();
          }
          if (more) {
            SPP_UNROLL for (uint32_t j = 0; j < COOP_CHUNK / 64; j++) lvl_buf[(ch + 1) & 1][lane + 64 * j] = pre[j];
          }
          __syncthreads();
        }
        break;
      }
      case COOP_POSEIDON: {
        const uint32_t t = pr[a + 1], h0 = pr[a + 2], out0 = pr[a + 3];
        if (t == 3) coop_poseidon<3>(dc, dc.pos3_rc, dc.pos3_mds, 57, W, h0, out0, P, p, lane);
        else coop_poseidon<5>(dc, dc.pos5_rc, dc.pos5_mds, 60, W, h0, out0, P, p, lane);
        break;
      }
      case COOP_POSEIDON2:
        coop_poseidon2(dc, W, pr[a + 1], pr[a + 2], P, p, lane);
        break;
      case COOP_GRUMPKIN:
        coop_grumpkin(dc, W, pr + a, P, p, lane);
        break;
      default:
        return;
    }
    __syncthreads();   // the next item reads wires other lanes have just written
  }
}
void launch_solve_coop(hipStream_t st, DevCircuit dc, DevCoop co, Fr* W, Fr* scratch, CoopTracks tracks, uint32_t P) {
  if (tracks.n == 0 || P == 0) return;
  hipLaunchKernelGGL(k_solve_coop, dim3(P, tracks.n), dim3(64), 0, st, dc, co, W, scratch, tracks, P);
}

// ---------------------------------------------------------------------------------------------------
// wide OP_BATCH_DIV: lane -> (chunk of DIV_CHUNK consecutive constraints, proof); one inversion per chunk
// ---------------------------------------------------------------------------------------------------
// (chunk = 32 for batches: one inversion per 32 divisions; 4 for small batches, where the lanes are free and the chain of
// prefix products is what a proof waits for)
__global__ void __launch_bounds__(64) k_batch_div(DevCircuit dc, Fr* __restrict__ W, Fr* __restrict__ scratch, uint32_t k0, uint32_t n,
                                                  uint32_t P, uint32_t chunk_len) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nchunks = (n + chunk_len - 1) / chunk_len;
  if (g >= (uint64_t)nchunks * P) return;
  const uint32_t p = (uint32_t)(g % P), chunk = (uint32_t)(g / P);
  const uint32_t i0 = chunk * chunk_len;
  const uint32_t cnt = n - i0 < chunk_len ? n - i0 : chunk_len;
  dev_div_range(dc, W, scratch + (size_t)i0 * P, k0 + i0, cnt, P, p);
}
// small batches: one division per lane, ONE inversion per wave -- the 64 denominators of a wave (whatever proofs they belong to)
// are inverted together: prefix and suffix products by shuffle scans (6 + 6 dependent products), lane 0 inverts the product of
// all, inv(d_i) = prefix_{i-1} * suffix_{i+1} * inv(all).  (64 lanes running the binary inversion each diverge on every step:
// 0.25 ms for the 264 divisions of one withdraw proof; this form is ~50 us.)
__global__ void __launch_bounds__(64) k_batch_div_wave(DevCircuit dc, Fr* __restrict__ W, uint32_t k0, uint32_t n, uint32_t P) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x;
  const bool live = g < (uint64_t)n * P;
  const uint32_t p = live ? (uint32_t)(g % P) : 0, k = k0 + (live ? (uint32_t)(g / P) : 0);
  Fr den = Fr::one();
  bool zero = true;
  if (live) {
    den = dev_row_dot(dc.B, dc.coeffs, k, 0, W, P, p);
    zero = den.is_zero();
    if (zero) den = Fr::one();
  }
  Fr pre = den, suf = den;                       // inclusive prefix / suffix products over the wave
  SPP_UNROLL for (uint32_t off = 1; off < 64; off <<= 1) {
    const Fr a = lane_get(pre, lane >= off ? lane - off : lane);
    const Fr b = lane_get(suf, lane + off < 64 ? lane + off : lane);
    if (lane >= off) pre = pre * a;
    if (lane + off < 64) suf = suf * b;
  }
  Fr inv_all = Fr::zero();
  if (lane == 63) inv_all = pre.inv();           // pre of lane 63 = product of all 64
  inv_all = lane_bcast<63>(inv_all);
  const Fr before = lane_get(pre, lane ? lane - 1 : 0), after = lane_get(suf, lane < 63 ? lane + 1 : 63);
  Fr inv = inv_all;
  if (lane > 0) inv = inv * before;
  if (lane < 63) inv = inv * after;
  if (!live) return;
  const uint32_t out = dc.A.wire[dc.A.rowptr[k]];
  W[(size_t)out * P + p] = zero ? Fr::zero() : dev_row_dot(dc.C, dc.coeffs, k, 0, W, P, p) * inv;
}
void launch_batch_div(hipStream_t st, DevCircuit dc, Fr* W, Fr* scratch, uint32_t k0, uint32_t n, uint32_t P) {
  if (n == 0) return;
  if ((uint64_t)n * P <= 16384) {
    hipLaunchKernelGGL(k_batch_div_wave, dim3((uint32_t)(((uint64_t)n * P + 63) / 64)), dim3(64), 0, st, dc, W, k0, n, P);
    return;
  }
  const uint32_t chunk_len = P >= 256 ? 32 : 4;
  uint64_t lanes = (uint64_t)((n + chunk_len - 1) / chunk_len) * P;
  hipLaunchKernelGGL(k_batch_div, dim3((uint32_t)((lanes + 63) / 64)), dim3(64), 0, st, dc, W, scratch, k0, n, P, chunk_len);
}

// ---------------------------------------------------------------------------------------------------
// wide OP_COUNT8: histogram of n looked-up values per proof with integer atomics, then conversion to Fr
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_count8_hist(DevCircuit dc, const Fr* __restrict__ W, uint32_t* __restrict__ counters, uint32_t h0,
                                                     uint32_t n, uint32_t P) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (uint64_t)n * P) return;
  const uint32_t p = (uint32_t)(g % P), i = (uint32_t)(g / P);
  Fr v = dev_row_dot(dc.H, dc.coeffs, h0 + i, 0, W, P, p);
  uint32_t c[8];
  v.to_canonical(c);
  if (c[0] < 256 && (c[1] | c[2] | c[3] | c[4] | c[5] | c[6] | c[7]) == 0) atomicAdd(&counters[(size_t)c[0] * P + p], 1u);
}
__global__ void __launch_bounds__(256) k_count8_store(const uint32_t* __restrict__ counters, Fr* __restrict__ W, uint32_t out0, uint32_t P) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= 256 * P) return;
  W[(size_t)out0 * P + g] = Fr::from_u64(counters[g]);
}
void launch_count8(hipStream_t st, DevCircuit dc, Fr* W, uint32_t* counters, uint32_t h0, uint32_t n, uint32_t out0, uint32_t P) {
  (void)hipMemsetAsync(counters, 0, sizeof(uint32_t) * 256 * (size_t)P, st);
  uint64_t lanes = (uint64_t)n * P;
  if (lanes) hipLaunchKernelGGL(k_count8_hist, dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, st, dc, W, counters, h0, n, P);
  hipLaunchKernelGGL(k_count8_store, dim3((256 * P + 255) / 256), dim3(256), 0, st, counters, W, out0, P);
}

// ---------------------------------------------------------------------------------------------------
// constraint evaluation + satisfaction check: lane -> (constraint k, proof p)
// ---------------------------------------------------------------------------------------------------
// Row dot product for the matrix evaluation.  Terms whose coefficient is a small integer (the audit circuit's 1 088
// quotient rows carry 1 024 public-key coefficients < 2^28 each, generate_audit.py:57-66,236-243) are summed as plain
// integers, c * (w*R) into a 320-bit accumulator (8 multiply-adds per term instead of a ~300-instruction Montgomery
// product), positives and negatives apart; the two sums are reduced once per row: the element whose Montgomery word is
// X = X_lo + 2^256 * X_hi is elem(X_lo mod p) + X_hi.
__device__ __forceinline__ void wide_mac(uint32_t (&acc)[10], uint32_t c, const Fr& x) {
  uint32_t carry = 0;
  SPP_UNROLL for (int i = 0; i < 8; i++) {
    const uint64_t t = (uint64_t)c * x.l[i] + acc[i] + carry;
    acc[i] = (uint32_t)t;
    carry = (uint32_t)(t >> 32);
  }
  const uint64_t t = (uint64_t)acc[8] + carry;
  acc[8] = (uint32_t)t;
  acc[9] += (uint32_t)(t >> 32);
}
__device__ __forceinline__ Fr wide_finish(const uint32_t (&acc)[10]) {
  Fr lo;
  SPP_UNROLL for (int i = 0; i < 8; i++) lo.l[i] = acc[i];
  Fr::cond_sub_2p(lo.l);   // < 2^256 < 6p  ->  < 4p  ->  < 2p
  Fr::cond_sub_2p(lo.l);
  if ((acc[8] | acc[9]) == 0) return lo;
  return lo + Fr::from_u64((uint64_t)acc[8] | ((uint64_t)acc[9] << 32));
}
// terms first, first + step, ... of row k (first = 0, step = 1: the whole row)
__device__ __forceinline__ Fr dev_row_dot_wide(const DevSparse& m, const Fr* __restrict__ coeffs, uint32_t k, const Fr* __restrict__ W,
                                                uint32_t P, uint32_t p, uint32_t first = 0, uint32_t step = 1) {
  Fr acc = Fr::zero();
  uint32_t pos[10], neg[10];
  SPP_UNROLL for (int i = 0; i < 10; i++) pos[i] = neg[i] = 0;
  bool any = false;
  const uint32_t b = m.rowptr[k] + first, e = m.rowptr[k + 1];
  for (uint32_t t = b; t < e; t += step) {
    const uint32_t ci = m.coeff[t], li = m.lit[t];
    const Fr w = W[(size_t)m.wire[t] * P + p];
    if (ci & COEFF_ONE) acc = acc + w;
    else if (ci & COEFF_MINUS_ONE) acc = acc - w;
    else if (li) {
      any = true;
      if (li & 0x80000000u) wide_mac(neg, li & 0x7fffffffu, w);
      else wide_mac(pos, li, w);
    } else acc = acc + coeffs[ci & COEFF_MASK] * w;
  }
  if (any) acc = acc + wide_finish(pos) - wide_finish(neg);
  return acc;
}

// ---- small rows (see DevCircuit) --------------------------------------------------------------------------------------------
// slot s, proof p: the wire's value as a small signed integer.  A value outside the range the lookup argument allows cannot
// satisfy the circuit (its lookup row fails): the proof is refused here already, and the rows built on it may hold anything.
__global__ void __launch_bounds__(256) k_small_extract(DevCircuit dc, const Fr* __restrict__ W, int16_t* __restrict__ small, uint32_t P,
                                                       uint32_t* __restrict__ status) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (uint64_t)dc.sm_nslots * P) return;
  const uint32_t p = (uint32_t)(g % P), s = (uint32_t)(g / P);
  uint32_t c[8], n[8];
  W[(size_t)dc.sm_wires[s] * P + p].to_canonical(c);
  canonical_negate<FrParams>(c, n);
  const bool pos = (c[1] | c[2] | c[3] | c[4] | c[5] | c[6] | c[7]) == 0 && c[0] < 32768u;
  const bool neg = (n[1] | n[2] | n[3] | n[4] | n[5] | n[6] | n[7]) == 0 && n[0] <= 32768u && (c[0] | c[1] | c[2] | c[3] | c[4] | c[5] | c[6] | c[7]) != 0;
  int v = pos ? (int)c[0] : neg ? -(int)n[0] : 0;
  const int lo = dc.sm_lo[s];
  if ((!pos && !neg) || v < lo || v > lo + 255) {
    atomicOr(&status[p], 1u);
    v = 0;
  }
  small[g] = (int16_t)v;
}
// small row r, proof p: sum of coefficient * small value as a 64-bit integer (|sum| < 2^31 * 2^15 * terms), stored as a field
// element where k_spmv_check expects the row's value
__global__ void __launch_bounds__(256) k_spmv_small_rows(DevCircuit dc, const int16_t* __restrict__ small, const Fr* __restrict__ W,
                                                         Fr* __restrict__ abc, uint32_t n, uint32_t P) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (uint64_t)dc.sm_nrows * P) return;
  const uint32_t p = (uint32_t)(g % P), r = (uint32_t)(g / P);
  const uint32_t b = dc.sm_rowptr[r], e = dc.sm_rowptr[r + 1];
  long long acc = 0;
  for (uint32_t t = b; t < e; t++) acc += (long long)dc.sm_coef[t] * (long long)small[(size_t)dc.sm_slot[t] * P + p];
  const uint32_t ro = dc.sm_row_out[r], mat = ro >> 30, k = ro & 0x3fffffffu;
  const Fr mag = Fr::from_u64((uint64_t)(acc < 0 ? -acc : acc));
  Fr v = acc < 0 ? mag.neg() : mag;
  for (uint32_t t = dc.sm_rest_ptr[r]; t < dc.sm_rest_ptr[r + 1]; t++) v = v + dc.coeffs[dc.sm_rest_coeff[t]] * W[(size_t)dc.sm_rest_wire[t] * P + p];
  abc[(size_t)mat * n * P + (size_t)k * P + p] = v;
}

// lane -> (run r of constraints with one shared B row, proof p); lanes past the last run zero-fill the padding rows
// n_constraints .. n-1 of the evaluation domain
__global__ void __launch_bounds__(256) k_spmv_check(DevCircuit dc, const Fr* __restrict__ W, Fr* __restrict__ abc, uint32_t n, uint32_t P,
                                                    uint32_t* __restrict__ status) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t lanes = (uint64_t)(dc.n_runs + (n - dc.n_constraints)) * P;
  if (g >= lanes) return;
  const uint32_t p = (uint32_t)(g % P), r = (uint32_t)(g / P);
  const uint64_t total = (uint64_t)n * P;
  if (r >= dc.n_runs) {
    const uint64_t o = (uint64_t)(dc.n_constraints + (r - dc.n_runs)) * P + p;
    abc[o] = Fr::zero();
    abc[total + o] = Fr::zero();
    abc[2 * total + o] = Fr::zero();
    return;
  }
  const uint32_t k0 = dc.run_start[r], k1 = dc.run_start[r + 1];
  const uint8_t* __restrict__ rs = dc.row_small;
  // the rows of a run share their B row; a small B row was stored for the run's first constraint by k_spmv_small_rows
  const Fr b = (rs && (rs[k0] & 2)) ? abc[total + (uint64_t)k0 * P + p] : dev_row_dot_wide(dc.B, dc.coeffs, k0, W, P, p);
  bool bad = false;
  for (uint32_t k = k0; k < k1; k++) {
    const uint64_t o = (uint64_t)k * P + p;
    const uint32_t f = rs ? rs[k] : 0u;
    const Fr a = (f & 1) ? abc[o] : dev_row_dot_wide(dc.A, dc.coeffs, k, W, P, p);
    const Fr c = (f & 4) ? abc[2 * total + o] : dev_row_dot_wide(dc.C, dc.coeffs, k, W, P, p);
    bad |= (a * b != c);
    abc[o] = a;
    abc[total + o] = b;
    abc[2 * total + o] = c;
  }
  if (bad) atomicOr(&status[p], 1u);
}
// long rows (DevCircuit::lg_rows): workgroup = 64 proofs x 16 term classes; lane (class s, proof p) sums the terms s, s + 16, ...
// of the row with coalesced witness reads, the 16 partial sums meet in LDS.  blockIdx.y = long row.
static constexpr uint32_t LONG_G = 16;
__global__ void __launch_bounds__(1024) k_spmv_long_rows(DevCircuit dc, const Fr* __restrict__ W, Fr* __restrict__ abc, uint32_t n, uint32_t P) {
  __shared__ Fr part[LONG_G][64];
  const uint32_t lane = threadIdx.x & 63, sub = threadIdx.x >> 6, p = blockIdx.x * 64 + lane;
  const uint32_t code = dc.lg_rows[blockIdx.y], mi = code >> 30, k = code & 0x3fffffffu;
  const DevSparse& m = mi == 0 ? dc.A : mi == 1 ? dc.B : dc.C;
  part[sub][lane] = p < P ? dev_row_dot_wide(m, dc.coeffs, k, W, P, p, sub, LONG_G) : Fr::zero();
  __syncthreads();
  if (sub != 0 || p >= P) return;
  Fr acc = part[0][lane];
  SPP_UNROLL for (uint32_t q = 1; q < LONG_G; q++) acc = acc + part[q][lane];
  abc[(uint64_t)mi * n * P + (uint64_t)k * P + p] = acc;
}

// small batches: SPMV_G lanes -> (constraint k, proof p), each lane takes every SPMV_G-th term of the three rows and the partial
// sums meet through wave shuffles.  The run kernel above saves the repeated B evaluations of a batch, but a run of thousands of
// constraints, or one row of 6 720 terms (the lookup sum of the audit circuit), is one lane's serial work there: 4.6 ms for a
// single audit proof.
static constexpr uint32_t SPMV_G = 16;
__device__ __forceinline__ Fr group_sum(Fr v) {
  SPP_UNROLL for (uint32_t off = SPMV_G / 2; off >= 1; off >>= 1) {
    Fr o;
    SPP_UNROLL for (int i = 0; i < 8; i++) o.l[i] = __shfl_xor(v.l[i], off);
    v = v + o;
  }
  return v;
}
__global__ void __launch_bounds__(256) k_spmv_check_rows(DevCircuit dc, const Fr* __restrict__ W, Fr* __restrict__ abc, uint32_t n, uint32_t P,
                                                         uint32_t* __restrict__ status) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t total = (uint64_t)n * P;
  const uint64_t e = g / SPMV_G;           // element (k, p); the grid is padded to whole groups, every lane reaches the shuffles
  const uint32_t sub = (uint32_t)(g % SPMV_G);
  const bool live = e < total;
  const uint32_t p = live ? (uint32_t)(e % P) : 0, k = live ? (uint32_t)(e / P) : dc.n_constraints;
  Fr a = Fr::zero(), b = Fr::zero(), c = Fr::zero();
  if (k < dc.n_constraints) {
    a = dev_row_dot_wide(dc.A, dc.coeffs, k, W, P, p, sub, SPMV_G);
    b = dev_row_dot_wide(dc.B, dc.coeffs, k, W, P, p, sub, SPMV_G);
    c = dev_row_dot_wide(dc.C, dc.coeffs, k, W, P, p, sub, SPMV_G);
  }
  a = group_sum(a);
  b = group_sum(b);
  c = group_sum(c);
  if (!live || sub != 0) return;
  if (a * b != c) atomicOr(&status[p], 1u);
  abc[e] = a;
  abc[total + e] = b;
  abc[2 * total + e] = c;
}
void launch_spmv_check(hipStream_t st, DevCircuit dc, const Fr* W, Fr* abc, uint32_t n, uint32_t P, uint32_t* status, int16_t* small) {
  // circuits with long rows, up to 64 proofs: with 16 lanes per row the longest row is what a small batch waits for (audit: 0.3 ms +
  // 33 us per proof against 4.4 ms for the run kernel's single lane on the 6 720-term row; measured equal near 128 proofs)
  if ((uint64_t)dc.n_runs * P < 65536 || (P <= 64 && dc.max_row_terms > 1024)) {
    const uint64_t lanes = (uint64_t)n * P * SPMV_G;
    hipLaunchKernelGGL(k_spmv_check_rows, dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, st, dc, W, abc, n, P, status);
    return;
  }
  if (small && dc.sm_nrows) {
    const uint64_t l1 = (uint64_t)dc.sm_nslots * P, l2 = (uint64_t)dc.sm_nrows * P;
    hipLaunchKernelGGL(k_small_extract, dim3((uint32_t)((l1 + 255) / 256)), dim3(256), 0, st, dc, W, small, P, status);
    hipLaunchKernelGGL(k_spmv_small_rows, dim3((uint32_t)((l2 + 255) / 256)), dim3(256), 0, st, dc, small, W, abc, n, P);
  } else {
    dc.row_small = dc.row_long;
  }
  if (dc.lg_n) hipLaunchKernelGGL(k_spmv_long_rows, dim3((P + 63) / 64, dc.lg_n), dim3(1024), 0, st, dc, W, abc, n, P);
  const uint64_t lanes = (uint64_t)(dc.n_runs + (n - dc.n_constraints)) * P;
  hipLaunchKernelGGL(k_spmv_check, dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, st, dc, W, abc, n, P, status);
}

}  // namespace spp
