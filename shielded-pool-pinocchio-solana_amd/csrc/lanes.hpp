// lanes.hpp -- cross-lane helpers of the cooperative (one wave per instance) kernels and the lane-parallel Poseidon2
// permutation they share: the solver's COOP_POSEIDON2 item (kernels_solve.hip) and the ciphertext sponge of small batches
// (kernels_witness.hip, ct_helper/src/main.nr:15-34).
#pragma once
#include "bn254.hpp"

namespace spp {

__device__ __forceinline__ Fr lane_get(const Fr& v, uint32_t src) {
  Fr r;
  SPP_UNROLL for (int i = 0; i < 8; i++) r.l[i] = (uint32_t)__shfl((int)v.l[i], (int)src);
  return r;
}
// the value of one FIXED lane in every lane: v_readlane_b32 (scalar path) instead of the LDS crossbar of ds_bpermute
template <int SRC>
__device__ __forceinline__ Fr lane_bcast(const Fr& v) {
  Fr r;
  SPP_UNROLL for (int i = 0; i < 8; i++) r.l[i] = (uint32_t)__builtin_amdgcn_readlane((int)v.l[i], SRC);
  return r;
}
// lane ^ 1 / lane ^ 2 inside each quad: one DPP move per word (quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E)
template <int CTRL>
__device__ __forceinline__ Fr lane_quad(const Fr& v) {
  Fr r;
  SPP_UNROLL for (int i = 0; i < 8; i++) r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], CTRL, 0xF, 0xF, true);
  return r;
}
__device__ __forceinline__ Fr lane_sel(bool c, const Fr& a, const Fr& b) {
  Fr r;
  SPP_UNROLL for (int i = 0; i < 8; i++) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}

// Poseidon2 (t = 4, RF = 8, RP = 56): lanes 0..3 hold the state, lanes 4..7 compute x^4 next to x^3 (full rounds); in a partial
// round lane 4 carries mu_0 * x alongside the S-box of lane 0, so that mu_0 * x^5 = (mu_0 * x) * x^4 is ready together with
// x^5: three dependent products per round instead of eight.  s: the state word of lanes 0..3 on entry and on return (other
// lanes: don't care).  emit(offset, x2, x3, x4, x5): called with the powers of every S-box input -- by lanes 0..3 with offset
// 16*round + 4*lane in the full rounds, by lane 0 with the running offset in the partial rounds (the solver stores them as
// witness wires; the sponge passes a no-op).
template <class Emit>
__device__ __forceinline__ Fr coop_p2_permute(const Fr* __restrict__ rc, const Fr* __restrict__ mus, Fr s, uint32_t lane, Emit&& emit) {
  const uint32_t l4 = lane & 3;
  const Fr mu = mus[l4];
  auto external = [&](const Fr& mine) {   // rows (5,7,1,3),(4,6,1,1),(1,3,5,7),(1,1,4,6) of the state held by lanes 0..3
    const Fr x = lane_bcast<0>(mine), y = lane_bcast<1>(mine), z = lane_bcast<2>(mine), w = lane_bcast<3>(mine);
    const Fr t0 = x + y, t1 = z + w, t2 = y.dbl() + t1, t3 = w.dbl() + t0;
    const Fr t4 = t1.dbl().dbl() + t3, t5 = t0.dbl().dbl() + t2;
    const Fr t6 = t3 + t5, t7 = t2 + t4;
    return lane_sel(l4 < 2, lane_sel(l4 == 0, t6, t5), lane_sel(l4 == 2, t7, t4));
  };
  s = external(s);
  uint32_t k = 0, out = 0;
  auto full_round = [&]() {
    const Fr x = s + rc[k + l4];
    const Fr x2 = x * x;
    const Fr t = lane_get(x2, l4);                       // lanes 4..7: x^2 of lane - 4
    const Fr R = t * lane_sel(lane < 4, x, t);           // lanes 0..3: x^3, lanes 4..7: x^4
    const Fr x4 = lane_get(R, l4 + 4);
    const Fr x5 = x4 * x;
    if (lane < 4) emit(out + 4 * lane, x2, R, x4, x5);
    out += 16;
    k += 4;
    s = external(x5);
  };
#pragma unroll 1
  for (int r = 0; r < 4; r++) full_round();
#pragma unroll 1
  for (int r = 0; r < 56; r++) {
    const Fr x = s + rc[k];                              // lane 0
    const Fr x0 = lane_bcast<0>(x);
    const Fr R1 = lane_sel(lane == 0, x0, lane_sel(lane < 4, s, x0)) * lane_sel(lane == 0, x0, mu);
    // R1: lane 0 x^2 | lanes 1..3 mu_i * s_i | lane 4 mu_0 * x
    const Fr x2 = lane_bcast<0>(R1);
    const Fr R2 = x2 * lane_sel(lane == 0, x0, x2);      // lane 0 x^3 | lane 4 x^4
    const Fr x4 = lane_bcast<4>(R2);
    const Fr R3 = x4 * lane_sel(lane == 0, x0, R1);      // lane 0 x^5 | lane 4 mu_0 * x^5
    if (lane == 0) emit(out, R1, R2, x4, R3);
    out += 4;
    k += 1;
    const Fr val = lane_sel(lane == 0, R3, s);
    Fr tot = val + lane_quad<0xB1>(val);
    tot = tot + lane_quad<0x4E>(tot);
    const Fr m0 = lane_bcast<4>(R3);
    s = lane_sel(lane == 0, m0, R1) + tot;
  }
#pragma unroll 1
  for (int r = 0; r < 4; r++) full_round();
  return s;
}

}  // namespace spp
