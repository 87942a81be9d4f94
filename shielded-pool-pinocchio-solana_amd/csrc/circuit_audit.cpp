// RLWE audit circuit -- R1CS for the statement of the reference's generated audit_circuit/src/main.nr
// (emitted by scripts/generate_audit.py:246-465; the .nr itself is absent from the reference, see
// .MISSING_LARGE_BLOBS, and is regenerated from demo-frontend/public/rlwe/rlwe_pk.json).
//
//   public : wa_commitment, ct_commitment                                   generate_audit.py:405-407
//   private: c0_packed[10], c1_packed[147], secret_key, r[1024], e1_sparse[64], e2[1024], k0[64], k1[1024]
//
//   1. (owner_x, owner_y) = secret_key * G on Grumpkin                     :418-426
//   2. wa_commitment == Poseidon(owner_x, owner_y)                         :428-430
//   3. c0[64], c1[1024] = 32-bit coefficients unpacked from the packed fields (7 per field)   :432-434
//   4. msg[64] = little-endian bytes of owner_x then owner_y               :436-441
//   5. r, e1, e2 in [-128, 127]  (value + 128 fits u8)                     :443-446
//   6. c0[i] + k0[i]*Q == <PK_B_ROW[i], r> + e1[i] + DELTA*msg[i]           :448-453
//   7. c1[i] + k1[i]*Q == <PK_A_ROW[i], r> + e2[i]                          :455-460
//   8. ct_commitment == Poseidon2 sponge(c0_packed ++ c1_packed)           :462-464
//
// Every "fits in 8/32 bits" check is an 8-bit table lookup proved with one log-derivative argument over
// a BSB22 commitment (the mechanism gnark uses for the reference's R1CS), which keeps the circuit at
// ~25 K constraints -- the reference's README quotes ~26 K -- and the proof at 388 bytes.
// Unpacking is stated as  packed == sum_j coeff_j * 2^(32 j)  with every coeff_j < 2^32; for honest inputs
// this equals the reference's repeated `val as u32` truncation (:318-333), for malformed ones it is stricter.
#include "circuit.hpp"

namespace spp {

static const uint64_t RLWE_Q = 167772161ull;
static const uint64_t RLWE_DELTA = 655360ull;
static const int RLWE_N = 1024, RLWE_SLOTS = 64, PACK_WIDTH = 7;

// 32-bit coefficients of a packed field as linear forms over byte limbs (each limb looked up)
static void unpack(Builder& b, const LC& packed, int ncoeff, std::vector<LC>& out) {
  std::vector<LC> limbs = b.to_limbs8(packed, 4 * ncoeff);
  for (int j = 0; j < ncoeff; j++) {
    LC c = limbs[4 * j] + limbs[4 * j + 1].scaled_u64(1u << 8) + limbs[4 * j + 2].scaled_u64(1u << 16) +
           limbs[4 * j + 3].scaled_u64(1u << 24);
    out.push_back(c);
  }
}

// canonical little-endian bytes of a field element (32 slots; bits 254, 255 are zero)
static std::vector<LC> byte_slots(Builder& b, const LC& v) {
  std::vector<LC> bits = b.to_bits(v, 254);
  uint32_t rm1[8];
  for (int i = 0; i < 8; i++) rm1[i] = FrParams::MOD(i);
  rm1[0] -= 1;
  b.assert_bits_leq_const(bits, rm1);
  std::vector<LC> slots(32);
  for (int i = 0; i < 32; i++)
    for (int j = 0; j < 8; j++) {
      int bi = 8 * i + j;
      if (bi < 254) slots[i] = slots[i] + bits[bi].scaled_u64(1ull << j);
    }
  return slots;
}

Circuit build_audit_circuit(const uint32_t* pk_a, const uint32_t* pk_b, bool native_hints) {
  Builder b(CIRCUIT_AUDIT);
  LC wa_commitment = b.public_input();
  LC ct_commitment = b.public_input();
  std::vector<LC> c0_packed, c1_packed, r, e1, e2, k0, k1;
  for (int i = 0; i < 10; i++) c0_packed.push_back(b.secret_input());
  for (int i = 0; i < 147; i++) c1_packed.push_back(b.secret_input());
  LC secret_key = b.secret_input();
  for (int i = 0; i < RLWE_N; i++) r.push_back(b.secret_input());
  for (int i = 0; i < RLWE_SLOTS; i++) e1.push_back(b.secret_input());
  for (int i = 0; i < RLWE_N; i++) e2.push_back(b.secret_input());
  for (int i = 0; i < RLWE_SLOTS; i++) k0.push_back(b.secret_input());
  for (int i = 0; i < RLWE_N; i++) k1.push_back(b.secret_input());
  const uint32_t r_wire0 = r[0].t[0].first;

  // 1. public key
  std::vector<LC> skbits = b.to_bits(secret_key, 254);
  uint32_t rm1[8];
  for (int i = 0; i < 8; i++) rm1[i] = FrParams::MOD(i);
  rm1[0] -= 1;
  b.assert_bits_leq_const(skbits, rm1);
  auto pk = gadget_grumpkin_fixed_base(b, skbits, native_hints);
  // materialise owner_x / owner_y as wires (they feed two hashes-worth of linear forms and two bit decompositions)
  LC owner_x = b.mul(pk.first, LC::constant(Fr::one()), true, false);
  LC owner_y = b.mul(pk.second, LC::constant(Fr::one()), true, false);

  // 2. wa_commitment
  b.assert_eq(gadget_poseidon_hash(b, {owner_x, owner_y}, native_hints), wa_commitment);

  // 3. unpack ciphertext
  std::vector<LC> c0, c1;
  for (int i = 0; i < 10; i++) unpack(b, c0_packed[i], std::min(PACK_WIDTH, RLWE_SLOTS - PACK_WIDTH * i), c0);
  for (int i = 0; i < 147; i++) unpack(b, c1_packed[i], std::min(PACK_WIDTH, RLWE_N - PACK_WIDTH * i), c1);

  // 4. message slots
  std::vector<LC> msg = byte_slots(b, owner_x);
  std::vector<LC> sy = byte_slots(b, owner_y);
  msg.insert(msg.end(), sy.begin(), sy.end());

  // 5. small-noise range proofs
  LC k128 = LC::constant_u64(128);
  for (auto& v : r) b.lookup8(v + k128);
  for (auto& v : e1) b.lookup8(v + k128);
  for (auto& v : e2) b.lookup8(v + k128);

  // 6./7. quotient equations; <row_k, r> with row_k[j] = poly[k-j] or (q - poly[k-j+n]) (generate_audit.py:57-66)
  Fr fq = Fr::from_u64(RLWE_Q);
  auto inner = [&](const uint32_t* poly, int k) {
    LC ip;
    ip.t.reserve(RLWE_N);
    for (int j = 0; j < RLWE_N; j++) {
      int idx = k - j;
      uint64_t cf = idx >= 0 ? poly[idx] : (poly[idx + RLWE_N] ? RLWE_Q - poly[idx + RLWE_N] : 0);
      if (cf) ip.t.push_back({r_wire0 + (uint32_t)j, Fr::from_u64(cf)});
    }
    return ip;
  };
  for (int i = 0; i < RLWE_SLOTS; i++)
    b.assert_eq(c0[i] + k0[i].scaled(fq), inner(pk_b, i) + e1[i] + msg[i].scaled_u64(RLWE_DELTA));
  for (int i = 0; i < RLWE_N; i++) b.assert_eq(c1[i] + k1[i].scaled(fq), inner(pk_a, i) + e2[i]);

  // 8. ct_commitment: Poseidon2 sponge, rate 3 (ct_helper/src/main.nr:15-34)
  std::vector<LC> packed = c0_packed;
  packed.insert(packed.end(), c1_packed.begin(), c1_packed.end());
  LC st[4];
  const int total = (int)packed.size(), full = total / 3;
  for (int i = 0; i < full; i++) {
    for (int j = 0; j < 3; j++) st[j] = st[j] + packed[3 * i + j];
    gadget_poseidon2_permute(b, st, native_hints);
  }
  int rem = total - 3 * full;
  if (rem >= 1) st[0] = st[0] + packed[3 * full];
  if (rem >= 2) st[1] = st[1] + packed[3 * full + 1];
  gadget_poseidon2_permute(b, st, native_hints);
  b.assert_eq(st[0], ct_commitment);

  return b.finish();
}

}  // namespace spp
