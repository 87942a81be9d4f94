// Micro-benchmark + exactness check: a 254-bit Montgomery multiplication over BN254 Fq built from FP64 FMAs
// on 5 limbs of 52 bits (the scheme of Emmart, Zheng, Weems, "Faster modular exponentiation using double precision floating
// point arithmetic on the GPU", ARITH 2018) next to the product the MSM kernels use (f29.hpp: 9 limbs of 29 bits, 81 + 81
// v_mad_u64_u32 into carry-free 64-bit columns).
//
//   fp64 exact : per 52x52 partial product  hi = fma_rz(a, b, 2^104), lo = fma_rz(a, b, (2^104 + 2^52) - hi); the bit patterns of hi
//                and lo are added into 64-bit integer columns (the exponent fields are a constant per column, folded into the
//                initial value): 2 FMA + 1 FP add + 2 integer 64-bit adds per partial product, 50 partial products, plus the
//                per-limb work of the reduction (low product by -p^-1, int <-> double conversions, carries).  Round toward zero
//                is set once per kernel in the MODE register.  Checked against integers on the host (this file, `check`).
//   fp64 floor : the same 50 partial products with ONLY the two FMAs each, chained through per-column double accumulators, and
//                integer work per COLUMN only -- what the arithmetic would cost if the sums could stay in floating point.  They
//                cannot: a sum of two lo parts needs 54 bits, so this variant computes garbage.  It is timed as a lower bound
//                for any FP64 scheme, nothing else.
//   f29        : F29<FqParams>::operator* (the product in the MSM inner loop).
//
// Each lane runs a chain x <- x * y (the multiplications of a point addition are mostly dependent in the same way); time per
// multiplication = kernel time / (chain length x waves per SIMD).  Occupancy is set through dynamic LDS.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I../../shielded-pool-pinocchio-solana_amd/csrc fp64_mul.hip -o fp64_mul
#include <hip/hip_runtime.h>
#include <cfenv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "f29.hpp"

using namespace spp;

// Rounding: the scheme needs round-toward-zero FMAs.  hipcc tracks the MODE register (SIModeRegister) and puts the rounding
// field back to nearest-even in front of every floating-point instruction it emits itself, so on the device every FP64
// operation of the multiplication is an `asm volatile` statement (invisible to that pass) and the kernel sets
// MODE.fp_round(f64) = toward zero once.  Volatile statements keep their order, so the products are written a row at a time
// (five independent hi, five sub, five lo) to keep dependent instructions apart.  The host check runs the same source
// with std::fma under fesetround(FE_TOWARDZERO).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double FMA_RZ(double a, double b, double c) {
  double r;
  asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ double SUB_RZ(double a, double b) {
  double r;
  asm volatile("v_add_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
#else
__host__ __device__ static inline double FMA_RZ(double a, double b, double c) { return std::fma(a, b, c); }
__host__ __device__ static inline double SUB_RZ(double a, double b) { return a - b; }
#endif

struct D52 {
  static constexpr uint64_t M52 = (1ull << 52) - 1;   // BN254 Fq in 5 x 52-bit limbs; Montgomery radix 2^260
};

// limbs of p and -p^-1 mod 2^52, filled by the host at start-up (exact integer arithmetic) and passed to the kernels
struct D52Consts {
  double p[5];
  double np;   // -p^-1 mod 2^52
};

__host__ __device__ inline uint64_t dbits(double x) {
  uint64_t u;
  memcpy(&u, &x, 8);
  return u;
}
__host__ __device__ inline double bitsd(uint64_t u) {
  double x;
  memcpy(&x, &u, 8);
  return x;
}

// exponent patterns carried by the hi / lo terms
static constexpr uint64_t EXP_HI = 0x467ull << 52;   // 2^104
static constexpr uint64_t EXP_LO = 0x433ull << 52;   // 2^52

// r = a * b / 2^260 mod p, r < 1.02 p for a, b < 1.02 p.  Exact.
__host__ __device__ inline void dmul52(const D52Consts& k, const double (&a)[5], const double (&b)[5], double (&r)[5]) {
  const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52, C3 = 0x1p52;
  // column k receives: lo terms of products with i + j = k, hi terms of products with i + j = k - 1 -- from a*b and from q*p
  // alike (q_i * p_j lands in column i + j): counts nlo[k] = 2 * #{(i,j): i+j=k}, nhi[k] = 2 * #{(i,j): i+j=k-1}
  uint64_t col[11];
#pragma unroll
  for (int c = 0; c < 11; c++) {
    const int nlo = c <= 8 ? (c < 5 ? c + 1 : 9 - c) : 0;
    const int nhi = (c >= 1 && c <= 9) ? ((c - 1) < 5 ? c : 10 - c) : 0;
    col[c] = 0ull - 2ull * ((uint64_t)nlo * EXP_LO + (uint64_t)nhi * EXP_HI);
  }
#pragma unroll
  for (int i = 0; i < 5; i++) {
    double hi[5], lo[5];
#pragma unroll
    for (int j = 0; j < 5; j++) hi[j] = FMA_RZ(a[i], b[j], C1);
#pragma unroll
    for (int j = 0; j < 5; j++) lo[j] = SUB_RZ(C2, hi[j]);
#pragma unroll
    for (int j = 0; j < 5; j++) lo[j] = FMA_RZ(a[i], b[j], lo[j]);
#pragma unroll
    for (int j = 0; j < 5; j++) {
      col[i + j] += dbits(lo[j]);
      col[i + j + 1] += dbits(hi[j]);
    }
  }
  // the q * p terms have not been added yet: their exponent patterns are still "owed" by the columns.  Add them as they come.
#pragma unroll
  for (int i = 0; i < 5; i++) {
    // the low 52 bits of column i are final once q_0 .. q_{i-1} have been added: every exponent pattern is a multiple of 2^52
    const uint64_t t = col[i] & D52::M52;
    const double td = SUB_RZ(bitsd(t | EXP_LO), C3);                    // t as a double (exact)
    const double qh = FMA_RZ(td, k.np, C1);
    const double ql = FMA_RZ(td, k.np, SUB_RZ(C2, qh));                 // 2^52 + (t * np mod 2^52)
    const double q = SUB_RZ(ql, C3);
    double hi[5], lo[5];
#pragma unroll
    for (int j = 0; j < 5; j++) hi[j] = FMA_RZ(q, k.p[j], C1);
#pragma unroll
    for (int j = 0; j < 5; j++) lo[j] = SUB_RZ(C2, hi[j]);
#pragma unroll
    for (int j = 0; j < 5; j++) lo[j] = FMA_RZ(q, k.p[j], lo[j]);
#pragma unroll
    for (int j = 0; j < 5; j++) {
      col[i + j] += dbits(lo[j]);
      col[i + j + 1] += dbits(hi[j]);
    }
    // column i is complete and its low 52 bits are zero: pass the carry on
    // (columns above still owe patterns, the carry is a plain integer, so nothing interferes)
    col[i + 1] += col[i] >> 52;
  }
#pragma unroll
  for (int c = 5; c < 10; c++) {
    r[c - 5] = SUB_RZ(bitsd((col[c] & D52::M52) | EXP_LO), C3);
    col[c + 1] += col[c] >> 52;
  }
}

// lower bound: two FMAs per partial product, nothing else per product (NOT a multiplication: see the header)
__device__ inline void dmul52_floor(const D52Consts& k, const double (&a)[5], const double (&b)[5], double (&r)[5]) {
  double ch[11], cl[11];
#pragma unroll
  for (int c = 0; c < 11; c++) {
    ch[c] = 0x1p104;
    cl[c] = 0x1p52;
  }
#pragma unroll
  for (int i = 0; i < 5; i++) {
#pragma unroll
    for (int j = 0; j < 5; j++) {
      ch[i + j + 1] = FMA_RZ(a[i], b[j], ch[i + j + 1]);
      cl[i + j] = FMA_RZ(a[i], b[j], cl[i + j]);
    }
  }
  uint64_t carry = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const uint64_t t = (dbits(ch[i]) + dbits(cl[i]) + carry) & D52::M52;   // per-column integer work
    const double td = SUB_RZ(bitsd(t | EXP_LO), 0x1p52);
    const double q = SUB_RZ(FMA_RZ(td, k.np, 0x1p52), 0x1p52);
#pragma unroll
    for (int j = 0; j < 5; j++) {
      ch[i + j + 1] = FMA_RZ(q, k.p[j], ch[i + j + 1]);
      cl[i + j] = FMA_RZ(q, k.p[j], cl[i + j]);
    }
    carry = (dbits(ch[i]) + dbits(cl[i]) + carry) >> 52;
  }
#pragma unroll
  for (int c = 5; c < 10; c++) {
    const uint64_t v = dbits(ch[c]) + dbits(cl[c]) + carry;
    r[c - 5] = SUB_RZ(bitsd((v & D52::M52) | EXP_LO), 0x1p52);
    carry = v >> 52;
  }
}

__device__ inline void set_f64_round_toward_zero() {
  // s_setreg_b32 hwreg(HW_REG_MODE, 2, 2), 3 : FP_ROUND field for f64/f16 = round toward zero
  __builtin_amdgcn_s_setreg(1 | (2 << 6) | (1 << 11), 3);
}

constexpr int CHAIN = 4096;

__global__ void __launch_bounds__(256) k_fp64_exact(D52Consts k, const double* in, double* out, int chain) {
  extern __shared__ char lds[];
  set_f64_round_toward_zero();
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double x[5], y[5], r[5];
  for (int i = 0; i < 5; i++) {
    x[i] = in[(g % 4096) * 10 + i];
    y[i] = in[(g % 4096) * 10 + 5 + i];
  }
#pragma unroll 1
  for (int it = 0; it < chain; it++) {
    dmul52(k, x, y, r);
    for (int i = 0; i < 5; i++) x[i] = r[i];
  }
  for (int i = 0; i < 5; i++) out[g * 5 + i] = x[i];
  if (chain < 0) lds[threadIdx.x] = 0;
}
__global__ void __launch_bounds__(256) k_fp64_floor(D52Consts k, const double* in, double* out, int chain) {
  extern __shared__ char lds[];
  set_f64_round_toward_zero();
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double x[5], y[5], r[5];
  for (int i = 0; i < 5; i++) {
    x[i] = in[(g % 4096) * 10 + i];
    y[i] = in[(g % 4096) * 10 + 5 + i];
  }
#pragma unroll 1
  for (int it = 0; it < chain; it++) {
    dmul52_floor(k, x, y, r);
    for (int i = 0; i < 5; i++) x[i] = r[i];
  }
  for (int i = 0; i < 5; i++) out[g * 5 + i] = x[i];
  if (chain < 0) lds[threadIdx.x] = 0;
}
__global__ void __launch_bounds__(256) k_f29(D52Consts, const double* in, double* out, int chain) {
  extern __shared__ char lds[];
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  F29<FqParams> x, y;
  for (int i = 0; i < 9; i++) {
    x.l[i] = (uint32_t)(dbits(in[(g % 4096) * 10 + (i % 5)]) >> (3 * i)) & F29<FqParams>::M;
    y.l[i] = (uint32_t)(dbits(in[(g % 4096) * 10 + 5 + (i % 5)]) >> (2 * i)) & F29<FqParams>::M;
  }
  x.l[8] &= 0xfffff;
  y.l[8] &= 0xfffff;
#pragma unroll 1
  for (int it = 0; it < chain; it++) x = x * y;
  for (int i = 0; i < 5; i++) out[g * 5 + i] = (double)x.l[i];
  if (chain < 0) lds[threadIdx.x] = 0;
}

// ---- host: exact integers (4 x 64 with unsigned __int128) for the check ----
typedef unsigned __int128 u128;
struct U320 { uint64_t w[5]; };   // little endian 64-bit words, 320 bits
static const uint64_t P64[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static void to52(const uint64_t w[4], double out[5]) {
  // 256-bit integer -> 5 limbs of 52 bits
  for (int i = 0; i < 5; i++) {
    const int bit = 52 * i, q = bit / 64, o = bit % 64;
    uint64_t v = w[q] >> o;
    if (o > 12 && q + 1 < 4) v |= w[q + 1] << (64 - o);
    out[i] = (double)(v & D52::M52);
  }
}
static void from52(const double in[5], uint64_t w[5]) {
  for (int i = 0; i < 5; i++) w[i] = 0;
  for (int i = 0; i < 5; i++) {
    const uint64_t v = (uint64_t)in[i];
    const int bit = 52 * i, q = bit / 64, o = bit % 64;
    w[q] |= v << o;
    if (o > 12 && q + 1 < 5) w[q + 1] |= v >> (64 - o);
  }
}
// (a * b) mod p by shift-and-add on 320-bit values (slow, host only)
static bool geq(const uint64_t a[5], const uint64_t b[5]) {
  for (int i = 4; i >= 0; i--) {
    if (a[i] != b[i]) return a[i] > b[i];
  }
  return true;
}
static void sub_(uint64_t a[5], const uint64_t b[5]) {
  u128 br = 0;
  for (int i = 0; i < 5; i++) {
    u128 d = (u128)a[i] - b[i] - br;
    a[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}
static void addmod(uint64_t a[5], const uint64_t b[5], const uint64_t p[5]) {
  u128 c = 0;
  for (int i = 0; i < 5; i++) {
    c += (u128)a[i] + b[i];
    a[i] = (uint64_t)c;
    c >>= 64;
  }
  while (geq(a, p)) sub_(a, p);
}
static void mulmod(const uint64_t a[5], const uint64_t b[5], const uint64_t p[5], uint64_t out[5]) {
  uint64_t acc[5] = {0, 0, 0, 0, 0}, cur[5];
  memcpy(cur, a, 40);
  while (geq(cur, p)) sub_(cur, p);
  for (int bit = 0; bit < 320; bit++) {
    if ((b[bit / 64] >> (bit % 64)) & 1) addmod(acc, cur, p);
    uint64_t t[5];
    memcpy(t, cur, 40);
    addmod(cur, t, p);
  }
  memcpy(out, acc, 40);
}

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd64() {
  rng_state ^= rng_state << 7;
  rng_state ^= rng_state >> 9;
  return rng_state * 0x2545f4914f6cdd1dull;
}

int main(int argc, char** argv) {
  const bool check_only = argc > 1 && !strcmp(argv[1], "check");
  // constants
  D52Consts k;
  to52(P64, k.p);
  {
    // -p^-1 mod 2^52 by Newton iteration on the low limb
    const uint64_t p0 = P64[0] & D52::M52;
    uint64_t inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - p0 * inv;
    k.np = (double)((0 - inv) & D52::M52);
  }
  uint64_t p5[5] = {P64[0], P64[1], P64[2], P64[3], 0};
  // ---- exactness check on the host, same source, round toward zero ----
  std::fesetround(FE_TOWARDZERO);
  // 2^260 mod p (to undo the Montgomery factor)
  uint64_t r260[5] = {0, 0, 0, 0, 16}, one[5] = {1, 0, 0, 0, 0}, R[5];
  mulmod(r260, one, p5, R);
  int bad = 0;
  const int NCHK = 2000;
  for (int t = 0; t < NCHK; t++) {
    uint64_t a[5] = {rnd64(), rnd64(), rnd64(), rnd64() >> 2, 0}, b[5] = {rnd64(), rnd64(), rnd64(), rnd64() >> 2, 0};
    if (t == 0) { memset(a, 0, 40); }
    if (t == 1) { memcpy(a, p5, 40); a[0] -= 1; memcpy(b, a, 40); }                          // p - 1 squared
    if (t == 2) { for (int i = 0; i < 4; i++) a[i] = b[i] = ~0ull; a[3] = b[3] = P64[3]; }   // every limb of the low words saturated
    while (geq(a, p5)) sub_(a, p5);
    while (geq(b, p5)) sub_(b, p5);
    double ad[5], bd[5], rd[5];
    to52(a, ad);
    to52(b, bd);
    dmul52(k, ad, bd, rd);
    uint64_t r[5], lhs[5], rhs[5];
    from52(rd, r);
    mulmod(r, R, p5, lhs);      // r * 2^260
    mulmod(a, b, p5, rhs);
    bool ok = memcmp(lhs, rhs, 40) == 0;
    for (int i = 0; i < 5; i++) ok = ok && rd[i] >= 0 && rd[i] < 0x1p52;
    uint64_t twop[5] = {0, 0, 0, 0, 0};
    addmod(twop, p5, p5);       // p mod p = 0 ... so build 2p by hand: r must stay below it
    u128 cy = 0;
    for (int i = 0; i < 5; i++) { cy += (u128)p5[i] * 2; twop[i] = (uint64_t)cy; cy >>= 64; }
    ok = ok && !geq(r, twop);
    if (!ok) bad++;
  }
  std::fesetround(FE_TONEAREST);
  printf("host check (round toward zero, %d products incl. 0, (p-1)^2, saturated limbs): %s\n", NCHK, bad ? "FAILED" : "exact");
  if (bad) return 1;
  if (check_only) return 0;

  // ---- device: same check on 4096 lanes, then timing ----
  std::vector<double> in(4096 * 10);
  std::vector<uint64_t> ia(4096 * 5), ib(4096 * 5);
  for (int g = 0; g < 4096; g++) {
    uint64_t a[5] = {rnd64(), rnd64(), rnd64(), rnd64() >> 2, 0}, b[5] = {rnd64(), rnd64(), rnd64(), rnd64() >> 2, 0};
    while (geq(a, p5)) sub_(a, p5);
    while (geq(b, p5)) sub_(b, p5);
    to52(a, &in[g * 10]);
    to52(b, &in[g * 10 + 5]);
    memcpy(&ia[g * 5], a, 40);
    memcpy(&ib[g * 5], b, 40);
  }
  double *d_in, *d_out;
  const size_t max_lanes = (size_t)256 * 8 * 256;
  hipMalloc((void**)&d_in, in.size() * 8);
  hipMalloc((void**)&d_out, max_lanes * 5 * 8);
  hipMemcpy(d_in, in.data(), in.size() * 8, hipMemcpyHostToDevice);
  {
    hipLaunchKernelGGL(k_fp64_exact, dim3(16), dim3(256), 0, 0, k, d_in, d_out, 1);
    std::vector<double> out(4096 * 5);
    hipMemcpy(out.data(), d_out, out.size() * 8, hipMemcpyDeviceToHost);
    int dbad = 0;
    for (int g = 0; g < 4096; g++) {
      uint64_t r[5], lhs[5], rhs[5];
      from52(&out[g * 5], r);
      mulmod(r, R, p5, lhs);
      mulmod(&ia[g * 5], &ib[g * 5], p5, rhs);
      if (memcmp(lhs, rhs, 40)) dbad++;
    }
    printf("device check (gfx950, MODE.fp_round = toward zero, 4096 products): %s\n", dbad ? "FAILED" : "exact");
    if (dbad) return 1;
  }
  typedef void (*kern_t)(D52Consts, const double*, double*, int);
  struct { const char* name; kern_t kern; } ks[] = {{"f29 (9x29, v_mad_u64_u32)", k_f29}, {"fp64 exact (5x52, Emmart)", k_fp64_exact},
                                                   {"fp64 floor (2 FMA/product)", k_fp64_floor}};
  double base[9] = {0};
  int bi = 0;
  for (int occ : {1, 2, 4}) {
    const size_t lds = occ == 1 ? 96 * 1024 : occ == 2 ? 64 * 1024 : 36 * 1024;
    for (auto& kk : ks) {
      hipFuncSetAttribute((const void*)kk.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      const int blocks = 256 * occ;
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipLaunchKernelGGL(kk.kern, dim3(blocks), dim3(256), lds, 0, k, d_in, d_out, CHAIN);
      hipDeviceSynchronize();
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kk.kern, dim3(blocks), dim3(256), lds, 0, k, d_in, d_out, CHAIN);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      // one wave per SIMD per block: a SIMD runs occ chains of CHAIN multiplications
      const double ns_per_mul = best * 1e6 / ((double)CHAIN * occ);
      const double cyc = ns_per_mul * 2.4;
      base[bi] = ns_per_mul;
      printf("waves/SIMD=%d  %-28s %8.3f ms  %7.1f ns per wave-multiplication per SIMD (%6.0f cycles at 2.4 GHz nominal)  x%.2f of f29\n", occ,
             kk.name, best, ns_per_mul, cyc, ns_per_mul / base[bi - (bi % 3)]);
      bi++;
    }
  }
  return 0;
}
