// BN254 prime-field arithmetic for CDNA4 (gfx950): Fp (base field) and Fr (scalar field).
//
// Element layout = the reference's in-memory layout (constantine, 64-bit build; SURVEY 8a,
// reference groth16/bn128/io.nim:60-92): 256-bit little-endian integer in Montgomery form with
// R = 2^256, canonical (< modulus).  On the device it is handled as 8 x u32 limbs, which is the
// same 32 bytes.
//
// Multiplication is product-scanning (Comba/FIPS) Montgomery: every 32x32 partial product is ONE
// v_mad_u64_u32 into a 64-bit column accumulator plus ONE v_addc_co_u32 collecting the carry-out
// into a third word.  Measured issue cost on MI355X (profiles/r01_ubench_int_issue_rates.txt):
// v_mad_u64_u32 ~4.9 cycles/wave, v_addc ~2.7 -- so 32-bit limbs with the wide mad beat 24-bit
// mads and fp64-FMA limb tricks on this chip.  No MFMA: carry-propagated big-int.
//
// The same header compiles with g++ (tests/cpu_kernels) so that the exact device formulas can be
// unit-tested against the oracle without a GPU; the product library only ever runs them on the GPU
// (and, for the O(1) prover mask algebra that the reference keeps on the host, in host code).
#pragma once
#include <stdint.h>
#include <utility>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FF_HD __host__ __device__ __forceinline__
// Fp2 products: inlined in release builds (measured on MI355X: G2 accumulate 4.09 ms inlined vs 4.85 ms as
// by-value calls vs 5.89 ms as by-reference calls).  -DG16_FP2_CALLS makes them real device functions, which
// cuts the G2 translation unit's compile time from minutes to ~15 s for development builds (`make DEV=1`).
#if defined(G16_FP2_CALLS)
#define FF_HD_CALL __host__ __device__ __attribute__((noinline))
#else
#define FF_HD_CALL __host__ __device__ __forceinline__
#endif
#else
#define FF_HD inline __attribute__((always_inline))
#define FF_HD_CALL inline
#endif
#if defined(__HIPCC__)
#define FF_HD_COLD __host__ __device__ __attribute__((noinline))   // rarely taken paths: keep them out of line
#else
#define FF_HD_COLD __attribute__((noinline))
#endif

namespace g16 {

struct alignas(16) u256 {
  uint32_t v[8];
};

// ---- 96-bit column accumulator primitives -------------------------------------------------------
// (lo:64, hi:32) += a*b
FF_HD void mac(uint64_t& lo, uint32_t& hi, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(lo), "+v"(hi)
      : "v"(a), "v"(b)
      : "vcc");
#else
  unsigned __int128 t = (unsigned __int128)lo + (uint64_t)a * b;
  lo = (uint64_t)t;
  hi += (uint32_t)(t >> 64);
#endif
}
// column shift: (lo,hi) >>= 32
FF_HD void acc_shift(uint64_t& lo, uint32_t& hi) {
  lo = (lo >> 32) | ((uint64_t)hi << 32);
  hi = 0;
}

#include "ff_mac.inc"

// carry-chain steps: lower to v_add_co/v_addc_co and v_sub_co/v_subb_co chains under hipcc
FF_HD uint32_t addc(uint32_t a, uint32_t b, uint32_t& c) {
#if defined(__clang__)
  unsigned co;
  uint32_t r = __builtin_addc(a, b, c, &co);
  c = co;
  return r;
#else
  uint64_t t = (uint64_t)a + b + c;
  c = (uint32_t)(t >> 32);
  return (uint32_t)t;
#endif
}
FF_HD uint32_t subc(uint32_t a, uint32_t b, uint32_t& bw) {
#if defined(__clang__)
  unsigned bo;
  uint32_t r = __builtin_subc(a, b, bw, &bo);
  bw = bo;
  return r;
#else
  uint64_t t = (uint64_t)a - b - bw;
  bw = (uint32_t)(t >> 63);
  return (uint32_t)t;
#endif
}

// ---- field parameters ---------------------------------------------------------------------------
// p = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47   (fields.nim:36)
struct FpParams {
  static constexpr uint32_t P0 = 0xd87cfd47u, P1 = 0x3c208c16u, P2 = 0x6871ca8du, P3 = 0x97816a91u,
                            P4 = 0x8181585du, P5 = 0xb85045b6u, P6 = 0xe131a029u, P7 = 0x30644e72u;
  static constexpr uint32_t INV = 0xe4866389u;  // -p^-1 mod 2^32
  // R mod p  (= fpMontR, io.nim:87)
  static constexpr uint32_t ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                      0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  // R^2 mod p
  static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                     0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
};
// r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001   (fields.nim:37)
struct FrParams {
  static constexpr uint32_t P0 = 0xf0000001u, P1 = 0x43e1f593u, P2 = 0x79b97091u, P3 = 0x2833e848u,
                            P4 = 0x8181585du, P5 = 0xb85045b6u, P6 = 0xe131a029u, P7 = 0x30644e72u;
  static constexpr uint32_t INV = 0xefffffffu;  // -r^-1 mod 2^32
  // R mod r  (= frMontR, io.nim:91)
  static constexpr uint32_t ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                      0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  // R^2 mod r
  static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                     0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
};

template <class PR>
struct Field {
  using T = u256;
  using Params = PR;

  template <int I>
  static FF_HD constexpr uint32_t P() {
    if constexpr (I == 0) return PR::P0;
    else if constexpr (I == 1) return PR::P1;
    else if constexpr (I == 2) return PR::P2;
    else if constexpr (I == 3) return PR::P3;
    else if constexpr (I == 4) return PR::P4;
    else if constexpr (I == 5) return PR::P5;
    else if constexpr (I == 6) return PR::P6;
    else return PR::P7;
  }
  static FF_HD uint32_t Pi(int i) {
    switch (i) {
      case 0: return PR::P0; case 1: return PR::P1; case 2: return PR::P2; case 3: return PR::P3;
      case 4: return PR::P4; case 5: return PR::P5; case 6: return PR::P6; default: return PR::P7;
    }
  }

  static FF_HD T zero() {
    T r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = 0;
    return r;
  }
  static FF_HD T one() {
    T r;
    r.v[0] = PR::ONE[0]; r.v[1] = PR::ONE[1]; r.v[2] = PR::ONE[2]; r.v[3] = PR::ONE[3];
    r.v[4] = PR::ONE[4]; r.v[5] = PR::ONE[5]; r.v[6] = PR::ONE[6]; r.v[7] = PR::ONE[7];
    return r;
  }
  static FF_HD T r2() {
    T r;
    r.v[0] = PR::R2[0]; r.v[1] = PR::R2[1]; r.v[2] = PR::R2[2]; r.v[3] = PR::R2[3];
    r.v[4] = PR::R2[4]; r.v[5] = PR::R2[5]; r.v[6] = PR::R2[6]; r.v[7] = PR::R2[7];
    return r;
  }
  static FF_HD bool is_zero(const T& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.v[i];
    return o == 0;
  }
  static FF_HD bool eq(const T& a, const T& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.v[i] ^ b.v[i];
    return o == 0;
  }

  // r = a - p, returns borrow (1 if a < p)
  static FF_HD uint32_t sub_p(T& r, const T& a) {
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = subc(a.v[i], Pi(i), bw);
    return bw;
  }
  // canonical reduce of a value known to be < 2p
  static FF_HD T reduce_once(const T& a) {
    T t;
    uint32_t borrow = sub_p(t, a);
    T r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = borrow ? a.v[i] : t.v[i];
    return r;
  }
  static FF_HD T add(const T& a, const T& b) {
    T s;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s.v[i] = addc(a.v[i], b.v[i], c);
    return reduce_once(s);  // a+b < 2p < 2^255: no carry out of 256 bits
  }
  static FF_HD T sub(const T& a, const T& b) {
    T d, t, r;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d.v[i] = subc(a.v[i], b.v[i], bw);
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t.v[i] = addc(d.v[i], Pi(i), c);
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = bw ? t.v[i] : d.v[i];  // borrow -> add p back
    return r;
  }
  static FF_HD T neg(const T& a) {
    T r;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = subc(Pi(i), a.v[i], bw);
    bool z = is_zero(a);
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = z ? 0u : r.v[i];
    return r;
  }
  static FF_HD T dbl(const T& a) { return add(a, a); }
  // a/2 mod p  (constantine div2, used by the reference's inverse NTT: ntt.nim:111-112,121)
  static FF_HD T div2(const T& a) {
    uint32_t mask = (a.v[0] & 1) ? 0xffffffffu : 0u;
    uint32_t s[9];
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = addc(a.v[i], Pi(i) & mask, c);
    s[8] = c;
    T r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (s[i] >> 1) | (s[i + 1] << 31);
    return r;
  }

  // ---- product-scanning Montgomery ---------------------------------------------------------------
  // column K of a*b: sum_{i=S}^{S+n-1} a_i * b_{K-i}
  template <int K, int S, int... I>
  static FF_HD void col_ab(uint64_t& lo, uint32_t& hi, const T& a, const T& b, std::integer_sequence<int, I...>) {
    macv(lo, hi, a.v[S + I]..., b.v[K - S - I]...);
  }
  // column K of m*p: sum_{i=S}^{S+n-1} m_i * p_{K-i}
  template <int K, int S, int... I>
  static FF_HD void col_mp(uint64_t& lo, uint32_t& hi, const uint32_t (&m)[8], std::integer_sequence<int, I...>) {
    macs(lo, hi, m[S + I]..., P<K - S - I>()...);
  }
  template <int K>
  static FF_HD void mul_col_lo(uint64_t& lo, uint32_t& hi, const T& a, const T& b, uint32_t (&m)[8]) {
    col_ab<K, 0>(lo, hi, a, b, std::make_integer_sequence<int, K + 1>{});
    if constexpr (K > 0) col_mp<K, 0>(lo, hi, m, std::make_integer_sequence<int, K>{});
    m[K] = (uint32_t)lo * PR::INV;
    macs(lo, hi, m[K], P<0>());
    acc_shift(lo, hi);
  }
  template <int K>
  static FF_HD void mul_col_hi(uint64_t& lo, uint32_t& hi, const T& a, const T& b, const uint32_t (&m)[8], T& r) {
    if constexpr (K < 15) {
      col_ab<K, K - 7>(lo, hi, a, b, std::make_integer_sequence<int, 15 - K>{});
      col_mp<K, K - 7>(lo, hi, m, std::make_integer_sequence<int, 15 - K>{});
    }
    r.v[K - 8] = (uint32_t)lo;
    acc_shift(lo, hi);
  }
  // Montgomery product a*b/R mod p.  128 v_mad_u64_u32 + 128 v_addc + 8 v_mul_lo.
  static FF_HD T mul(const T& a, const T& b) {
    uint32_t m[8];
    T r;
    uint64_t lo = 0;
    uint32_t hi = 0;
    mul_col_lo<0>(lo, hi, a, b, m); mul_col_lo<1>(lo, hi, a, b, m);
    mul_col_lo<2>(lo, hi, a, b, m); mul_col_lo<3>(lo, hi, a, b, m);
    mul_col_lo<4>(lo, hi, a, b, m); mul_col_lo<5>(lo, hi, a, b, m);
    mul_col_lo<6>(lo, hi, a, b, m); mul_col_lo<7>(lo, hi, a, b, m);
    mul_col_hi<8>(lo, hi, a, b, m, r); mul_col_hi<9>(lo, hi, a, b, m, r);
    mul_col_hi<10>(lo, hi, a, b, m, r); mul_col_hi<11>(lo, hi, a, b, m, r);
    mul_col_hi<12>(lo, hi, a, b, m, r); mul_col_hi<13>(lo, hi, a, b, m, r);
    mul_col_hi<14>(lo, hi, a, b, m, r); mul_col_hi<15>(lo, hi, a, b, m, r);
    return reduce_once(r);  // inputs < p  =>  result < 2p
  }
  static FF_HD T sqr(const T& a) { return mul(a, a); }

  // Montgomery reduction alone: a/R mod p  (Montgomery form -> standard form; msm.nim:42-44 `toBig`)
  template <int K>
  static FF_HD void red_col_lo(uint64_t& lo, uint32_t& hi, const T& a, uint32_t (&m)[8]) {
    if constexpr (K > 0) col_mp<K, 0>(lo, hi, m, std::make_integer_sequence<int, K>{});
    uint64_t old = lo;
    lo += a.v[K];
    hi += (lo < old) ? 1u : 0u;
    m[K] = (uint32_t)lo * PR::INV;
    macs(lo, hi, m[K], P<0>());
    acc_shift(lo, hi);
  }
  template <int K>
  static FF_HD void red_col_hi(uint64_t& lo, uint32_t& hi, const uint32_t (&m)[8], T& r) {
    if constexpr (K < 15) col_mp<K, K - 7>(lo, hi, m, std::make_integer_sequence<int, 15 - K>{});
    r.v[K - 8] = (uint32_t)lo;
    acc_shift(lo, hi);
  }
  static FF_HD T from_mont(const T& a) {
    uint32_t m[8];
    T r;
    uint64_t lo = 0;
    uint32_t hi = 0;
    red_col_lo<0>(lo, hi, a, m); red_col_lo<1>(lo, hi, a, m); red_col_lo<2>(lo, hi, a, m);
    red_col_lo<3>(lo, hi, a, m); red_col_lo<4>(lo, hi, a, m); red_col_lo<5>(lo, hi, a, m);
    red_col_lo<6>(lo, hi, a, m); red_col_lo<7>(lo, hi, a, m);
    red_col_hi<8>(lo, hi, m, r); red_col_hi<9>(lo, hi, m, r); red_col_hi<10>(lo, hi, m, r);
    red_col_hi<11>(lo, hi, m, r); red_col_hi<12>(lo, hi, m, r); red_col_hi<13>(lo, hi, m, r);
    red_col_hi<14>(lo, hi, m, r); red_col_hi<15>(lo, hi, m, r);
    return reduce_once(r);
  }
  static FF_HD T to_mont(const T& a) { return mul(a, r2()); }

  // ---- Montgomery dot product  (a*b + c*d) / R mod p  in ONE interleaved product-scanning pass --------------
  // Used by Fp2::mul: both components of an Fp2 product are sums of two base-field products, so they need one
  // reduction each instead of three for Karatsuba -- and no additions or subtractions around the products.
  // Column sums hold up to 24 partial products (< 2^69): the 96-bit accumulator suffices.  Operands <= p give a
  // result < p (2p/R + 1) < 2p, made canonical by one conditional subtraction.
  template <int K>
  static FF_HD void mul2_col_lo(uint64_t& lo, uint32_t& hi, const T& a, const T& b, const T& c, const T& d,
                                uint32_t (&m)[8]) {
    col_ab<K, 0>(lo, hi, a, b, std::make_integer_sequence<int, K + 1>{});
    col_ab<K, 0>(lo, hi, c, d, std::make_integer_sequence<int, K + 1>{});
    if constexpr (K > 0) col_mp<K, 0>(lo, hi, m, std::make_integer_sequence<int, K>{});
    m[K] = (uint32_t)lo * PR::INV;
    macs(lo, hi, m[K], P<0>());
    acc_shift(lo, hi);
  }
  template <int K>
  static FF_HD void mul2_col_hi(uint64_t& lo, uint32_t& hi, const T& a, const T& b, const T& c, const T& d,
                                const uint32_t (&m)[8], T& r) {
    if constexpr (K < 15) {
      col_ab<K, K - 7>(lo, hi, a, b, std::make_integer_sequence<int, 15 - K>{});
      col_ab<K, K - 7>(lo, hi, c, d, std::make_integer_sequence<int, 15 - K>{});
      col_mp<K, K - 7>(lo, hi, m, std::make_integer_sequence<int, 15 - K>{});
    }
    r.v[K - 8] = (uint32_t)lo;
    acc_shift(lo, hi);
  }
  static FF_HD T mul2(const T& a, const T& b, const T& c, const T& d) {
    uint32_t m[8];
    T r;
    uint64_t lo = 0;
    uint32_t hi = 0;
    mul2_col_lo<0>(lo, hi, a, b, c, d, m); mul2_col_lo<1>(lo, hi, a, b, c, d, m);
    mul2_col_lo<2>(lo, hi, a, b, c, d, m); mul2_col_lo<3>(lo, hi, a, b, c, d, m);
    mul2_col_lo<4>(lo, hi, a, b, c, d, m); mul2_col_lo<5>(lo, hi, a, b, c, d, m);
    mul2_col_lo<6>(lo, hi, a, b, c, d, m); mul2_col_lo<7>(lo, hi, a, b, c, d, m);
    mul2_col_hi<8>(lo, hi, a, b, c, d, m, r); mul2_col_hi<9>(lo, hi, a, b, c, d, m, r);
    mul2_col_hi<10>(lo, hi, a, b, c, d, m, r); mul2_col_hi<11>(lo, hi, a, b, c, d, m, r);
    mul2_col_hi<12>(lo, hi, a, b, c, d, m, r); mul2_col_hi<13>(lo, hi, a, b, c, d, m, r);
    mul2_col_hi<14>(lo, hi, a, b, c, d, m, r); mul2_col_hi<15>(lo, hi, a, b, c, d, m, r);
    return reduce_once(r);
  }
  // four-term dot product (a*b + c*d + e*f + g*h) / R mod p  (column sums < 2^70; result < p (4p/R + 1) < 2p)
  template <int K>
  static FF_HD void mul4_col_lo(uint64_t& lo, uint32_t& hi, const T* x, uint32_t (&m)[8]) {
    col_ab<K, 0>(lo, hi, x[0], x[1], std::make_integer_sequence<int, K + 1>{});
    col_ab<K, 0>(lo, hi, x[2], x[3], std::make_integer_sequence<int, K + 1>{});
    col_ab<K, 0>(lo, hi, x[4], x[5], std::make_integer_sequence<int, K + 1>{});
    col_ab<K, 0>(lo, hi, x[6], x[7], std::make_integer_sequence<int, K + 1>{});
    if constexpr (K > 0) col_mp<K, 0>(lo, hi, m, std::make_integer_sequence<int, K>{});
    m[K] = (uint32_t)lo * PR::INV;
    macs(lo, hi, m[K], P<0>());
    acc_shift(lo, hi);
  }
  template <int K>
  static FF_HD void mul4_col_hi(uint64_t& lo, uint32_t& hi, const T* x, const uint32_t (&m)[8], T& r) {
    if constexpr (K < 15) {
      col_ab<K, K - 7>(lo, hi, x[0], x[1], std::make_integer_sequence<int, 15 - K>{});
      col_ab<K, K - 7>(lo, hi, x[2], x[3], std::make_integer_sequence<int, 15 - K>{});
      col_ab<K, K - 7>(lo, hi, x[4], x[5], std::make_integer_sequence<int, 15 - K>{});
      col_ab<K, K - 7>(lo, hi, x[6], x[7], std::make_integer_sequence<int, 15 - K>{});
      col_mp<K, K - 7>(lo, hi, m, std::make_integer_sequence<int, 15 - K>{});
    }
    r.v[K - 8] = (uint32_t)lo;
    acc_shift(lo, hi);
  }
  static FF_HD T mul4(const T& a, const T& b, const T& c, const T& d, const T& e, const T& f, const T& g, const T& h) {
    const T x[8] = {a, b, c, d, e, f, g, h};
    uint32_t m[8];
    T r;
    uint64_t lo = 0;
    uint32_t hi = 0;
    mul4_col_lo<0>(lo, hi, x, m); mul4_col_lo<1>(lo, hi, x, m); mul4_col_lo<2>(lo, hi, x, m);
    mul4_col_lo<3>(lo, hi, x, m); mul4_col_lo<4>(lo, hi, x, m); mul4_col_lo<5>(lo, hi, x, m);
    mul4_col_lo<6>(lo, hi, x, m); mul4_col_lo<7>(lo, hi, x, m);
    mul4_col_hi<8>(lo, hi, x, m, r); mul4_col_hi<9>(lo, hi, x, m, r); mul4_col_hi<10>(lo, hi, x, m, r);
    mul4_col_hi<11>(lo, hi, x, m, r); mul4_col_hi<12>(lo, hi, x, m, r); mul4_col_hi<13>(lo, hi, x, m, r);
    mul4_col_hi<14>(lo, hi, x, m, r); mul4_col_hi<15>(lo, hi, x, m, r);
    return reduce_once(r);
  }
  // a*b - c*d  in one Montgomery pass (the Y3 line of every addition / doubling formula)
  static FF_HD T mulsub(const T& a, const T& b, const T& c, const T& d) { return mul2(a, b, neg_raw(c), d); }

  // p - a as a plain integer (a <= p): a representative of -a in [0, p] (p itself for a = 0), fine as a mul2 operand
  static FF_HD T neg_raw(const T& a) {
    T r;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = subc(Pi(i), a.v[i], bw);
    return r;
  }

  // a^e for a 256-bit exponent given as 8 limbs (vartime; exponents here are public constants)
  static FF_HD T pow(const T& a, const uint32_t (&e)[8]) {
    T r = one();
    bool started = false;
    for (int i = 7; i >= 0; --i) {
      for (int b = 31; b >= 0; --b) {
        if (started) r = sqr(r);
        if ((e[i] >> b) & 1) {
          r = started ? mul(r, a) : a;
          started = true;
        }
      }
    }
    return r;
  }
  // a^(p-2)   (0 -> 0): Fermat inversion, ~380 modmuls
  static FF_HD T inv_fermat(const T& a) {
    uint32_t e[8];
    e[0] = PR::P0 - 2; e[1] = PR::P1; e[2] = PR::P2; e[3] = PR::P3;
    e[4] = PR::P4; e[5] = PR::P5; e[6] = PR::P6; e[7] = PR::P7;
    return pow(a, e);
  }
  // ---- helpers for the binary extended Euclid below (plain 256-bit integers) ----
  static FF_HD bool geq(const T& a, const T& b) {
    T t;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t.v[i] = subc(a.v[i], b.v[i], bw);
    return bw == 0;
  }
  static FF_HD T modulus() {
    T m;
#pragma unroll
    for (int i = 0; i < 8; ++i) m.v[i] = Pi(i);
    return m;
  }
  // a < modulus: the only encoding of a residue this library produces and (at the verifier) accepts
  static FF_HD bool is_canonical(const T& a) { return !geq(a, modulus()); }
  static FF_HD T raw_sub(const T& a, const T& b) {
    T t;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t.v[i] = subc(a.v[i], b.v[i], bw);
    return t;
  }
  static FF_HD T shr1(const T& a) {
    T r;
#pragma unroll
    for (int i = 0; i < 7; ++i) r.v[i] = (a.v[i] >> 1) | (a.v[i + 1] << 31);
    r.v[7] = a.v[7] >> 1;
    return r;
  }
  static FF_HD bool is_one_raw(const T& a) {
    uint32_t o = a.v[0] ^ 1u;
#pragma unroll
    for (int i = 1; i < 8; ++i) o |= a.v[i];
    return o == 0;
  }
  static FF_HD T r3() {   // R^3 mod p = R2 * R2 / R * ... computed once per call: mul(R2, R2) = R^3
    return mul(r2(), r2());
  }
  // ---- modular inverse -----------------------------------------------------------------------------------------
  // Bernstein-Yang division steps ("safegcd", https://gcd.cr.yp.to) in the batched form popularised by libsecp256k1's
  // modinv32: the values live in 9 signed limbs of 30 bits; 30 division steps at a time are run on the LOW WORDS of
  // (f, g) only -- a handful of 32-bit instructions per step -- and yield a 2x2 transition matrix that is then applied
  // to the full-size (f, g) and (d, e) with 64-bit multiply-adds.  Variable time: a batch stops dividing as soon as its
  // 30 steps are used up, the loop ends when g = 0 (~19 batches for a 254-bit prime).
  // One inversion costs about a tenth of the binary extended Euclid of rounds 1-2 (`inv_eea`, kept below for the
  // micro-benchmark): every step of that one shifts and subtracts four 256-bit values.  tools/ubench_inv measures both.
  struct S30 {
    int32_t v[9];
  };
  static FF_HD constexpr uint32_t PC(int i) {   // modulus limb i (32-bit), 0 beyond
    return i == 0 ? PR::P0 : i == 1 ? PR::P1 : i == 2 ? PR::P2 : i == 3 ? PR::P3 : i == 4 ? PR::P4 : i == 5 ? PR::P5
           : i == 6 ? PR::P6 : i == 7 ? PR::P7 : 0u;
  }
  static FF_HD constexpr int32_t mod30(int i) {   // modulus limb i in radix 2^30
    const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
    const uint64_t two = (uint64_t)PC(w) | ((uint64_t)PC(w + 1) << 32);
    return (int32_t)((two >> sh) & 0x3fffffffu);
  }
  static FF_HD constexpr uint32_t modinv30() {   // modulus^-1 mod 2^30 (Newton on the low word; the modulus is odd)
    uint32_t x = PC(0);                          // correct to 3 bits
    for (int k = 0; k < 4; ++k) x *= 2u - PC(0) * x;
    return x & 0x3fffffffu;
  }
  static FF_HD S30 to_s30(const T& x) {
    S30 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
      const uint64_t two = (uint64_t)x.v[w] | (w + 1 < 8 ? (uint64_t)x.v[w + 1] << 32 : 0);
      r.v[i] = (int32_t)((two >> sh) & 0x3fffffffu);
    }
    return r;
  }
  static FF_HD T from_s30(const S30& a) {   // a normalized: limbs in [0, 2^30), value < 2^256
    T x;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const int bit = 32 * w, i = bit / 30, sh = bit - 30 * i;
      uint64_t t = (uint64_t)(uint32_t)a.v[i] >> sh;
      t |= (uint64_t)(uint32_t)a.v[i + 1] << (30 - sh);
      if (i + 2 < 9 && 60 - sh < 32) t |= (uint64_t)(uint32_t)a.v[i + 2] << (60 - sh);
      x.v[w] = (uint32_t)t;
    }
    return x;
  }
  // up to 30 division steps on the low words f0 (odd), g0; eta = -delta.  -> new eta; t = (u, v, q, r) with
  // t * (f, g) = 2^30 * (f', g').  Zero bits of g are shifted out in one go; up to 8 bits of g are cancelled per
  // addition of a multiple of f (w = -g / f mod 2^k by Newton, where modinv32 reads a 128-byte table).
  static FF_HD int32_t divsteps30(int32_t eta, uint32_t f0, uint32_t g0, int32_t (&t)[4]) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
    int i = 30;
    for (;;) {
      const int zeros = __builtin_ctz(g | (0xffffffffu << i));
      g >>= zeros;
      u <<= zeros;
      v <<= zeros;
      eta -= zeros;
      i -= zeros;
      if (i == 0) break;
      if (eta < 0) {
        uint32_t tmp;
        eta = -eta;
        tmp = f, f = g, g = 0u - tmp;
        tmp = u, u = q, q = 0u - tmp;
        tmp = v, v = r, r = 0u - tmp;
      }
      // cancel min(eta + 1, i, 8) low bits of g: no more than i are left, and after eta + 1 the sign of eta flips
      const int limit = eta + 1 > i ? i : eta + 1;
      const uint32_t m = (0xffffffffu >> (32 - limit)) & 255u;
      uint32_t finv = f;                      // f^-1 mod 2^3 (f odd), then 2^6, then 2^12
      finv *= 2u - f * finv;
      finv *= 2u - f * finv;
      const uint32_t w = (g * (0u - finv)) & m;
      g += f * w;
      q += u * w;
      r += v * w;
    }
    t[0] = (int32_t)u, t[1] = (int32_t)v, t[2] = (int32_t)q, t[3] = (int32_t)r;
    return eta;
  }
  // (d, e) <- t * (d, e) / 2^30 mod p, limbs kept in (-2^30, 2^30), values in (-2p, p)
  static FF_HD void update_de30(S30& d, S30& e, const int32_t (&t)[4]) {
    constexpr int32_t M30 = 0x3fffffff;
    const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
    const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
    int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
    int64_t cd = (int64_t)u * d.v[0] + (int64_t)v * e.v[0];
    int64_t ce = (int64_t)q * d.v[0] + (int64_t)r * e.v[0];
    md -= (int32_t)((modinv30() * (uint32_t)cd + (uint32_t)md) & M30);
    me -= (int32_t)((modinv30() * (uint32_t)ce + (uint32_t)me) & M30);
    cd += (int64_t)mod30(0) * md;
    ce += (int64_t)mod30(0) * me;
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
      cd += (int64_t)u * d.v[i] + (int64_t)v * e.v[i] + (int64_t)mod30(i) * md;
      ce += (int64_t)q * d.v[i] + (int64_t)r * e.v[i] + (int64_t)mod30(i) * me;
      d.v[i - 1] = (int32_t)cd & M30;
      cd >>= 30;
      e.v[i - 1] = (int32_t)ce & M30;
      ce >>= 30;
    }
    d.v[8] = (int32_t)cd;
    e.v[8] = (int32_t)ce;
  }
  // (f, g) <- t * (f, g) / 2^30 (exact)
  static FF_HD void update_fg30(S30& f, S30& g, const int32_t (&t)[4]) {
    constexpr int32_t M30 = 0x3fffffff;
    const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
    int64_t cf = (int64_t)u * f.v[0] + (int64_t)v * g.v[0];
    int64_t cg = (int64_t)q * f.v[0] + (int64_t)r * g.v[0];
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
      cf += (int64_t)u * f.v[i] + (int64_t)v * g.v[i];
      cg += (int64_t)q * f.v[i] + (int64_t)r * g.v[i];
      f.v[i - 1] = (int32_t)cf & M30;
      cf >>= 30;
      g.v[i - 1] = (int32_t)cg & M30;
      cg >>= 30;
    }
    f.v[8] = (int32_t)cf;
    g.v[8] = (int32_t)cg;
  }
  // r in (-2p, p) -> [0, p), negated first if sign < 0
  static FF_HD void normalize30(S30& r, int32_t sign) {
    constexpr int32_t M30 = 0x3fffffff;
    const int32_t add1 = r.v[8] >> 31, neg = sign >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      r.v[i] += mod30(i) & add1;
      r.v[i] = (r.v[i] ^ neg) - neg;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      r.v[i + 1] += r.v[i] >> 30;
      r.v[i] &= M30;
    }
    const int32_t add2 = r.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] += mod30(i) & add2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      r.v[i + 1] += r.v[i] >> 30;
      r.v[i] &= M30;
    }
  }
  // x^-1 mod p of a canonical x (as an integer, NOT Montgomery); x != 0
  static FF_HD T inv_raw(const T& x) {
    S30 d, e, f, g = to_s30(x);
#pragma unroll
    for (int i = 0; i < 9; ++i) d.v[i] = 0, e.v[i] = 0, f.v[i] = mod30(i);
    e.v[0] = 1;
    int32_t eta = -1;
    for (;;) {
      int32_t t[4];
      eta = divsteps30(eta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
      update_de30(d, e, t);
      update_fg30(f, g, t);
      int32_t any = 0;
#pragma unroll
      for (int i = 0; i < 9; ++i) any |= g.v[i];
      if (any == 0) break;
    }
    normalize30(d, f.v[8]);   // f = +-1 now, d = +-x^-1
    return from_s30(d);
  }
  // In: a*R, out: a^-1 * R.  0 -> 0.
  static FF_HD T inv(const T& a) {
    if (is_zero(a)) return a;
    // (aR)^-1 = a^-1 R^-1 as an integer; one Montgomery product with R^3 gives a^-1 R
    return mul(inv_raw(a), r3());
  }
  // Rounds 1-2: binary extended Euclidean algorithm (vartime; ~2*254 shift/subtract steps of 8-limb integer ops).
  // Same contract as inv().  Kept for tools/ubench_inv.
  static FF_HD T inv_eea(const T& a) {
    if (is_zero(a)) return a;
    T u = a, v, x1 = zero(), x2 = zero();
#pragma unroll
    for (int i = 0; i < 8; ++i) v.v[i] = Pi(i);
    x1.v[0] = 1;   // invariants: x1 * a == u, x2 * a == v (mod p)
    while (!is_one_raw(u) && !is_one_raw(v)) {
      while ((u.v[0] & 1) == 0) {
        u = shr1(u);
        x1 = div2(x1);
      }
      while ((v.v[0] & 1) == 0) {
        v = shr1(v);
        x2 = div2(x2);
      }
      if (geq(u, v)) {
        u = raw_sub(u, v);
        x1 = sub(x1, x2);
      } else {
        v = raw_sub(v, u);
        x2 = sub(x2, x1);
      }
    }
    return mul(is_one_raw(u) ? x1 : x2, r3());
  }
  static FF_HD T mul_small(const T& a, uint32_t k) {  // k in {2,3,4,8}: via doublings/adds
    T r = a;
    if (k == 2) return dbl(a);
    if (k == 3) return add(dbl(a), a);
    if (k == 4) return dbl(dbl(a));
    if (k == 8) return dbl(dbl(dbl(a)));
    (void)r;
    return a;
  }
};

using Fp = Field<FpParams>;
using Fr = Field<FrParams>;

// ---- Fp2 = Fp[u]/(u^2+1): (c0, c1), 64 bytes  (fields.nim:27-32, coords:[i,u]) -------------------
struct fp2_t {
  u256 c0, c1;
};

struct Fp2 {
  using T = fp2_t;
  static FF_HD T zero() { return T{Fp::zero(), Fp::zero()}; }
  static FF_HD T one() { return T{Fp::one(), Fp::zero()}; }
  static FF_HD bool is_zero(const T& a) { return Fp::is_zero(a.c0) && Fp::is_zero(a.c1); }
  static FF_HD bool eq(const T& a, const T& b) { return Fp::eq(a.c0, b.c0) && Fp::eq(a.c1, b.c1); }
  static FF_HD T add(const T& a, const T& b) { return T{Fp::add(a.c0, b.c0), Fp::add(a.c1, b.c1)}; }
  static FF_HD T sub(const T& a, const T& b) { return T{Fp::sub(a.c0, b.c0), Fp::sub(a.c1, b.c1)}; }
  static FF_HD T neg(const T& a) { return T{Fp::neg(a.c0), Fp::neg(a.c1)}; }
  static FF_HD T dbl(const T& a) { return T{Fp::dbl(a.c0), Fp::dbl(a.c1)}; }
  // (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u : two Montgomery dot products (Fp::mul2)
  static FF_HD_CALL T mul(const T& a, const T& b) {
    return T{Fp::mul2(a.c0, b.c0, Fp::neg_raw(a.c1), b.c1), Fp::mul2(a.c0, b.c1, a.c1, b.c0)};
  }
  // a*b - c*d over Fp2: each component is a four-term dot product
  //   re = a0 b0 - a1 b1 - c0 d0 + c1 d1 ,   im = a0 b1 + a1 b0 - c0 d1 - c1 d0
  static FF_HD_CALL T mulsub(const T& a, const T& b, const T& c, const T& d) {
    const u256 na1 = Fp::neg_raw(a.c1), nc0 = Fp::neg_raw(c.c0), nc1 = Fp::neg_raw(c.c1);
    return T{Fp::mul4(a.c0, b.c0, na1, b.c1, nc0, d.c0, c.c1, d.c1),
             Fp::mul4(a.c0, b.c1, a.c1, b.c0, nc0, d.c1, nc1, d.c0)};
  }
  // complex squaring: 2 base-field products
  static FF_HD_CALL T sqr(const T& a) {
    u256 t = Fp::mul(a.c0, a.c1);
    u256 c0 = Fp::mul(Fp::add(a.c0, a.c1), Fp::sub(a.c0, a.c1));
    return T{c0, Fp::dbl(t)};
  }
  static FF_HD T inv(const T& a) {
    u256 n = Fp::add(Fp::sqr(a.c0), Fp::sqr(a.c1));
    u256 d = Fp::inv(n);
    return T{Fp::mul(a.c0, d), Fp::neg(Fp::mul(a.c1, d))};
  }
  static FF_HD T mul_small(const T& a, uint32_t k) { return T{Fp::mul_small(a.c0, k), Fp::mul_small(a.c1, k)}; }
};

}  // namespace g16
