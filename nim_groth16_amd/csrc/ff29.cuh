// BN254 Fp in reduced radix for the bucket-accumulation kernels: 9 limbs of 29 bits, Montgomery radix 2^261.
//
// Why: on gfx950 a 32x32->64 multiply-add (v_mad_u64_u32) issues in ~4.9 cycles/wave and every carry
// instruction behind it (v_addc_co_u32) in ~4.3 (profiles/r01_ubench_int_issue_rates.txt).  With full 32-bit
// limbs (ff.cuh) every partial product needs one of each.  With 29-bit limbs a whole column of the product
// (<= 9 a_i*b_j plus 9 m_i*p_j) fits the 64-bit accumulator of v_mad_u64_u32, so a column is a carry-free
// chain of multiply-adds followed by one 64-bit shift: 1161 -> 908 SIMD cycles per wave-multiplication
// measured (tools/ubench_mul29.hip), and a dedicated squaring and multi-term dot products come for free.
//
// Values are lazily reduced: limbs may exceed 29 bits and values may exceed p, within bounds that are stated
// at each function and PROVEN for the mixed-addition and general-addition sequences by the interval model tools/ff29_model.py
// (it executes the same operation sequence on worst-case bounds and checks every 64-bit column, every 32-bit
// limb and the loop invariant).  "normalized" = limbs 0..7 < 2^29; the top limb carries whatever is left.
//
// The representation stays inside the accumulate -> heavy -> reduce1 kernels: point tables are converted once at
// registration (Ec29::tab_from_std, packed 64/128-byte entries) and the 16-bucket chunk sums are converted back
// to the 8x32 / R=2^256 layout of ff.cuh (Ec29::to_std) before the latency-bound tail (reduce2, fold) sees them.
#pragma once
#include <utility>

#include "ff.cuh"

// Every multiply-add of a column is pinned behind the previous one by an empty (non-volatile) asm, so that the
// column sum is ONE serial v_mad_u64_u32 chain starting from the carry of the previous column.  Left alone, the
// compiler starts each column from zero (shorter critical path, which 4 waves/SIMD do not need) and joins it with
// the carry through an extra 64-bit add per column (144 v_lshl_add_u64 per mixed addition).  The asm costs an
// s_nop each (hazard padding), still a net win where occupancy hides the longer chain: msm_accum_g1 (4 waves/SIMD)
// 1.007 -> 0.956 ms.  At 2 waves/SIMD (G2 accumulate) one serial chain per wave stalls, so that kernel computes its
// Fp2 products as two interleaved pinned chains (dot_pair, -DG16_F29_PAIR): 2.83 -> 2.69 ms.  The reduce kernels
// (1 wave/SIMD) get slower with pins; only the two accumulate translation units are built with -DG16_F29_SERIAL
// (Makefile).  (A *volatile* asm orders all field operations against each other and made the kernel 3x slower.)
#if defined(__HIP_DEVICE_COMPILE__) && defined(G16_F29_SERIAL)
#define F29_MAC(acc, a, b)              \
  do {                                  \
    acc += (uint64_t)(a) * (b);         \
    asm("" : "+v"(acc));                \
  } while (0)
#else
#define F29_MAC(acc, a, b) acc += (uint64_t)(a) * (b)
#endif

namespace g16 {

struct fe29 {
  uint32_t v[9];
};

#include "ff29_mac.inc"

struct Limbs29 {
  uint32_t v[9];
};
// the two BN254 fields in this radix.  PL: the modulus; N0 = -m^-1, PINV0 = m^-1 (mod 2^29); ONE = 2^261, C_IN = 2^266
// (x*2^256 -> x*2^261), C_OUT = 2^256 (x*2^261 -> x*2^256), all mod m.
struct Fp29Params {
  static constexpr uint32_t N0 = 0x4866389u, PINV0 = 0x1b799c77u;
  static constexpr Limbs29 PL = {{0x187cfd47u, 0x10460b6u, 0x1c72a34fu, 0x2d522d0u, 0x1585d978u, 0x2db40c0u,
                                  0xa6e141u, 0xe5c2634u, 0x30644eu}};
  static constexpr Limbs29 ONE = {{0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x14c0419u, 0xaa36fb9u, 0x1d4240ceu,
                                   0x11d54c07u, 0x52ac7a8u, 0xdc836u}};
  static constexpr Limbs29 C_IN = {{0x13349ca1u, 0x1a5d84a8u, 0xa3e5cacu, 0x100249e0u, 0x12b951e8u, 0xe92d304u,
                                    0x14cb95b3u, 0x41b9d3du, 0x58003u}};
  static constexpr Limbs29 C_OUT = {{0x58f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u, 0x11b7bc3cu,
                                     0x1cbd99bau, 0x183340fbu, 0xe0a77u}};
};
template <class PR>
struct Field29 {
  static constexpr int B = 29, L = 9;
  static constexpr uint32_t MASK = (1u << B) - 1;
  static constexpr uint32_t N0 = PR::N0;        // -m^-1 mod 2^29
  static constexpr uint32_t PINV0 = PR::PINV0;  //  m^-1 mod 2^29
  using Limbs = Limbs29;
  static constexpr Limbs PL = PR::PL;
  static constexpr Limbs ONE = PR::ONE;      // 2^261 mod m
  static constexpr Limbs C_IN = PR::C_IN;    // 2^266 mod m: x*2^256 -> x*2^261
  static constexpr Limbs C_OUT = PR::C_OUT;  // 2^256 mod m: x*2^261 -> x*2^256

  // limbs of MULT*p with LIFT units borrowed from every higher limb into the one below ("borrow form"):
  // same value, but limbs 0..7 >= LIFT*(2^29-1), so that  a + K - b  stays non-negative limb by limb
  template <uint32_t MULT, uint32_t LIFT>
  static constexpr Limbs kform() {
    Limbs k{};
    uint64_t carry = 0;
    for (int i = 0; i < L; ++i) {
      uint64_t t = (uint64_t)PL.v[i] * MULT + carry;
      k.v[i] = i < L - 1 ? (uint32_t)(t & MASK) : (uint32_t)t;
      carry = t >> B;
    }
    for (int i = 0; i < L - 1; ++i) {
      k.v[i] += LIFT << B;
      k.v[i + 1] -= LIFT;
    }
    return k;
  }

  static FF_HD fe29 zero() {
    fe29 r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = 0;
    return r;
  }
  static FF_HD fe29 one() {
    fe29 r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = ONE.v[i];
    return r;
  }
  static FF_HD fe29 constant(const Limbs& c) {
    fe29 r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = c.v[i];
    return r;
  }
  static FF_HD bool limbs_zero(const fe29& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) o |= a.v[i];
    return o == 0;
  }

  // ---- column chains as single asm statements (-DG16_F29_ASM: the G1 accumulate kernel) ----------------------
  // Column k of a product: the a_i*b_(k-i) for i = lo..hi, then the m_i*p_(k-i) of the interleaved Montgomery
  // reduction.  Each part is ONE asm statement of back-to-back v_mad_u64_u32 on the same accumulator
  // (ff29_mac.inc): a serial chain without the compiler's per-column join add, like the pins of F29_MAC, but
  // with one hipcc s_nop per statement instead of one per product.
  template <int K>
  static constexpr int col_lo() { return K < L ? 0 : K - L + 1; }
  template <int K>
  static constexpr int col_cnt() { return K < L ? K + 1 : 2 * L - 1 - K; }   // products a_i*b_(K-i)
  template <int K>
  static constexpr int red_cnt() { return K < L ? K : 2 * L - 1 - K; }       // m_i*p_(K-i) with m_i already known
  template <int K, int... I>
  static FF_HD void col_mul(uint64_t& acc, const fe29& a, const fe29& b, std::integer_sequence<int, I...>) {
    mac29v(acc, a.v[col_lo<K>() + I]..., b.v[K - col_lo<K>() - I]...);
  }
  template <int K, int... I>
  static FF_HD void col_red(uint64_t& acc, const uint32_t (&m)[L], std::integer_sequence<int, I...>) {
    mac29s(acc, m[col_lo<K>() + I]..., PL.v[K - col_lo<K>() - I]...);
  }
  // squaring: the products i < K-i against the doubled operand, i == K-i against itself
  template <int K, int I>
  static FF_HD uint32_t sqr_rhs(const fe29& a, const uint32_t (&a2)[L]) {
    constexpr int i = col_lo<K>() + I, j = K - i;
    static_assert(i <= j, "upper triangle");
    return i == j ? a.v[j] : a2[j];
  }
  template <int K, int... I>
  static FF_HD void col_sqr(uint64_t& acc, const fe29& a, const uint32_t (&a2)[L], std::integer_sequence<int, I...>) {
    mac29v(acc, a.v[col_lo<K>() + I]..., sqr_rhs<K, I>(a, a2)...);
  }
  template <int K>
  static FF_HD void col_finish(uint64_t& acc, uint32_t (&m)[L], fe29& r) {
    if constexpr (red_cnt<K>() > 0) col_red<K>(acc, m, std::make_integer_sequence<int, red_cnt<K>()>{});
    if constexpr (K < L) {
      m[K] = ((uint32_t)acc * N0) & MASK;
      acc += (uint64_t)m[K] * PL.v[0];
    } else {
      r.v[K - L] = (uint32_t)acc & MASK;
    }
    acc >>= B;
  }
  template <int NP, int K>
  static FF_HD void dot_col(uint64_t& acc, uint32_t (&m)[L], fe29& r, const fe29& a0, const fe29& b0, const fe29& a1,
                            const fe29& b1, const fe29& a2, const fe29& b2, const fe29& a3, const fe29& b3) {
    using Seq = std::make_integer_sequence<int, col_cnt<K>()>;
    col_mul<K>(acc, a0, b0, Seq{});
    if constexpr (NP > 1) col_mul<K>(acc, a1, b1, Seq{});
    if constexpr (NP > 2) col_mul<K>(acc, a2, b2, Seq{});
    if constexpr (NP > 3) col_mul<K>(acc, a3, b3, Seq{});
    col_finish<K>(acc, m, r);
  }
  template <int NP, int... K>
  static FF_HD fe29 dot_asm(const fe29& a0, const fe29& b0, const fe29& a1, const fe29& b1, const fe29& a2,
                            const fe29& b2, const fe29& a3, const fe29& b3, std::integer_sequence<int, K...>) {
    uint32_t m[L];
    fe29 r;
    uint64_t acc = 0;
    (dot_col<NP, K>(acc, m, r, a0, b0, a1, b1, a2, b2, a3, b3), ...);
    r.v[L - 1] = (uint32_t)acc;
    return r;
  }
  template <int K>
  static FF_HD void sqr_col(uint64_t& acc, uint32_t (&m)[L], fe29& r, const fe29& a, const uint32_t (&a2)[L]) {
    col_sqr<K>(acc, a, a2, std::make_integer_sequence<int, (col_cnt<K>() + 1) / 2>{});
    col_finish<K>(acc, m, r);
  }
  template <int... K>
  static FF_HD fe29 sqr_asm(const fe29& a, std::integer_sequence<int, K...>) {
    uint32_t m[L], a2[L];
    fe29 r;
#pragma unroll
    for (int i = 0; i < L; ++i) a2[i] = a.v[i] << 1;
    uint64_t acc = 0;
    (sqr_col<K>(acc, m, r, a, a2), ...);
    r.v[L - 1] = (uint32_t)acc;
    return r;
  }

  // (sum over NP pairs of a_j*b_j) / 2^261 mod p.  Every 64-bit column sum must stay below 2^64: with limb
  // bounds la_j, lb_j that is  9*(sum_j la_j*lb_j + 2^58) < 2^64.  Result: normalized, < (sum a_j b_j)/2^261 + p.
  template <int NP>
  static FF_HD fe29 dot(const fe29& a0, const fe29& b0, const fe29& a1, const fe29& b1, const fe29& a2,
                        const fe29& b2, const fe29& a3, const fe29& b3) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(G16_F29_ASM)
    return dot_asm<NP>(a0, b0, a1, b1, a2, b2, a3, b3, std::make_integer_sequence<int, 2 * L - 1>{});
#endif
    uint32_t m[L];
    fe29 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * L - 1; ++k) {
      const int lo = k < L ? 0 : k - L + 1, hi = k < L ? k : L - 1;
#pragma unroll
      for (int i = lo; i <= hi; ++i) {
        F29_MAC(acc, a0.v[i], b0.v[k - i]);
        if constexpr (NP > 1) F29_MAC(acc, a1.v[i], b1.v[k - i]);
        if constexpr (NP > 2) F29_MAC(acc, a2.v[i], b2.v[k - i]);
        if constexpr (NP > 3) F29_MAC(acc, a3.v[i], b3.v[k - i]);
      }
#pragma unroll
      for (int i = lo; i <= hi; ++i)
        if (!(k < L && i == k)) F29_MAC(acc, m[i], PL.v[k - i]);
      if (k < L) {
        m[k] = ((uint32_t)acc * N0) & MASK;
        F29_MAC(acc, m[k], PL.v[0]);
      } else {
        r.v[k - L] = (uint32_t)acc & MASK;
      }
      acc >>= B;
    }
    r.v[L - 1] = (uint32_t)acc;
    return r;
  }
  // ---- chain PAIRS as asm statements (-DG16_F29_PAIR_ASM: the G2 accumulate kernel) ---------------------------------
  // dot_pair below with its two lockstep chains written as multi-instruction asm statements (mac29p / mac29ps,
  // ff29_mac.inc): the same instruction order as the pinned C++ chains, one hipcc s_nop per statement (<= 14
  // multiply-adds) instead of one per product.  An asm statement takes <= 30 operands: <= 7 product pairs of VGPR
  // factors, so a full column part (9 products per chain) is two statements.
  template <int K, int OFF, int... I>
  static FF_HD void pair_mul(uint64_t& acc, uint64_t& bcc, const fe29& a, const fe29& b, const fe29& c, const fe29& d,
                             std::integer_sequence<int, I...>) {
    mac29p(acc, bcc, a.v[col_lo<K>() + OFF + I]..., b.v[K - col_lo<K>() - OFF - I]..., c.v[col_lo<K>() + OFF + I]...,
           d.v[K - col_lo<K>() - OFF - I]...);
  }
  template <int K>
  static FF_HD void pair_term(uint64_t& acc, uint64_t& bcc, const fe29& a, const fe29& b, const fe29& c, const fe29& d) {
    constexpr int n = col_cnt<K>(), first = n <= 7 ? n : (n + 1) / 2;
    pair_mul<K, 0>(acc, bcc, a, b, c, d, std::make_integer_sequence<int, first>{});
    if constexpr (n > first) pair_mul<K, first>(acc, bcc, a, b, c, d, std::make_integer_sequence<int, n - first>{});
  }
  template <int K, int... I>
  static FF_HD void pair_red(uint64_t& acc, uint64_t& bcc, const uint32_t (&m)[L], const uint32_t (&n)[L],
                             std::integer_sequence<int, I...>) {
    mac29ps(acc, bcc, m[col_lo<K>() + I]..., n[col_lo<K>() + I]..., PL.v[K - col_lo<K>() - I]...);
  }
  template <int NP, int K>
  static FF_HD void pair_col(uint64_t& acc, uint64_t& bcc, uint32_t (&m)[L], uint32_t (&n)[L], fe29& r0, fe29& r1,
                             const fe29& a0, const fe29& b0, const fe29& a1, const fe29& b1, const fe29& a2,
                             const fe29& b2, const fe29& a3, const fe29& b3, const fe29& c0, const fe29& d0,
                             const fe29& c1, const fe29& d1, const fe29& c2, const fe29& d2, const fe29& c3,
                             const fe29& d3) {
    pair_term<K>(acc, bcc, a0, b0, c0, d0);
    if constexpr (NP > 1) pair_term<K>(acc, bcc, a1, b1, c1, d1);
    if constexpr (NP > 2) pair_term<K>(acc, bcc, a2, b2, c2, d2);
    if constexpr (NP > 3) pair_term<K>(acc, bcc, a3, b3, c3, d3);
    if constexpr (red_cnt<K>() > 0) pair_red<K>(acc, bcc, m, n, std::make_integer_sequence<int, red_cnt<K>()>{});
    if constexpr (K < L) {
      m[K] = ((uint32_t)acc * N0) & MASK;
      n[K] = ((uint32_t)bcc * N0) & MASK;
      acc += (uint64_t)m[K] * PL.v[0];
      bcc += (uint64_t)n[K] * PL.v[0];
    } else {
      r0.v[K - L] = (uint32_t)acc & MASK;
      r1.v[K - L] = (uint32_t)bcc & MASK;
    }
    acc >>= B;
    bcc >>= B;
  }
  template <int NP, int... K>
  static FF_HD void dot_pair_asm(fe29& r0, fe29& r1, const fe29& a0, const fe29& b0, const fe29& a1, const fe29& b1,
                                 const fe29& a2, const fe29& b2, const fe29& a3, const fe29& b3, const fe29& c0,
                                 const fe29& d0, const fe29& c1, const fe29& d1, const fe29& c2, const fe29& d2,
                                 const fe29& c3, const fe29& d3, std::integer_sequence<int, K...>) {
    uint32_t m[L], n[L];
    uint64_t acc = 0, bcc = 0;
    (pair_col<NP, K>(acc, bcc, m, n, r0, r1, a0, b0, a1, b1, a2, b2, a3, b3, c0, d0, c1, d1, c2, d2, c3, d3), ...);
    r0.v[L - 1] = (uint32_t)acc;
    r1.v[L - 1] = (uint32_t)bcc;
  }

  // two independent NP-term dot products column by column in lockstep (r0 = sum a_j*b_j, r1 = sum c_j*d_j): with
  // the serial pins of F29_MAC each result is one multiply-add chain without join adds, and the two chains give
  // a wave the instruction-level parallelism that a single serial chain lacks at 2 waves/SIMD (G2 kernels)
  template <int NP>
  static FF_HD void dot_pair(fe29& r0, fe29& r1, const fe29& a0, const fe29& b0, const fe29& a1, const fe29& b1,
                             const fe29& a2, const fe29& b2, const fe29& a3, const fe29& b3, const fe29& c0,
                             const fe29& d0, const fe29& c1, const fe29& d1, const fe29& c2, const fe29& d2,
                             const fe29& c3, const fe29& d3) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(G16_F29_PAIR_ASM)
    dot_pair_asm<NP>(r0, r1, a0, b0, a1, b1, a2, b2, a3, b3, c0, d0, c1, d1, c2, d2, c3, d3,
                     std::make_integer_sequence<int, 2 * L - 1>{});
    return;
#endif
    uint32_t m[L], n[L];
    uint64_t acc = 0, bcc = 0;
#pragma unroll
    for (int k = 0; k < 2 * L - 1; ++k) {
      const int lo = k < L ? 0 : k - L + 1, hi = k < L ? k : L - 1;
#pragma unroll
      for (int i = lo; i <= hi; ++i) {
        F29_MAC(acc, a0.v[i], b0.v[k - i]);
        F29_MAC(bcc, c0.v[i], d0.v[k - i]);
        if constexpr (NP > 1) F29_MAC(acc, a1.v[i], b1.v[k - i]);
        if constexpr (NP > 1) F29_MAC(bcc, c1.v[i], d1.v[k - i]);
        if constexpr (NP > 2) F29_MAC(acc, a2.v[i], b2.v[k - i]);
        if constexpr (NP > 2) F29_MAC(bcc, c2.v[i], d2.v[k - i]);
        if constexpr (NP > 3) F29_MAC(acc, a3.v[i], b3.v[k - i]);
        if constexpr (NP > 3) F29_MAC(bcc, c3.v[i], d3.v[k - i]);
      }
#pragma unroll
      for (int i = lo; i <= hi; ++i)
        if (!(k < L && i == k)) {
          F29_MAC(acc, m[i], PL.v[k - i]);
          F29_MAC(bcc, n[i], PL.v[k - i]);
        }
      if (k < L) {
        m[k] = ((uint32_t)acc * N0) & MASK;
        n[k] = ((uint32_t)bcc * N0) & MASK;
        F29_MAC(acc, m[k], PL.v[0]);
        F29_MAC(bcc, n[k], PL.v[0]);
      } else {
        r0.v[k - L] = (uint32_t)acc & MASK;
        r1.v[k - L] = (uint32_t)bcc & MASK;
      }
      acc >>= B;
      bcc >>= B;
    }
    r0.v[L - 1] = (uint32_t)acc;
    r1.v[L - 1] = (uint32_t)bcc;
  }
  static FF_HD fe29 mul(const fe29& a, const fe29& b) { return dot<1>(a, b, a, b, a, b, a, b); }
  static FF_HD fe29 dot2(const fe29& a0, const fe29& b0, const fe29& a1, const fe29& b1) {
    return dot<2>(a0, b0, a1, b1, a0, b0, a0, b0);
  }
  static FF_HD fe29 dot4(const fe29& a0, const fe29& b0, const fe29& a1, const fe29& b1, const fe29& a2,
                         const fe29& b2, const fe29& a3, const fe29& b3) {
    return dot<4>(a0, b0, a1, b1, a2, b2, a3, b3);
  }
  // a^2 / 2^261: the cross products once, against the doubled operand (45 + 81 multiply-adds instead of 162)
  static FF_HD fe29 sqr(const fe29& a) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(G16_F29_ASM)
    return sqr_asm(a, std::make_integer_sequence<int, 2 * L - 1>{});
#endif
    uint32_t m[L], a2[L];
    fe29 r;
#pragma unroll
    for (int i = 0; i < L; ++i) a2[i] = a.v[i] << 1;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * L - 1; ++k) {
      const int lo = k < L ? 0 : k - L + 1, hi = k < L ? k : L - 1;
#pragma unroll
      for (int i = lo; i <= hi; ++i) {
        if (i < k - i) F29_MAC(acc, a.v[i], a2[k - i]);
        if (i == k - i) F29_MAC(acc, a.v[i], a.v[i]);
      }
#pragma unroll
      for (int i = lo; i <= hi; ++i)
        if (!(k < L && i == k)) F29_MAC(acc, m[i], PL.v[k - i]);
      if (k < L) {
        m[k] = ((uint32_t)acc * N0) & MASK;
        F29_MAC(acc, m[k], PL.v[0]);
      } else {
        r.v[k - L] = (uint32_t)acc & MASK;
      }
      acc >>= B;
    }
    r.v[L - 1] = (uint32_t)acc;
    return r;
  }

  static FF_HD fe29 add(const fe29& a, const fe29& b) {
    fe29 r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = a.v[i] + b.v[i];
    return r;
  }
  // a - b + MULT*p, limb by limb (no carries): kform<MULT,LIFT> must cover the limb bounds of b
  template <uint32_t MULT, uint32_t LIFT>
  static FF_HD fe29 subk(const fe29& a, const fe29& b) {
    constexpr Limbs k = kform<MULT, LIFT>();
    fe29 r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = a.v[i] + k.v[i] - b.v[i];
    return r;
  }
  // a - b - 2c + MULT*p
  template <uint32_t MULT, uint32_t LIFT>
  static FF_HD fe29 subk2(const fe29& a, const fe29& b, const fe29& c) {
    constexpr Limbs k = kform<MULT, LIFT>();
    fe29 r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = a.v[i] + k.v[i] - b.v[i] - 2 * c.v[i];
    return r;
  }
  template <uint32_t MULT, uint32_t LIFT>
  static FF_HD fe29 negk(const fe29& b) {
    constexpr Limbs k = kform<MULT, LIFT>();
    fe29 r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = k.v[i] - b.v[i];
    return r;
  }
  // carry propagation: limbs 0..7 < 2^29 afterwards, same value
  static FF_HD fe29 norm(const fe29& a) {
    fe29 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < L - 1; ++i) {
      uint32_t t = a.v[i] + c;
      r.v[i] = t & MASK;
      c = t >> B;
    }
    r.v[L - 1] = a.v[L - 1] + c;
    return r;
  }
  // unique representative in [0, p) of a normalized value < 2*MAXMULT*p (MAXMULT a power of two)
  template <uint32_t MAXMULT>
  static FF_HD fe29 canon(fe29 a) {
    if constexpr (MAXMULT > 1) a = canon_step<MAXMULT>(a);
    if constexpr (MAXMULT > 1) return canon<MAXMULT / 2>(a);
    return canon_step<1>(a);
  }
  template <uint32_t MULT>
  static FF_HD fe29 canon_step(const fe29& a) {  // a >= MULT*p ? a - MULT*p : a
    constexpr Limbs k = kform<MULT, 0>();
    fe29 t;
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < L - 1; ++i) {
      int32_t d = (int32_t)a.v[i] - (int32_t)k.v[i] + borrow;
      t.v[i] = (uint32_t)d & MASK;
      borrow = d >> B;
    }
    int32_t top = (int32_t)a.v[L - 1] - (int32_t)k.v[L - 1] + borrow;
    t.v[L - 1] = (uint32_t)top;
    fe29 r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = top < 0 ? a.v[i] : t.v[i];
    return r;
  }
  // cheap necessary condition for a == 0 mod p of a normalized value < 32p: a = j*p forces a_0*p^-1 = j < 32
  // (mod 2^29); false positives with probability 2^-24
  static FF_HD bool maybe_zero(const fe29& a) { return ((a.v[0] * PINV0) & MASK) < 32u; }
  template <uint32_t MAXMULT>
  static FF_HD bool is_zero_exact(const fe29& a) {
    return limbs_zero(canon<MAXMULT>(a));
  }

  // ---- layout conversion with ff.cuh (8 x 32-bit limbs) ----
  static FF_HD fe29 relimb(const u256& x) {
    fe29 r;
#pragma unroll
    for (int i = 0; i < L; ++i) {
      const int bit = B * i, w = bit >> 5, sh = bit & 31;
#if defined(__HIP_DEVICE_COMPILE__)
      // one funnel shift per limb (v_alignbit_b32) instead of 64-bit shifts
      uint32_t t = sh == 0 ? x.v[w] : (w + 1 < 8 ? __builtin_amdgcn_alignbit(x.v[w + 1], x.v[w], sh) : x.v[w] >> sh);
      r.v[i] = t & MASK;
#else
      uint64_t two = (uint64_t)x.v[w] | (w + 1 < 8 ? (uint64_t)x.v[w + 1] << 32 : 0);
      r.v[i] = (uint32_t)(two >> sh) & MASK;
#endif
    }
    return r;
  }
  static FF_HD u256 relimb(const fe29& a) {  // a canonical
    u256 x;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const int bit = 32 * w, i = bit / B, sh = bit - B * i;
      uint64_t t = (uint64_t)a.v[i] >> sh;
      t |= (uint64_t)a.v[i + 1] << (B - sh);
      if (i + 2 < L && 2 * B - sh < 32) t |= (uint64_t)a.v[i + 2] << (2 * B - sh);
      x.v[w] = (uint32_t)t;
    }
    return x;
  }
  // Montgomery form of ff.cuh (x*2^256 mod p, canonical)  <->  canonical x*2^261 mod p
  static FF_HD fe29 from_std(const u256& x) { return canon<1>(mul(relimb(x), constant(C_IN))); }
  // any normalized value < 13p
  static FF_HD u256 to_std(const fe29& a) { return relimb(canon<1>(mul(a, constant(C_OUT)))); }
};
using Fp29 = Field29<Fp29Params>;

}  // namespace g16
