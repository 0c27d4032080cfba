// NTT passes with the butterflies in the reduced-radix field (9 x 29-bit limbs, ff29.cuh) over the SCALAR field r.
// Same decomposition, same butterflies in the same order and the same canonical outputs as ntt.cuh (which follows
// groth16/math/ntt.nim:17-161); what changes is the arithmetic inside a pass:
//
//  * Data stay in the reference's Montgomery form x * 2^256 (what the 8 x 32 kernels, the MSM scalars and the C ABI
//    use) but live as 9 x 29-bit limbs, lazily reduced, from the load of a pass to its store.  Every multiplication
//    of an NTT is data x twiddle, so the twiddle tables are kept in the 2^261 form of ff29.cuh and
//    mul(x * 2^256, w * 2^261) = x w * 2^256 needs no conversion.  (The one data x data product, A1*B1 of the fused
//    last pass, is followed by one multiplication by the constant 2^266.)
//  * A multiplication is 162 carry-free multiply-adds + 54 other instructions instead of 128 + 128 carry
//    instructions + ..., additions and subtractions are limb-wise without carries or comparisons (a - b adds a
//    multiple of r in borrow form), and nothing is compared with r inside the tile.
//  * Bounds (exercised AT the bounds, maximal limbs included, against Python integers by the CPU test
//    tests/test_device_headers_cpu.py::test_ntt29_radix4_group_at_its_bounds; derivation in group4 below): a value
//    that enters a radix-4 round trip below V r leaves it below (4 V + 1) r, products are below 6 r; the third round
//    trip brings every output that did not come out of a multiplication back below 2 r (weak_reduce: one estimated
//    quotient from the top limb), so V <= 197 at the end of a 10-stage pass and every limb stays below 2^32, every
//    product column below 2^64.  Stores to HBM are canonical (the next pass and the caller see exactly ntt.cuh's bytes).
//  * LDS holds 36 B per element, struct-of-arrays (limb-major: conflict-free 4-byte accesses): a 2048-element tile is
//    72 KB, two workgroups per CU as before; the inner twiddles come from a compact 18-KB table in global memory (L1 /
//    L2 resident) instead of LDS.
#pragma once
#include "ff29.cuh"
#if defined(__HIPCC__)
#include "ntt.cuh"   // tile order, bit reversal, the 8 x 32 tables the reduced-radix ones are converted from
#endif

namespace g16 {

struct Ntt29 {
  using F = Fr29;
  static constexpr int L = 9;
  static constexpr uint32_t MASK = F::MASK;
  static constexpr uint32_t RTOP1 = 0x30644eu + 1u;                                  // (r >> 232) + 1
  static constexpr uint64_t RECIP = ((uint64_t(1) << 48) + RTOP1 - 1) / RTOP1;      // ceil(2^48 / RTOP1)
  static constexpr uint32_t PROD_MULT = 8;   // products are < 6 r: 8 r in borrow form covers them limb by limb

  // mult * r with `lift` units borrowed from every higher limb into the one below: same value, limbs 0..7 >=
  // lift * (2^29 - 1), so that  a + K - b  stays non-negative limb by limb for b with limbs <= lift * (2^29 - 1) and
  // value < mult * r (top limb)
  static FF_HD fe29 kmult(uint32_t mult, uint32_t lift) {
    fe29 k;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
      const uint64_t t = (uint64_t)F::PL.v[i] * mult + c;
      k.v[i] = i < L - 1 ? (uint32_t)(t & MASK) : (uint32_t)t;
      c = t >> 29;
    }
#pragma unroll
    for (int i = 0; i < L - 1; ++i) {
      k.v[i] += lift << 29;
      k.v[i + 1] -= lift;
    }
    return k;
  }
  static FF_HD fe29 addl(const fe29& a, const fe29& b) {
    fe29 r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = a.v[i] + b.v[i];
    return r;
  }
  static FF_HD fe29 subl(const fe29& a, const fe29& b, const fe29& k) {   // a - b + k
    fe29 r;
#pragma unroll
    for (int i = 0; i < L; ++i) r.v[i] = a.v[i] + k.v[i] - b.v[i];
    return r;
  }
  // normalized x < 2^264  ->  normalized, same residue, < 2 r.  q = floor(top / ((r >> 232) + 1)) never exceeds
  // floor(x / r) and falls short of it by at most one.
  static FF_HD fe29 weak_reduce(const fe29& x) {
    const uint32_t q = (uint32_t)(((uint64_t)x.v[L - 1] * RECIP) >> 48);
    fe29 r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
      const int64_t t = (int64_t)x.v[i] - (int64_t)((uint64_t)q * F::PL.v[i]) + c;
      r.v[i] = i < L - 1 ? (uint32_t)t & MASK : (uint32_t)t;
      c = t >> 29;
    }
    return r;
  }
  // one radix-4 group = the radix-2 DIF stages with half-distances h and h/2 on four elements (ntt_tile_stages).
  // Inputs normalized, < V r;  kA = kmult(V + 1, 1), kB = kmult(2 V + 1, 2);  p_nz: the group's offset p is not 0
  // (p == 0: stage A's first twiddle and stage B's twiddle are 1 and those products are skipped);  reduce: this is
  // the round trip that brings the unmultiplied outputs back below 2 r.
  static FF_HD void group4(fe29& x0, fe29& x1, fe29& x2, fe29& x3, bool p_nz, const fe29& tA2, const fe29& tA3,
                           const fe29& tB, const fe29& kA, const fe29& kB, bool reduce) {
    constexpr Limbs29 K3 = F::template kform<PROD_MULT, 1>();
    fe29 a0 = addl(x0, x2), a1 = addl(x1, x3);
    fe29 a2 = subl(x0, x2, kA), a3 = subl(x1, x3, kA);
    a3 = F::mul(a3, tA3);
    a2 = p_nz ? F::mul(a2, tA2) : F::norm(a2);
    fe29 b1 = subl(a0, a1, kB);
    fe29 b0 = F::norm(addl(a0, a1));
    b1 = p_nz ? F::mul(b1, tB) : F::norm(b1);
    fe29 b3 = subl(a2, a3, F::constant(K3));
    fe29 b2 = F::norm(addl(a2, a3));
    b3 = p_nz ? F::mul(b3, tB) : F::norm(b3);
    if (reduce) {
      b0 = weak_reduce(b0);
      if (!p_nz) {
        b1 = weak_reduce(b1);
        b2 = weak_reduce(b2);
        b3 = weak_reduce(b3);
      }
    }
    x0 = b0, x1 = b1, x2 = b2, x3 = b3;
  }
  // one radix-2 stage on two elements (odd rho): inputs normalized < V r, kA = kmult(V + 1, 1)
  static FF_HD void stage2(fe29& a, fe29& c, bool p_nz, const fe29& t, const fe29& kA) {
    fe29 sum = F::norm(addl(a, c));
    fe29 dif = subl(a, c, kA);
    dif = p_nz ? F::mul(dif, t) : F::norm(dif);
    a = sum, c = dif;
  }
  // value bound (in units of r) after a round trip / after the designated reducing round trip / after a radix-2 stage
  // (unmultiplied outputs: b0 < 4 V, b1 < 4 V + 1, b2 < 2 V + 7, b3 < 2 V + 9; multiplied ones: < 6, b2 < 12)
  static FF_HD uint32_t v_after_group(uint32_t V, bool reduce) {
    const uint32_t u = 4 * V + 1 > 2 * V + 9 ? 4 * V + 1 : 2 * V + 9;
    return reduce || u < 12 ? 12 : u;
  }
  static FF_HD uint32_t v_after_stage2(uint32_t V) { return 2 * V + 1 > 6 ? 2 * V + 1 : 6; }
  static constexpr int REDUCE_TRIP = 2;   // round trips are counted from 0

  // pass boundary: HBM holds canonical 8 x 32 values
  static FF_HD fe29 load(const u256& x) { return F::relimb(x); }
  static FF_HD u256 store_small(const fe29& v) { return F::relimb(F::template canon<1>(v)); }   // normalized, < 2 r
  static FF_HD u256 store_any(const fe29& v) { return store_small(weak_reduce(v)); }            // normalized, < 2^264
};

#if defined(__HIPCC__)
// ---- tables ------------------------------------------------------------------------------------------------------
// out[i] = in[i] * 2^5 : the 2^256 Montgomery form of ntt.cuh's tables -> the 2^261 form, canonical 9 x 29 limbs
static __global__ void __launch_bounds__(256) ntt29_convert_table(const u256* __restrict__ in, fe29* __restrict__ out,
                                                                  uint32_t count) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < count) out[i] = Fr29::from_std(in[i]);
}
// compact inner-stage twiddles: twc[dir * half + t] = w_R^(+-t), R = 2^rmax = min(2^10, n), t < R/2 (dir 1: inverse)
static __global__ void __launch_bounds__(256) ntt29_compact_twiddles(const u256* __restrict__ tw, fe29* __restrict__ twc,
                                                                     uint32_t log2n, uint32_t rmax) {
  const uint32_t half = (1u << rmax) >> 1, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * half) return;
  const uint32_t dir = i >= half ? 1u : 0u, t = i - dir * half;
  twc[i] = Fr29::from_std(ntt_tw(tw, t << (log2n - rmax), log2n, (int)dir));
}

// LDS tile, limb-major: element idx, limb k at lds[k * tile + idx]
__device__ __forceinline__ fe29 lds29_get(const uint32_t* lds, uint32_t tile, uint32_t idx) {
  fe29 r;
#pragma unroll
  for (int k = 0; k < 9; ++k) r.v[k] = lds[k * tile + idx];
  return r;
}
__device__ __forceinline__ void lds29_put(uint32_t* lds, uint32_t tile, uint32_t idx, const fe29& v) {
#pragma unroll
  for (int k = 0; k < 9; ++k) lds[k * tile + idx] = v.v[k];
}

// rho DIF stages on the tile (ntt_tile_stages of ntt.cuh, reduced-radix arithmetic).  twc: compact inner twiddles of
// this direction, twc[t << tshift] = w_R^t.  Returns the value bound V of the tile's elements (units of r).
template <int BLOCK>
__device__ __forceinline__ uint32_t ntt29_tile_stages(uint32_t* lds, const fe29* __restrict__ twc, uint32_t tshift,
                                                      uint32_t rho, uint32_t log2b, uint32_t tile) {
  using N = Ntt29;
  const uint32_t tid = threadIdx.x, B = 1u << log2b;
  uint32_t lh = rho, V = 1;
  if (lh & 1u) {
    --lh;
    const uint32_t h = 1u << lh;
    const fe29 kA = N::kmult(V + 1, 1);
    for (uint32_t bf = tid; bf < (tile >> 1); bf += BLOCK) {
      const uint32_t b = bf & (B - 1), pi = bf >> log2b;
      const uint32_t p = pi & (h - 1);
      const uint32_t i = ((pi >> lh) << (lh + 1)) | p;
      const uint32_t ia = (i << log2b) | b, ib = ((i + h) << log2b) | b;
      fe29 a = lds29_get(lds, tile, ia), c = lds29_get(lds, tile, ib);
      N::stage2(a, c, p != 0, twc[(p << (rho - lh - 1)) << tshift], kA);
      lds29_put(lds, tile, ia, a);
      lds29_put(lds, tile, ib, c);
    }
    V = N::v_after_stage2(V);
    __syncthreads();
  }
  int trip = 0;
  while (lh >= 2) {
    lh -= 2;
    const uint32_t hh = 1u << lh, h = hh << 1;
    const uint32_t sA = rho - lh - 2;
    const bool reduce = trip == N::REDUCE_TRIP;
    const fe29 kA = N::kmult(V + 1, 1), kB = N::kmult(2 * V + 1, 2);
    for (uint32_t g = tid; g < (tile >> 2); g += BLOCK) {
      const uint32_t b = g & (B - 1), pi = g >> log2b;
      const uint32_t p = pi & (hh - 1);
      const uint32_t i = ((pi >> lh) << (lh + 2)) | p;
      const uint32_t i0 = (i << log2b) | b, i1 = ((i + hh) << log2b) | b, i2 = ((i + h) << log2b) | b,
                     i3 = ((i + h + hh) << log2b) | b;
      fe29 x0 = lds29_get(lds, tile, i0), x1 = lds29_get(lds, tile, i1), x2 = lds29_get(lds, tile, i2),
           x3 = lds29_get(lds, tile, i3);
      const fe29 tA3 = twc[((p + hh) << sA) << tshift];
      fe29 tA2 = tA3, tB = tA3;
      if (p) {
        tA2 = twc[(p << sA) << tshift];
        tB = twc[(p << (sA + 1)) << tshift];
      }
      N::group4(x0, x1, x2, x3, p != 0, tA2, tA3, tB, kA, kB, reduce);
      lds29_put(lds, tile, i0, x0);
      lds29_put(lds, tile, i1, x1);
      lds29_put(lds, tile, i2, x2);
      lds29_put(lds, tile, i3, x3);
    }
    V = N::v_after_group(V, reduce);
    ++trip;
    __syncthreads();
  }
  return V;
}

// w^(+-e) in the 2^261 form: the table entry and whether it enters negated (ntt_tw of ntt.cuh)
__device__ __forceinline__ fe29 ntt29_tw(const fe29* __restrict__ tw29, uint32_t e, uint32_t log2n, int inverse,
                                         bool& negate) {
  const uint32_t n = 1u << log2n, half = n >> 1;
  if (inverse && e) e = n - e;
  negate = e >= half;
  return tw29[negate ? e - half : e];
}

// one pass (ntt_pass of ntt.cuh).  tw29: w^i (i < n/2), 1/n at n/2; twc: compact inner twiddles of this direction;
// scale29 (last pass): per-output factor replacing 1/n (coset shift), 2^261 form.
template <int BLOCK>
static __global__ void __launch_bounds__(BLOCK, 4) ntt29_pass(const u256* __restrict__ x, u256* __restrict__ y,
                                                        const fe29* __restrict__ tw29, const fe29* __restrict__ twc,
                                                        uint32_t tshift, uint32_t log2n, uint32_t log2s, uint32_t rho,
                                                        uint32_t log2b, int inverse, int last, size_t xstride,
                                                        size_t ystride, const fe29* __restrict__ scale29) {
  using N = Ntt29;
  using F = Fr29;
  extern __shared__ __align__(16) unsigned char smem[];
  uint32_t* lds = reinterpret_cast<uint32_t*>(smem);
  x += xstride * blockIdx.y;
  y += ystride * blockIdx.y;
  const uint32_t B = 1u << log2b;
  const uint32_t nR = 1u << (log2n - rho);
  const uint32_t base0 = ntt_tile_of_block(blockIdx.x, gridDim.x) << log2b;
  const uint32_t tile = (1u << rho) << log2b;
  const uint32_t tid = threadIdx.x;
  for (uint32_t e = tid; e < tile; e += BLOCK) {
    const uint32_t b = e & (B - 1), r = e >> log2b;
    lds29_put(lds, tile, e, N::load(x[(size_t)base0 + b + (size_t)nR * r]));
  }
  __syncthreads();
  ntt29_tile_stages<BLOCK>(lds, twc, tshift, rho, log2b, tile);
  const uint32_t s_mask = (1u << log2s) - 1;
  constexpr Limbs29 KP = F::template kform<Ntt29::PROD_MULT, 1>();
  for (uint32_t e = tid; e < tile; e += BLOCK) {
    const uint32_t b = e & (B - 1), q = e >> log2b;
    const uint32_t base = base0 + b;
    const uint32_t k = base & s_mask, j = base >> log2s;
    fe29 v = lds29_get(lds, tile, (bitrev(q, rho) << log2b) | b);
    const size_t oi = (size_t)k + ((size_t)(((size_t)j << rho) + q) << log2s);
    u256 out;
    if (!last) {
      const uint64_t ex = ((uint64_t)j * q) << log2s;
      const uint32_t em = (uint32_t)(ex & ((1ull << log2n) - 1));
      if (em) {
        bool neg;
        const fe29 t = ntt29_tw(tw29, em, log2n, inverse, neg);
        v = F::mul(v, t);
        if (neg) v = F::norm(N::subl(F::zero(), v, F::constant(KP)));
      }
      out = N::store_any(v);
    } else if (scale29) {
      out = N::store_any(F::mul(v, scale29[oi]));
    } else if (inverse) {
      out = N::store_any(F::mul(v, tw29[(1u << log2n) >> 1]));   // * 1/n
    } else {
      out = N::store_any(v);
    }
    y[oi] = out;
  }
}

// last forward pass of the quotient with A1*B1 - C1 fused (ntt_last_pass_abc of ntt.cuh)
template <int BLOCK, int TILE>
static __global__ void __launch_bounds__(BLOCK, 4) ntt29_last_pass_abc(const u256* __restrict__ x, u256* __restrict__ out,
                                                                 const fe29* __restrict__ twc, uint32_t tshift,
                                                                 uint32_t log2n, uint32_t log2s, uint32_t rho,
                                                                 uint32_t log2b, size_t xstride, int mul_invz) {
  using N = Ntt29;
  using F = Fr29;
  extern __shared__ __align__(16) unsigned char smem[];
  uint32_t* lds = reinterpret_cast<uint32_t*>(smem);
  constexpr int PER = TILE / BLOCK;
  const uint32_t B = 1u << log2b;
  const uint32_t nR = 1u << (log2n - rho);
  const uint32_t base0 = ntt_tile_of_block(blockIdx.x, gridDim.x) << log2b;
  const uint32_t tile = (1u << rho) << log2b;
  const uint32_t tid = threadIdx.x;
  // 2^266 mod r in the 9 x 29 limbs: (A 2^256)(B 2^256) / 2^261 = A B 2^251, times 2^266 / 2^261 -> A B 2^256;
  // JensGroth additionally * invZ1 = -1/2 (prover.nim:127-128, 141)
  fe29 acc[PER];
  uint32_t V = 1;
#pragma unroll 1
  for (int v = 0; v < 3; ++v) {
    const u256* xv = x + xstride * v;
    for (uint32_t e = tid; e < tile; e += BLOCK) {
      const uint32_t b = e & (B - 1), r = e >> log2b;
      lds29_put(lds, tile, e, N::load(xv[(size_t)base0 + b + (size_t)nR * r]));
    }
    __syncthreads();
    V = ntt29_tile_stages<BLOCK>(lds, twc, tshift, rho, log2b, tile);
    const fe29 kC = N::kmult(V + 1, 1);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const uint32_t e = tid + i * BLOCK;
      if (e < tile) {
        const uint32_t b = e & (B - 1), q = e >> log2b;
        const fe29 z = lds29_get(lds, tile, (bitrev(q, rho) << log2b) | b);
        if (v == 0) acc[i] = z;
        else if (v == 1) acc[i] = F::mul(F::mul(acc[i], z), F::constant(F::C_IN));
        else acc[i] = F::norm(N::subl(acc[i], z, kC));
      }
    }
    __syncthreads();
  }
  const uint32_t s_mask = (1u << log2s) - 1;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const uint32_t e = tid + i * BLOCK;
    if (e < tile) {
      const uint32_t b = e & (B - 1), q = e >> log2b;
      const uint32_t base = base0 + b;
      const uint32_t k = base & s_mask, j = base >> log2s;
      const size_t oi = (size_t)k + ((size_t)(((size_t)j << rho) + q) << log2s);
      u256 r = N::store_any(acc[i]);
      if (mul_invz) r = Fr::neg(Fr::div2(r));   // invZ1 = 1/(eta^n - 1) = -1/2
      out[oi] = r;
    }
  }
}
#endif  // __HIPCC__

}  // namespace g16
