// Radix-2 NTT over BN254 Fr for CDNA4.  Replaces forwardNTT / inverseNTT of the reference
// (groth16/math/ntt.nim:17-161): natural order in and out, forward unscaled
// y_k = sum_i x_i w^(ik), inverse scaled by 1/n, w = gen28^(2^(28-log2 n)) (math/domain.nim:26-33).
//
// The reference recursion is replaced by a Stockham auto-sort decomposition in <= 3 passes (2 up to n = 2^20); every
// pass performs up to 10 radix-2 DIF butterfly stages of a size-R sub-transform inside LDS:
//   pass with stride s (product of earlier radices), l = n / (s R):
//     for base = k + s*j  (k < s, j < l):   Z_q = sum_r x[base + (n/R) r] w_R^(r q)      (LDS butterflies)
//                                           y[k + s (R j + q)] = Z_q * w_n^(s j q)        (inter-pass twiddle)
// A workgroup owns B consecutive bases, so global reads/writes are B*32-byte contiguous segments.
// HBM traffic: 64 B per element per pass (algorithmic minimum for one pass: 64 B/element).
#pragma once
#include "ff.cuh"

namespace g16 {

// gen28 (math/domain.nim:26), standard form 0x2a3c09f0a58a7e85...9bd61b6e725b19f0, little-endian limbs
__device__ __forceinline__ u256 ntt_gen28() {
  u256 g;
  g.v[0] = 0x725b19f0u; g.v[1] = 0x9bd61b6eu; g.v[2] = 0x41112ed4u; g.v[3] = 0x402d111eu;
  g.v[4] = 0x8ef62abcu; g.v[5] = 0x00e0a7ebu; g.v[6] = 0xa58a7e85u; g.v[7] = 0x2a3c09f0u;
  return Fr::to_mont(g);
}
__device__ __forceinline__ u256 ntt_omega(uint32_t log2n) {
  u256 w = ntt_gen28();
  for (uint32_t i = log2n; i < 28; ++i) w = Fr::sqr(w);
  return w;
}
__device__ __forceinline__ u256 fr_pow_u32(u256 b, uint32_t e) {
  u256 r = Fr::one();
  while (e) {
    if (e & 1) r = Fr::mul(r, b);
    b = Fr::sqr(b);
    e >>= 1;
  }
  return r;
}

// tw[i] = w^i for i < n/2 ;  tw[n/2] = 1/n   (both Montgomery).  One table serves both directions:
// w^(-e) = w^(n-e) and w^(e) = -w^(e-n/2) for e >= n/2.
static __global__ void __launch_bounds__(256) ntt_make_twiddles(u256* __restrict__ tw, uint32_t log2n) {
  const uint32_t half = log2n ? (1u << (log2n - 1)) : 0u;
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < half) {
    tw[i] = fr_pow_u32(ntt_omega(log2n), i);
  } else if (i == half) {
    // 1/n = (1/2)^log2n : div2 applied to one  (ntt.nim folds the same halving into every level)
    u256 h = Fr::one();
    for (uint32_t k = 0; k < log2n; ++k) h = Fr::div2(h);
    tw[half] = h;
  }
}

// w^(+-e), e in [0, n)
__device__ __forceinline__ u256 ntt_tw(const u256* __restrict__ tw, uint32_t e, uint32_t log2n, int inverse) {
  const uint32_t n = 1u << log2n, half = n >> 1;
  if (inverse && e) e = n - e;
  if (e < half) return tw[e];
  return Fr::neg(tw[e - half]);
}

// Workgroup geometry.  A workgroup owns a tile of B bases x R points (R = 2^rho), B * R = TILE elements in LDS plus
// R/2 inner twiddles.  Three geometries are instantiated; G16_NTT_TILE (read once per process) selects one:
//   2048 (default)  512 threads, 64 KB tiles, rho <= 10: a 2^20 transform is TWO passes (10 + 10 stages; 2^21..2^30:
//                   three) and two workgroups fit a CU, so one's load phase overlaps the other's butterflies;
//                   B = 2 at rho = 10: 64-byte segments, neighbouring tiles on the same XCD (ntt_tile_of_block)
//   4096            1024 threads, 128 KB tiles, rho <= 10, B >= 4 (whole 128-byte lines), one workgroup per CU
//   1024            256 threads, 32 KB tiles, rho <= 8: three passes at 2^20 (round 1's geometry)
// Same box, 2^20 (tools/ab_ntt_tile.sh, profiles/r02_ab_ntt_tile.txt): quotient 0.851 / 0.818 / 0.811 ms and
// 107.6 / 108.7 / 110.4 proofs/s for 4096 / 2048 / 1024.  The passes are VALU-bound, so the extra pass of the small
// tiles costs HBM bytes (1.14 GB per quotient instead of 0.74 GB), not time, and their small workgroups slot in more
// easily beside the register-filling accumulate waves of the other streams (+1.5 % proofs/s); the default keeps the
// two-pass traffic and has the lowest single-proof latency.
constexpr int NTT_BLOCK = 1024;
constexpr int NTT_TILE = 4096;
constexpr uint32_t NTT_MAX_RHO = 10;
constexpr int NTT_BLOCK_SMALL = 256;
constexpr int NTT_TILE_SMALL = 1024;
constexpr uint32_t NTT_MAX_RHO_SMALL = 8;
constexpr int NTT_BLOCK_MID = 512;
constexpr int NTT_TILE_MID = 2048;

__device__ __forceinline__ uint32_t bitrev(uint32_t x, uint32_t bits) { return bits ? __brev(x) >> (32 - bits) : 0u; }

// Tile order: consecutive tiles touch neighbouring 128-byte lines of every strided row, and workgroups are dealt
// round-robin over the 8 XCDs (blocks b and b + 8 share an L2).  Giving XCD x the contiguous tile range
// [x*T/8, (x+1)*T/8) keeps those neighbours in ONE L2 instead of fetching the line into two.  Speed only: any
// bijection is correct.
__device__ __forceinline__ uint32_t ntt_tile_of_block(uint32_t bid, uint32_t ntiles) {
  return (ntiles & 7u) ? bid : (bid & 7u) * (ntiles >> 3) + (bid >> 3);
}

// rho DIF stages on the tile in LDS (rows r = 0..R-1 of B elements; output row bitrev(q) holds Z_q).
// Two stages per LDS round trip: a thread takes the four elements {i, i+h/2, i+h, i+3h/2} of a radix-4 group into
// registers, runs the stage with half-distance h on the pairs (0,2), (1,3) and the stage with half-distance h/2 on
// (0,1), (2,3), and writes them back -- the same radix-2 butterflies in the same order as the reference recursion
// (ntt.nim:47-50: t = w^j * odd; (even + t, even - t), here in DIF form), so the results are bit-identical, with half
// the LDS traffic, half the barriers and half the index arithmetic.  An odd rho starts with one plain radix-2 stage.
// (A prime field has no cheap 4th root of unity: the multiplication count is unchanged, 4 per group.)
template <int BLOCK>
__device__ __forceinline__ void ntt_tile_stages(u256* lds, const u256* twl, uint32_t rho, uint32_t log2b,
                                                uint32_t tile) {
  const uint32_t tid = threadIdx.x, B = 1u << log2b;
  uint32_t lh = rho;   // stages still to run; the next one has half-distance 2^(lh-1)
  if (lh & 1u) {
    --lh;
    const uint32_t h = 1u << lh;
    for (uint32_t bf = tid; bf < (tile >> 1); bf += BLOCK) {
      uint32_t b = bf & (B - 1), pi = bf >> log2b;
      uint32_t p = pi & (h - 1);
      uint32_t i = ((pi >> lh) << (lh + 1)) | p;
      uint32_t ia = (i << log2b) | b, ib = ((i + h) << log2b) | b;
      u256 a = lds[ia], c = lds[ib];
      u256 sum = Fr::add(a, c), dif = Fr::sub(a, c);
      if (p) dif = Fr::mul(dif, twl[p << (rho - lh - 1)]);  // p == 0: twiddle 1
      lds[ia] = sum;
      lds[ib] = dif;
    }
    __syncthreads();
  }
  while (lh >= 2) {
    lh -= 2;                              // stage A: half-distance h = 2^(lh+1); stage B: hh = 2^lh
    const uint32_t hh = 1u << lh, h = hh << 1;
    const uint32_t sA = rho - lh - 2;     // twiddle stride of stage A: w_(2h)^p = w_R^(p << sA); stage B: << (sA + 1)
    for (uint32_t g = tid; g < (tile >> 2); g += BLOCK) {
      const uint32_t b = g & (B - 1), pi = g >> log2b;
      const uint32_t p = pi & (hh - 1);
      const uint32_t i = ((pi >> lh) << (lh + 2)) | p;
      const uint32_t i0 = (i << log2b) | b, i1 = ((i + hh) << log2b) | b, i2 = ((i + h) << log2b) | b,
                     i3 = ((i + h + hh) << log2b) | b;
      const u256 x0 = lds[i0], x1 = lds[i1], x2 = lds[i2], x3 = lds[i3];
      // stage A (distance h): offsets p and p + hh inside the 2h-block
      u256 a0 = Fr::add(x0, x2), a2 = Fr::sub(x0, x2);
      u256 a1 = Fr::add(x1, x3), a3 = Fr::sub(x1, x3);
      if (p) a2 = Fr::mul(a2, twl[p << sA]);
      a3 = Fr::mul(a3, twl[(p + hh) << sA]);
      // stage B (distance hh): offset p inside either h-block
      u256 b0 = Fr::add(a0, a1), b1 = Fr::sub(a0, a1);
      u256 b2 = Fr::add(a2, a3), b3 = Fr::sub(a2, a3);
      if (p) {
        const u256 tb = twl[p << (sA + 1)];
        b1 = Fr::mul(b1, tb);
        b3 = Fr::mul(b3, tb);
      }
      lds[i0] = b0;
      lds[i1] = b1;
      lds[i2] = b2;
      lds[i3] = b3;
    }
    __syncthreads();
  }
}

// one pass: rho radix-2 stages of B sub-transforms of size R = 2^rho per workgroup.
// blockIdx.y selects one of several independent vectors (x + y*xstride -> y + y*ystride): the prover
// transforms Az, Bz, Cz together (prover.nim:167-169 runs them as three tasks).
// `scale` (last pass only): out[i] *= scale[i] instead of the plain 1/n of the inverse transform -- used
// to fuse multiplyByPowers (prover.nim:96-106) into the inverse NTT of shiftEvalDomain (prover.nim:109-113).
// `mul_src` (first pass of the quotient only): vector 2 of the batch is not read but FORMED while loading,
// Cz[i] = Az[i] * Bz[i] from vectors 0 and 1 (prover.nim:69-72: buildABC's pointwise product) -- Cz is consumed by this
// transform alone, so it never exists in HBM (no extra kernel between buildABC and the quotient, 64 n bytes less).
template <int BLOCK>
static __global__ void __launch_bounds__(BLOCK) ntt_pass(const u256* __restrict__ x, u256* __restrict__ y,
                                                      const u256* __restrict__ tw, uint32_t log2n, uint32_t log2s,
                                                      uint32_t rho, uint32_t log2b, int inverse, int last,
                                                      size_t xstride, size_t ystride,
                                                      const u256* __restrict__ scale, int mul_src) {
  extern __shared__ __align__(16) unsigned char smem[];
  u256* lds = reinterpret_cast<u256*>(smem);
  x += xstride * blockIdx.y;
  y += ystride * blockIdx.y;
  const uint32_t R = 1u << rho, B = 1u << log2b;
  const uint32_t nR = 1u << (log2n - rho);  // n / R = number of bases = input stride between r's
  const uint32_t base0 = ntt_tile_of_block(blockIdx.x, gridDim.x) << log2b;
  const uint32_t tile = R << log2b;
  const uint32_t tid = threadIdx.x;

  // inner-stage twiddles w_R^t (t < R/2) once per workgroup into LDS, behind the tile
  u256* twl = lds + tile;
  for (uint32_t t = tid; t < (R >> 1); t += BLOCK) twl[t] = ntt_tw(tw, t << (log2n - rho), log2n, inverse);
  // load: element (r, b) <- x[base0 + b + nR * r]   (LDS index r*B + b)
  if (mul_src && blockIdx.y == 2) {   // block-uniform
    const u256* xa = x - 2 * xstride;
    const u256* xb = x - xstride;
    for (uint32_t e = tid; e < tile; e += BLOCK) {
      uint32_t b = e & (B - 1), r = e >> log2b;
      const size_t gi = (size_t)base0 + b + (size_t)nR * r;
      lds[e] = Fr::mul(xa[gi], xb[gi]);
    }
  } else {
    for (uint32_t e = tid; e < tile; e += BLOCK) {
      uint32_t b = e & (B - 1), r = e >> log2b;
      lds[e] = x[(size_t)base0 + b + (size_t)nR * r];
    }
  }
  __syncthreads();

  // rho DIF stages: half distance h = R/2 ... 1 ; twiddle w_(2h)^p = w_R^(p * R/(2h))
  ntt_tile_stages<BLOCK>(lds, twl, rho, log2b, tile);

  // store: Z_q (at LDS row bitrev(q)) * w^(s j q)  ->  y[k + s (R j + q)]
  const uint32_t s_mask = (1u << log2s) - 1;
  for (uint32_t e = tid; e < tile; e += BLOCK) {
    uint32_t b = e & (B - 1), q = e >> log2b;
    uint32_t base = base0 + b;
    uint32_t k = base & s_mask, j = base >> log2s;
    u256 v = lds[(bitrev(q, rho) << log2b) | b];
    if (!last) {
      // exponent s*j*q < n ; computed mod n in 64 bits
      uint64_t ex = ((uint64_t)j * q) << log2s;
      uint32_t em = (uint32_t)(ex & ((1ull << log2n) - 1));
      if (em) v = Fr::mul(v, ntt_tw(tw, em, log2n, inverse));
    }
    const size_t oi = (size_t)k + ((size_t)(((size_t)j << rho) + q) << log2s);
    if (last) {
      if (scale) v = Fr::mul(v, scale[oi]);
      else if (inverse) v = Fr::mul(v, tw[(1u << log2n) >> 1]);  // * 1/n
    }
    y[oi] = v;
  }
}

// LAST forward pass of the quotient with the pointwise step fused in: the workgroup runs the same tile of the three
// vectors A, B, C (x, x + xstride, x + 2 xstride) one after the other through LDS and keeps, per output element,
//   acc = A1 ;  acc *= B1 ;  acc -= C1          ( ys[j] = A1[j]*B1[j] - C1[j], prover.nim:175-176 )
// in registers ( * invZ1 = -1/2 for the JensGroth flavour, prover.nim:127-128,141 ), so the three transformed
// vectors are never written to HBM: out receives n elements instead of 3n written + 3n read + n written.
template <int BLOCK, int TILE>
static __global__ void __launch_bounds__(BLOCK) ntt_last_pass_abc(const u256* __restrict__ x, u256* __restrict__ out,
                                                               const u256* __restrict__ tw, uint32_t log2n,
                                                               uint32_t log2s, uint32_t rho, uint32_t log2b,
                                                               size_t xstride, int mul_invz) {
  extern __shared__ __align__(16) unsigned char smem[];
  u256* lds = reinterpret_cast<u256*>(smem);
  constexpr int PER = TILE / BLOCK;   // output elements per thread (the tile may be smaller: guarded)
  const uint32_t R = 1u << rho, B = 1u << log2b;
  const uint32_t nR = 1u << (log2n - rho);
  const uint32_t base0 = ntt_tile_of_block(blockIdx.x, gridDim.x) << log2b;
  const uint32_t tile = R << log2b;
  const uint32_t tid = threadIdx.x;
  u256* twl = lds + tile;
  for (uint32_t t = tid; t < (R >> 1); t += BLOCK) twl[t] = ntt_tw(tw, t << (log2n - rho), log2n, 0);
  u256 acc[PER];
#pragma unroll 1
  for (int v = 0; v < 3; ++v) {
    const u256* xv = x + xstride * v;
    for (uint32_t e = tid; e < tile; e += BLOCK) {
      uint32_t b = e & (B - 1), r = e >> log2b;
      lds[e] = xv[(size_t)base0 + b + (size_t)nR * r];
    }
    __syncthreads();
    ntt_tile_stages<BLOCK>(lds, twl, rho, log2b, tile);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const uint32_t e = tid + i * BLOCK;
      if (e < tile) {
        uint32_t b = e & (B - 1), q = e >> log2b;
        u256 z = lds[(bitrev(q, rho) << log2b) | b];
        acc[i] = v == 0 ? z : v == 1 ? Fr::mul(acc[i], z) : Fr::sub(acc[i], z);
      }
    }
    __syncthreads();   // the tile is overwritten by the next vector
  }
  const uint32_t s_mask = (1u << log2s) - 1;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const uint32_t e = tid + i * BLOCK;
    if (e < tile) {
      uint32_t b = e & (B - 1), q = e >> log2b;
      uint32_t base = base0 + b;
      uint32_t k = base & s_mask, j = base >> log2s;
      const size_t oi = (size_t)k + ((size_t)(((size_t)j << rho) + q) << log2s);
      u256 r = acc[i];
      if (mul_invz) r = Fr::neg(Fr::div2(r));   // invZ1 = 1/(eta^n - 1) = -1/2
      out[oi] = r;
    }
  }
}

// scale tables for the coset shift, eta = w_(2n) (prover.nim:163):
//   mode 0: tab[i] = eta^i / n      (inverse NTT + multiplyByPowers(eta),    prover.nim:110-112)
//   mode 1: tab[i] = eta^(-i) / n   (inverse NTT + multiplyByPowers(1/eta),  prover.nim:142-143)
static __global__ void __launch_bounds__(256) ntt_make_coset_table(u256* __restrict__ tab, uint32_t log2n, int mode) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const uint32_t n = 1u << log2n;
  if (i >= n) return;
  u256 ninv = Fr::one();
  for (uint32_t k = 0; k < log2n; ++k) ninv = Fr::div2(ninv);
  u256 eta = ntt_omega(log2n + 1);
  uint32_t e = mode ? (2 * n - i) % (2 * n) : i;   // eta^(2n) = 1
  tab[i] = Fr::mul(fr_pow_u32(eta, e), ninv);
}

// ys[j] = A1[j]*B1[j] - C1[j]   (prover.nim:175-176) on separately held vectors: the multi-GPU task-parallel quotient
// (one coset pipeline per rank) forms its slice of the H scalars from three received slices.  The single-GPU path
// has this step fused into ntt_last_pass_abc.
static __global__ void __launch_bounds__(256) fr_abc_pointwise(const u256* __restrict__ a, const u256* __restrict__ b,
                                                        const u256* __restrict__ c, u256* __restrict__ out,
                                                        uint32_t n) {
  uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  out[j] = Fr::sub(Fr::mul(a[j], b[j]), c[j]);
}

}  // namespace g16
