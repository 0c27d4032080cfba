// Pippenger multi-scalar multiplication kernels for CDNA4 (G1 and G2 share the templates).
// Replaces the hot loop behind msmConstantineG1/G2 (reference groth16/bn128/msm.nim:35-83, i.e.
// constantine's multiScalarMul_vartime) -- same inputs (Fr scalars, affine points, (0,0) = inf),
// same canonical affine output.
//
// Pipeline (one stream, no host round trip inside):
//   sort   part_pass<false/true>, scan1_*, bucket_hist, bucket_place, perm_scatter, msm_make_extra
//          scalar -> signed c-bit digits; two-level LDS partition sort of the (bucket, point index | sign) entries; bucket
//          offsets, split-bucket segment list, buckets ordered by size.  (msm_count / msm_scatter / scan_* / perm_hist:
//          the global-atomic variant, used when the partition count exceeds the LDS histogram and by G16_MSM_SORT=a)
//   accum  msm_accum      one thread per bucket *segment* (<= L entries): XYZZ += affine table point, 9x29 field
//   heavy  msm_heavy      buckets that were split in >1 segment: one thread (few segments) / LDS tree (many)
//   reduce msm_reduce1/2  sum_k k*B_k: 16-bucket chunk running sums, then per slice an LDS suffix scan + tree
//   fold   msm_fold_classes / msm_fold_merged (registered sets: slices of the class / plain bucket set) or msm_fold
//          (windows, Horner) -> one point; canonical affine on request
// Every stage from accum on takes up to MSM_BATCH_MAX jobs per launch (blockIdx.y).
//
// Load balance: work is cut by *entries*, not by buckets -- a circom witness puts ~30 % of all
// scalars in bucket (w=0, d=1); that bucket becomes ~N/L segments handled by N/L threads.
#pragma once
#include "ec29.cuh"
#include "msm_params.hpp"

namespace g16 {

constexpr int MSM_BLOCK = 256;
#ifndef G16_ACC_BLOCK
#define G16_ACC_BLOCK 256
#endif
constexpr int ACC_BLOCK = G16_ACC_BLOCK;   // workgroup size of the accumulate kernels (no LDS, no barriers: dispatch granularity only)
#ifndef G16_G2_WAVES
#define G16_G2_WAVES 2
#endif
constexpr int FR_BITS = 254;

// Wave priority of the latency-bound kernels (split-bucket combine, reduce, fold; the sort): their few waves share the
// SIMDs with the register-filling accumulate waves of the other streams, and every instruction they wait for the
// arbiter lengthens a dependency chain the proof's end waits for.  s_setprio 3 = issue before priority-0 waves.
#ifndef G16_TAIL_PRIO
#define G16_TAIL_PRIO 0
#endif
__device__ __forceinline__ void tail_prio() {
#if G16_TAIL_PRIO
  __builtin_amdgcn_s_setprio(G16_TAIL_PRIO);
#endif
}
#ifndef G16_G1_WAVES
#define G16_G1_WAVES 4
#endif
// Register budget of the latency-bound kernels, as waves per SIMD they must allow (1 = whatever the compiler takes).
// A wave can only start on a SIMD with that many free registers: while the accumulate kernels of other streams keep
// refilling the SIMDs with 117-register (G1) / 241-register (G2) waves, a 481-register reduce1<G2> wave finds room
// only when a whole SIMD drains -- the tails of a proof then wait for the accumulate launches to END instead of
// slotting in behind single exiting waves.
#ifndef G16_TAIL_WAVES_G1
#define G16_TAIL_WAVES_G1 1
#endif
#ifndef G16_TAIL_WAVES_G2
#define G16_TAIL_WAVES_G2 1
#endif
template <class C>
constexpr int tail_waves() { return sizeof(typename C::Aff) == 64 ? G16_TAIL_WAVES_G1 : G16_TAIL_WAVES_G2; }

// ---- batched launches -------------------------------------------------------------------------------------------
// Every stage kernel from the accumulation to the fold takes up to MSM_BATCH_MAX independent MSMs per launch
// (blockIdx.y = job).  The three G1 MSMs of a proof that consume the witness (A1, B1, C1: prover.nim:282-302) run as
// ONE launch sequence on ONE stream: their latency-bound reduce / fold chains run three wide, a proof needs ~15
// launches and 2 streams less, and the in-flight proofs of a GPU fit the hardware queues without sharing them.
constexpr int MSM_BATCH_MAX = 3;
template <class C>
struct MsmJob {
  const typename Ec29<C>::Tab* points;   // window tables (registered set) or reduced-radix entries (one-shot)
  const uint32_t* entries;               // the bucket arrangement it runs against (g16_ctx::MsmSort)
  const uint32_t* offset;
  const uint2* xseg;
  const uint32_t* info;
  const uint32_t* perm;
  const uint32_t* heavy;
  const uint32_t* xoff;
  typename Ec29<C>::Acc* partial;        // bucket / segment sums (reduced radix)
  const typename Ec29<C>::Acc* init;     // optional: bucket sums of another MSM over the same bucket set to start from
  typename C::Acc* chunkR;               // reduce1 -> reduce2
  typename C::Acc* chunkA;
  typename C::Acc* wsum;                 // reduce2 -> fold (2 * 64 + 2 accumulators)
  typename C::Aff* out_aff;              // either may be null
  typename C::Acc* out_acc;
};
template <class C>
struct MsmBatch {
  MsmJob<C> job[MSM_BATCH_MAX];
};


// signed c-bit digits of a scalar, least significant window first.  The canonical scalar stays in its 8 registers
// and is shifted right by c bits per window (8 funnel shifts with static register indices, c <= 22 < 32): no LDS
// staging, no dynamic limb indexing, and any workgroup size.  (Rounds 1-2 staged the limbs in LDS and indexed them by
// window: 36 KB of LDS per 1024 threads and two LDS reads per digit.)
// `live` (optional): bitmap over the pairs; a cleared bit drops the pair before its scalar is even read -- the entry
// lists of point sets with many points at infinity ((0,0) for every wire absent from a matrix, curves.nim:95-107)
// then hold only the points that can contribute.
// ---- class bucket set of a registered point set with multiplier tables {1, 2} (MsmParams::mtab == 2) ---------------
// A signed digit of magnitude t in [1, H], H = 2^(c-1), is served from the table of 2^(c w) P or of 2 * 2^(c w) P:
//   t = 2^s * b,  s = ctz(t) & 1,  so that b = 4^z * u with u odd (an even power of two times an odd number),
// and the bucket is b's, with the entry pointing into table s.  Only the b of that form are buckets: 2/3 of [1, H] --
// the running-sum reduction (2 general additions per bucket, 8.6 % of a proof's instructions in round 3) shrinks by a
// third for the same 13 accumulated entries per scalar.  Bucket order = (class z, v) with u = 2 v + 1: inside a class
// the weights 4^z (2 v + 1) are an arithmetic progression, which is all the running sums need (msm_fold_classes).
// Classes z = 0, 1, 2 hold t with ctz(t) < 6; the rest (t = 64 x) go unmultiplied to class X, weight 64 x.  Sizes
// H/2, H/8, H/32, H/64: 43 slices of 2^(c-7) buckets.  Every bucket receives one or two digit values (u > H/2 or an
// odd-length chain end: one), so bucket loads are 1x or 2x the load of the plain set -- like its two rounds of tasks.
// (msm_class_bucket itself lives in msm_params.hpp: host and device code, and a CPU test, share it)

template <class EMIT>
__device__ __forceinline__ void msm_digits(const u256* __restrict__ scalars, const uint32_t* __restrict__ live, uint32_t i,
                                           const MsmParams& P, EMIT&& emit) {
  if (live && !((live[i >> 5] >> (i & 31)) & 1u)) return;
  u256 s = scalars[i];
  if (P.scalars_mont) s = Fr::from_mont(s);
  if (Fr::is_zero(s)) return;
  const uint32_t c = P.c, half = 1u << (c - 1), mask = (1u << c) - 1;
  uint32_t carry = 0;
  for (uint32_t w = 0; w < P.nwin; ++w) {
    uint32_t raw = (s.v[0] & mask) + carry;
    uint32_t neg = raw > half ? 1u : 0u;
    uint32_t mag = neg ? (1u << c) - raw : raw;
    carry = neg;
    if (mag) {
      if (P.mtab == 2) {   // (window + table selector, class bucket): the entry then indexes table [s][w]
        uint32_t sel;
        const uint32_t b = msm_class_bucket(mag, c, sel);
        emit(w + sel * P.nwin, b, neg);
      } else {
        emit(w, mag - 1, neg);
      }
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) s.v[j] = __builtin_amdgcn_alignbit(s.v[j + 1], s.v[j], c);
    s.v[7] >>= c;
  }
}

static __global__ void __launch_bounds__(MSM_BLOCK) msm_count(const u256* __restrict__ scalars,
                                                              const uint32_t* __restrict__ live, MsmParams P,
                                                              uint32_t* __restrict__ count) {
  uint32_t i = blockIdx.x * MSM_BLOCK + threadIdx.x;
  if (i >= P.n) return;
  const uint32_t bshift = P.c - 1;
  msm_digits(scalars, live, i, P, [&](uint32_t w, uint32_t k, uint32_t) {
    atomicAdd(&count[(P.tables ? 0u : (w << bshift)) + k], 1u);
  });
}

static __global__ void __launch_bounds__(MSM_BLOCK) msm_scatter(const u256* __restrict__ scalars,
                                                         const uint32_t* __restrict__ live, MsmParams P,
                                                         const uint32_t* __restrict__ offset,
                                                         uint32_t* __restrict__ cursor,
                                                         uint32_t* __restrict__ entries) {
  uint32_t i = blockIdx.x * MSM_BLOCK + threadIdx.x;
  if (i >= P.n) return;
  const uint32_t bshift = P.c - 1;
  msm_digits(scalars, live, i, P, [&](uint32_t w, uint32_t k, uint32_t neg) {
    uint32_t b = (P.tables ? 0u : (w << bshift)) + k;
    uint32_t pos = offset[b] + atomicAdd(&cursor[b], 1u);
    // entry = point index (table-major when tables are used) | sign in bit 31
    uint32_t pidx = P.tables ? (w * P.n + i) : i;
    entries[pos] = pidx | (neg << 31);
  });
}

// ---- two-level partition sort (no global atomics) -------------------------------------------------------
// Global atomics execute at the memory side on MI355X (~26 G/s measured for scattered 4-byte adds), which
// made the histogram + scatter above cost 1.6 ms per 2^20 scalars.  Here the (window, bucket) key is split
// into a partition id (window, high bucket bits; <= 4096 partitions) and <= 8 low bits:
//   part_pass<false>  per tile of 4096 scalars: LDS histogram over partitions -> tile_hist[part][tile]
//   (exclusive scan of tile_hist, partition-major = final bucket order)
//   part_pass<true>   recompute digits; rank inside (tile, partition) from LDS atomics; tmp entry =
//                     { low bits | sign<<8 , point index (table-major for registered sets) }
//   bucket_hist/place 8 workgroups per partition: LDS histograms over the low bits -> count[], offset[],
//                     then placement of the final entries in bucket order
// LDS counter increment that returns the old value, robust against one hot key per wave: skewed scalars
// (a circom witness is ~30 % ones; a short top window maps every scalar to a handful of buckets) would
// otherwise serialise 64 same-address LDS atomics per wave-instruction.  If more than 8 lanes of the wave
// carry the first active lane's key, that group issues ONE atomic and ranks itself by prefix popcount.
__device__ __forceinline__ uint32_t lds_rank_add(uint32_t* ctr, uint32_t key) {
  const uint32_t leader = __builtin_amdgcn_readfirstlane(key);
  const unsigned long long m = __ballot(key == leader);
  if (__popcll(m) > 8) {
    uint32_t r = 0;
    if (key == leader) {
      const uint32_t lane = threadIdx.x & 63;
      const uint32_t before = __popcll(m & ((1ull << lane) - 1));
      uint32_t base = 0;
      if (before == 0) base = atomicAdd(&ctr[leader], (uint32_t)__popcll(m));
      base = __shfl(base, __ffsll((long long)m) - 1, 64);
      r = base + before;
    } else {
      r = atomicAdd(&ctr[key], 1u);
    }
    return r;
  }
  return atomicAdd(&ctr[key], 1u);
}

// 1024 threads x 4 scalars: the pass is a chain of dependent LDS atomics and scattered 8-byte stores per thread, and a
// 2^20-scalar sort has only 256 tiles -- one workgroup per CU -- so the workgroup is as wide as it gets (16 waves per
// CU hide that latency; rounds 1-2 ran 256 threads x 16 scalars = ONE wave per SIMD)
// Low bucket bits sorted inside a partition (bucket_hist / bucket_place: one thread per low value).  Round 4: 9 instead
// of 8 -- half as many partitions (688 for the class set of c = 20), hence half as many (tile, partition) runs that
// part_pass<true> keeps open at once: 32 tiles per XCD x 1376 runs x one active 128-byte line were 5.6 MB against a
// 4-MB L2, and lines left the L2 half written (WRITE_SIZE 3.4 x the record bytes); 688 runs are 2.8 MB.
constexpr int BS_LOG = 9;
constexpr int BS_LOW = 1 << BS_LOG;
constexpr int PART_BLOCK = 1024;
constexpr int PART_PER_THREAD = 4;
constexpr int PART_TILE = PART_BLOCK * PART_PER_THREAD;  // scalars per workgroup
constexpr int PART_MAX = 8192;                           // max partitions (LDS histogram, 32 KB)

// tile of a partition pass: XCD x takes a contiguous range of tiles, so that neighbouring tiles -- whose (tile,
// partition) runs of tmp records share cache lines at their ends -- mostly write through the same L2
__device__ __forceinline__ uint32_t part_tile_of_block(uint32_t bid, uint32_t ntiles) {
  return (ntiles & 7u) ? bid : (bid & 7u) * (ntiles >> 3) + (bid >> 3);
}
template <bool SCATTER>
static __global__ void __launch_bounds__(PART_BLOCK) part_pass(const u256* __restrict__ scalars,
                                                               const uint32_t* __restrict__ live, MsmParams P,
                                                               uint32_t lo_bits, uint32_t nparts, uint32_t ntiles,
                                                               uint32_t* __restrict__ tile_hist,
                                                               uint2* __restrict__ tmp) {
  __shared__ uint32_t hist[PART_MAX];
  const uint32_t tile = part_tile_of_block(blockIdx.x, ntiles);
  for (uint32_t p = threadIdx.x; p < nparts; p += PART_BLOCK)
    hist[p] = SCATTER ? tile_hist[(size_t)p * ntiles + tile] : 0u;   // SCATTER: exclusive base of (part, tile)
  __syncthreads();
  const uint32_t hi_bits = P.c - 1 - lo_bits, lo_mask = (1u << lo_bits) - 1;
  for (uint32_t r = 0; r < PART_PER_THREAD; ++r) {
    uint32_t i = tile * PART_TILE + r * PART_BLOCK + threadIdx.x;
    if (i < P.n) {
      msm_digits(scalars, live, i, P, [&](uint32_t w, uint32_t k, uint32_t neg) {
        uint32_t part = (P.tables ? 0u : (w << hi_bits)) | (k >> lo_bits);
        uint32_t pos = lds_rank_add(hist, part);
        if (SCATTER) tmp[pos] = make_uint2((k & lo_mask) | (neg << BS_LOG), P.tables ? w * P.n + i : i);
      });
    }
  }
  if (!SCATTER) {
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < nparts; p += PART_BLOCK) tile_hist[(size_t)p * ntiles + tile] = hist[p];
  }
}

// A partition can be arbitrarily large (skewed scalars), so each one is cut into BS_SPLIT slices handled by
// separate workgroups: bucket_hist counts the low bits per slice, bucket_place derives count[]/offset[] from
// the slice histograms and writes the final entries.
// Which (partition, slice) a workgroup of bucket_hist / bucket_place takes.  Workgroups are dealt round robin over the 8
// XCDs (block b -> XCD b % 8), each with its own L2: with the plain order part = b / 8 the eight slices of a partition
// -- which scatter 4-byte entries into the SAME 40-KB output range -- sat on eight different XCDs, and every 128-byte
// line left eight L2s as a partial write (WRITE_SIZE 5.3 x the entry bytes, profiles/r03_pmc_hbm_traffic_2p20.json).
// This order gives all slices of a partition the same b % 8, so that one L2 merges the line.  Speed / traffic only:
// any bijection is correct.
// (bs_block itself lives in msm_params.hpp: a CPU test checks that it is a bijection)
__device__ __forceinline__ void bs_slice(uint32_t start, uint32_t end, uint32_t q, uint32_t& lo, uint32_t& hi) {
  const uint32_t len = end - start;
  lo = start + (uint32_t)(((uint64_t)len * q) / BS_SPLIT);
  hi = start + (uint32_t)(((uint64_t)len * (q + 1)) / BS_SPLIT);
}
static __global__ void __launch_bounds__(BS_LOW) bucket_hist(const uint2* __restrict__ tmp,
                                                          const uint32_t* __restrict__ part_base, uint32_t ntiles,
                                                          uint32_t nparts, const uint32_t* __restrict__ total,
                                                          uint32_t* __restrict__ slice_hist) {
  __shared__ uint32_t hist[BS_LOW];
  uint32_t part, q;
  bs_block(blockIdx.x, nparts, part, q);
  const uint32_t tid = threadIdx.x;
  const uint32_t start = part_base[(size_t)part * ntiles];
  const uint32_t end = part + 1 < nparts ? part_base[(size_t)(part + 1) * ntiles] : total[0];
  uint32_t lo, hi;
  bs_slice(start, end, q, lo, hi);
  hist[tid] = 0;
  __syncthreads();
  for (uint32_t j = lo + tid; j < hi; j += BS_LOW) lds_rank_add(hist, tmp[j].x & (uint32_t)(BS_LOW - 1));
  __syncthreads();
  slice_hist[((size_t)part * BS_SPLIT + q) * BS_LOW + tid] = hist[tid];
}
// `fused` (lo_bits == BS_LOG: one partition = one PERM_BLOCK-bucket block of the size-order permutation): the q == 0 workgroup of
// every partition holds the final bucket counts in registers anyway, so it also does what rounds 1-2 ran five more
// launches for -- the extra-segment bookkeeping of split buckets (xoff[], heavy list; split buckets are rare, so
// their range of extra-segment slots comes from one global atomic each instead of a device-wide scan) and the size
// histogram of the permutation (perm_hist).
constexpr int PERM_BINS = 256;   // size classes of the bucket-order permutation (perm_hist / perm_scatter below)
// extra segments of a bucket of cnt entries cut into segments of L: max(ceil(cnt / L) - 1, 0)
__device__ __forceinline__ uint32_t extra_segs(uint32_t cnt, uint32_t L) { return cnt > L ? (cnt - 1) / L : 0u; }
static __global__ void __launch_bounds__(BS_LOW) bucket_place(const uint2* __restrict__ tmp,
                                                           const uint32_t* __restrict__ part_base, uint32_t ntiles,
                                                           uint32_t nparts, const uint32_t* __restrict__ total,
                                                           const uint32_t* __restrict__ slice_hist, MsmParams P,
                                                           uint32_t lo_bits, uint32_t* __restrict__ count,
                                                           uint32_t* __restrict__ offset,
                                                           uint32_t* __restrict__ entries, uint32_t fused,
                                                           uint32_t* __restrict__ xoff, uint32_t* __restrict__ heavy,
                                                           uint32_t* __restrict__ info, uint32_t* __restrict__ ghist,
                                                           uint32_t* __restrict__ blk_base) {
  __shared__ uint32_t cur[BS_LOW];
  __shared__ uint32_t szh[PERM_BINS];
  uint32_t part, q;
  bs_block(blockIdx.x, nparts, part, q);
  const uint32_t tid = threadIdx.x;
  const uint32_t start = part_base[(size_t)part * ntiles];
  const uint32_t end = part + 1 < nparts ? part_base[(size_t)(part + 1) * ntiles] : total[0];
  // bucket totals over all slices, and the part of each bucket that belongs to earlier slices
  uint32_t mine = 0, before = 0;
#pragma unroll
  for (int k = 0; k < BS_SPLIT; ++k) {
    uint32_t h = slice_hist[((size_t)part * BS_SPLIT + k) * BS_LOW + tid];
    mine += h;
    if ((uint32_t)k < q) before += h;
  }
  // exclusive scan of the bucket totals (Hillis-Steele in LDS)
  cur[tid] = mine;
  if (tid < PERM_BINS) szh[tid] = 0;
  __syncthreads();
  for (int d = 1; d < BS_LOW; d <<= 1) {
    uint32_t add = (int)tid >= d ? cur[tid - d] : 0u;
    __syncthreads();
    cur[tid] += add;
    __syncthreads();
  }
  const uint32_t excl = cur[tid] - mine;
  __syncthreads();
  cur[tid] = start + excl + before;
  if (q == 0) {
    // bucket index of (part, low bits): partitions are numbered in bucket order in both modes
    if (tid < (1u << lo_bits)) {
      uint32_t b = (part << lo_bits) + tid;
      count[b] = mine;
      offset[b] = start + excl;
      if (fused) {
        const uint32_t e = extra_segs(mine, P.seg);
        uint32_t x0 = 0;
        if (e) {
          x0 = atomicAdd(&info[1], e);
          heavy[atomicAdd(&info[2], 1u)] = b;
        }
        xoff[b] = x0;
        atomicAdd(&szh[mine < PERM_BINS - 1 ? mine : PERM_BINS - 1], 1u);
      }
    }
    if (part == nparts - 1 && tid == 0) offset[P.nbuckets] = end;
  }
  __syncthreads();
  if (q == 0 && fused && tid < PERM_BINS) {
    const uint32_t m = szh[tid];
    blk_base[(size_t)part * PERM_BINS + tid] = m ? atomicAdd(&ghist[tid], m) : 0u;
  }
  uint32_t lo, hi;
  bs_slice(start, end, q, lo, hi);
  for (uint32_t j = lo + tid; j < hi; j += BS_LOW) {
    uint2 e = tmp[j];
    uint32_t pos = lds_rank_add(cur, e.x & (uint32_t)(BS_LOW - 1));
    entries[pos] = e.y | (((e.x >> BS_LOG) & 1u) << 31);
  }
}

// plain exclusive scan of a u32 array in place (three launches, reuses the tile machinery below)
// ---- three-phase exclusive scan over the bucket histogram ------------------------------------------
// produces offset[b] = sum_{b'<b} count[b'], xoff[b] = sum_{b'<b} extra(b') with
// extra(b) = max(ceil(count/L) - 1, 0), and appends buckets with extra(b) > 0 to the heavy list.
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;                       // per thread
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;  // 2048 buckets per workgroup

// split buckets with at least this many extra segments are combined by a workgroup (msm_heavy), the others by
// one thread each (phase 1 of msm_heavy)
constexpr uint32_t HEAVY_MIN = 12;

__device__ __forceinline__ uint2 block_excl_scan2(uint2 v, uint2* total) {
  // exclusive scan of (x,y) across SCAN_BLOCK threads: wave shuffles + LDS across the 4 waves
  __shared__ uint2 wsum[SCAN_BLOCK / 64];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint2 inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t x = __shfl_up(inc.x, d, 64), y = __shfl_up(inc.y, d, 64);
    if (lane >= (uint32_t)d) { inc.x += x; inc.y += y; }
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint2 base = make_uint2(0, 0), tot = make_uint2(0, 0);
#pragma unroll
  for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
    uint2 s = wsum[w];
    if ((uint32_t)w < wave) { base.x += s.x; base.y += s.y; }
    tot.x += s.x; tot.y += s.y;
  }
  __syncthreads();
  if (total) *total = tot;
  return make_uint2(base.x + inc.x - v.x, base.y + inc.y - v.y);
}

static __global__ void __launch_bounds__(SCAN_BLOCK) scan_tile_sums(const uint32_t* __restrict__ count, uint32_t nb,
                                                             uint32_t L, uint2* __restrict__ tile_sum) {
  uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint2 s = make_uint2(0, 0);
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    uint32_t b = base + j;
    uint32_t cnt = b < nb ? count[b] : 0u;
    s.x += cnt;
    s.y += extra_segs(cnt, L);
  }
  uint2 tot;
  block_excl_scan2(s, &tot);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}
// single workgroup: exclusive scan of the tile sums in place; totals -> info[0..1]
static __global__ void __launch_bounds__(SCAN_BLOCK) scan_tiles(uint2* __restrict__ tile_sum, uint32_t ntiles,
                                                         uint32_t* __restrict__ info) {
  uint2 carry = make_uint2(0, 0);
  for (uint32_t base = 0; base < ntiles; base += SCAN_BLOCK) {
    uint32_t t = base + threadIdx.x;
    uint2 v = t < ntiles ? tile_sum[t] : make_uint2(0, 0);
    uint2 tot;
    uint2 ex = block_excl_scan2(v, &tot);
    if (t < ntiles) tile_sum[t] = make_uint2(ex.x + carry.x, ex.y + carry.y);
    carry.x += tot.x;
    carry.y += tot.y;
  }
  if (threadIdx.x == 0) {
    info[0] = carry.x;  // total entries
    info[1] = carry.y;  // total extra segments
  }
}
static __global__ void __launch_bounds__(SCAN_BLOCK) scan_apply(const uint32_t* __restrict__ count, uint32_t nb, uint32_t L,
                                                         const uint2* __restrict__ tile_sum,
                                                         uint32_t* __restrict__ offset, uint32_t* __restrict__ xoff,
                                                         uint32_t* __restrict__ heavy, uint32_t* __restrict__ info) {
  uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t cnt[SCAN_ITEMS];
  uint2 s = make_uint2(0, 0);
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    uint32_t b = base + j;
    cnt[j] = b < nb ? count[b] : 0u;
    s.x += cnt[j];
    s.y += extra_segs(cnt[j], L);
  }
  uint2 ex = block_excl_scan2(s, nullptr);
  uint2 t0 = tile_sum[blockIdx.x];
  uint32_t o = t0.x + ex.x, x = t0.y + ex.y;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    uint32_t b = base + j;
    if (b < nb) {
      offset[b] = o;
      xoff[b] = x;
      uint32_t e = extra_segs(cnt[j], L);
      if (e) heavy[atomicAdd(&info[2], 1u)] = b;   // every split bucket: msm_make_extra lists its segments
      o += cnt[j];
      x += e;
    }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_BLOCK - 1) offset[nb] = o;
}

// in-place exclusive scan of a plain u32 array (used for tile_hist): same three phases
static __global__ void __launch_bounds__(SCAN_BLOCK) scan1_tile_sums(const uint32_t* __restrict__ v, uint32_t n,
                                                                     uint2* __restrict__ tile_sum) {
  uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint2 s = make_uint2(0, 0);
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) s.x += base + j < n ? v[base + j] : 0u;
  uint2 tot;
  block_excl_scan2(s, &tot);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}
static __global__ void __launch_bounds__(SCAN_BLOCK) scan1_apply(uint32_t* __restrict__ v, uint32_t n,
                                                                 const uint2* __restrict__ tile_sum) {
  uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t x[SCAN_ITEMS];
  uint2 s = make_uint2(0, 0);
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    x[j] = base + j < n ? v[base + j] : 0u;
    s.x += x[j];
  }
  uint2 ex = block_excl_scan2(s, nullptr);
  uint32_t o = tile_sum[blockIdx.x].x + ex.x;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    if (base + j < n) v[base + j] = o;
    o += x[j];
  }
}

// extra-segment descriptors: xseg[xoff[b] + s - 1] = (b, s) for s = 1..extra(b)
static __global__ void __launch_bounds__(MSM_BLOCK) msm_make_extra(const uint32_t* __restrict__ heavy,
                                                            const uint32_t* __restrict__ info,
                                                            const uint32_t* __restrict__ offset,
                                                            const uint32_t* __restrict__ xoff, uint32_t L,
                                                            uint32_t max_extra, uint2* __restrict__ xseg) {
  // one workgroup per heavy bucket (grid-stride), threads stride over its segments
  const uint32_t nheavy = info[2];
  for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
    uint32_t b = heavy[h];
    uint32_t cnt = offset[b + 1] - offset[b];
    uint32_t e = extra_segs(cnt, L), x0 = xoff[b];
    for (uint32_t s = threadIdx.x; s < e; s += MSM_BLOCK)
      if (x0 + s < max_extra) xseg[x0 + s] = make_uint2(b, s + 1);
  }
}

// ---- bucket order by size ------------------------------------------------------------------------------
// Bucket sizes are ~Poisson(avg): a wave that takes 64 consecutive buckets runs as long as its largest one
// (~70 % lane efficiency at avg 32).  perm[] lists the buckets by descending min(count, 255), so the 64
// lanes of a wave get (nearly) equal trip counts.  Two small kernels: per-block LDS histogram + one global
// atomic per (block, size class) to reserve a range, then ranks from LDS atomics.
constexpr int PERM_BLOCK = BS_LOW;   // = one partition of the fused sort (bucket_place writes blk_base per partition)
static __global__ void __launch_bounds__(PERM_BLOCK) perm_hist(const uint32_t* __restrict__ count, uint32_t nb,
                                                               uint32_t* __restrict__ ghist,
                                                               uint32_t* __restrict__ blk_base) {
  __shared__ uint32_t h[PERM_BINS];
  if (threadIdx.x < PERM_BINS) h[threadIdx.x] = 0;
  __syncthreads();
  uint32_t b = blockIdx.x * PERM_BLOCK + threadIdx.x;
  if (b < nb) {
    uint32_t c = count[b];
    atomicAdd(&h[c < PERM_BINS - 1 ? c : PERM_BINS - 1], 1u);
  }
  __syncthreads();
  if (threadIdx.x < PERM_BINS) {
    uint32_t mine = h[threadIdx.x];
    blk_base[(size_t)blockIdx.x * PERM_BINS + threadIdx.x] = mine ? atomicAdd(&ghist[threadIdx.x], mine) : 0u;
  }
}
static __global__ void __launch_bounds__(PERM_BLOCK) perm_scatter(const uint32_t* __restrict__ count, uint32_t nb,
                                                                  const uint32_t* __restrict__ ghist,
                                                                  const uint32_t* __restrict__ blk_base,
                                                                  uint32_t* __restrict__ perm) {
  __shared__ uint32_t start[PERM_BINS];   // first position of size class v (descending order)
  __shared__ uint32_t cur[PERM_BINS];
  // suffix sums of the 256-bin histogram: start[v] = sum_{v' > v} ghist[v']   (threads >= PERM_BINS only wait)
  const bool bin = threadIdx.x < PERM_BINS;
  if (bin) {
    start[threadIdx.x] = ghist[threadIdx.x];
    cur[threadIdx.x] = 0;
  }
  __syncthreads();
  for (int d = 1; d < PERM_BINS; d <<= 1) {
    uint32_t add = bin && (int)threadIdx.x + d < PERM_BINS ? start[threadIdx.x + d] : 0u;
    __syncthreads();
    if (bin) start[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t incl = bin ? start[threadIdx.x] : 0u;
  __syncthreads();
  if (bin) start[threadIdx.x] = incl - ghist[threadIdx.x];
  __syncthreads();
  uint32_t b = blockIdx.x * PERM_BLOCK + threadIdx.x;
  if (b < nb) {
    uint32_t c = count[b];
    uint32_t v = c < PERM_BINS - 1 ? c : PERM_BINS - 1;
    uint32_t r = atomicAdd(&cur[v], 1u);
    perm[start[v] + blk_base[(size_t)blockIdx.x * PERM_BINS + v] + r] = b;
  }
}

// ---- K4: bucket-segment accumulation -----------------------------------------------------------
// The accumulate kernels work in the reduced-radix field of ff29.cuh (9 x 29-bit limbs: carry-free columns,
// ~25 % fewer VALU cycles per multiplication than the 8 x 32 form).  They read point tables holding that
// field's canonical values packed in 64 B per G1 point / 128 B per G2 point (written by msm_precompute /
// points_to29, unpacked by bit slicing) and write each bucket sum back in the standard XYZZ layout, which is
// all the later stages ever see.
//
// occupancy target: G1 fits 4 waves/SIMD (<=128 VGPRs); G2 is bounded to 256 registers (2 waves/SIMD;
// 3 waves/SIMD with spills measured slower: 3.88 ms vs 3.52 ms on the 8x32 kernel)
template <class C>
__global__ void __launch_bounds__(ACC_BLOCK, sizeof(typename C::Aff) == 64 ? G16_G1_WAVES : G16_G2_WAVES)
msm_accum(const MsmBatch<C> B, MsmParams P, unsigned long long* __restrict__ clk) {
  using E = Ec29<C>;
  const MsmJob<C>& J = B.job[blockIdx.y];
  const typename E::Tab* __restrict__ points = J.points;
  const uint32_t* __restrict__ entries = J.entries;
  const uint32_t* __restrict__ offset = J.offset;
  // profiling only: shader-clock and 100 MHz stamps around thread 0's task (g16_profile_clock)
  unsigned long long t0 = 0, r0 = 0;
  const bool stamp = clk != nullptr && threadIdx.x == 0;
  if (stamp) t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  // task order = dispatch order: the extra segments of split buckets (the longest tasks, L entries each)
  // first, then the buckets by descending size, so that no long task is left for the tail of the launch
  const uint32_t t = blockIdx.x * ACC_BLOCK + threadIdx.x;
  const uint32_t nx = J.info[1] < P.max_extra ? J.info[1] : P.max_extra;
  uint32_t b, s, slot;
  if (t < nx) {
    uint2 d = J.xseg[t];
    b = d.x;
    s = d.y;
    slot = P.nbuckets + t;
  } else {
    if (t - nx >= P.nbuckets) return;
    b = J.perm[t - nx];   // equal trip counts inside a wave
    s = 0;
    slot = b;
  }
  uint32_t beg = offset[b], end = offset[b + 1];
  beg += s * P.seg;
  if (end > beg + P.seg) end = beg + P.seg;
  // `init`: this MSM continues the bucket sums of another one over the same bucket set (C1 -> H1 of a proof: only
  // their sum enters pi_c, prover.nim:301-302, so the pair needs ONE bucket reduction)
  typename E::Acc acc = E::acc_inf();
  if (J.init != nullptr && s == 0) acc = J.init[b];
  for (uint32_t j = beg; j < end; ++j) {
    const uint32_t e = entries[j];   // point index (table-major) | sign in bit 31
    E::madd(acc, points + (e & 0x7fffffffu), e >> 31);
  }
  J.partial[slot] = acc;   // stays in the reduced-radix form: msm_heavy / msm_reduce1 consume it as is
  if (stamp) {
    atomicAdd(&clk[0], __builtin_amdgcn_s_memtime() - t0);
    atomicAdd(&clk[1], __builtin_amdgcn_s_memrealtime() - r0);
  }
}

// affine points in the reference layout (64 / 128 B, Montgomery R = 2^256) -> reduced-radix table entries
template <class C>
__global__ void __launch_bounds__(MSM_BLOCK) points_to29(const typename C::Aff* __restrict__ points, uint32_t n,
                                                         typename Ec29<C>::Tab* __restrict__ out) {
  uint32_t i = blockIdx.x * MSM_BLOCK + threadIdx.x;
  if (i < n) out[i] = Ec29<C>::tab_from_std(points[i]);
}

// ---- K5: combine the segments of split buckets (one workgroup per heavy bucket) ---------------------
template <class C, int BLOCK>
__device__ __forceinline__ typename C::Acc block_sum(typename C::Acc v, typename C::Acc* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
#pragma unroll 1
  for (int stride = BLOCK / 2; stride > 0; stride >>= 1) {
    if ((int)threadIdx.x < stride) {
      typename C::Acc a = sh[threadIdx.x];
      C::add(a, sh[threadIdx.x + stride]);
      sh[threadIdx.x] = a;
    }
    __syncthreads();
  }
  typename C::Acc r = sh[0];
  __syncthreads();
  return r;
}

// the bucket partials between msm_accum and msm_reduce1 are reduced-radix XYZZ (Ec29<C>::Acc: 144 B G1, 288 B G2);
// HEAVY_BLOCK threads x one partial each must fit the LDS budget of a workgroup
template <class C>
constexpr int heavy_block() { return sizeof(typename C::Aff) == 64 ? 256 : 128; }
// One launch for every split bucket of the batch: buckets with only a few extra segments are summed by one thread
// each (phase 1), buckets with >= HEAVY_MIN by a workgroup with an LDS tree (phase 2).  Both phases stride over the
// same list and touch disjoint buckets.  (Rounds 1-3: two launches, msm_heavy + msm_heavy_small.)
template <class C>
__global__ void __launch_bounds__(heavy_block<C>(), tail_waves<C>()) msm_heavy(const MsmBatch<C> B, MsmParams P) {
  tail_prio();
  using E = Ec29<C>;
  constexpr int HEAVY_BLOCK = heavy_block<C>();
  extern __shared__ __align__(16) unsigned char smem[];
  typename E::Acc* sh = reinterpret_cast<typename E::Acc*>(smem);
  const MsmJob<C>& J = B.job[blockIdx.y];
  const uint32_t* __restrict__ heavy = J.heavy;
  const uint32_t* __restrict__ offset = J.offset;
  const uint32_t* __restrict__ xoff = J.xoff;
  typename E::Acc* __restrict__ partial = J.partial;
  const uint32_t nheavy = J.info[2];
  for (uint32_t h = blockIdx.x * HEAVY_BLOCK + threadIdx.x; h < nheavy; h += gridDim.x * HEAVY_BLOCK) {
    const uint32_t b = heavy[h];
    const uint32_t e = extra_segs(offset[b + 1] - offset[b], P.seg), x0 = xoff[b];
    if (e >= HEAVY_MIN) continue;
    typename E::Acc acc = partial[b];
    for (uint32_t k = 0; k < e; ++k) E::add(acc, partial[P.nbuckets + x0 + k]);
    partial[b] = acc;
  }
  for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
    uint32_t b = heavy[h];
    uint32_t e = extra_segs(offset[b + 1] - offset[b], P.seg), x0 = xoff[b];
    if (e < HEAVY_MIN) continue;   // phase 1's (uniform per workgroup: no barrier is skipped unevenly)
    typename E::Acc acc = E::acc_inf();
    // segment 0 lives at partial[b]; segments 1..e at partial[nbuckets + x0 + s - 1]
    for (uint32_t s = threadIdx.x; s <= e; s += HEAVY_BLOCK) {
      uint32_t idx = s == 0 ? b : P.nbuckets + x0 + s - 1;
      E::add(acc, partial[idx]);
    }
    typename E::Acc r = block_sum<E, HEAVY_BLOCK>(acc, sh);
    if (threadIdx.x == 0) partial[b] = r;
  }
}

// ---- K6: bucket reduction  S_w = sum_{k=1}^{K} k * B_{w,k},  K = 2^(c-1) -----------------------------
// stage 1: thread per chunk of RC consecutive buckets: R_j = sum B_k, A_j = sum (k - j*RC) B_k
// (RC = msm_red_chunk, g16_internal.hpp: 16, or 4 for small bucket sets)
// A job that continues another MSM's bucket sums (MsmJob::init): a bucket without entries of its own still
// holds the other MSM's sum, so every bucket is added (infinity is skipped inside the addition).
template <class C>
__global__ void __launch_bounds__(MSM_BLOCK, tail_waves<C>()) msm_reduce1(const MsmBatch<C> B, uint32_t nbuckets, uint32_t rc) {
  tail_prio();
  using E = Ec29<C>;   // running sums in the reduced-radix field; the chunk sums leave in the standard layout
  const MsmJob<C>& J = B.job[blockIdx.y];
  const typename E::Acc* __restrict__ partial = J.partial;
  const uint32_t* __restrict__ offset = J.offset;
  const bool always = J.init != nullptr;
  uint32_t j = blockIdx.x * MSM_BLOCK + threadIdx.x;
  uint32_t b0 = j * rc;
  if (b0 >= nbuckets) return;
  typename E::Acc run = E::acc_inf(), acc = E::acc_inf();
#pragma unroll 1
  for (int k = (int)rc - 1; k >= 0; --k) {
    uint32_t b = b0 + k;
    if (b < nbuckets && (always || offset[b + 1] != offset[b])) E::add(run, partial[b]);
    E::add(acc, run);
  }
  J.chunkR[j] = E::to_std(run);
  J.chunkA[j] = E::to_std(acc);
}
// stage 2: one workgroup per reduction set (a window, or a slice of the merged bucket set) over its M chunks
// (chunk m holds buckets m*RC+1 .. (m+1)*RC of the set):
//   S = sum_m A_m + RC * sum_m m * R_m ,      sum_m m R_m = sum_t [ wsum_t + lo_t * run_t ]
// Thread t owns chunks [lo_t, lo_t + per): run_t = sum R, wsum_t = sum (m - lo_t) R_m, sumA_t = sum A_m by
// running sums; sum_t lo_t run_t = per * sum_{t>=1} suffix(t) with suffix = inclusive suffix scan of run_t
// (Hillis-Steele in LDS).  Each thread then forms  v_t = RC * (per * suffix_t[t>=1] + wsum_t) + sumA_t  with a
// few doublings, and ONE LDS tree adds the v_t.  The whole kernel is a latency chain of ~40 group operations,
// so the block is as wide as the register budget allows (BLOCK = 512 for G1, 256 for G2).
template <class C, int BLOCK>
__global__ void __launch_bounds__(BLOCK, tail_waves<C>()) msm_reduce2(const MsmBatch<C> B, uint32_t chunks_per_window,
                                                                      uint32_t rc) {
  tail_prio();
  extern __shared__ __align__(16) unsigned char smem[];
  typename C::Acc* sh = reinterpret_cast<typename C::Acc*>(smem);
  const MsmJob<C>& J = B.job[blockIdx.y];
  typename C::Acc* __restrict__ window_sum = J.wsum;        // [0, 64]: set sums
  typename C::Acc* __restrict__ window_tot = J.wsum + 65;   // [65, 129]: sum of every bucket of the set
  const uint32_t w = blockIdx.x, M = chunks_per_window;
  const typename C::Acc* R = J.chunkR + (size_t)w * M;
  const typename C::Acc* A = J.chunkA + (size_t)w * M;
  const uint32_t per = (M + BLOCK - 1) / BLOCK;
  const uint32_t lo = threadIdx.x * per, hi = (lo + per < M) ? lo + per : M;
  typename C::Acc sumA = C::acc_inf(), run = C::acc_inf(), wsum = C::acc_inf();
  if (lo < M) {
    for (uint32_t m = hi; m-- > lo;) {
      C::add(wsum, run);   // each R_m' ends up counted (m' - lo) times
      C::add(run, R[m]);
      C::add(sumA, A[m]);
    }
  }
  // inclusive suffix scan of the slice totals
  sh[threadIdx.x] = run;
  __syncthreads();
  typename C::Acc incl = run;
#pragma unroll 1
  for (int d = 1; d < BLOCK; d <<= 1) {
    typename C::Acc other = C::acc_inf();
    const bool has = (int)threadIdx.x + d < BLOCK;
    if (has) other = sh[threadIdx.x + d];
    __syncthreads();
    if (has) C::add(incl, other);
    sh[threadIdx.x] = incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) window_tot[w] = incl;   // sum of every bucket of this set
  // v_t = RC * (per * suffix_t + wsum_t) + sumA_t
  typename C::Acc v = (threadIdx.x >= 1 && lo < M) ? C::mul_small(incl, per) : C::acc_inf();
  C::add(v, wsum);
  v = C::mul_small(v, rc);
  C::add(v, sumA);
  __syncthreads();
  typename C::Acc tot = block_sum<C, BLOCK>(v, sh);
  if (threadIdx.x == 0) window_sum[w] = tot;
}

// ---- wave-cooperative doubling for the serial tails -----------------------------------------------------
// A doubling chain on ONE lane is bound by the issue rate of its wave (~2800 instructions, ~7 us per doubling on
// MI355X) while 63 lanes idle.  The 9 multiplications + 1 two-term product of dbl-2008-s-1 fall into three levels
// of mutually independent products:
//   level 1:  v = u^2          xx = x^2                        (u = 2y)
//   level 2:  w = u v          s  = x v          mm = m^2      (m = 3 xx)
//   level 3:  y3 = m (s - x3) - w y      zz3 = v zz      zzz3 = w zzz        (x3 = mm - 2s)
// Lanes 0, 1, 2 of the wave take one product of a level each -- the SAME instruction stream on different operands,
// so nothing diverges -- and the results travel between levels by ds_bpermute.  One doubling then costs 3 (+1/2)
// multiplications of wave time instead of 10 1/2.  Every lane must hold the same COORDINATES (not merely the same
// point: products computed by different lanes are combined) on entry, and does on exit; the formulas map the all-zero
// infinity to itself, so no lane branches.
__device__ __forceinline__ u256 wave_get(const u256& v, int src) {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = __shfl(v.v[i], src, 64);
  return r;
}
__device__ __forceinline__ fp2_t wave_get(const fp2_t& v, int src) { return fp2_t{wave_get(v.c0, src), wave_get(v.c1, src)}; }
template <class E>
__device__ __forceinline__ E sel3(uint32_t role, const E& a, const E& b, const E& c) {
  return role == 0 ? a : (role == 1 ? b : c);
}
template <class C>
__device__ __forceinline__ typename C::Acc dbl_coop(const typename C::Acc& p) {
  using F = typename C::Field;
  using E = typename C::E;
  const uint32_t lane = threadIdx.x & 63, role = lane < 2 ? lane : 2;
  const E u = F::dbl(p.y);
  const E l1 = F::sqr(role == 0 ? u : p.x);
  const E v = wave_get(l1, 0), xx = wave_get(l1, 1);
  const E m = F::add(F::dbl(xx), xx);
  const E l2 = F::mul(sel3(role, u, p.x, m), role == 2 ? m : v);
  const E w = wave_get(l2, 0), s = wave_get(l2, 1), mm = wave_get(l2, 2);
  const E x3 = F::sub(mm, F::dbl(s));
  // a b - c d on every lane (c = 0 on lanes 1, 2: the plain product through the same code path)
  const E l3 = F::mulsub(sel3(role, m, v, w), sel3(role, F::sub(s, x3), p.zz, p.zzz), role == 0 ? w : F::zero(), p.y);
  return typename C::Acc{x3, wave_get(l3, 0), wave_get(l3, 1), wave_get(l3, 2)};
}

// ---- quad-cooperative group operations for the latency chains ------------------------------------------------------
// reduce2 and the folds are chains of DEPENDENT group operations run by a handful of waves: one addition on one lane
// is bound by the issue rate of its wave (~4.5 us G1, ~20 us G2) while most of the GPU idles.  The 14 multiplications
// of add-2008-s fall into four levels of mutually independent products:
//   level 1:  u1 = x1 zz2     u2 = x2 zz1     s1 = y1 zzz2    s2 = y2 zzz1          (p = u2 - u1, r = s2 - s1)
//   level 2:  pp = p^2        rr = r^2        z12 = zz1 zz2   z123 = zzz1 zzz2
//   level 3:  ppp = p pp      q = u1 pp       zz3 = z12 pp                          (x3 = rr - ppp - 2 q)
//   level 4:  r (q - x3)      s1 ppp          zzz3 = z123 ppp                       (y3 = the difference)
// The four lanes of a quad hold the SAME operands (the same coordinates, not merely the same point), take one product
// of a level each -- one instruction stream, different operands, nothing diverges inside a quad -- and exchange the
// results by quad-permute DPP moves (no LDS, no barrier).  An addition then costs 4 multiplications of wave time
// instead of 14, a doubling (dbl_coop's three levels) 3 instead of 10 1/2.  Every lane of the quad returns the same
// coordinates.  Branches (infinity, P = +-Q) are uniform inside a quad by construction.
template <int I>
__device__ __forceinline__ uint32_t quad_word(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, I * 0x55, 0xf, 0xf, false);   // quad_perm:[I,I,I,I]
}
template <int I>
__device__ __forceinline__ u256 quad_get(const u256& v) {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = quad_word<I>(v.v[i]);
  return r;
}
template <int I>
__device__ __forceinline__ fp2_t quad_get(const fp2_t& v) { return fp2_t{quad_get<I>(v.c0), quad_get<I>(v.c1)}; }
template <class E>
__device__ __forceinline__ E sel4(uint32_t role, const E& a, const E& b, const E& c, const E& d) {
  return role == 0 ? a : (role == 1 ? b : (role == 2 ? c : d));
}
template <class C>
__device__ __forceinline__ typename C::Acc dbl_quad(const typename C::Acc& p) {
  using F = typename C::Field;
  using E = typename C::E;
  const uint32_t q = threadIdx.x & 3, role = q < 2 ? q : 2;   // lane 3 repeats lane 2's product
  const E u = F::dbl(p.y);
  const E l1 = F::sqr(role == 0 ? u : p.x);
  const E v = quad_get<0>(l1), xx = quad_get<1>(l1);
  const E m = F::add(F::dbl(xx), xx);
  const E l2 = F::mul(sel3(role, u, p.x, m), role == 2 ? m : v);
  const E w = quad_get<0>(l2), s = quad_get<1>(l2), mm = quad_get<2>(l2);
  const E x3 = F::sub(mm, F::dbl(s));
  const E l3 = F::mulsub(sel3(role, m, v, w), sel3(role, F::sub(s, x3), p.zz, p.zzz), role == 0 ? w : F::zero(), p.y);
  return typename C::Acc{x3, quad_get<0>(l3), quad_get<1>(l3), quad_get<2>(l3)};   // infinity (all zero) maps to itself
}
template <class C>
__device__ __forceinline__ typename C::Acc add_quad(const typename C::Acc& a, const typename C::Acc& b) {
  using F = typename C::Field;
  using E = typename C::E;
  if (C::is_inf(b)) return a;
  if (C::is_inf(a)) return b;
  const uint32_t role = threadIdx.x & 3;
  const E l1 = F::mul(sel4(role, a.x, b.x, a.y, b.y), sel4(role, b.zz, a.zz, b.zzz, a.zzz));
  const E u1 = quad_get<0>(l1), u2 = quad_get<1>(l1), s1 = quad_get<2>(l1), s2 = quad_get<3>(l1);
  const E p = F::sub(u2, u1), r = F::sub(s2, s1);
  if (F::is_zero(p)) return F::is_zero(r) ? dbl_quad<C>(a) : C::acc_inf();
  const E l2 = F::mul(sel4(role, p, r, a.zz, a.zzz), sel4(role, p, r, b.zz, b.zzz));
  const E pp = quad_get<0>(l2), rr = quad_get<1>(l2), z12 = quad_get<2>(l2), z123 = quad_get<3>(l2);
  const E l3 = F::mul(sel4(role, p, u1, z12, z12), pp);
  const E ppp = quad_get<0>(l3), qq = quad_get<1>(l3), zz3 = quad_get<2>(l3);
  const E x3 = F::sub(F::sub(rr, ppp), F::dbl(qq));
  const E l4 = F::mul(sel4(role, r, s1, z123, z123), role == 0 ? F::sub(qq, x3) : ppp);
  return typename C::Acc{x3, F::sub(quad_get<0>(l4), quad_get<1>(l4)), zz3, quad_get<2>(l4)};
}
// k * P for a small k that is uniform over the quad
template <class C>
__device__ __forceinline__ typename C::Acc mul_small_quad(const typename C::Acc& p, uint32_t k) {
  typename C::Acc r = C::acc_inf();
#pragma unroll 1
  for (int b = k ? 31 - __clz(k) : -1; b >= 0; --b) {
    r = dbl_quad<C>(r);
    if ((k >> b) & 1) r = add_quad<C>(r, p);
  }
  return r;
}
// sum over the SLOTS quads of a workgroup (4 * SLOTS threads), returned on every thread
template <class C, int SLOTS>
__device__ __forceinline__ typename C::Acc block_sum_quad(const typename C::Acc& v, typename C::Acc* sh) {
  const int slot = threadIdx.x >> 2;
  __syncthreads();
  if ((threadIdx.x & 3) == 0) sh[slot] = v;
  __syncthreads();
#pragma unroll 1
  for (int stride = SLOTS / 2; stride > 0; stride >>= 1) {
    if (slot < stride) {
      const typename C::Acc a = add_quad<C>(sh[slot], sh[slot + stride]);
      if ((threadIdx.x & 3) == 0) sh[slot] = a;
    }
    __syncthreads();
  }
  const typename C::Acc r = sh[0];
  __syncthreads();
  return r;
}

// stage 2 of the bucket reduction (msm_reduce2's algorithm and outputs) with a quad per slot: 4 * SLOTS threads
template <class C, int SLOTS>
__global__ void __launch_bounds__(4 * SLOTS, tail_waves<C>()) msm_reduce2_quad(const MsmBatch<C> B, uint32_t chunks_per_window,
                                                                               uint32_t rc) {
  tail_prio();
  extern __shared__ __align__(16) unsigned char smem[];
  typename C::Acc* sh = reinterpret_cast<typename C::Acc*>(smem);
  const MsmJob<C>& J = B.job[blockIdx.y];
  const uint32_t w = blockIdx.x, M = chunks_per_window;
  const typename C::Acc* R = J.chunkR + (size_t)w * M;
  const typename C::Acc* A = J.chunkA + (size_t)w * M;
  const uint32_t slot = threadIdx.x >> 2, q = threadIdx.x & 3;
  const uint32_t per = (M + SLOTS - 1) / SLOTS;
  const uint32_t lo = slot * per, hi = (lo + per < M) ? lo + per : M;
  typename C::Acc sumA = C::acc_inf(), run = C::acc_inf(), wsum = C::acc_inf();
  if (lo < M) {
#pragma unroll 1
    for (uint32_t m = hi; m-- > lo;) {
      wsum = add_quad<C>(wsum, run);
      run = add_quad<C>(run, R[m]);
      sumA = add_quad<C>(sumA, A[m]);
    }
  }
  if (q == 0) sh[slot] = run;
  __syncthreads();
  typename C::Acc incl = run;
#pragma unroll 1
  for (int d = 1; d < SLOTS; d <<= 1) {
    typename C::Acc other = C::acc_inf();
    const bool has = (int)slot + d < SLOTS;
    if (has) other = sh[slot + d];
    __syncthreads();
    if (has) incl = add_quad<C>(incl, other);
    if (q == 0) sh[slot] = incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) J.wsum[65 + w] = incl;
  typename C::Acc v = (slot >= 1 && lo < M) ? mul_small_quad<C>(incl, per) : C::acc_inf();
  v = add_quad<C>(v, wsum);
  v = mul_small_quad<C>(v, rc);
  v = add_quad<C>(v, sumA);
  const typename C::Acc tot = block_sum_quad<C, SLOTS>(v, sh);
  if (threadIdx.x == 0) J.wsum[w] = tot;
}

// ---- K7: fold windows + canonical affine ------------------------------------------------------------
// c > 0: Horner  sum_w 2^(c w) S_w  (c doublings per window: a serial chain of ~254 doublings);
// c == 0: the window sums already carry their 2^(c w) factor (precomputed tables) -> plain sum.
// One wave; every lane carries the same running point, the doublings are wave-cooperative (dbl_coop).
template <class C>
__global__ void __launch_bounds__(64) msm_fold(const MsmBatch<C> B, uint32_t nwin, uint32_t c) {
  tail_prio();
  const MsmJob<C>& J = B.job[blockIdx.y];
  const typename C::Acc* __restrict__ window_sum = J.wsum;
  typename C::Aff* __restrict__ out_aff = J.out_aff;
  typename C::Acc* __restrict__ out_acc = J.out_acc;
  typename C::Acc r = C::acc_inf();
  for (int w = (int)nwin - 1; w >= 0; --w) {
    if (!C::is_inf(r))   // uniform: r is the same on every lane
#pragma unroll 1
      for (uint32_t i = 0; i < c; ++i) r = dbl_coop<C>(r);
    C::add(r, window_sum[w]);
  }
  if (threadIdx.x == 0) {
    if (out_acc) *out_acc = r;
    if (out_aff) *out_aff = C::to_affine(r);
  }
}

// ---- K7': fold for the merged bucket set of a registered point set ------------------------------------
// The 2^(c-1) buckets were reduced in `nsets` <= 64 slices of Ks = 2^log2ks buckets with slice-local weights
// 1..Ks; slice v starts at bucket v*Ks, so   S = sum_v S_v + Ks * sum_v v * Tot_v .
// Two waves, no LDS trees: wave 1 all-reduces the S_v; wave 0 forms sum_v v Tot_v as the sum over v >= 1 of the
// suffix sums of Tot (one shuffle scan + one shuffle all-reduce: 12 additions, no per-lane double-and-add), applies
// the log2(Ks) doublings cooperatively (dbl_coop) and finishes.
template <class C>
__device__ __forceinline__ typename C::Acc wave_get_acc(const typename C::Acc& a, int src) {
  return typename C::Acc{wave_get(a.x, src), wave_get(a.y, src), wave_get(a.zz, src), wave_get(a.zzz, src)};
}
template <class C>
__device__ __forceinline__ typename C::Acc wave_allreduce(typename C::Acc v) {
  const int lane = threadIdx.x & 63;
#pragma unroll 1
  for (int d = 1; d < 64; d <<= 1) {
    typename C::Acc other = wave_get_acc<C>(v, lane ^ d);
    C::add(v, other);
  }
  // Every lane now holds the same POINT but not the same coordinates: a + b and b + a differ in the sign of
  // (Y, ZZZ).  Callers that go on cooperatively (dbl_coop mixes coordinates across lanes) need one representation:
  // lane 0's.
  return wave_get_acc<C>(v, 0);
}
template <class C>
__global__ void __launch_bounds__(128, tail_waves<C>()) msm_fold_merged(const MsmBatch<C> B, uint32_t nsets, uint32_t log2ks) {
  tail_prio();
  __shared__ typename C::Acc ysum;
  const MsmJob<C>& J = B.job[blockIdx.y];
  const typename C::Acc* __restrict__ set_sum = J.wsum;
  const typename C::Acc* __restrict__ set_tot = J.wsum + 65;
  typename C::Aff* __restrict__ out_aff = J.out_aff;
  typename C::Acc* __restrict__ out_acc = J.out_acc;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave == 1) {
    typename C::Acc y = (uint32_t)lane < nsets ? set_sum[lane] : C::acc_inf();
    y = wave_allreduce<C>(y);
    if (lane == 0) ysum = y;
  }
  typename C::Acc xs = C::acc_inf();
  if (wave == 0) {
    // inclusive suffix scan of Tot over the lanes, then sum the suffixes of v = 1 .. nsets-1
    typename C::Acc suf = (uint32_t)lane < nsets ? set_tot[lane] : C::acc_inf();
#pragma unroll 1
    for (int d = 1; d < 64; d <<= 1) {
      typename C::Acc other = wave_get_acc<C>(suf, (lane + d) & 63);
      if (lane + d < 64) C::add(suf, other);
    }
    if (lane == 0) suf = C::acc_inf();   // v = 0 carries weight 0
    xs = wave_allreduce<C>(suf);
#pragma unroll 1
    for (uint32_t i = 0; i < log2ks; ++i) xs = dbl_coop<C>(xs);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    C::add(xs, ysum);
    if (out_acc) *out_acc = xs;
    if (out_aff) *out_aff = C::to_affine(xs);
  }
}

// ---- K7'': fold for the class bucket set (MsmParams::mtab == 2, msm_class_bucket) --------------------------------
// 43 slices of Ks = 2^log2ks buckets, reduced with slice-local weights l = 1..Ks (set_sum W_v, set_tot T_v): slices
// 0..31 are class 0, 32..39 class 1, 40..41 class 2 (bucket weight 4^z (2 (sigma Ks + l - 1) + 1), sigma = slice
// inside the class), slice 42 is class X (weight 64 l).  Hence
//   S = sum_v y_v + 2 Ks * sum_v e_v T_v ,   y_v = 4^z (2 W_v - T_v)  (X: 64 W_v),   e_v = 4^z sigma  (X: 0; all < 32).
// Wave 1 forms the y_v (one doubling, one addition, <= 5 more doublings per lane) and all-reduces them; wave 0 forms
// sum e_v T_v as the sum over sigma >= 1 of the per-class suffix sums of T (segmented shuffle scan), scaled by 4^z per
// lane, all-reduces, applies the log2(2 Ks) doublings cooperatively and finishes.
template <class C>
__global__ void __launch_bounds__(128, tail_waves<C>()) msm_fold_classes(const MsmBatch<C> B, uint32_t log2ks) {
  tail_prio();
  __shared__ typename C::Acc ysum;
  const MsmJob<C>& J = B.job[blockIdx.y];
  const typename C::Acc* __restrict__ set_sum = J.wsum;
  const typename C::Acc* __restrict__ set_tot = J.wsum + 65;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // class, slice inside the class, first lane of the next class
  const int z = lane < 32 ? 0 : lane < 40 ? 1 : lane < 42 ? 2 : 3;
  const int first = z == 0 ? 0 : z == 1 ? 32 : z == 2 ? 40 : 42, end = z == 0 ? 32 : z == 1 ? 40 : z == 2 ? 42 : 43;
  const bool live = lane < (int)MSM_CLASS_SLICES;
  if (wave == 1) {
    typename C::Acc y = live ? set_sum[lane] : C::acc_inf();
    y = C::dbl(y);                                             // 2 W
    if (live && z < 3) C::add(y, C::neg(set_tot[lane]));       // - T
    const int nd = z == 0 ? 0 : z == 1 ? 2 : z == 2 ? 4 : 5;  // 4^z; X: 2 W -> 64 W
#pragma unroll 1
    for (int i = 0; i < 5; ++i) {
      typename C::Acc d = C::dbl(y);
      if (i < nd) y = d;
    }
    y = wave_allreduce<C>(y);
    if (lane == 0) ysum = y;
  }
  typename C::Acc xs = C::acc_inf();
  if (wave == 0) {
    // inclusive suffix scan of T inside each class, then the suffixes of sigma >= 1, scaled by 4^z
    typename C::Acc suf = (live && z < 3) ? set_tot[lane] : C::acc_inf();
#pragma unroll 1
    for (int d = 1; d < 32; d <<= 1) {
      typename C::Acc other = wave_get_acc<C>(suf, (lane + d) & 63);
      if (lane + d < end && z < 3) C::add(suf, other);
    }
    if (lane == first || z == 3 || !live) suf = C::acc_inf();   // sigma = 0 carries weight 0
    const int nd = z == 1 ? 2 : z == 2 ? 4 : 0;
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
      typename C::Acc d = C::dbl(suf);
      if (i < nd) suf = d;
    }
    xs = wave_allreduce<C>(suf);
#pragma unroll 1
    for (uint32_t i = 0; i < log2ks + 1; ++i) xs = dbl_coop<C>(xs);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    C::add(xs, ysum);
    if (J.out_acc) *J.out_acc = xs;
    if (J.out_aff) *J.out_aff = C::to_affine(xs);
  }
}

// msm_fold_classes with a quad per slice: threads 0..255 (role A) form  sum e_v T_v  and finish, threads 256..511
// (role B) form  sum y_v.  The barriers are shared, so B's slice-local doublings run while A scales its suffixes.
template <class C>
__global__ void __launch_bounds__(512, tail_waves<C>()) msm_fold_classes_quad(const MsmBatch<C> B, uint32_t log2ks) {
  tail_prio();
  extern __shared__ __align__(16) unsigned char smem[];
  typename C::Acc* shA = reinterpret_cast<typename C::Acc*>(smem);   // 64 + 64 accumulators
  const MsmJob<C>& J = B.job[blockIdx.y];
  const typename C::Acc* __restrict__ set_sum = J.wsum;
  const typename C::Acc* __restrict__ set_tot = J.wsum + 65;
  const bool roleB = threadIdx.x >= 256;
  typename C::Acc* sh = shA + (roleB ? 64 : 0);
  const int lane = (threadIdx.x & 255) >> 2, q = threadIdx.x & 3;   // "lane" = slice, as in msm_fold_classes
  const int z = lane < 32 ? 0 : lane < 40 ? 1 : lane < 42 ? 2 : 3;
  const int first = z == 0 ? 0 : z == 1 ? 32 : z == 2 ? 40 : 42, end = z == 0 ? 32 : z == 1 ? 40 : z == 2 ? 42 : 43;
  const bool live = lane < (int)MSM_CLASS_SLICES;
  // A: inclusive suffix scan of T inside each class
  typename C::Acc val = (!roleB && live && z < 3) ? set_tot[lane] : C::acc_inf();
  if (q == 0) sh[lane] = val;
  __syncthreads();
#pragma unroll 1
  for (int d = 1; d < 32; d <<= 1) {
    const bool has = !roleB && lane + d < end && z < 3;
    typename C::Acc other = C::acc_inf();
    if (has) other = sh[lane + d];
    __syncthreads();
    if (has) val = add_quad<C>(val, other);
    if (q == 0) sh[lane] = val;
    __syncthreads();
  }
  if (!roleB) {
    if (lane == first || z == 3 || !live) val = C::acc_inf();   // sigma = 0 carries weight 0
    const int nd = z == 1 ? 2 : z == 2 ? 4 : 0;
#pragma unroll 1
    for (int i = 0; i < nd; ++i) val = dbl_quad<C>(val);
  } else {
    val = live ? set_sum[lane] : C::acc_inf();
    val = dbl_quad<C>(val);                                                  // 2 W
    if (live && z < 3) val = add_quad<C>(val, C::neg(set_tot[lane]));        // - T
    const int nd = z == 0 ? 0 : z == 1 ? 2 : z == 2 ? 4 : 5;                // 4^z; X: 2 W -> 64 W
#pragma unroll 1
    for (int i = 0; i < nd; ++i) val = dbl_quad<C>(val);
  }
  // both sums: the same tree on either half (block_sum_quad over 64 slots, indexed inside the role)
  __syncthreads();
  if (q == 0) sh[lane] = val;
  __syncthreads();
#pragma unroll 1
  for (int stride = 32; stride > 0; stride >>= 1) {
    if (lane < stride) {
      const typename C::Acc a = add_quad<C>(sh[lane], sh[lane + stride]);
      if (q == 0) sh[lane] = a;
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) {
    typename C::Acc xs = shA[0];
#pragma unroll 1
    for (uint32_t i = 0; i < log2ks + 1; ++i) xs = dbl_quad<C>(xs);
    xs = add_quad<C>(xs, shA[64]);
    if (threadIdx.x == 0) {
      if (J.out_acc) *J.out_acc = xs;
      if (J.out_aff) *J.out_aff = C::to_affine(xs);
    }
  }
}

// ---- registration-time precomputation:  table[w][i] = 2^(c w) * P_i  (affine), w = 0..nwin-1 ---------
// One thread per point: c doublings per window, one inversion per stored point.  Runs once per circuit
// (the reference loads ProverPoints once per zkey, zkey_types.nim:36-41); trades HBM capacity
// (nwin x the point set) for the removal of the serial doubling chain from every MSM.
// mtab == 2: a second set of tables [1][w][i] = 2 * 2^(c w) P_i behind the first (MsmParams::mtab)
template <class C>
__global__ void __launch_bounds__(MSM_BLOCK) msm_precompute(const typename C::Aff* __restrict__ points, uint32_t n,
                                                            uint32_t c, uint32_t nwin, uint32_t mtab,
                                                            typename Ec29<C>::Tab* __restrict__ tables) {
  uint32_t i = blockIdx.x * MSM_BLOCK + threadIdx.x;
  if (i >= n) return;
  typename C::Aff p = points[i];
  tables[i] = Ec29<C>::tab_from_std(p);
  typename C::Acc acc = C::from_affine(p);
  for (uint32_t w = 0; w < nwin; ++w) {
    if (w) tables[(size_t)w * n + i] = Ec29<C>::tab_from_std(C::to_affine(acc));
    acc = C::dbl(acc);
    if (mtab == 2) tables[((size_t)nwin + w) * n + i] = Ec29<C>::tab_from_std(C::to_affine(acc));
    for (uint32_t k = 1; k < c; ++k) acc = C::dbl(acc);
  }
}

// ---- on-curve check of a point array: y^2 == x^3 + b, (0,0) = infinity is accepted ------------------------
// the reference asserts this for every point it loads (mkG1 / mkG2, curves.nim:95-107; loadPointsG1/G2, io.nim:240-250)
template <class C>
__global__ void __launch_bounds__(256) points_on_curve(const typename C::Aff* __restrict__ pts, uint32_t n,
                                                       typename C::E b, uint32_t* __restrict__ first_bad) {
  using F = typename C::Field;
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  typename C::Aff p = pts[i];
  if (C::is_inf(p)) return;
  typename C::E lhs = F::sqr(p.y);
  typename C::E rhs = F::add(F::mul(F::sqr(p.x), p.x), b);
  if (!F::eq(lhs, rhs)) atomicMin(first_bad, i);
}

// ---- live bitmap of a point array: bit i set <=> point i is not the point at infinity (0,0) ------------------------
// bitmap: ceil(n/32) words, zero-initialised is not required (every word is written); *n_inf += infinity points
template <class C>
__global__ void __launch_bounds__(256) points_live_bitmap(const typename C::Aff* __restrict__ pts, uint32_t n,
                                                          uint32_t* __restrict__ bitmap, uint32_t* __restrict__ n_inf) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const bool livep = i < n && !C::is_inf(pts[i]);
  const unsigned long long m = __ballot(livep);
  const uint32_t lane = threadIdx.x & 63;
  if (lane == 0 && i < n) {
    bitmap[i >> 5] = (uint32_t)m;
    if (i + 32 < n) bitmap[(i >> 5) + 1] = (uint32_t)(m >> 32);
    const uint32_t valid = n - i < 64 ? n - i : 64;
    const uint32_t dead = valid - (uint32_t)__popcll(m);
    if (dead) atomicAdd(n_inf, dead);
  }
}
// a |= b (word-wise); *n_dead += pairs that are dead in the union
static __global__ void __launch_bounds__(256) bitmap_or(uint32_t* __restrict__ out, const uint32_t* __restrict__ a,
                                                        const uint32_t* __restrict__ b, uint32_t n,
                                                        uint32_t* __restrict__ n_dead) {
  const uint32_t w = blockIdx.x * 256 + threadIdx.x;
  if (w >= (n + 31) / 32) return;
  uint32_t v = a[w] | b[w];
  out[w] = v;
  const uint32_t valid = n - 32 * w < 32 ? n - 32 * w : 32;
  const uint32_t mask = valid == 32 ? 0xffffffffu : (1u << valid) - 1;
  const uint32_t dead = valid - __popc(v & mask);
  if (dead) atomicAdd(n_dead, dead);
}

// ---- fixed-base multiples of the group generator: out[i] = k_i * G  (fake_setup.nim:258-261 `y ** gen`) ----
// table[w*255 + d-1] = d * 2^(8w) * G, w < 32, d = 1..255 (built once per context)
template <class C>
__global__ void __launch_bounds__(256) fixed_base_table(typename C::Aff gen, typename C::Aff* __restrict__ table) {
  uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 32 * 255) return;
  uint32_t w = t / 255, d = t % 255 + 1;
  typename C::Acc acc = C::mul_small(C::from_affine(gen), d);
  for (uint32_t i = 0; i < 8 * w; ++i) acc = C::dbl(acc);
  table[t] = C::to_affine(acc);
}
template <class C>
__global__ void __launch_bounds__(256) fixed_base_mul(const u256* __restrict__ scalars, uint32_t mont, uint32_t n,
                                                      const typename C::Aff* __restrict__ table,
                                                      typename C::Aff* __restrict__ out) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  u256 s = scalars[i];
  if (mont) s = Fr::from_mont(s);
  typename C::Acc acc = C::acc_inf();
#pragma unroll 1
  for (uint32_t j = 0; j < 8; ++j) {
    uint32_t limb = s.v[0];
    // rotate limbs so that the index stays static (a runtime-indexed register array would go to scratch)
#pragma unroll
    for (int q = 0; q < 7; ++q) s.v[q] = s.v[q + 1];
    s.v[7] = limb;
#pragma unroll 1
    for (uint32_t b = 0; b < 4; ++b) {
      uint32_t d = (limb >> (8 * b)) & 0xff;
      if (d) C::madd(acc, table[(4 * j + b) * 255 + d - 1]);
    }
  }
  out[i] = C::to_affine(acc);
}

}  // namespace g16
