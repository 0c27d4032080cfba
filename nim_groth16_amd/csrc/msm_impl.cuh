// Host launch sequence of the MSM pipeline, instantiated once per group in msm_g1.hip / msm_g2.hip.
#pragma once
#include "g16_internal.hpp"
#include "msm.cuh"

using namespace g16;

// ---- MSM ------------------------------------------------------------------------------------------
static uint32_t floor_log2(size_t x) {
  uint32_t k = 0;
  while (x >>= 1) ++k;
  return k;
}

// Window size by a cost model: accumulation = n * nwin mixed adds (~10 modmul each); bucket reduction =
// 2 XYZZ adds (~14 modmul each) per bucket, over nwin bucket sets -- or over ONE set when the points come
// with precomputed 2^(c w) tables (`merged`).  A short top window (t = 254 - (nwin-1) c bits) would map all n
// scalars onto 2^t buckets, so candidates need t >= min(c-2, 6).  2^20 points: c = 16 plain, c = 20 merged
// (13 tables instead of 16 windows).
static uint32_t pick_window_cost(size_t n, bool merged, const char* env_name, uint32_t cmax) {
  if (const char* env = getenv(env_name)) {
    int v = atoi(env);
    if (v >= 5 && v <= 22) return (uint32_t)v;
  }
  uint32_t best = 5;
  double best_cost = 1e300;
  for (uint32_t c = 5; c <= cmax; ++c) {
    const uint32_t nwin = FR_BITS / c + 1;
    if (((size_t)nwin * n) >> 31) continue;   // table index / entry count must fit 31 bits
    const uint32_t t = FR_BITS - (nwin - 1) * c, tmin = c - 2 < 6 ? c - 2 : 6;
    if (t < tmin && c > 5) continue;
    const double sets = merged ? 1.0 : (double)nwin;
    const double cost = 10.0 * (double)n * nwin + 28.0 * sets * (double)(1u << (c - 1));
    if (cost < best_cost) {
      best_cost = cost;
      best = c;
    }
  }
  return best;
}
static uint32_t pick_window(size_t n) { return pick_window_cost(n ? n : 1, false, "G16_MSM_WINDOW", 16); }

static MsmParams msm_params(size_t n, uint32_t flags, uint32_t table_c) {
  MsmParams P;
  P.n = (uint32_t)n;
  P.c = table_c ? table_c : pick_window(n);
  P.nwin = FR_BITS / P.c + 1;
  P.tables = table_c ? 1u : 0u;
  P.nbuckets = P.tables ? (1u << (P.c - 1)) : (P.nwin << (P.c - 1));
  // segment length L: one accumulate task handles <= L entries.  A task is a serial chain of L mixed adds
  // (~23 us each with 4 waves per SIMD), so L also bounds the tail of the launch; ~1.25 x the mean bucket size
  // keeps most buckets in one segment, the rest get 1-2 short extra segments that msm_reduce1 absorbs.
  size_t avg = ((size_t)n * P.nwin) / P.nbuckets + 1;
  P.seg = (uint32_t)(((avg + avg / 4 + 15) / 16) * 16);
  if (P.seg < 32) P.seg = 32;
  if (const char* env = getenv("G16_MSM_SEG")) {
    int v = atoi(env);
    if (v >= 8 && v <= 4096) P.seg = (uint32_t)v;
  }
  P.scalars_mont = (flags & G16_SCALARS_MONT) ? 1u : 0u;
  P.max_extra = (uint32_t)(((size_t)P.n * P.nwin) / P.seg + 1);
  return P;
}

// ---- phase 1: scalars -> bucket arrangement (count, scan, scatter, extra-segment list) ---------------------
static int32_t msm_sort_device(g16_ctx* ctx, hipStream_t st, const void* d_scalars, uint32_t flags, size_t n,
                               uint32_t table_c, g16_ctx::MsmSort& S) {
  const MsmParams P = msm_params(n, flags, table_c);
  S.P = P;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t r = o;
    o += (bytes + 255) & ~size_t(255);
    return r;
  };
  const size_t nb = P.nbuckets;
  const size_t o_count = take(nb * 4), o_cursor = take(nb * 4), o_offset = take((nb + 1) * 4), o_xoff = take(nb * 4),
               o_heavy = take(nb * 4), o_info = take(64), o_tiles = take(((nb + SCAN_TILE - 1) / SCAN_TILE) * 8),
               o_entries = take((size_t)P.n * P.nwin * 4), o_xseg = take((size_t)P.max_extra * 8),
               o_perm = take(nb * 4), o_ghist = take(PERM_BINS * 4),
               o_blk = take(((nb + PERM_BLOCK - 1) / PERM_BLOCK) * PERM_BINS * 4);
  // partition sort (see msm.cuh): low bits <= 8, partitions = nwin << hi_bits
  const uint32_t lo_bits = P.c - 1 < 8 ? P.c - 1 : 8;
  const uint32_t nparts = P.nbuckets >> lo_bits;
  const uint32_t ptiles = (P.n + PART_TILE - 1) / PART_TILE;
  const char* env_sort = getenv("G16_MSM_SORT");
  const bool use_part = nparts <= PART_MAX && !(env_sort && env_sort[0] == 'a');
  const size_t nth = (size_t)nparts * ptiles;
  const size_t o_thist = take(use_part ? nth * 4 : 4), o_tmp = take(use_part ? (size_t)P.n * P.nwin * 8 : 8),
               o_tiles2 = take(((nth + SCAN_TILE - 1) / SCAN_TILE) * 8 + 8),
               o_shist = take(use_part ? (size_t)nparts * BS_SPLIT * 256 * 4 : 4);
  int32_t rc = ensure(ctx, S.buf, o);
  if (rc) return rc;
  char* ws = (char*)S.buf.p;
  S.count = (uint32_t*)(ws + o_count);
  S.cursor = (uint32_t*)(ws + o_cursor);
  S.offset = (uint32_t*)(ws + o_offset);
  S.xoff = (uint32_t*)(ws + o_xoff);
  S.heavy = (uint32_t*)(ws + o_heavy);
  S.info = (uint32_t*)(ws + o_info);
  S.tiles = (uint2*)(ws + o_tiles);
  S.entries = (uint32_t*)(ws + o_entries);
  S.xseg = (uint2*)(ws + o_xseg);
  S.perm = (uint32_t*)(ws + o_perm);
  S.ghist = (uint32_t*)(ws + o_ghist);
  S.blk_base = (uint32_t*)(ws + o_blk);
  S.tile_hist = (uint32_t*)(ws + o_thist);
  S.tmp = (uint2*)(ws + o_tmp);
  S.tiles2 = (uint2*)(ws + o_tiles2);
  S.slice_hist = (uint32_t*)(ws + o_shist);
  const auto* scalars = (const u256*)d_scalars;
  HIPCHK(ctx, hipMemsetAsync(ws + o_count, 0, o_offset - o_count, st));  // count + cursor are adjacent
  HIPCHK(ctx, hipMemsetAsync(S.info, 0, 64, st));
  HIPCHK(ctx, hipMemsetAsync(S.ghist, 0, PERM_BINS * 4, st));
  const uint32_t nblk = (P.n + MSM_BLOCK - 1) / MSM_BLOCK;
  const uint32_t ntiles = (P.nbuckets + SCAN_TILE - 1) / SCAN_TILE;
  if (use_part) {
    const uint32_t nt2 = (uint32_t)((nth + SCAN_TILE - 1) / SCAN_TILE);
    KLAUNCH_ON(ctx, st, "msm_part_count", part_pass<false>, ptiles, PART_BLOCK, 0, scalars, P, lo_bits, nparts, ptiles,
               S.tile_hist, S.tmp);
    KLAUNCH_ON(ctx, st, "msm_scan", scan1_tile_sums, nt2, SCAN_BLOCK, 0, S.tile_hist, (uint32_t)nth, S.tiles2);
    KLAUNCH_ON(ctx, st, "msm_scan", scan_tiles, 1, SCAN_BLOCK, 0, S.tiles2, nt2, S.info + 8);  // total -> info[8]
    KLAUNCH_ON(ctx, st, "msm_scan", scan1_apply, nt2, SCAN_BLOCK, 0, S.tile_hist, (uint32_t)nth, S.tiles2);
    KLAUNCH_ON(ctx, st, "msm_part_scatter", part_pass<true>, ptiles, PART_BLOCK, 0, scalars, P, lo_bits, nparts,
               ptiles, S.tile_hist, S.tmp);
    KLAUNCH_ON(ctx, st, "msm_bucket_sort", bucket_hist, nparts * BS_SPLIT, 256, 0, S.tmp, S.tile_hist, ptiles, nparts,
               S.info + 8, S.slice_hist);
    KLAUNCH_ON(ctx, st, "msm_bucket_sort", bucket_place, nparts * BS_SPLIT, 256, 0, S.tmp, S.tile_hist, ptiles, nparts,
               S.info + 8, S.slice_hist, P, lo_bits, S.count, S.offset, S.entries);
  } else {
    KLAUNCH_ON(ctx, st, "msm_count", msm_count, nblk, MSM_BLOCK, 0, scalars, P, S.count);
  }
  KLAUNCH_ON(ctx, st, "msm_scan", scan_tile_sums, ntiles, SCAN_BLOCK, 0, S.count, P.nbuckets, P.seg, S.tiles);
  KLAUNCH_ON(ctx, st, "msm_scan", scan_tiles, 1, SCAN_BLOCK, 0, S.tiles, ntiles, S.info);
  KLAUNCH_ON(ctx, st, "msm_scan", scan_apply, ntiles, SCAN_BLOCK, 0, S.count, P.nbuckets, P.seg, S.tiles, S.offset,
             S.xoff, S.heavy, S.info);
  const uint32_t pblk = (P.nbuckets + PERM_BLOCK - 1) / PERM_BLOCK;
  KLAUNCH_ON(ctx, st, "msm_perm", perm_hist, pblk, PERM_BLOCK, 0, S.count, P.nbuckets, S.ghist, S.blk_base);
  KLAUNCH_ON(ctx, st, "msm_perm", perm_scatter, pblk, PERM_BLOCK, 0, S.count, P.nbuckets, S.ghist, S.blk_base,
             S.perm);
  if (!use_part)
    KLAUNCH_ON(ctx, st, "msm_scatter", msm_scatter, nblk, MSM_BLOCK, 0, scalars, P, S.offset, S.cursor, S.entries);
  KLAUNCH_ON(ctx, st, "msm_make_extra", msm_make_extra, 512, MSM_BLOCK, 0, S.heavy, S.info, S.offset, S.xoff, P.seg,
             P.max_extra, S.xseg);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}

// ---- phase 2: accumulate + reduce one point set against a bucket arrangement --------------------------------
// d_out_aff / d_out_acc: device pointers (either may be null).
template <class C>
static int32_t msm_reduce_device(g16_ctx* ctx, hipStream_t st, g16_ctx::Buf& acc, const g16_ctx::MsmSort& S,
                                 const void* d_points, typename C::Aff* d_out_aff, typename C::Acc* d_out_acc) {
  const MsmParams& P = S.P;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t r = o;
    o += (bytes + 255) & ~size_t(255);
    return r;
  };
  const size_t nchunks = P.nbuckets / RED_CHUNK;
  const size_t o_partial = take(((size_t)P.nbuckets + P.max_extra) * sizeof(typename C::Acc)),
               o_chunkR = take(nchunks * sizeof(typename C::Acc)), o_chunkA = take(nchunks * sizeof(typename C::Acc)),
               o_wsum = take((size_t)(2 * 64 + 2) * sizeof(typename C::Acc));
  int32_t rc = ensure(ctx, acc, o);
  if (rc) return rc;
  char* ws = (char*)acc.p;
  auto* partial = (typename C::Acc*)(ws + o_partial);
  auto* chunkR = (typename C::Acc*)(ws + o_chunkR);
  auto* chunkA = (typename C::Acc*)(ws + o_chunkA);
  auto* wsum = (typename C::Acc*)(ws + o_wsum);
  const auto* points = (const typename C::Aff*)d_points;
  const bool g2 = sizeof(typename C::Aff) == 128;
  const uint32_t ntask = P.nbuckets + P.max_extra;
  KLAUNCH_ON(ctx, st, g2 ? "msm_accum_g2" : "msm_accum_g1", msm_accum<C>, (ntask + MSM_BLOCK - 1) / MSM_BLOCK,
             MSM_BLOCK, 0, points, S.entries, S.offset, S.xseg, S.info, S.perm, P, partial);
  KLAUNCH_ON(ctx, st, g2 ? "msm_heavy_g2" : "msm_heavy_g1", msm_heavy<C>, 1024, HEAVY_BLOCK,
             HEAVY_BLOCK * sizeof(typename C::Acc), S.heavy, S.info, S.offset, S.xoff, P, partial);
  KLAUNCH_ON(ctx, st, g2 ? "msm_heavy_g2" : "msm_heavy_g1", msm_heavy_small<C>, 256, MSM_BLOCK, 0, S.heavy, S.info,
             S.offset, S.xoff, P, partial);
  KLAUNCH_ON(ctx, st, g2 ? "msm_reduce1_g2" : "msm_reduce1_g1", msm_reduce1<C>,
             (uint32_t)((nchunks + MSM_BLOCK - 1) / MSM_BLOCK), MSM_BLOCK, 0, partial, S.offset, P.nbuckets, chunkR,
             chunkA);
  // reduction sets: the windows themselves, or <= 64 slices of 2048 chunks of the merged bucket set
  uint32_t nsets = P.nwin, log2ks = 0;
  if (P.tables) {
    uint32_t cps = nchunks < 2048 ? (uint32_t)nchunks : 2048u;
    nsets = (uint32_t)(nchunks / cps);
    for (uint32_t ks = cps * RED_CHUNK; ks > 1; ks >>= 1) ++log2ks;
  }
  auto* wtot = wsum + 65;
  constexpr int R2B = sizeof(typename C::Aff) == 64 ? 512 : 256;   // as wide as the register budget allows
  KLAUNCH_ON(ctx, st, g2 ? "msm_reduce2_g2" : "msm_reduce2_g1", (msm_reduce2<C, R2B>), nsets, R2B,
             R2B * sizeof(typename C::Acc), chunkR, chunkA, (uint32_t)(nchunks / nsets), wsum, wtot);
  if (P.tables)
    KLAUNCH_ON(ctx, st, g2 ? "msm_fold_g2" : "msm_fold_g1", msm_fold_merged<C>, 1, 64, 64 * sizeof(typename C::Acc),
               wsum, wtot, nsets, log2ks, d_out_aff, d_out_acc);
  else
    KLAUNCH_ON(ctx, st, g2 ? "msm_fold_g2" : "msm_fold_g1", msm_fold<C>, 1, 64, 0, wsum, nsets, P.c, d_out_aff,
               d_out_acc);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}

// one complete MSM on the context's main stream
template <class C>
static int32_t msm_device(g16_ctx* ctx, const void* d_scalars, uint32_t flags, const void* d_points, size_t n,
                          typename C::Aff* d_out_aff, typename C::Acc* d_out_acc, uint32_t table_c) {
  int32_t rc = msm_sort_device(ctx, ctx->stream, d_scalars, flags, n, table_c, ctx->sort[0]);
  if (rc) return rc;
  return msm_reduce_device<C>(ctx, ctx->stream, ctx->lane[0].acc, ctx->sort[0], d_points, d_out_aff, d_out_acc);
}

// sum of XYZZ partials -> affine (the `res += sync pending[k]` of msm.nim:117-119 across GPUs)
template <class C>
__global__ void sum_partials_kernel(const typename C::Acc* __restrict__ parts, uint32_t count,
                                    typename C::Aff* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  typename C::Acc r = C::acc_inf();
  for (uint32_t i = 0; i < count; ++i) C::add(r, parts[i]);
  *out = C::to_affine(r);
}

template <class C>
static int32_t sum_partials_device(g16_ctx* ctx, const void* d_parts, uint32_t count, void* d_out_aff) {
  KLAUNCH(ctx, "sum_partials", sum_partials_kernel<C>, 1, 64, 0, (const typename C::Acc*)d_parts, count,
          (typename C::Aff*)d_out_aff);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}

static uint32_t pick_table_window(size_t n) { return pick_window_cost(n ? n : 1, true, "G16_TABLE_WINDOW", 22); }

template <class C>
static int32_t precompute_device(g16_ctx* ctx, const void* d_points, size_t n, uint32_t c, void* d_tables) {
  const uint32_t nwin = FR_BITS / c + 1;
  KLAUNCH(ctx, "msm_precompute", msm_precompute<C>, (uint32_t)((n + MSM_BLOCK - 1) / MSM_BLOCK), MSM_BLOCK, 0,
          (const typename C::Aff*)d_points, (uint32_t)n, c, nwin, (typename C::Aff*)d_tables);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}

// gen: group generator in Montgomery affine form; d_table: 32*255 points (built here when !table_ready)
template <class C>
static int32_t fixed_base_device(g16_ctx* ctx, const typename C::Aff& gen, void* d_table, bool table_ready,
                                 const void* d_scalars, uint32_t mont, size_t n, void* d_out) {
  if (!table_ready)
    KLAUNCH(ctx, "fixed_base_table", fixed_base_table<C>, (32 * 255 + 255) / 256, 256, 0, gen,
            (typename C::Aff*)d_table);
  if (n)
    KLAUNCH(ctx, "fixed_base_mul", fixed_base_mul<C>, (uint32_t)((n + 255) / 256), 256, 0, (const u256*)d_scalars,
            mont, (uint32_t)n, (const typename C::Aff*)d_table, (typename C::Aff*)d_out);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}
