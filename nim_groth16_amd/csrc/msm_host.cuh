// Non-template host logic of the MSM pipeline: window choice, launch parameters and the sort phase (msm_sort.hip).
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
static uint32_t pick_window_cost(size_t n, bool merged, int forced, uint32_t cmax) {
  if (forced) return (uint32_t)forced;   // G16_MSM_WINDOW / G16_TABLE_WINDOW (g16_env: read once per process)
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
static uint32_t pick_window(size_t n) { return pick_window_cost(n ? n : 1, false, g16_env().msm_window, 16); }

// table_cfg: 0 for a plain point array, else the window bits of a registered set | its multiplier tables << 8
// (g16_points::cfg)
static MsmParams msm_params(size_t n, uint32_t flags, uint32_t table_cfg) {
  MsmParams P;
  const uint32_t table_c = table_cfg & 0xffu;
  P.n = (uint32_t)n;
  P.c = table_c ? table_c : pick_window(n);
  P.nwin = FR_BITS / P.c + 1;
  P.tables = table_c ? 1u : 0u;
  P.mtab = table_c && (table_cfg >> 8) == 2 ? 2u : 1u;
  P.nbuckets = P.tables ? msm_table_buckets(P.c, P.mtab) : (P.nwin << (P.c - 1));
  // segment length L: one accumulate task handles <= L entries.  A task is a serial chain of L mixed adds
  // (~23 us each with 4 waves per SIMD), so L also bounds the tail of the launch; ~1.25 x the mean bucket size
  // keeps most buckets in one segment, the rest get 1-2 short extra segments that msm_reduce1 absorbs.
  // (class bucket set: a bucket serves one or two digit values -- size the segment for the two-value buckets, or most
  // of them are split: 112 instead of 121 proofs/s, profiles/r04_ab_mtab_seg.txt)
  size_t avg = P.mtab == 2 ? ((size_t)n * P.nwin * 2) / (size_t(1) << (P.c - 1)) + 1 : ((size_t)n * P.nwin) / P.nbuckets + 1;
  P.seg = (uint32_t)(((avg + avg / 4 + 15) / 16) * 16);
  // few, long buckets (small windows / small point sets): cut them so that the launch still has ~64 k tasks --
  // a task is a serial chain, and 2^11 buckets of 1500 entries each would otherwise run as 2^11 threads
  const size_t cap = (((size_t)n * P.nwin / 65536 + 15) / 16) * 16;
  if (P.seg > cap) P.seg = (uint32_t)cap;
  if (P.seg < 32) P.seg = 32;
  if (g16_env().msm_seg) P.seg = (uint32_t)g16_env().msm_seg;
  P.scalars_mont = (flags & G16_SCALARS_MONT) ? 1u : 0u;
  P.max_extra = (uint32_t)(((size_t)P.n * P.nwin) / P.seg + 1);
  return P;
}

// ---- phase 1: scalars -> bucket arrangement (count, scan, scatter, extra-segment list) ---------------------
static int32_t msm_sort_device(g16_ctx* ctx, hipStream_t st, const void* d_scalars, uint32_t flags, size_t n,
                               uint32_t table_cfg, g16_ctx::MsmSort& S, const uint32_t* d_live = nullptr) {
  const MsmParams P = msm_params(n, flags, table_cfg);
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
  // partition sort (see msm.cuh): low bits <= BS_LOG (as many as divide the bucket count: the class set of a
  // registered set is 43 * 2^(c-7) buckets), partitions = nwin << hi_bits
  uint32_t lo_bits = P.c - 1 < (uint32_t)BS_LOG ? P.c - 1 : (uint32_t)BS_LOG;
  while (lo_bits && (P.nbuckets & ((1u << lo_bits) - 1))) --lo_bits;
  const uint32_t nparts = P.nbuckets >> lo_bits;
  const uint32_t ptiles = (P.n + PART_TILE - 1) / PART_TILE;
  const bool use_part = nparts <= PART_MAX && g16_env().msm_sort != 'a';
  const size_t nth = (size_t)nparts * ptiles;
  const size_t o_thist = take(use_part ? nth * 4 : 4), o_tmp = take(use_part ? (size_t)P.n * P.nwin * 8 : 8),
               o_tiles2 = take(((nth + SCAN_TILE - 1) / SCAN_TILE) * 8 + 8),
               o_shist = take(use_part ? (size_t)nparts * BS_SPLIT * BS_LOW * 4 : 4);
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
  // lo_bits == BS_LOG: bucket_place also produces xoff / heavy / the size histogram (see msm.cuh); count[] and
  // offset[] are fully written by it, so the partition path clears only the two small counter blocks
  const bool fused = use_part && lo_bits == (uint32_t)BS_LOG;
  if (!use_part) HIPCHK(ctx, hipMemsetAsync(ws + o_count, 0, o_offset - o_count, st));  // count + cursor are adjacent
  HIPCHK(ctx, hipMemsetAsync(S.info, 0, 64, st));
  HIPCHK(ctx, hipMemsetAsync(S.ghist, 0, PERM_BINS * 4, st));
  const uint32_t nblk = (P.n + MSM_BLOCK - 1) / MSM_BLOCK;
  const uint32_t ntiles = (P.nbuckets + SCAN_TILE - 1) / SCAN_TILE;
  if (use_part) {
    const uint32_t nt2 = (uint32_t)((nth + SCAN_TILE - 1) / SCAN_TILE);
    KLAUNCH_ON(ctx, st, "msm_part_count", part_pass<false>, ptiles, PART_BLOCK, 0, scalars, d_live, P, lo_bits, nparts, ptiles,
               S.tile_hist, S.tmp);
    KLAUNCH_ON(ctx, st, "msm_scan", scan1_tile_sums, nt2, SCAN_BLOCK, 0, S.tile_hist, (uint32_t)nth, S.tiles2);
    KLAUNCH_ON(ctx, st, "msm_scan", scan_tiles, 1, SCAN_BLOCK, 0, S.tiles2, nt2, S.info + 8);  // total -> info[8]
    KLAUNCH_ON(ctx, st, "msm_scan", scan1_apply, nt2, SCAN_BLOCK, 0, S.tile_hist, (uint32_t)nth, S.tiles2);
    KLAUNCH_ON(ctx, st, "msm_part_scatter", part_pass<true>, ptiles, PART_BLOCK, 0, scalars, d_live, P, lo_bits, nparts,
               ptiles, S.tile_hist, S.tmp);
    KLAUNCH_ON(ctx, st, "msm_bucket_sort", bucket_hist, nparts * BS_SPLIT, BS_LOW, 0, S.tmp, S.tile_hist, ptiles, nparts,
               S.info + 8, S.slice_hist);
    KLAUNCH_ON(ctx, st, "msm_bucket_sort", bucket_place, nparts * BS_SPLIT, BS_LOW, 0, S.tmp, S.tile_hist, ptiles, nparts,
               S.info + 8, S.slice_hist, P, lo_bits, S.count, S.offset, S.entries, fused ? 1u : 0u, S.xoff, S.heavy,
               S.info, S.ghist, S.blk_base);
  } else {
    KLAUNCH_ON(ctx, st, "msm_count", msm_count, nblk, MSM_BLOCK, 0, scalars, d_live, P, S.count);
  }
  const uint32_t pblk = (P.nbuckets + PERM_BLOCK - 1) / PERM_BLOCK;
  if (!fused) {
    KLAUNCH_ON(ctx, st, "msm_scan", scan_tile_sums, ntiles, SCAN_BLOCK, 0, S.count, P.nbuckets, P.seg, S.tiles);
    KLAUNCH_ON(ctx, st, "msm_scan", scan_tiles, 1, SCAN_BLOCK, 0, S.tiles, ntiles, S.info);
    KLAUNCH_ON(ctx, st, "msm_scan", scan_apply, ntiles, SCAN_BLOCK, 0, S.count, P.nbuckets, P.seg, S.tiles, S.offset,
               S.xoff, S.heavy, S.info);
    KLAUNCH_ON(ctx, st, "msm_perm", perm_hist, pblk, PERM_BLOCK, 0, S.count, P.nbuckets, S.ghist, S.blk_base);
  }
  KLAUNCH_ON(ctx, st, "msm_perm", perm_scatter, pblk, PERM_BLOCK, 0, S.count, P.nbuckets, S.ghist, S.blk_base,
             S.perm);
  if (!use_part)
    KLAUNCH_ON(ctx, st, "msm_scatter", msm_scatter, nblk, MSM_BLOCK, 0, scalars, d_live, P, S.offset, S.cursor, S.entries);
  KLAUNCH_ON(ctx, st, "msm_make_extra", msm_make_extra, 512, MSM_BLOCK, 0, S.heavy, S.info, S.offset, S.xoff, P.seg,
             P.max_extra, S.xseg);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}

static uint32_t pick_table_window(size_t n) { return pick_window_cost(n ? n : 1, true, g16_env().table_window, 22); }
// multiplier tables of a registered set with window c: the 43 slices of 2^(c-7) buckets of the class bucket set must
// be whole 256-bucket partitions of the sort
static uint32_t pick_table_mtab(uint32_t c) { return g16_env().mtab == 2 && c >= 15 ? 2u : 1u; }

