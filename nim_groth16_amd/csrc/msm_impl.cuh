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

// window size: ~32 entries per bucket on average keeps bucket-accumulation (n*nwin madds) and bucket
// reduction (2*nbuckets adds) balanced; RED_CHUNK needs 2^(c-1) >= 16.
static uint32_t pick_window(size_t n) {
  int c = (int)floor_log2(n ? n : 1) - 4;
  if (c < 5) c = 5;
  if (c > 16) c = 16;
  const char* env = getenv("G16_MSM_WINDOW");
  if (env) {
    int v = atoi(env);
    if (v >= 5 && v <= 20) c = v;
  }
  return (uint32_t)c;
}

template <class C>
struct MsmLayout {
  size_t count, cursor, offset, xoff, heavy, info, tiles, entries, xseg, partial, chunkR, chunkA, wsum, total;
  MsmLayout(const MsmParams& P) {
    size_t o = 0;
    auto take = [&](size_t bytes) {
      size_t r = o;
      o += (bytes + 255) & ~size_t(255);
      return r;
    };
    size_t nb = P.nbuckets;
    count = take(nb * 4);
    cursor = take(nb * 4);
    offset = take((nb + 1) * 4);
    xoff = take(nb * 4);
    heavy = take(nb * 4);
    info = take(64);
    tiles = take(((nb + SCAN_TILE - 1) / SCAN_TILE) * 8);
    entries = take((size_t)P.n * P.nwin * 4);
    xseg = take((size_t)P.max_extra * 8);
    partial = take((nb + P.max_extra) * sizeof(typename C::Acc));
    size_t nchunks = nb / RED_CHUNK;
    chunkR = take(nchunks * sizeof(typename C::Acc));
    chunkA = take(nchunks * sizeof(typename C::Acc));
    wsum = take((size_t)(P.nwin + 1) * sizeof(typename C::Acc));
    total = o;
  }
};

// d_out_aff / d_out_acc: device pointers (either may be null).  table_c != 0: d_points holds the
// precomputed tables of a registered point set built with window size table_c.
template <class C>
static int32_t msm_device(g16_ctx* ctx, const void* d_scalars, uint32_t flags, const void* d_points, size_t n,
                          typename C::Aff* d_out_aff, typename C::Acc* d_out_acc, uint32_t table_c) {
  MsmParams P;
  P.n = (uint32_t)n;
  P.c = table_c ? table_c : pick_window(n);
  P.nwin = FR_BITS / P.c + 1;
  P.tables = table_c ? 1u : 0u;
  P.nbuckets = P.nwin << (P.c - 1);
  size_t avg = (n >> (P.c - 1)) + 1;
  P.seg = (uint32_t)(((2 * avg + 31) / 32) * 32);
  if (P.seg < 64) P.seg = 64;
  P.scalars_mont = (flags & G16_SCALARS_MONT) ? 1u : 0u;
  P.max_extra = (uint32_t)(((size_t)P.n * P.nwin) / P.seg + 1);
  MsmLayout<C> L(P);
  int32_t rc = ensure(ctx, ctx->ws, L.total);
  if (rc) return rc;
  char* ws = (char*)ctx->ws.p;
  auto* count = (uint32_t*)(ws + L.count);
  auto* cursor = (uint32_t*)(ws + L.cursor);
  auto* offset = (uint32_t*)(ws + L.offset);
  auto* xoff = (uint32_t*)(ws + L.xoff);
  auto* heavy = (uint32_t*)(ws + L.heavy);
  auto* info = (uint32_t*)(ws + L.info);
  auto* tiles = (uint2*)(ws + L.tiles);
  auto* entries = (uint32_t*)(ws + L.entries);
  auto* xseg = (uint2*)(ws + L.xseg);
  auto* partial = (typename C::Acc*)(ws + L.partial);
  auto* chunkR = (typename C::Acc*)(ws + L.chunkR);
  auto* chunkA = (typename C::Acc*)(ws + L.chunkA);
  auto* wsum = (typename C::Acc*)(ws + L.wsum);
  const auto* scalars = (const u256*)d_scalars;
  const auto* points = (const typename C::Aff*)d_points;

  // count + cursor are adjacent: one memset
  HIPCHK(ctx, hipMemsetAsync(ws + L.count, 0, L.offset - L.count, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(info, 0, 64, ctx->stream));
  const uint32_t nblk = (P.n + MSM_BLOCK - 1) / MSM_BLOCK;
  const uint32_t ntiles = (P.nbuckets + SCAN_TILE - 1) / SCAN_TILE;
  const bool g2 = sizeof(typename C::Aff) == 128;
  KLAUNCH(ctx, g2 ? "msm_count_g2" : "msm_count_g1", msm_count, nblk, MSM_BLOCK, 0, scalars, P, count);
  KLAUNCH(ctx, "msm_scan", scan_tile_sums, ntiles, SCAN_BLOCK, 0, count, P.nbuckets, P.seg, tiles);
  KLAUNCH(ctx, "msm_scan", scan_tiles, 1, SCAN_BLOCK, 0, tiles, ntiles, info);
  KLAUNCH(ctx, "msm_scan", scan_apply, ntiles, SCAN_BLOCK, 0, count, P.nbuckets, P.seg, tiles, offset, xoff, heavy,
          info);
  KLAUNCH(ctx, g2 ? "msm_scatter_g2" : "msm_scatter_g1", msm_scatter, nblk, MSM_BLOCK, 0, scalars, P, offset, cursor,
          entries);
  KLAUNCH(ctx, "msm_make_extra", msm_make_extra, 512, MSM_BLOCK, 0, heavy, info, offset, xoff, P.seg, P.max_extra,
          xseg);
  const uint32_t ntask = P.nbuckets + P.max_extra;
  KLAUNCH(ctx, g2 ? "msm_accum_g2" : "msm_accum_g1", msm_accum<C>, (ntask + MSM_BLOCK - 1) / MSM_BLOCK, MSM_BLOCK, 0,
          points, entries, offset, xseg, info, P, partial);
  KLAUNCH(ctx, g2 ? "msm_heavy_g2" : "msm_heavy_g1", msm_heavy<C>, 1024, HEAVY_BLOCK,
          HEAVY_BLOCK * sizeof(typename C::Acc), heavy, info, offset, xoff, P, partial);
  const uint32_t nchunks = P.nbuckets / RED_CHUNK;
  KLAUNCH(ctx, g2 ? "msm_reduce1_g2" : "msm_reduce1_g1", msm_reduce1<C>, (nchunks + MSM_BLOCK - 1) / MSM_BLOCK,
          MSM_BLOCK, 0, partial, offset, P.nbuckets, chunkR, chunkA);
  const uint32_t nsets = P.nwin;
  KLAUNCH(ctx, g2 ? "msm_reduce2_g2" : "msm_reduce2_g1", msm_reduce2<C>, nsets, RED2_BLOCK,
          RED2_BLOCK * sizeof(typename C::Acc), chunkR, chunkA, nchunks / nsets, wsum);
  KLAUNCH(ctx, g2 ? "msm_fold_g2" : "msm_fold_g1", msm_fold<C>, 1, 64, 0, wsum, nsets, P.tables ? 0u : P.c, d_out_aff, d_out_acc);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
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
