// The sort phase (shared by G1 and G2: it only looks at scalars), window choice, and the glue that strings the
// per-curve stages of an MSM together.
#include "msm_host.cuh"

uint32_t g16_pick_window_g1(size_t n) { return pick_table_window(n); }

int32_t g16_msm_sort(g16_ctx* ctx, hipStream_t stream, const void* d_scalars, uint32_t flags, size_t n, uint32_t table_c,
                     g16_ctx::MsmSort& sort, const uint32_t* d_live) {
  return msm_sort_device(ctx, stream, d_scalars, flags, n, table_c, sort, d_live);
}

// phase 2: accumulate + reduce one point set against a bucket arrangement; d_out_aff / d_out_acc: device
// pointers (either may be null).  acc_bytes = sizeof(XYZZ accumulator) of the group (128 for G1, 256 for G2).
static int32_t msm_reduce(g16_ctx* ctx, hipStream_t st, g16_ctx::Buf& acc, const g16_ctx::MsmSort& S, int group,
                          const void* d_points, void* d_out_aff, void* d_out_acc) {
  const MsmParams& P = S.P;
  const size_t asz = group == 1 ? 128 : 256;    // standard XYZZ (chunk sums and later)
  const size_t psz29 = group == 1 ? 144 : 288;  // reduced-radix XYZZ (bucket partials: accumulate -> heavy -> reduce1)
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t r = o;
    o += (bytes + 255) & ~size_t(255);
    return r;
  };
  const size_t nchunks = P.nbuckets / RED_CHUNK;
  const size_t o_partial = take(((size_t)P.nbuckets + P.max_extra) * psz29), o_chunkR = take(nchunks * asz),
               o_chunkA = take(nchunks * asz), o_wsum = take((size_t)(2 * 64 + 2) * asz);
  int32_t rc = ensure(ctx, acc, o);
  if (rc) return rc;
  char* ws = (char*)acc.p;
  void *partial = ws + o_partial, *chunkR = ws + o_chunkR, *chunkA = ws + o_chunkA, *wsum = ws + o_wsum;
  if (group == 1) {
    if ((rc = g16_st_accum_g1(ctx, st, S, d_points, partial))) return rc;
    if ((rc = g16_st_heavy_g1(ctx, st, S, partial))) return rc;
    if ((rc = g16_st_reduce1_g1(ctx, st, S, partial, chunkR, chunkA))) return rc;
    return g16_st_reduce2_g1(ctx, st, S, chunkR, chunkA, wsum, d_out_aff, d_out_acc);
  }
  if ((rc = g16_st_accum_g2(ctx, st, S, d_points, partial))) return rc;
  if ((rc = g16_st_heavy_g2(ctx, st, S, partial))) return rc;
  if ((rc = g16_st_reduce1_g2(ctx, st, S, partial, chunkR, chunkA))) return rc;
  return g16_st_reduce2_g2(ctx, st, S, chunkR, chunkA, wsum, d_out_aff, d_out_acc);
}
int32_t g16_msm_reduce_g1(g16_ctx* ctx, hipStream_t stream, g16_ctx::Buf& acc, const g16_ctx::MsmSort& sort,
                          const void* d_points, void* d_out_aff, void* d_out_acc) {
  return msm_reduce(ctx, stream, acc, sort, 1, d_points, d_out_aff, d_out_acc);
}
int32_t g16_msm_reduce_g2(g16_ctx* ctx, hipStream_t stream, g16_ctx::Buf& acc, const g16_ctx::MsmSort& sort,
                          const void* d_points, void* d_out_aff, void* d_out_acc) {
  return msm_reduce(ctx, stream, acc, sort, 2, d_points, d_out_aff, d_out_acc);
}
// one complete MSM on the context's main stream
static int32_t msm_device(g16_ctx* ctx, int group, const void* d_scalars, uint32_t flags, const void* d_points, size_t n,
                          void* d_out_aff, void* d_out_acc, uint32_t table_c, const uint32_t* d_live) {
  ctx->sort[0].narrow_tail = false;   // stand-alone MSM: nothing overlaps its tail, the short chain wins
  int32_t rc = msm_sort_device(ctx, ctx->stream, d_scalars, flags, n, table_c, ctx->sort[0], d_live);
  if (rc) return rc;
  if (table_c == 0 && n) {   // plain point array: the accumulate kernel reads reduced-radix entries
    const size_t esz = group == 1 ? 64 : 128;
    if ((rc = ensure(ctx, ctx->stage_p29, n * esz))) return rc;
    rc = group == 1 ? g16_to29_device_g1(ctx, ctx->stream, d_points, n, ctx->stage_p29.p)
                    : g16_to29_device_g2(ctx, ctx->stream, d_points, n, ctx->stage_p29.p);
    if (rc) return rc;
    d_points = ctx->stage_p29.p;
  }
  return msm_reduce(ctx, ctx->stream, ctx->lane[0].acc, ctx->sort[0], group, d_points, d_out_aff, d_out_acc);
}
int32_t g16_msm_device_g1(g16_ctx* ctx, const void* s, uint32_t f, const void* p, size_t n, void* aff, void* acc,
                          uint32_t table_c, const uint32_t* d_live) {
  return msm_device(ctx, 1, s, f, p, n, aff, acc, table_c, d_live);
}
int32_t g16_msm_device_g2(g16_ctx* ctx, const void* s, uint32_t f, const void* p, size_t n, void* aff, void* acc,
                          uint32_t table_c, const uint32_t* d_live) {
  return msm_device(ctx, 2, s, f, p, n, aff, acc, table_c, d_live);
}
