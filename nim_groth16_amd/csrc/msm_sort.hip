// The sort phase (shared by G1 and G2: it only looks at scalars), window choice, and the glue that strings the
// per-curve stages of an MSM together.
#include "msm_host.cuh"

uint32_t g16_pick_window_g1(size_t n) { return pick_table_window(n); }
uint32_t g16_pick_mtab(uint32_t c) { return pick_table_mtab(c); }

int32_t g16_msm_sort(g16_ctx* ctx, hipStream_t stream, const void* d_scalars, uint32_t flags, size_t n, uint32_t table_c,
                     g16_ctx::MsmSort& sort, const uint32_t* d_live) {
  return msm_sort_device(ctx, stream, d_scalars, flags, n, table_c, sort, d_live);
}

// phase 2: accumulate + reduce point sets against their bucket arrangements, `n_accum` jobs per launch (see
// g16_internal.hpp).  Workspace of a job: bucket sums (reduced radix) | chunkR | chunkA | wsum.
template <class C>
static int32_t msm_batch(g16_ctx* ctx, hipStream_t st, const g16_msm_run* runs, int n_accum, int n_tail,
                         hipEvent_t after_heavy) {
  if (n_accum < 1 || n_accum > MSM_BATCH_MAX || n_tail < 0 || n_tail > n_accum) {
    ctx->err = "bad MSM batch";
    return G16_EINVAL;
  }
  const MsmParams& P = runs[0].sort->P;
  constexpr bool g2 = sizeof(typename C::Aff) == 128;
  constexpr size_t asz = sizeof(typename C::Acc);             // standard XYZZ (chunk sums and later): 128 / 256 B
  constexpr size_t psz29 = sizeof(typename Ec29<C>::Acc);     // reduced-radix XYZZ (bucket sums): 144 / 288 B
  static_assert(asz == (g2 ? 256 : 128) && psz29 == (g2 ? 288 : 144), "accumulator sizes");
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t r = o;
    o += (bytes + 255) & ~size_t(255);
    return r;
  };
  const size_t nchunks = P.nbuckets / msm_red_chunk(P);
  const size_t o_partial = take(((size_t)P.nbuckets + P.max_extra) * psz29), o_chunkR = take(nchunks * asz),
               o_chunkA = take(nchunks * asz), o_wsum = take((size_t)(2 * 64 + 2) * asz);
  MsmBatch<C> B;
  memset(&B, 0, sizeof B);
  for (int j = 0; j < n_accum; ++j) {
    const g16_ctx::MsmSort& S = *runs[j].sort;
    const MsmParams& Q = S.P;
    if (Q.n != P.n || Q.c != P.c || Q.nwin != P.nwin || Q.nbuckets != P.nbuckets || Q.seg != P.seg ||
        Q.tables != P.tables || Q.mtab != P.mtab || Q.max_extra != P.max_extra) {
      ctx->err = "MSM batch: the jobs do not share their launch parameters";
      return G16_EINVAL;
    }
    int32_t rc = ensure(ctx, *runs[j].acc, o);
    if (rc) return rc;
    char* ws = (char*)runs[j].acc->p;
    MsmJob<C>& J = B.job[j];
    J.points = (const typename Ec29<C>::Tab*)runs[j].d_points;
    J.entries = S.entries;
    J.offset = S.offset;
    J.xseg = S.xseg;
    J.info = S.info;
    J.perm = S.perm;
    J.heavy = S.heavy;
    J.xoff = S.xoff;
    J.partial = (typename Ec29<C>::Acc*)(ws + o_partial);
    J.init = (const typename Ec29<C>::Acc*)runs[j].init_partial;
    J.chunkR = (typename C::Acc*)(ws + o_chunkR);
    J.chunkA = (typename C::Acc*)(ws + o_chunkA);
    J.wsum = (typename C::Acc*)(ws + o_wsum);
    J.out_aff = (typename C::Aff*)runs[j].d_out_aff;
    J.out_acc = (typename C::Acc*)runs[j].d_out_acc;
  }
  // (the bucket sums come first in the workspace: g16_msm_partial_ptr)
  int32_t rc;
  if ((rc = g2 ? g16_st_accum_g2(ctx, st, P, &B, n_accum) : g16_st_accum_g1(ctx, st, P, &B, n_accum))) return rc;
  if ((rc = g2 ? g16_st_heavy_g2(ctx, st, P, &B, n_accum) : g16_st_heavy_g1(ctx, st, P, &B, n_accum))) return rc;
  if (after_heavy) HIPCHK(ctx, hipEventRecord(after_heavy, st));
  if (!n_tail) return G16_OK;
  if ((rc = g2 ? g16_st_reduce1_g2(ctx, st, P, &B, n_tail) : g16_st_reduce1_g1(ctx, st, P, &B, n_tail))) return rc;
  const bool narrow = runs[0].sort->narrow_tail;
  return g2 ? g16_st_reduce2_g2(ctx, st, P, narrow, &B, n_tail) : g16_st_reduce2_g1(ctx, st, P, narrow, &B, n_tail);
}
int32_t g16_msm_batch(g16_ctx* ctx, hipStream_t stream, int group, const g16_msm_run* runs, int n_accum, int n_tail,
                      hipEvent_t after_heavy) {
  return group == 1 ? msm_batch<G1>(ctx, stream, runs, n_accum, n_tail, after_heavy)
                    : msm_batch<G2>(ctx, stream, runs, n_accum, n_tail, after_heavy);
}
int32_t g16_msm_reduce_g1(g16_ctx* ctx, hipStream_t stream, g16_ctx::Buf& acc, const g16_ctx::MsmSort& sort,
                          const void* d_points, void* d_out_aff, void* d_out_acc) {
  const g16_msm_run run{&sort, &acc, d_points, d_out_aff, d_out_acc, nullptr};
  return msm_batch<G1>(ctx, stream, &run, 1, 1, nullptr);
}
int32_t g16_msm_reduce_g2(g16_ctx* ctx, hipStream_t stream, g16_ctx::Buf& acc, const g16_ctx::MsmSort& sort,
                          const void* d_points, void* d_out_aff, void* d_out_acc) {
  const g16_msm_run run{&sort, &acc, d_points, d_out_aff, d_out_acc, nullptr};
  return msm_batch<G2>(ctx, stream, &run, 1, 1, nullptr);
}
// one complete MSM on the context's main stream
static int32_t msm_device(g16_ctx* ctx, int group, const void* d_scalars, uint32_t flags, const void* d_points, size_t n,
                          void* d_out_aff, void* d_out_acc, uint32_t table_c, const uint32_t* d_live) {
  ctx->sort[0].narrow_tail = false;   // stand-alone MSM: nothing overlaps its tail, the short chain wins
  int32_t rc = msm_sort_device(ctx, ctx->stream, d_scalars, flags, n, table_c, ctx->sort[0], d_live);
  if (rc) return rc;
  if ((table_c & 0xffu) == 0 && n) {   // plain point array: the accumulate kernel reads reduced-radix entries
    const size_t esz = group == 1 ? 64 : 128;
    if ((rc = ensure(ctx, ctx->stage_p29, n * esz))) return rc;
    rc = group == 1 ? g16_to29_device_g1(ctx, ctx->stream, d_points, n, ctx->stage_p29.p)
                    : g16_to29_device_g2(ctx, ctx->stream, d_points, n, ctx->stage_p29.p);
    if (rc) return rc;
    d_points = ctx->stage_p29.p;
  }
  return group == 1 ? g16_msm_reduce_g1(ctx, ctx->stream, ctx->lane[0].acc, ctx->sort[0], d_points, d_out_aff, d_out_acc)
                    : g16_msm_reduce_g2(ctx, ctx->stream, ctx->lane[0].acc, ctx->sort[0], d_points, d_out_aff, d_out_acc);
}
int32_t g16_msm_device_g1(g16_ctx* ctx, const void* s, uint32_t f, const void* p, size_t n, void* aff, void* acc,
                          uint32_t table_c, const uint32_t* d_live) {
  return msm_device(ctx, 1, s, f, p, n, aff, acc, table_c, d_live);
}
int32_t g16_msm_device_g2(g16_ctx* ctx, const void* s, uint32_t f, const void* p, size_t n, void* aff, void* acc,
                          uint32_t table_c, const uint32_t* d_live) {
  return msm_device(ctx, 2, s, f, p, n, aff, acc, table_c, d_live);
}
