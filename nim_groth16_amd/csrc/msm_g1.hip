#include "msm_impl.cuh"
int32_t g16_msm_device_g1(g16_ctx* ctx, const void* s, uint32_t f, const void* p, size_t n, void* aff, void* acc,
                          uint32_t table_c) {
  return msm_device<G1>(ctx, s, f, p, n, (g1_aff*)aff, (g1_acc*)acc, table_c);
}
int32_t g16_sum_partials_device_g1(g16_ctx* ctx, const void* parts, uint32_t count, void* out) {
  return sum_partials_device<G1>(ctx, parts, count, out);
}
int32_t g16_precompute_device_g1(g16_ctx* ctx, const void* d_points, size_t n, uint32_t c, void* d_tables) {
  return precompute_device<G1>(ctx, d_points, n, c, d_tables);
}
uint32_t g16_pick_window_g1(size_t n) { return pick_table_window(n); }
int32_t g16_fixed_base_device_g1(g16_ctx* ctx, void* d_table, bool ready, const void* d_s, uint32_t mont, size_t n,
                                 void* d_out) {
  // gen1 = (1, 2)  (curves.nim:112-113)
  g1_aff g{Fp::one(), Fp::dbl(Fp::one())};
  return fixed_base_device<G1>(ctx, g, d_table, ready, d_s, mont, n, d_out);
}
int32_t g16_msm_sort(g16_ctx* ctx, hipStream_t stream, const void* d_scalars, uint32_t flags, size_t n, uint32_t table_c,
                     g16_ctx::MsmSort& sort) {
  return msm_sort_device(ctx, stream, d_scalars, flags, n, table_c, sort);
}
int32_t g16_msm_reduce_g1(g16_ctx* ctx, hipStream_t stream, g16_ctx::Buf& acc, const g16_ctx::MsmSort& sort,
                          const void* d_points, void* d_out_aff, void* d_out_acc) {
  return msm_reduce_device<G1>(ctx, stream, acc, sort, d_points, (g1_aff*)d_out_aff, (g1_acc*)d_out_acc);
}
