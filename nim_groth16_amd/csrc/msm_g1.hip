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
uint32_t g16_pick_window_g1(size_t n) { return pick_window(n); }
