#include "msm_impl.cuh"
int32_t g16_msm_device_g2(g16_ctx* ctx, const void* s, uint32_t f, const void* p, size_t n, void* aff, void* acc) {
  return msm_device<G2>(ctx, s, f, p, n, (g2_aff*)aff, (g2_acc*)acc, "g2");
}
int32_t g16_sum_partials_device_g2(g16_ctx* ctx, const void* parts, uint32_t count, void* out) {
  return sum_partials_device<G2>(ctx, parts, count, out);
}
