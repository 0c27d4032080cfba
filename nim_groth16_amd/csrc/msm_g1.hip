#include "msm_impl.cuh"
int32_t g16_msm_device_g1(g16_ctx* ctx, const void* s, uint32_t f, const void* p, size_t n, void* aff, void* acc) {
  return msm_device<G1>(ctx, s, f, p, n, (g1_aff*)aff, (g1_acc*)acc, "g1");
}
int32_t g16_sum_partials_device_g1(g16_ctx* ctx, const void* parts, uint32_t count, void* out) {
  return sum_partials_device<G1>(ctx, parts, count, out);
}
