// G2 second bucket-reduction stage, fold, and the sum of per-GPU partials
#include "msm_stage.cuh"
int32_t g16_st_reduce2_g2(g16_ctx* ctx, hipStream_t st, const MsmParams& P, bool narrow_tail, const void* batch,
                          uint32_t ny) {
  return stage_reduce2_fold<G2>(ctx, st, P, narrow_tail, batch, ny);
}
int32_t g16_sum_partials_device_g2(g16_ctx* ctx, const void* parts, uint32_t count, void* out) {
  return sum_partials_device<G2>(ctx, parts, count, out);
}
