// G1 second bucket-reduction stage, fold, and the sum of per-GPU partials
#include "msm_stage.cuh"
int32_t g16_st_reduce2_g1(g16_ctx* ctx, hipStream_t st, const MsmParams& P, bool narrow_tail, const void* batch,
                          uint32_t ny) {
  return stage_reduce2_fold<G1>(ctx, st, P, narrow_tail, batch, ny);
}
int32_t g16_sum_partials_device_g1(g16_ctx* ctx, const void* parts, uint32_t count, void* out) {
  return sum_partials_device<G1>(ctx, parts, count, out);
}
