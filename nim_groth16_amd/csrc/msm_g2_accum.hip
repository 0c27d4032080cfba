// G2 bucket-segment accumulation (the dominant kernel of the MSM)
#include "msm_stage.cuh"
int32_t g16_st_accum_g2(g16_ctx* ctx, hipStream_t st, const MsmParams& P, const void* batch, uint32_t ny) {
  return stage_accum<G2>(ctx, st, P, batch, ny);
}
