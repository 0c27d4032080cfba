// G1 bucket-segment accumulation (the dominant kernel of the MSM)
#include "msm_stage.cuh"
int32_t g16_st_accum_g1(g16_ctx* ctx, hipStream_t st, const g16_ctx::MsmSort& S, const void* points, void* partial) {
  return stage_accum<G1>(ctx, st, S, points, partial);
}
