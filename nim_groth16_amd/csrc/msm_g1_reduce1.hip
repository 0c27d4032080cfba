// G1 split-bucket combine and first bucket-reduction stage
#include "msm_stage.cuh"
int32_t g16_st_heavy_g1(g16_ctx* ctx, hipStream_t st, const MsmParams& P, const void* batch, uint32_t ny) {
  return stage_heavy<G1>(ctx, st, P, batch, ny);
}
int32_t g16_st_reduce1_g1(g16_ctx* ctx, hipStream_t st, const MsmParams& P, const void* batch, uint32_t ny) {
  return stage_reduce1<G1>(ctx, st, P, batch, ny);
}
