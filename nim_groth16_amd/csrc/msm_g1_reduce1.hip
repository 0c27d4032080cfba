// G1 split-bucket combine and first bucket-reduction stage
#include "msm_stage.cuh"
int32_t g16_st_heavy_g1(g16_ctx* ctx, hipStream_t st, const g16_ctx::MsmSort& S, void* partial) {
  return stage_heavy<G1>(ctx, st, S, partial);
}
int32_t g16_st_reduce1_g1(g16_ctx* ctx, hipStream_t st, const g16_ctx::MsmSort& S, const void* partial, void* chunkR,
                          void* chunkA) {
  return stage_reduce1<G1>(ctx, st, S, partial, chunkR, chunkA);
}
