// G1 registration-time tables, fixed-base multiples of gen1, on-curve check
#include "msm_stage.cuh"
int32_t g16_to29_device_g1(g16_ctx* ctx, hipStream_t st, const void* d_points, size_t n, void* d_out) {
  return to29_device<G1>(ctx, st, d_points, n, d_out);
}
int32_t g16_precompute_device_g1(g16_ctx* ctx, const void* d_points, size_t n, uint32_t c, uint32_t mtab,
                                 void* d_tables) {
  return precompute_device<G1>(ctx, d_points, n, c, mtab, d_tables);
}
int32_t g16_fixed_base_device_g1(g16_ctx* ctx, void* d_table, bool ready, const void* d_s, uint32_t mont, size_t n,
                                 void* d_out) {
  // gen1 = (1, 2)  (curves.nim:112-113)
  g1_aff g{Fp::one(), Fp::dbl(Fp::one())};
  return fixed_base_device<G1>(ctx, g, d_table, ready, d_s, mont, n, d_out);
}
int32_t g16_on_curve_device_g1(g16_ctx* ctx, const void* d_points, size_t n, uint32_t* d_first_bad) {
  // y^2 = x^3 + 3  (curves.nim:54-67)
  const u256 b = Fp::add(Fp::dbl(Fp::one()), Fp::one());
  return on_curve_device<G1>(ctx, d_points, n, b, d_first_bad);
}
int32_t g16_live_bitmap_device_g1(g16_ctx* ctx, const void* d_points, size_t n, uint32_t* d_bitmap, uint32_t* d_n_inf) {
  if (n)
    KLAUNCH(ctx, "points_live_bitmap", points_live_bitmap<G1>, (uint32_t)((n + 255) / 256), 256, 0,
            (const G1::Aff*)d_points, (uint32_t)n, d_bitmap, d_n_inf);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}
int32_t g16_bitmap_or_device(g16_ctx* ctx, uint32_t* d_out, const uint32_t* d_a, const uint32_t* d_b, size_t n,
                             uint32_t* d_n_dead) {
  if (n)
    KLAUNCH(ctx, "bitmap_or", bitmap_or, (uint32_t)(((n + 31) / 32 + 255) / 256), 256, 0, d_out, d_a, d_b, (uint32_t)n,
            d_n_dead);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}
