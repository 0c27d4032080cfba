// G2 registration-time tables, fixed-base multiples of gen2, on-curve check
#include "msm_stage.cuh"
static u256 std_fp(uint32_t a7, uint32_t a6, uint32_t a5, uint32_t a4, uint32_t a3, uint32_t a2, uint32_t a1, uint32_t a0) {
  u256 v;
  v.v[0] = a0; v.v[1] = a1; v.v[2] = a2; v.v[3] = a3; v.v[4] = a4; v.v[5] = a5; v.v[6] = a6; v.v[7] = a7;
  return Fp::to_mont(v);   // standard form -> Montgomery
}
int32_t g16_to29_device_g2(g16_ctx* ctx, hipStream_t st, const void* d_points, size_t n, void* d_out) {
  return to29_device<G2>(ctx, st, d_points, n, d_out);
}
int32_t g16_precompute_device_g2(g16_ctx* ctx, const void* d_points, size_t n, uint32_t c, uint32_t mtab,
                                 void* d_tables) {
  return precompute_device<G2>(ctx, d_points, n, c, mtab, d_tables);
}
int32_t g16_fixed_base_device_g2(g16_ctx* ctx, void* d_table, bool ready, const void* d_s, uint32_t mont, size_t n,
                                 void* d_out) {
  // gen2 (curves.nim:115-121), standard form -> Montgomery
  g2_aff g;
  g.x.c0 = std_fp(0x1adcd0edu, 0x10df9cb8u, 0x7040f466u, 0x55e3808fu, 0x98aa68a5u, 0x70acf5b0u, 0xbde23fabu, 0x1f149701u);
  g.x.c1 = std_fp(0x09e847e9u, 0xf05a6082u, 0xc3cd2a1du, 0x0a3a82e6u, 0xfbfbe620u, 0xf7f31269u, 0xfa15d21cu, 0x1c13b23bu);
  g.y.c0 = std_fp(0x056c0116u, 0x8a531946u, 0x1f7ca7aau, 0x19d4fcfdu, 0x1c7cdf52u, 0xdbfc4cbeu, 0xe6f91525u, 0x0b7f6fc8u);
  g.y.c1 = std_fp(0x0efe500au, 0x2d02dd77u, 0xf5f40132u, 0x9f30895du, 0xf553b878u, 0xfc3c0dadu, 0xaaa86456u, 0xa623235cu);
  return fixed_base_device<G2>(ctx, g, d_table, ready, d_s, mont, n, d_out);
}
int32_t g16_on_curve_device_g2(g16_ctx* ctx, const void* d_points, size_t n, uint32_t* d_first_bad) {
  // y^2 = x^3 + 3/(9+u): twistCoeffB (curves.nim:75-77)
  fp2_t b;
  b.c0 = std_fp(0x2b149d40u, 0xceb8aaaeu, 0x81be1899u, 0x1be06ac3u, 0xb5b4c5e5u, 0x59dbefa3u, 0x3267e6dcu, 0x24a138e5u);
  b.c1 = std_fp(0x009713b0u, 0x3af0fed4u, 0xcd2cafadu, 0xeed8fdf4u, 0xa74fa084u, 0xe52d1852u, 0xe4a2bd06u, 0x85c315d2u);
  return on_curve_device<G2>(ctx, d_points, n, b, d_first_bad);
}
int32_t g16_live_bitmap_device_g2(g16_ctx* ctx, const void* d_points, size_t n, uint32_t* d_bitmap, uint32_t* d_n_inf) {
  if (n)
    KLAUNCH(ctx, "points_live_bitmap", points_live_bitmap<G2>, (uint32_t)((n + 255) / 256), 256, 0,
            (const G2::Aff*)d_points, (uint32_t)n, d_bitmap, d_n_inf);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}
