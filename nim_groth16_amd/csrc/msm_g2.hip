#include "msm_impl.cuh"
int32_t g16_msm_device_g2(g16_ctx* ctx, const void* s, uint32_t f, const void* p, size_t n, void* aff, void* acc,
                          uint32_t table_c) {
  return msm_device<G2>(ctx, s, f, p, n, (g2_aff*)aff, (g2_acc*)acc, table_c);
}
int32_t g16_sum_partials_device_g2(g16_ctx* ctx, const void* parts, uint32_t count, void* out) {
  return sum_partials_device<G2>(ctx, parts, count, out);
}
int32_t g16_precompute_device_g2(g16_ctx* ctx, const void* d_points, size_t n, uint32_t c, void* d_tables) {
  return precompute_device<G2>(ctx, d_points, n, c, d_tables);
}
uint32_t g16_pick_window_g2(size_t n) { return pick_table_window(n); }
int32_t g16_fixed_base_device_g2(g16_ctx* ctx, void* d_table, bool ready, const void* d_s, uint32_t mont, size_t n,
                                 void* d_out) {
  // gen2 (curves.nim:115-121), standard form -> Montgomery
  auto std_fp = [](uint32_t a7, uint32_t a6, uint32_t a5, uint32_t a4, uint32_t a3, uint32_t a2, uint32_t a1,
                   uint32_t a0) {
    u256 v;
    v.v[0] = a0; v.v[1] = a1; v.v[2] = a2; v.v[3] = a3; v.v[4] = a4; v.v[5] = a5; v.v[6] = a6; v.v[7] = a7;
    return Fp::to_mont(v);
  };
  g2_aff g;
  g.x.c0 = std_fp(0x1adcd0edu, 0x10df9cb8u, 0x7040f466u, 0x55e3808fu, 0x98aa68a5u, 0x70acf5b0u, 0xbde23fabu, 0x1f149701u);
  g.x.c1 = std_fp(0x09e847e9u, 0xf05a6082u, 0xc3cd2a1du, 0x0a3a82e6u, 0xfbfbe620u, 0xf7f31269u, 0xfa15d21cu, 0x1c13b23bu);
  g.y.c0 = std_fp(0x056c0116u, 0x8a531946u, 0x1f7ca7aau, 0x19d4fcfdu, 0x1c7cdf52u, 0xdbfc4cbeu, 0xe6f91525u, 0x0b7f6fc8u);
  g.y.c1 = std_fp(0x0efe500au, 0x2d02dd77u, 0xf5f40132u, 0x9f30895du, 0xf553b878u, 0xfc3c0dadu, 0xaaa86456u, 0xa623235cu);
  return fixed_base_device<G2>(ctx, g, d_table, ready, d_s, mont, n, d_out);
}
int32_t g16_msm_reduce_g2(g16_ctx* ctx, hipStream_t stream, g16_ctx::Buf& acc, const g16_ctx::MsmSort& sort,
                          const void* d_points, void* d_out_aff, void* d_out_acc) {
  return msm_reduce_device<G2>(ctx, stream, acc, sort, d_points, (g2_aff*)d_out_aff, (g2_acc*)d_out_acc);
}
