/* A plain-C caller of the C ABI (include/g16hip.h): no Python, no torch -- what a Nim/C/Go host links against.
 *
 *   gcc -O2 -Iinclude examples/c_abi_demo.c -Lnim_groth16_amd/csrc -lg16hip -Wl,-rpath,$PWD/nim_groth16_amd/csrc -o demo
 *
 * Checks, with nothing but the library itself:
 *   MSM:   sum s_i * (k_i G)  ==  (sum s_i k_i) G          (points made by g16_fixed_base_g1/g2, canonical scalars)
 *   NTT:   inverse(forward(x)) == x
 *   pairing: e(2 G1, 3 G2) == e(6 G1, G2)
 *   sparse product (the kernel of buildABC): a permutation matrix with Montgomery ones moves x's elements, empty rows give 0
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "g16hip.h"

#define CHECK(call)                                                                        \
  do {                                                                                     \
    int32_t rc__ = (call);                                                                 \
    if (rc__ != G16_OK) {                                                                  \
      fprintf(stderr, "%s -> %d (%s)\n", #call, rc__, ctx ? g16_last_error(ctx) : "?");    \
      return 1;                                                                            \
    }                                                                                      \
  } while (0)

static void small(uint8_t* fr, uint64_t v) { /* canonical little-endian Fr (the .wtns layout, G16_SCALARS_STD) */
  memset(fr, 0, 32);
  memcpy(fr, &v, 8);
}

int main(void) {
  g16_ctx* ctx = NULL;
  CHECK(g16_ctx_create(0, &ctx));
  CHECK(g16_selftest(ctx));

  enum { N = 1000 };
  static uint8_t ks[N * 32], ss[N * 32], g1[N * 64], g2[N * 128], e[32], exp1[64], got1[64], exp2[128], got2[128];
  uint64_t dot = 0;
  for (int i = 0; i < N; ++i) {
    uint64_t k = 3 + 7ull * i, s = 1000003ull * (i + 1) % 65521ull;
    small(ks + 32 * i, k);
    small(ss + 32 * i, s);
    dot += k * s;
  }
  small(e, dot);
  CHECK(g16_fixed_base_g1(ctx, ks, G16_SCALARS_STD, N, g1));
  CHECK(g16_fixed_base_g2(ctx, ks, G16_SCALARS_STD, N, g2));
  CHECK(g16_fixed_base_g1(ctx, e, G16_SCALARS_STD, 1, exp1));
  CHECK(g16_fixed_base_g2(ctx, e, G16_SCALARS_STD, 1, exp2));
  CHECK(g16_msm_g1(ctx, ss, G16_SCALARS_STD, g1, N, got1));
  CHECK(g16_msm_g2(ctx, ss, G16_SCALARS_STD, g2, N, got2));
  if (memcmp(exp1, got1, 64) || memcmp(exp2, got2, 128)) {
    fprintf(stderr, "MSM mismatch\n");
    return 1;
  }

  /* registered point set: tables once, many MSMs */
  g16_points* h = NULL;
  CHECK(g16_points_register_g1(ctx, g1, N, &h));
  CHECK(g16_msm_points(ctx, h, ss, G16_SCALARS_STD, got1));
  g16_points_release(h);
  if (memcmp(exp1, got1, 64)) {
    fprintf(stderr, "registered MSM mismatch\n");
    return 1;
  }

  static uint8_t x[256 * 32], y[256 * 32], z[256 * 32];
  for (int i = 0; i < 256 * 32; ++i) x[i] = (uint8_t)(i * 131 + 7);
  for (int i = 0; i < 256; ++i) x[32 * i + 31] &= 0x0f; /* < r: valid Montgomery residues */
  CHECK(g16_ntt_fr(ctx, x, y, 8, 0));
  CHECK(g16_ntt_fr(ctx, y, z, 8, 1));
  if (memcmp(x, z, sizeof x)) {
    fprintf(stderr, "NTT round trip mismatch\n");
    return 1;
  }

  static uint8_t c[3 * 32], p1[3 * 64], p2[3 * 128], gt[3 * G16_GT_BYTES], pa[2 * 64], pb[2 * 128];
  small(c, 2), small(c + 32, 6), small(c + 64, 1);
  CHECK(g16_fixed_base_g1(ctx, c, G16_SCALARS_STD, 3, p1)); /* 2G1, 6G1, G1 */
  small(c, 3), small(c + 32, 1), small(c + 64, 1);
  CHECK(g16_fixed_base_g2(ctx, c, G16_SCALARS_STD, 3, p2)); /* 3G2, G2, G2 */
  memcpy(pa, p1, 128);
  memcpy(pb, p2, 256);
  CHECK(g16_pairing(ctx, pa, pb, 2, gt));
  if (memcmp(gt, gt + G16_GT_BYTES, G16_GT_BYTES)) {
    fprintf(stderr, "pairing is not bilinear\n");
    return 1;
  }
  /* y = M x with g16_spmv_fr: M[i][(7 i + 3) mod 256] = 1 for i < 300 except the empty rows i = 0 mod 50;
   * 1 in Montgomery form = R mod r (frMontR, reference groth16/bn128/io.nim:91), little-endian */
  static const uint8_t one_mont[32] = {0xfb, 0xff, 0xff, 0x4f, 0x1c, 0x34, 0x96, 0xac, 0x29, 0xcd, 0x60, 0x9f, 0x95, 0x76, 0xfc, 0x36,
                                       0x2e, 0x46, 0x79, 0x78, 0x6f, 0xa3, 0x6e, 0x66, 0x2f, 0xdf, 0x07, 0x9a, 0xc1, 0x77, 0x0a, 0x0e};
  enum { ROWS = 300 };
  static uint32_t mrow[ROWS], mcol[ROWS];
  static uint8_t mval[ROWS * 32], yv[ROWS * 32];
  size_t nnz = 0;
  for (uint32_t i = 0; i < ROWS; ++i) {
    if (i % 50 == 0) continue;
    mrow[nnz] = i, mcol[nnz] = (7 * i + 3) % 256;
    memcpy(mval + 32 * nnz, one_mont, 32);
    ++nnz;
  }
  CHECK(g16_spmv_fr(ctx, mrow, mcol, mval, nnz, x, 256, ROWS, yv));
  for (uint32_t i = 0; i < ROWS; ++i) {
    static const uint8_t zero[32] = {0};
    const uint8_t* want = i % 50 == 0 ? zero : x + 32 * ((7 * i + 3) % 256);
    if (memcmp(yv + 32 * i, want, 32)) {
      fprintf(stderr, "sparse product mismatch in row %u\n", i);
      return 1;
    }
  }
  g16_ctx_destroy(ctx);
  printf("C ABI demo OK: MSM G1/G2 (n=%d), registered MSM, NTT round trip, pairing bilinearity, sparse product\n", N);
  return 0;
}
