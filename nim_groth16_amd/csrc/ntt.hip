// Launch sequences of the NTT passes and of the quotient pipeline built from them.
#include "g16_internal.hpp"
#include "ntt.cuh"

using namespace g16;

static int32_t ensure_twiddles(g16_ctx* ctx, uint32_t log2n) {
  // tw[i] = w^i (i < n/2), tw[n/2] = 1/n ; serves both directions (ntt.nim:64-69, 148-153)
  if (ctx->tw_log2n == log2n) return G16_OK;
  const size_t n = size_t(1) << log2n;
  int32_t rc;
  if ((rc = ensure(ctx, ctx->ntt_tw, (n / 2 + 1) * 32))) return rc;
  KLAUNCH(ctx, "ntt_twiddles", ntt_make_twiddles, (uint32_t)((n / 2 + 1 + 255) / 256), 256, 0, (u256*)ctx->ntt_tw.p,
          log2n);
  ctx->tw_log2n = log2n;
  return G16_OK;
}

// `batch` independent transforms: in + b*in_stride -> out + b*out_stride (strides in elements).
// in may equal out.  scale: optional per-output-index factor applied in the last pass (replaces 1/n).
static int32_t ntt_batched(g16_ctx* ctx, const u256* in, size_t in_stride, u256* out, size_t out_stride,
                           uint32_t batch, uint32_t log2n, int inverse, const u256* scale) {
  const size_t n = size_t(1) << log2n;
  int32_t rc;
  if ((rc = ensure_twiddles(ctx, log2n))) return rc;
  if (log2n == 0) {
    for (uint32_t b = 0; b < batch; ++b)
      HIPCHK(ctx, hipMemcpyAsync(out + b * out_stride, in + b * in_stride, 32, hipMemcpyDeviceToDevice, ctx->stream));
    return G16_OK;  // n = 1: 1/n = 1 and eta^0 = 1
  }
  const uint32_t npass = (log2n + 7) / 8;
  if ((rc = ensure(ctx, ctx->ntt_tmp, 2 * n * 32 * batch))) return rc;
  u256* tmpA = (u256*)ctx->ntt_tmp.p;
  u256* tmpB = tmpA + n * batch;
  const u256* src = in;
  size_t src_stride = in_stride;
  uint32_t log2s = 0;
  for (uint32_t p = 0; p < npass; ++p) {
    uint32_t rho = log2n / npass + (p < log2n % npass ? 1u : 0u);
    uint32_t log2b = 10 - rho;   // 1024-element (32 KB) tiles: 4 workgroups per CU
    if (log2b > log2n - rho) log2b = log2n - rho;
    const bool last = p + 1 == npass;
    u256* dst = last ? out : ((p & 1) ? tmpB : tmpA);
    const size_t dst_stride = last ? out_stride : n;
    const dim3 grid(1u << (log2n - rho - log2b), batch);
    const size_t shmem = (size_t(32) << (rho + log2b)) + (size_t(16) << rho);   // tile + R/2 twiddles
    KLAUNCH(ctx, inverse ? "ntt_pass_inv" : "ntt_pass_fwd", ntt_pass, grid, NTT_BLOCK, shmem, src, dst,
            (const u256*)ctx->ntt_tw.p, log2n, log2s, rho, log2b, inverse, last ? 1 : 0, src_stride, dst_stride,
            last ? scale : (const u256*)nullptr);
    src = dst;
    src_stride = dst_stride;
    log2s += rho;
  }
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}

int32_t g16_ntt_device(g16_ctx* ctx, const void* d_src, void* d_dst, uint32_t log2n, int inverse) {
  return ntt_batched(ctx, (const u256*)d_src, 0, (u256*)d_dst, 0, 1, log2n, inverse, nullptr);
}

static int32_t ensure_coset(g16_ctx* ctx, uint32_t log2n, int mode) {
  if (ctx->coset_log2n[mode] == log2n) return G16_OK;
  const size_t n = size_t(1) << log2n;
  int32_t rc;
  if ((rc = ensure(ctx, ctx->coset[mode], n * 32))) return rc;
  KLAUNCH(ctx, "ntt_coset_table", ntt_make_coset_table, (uint32_t)((n + 255) / 256), 256, 0,
          (u256*)ctx->coset[mode].p, log2n, mode);
  ctx->coset_log2n[mode] = log2n;
  return G16_OK;
}

int32_t g16_quotient_device(g16_ctx* ctx, const void* d_a, const void* d_b, const void* d_c, uint32_t log2n,
                            int flavour, void* d_out) {
  if (log2n > 27) {
    ctx->err = "quotient needs the 2n domain: log2n <= 27";
    return G16_EINVAL;
  }
  const size_t n = size_t(1) << log2n;
  int32_t rc;
  if ((rc = ensure(ctx, ctx->quot, 6 * n * 32))) return rc;
  if ((rc = ensure_coset(ctx, log2n, 0))) return rc;
  u256* X = (u256*)ctx->quot.p;
  u256* Y = X + 3 * n;
  HIPCHK(ctx, hipMemcpyAsync(X, d_a, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(X + n, d_b, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(X + 2 * n, d_c, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
  // shiftEvalDomain x3 (prover.nim:109-113, 167-169): iNTT with eta^i/n folded in, then forward NTT
  if ((rc = ntt_batched(ctx, X, n, Y, n, 3, log2n, 1, (const u256*)ctx->coset[0].p))) return rc;
  if ((rc = ntt_batched(ctx, Y, n, X, n, 3, log2n, 0, nullptr))) return rc;
  const uint32_t grid = (uint32_t)((n + 255) / 256);
  if (flavour == 1) {  // Snarkjs: ys = A1*B1 - C1   (prover.nim:175-176)
    KLAUNCH(ctx, "fr_abc_pointwise", fr_abc_pointwise, grid, 256, 0, X, X + n, X + 2 * n, (u256*)d_out, (uint32_t)n, 0,
            log2n);
  } else {  // JensGroth: (A1*B1 - C1) * invZ1, iNTT, * eta^-i   (prover.nim:141-143)
    if ((rc = ensure_coset(ctx, log2n, 1))) return rc;
    KLAUNCH(ctx, "fr_abc_pointwise", fr_abc_pointwise, grid, 256, 0, X, X + n, X + 2 * n, Y, (uint32_t)n, 1, log2n);
    if ((rc = ntt_batched(ctx, Y, n, (u256*)d_out, n, 1, log2n, 1, (const u256*)ctx->coset[1].p))) return rc;
  }
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}
