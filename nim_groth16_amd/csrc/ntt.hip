#include "g16_internal.hpp"
#include "ntt.cuh"

using namespace g16;

int32_t g16_ntt_device(g16_ctx* ctx, const void* d_src_, void* d_dst_, uint32_t log2n, int inverse) {
  const u256* d_src = (const u256*)d_src_;
  u256* d_dst = (u256*)d_dst_;
  const size_t n = size_t(1) << log2n;
  int32_t rc;
  // twiddle table  tw[i] = w^i (i < n/2), tw[n/2] = 1/n ; serves both directions (ntt.nim:64-69, 148-153)
  if (ctx->tw_log2n != log2n) {
    if ((rc = ensure(ctx, ctx->ntt_tw, (n / 2 + 1) * 32))) return rc;
    KLAUNCH(ctx, "ntt_twiddles", ntt_make_twiddles, (uint32_t)((n / 2 + 1 + 255) / 256), 256, 0,
            (u256*)ctx->ntt_tw.p, log2n);
    ctx->tw_log2n = log2n;
  }
  if (log2n == 0) {
    HIPCHK(ctx, hipMemcpyAsync(d_dst, d_src, 32, hipMemcpyDeviceToDevice, ctx->stream));
    return G16_OK;
  }
  const uint32_t npass = (log2n + 7) / 8;
  if ((rc = ensure(ctx, ctx->ntt_tmp, 2 * n * 32))) return rc;
  u256* tmpA = (u256*)ctx->ntt_tmp.p;
  u256* tmpB = tmpA + n;
  const u256* in = d_src;
  uint32_t log2s = 0;
  for (uint32_t p = 0; p < npass; ++p) {
    uint32_t rho = log2n / npass + (p < log2n % npass ? 1u : 0u);
    uint32_t log2b = 11 - rho;
    if (log2b > log2n - rho) log2b = log2n - rho;
    const bool last = p + 1 == npass;
    u256* out = last ? d_dst : ((p & 1) ? tmpB : tmpA);
    const uint32_t grid = 1u << (log2n - rho - log2b);
    const size_t shmem = (size_t(32) << (rho + log2b));
    KLAUNCH(ctx, inverse ? "ntt_pass_inv" : "ntt_pass_fwd", ntt_pass, grid, NTT_BLOCK, shmem, in,
            out, (const u256*)ctx->ntt_tw.p, log2n, log2s, rho, log2b, inverse, last ? 1 : 0);
    in = out;
    log2s += rho;
  }
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}

