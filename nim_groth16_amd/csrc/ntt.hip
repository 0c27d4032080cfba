// Launch sequences of the NTT passes and of the quotient pipeline built from them.
#include <atomic>
#include <mutex>

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

// dynamic LDS of a pass: tile + R/2 inner twiddles.  Up to 144 KB: above the 64 KB default, so the kernels are
// opted in once per process.
static size_t pass_shmem(uint32_t rho, uint32_t log2b) { return (size_t(32) << (rho + log2b)) + (size_t(16) << rho); }
static int32_t ntt_kernels_init(g16_ctx* ctx) {
  // per device (the attribute belongs to the device's copy of the code object).  Contexts of one device are used from
  // several host threads at once (the in-flight proofs of bench.py): the flag is published with release semantics
  // only after all four opt-ins succeeded, and the opt-ins themselves are serialised.
  static std::atomic<bool> done_on[64];
  static std::mutex mu;
  std::atomic<bool>& done = done_on[ctx->device & 63];
  if (done.load(std::memory_order_acquire)) return G16_OK;
  std::lock_guard<std::mutex> lock(mu);
  if (done.load(std::memory_order_relaxed)) return G16_OK;
  const int max_shmem = (int)pass_shmem(NTT_MAX_RHO, 2);
  static_assert((size_t(32) * NTT_TILE) + (size_t(16) << NTT_MAX_RHO) <= 160 * 1024, "tile + twiddles must fit the LDS");
  HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass<NTT_BLOCK>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, max_shmem));
  HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_last_pass_abc<NTT_BLOCK, NTT_TILE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, max_shmem));
  HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass<NTT_BLOCK_MID>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, max_shmem));
  HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_last_pass_abc<NTT_BLOCK_MID, NTT_TILE_MID>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, max_shmem));
  done.store(true, std::memory_order_release);
  return G16_OK;
}

// `batch` independent transforms: in + b*in_stride -> out + b*out_stride (strides in elements).
// `in` may equal `out` only for multi-pass sizes (log2n > NTT_MAX_RHO), where the first pass writes to a temporary.
// scale: optional per-output-index factor applied in the last pass (replaces 1/n).
// fuse_abc (forward, batch == 3 only): the last pass multiplies / subtracts the three transformed vectors on the fly
// and writes ONE vector to `out` (ntt_last_pass_abc); fuse_abc == 2 additionally multiplies by invZ1.
static int32_t ntt_batched(g16_ctx* ctx, const u256* in, size_t in_stride, u256* out, size_t out_stride,
                           uint32_t batch, uint32_t log2n, int inverse, const u256* scale, int fuse_abc = 0,
                           int c_from_ab = 0) {
  const size_t n = size_t(1) << log2n;
  int32_t rc;
  if ((rc = ensure_twiddles(ctx, log2n))) return rc;
  if ((rc = ntt_kernels_init(ctx))) return rc;
  if (log2n == 0 && !fuse_abc) {
    for (uint32_t b = 0; b < batch; ++b)
      HIPCHK(ctx, hipMemcpyAsync(out + b * out_stride, in + b * in_stride, 32, hipMemcpyDeviceToDevice, ctx->stream));
    return G16_OK;  // n = 1: 1/n = 1 and eta^0 = 1
  }
  // passes of <= 10 stages: one up to 2^10, two up to 2^20, three beyond (small geometry: <= 8 stages per pass)
  const bool small = g16_env().ntt_tile == NTT_TILE_SMALL, mid = g16_env().ntt_tile == NTT_TILE_MID;
  const uint32_t max_rho = small ? NTT_MAX_RHO_SMALL : NTT_MAX_RHO, log2tile = small ? 10 : mid ? 11 : 12;
  const uint32_t npass = log2n ? (log2n + max_rho - 1) / max_rho : 1;
  u256 *tmpA = nullptr, *tmpB = nullptr;
  if (npass > 1) {
    if ((rc = ensure(ctx, ctx->ntt_tmp, (npass > 2 ? 2 : 1) * n * 32 * batch))) return rc;
    tmpA = (u256*)ctx->ntt_tmp.p;
    tmpB = tmpA + n * batch;
  }
  const u256* src = in;
  size_t src_stride = in_stride;
  uint32_t log2s = 0;
  for (uint32_t p = 0; p < npass; ++p) {
    uint32_t rho = log2n / npass + (p < log2n % npass ? 1u : 0u);
    uint32_t log2b = log2tile - rho;
    if (log2b > log2n - rho) log2b = log2n - rho;
    const bool last = p + 1 == npass;
    const size_t shmem = pass_shmem(rho, log2b);
    const uint32_t ntiles = 1u << (log2n - rho - log2b);
    if (last && fuse_abc) {
      if (small)
        KLAUNCH(ctx, "ntt_last_pass_abc", (ntt_last_pass_abc<NTT_BLOCK_SMALL, NTT_TILE_SMALL>), ntiles, NTT_BLOCK_SMALL,
                shmem, src, out, (const u256*)ctx->ntt_tw.p, log2n, log2s, rho, log2b, src_stride, fuse_abc == 2 ? 1 : 0);
      else if (mid)
        KLAUNCH(ctx, "ntt_last_pass_abc", (ntt_last_pass_abc<NTT_BLOCK_MID, NTT_TILE_MID>), ntiles, NTT_BLOCK_MID,
                shmem, src, out, (const u256*)ctx->ntt_tw.p, log2n, log2s, rho, log2b, src_stride, fuse_abc == 2 ? 1 : 0);
      else
        KLAUNCH(ctx, "ntt_last_pass_abc", (ntt_last_pass_abc<NTT_BLOCK, NTT_TILE>), ntiles, NTT_BLOCK, shmem, src, out,
                (const u256*)ctx->ntt_tw.p, log2n, log2s, rho, log2b, src_stride, fuse_abc == 2 ? 1 : 0);
      break;
    }
    u256* dst = last ? out : ((p & 1) ? tmpB : tmpA);
    const size_t dst_stride = last ? out_stride : n;
    if (small)
      KLAUNCH(ctx, inverse ? "ntt_pass_inv" : "ntt_pass_fwd", ntt_pass<NTT_BLOCK_SMALL>, dim3(ntiles, batch),
              NTT_BLOCK_SMALL, shmem, src, dst, (const u256*)ctx->ntt_tw.p, log2n, log2s, rho, log2b, inverse,
              last ? 1 : 0, src_stride, dst_stride, last ? scale : (const u256*)nullptr, p == 0 ? c_from_ab : 0);
    else if (mid)
      KLAUNCH(ctx, inverse ? "ntt_pass_inv" : "ntt_pass_fwd", ntt_pass<NTT_BLOCK_MID>, dim3(ntiles, batch),
              NTT_BLOCK_MID, shmem, src, dst, (const u256*)ctx->ntt_tw.p, log2n, log2s, rho, log2b, inverse,
              last ? 1 : 0, src_stride, dst_stride, last ? scale : (const u256*)nullptr, p == 0 ? c_from_ab : 0);
    else
      KLAUNCH(ctx, inverse ? "ntt_pass_inv" : "ntt_pass_fwd", ntt_pass<NTT_BLOCK>, dim3(ntiles, batch), NTT_BLOCK, shmem,
              src, dst, (const u256*)ctx->ntt_tw.p, log2n, log2s, rho, log2b, inverse, last ? 1 : 0, src_stride,
              dst_stride, last ? scale : (const u256*)nullptr, p == 0 ? c_from_ab : 0);
    src = dst;
    src_stride = dst_stride;
    log2s += rho;
  }
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}

int32_t g16_ntt_device(g16_ctx* ctx, const void* d_src, void* d_dst, uint32_t log2n, int inverse) {
  const uint32_t one_pass = g16_env().ntt_tile == NTT_TILE_SMALL ? NTT_MAX_RHO_SMALL : NTT_MAX_RHO;
  if (d_src == d_dst && log2n <= one_pass && log2n > 0) {   // single pass: never in place
    int32_t rc = ensure(ctx, ctx->ntt_tmp, (size_t(32) << log2n));
    if (rc) return rc;
    if ((rc = ntt_batched(ctx, (const u256*)d_src, 0, (u256*)ctx->ntt_tmp.p, 0, 1, log2n, inverse, nullptr))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(d_dst, ctx->ntt_tmp.p, size_t(32) << log2n, hipMemcpyDeviceToDevice, ctx->stream));
    return G16_OK;
  }
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

// Quotient pipeline.  Snarkjs flavour at n = 2^20: FOUR launches, each reading or writing every vector once --
//   inverse pass 1 (A|B|C batched)  ->  inverse pass 2 with eta^i/n fused  ->  forward pass 1  ->  forward pass 2
//   with A1*B1 - C1 fused (one vector out).
// Az | Bz | Cz are read where they lie when they are contiguous (they are, in the per-proof buffer of
// g16_prove_partials): no staging copies.  HBM bytes: (6 + 6 + 6 + 4) * 32 n = 704 n.
// c_from_ab: d_c is not read -- Cz = Az * Bz is formed by the first pass while it loads (ntt.cuh, `mul_src`); needs the
// contiguous layout Az | Bz | (room for Cz) and log2n >= 1
int32_t g16_quotient_device(g16_ctx* ctx, const void* d_a, const void* d_b, const void* d_c, uint32_t log2n,
                            int flavour, void* d_out, int c_from_ab) {
  if (log2n > 27) {
    ctx->err = "quotient needs the 2n domain: log2n <= 27";
    return G16_EINVAL;
  }
  const size_t n = size_t(1) << log2n;
  int32_t rc;
  if ((rc = ensure(ctx, ctx->quot, 4 * n * 32))) return rc;
  if ((rc = ensure_coset(ctx, log2n, 0))) return rc;
  u256* X = (u256*)ctx->quot.p;   // 3n: coset coefficients  |  n: JensGroth intermediate
  u256* Y = X + 3 * n;
  const u256* in = (const u256*)d_a;
  if (c_from_ab && ((const u256*)d_b != in + n || (const u256*)d_c != in + 2 * n || log2n == 0)) {
    ctx->err = "quotient: Cz can only be formed on the fly from contiguous Az | Bz | Cz";
    return G16_EINVAL;
  }
  if ((const u256*)d_b != in + n || (const u256*)d_c != in + 2 * n) {   // scattered inputs (generic C-ABI callers)
    HIPCHK(ctx, hipMemcpyAsync(X, d_a, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(X + n, d_b, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(X + 2 * n, d_c, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
    in = X;
    if (log2n <= (g16_env().ntt_tile == NTT_TILE_SMALL ? NTT_MAX_RHO_SMALL : NTT_MAX_RHO)) {   // a single pass cannot run in place: stage behind the work area
      if ((rc = ensure(ctx, ctx->ntt_tmp, 3 * n * 32))) return rc;
      HIPCHK(ctx, hipMemcpyAsync(ctx->ntt_tmp.p, X, 3 * n * 32, hipMemcpyDeviceToDevice, ctx->stream));
      in = (const u256*)ctx->ntt_tmp.p;
    }
  }
  // shiftEvalDomain x3 (prover.nim:109-113, 167-169): iNTT with eta^i/n folded in, then forward NTT whose last pass
  // forms A1*B1 - C1 (prover.nim:175-176) [* invZ1, prover.nim:141]
  if ((rc = ntt_batched(ctx, in, n, X, n, 3, log2n, 1, (const u256*)ctx->coset[0].p, 0, c_from_ab))) return rc;
  if (flavour == 1) {  // Snarkjs
    if ((rc = ntt_batched(ctx, X, n, (u256*)d_out, 0, 3, log2n, 0, nullptr, 1))) return rc;
  } else {  // JensGroth: ... * invZ1, iNTT, * eta^-i   (prover.nim:141-143)
    if ((rc = ensure_coset(ctx, log2n, 1))) return rc;
    if ((rc = ntt_batched(ctx, X, n, Y, 0, 3, log2n, 0, nullptr, 2))) return rc;
    if ((rc = ntt_batched(ctx, Y, n, (u256*)d_out, n, 1, log2n, 1, (const u256*)ctx->coset[1].p))) return rc;
  }
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}

// shiftEvalDomain of ONE vector (prover.nim:109-113): values on H -> values on the coset eta*H.  The unit of the
// task-parallel quotient of a sharded proof: the reference runs the A, B, C pipelines as three Taskpool tasks
// (prover.nim:167-169); across GPUs each task lives on its own rank.  d_in is not modified; d_out != d_in.
int32_t g16_coset_pipeline_device(g16_ctx* ctx, const void* d_in, uint32_t log2n, void* d_out) {
  if (log2n > 27) {
    ctx->err = "quotient needs the 2n domain: log2n <= 27";
    return G16_EINVAL;
  }
  const size_t n = size_t(1) << log2n;
  int32_t rc;
  if ((rc = ensure(ctx, ctx->quot, 4 * n * 32))) return rc;
  if ((rc = ensure_coset(ctx, log2n, 0))) return rc;
  u256* X = (u256*)ctx->quot.p;
  if ((rc = ntt_batched(ctx, (const u256*)d_in, n, X, n, 1, log2n, 1, (const u256*)ctx->coset[0].p))) return rc;
  return ntt_batched(ctx, X, n, (u256*)d_out, n, 1, log2n, 0, nullptr);
}

int32_t g16_abc_pointwise_device(g16_ctx* ctx, const void* d_a, const void* d_b, const void* d_c, size_t count,
                                 void* d_out) {
  if (count)
    KLAUNCH(ctx, "fr_abc_pointwise", fr_abc_pointwise, (uint32_t)((count + 255) / 256), 256, 0, (const u256*)d_a,
            (const u256*)d_b, (const u256*)d_c, (u256*)d_out, (uint32_t)count);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}
