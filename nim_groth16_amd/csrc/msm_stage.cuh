// Per-curve launch templates of the accumulate / reduce stages and of the auxiliary point kernels.  Each stage is
// instantiated in its own translation unit (msm_g1_*.hip, msm_g2_*.hip): the fully inlined G2 kernels take
// minutes to compile, and `make -j` builds the stages in parallel.
#pragma once
#include "g16_internal.hpp"
#include "msm.cuh"

using namespace g16;

// `batch`: MsmBatch<C> with `ny` jobs that share the launch parameters P (same n, same window: the G1 MSMs of a proof)
template <class C>
static int32_t stage_accum(g16_ctx* ctx, hipStream_t st, const MsmParams& P, const void* batch, uint32_t ny) {
  const bool g2 = sizeof(typename C::Aff) == 128;
  const uint32_t ntask = P.nbuckets + P.max_extra;
  KLAUNCH_ON(ctx, st, g2 ? "msm_accum_g2" : "msm_accum_g1", msm_accum<C>, dim3((ntask + ACC_BLOCK - 1) / ACC_BLOCK, ny),
             ACC_BLOCK, 0, *(const MsmBatch<C>*)batch, P, ctx->profiling ? ctx->clk_buf : (unsigned long long*)nullptr);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}

template <class C>
static int32_t stage_heavy(g16_ctx* ctx, hipStream_t st, const MsmParams& P, const void* batch, uint32_t ny) {
  const bool g2 = sizeof(typename C::Aff) == 128;
  // Grid size: the kernel loops grid-stride over the list of split buckets (all but empty for uniform or circom-like
  // scalars).  In the timeline of a proof this launch looks expensive (milliseconds, against 0.05 ms alone) because
  // its workgroups queue behind the accumulate waves of the other streams; shrinking the grid to 128 workgroups was
  // measured in round 2 (profiles/r02_ab_heavy_grid.txt): no change in proofs/s or latency -- the in-order reduce
  // behind it waits for the same slots -- and 30 % slower MSMs for scalars with thousands of split buckets
  // (tools/perf_skew.py "256 values": 3.92 -> 5.11 ms).
  const uint32_t hgrid = g16_env().heavy_grid ? (uint32_t)g16_env().heavy_grid : (ny > 1 ? 512u : 1024u);
  KLAUNCH_ON(ctx, st, g2 ? "msm_heavy_g2" : "msm_heavy_g1", msm_heavy<C>, dim3(hgrid, ny), heavy_block<C>(),
             heavy_block<C>() * sizeof(typename Ec29<C>::Acc), *(const MsmBatch<C>*)batch, P);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}

template <class C>
static int32_t stage_reduce1(g16_ctx* ctx, hipStream_t st, const MsmParams& P, const void* batch, uint32_t ny) {
  const bool g2 = sizeof(typename C::Aff) == 128;
  const uint32_t rc = msm_red_chunk(P);
  const size_t nchunks = P.nbuckets / rc;
  KLAUNCH_ON(ctx, st, g2 ? "msm_reduce1_g2" : "msm_reduce1_g1", msm_reduce1<C>,
             dim3((uint32_t)((nchunks + MSM_BLOCK - 1) / MSM_BLOCK), ny), MSM_BLOCK, 0, *(const MsmBatch<C>*)batch,
             P.nbuckets, rc);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}

// MsmJob::wsum: room for 2 * 64 + 2 accumulators
// narrow_tail: the caller overlaps this tail with other work (only read by the one-lane kernels, below)
template <class C>
static int32_t stage_reduce2_fold(g16_ctx* ctx, hipStream_t st, const MsmParams& P, bool narrow_tail, const void* batch,
                                  uint32_t ny) {
  const bool g2 = sizeof(typename C::Aff) == 128;
  const uint32_t rc = msm_red_chunk(P);
  const size_t nchunks = P.nbuckets / rc;
  const MsmBatch<C>& B = *(const MsmBatch<C>*)batch;
  // reduction sets: the windows themselves, or <= 64 slices of the merged bucket set.  reduce2 is a latency chain
  // whose length grows with the chunks per thread, so the slices are as small as the 64 lanes of msm_fold_merged
  // allow: 512 chunks (2^13 buckets) per slice at c = 20 -> 64 workgroups, one chunk per thread (G1) / two (G2).
  // G16_RED_SLICE = log2(chunks per slice) overrides it (experiments).
  uint32_t nsets = P.nwin, log2ks = 0;
  if (P.tables && P.mtab == 2) {   // class bucket set: 43 slices of 2^(c-7) buckets (msm_class_bucket)
    nsets = MSM_CLASS_SLICES;
    log2ks = P.c - 7;
  } else if (P.tables) {
    const uint32_t want = 1u << (g16_env().red_slice_log2 ? g16_env().red_slice_log2 : 9);
    uint32_t cps = nchunks < want ? (uint32_t)nchunks : want;
    while (nchunks / cps > 64) cps <<= 1;
    nsets = (uint32_t)(nchunks / cps);
    for (uint32_t ks = cps * rc; ks > 1; ks >>= 1) ++log2ks;
  }
  // reduce2 is a latency chain (serial chunk sums -> Hillis-Steele suffix scan -> tree).
  // Default: the quad-cooperative kernels (msm.cuh: an addition in 4 multiplications of wave time instead of 14;
  // tools/ubench_quad.hip: 2.1-2.4 x per operation), 64 slots per slice (128 for G1 slices of >= 512 chunks; G2 at 512
  // threads would have to live in 256 registers).  Shards: 2.71-2.94 -> 2.61-2.82 ms per rank at G = 8; stand-alone 2^20
  // MSM 2.18 -> 2.00 ms (G1), 5.21 -> 4.87 (G2); 2^20 proofs: single-proof latency 10.79 -> 10.59 and 10.84 -> 10.54 ms in
  // two sessions, proofs/s 121.57 -> 120.91 and 119.23 -> 120.63, i.e. inside the noise (profiles/r04_ab_tail_quad.txt,
  // r04_ab_g2first_2p20.txt, r04_perf_reg_quad.txt).
  // G16_TAIL_QUAD=0 or any G16_R2_WIDTH selects the one-lane-per-slot kernels of rounds 1-3.  There every scan step costs
  // one group addition on EVERY wave of the workgroup: wide workgroups (512 / 256 threads: one chunk per thread) have the
  // shortest chain; narrow ones (128 / 64 threads: four chunks per thread, work-efficient serial sums, a 7- / 6-step
  // scan) issue ~2.5x fewer wave-instructions for a ~20 % longer chain and were the choice inside proofs (same-box A/B,
  // profiles/r03_ab_knobs.txt: 111.6 -> 114.7 proofs/s, 11.67 -> 11.85 ms), wide for stand-alone MSMs and for 4-bucket
  // chunks.  G16_R2_WIDTH = 0 / 1 / 2 forces wide / narrow / a single wave per slice.
  constexpr bool is_g1 = sizeof(typename C::Aff) == 64;
  constexpr int R2B = is_g1 ? 512 : 256;
  constexpr int R2N = R2B / 4;
  const uint32_t cps = (uint32_t)(nchunks / nsets);
  const char* nm = g2 ? "msm_reduce2_g2" : "msm_reduce2_g1";
  const bool quad = g16_env().r2_width < 0 && g16_env().tail_quad != 0;
  if (quad) {
    if (is_g1 && cps >= 512)
      KLAUNCH_ON(ctx, st, nm, (msm_reduce2_quad<C, is_g1 ? 128 : 64>), dim3(nsets, ny), 512, 128 * sizeof(typename C::Acc), B,
                 cps, rc);
    else
      KLAUNCH_ON(ctx, st, nm, (msm_reduce2_quad<C, 64>), dim3(nsets, ny), 256, 64 * sizeof(typename C::Acc), B, cps, rc);
  } else {
    switch (g16_env().r2_width >= 0 ? g16_env().r2_width : (narrow_tail && rc > 4 ? 1 : 0)) {
      case 0:
        KLAUNCH_ON(ctx, st, nm, (msm_reduce2<C, R2B>), dim3(nsets, ny), R2B, R2B * sizeof(typename C::Acc), B, cps, rc);
        break;
      case 2:
        KLAUNCH_ON(ctx, st, nm, (msm_reduce2<C, 64>), dim3(nsets, ny), 64, 64 * sizeof(typename C::Acc), B, cps, rc);
        break;
      default:
        KLAUNCH_ON(ctx, st, nm, (msm_reduce2<C, R2N>), dim3(nsets, ny), R2N, R2N * sizeof(typename C::Acc), B, cps, rc);
    }
  }
  if (P.tables && P.mtab == 2 && quad)
    KLAUNCH_ON(ctx, st, g2 ? "msm_fold_g2" : "msm_fold_g1", msm_fold_classes_quad<C>, dim3(1, ny), 512,
               128 * sizeof(typename C::Acc), B, log2ks);
  else if (P.tables && P.mtab == 2)
    KLAUNCH_ON(ctx, st, g2 ? "msm_fold_g2" : "msm_fold_g1", msm_fold_classes<C>, dim3(1, ny), 128, 0, B, log2ks);
  else if (P.tables)
    KLAUNCH_ON(ctx, st, g2 ? "msm_fold_g2" : "msm_fold_g1", msm_fold_merged<C>, dim3(1, ny), 128, 0, B, nsets, log2ks);
  else
    KLAUNCH_ON(ctx, st, g2 ? "msm_fold_g2" : "msm_fold_g1", msm_fold<C>, dim3(1, ny), 64, 0, B, nsets, P.c);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}

// sum of XYZZ partials -> affine (the `res += sync pending[k]` of msm.nim:117-119 across GPUs)
template <class C>
__global__ void sum_partials_kernel(const typename C::Acc* __restrict__ parts, uint32_t count,
                                    typename C::Aff* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  typename C::Acc r = C::acc_inf();
  for (uint32_t i = 0; i < count; ++i) C::add(r, parts[i]);
  *out = C::to_affine(r);
}

template <class C>
static int32_t sum_partials_device(g16_ctx* ctx, const void* d_parts, uint32_t count, void* d_out_aff) {
  KLAUNCH(ctx, "sum_partials", sum_partials_kernel<C>, 1, 64, 0, (const typename C::Acc*)d_parts, count,
          (typename C::Aff*)d_out_aff);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}

template <class C>
static int32_t precompute_device(g16_ctx* ctx, const void* d_points, size_t n, uint32_t c, uint32_t mtab,
                                 void* d_tables) {
  const uint32_t nwin = FR_BITS / c + 1;
  KLAUNCH(ctx, "msm_precompute", msm_precompute<C>, (uint32_t)((n + MSM_BLOCK - 1) / MSM_BLOCK), MSM_BLOCK, 0,
          (const typename C::Aff*)d_points, (uint32_t)n, c, nwin, mtab, (typename Ec29<C>::Tab*)d_tables);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}

// reference-layout points -> reduced-radix entries (the one-shot MSM entry points, which have no tables)
template <class C>
static int32_t to29_device(g16_ctx* ctx, hipStream_t st, const void* d_points, size_t n, void* d_out) {
  if (n)
    KLAUNCH_ON(ctx, st, "points_to29", points_to29<C>, (uint32_t)((n + MSM_BLOCK - 1) / MSM_BLOCK), MSM_BLOCK, 0,
               (const typename C::Aff*)d_points, (uint32_t)n, (typename Ec29<C>::Tab*)d_out);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}

// gen: group generator in Montgomery affine form; d_table: 32*255 points (built here when !table_ready)
template <class C>
static int32_t fixed_base_device(g16_ctx* ctx, const typename C::Aff& gen, void* d_table, bool table_ready,
                                 const void* d_scalars, uint32_t mont, size_t n, void* d_out) {
  if (!table_ready)
    KLAUNCH(ctx, "fixed_base_table", fixed_base_table<C>, (32 * 255 + 255) / 256, 256, 0, gen,
            (typename C::Aff*)d_table);
  if (n)
    KLAUNCH(ctx, "fixed_base_mul", fixed_base_mul<C>, (uint32_t)((n + 255) / 256), 256, 0, (const u256*)d_scalars,
            mont, (uint32_t)n, (const typename C::Aff*)d_table, (typename C::Aff*)d_out);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}

// first_bad (device u32) must hold 0xffffffff on entry
template <class C>
static int32_t on_curve_device(g16_ctx* ctx, const void* d_points, size_t n, const typename C::E& b,
                               uint32_t* d_first_bad) {
  if (n)
    KLAUNCH(ctx, "points_on_curve", points_on_curve<C>, (uint32_t)((n + 255) / 256), 256, 0,
            (const typename C::Aff*)d_points, (uint32_t)n, b, d_first_bad);
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}
