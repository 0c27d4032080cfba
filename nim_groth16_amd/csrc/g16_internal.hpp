// Internal declarations shared by the translation units of libg16hip.so (not part of the C ABI).
#pragma once
#include "../../include/g16hip.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>


#include "msm_params.hpp"

// Experiment knobs (G16_* environment variables), read ONCE per process -- at the first g16_ctx_create -- and never
// again on the per-proof path.  0 / '\0' = not set.
struct G16Env {
  int msm_window = 0;      // G16_MSM_WINDOW   5..22: window bits of the one-shot MSMs
  int table_window = 0;    // G16_TABLE_WINDOW 5..22: window bits of registered point sets
  int msm_seg = 0;         // G16_MSM_SEG      8..4096: accumulate segment length
  char msm_sort = 0;       // G16_MSM_SORT     'a': atomic histogram/scatter instead of the partition sort
  int g1_lanes[3] = {3, 2, 0};   // G16_G1_LANES  lanes of the A1 / B1 / C1 MSMs (three digits from {0,2,3}).  C1 goes first,
                                 // on the sort's own lane: the H accumulation continues C1's bucket sums, so a late C1
                                 // delays the last chain of the proof (single-proof latency 11.2 -> 10.8 ms, same
                                 // throughput: profiles/r04_ab_g1_lanes.txt; rounds 1-3: 0, 2, 3)
  char stream_prio[7] = "lhllln";   // G16_STREAM_PRIO  six characters from {h, n, l}
  int red_slice_log2 = 0;  // G16_RED_SLICE    log2 of the chunks per reduce2 slice of a merged bucket set (8..11)
  int inf_compact_pct = 10;   // G16_INF_COMPACT  point sets with at least this percentage of (0,0) points get their own
                              // entry lists without them (0: always, 101: never); prover.hip
  int r2_width = -1;       // G16_R2_WIDTH  0: reduce2 with 512 / 256-thread workgroups, 1: 128 / 64, 2: 64 / 64; unset:
                           // narrow inside proofs, wide for stand-alone MSMs (msm_stage.cuh)
  int ntt_tile = 2048;            // G16_NTT_TILE = 1024 | 2048 | 4096: NTT workgroup geometry (ntt.cuh)
  // launch order of a proof (the defaults are the measured optimum: profiles/r05_ab_quotient_first*.txt):
  int quotient_first = 1;         // G16_QUOTIENT_FIRST=0: enqueue the witness MSMs before buildABC + quotient + sort(qs) (rounds 1-4)
  int lanes_after_quotient = 0;   // G16_LANES_AFTER_QUOTIENT=1 (with the above): the witness accumulations wait for them
  int g1_batch = 0;               // G16_G1_BATCH=1: ONE batched launch sequence (blockIdx.y = MSM) for A1, B1, C1 on one stream
                                  // instead of one stream and one sequence per G1 MSM (prover.hip; measured slower)
  int mtab = 2;                   // G16_MTAB=1: registered sets without the second multiplier table / class bucket set
  int chain_ch = 1;               // G16_CHAIN_CH=0: C1 and H1 as two MSMs instead of H1 continuing C1's bucket sums
  int tail_quad = 1;              // G16_TAIL_QUAD=0: reduce2 / fold with one lane per slot instead of a cooperating quad (msm.cuh,
                                  // msm_stage.cuh)
  int red_chunk = 0;              // G16_RED_CHUNK = 2 | 4 | 8 | 16: buckets per thread of msm_reduce1 (unset: msm_red_chunk)
  int cu_split = 0;               // G16_CU_SPLIT=k (1..24): main stream on k CUs per XCD, MSM lanes on the other 32 - k (g16hip.hip)
  int heavy_grid = 0;             // G16_HEAVY_GRID: workgroups of msm_heavy (unset: 1024 / 512; msm_stage.cuh)
  int cz_on_the_fly = 1;          // G16_CZ_FLY=0: buildABC writes Cz with a kernel of its own instead of the quotient's first
                                  // pass forming it while loading (ntt.cuh mul_src)
  int abc_dict = 1;               // G16_ABC_DICT=0: buildABC reads a 32-byte value per entry even when the key's coefficients
                                  // come from a small set (spmv.hip: value dictionary)
  int g2_first = -1;              // G16_G2_FIRST = 0 | 1 | 2: A1 and B1 (2: C1 too) accumulate after B2 (unset: 1 for small shards,
                                  // prover.hip)
};
const G16Env& g16_env();

// Buckets per thread of the first reduction stage (msm_reduce1).  16 keeps the chunk records (two accumulators per
// chunk) and reduce2's work small where the reduction is throughput: a 2^20 proof's 352 k buckets.  A small bucket set
// (a shard's point sets, a small MSM) is a latency chain on a mostly empty GPU: with 4 the chain of reduce1 is 8
// additions instead of 32 and reduce2, run wide, pays 2 more scan steps (a G2 addition is ~20 us of wave time).
inline uint32_t msm_red_chunk(const g16::MsmParams& P) {
  if (g16_env().red_chunk) return (uint32_t)g16_env().red_chunk;
  return P.nbuckets <= (1u << 17) ? 4u : 16u;
}

struct ProfEntry {
  const char* name;
  hipEvent_t e0, e1;
};

struct g16_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  // growable device buffers
  struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
  };
  // MSM workspaces.  A "sort" is the bucket arrangement of ONE scalar vector (shared by every MSM that
  // uses those scalars: the witness feeds A1, B1, B2 and C1, prover.nim:282-302); a "lane" is one
  // accumulate/reduce pipeline with its own stream so that independent MSMs overlap.
  struct MsmSort {
    Buf buf;
    g16::MsmParams P;
    bool narrow_tail = false;   // reduce2 geometry: see stage_reduce2_fold (set by the prover's lanes: throughput)
    uint32_t *count = nullptr, *cursor = nullptr, *offset = nullptr, *xoff = nullptr, *heavy = nullptr,
             *info = nullptr, *entries = nullptr, *perm = nullptr, *ghist = nullptr, *blk_base = nullptr, *tile_hist = nullptr;
    uint2* tmp = nullptr;
    uint32_t* slice_hist = nullptr;
    uint2* tiles2 = nullptr;
    uint2* tiles = nullptr;
    uint2* xseg = nullptr;
  };
  struct MsmLane {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    Buf acc;
  };
  MsmSort sort[4];   // 0: witness (all pairs)  1: H scalars  2: witness, A1's live pairs  3: witness, B1/B2's live pairs
  MsmLane lane[5];
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_q = nullptr, ev_b2 = nullptr, ev_c = nullptr, ev_g2 = nullptr;
  Buf stage_s;   // staged scalars (host-pointer API)
  Buf stage_p;   // staged points
  Buf stage_p29; // the same points as reduced-radix entries (one-shot MSMs; registered sets keep their own tables)
  Buf stage_o;   // result slot
  Buf ntt_tw;    // twiddle table
  Buf ntt_tmp;   // ping-pong buffers
  Buf coset[2];  // eta^(+-i)/n tables (ntt_make_coset_table)
  Buf quot;      // 6n work area of the quotient pipeline
  Buf prove;     // per-proof scalars: witness, Az|Bz|Cz, qs
  Buf fb_table[2];  // fixed-base tables of gen1 / gen2
  bool fb_ready[2] = {false, false};
  unsigned long long* clk_buf = nullptr;   // {sum d_memtime, sum d_memrealtime} of the accumulate kernels (g16_profile_clock)
  const void* shard_begun = nullptr;   // key of a g16_prove_partials_begin that still awaits its _end
  uint32_t tw_log2n = 0xffffffffu;
  uint32_t coset_log2n[2] = {0xffffffffu, 0xffffffffu};
  // profiling
  bool profiling = false;
  bool prof_accum_only = false;   // g16_profile_enable(ctx, 2): only the bucket-accumulation kernels
  std::vector<ProfEntry> prof;
  std::vector<hipEvent_t> free_events;
};

#define HIPCHK(ctx, call)                                                                      \
  do {                                                                                         \
    hipError_t e__ = (call);                                                                   \
    if (e__ != hipSuccess) {                                                                   \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                         \
      return e__ == hipErrorOutOfMemory ? G16_ENOMEM : G16_EHIP;                               \
    }                                                                                          \
  } while (0)

// waits for everything this context has queued: the main stream AND the five MSM lane streams.  Called before a
// workspace buffer is freed or regrown, on every error exit of a multi-stream launch sequence and at teardown, so
// that no lane kernel can still be reading a buffer that is about to go away.
inline void ctx_quiesce(g16_ctx* ctx) {
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (auto& l : ctx->lane)
    if (l.stream) (void)hipStreamSynchronize(l.stream);
}

// Entry protocol of every context-taking C-ABI function.
//  * The calling thread's current HIP device becomes the context's: allocations, event creation and launches bind to
//    the CURRENT device, and a host thread starts on device 0 -- a context of GPU 3 used from a fresh worker thread
//    would otherwise allocate its workspaces on GPU 0.
//  * A g16_prove_partials_begin whose _end never came (the caller's exchange failed, or it simply moved on) still has
//    four witness lanes reading the sort and per-proof buffers: any other compute call on the context first drains
//    them and cancels the pending proof (its _end then returns G16_EINVAL).  `keep_shard`: calls that do not touch
//    the workspaces (_end itself, synchronize, the profiling getters).
inline int32_t ctx_enter(g16_ctx* ctx, bool keep_shard = false) {
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (ctx->shard_begun && !keep_shard) {
    ctx_quiesce(ctx);
    ctx->shard_begun = nullptr;
  }
  return G16_OK;
}
#define CTX_ENTER(ctx)                              \
  do {                                              \
    if (int32_t rc__ = ctx_enter(ctx)) return rc__; \
  } while (0)
#define CTX_ENTER_KEEP(ctx)                               \
  do {                                                    \
    if (int32_t rc__ = ctx_enter(ctx, true)) return rc__; \
  } while (0)

inline int32_t ensure(g16_ctx* ctx, g16_ctx::Buf& b, size_t bytes) {
  if (b.bytes >= bytes) return G16_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));   // hipMalloc binds to the calling thread's current device
  if (b.p) {
    ctx_quiesce(ctx);
    HIPCHK(ctx, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
  }
  size_t want = bytes + bytes / 8 + 4096;
  HIPCHK(ctx, hipMalloc(&b.p, want));
  b.bytes = want;
  return G16_OK;
}

// ---- profiling helpers ----------------------------------------------------------------------------
struct ProfScope {
  g16_ctx* ctx;
  bool on;
  ProfEntry e;
  hipStream_t st;
  ProfScope(g16_ctx* c, const char* name, hipStream_t stream = nullptr)
      : ctx(c), on(c->profiling && (!c->prof_accum_only || strncmp(name, "msm_accum", 9) == 0)),
        st(stream ? stream : c->stream) {
    if (!on) return;
    e.name = name;
    auto get = [&](hipEvent_t& ev) {
      if (!ctx->free_events.empty()) {
        ev = ctx->free_events.back();
        ctx->free_events.pop_back();
      } else {
        (void)hipEventCreate(&ev);
      }
    };
    get(e.e0);
    get(e.e1);
    (void)hipEventRecord(e.e0, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(e.e1, st);
    ctx->prof.push_back(e);
  }
};
#define KLAUNCH_ON(ctx, stream_, name, kernel, grid, block, shmem, ...)                        \
  do {                                                                                         \
    ProfScope ps__(ctx, name, stream_);                                                        \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), shmem, stream_, __VA_ARGS__);          \
  } while (0)
#define KLAUNCH(ctx, name, kernel, grid, block, shmem, ...) \
  KLAUNCH_ON(ctx, (ctx)->stream, name, kernel, grid, block, shmem, __VA_ARGS__)


// implemented in msm_g1.hip / msm_g2.hip / ntt.hip
// table_c == 0: d_points = n affine points; else g16_points::cfg() and d_points = the tables of that registered set
int32_t g16_msm_device_g1(g16_ctx* ctx, const void* d_scalars, uint32_t flags, const void* d_points, size_t n,
                          void* d_out_aff, void* d_out_acc, uint32_t table_c, const uint32_t* d_live = nullptr);
int32_t g16_msm_device_g2(g16_ctx* ctx, const void* d_scalars, uint32_t flags, const void* d_points, size_t n,
                          void* d_out_aff, void* d_out_acc, uint32_t table_c, const uint32_t* d_live = nullptr);
// the live bitmap of a registered set if it holds enough (0,0) points for its own entry lists to pay (G16_INF_COMPACT)
struct g16_points;
const uint32_t* g16_points_live_if_sparse(const g16_points* p);
// the two halves of an MSM: (1) arrange one scalar vector into buckets, (2) accumulate + reduce a point set
// against that arrangement.  Several point sets may share one sort (same scalars, same n, same c).
// d_live (optional): bitmap over the n pairs; pairs with a cleared bit get no entries (point sets with (0,0) points)
int32_t g16_msm_sort(g16_ctx* ctx, hipStream_t stream, const void* d_scalars, uint32_t flags, size_t n,
                     uint32_t table_c, g16_ctx::MsmSort& sort, const uint32_t* d_live = nullptr);
// Several MSMs of one group as ONE launch sequence on one stream (every stage kernel takes blockIdx.y = job): the jobs
// must share the launch parameters (same n, same window: e.g. the G1 MSMs of a proof that consume the witness).
//   n_accum jobs are accumulated (+ split buckets combined); the first n_tail <= n_accum of them are reduced and folded
//   init_partial: the bucket sums (after `after_heavy`) of another job over the same bucket set, to continue from
//   after_heavy (optional): recorded on the stream once every job's bucket sums are final
struct g16_msm_run {
  const g16_ctx::MsmSort* sort;
  g16_ctx::Buf* acc;            // workspace of this job (bucket sums first: see g16_msm_partial_ptr)
  const void* d_points;
  void* d_out_aff;              // either may be null
  void* d_out_acc;
  const void* init_partial;
};
int32_t g16_msm_batch(g16_ctx* ctx, hipStream_t stream, int group, const g16_msm_run* runs, int n_accum, int n_tail,
                      hipEvent_t after_heavy);
inline const void* g16_msm_partial_ptr(const g16_ctx::Buf& acc) { return acc.p; }
int32_t g16_msm_reduce_g1(g16_ctx* ctx, hipStream_t stream, g16_ctx::Buf& acc, const g16_ctx::MsmSort& sort,
                          const void* d_points, void* d_out_aff, void* d_out_acc);
int32_t g16_msm_reduce_g2(g16_ctx* ctx, hipStream_t stream, g16_ctx::Buf& acc, const g16_ctx::MsmSort& sort,
                          const void* d_points, void* d_out_aff, void* d_out_acc);
int32_t g16_lanes_init(g16_ctx* ctx);
int g16_stream_priority(int index);
// per-curve stages of phase 2, each compiled in its own translation unit (msm_g{1,2}_{accum,reduce1,reduce2}.hip).
// `batch`: a g16::MsmBatch<G1 / G2> (msm.cuh) of `ny` jobs that share the launch parameters P.
#define G16_DECL_STAGES(g)                                                                                            \
  int32_t g16_st_accum_##g(g16_ctx*, hipStream_t, const g16::MsmParams& P, const void* batch, uint32_t ny);           \
  int32_t g16_st_heavy_##g(g16_ctx*, hipStream_t, const g16::MsmParams& P, const void* batch, uint32_t ny);           \
  int32_t g16_st_reduce1_##g(g16_ctx*, hipStream_t, const g16::MsmParams& P, const void* batch, uint32_t ny);         \
  int32_t g16_st_reduce2_##g(g16_ctx*, hipStream_t, const g16::MsmParams& P, bool narrow_tail, const void* batch,     \
                             uint32_t ny);
G16_DECL_STAGES(g1)
G16_DECL_STAGES(g2)
int32_t g16_to29_device_g1(g16_ctx* ctx, hipStream_t st, const void* d_points, size_t n, void* d_out);
int32_t g16_to29_device_g2(g16_ctx* ctx, hipStream_t st, const void* d_points, size_t n, void* d_out);
int32_t g16_precompute_device_g1(g16_ctx* ctx, const void* d_points, size_t n, uint32_t c, uint32_t mtab, void* d_tables);
int32_t g16_precompute_device_g2(g16_ctx* ctx, const void* d_points, size_t n, uint32_t c, uint32_t mtab, void* d_tables);
uint32_t g16_pick_window_g1(size_t n);
uint32_t g16_pick_mtab(uint32_t c);
int32_t g16_on_curve_device_g1(g16_ctx* ctx, const void* d_points, size_t n, uint32_t* d_first_bad);
int32_t g16_on_curve_device_g2(g16_ctx* ctx, const void* d_points, size_t n, uint32_t* d_first_bad);
int32_t g16_fixed_base_device_g1(g16_ctx* ctx, void* d_table, bool ready, const void* d_s, uint32_t mont, size_t n,
                                 void* d_out);
int32_t g16_fixed_base_device_g2(g16_ctx* ctx, void* d_table, bool ready, const void* d_s, uint32_t mont, size_t n,
                                 void* d_out);

// device-resident point set with precomputed window tables.  Immutable after registration and tied to a DEVICE,
// not to the context that created it: every context of that device may run MSMs against it concurrently (the
// in-flight proofs of one GPU share one key), and it may be released before or after any context.
struct g16_points {
  int device = 0;
  int group = 1;          // 1: G1 (64-byte points), 2: G2 (128-byte points)
  size_t n = 0;
  uint32_t c = 0, nwin = 0;
  uint32_t mtab = 1;         // multiplier tables per window: 1, or 2 = {1, 2} with the class bucket set (msm.cuh)
  uint32_t cfg() const { return c | (mtab << 8); }   // the `table_cfg` of g16_msm_sort / g16_msm_device_*
  void* d_tables = nullptr;  // mtab * nwin * n affine points: [m][w][i] = 2^(c w + m) P_i
  uint32_t* d_live = nullptr;   // bitmap: bit i set <=> point i is not (0,0); ceil(n/32) words
  size_t n_inf = 0;             // points at infinity in the set
};
// *d_n_inf (device u32, zeroed by the caller) += number of (0,0) points; bitmap: ceil(n/32) words
int32_t g16_live_bitmap_device_g1(g16_ctx* ctx, const void* d_points, size_t n, uint32_t* d_bitmap, uint32_t* d_n_inf);
int32_t g16_live_bitmap_device_g2(g16_ctx* ctx, const void* d_points, size_t n, uint32_t* d_bitmap, uint32_t* d_n_inf);
int32_t g16_bitmap_or_device(g16_ctx* ctx, uint32_t* d_out, const uint32_t* d_a, const uint32_t* d_b, size_t n,
                             uint32_t* d_n_dead);
int32_t g16_sum_partials_device_g1(g16_ctx* ctx, const void* d_parts, uint32_t count, void* d_out_aff);
int32_t g16_sum_partials_device_g2(g16_ctx* ctx, const void* d_parts, uint32_t count, void* d_out_aff);
// row-binned sparse matrices over Fr (spmv.hip): nmat = 2 -> the A and B matrices of a key, apply = buildABC
struct g16_spmat;
int32_t g16_spmat_create(g16_ctx* ctx, uint32_t nmat, uint32_t nrows, size_t nnz, const uint32_t* vrow,
                         size_t vrow_stride, const uint32_t* col, size_t col_stride, const void* val_base,
                         size_t val_stride, g16_spmat** out, bool values_r2 = false);
void g16_spmat_destroy(g16_spmat* m);
void g16_spmat_info(const g16_spmat* m, size_t out[10]);   // dictionary size (0: plain values), virtual rows per bin
// nmat == 2: d_out = Az | Bz | Cz; need_cz = false: Cz may be left unwritten (the quotient forms it on the fly)
int32_t g16_spmat_apply(g16_ctx* ctx, const g16_spmat* m, const void* d_x, uint32_t x_mont, void* d_out,
                        bool need_cz = true);
int32_t g16_ntt_device(g16_ctx* ctx, const void* d_src, void* d_dst, uint32_t log2n, int inverse);
// computeSnarkjsScalarCoeffs (flavour 1, prover.nim:158-181) / computeQuotientPointwise (flavour 0, :118-148)
// d_a, d_b, d_c, d_out: n elements each (device); inputs are not modified
int32_t g16_quotient_device(g16_ctx* ctx, const void* d_a, const void* d_b, const void* d_c, uint32_t log2n,
                            int flavour, void* d_out, int c_from_ab = 0);
// one coset pipeline (shiftEvalDomain, prover.nim:109-113) and the pointwise step on separately held slices: the
// pieces of the task-parallel quotient of a sharded proof
int32_t g16_coset_pipeline_device(g16_ctx* ctx, const void* d_in, uint32_t log2n, void* d_out);
int32_t g16_abc_pointwise_device(g16_ctx* ctx, const void* d_a, const void* d_b, const void* d_c, size_t count,
                                 void* d_out);
