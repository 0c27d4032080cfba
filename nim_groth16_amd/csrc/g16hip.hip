// Host side of the C ABI (include/g16hip.h): context, HBM workspace, launch sequences.
// No CPU compute path exists here: if HIP is unavailable every call returns G16_ENODEV.
#include "g16_internal.hpp"
#include "ec.cuh"

using namespace g16;

// ---- environment knobs: parsed once ----------------------------------------------------------------
static G16Env read_env() {
  G16Env e;
  auto num = [](const char* name, int lo, int hi) {
    const char* v = getenv(name);
    if (!v) return 0;
    int x = atoi(v);
    return x >= lo && x <= hi ? x : 0;
  };
  e.msm_window = num("G16_MSM_WINDOW", 5, 22);
  e.table_window = num("G16_TABLE_WINDOW", 5, 22);
  e.msm_seg = num("G16_MSM_SEG", 8, 4096);
  e.red_slice_log2 = num("G16_RED_SLICE", 8, 11);
  e.cu_split = num("G16_CU_SPLIT", 1, 24);
  e.heavy_grid = num("G16_HEAVY_GRID", 1, 4096);
  if (const char* v = getenv("G16_INF_COMPACT")) e.inf_compact_pct = atoi(v) < 0 ? 0 : atoi(v) > 101 ? 101 : atoi(v);
  if (const char* v = getenv("G16_R2_WIDTH")) e.r2_width = v[0] == '0' ? 0 : v[0] == '2' ? 2 : 1;
  if (const char* v = getenv("G16_LANES_AFTER_QUOTIENT")) e.lanes_after_quotient = v[0] != '0';
  if (const char* v = getenv("G16_QUOTIENT_FIRST")) e.quotient_first = v[0] != '0';
  if (const char* v = getenv("G16_G1_BATCH")) e.g1_batch = v[0] != '0';
  if (const char* v = getenv("G16_CHAIN_CH")) e.chain_ch = v[0] != '0';
  if (const char* v = getenv("G16_ABC_DICT")) e.abc_dict = v[0] != '0';
  if (const char* v = getenv("G16_CZ_FLY")) e.cz_on_the_fly = v[0] != '0';
  if (const char* v = getenv("G16_RED_CHUNK")) e.red_chunk = atoi(v) == 2 || atoi(v) == 4 || atoi(v) == 8 || atoi(v) == 16 ? atoi(v) : 0;
  if (const char* v = getenv("G16_TAIL_QUAD")) e.tail_quad = v[0] != '0';
  if (const char* v = getenv("G16_G2_FIRST")) e.g2_first = v[0] == '2' ? 2 : v[0] != '0';
  if (const char* v = getenv("G16_MTAB")) e.mtab = v[0] == '1' ? 1 : 2;
  if (const char* v = getenv("G16_NTT_TILE")) e.ntt_tile = atoi(v) == 1024 ? 1024 : atoi(v) == 4096 ? 4096 : 2048;
  if (const char* v = getenv("G16_MSM_SORT")) e.msm_sort = v[0];
  if (const char* v = getenv("G16_G1_LANES"))
    if (strlen(v) == 3 && strspn(v, "023") == 3)
      for (int i = 0; i < 3; ++i) e.g1_lanes[i] = v[i] - '0';
  if (const char* v = getenv("G16_STREAM_PRIO"))
    if (strlen(v) >= 6 && strspn(v, "hnl") >= 6) memcpy(e.stream_prio, v, 6);
  return e;
}
const G16Env& g16_env() {
  static const G16Env env = read_env();   // thread-safe one-time initialisation
  return env;
}

// ---- CU partition (experiment, G16_CU_SPLIT=k) -----------------------------------------------------
// With k > 0 the context's main stream (witness upload, buildABC, NTTs, sorts, the fold of H) is created over k CUs of
// every XCD and the five MSM lanes over the remaining 32 - k: the front of a proof then owns CUs instead of waiting for
// accumulate waves of other streams to drain.  CU-mask bit i is CU (i / 8) of XCD (i % 8) on this part (the mask is dealt
// round robin over the XCCs).  Such streams have no priority.  Measured: profiles/r05_ab_cu_split.txt.
static hipError_t stream_create(hipStream_t* st, int prio_index, bool front) {
  const int k = g16_env().cu_split;
  if (k <= 0) return hipStreamCreateWithPriority(st, hipStreamNonBlocking, g16_stream_priority(prio_index));
  uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int bit = 0; bit < 256; ++bit) {
    const bool in_front = bit / 8 < k;
    if (in_front == front) mask[bit / 32] |= 1u << (bit % 32);
  }
  return hipExtStreamCreateWithCUMask(st, 8, mask);
}

// ---- context ------------------------------------------------------------------------------------
extern "C" int32_t g16_ctx_create(int32_t device, g16_ctx** out) {
  if (!out) return G16_EINVAL;
  *out = nullptr;
  (void)g16_env();
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return G16_ENODEV;
  if (device < 0 || device >= ndev) return G16_EINVAL;
  if (hipSetDevice(device) != hipSuccess) return G16_ENODEV;
  g16_ctx* ctx = new (std::nothrow) g16_ctx();
  if (!ctx) return G16_ENOMEM;
  ctx->device = device;
  if (stream_create(&ctx->stream, 5, true) != hipSuccess) {
    delete ctx;
    return G16_ENODEV;
  }
  ctx->own_stream = true;
  if (g16_lanes_init(ctx) != G16_OK) {
    g16_ctx_destroy(ctx);
    return G16_ENODEV;
  }
  *out = ctx;
  return G16_OK;
}

// Stream priorities.  Index 0..4 = lanes (0: witness sort + C1, 1: B2, 2: B1, 3: A1, 4: spare), 5 = main stream
// (buildABC, quotient NTTs, H).  Default "lhllln" (G2 high, main normal, G1 lanes low): the G2 MSM has the longest
// latency chain of a proof; prioritising it lets its reduce/fold tail overlap the G1 accumulations of this and
// of the other in-flight proofs (measured with 3 proofs in flight: 78 proofs/s flat, 90 "lhlllh", 97 "lhllln").
// G16_STREAM_PRIO = six characters from {h, n, l} overrides it (read once per process, g16_env).
int g16_stream_priority(int index) {
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // numerically: hi <= lo
  const char* cfg = g16_env().stream_prio;
  switch (cfg[index]) {
    case 'h': return hi;
    case 'l': return lo;
    default: return (lo + hi) / 2;
  }
}

int32_t g16_lanes_init(g16_ctx* ctx) {
  for (int i = 0; i < 5; ++i) {
    auto& l = ctx->lane[i];
    if (stream_create(&l.stream, i, false) != hipSuccess) return G16_EHIP;
    if (hipEventCreateWithFlags(&l.done, hipEventDisableTiming) != hipSuccess) return G16_EHIP;
  }
  if (hipEventCreateWithFlags(&ctx->ev_a, hipEventDisableTiming) != hipSuccess) return G16_EHIP;
  if (hipEventCreateWithFlags(&ctx->ev_b, hipEventDisableTiming) != hipSuccess) return G16_EHIP;
  if (hipEventCreateWithFlags(&ctx->ev_q, hipEventDisableTiming) != hipSuccess) return G16_EHIP;
  if (hipEventCreateWithFlags(&ctx->ev_b2, hipEventDisableTiming) != hipSuccess) return G16_EHIP;
  if (hipEventCreateWithFlags(&ctx->ev_c, hipEventDisableTiming) != hipSuccess) return G16_EHIP;
  if (hipEventCreateWithFlags(&ctx->ev_g2, hipEventDisableTiming) != hipSuccess) return G16_EHIP;
  return G16_OK;
}

extern "C" void g16_ctx_destroy(g16_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  ctx_quiesce(ctx);
  for (auto& l : ctx->lane) {
    if (l.stream) (void)hipStreamDestroy(l.stream);
    if (l.done) (void)hipEventDestroy(l.done);
    if (l.acc.p) (void)hipFree(l.acc.p);
  }
  for (auto& srt : ctx->sort)
    if (srt.buf.p) (void)hipFree(srt.buf.p);
  if (ctx->ev_a) (void)hipEventDestroy(ctx->ev_a);
  if (ctx->ev_b) (void)hipEventDestroy(ctx->ev_b);
  if (ctx->ev_q) (void)hipEventDestroy(ctx->ev_q);
  if (ctx->ev_b2) (void)hipEventDestroy(ctx->ev_b2);
  if (ctx->ev_c) (void)hipEventDestroy(ctx->ev_c);
  if (ctx->ev_g2) (void)hipEventDestroy(ctx->ev_g2);
  for (g16_ctx::Buf* b : {&ctx->stage_s, &ctx->stage_p, &ctx->stage_p29, &ctx->stage_o, &ctx->ntt_tw, &ctx->ntt_tmp,
                          &ctx->coset[0], &ctx->coset[1], &ctx->quot, &ctx->prove, &ctx->fb_table[0],
                          &ctx->fb_table[1]})
    if (b->p) (void)hipFree(b->p);
  for (auto& e : ctx->prof) {
    (void)hipEventDestroy(e.e0);
    (void)hipEventDestroy(e.e1);
  }
  for (auto& e : ctx->free_events) (void)hipEventDestroy(e);
  if (ctx->clk_buf) (void)hipFree(ctx->clk_buf);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" const char* g16_last_error(const g16_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int32_t g16_ctx_set_stream(g16_ctx* ctx, void* hip_stream) {
  if (!ctx) return G16_EINVAL;
  CTX_ENTER(ctx);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->own_stream) {
    (void)hipStreamDestroy(ctx->stream);
    ctx->own_stream = false;
  }
  if (hip_stream) {
    ctx->stream = (hipStream_t)hip_stream;
  } else {
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
  }
  return G16_OK;
}

// waits for everything the context has queued (main stream and lanes) and forgets a pending
// g16_prove_partials_begin: what a caller does after a failed exchange
extern "C" int32_t g16_ctx_cancel(g16_ctx* ctx) {
  if (!ctx) return G16_EINVAL;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx_quiesce(ctx);
  ctx->shard_begun = nullptr;
  return G16_OK;
}

extern "C" int32_t g16_ctx_synchronize(g16_ctx* ctx) {
  if (!ctx) return G16_EINVAL;
  CTX_ENTER_KEEP(ctx);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return G16_OK;
}

extern "C" int32_t g16_profile_enable(g16_ctx* ctx, int32_t on) {
  if (!ctx) return G16_EINVAL;
  CTX_ENTER_KEEP(ctx);
  if (on && !ctx->clk_buf) {
    HIPCHK(ctx, hipMalloc((void**)&ctx->clk_buf, 16));
    HIPCHK(ctx, hipMemset(ctx->clk_buf, 0, 16));
  }
  ctx->profiling = on != 0;
  ctx->prof_accum_only = on == 2;
  return G16_OK;
}
extern "C" int32_t g16_profile_reset(g16_ctx* ctx) {
  if (!ctx) return G16_EINVAL;
  CTX_ENTER_KEEP(ctx);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& e : ctx->prof) {
    ctx->free_events.push_back(e.e0);
    ctx->free_events.push_back(e.e1);
  }
  ctx->prof.clear();
  return G16_OK;
}
extern "C" int32_t g16_profile_report(g16_ctx* ctx, char* buf, size_t buflen) {
  if (!ctx || !buf || buflen < 3) return G16_EINVAL;
  CTX_ENTER_KEEP(ctx);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  std::map<std::string, std::pair<int, double>> agg;
  for (auto& e : ctx->prof) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, e.e0, e.e1) == hipSuccess) {
      auto& a = agg[e.name];
      a.first++;
      a.second += ms;
    }
  }
  std::string s = "{";
  bool first = true;
  for (auto& kv : agg) {
    char tmp[256];
    snprintf(tmp, sizeof tmp, "%s\"%s\": {\"calls\": %d, \"total_ms\": %.6f}", first ? "" : ", ", kv.first.c_str(),
             kv.second.first, kv.second.second);
    s += tmp;
    first = false;
  }
  s += "}";
  if (s.size() + 1 > buflen) return G16_EINVAL;
  memcpy(buf, s.c_str(), s.size() + 1);
  return G16_OK;
}

// ---- sustained shader clock of the accumulate kernels -------------------------------------------------------
// While profiling is on, thread 0 of every msm_accum workgroup stamps the shader clock counter (s_memtime: one tick
// per shader cycle) and the constant 100 MHz counter (s_memrealtime) around its task and adds both deltas to two
// 64-bit sums (msm.cuh).  sum(d_memtime) / sum(d_memrealtime) * 100 MHz is the clock the chip sustained WHILE THAT
// KERNEL RAN -- the accumulate kernels are power-limited (~2.0 GHz against 2.4 GHz nominal), so the VALU roofline
// must not borrow a clock from another box, another kernel or the idle gaps between launches.
extern "C" int32_t g16_profile_clock(g16_ctx* ctx, double* ghz) {
  if (!ctx || !ghz) return G16_EINVAL;
  CTX_ENTER_KEEP(ctx);
  *ghz = 0.0;
  if (!ctx->clk_buf) return G16_OK;
  ctx_quiesce(ctx);
  unsigned long long h[2] = {0, 0};
  HIPCHK(ctx, hipMemcpy(h, ctx->clk_buf, 16, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemset(ctx->clk_buf, 0, 16));
  if (h[1]) *ghz = (double)h[0] / (double)h[1] * 0.1;
  return G16_OK;
}

template <class C>
static int32_t msm_entry(g16_ctx* ctx, const void* scalars, uint32_t flags, const void* points, size_t n, void* out,
                         bool on_device, bool partial, const char* tag) {
  if (!ctx) return G16_EINVAL;
  if (!out || (n > 0 && (!scalars || !points))) {
    ctx->err = "null pointer argument";
    return G16_EINVAL;
  }
  if (n >= (size_t(1) << 27)) {
    ctx->err = "n too large (must be < 2^27)";
    return G16_EINVAL;
  }
  CTX_ENTER(ctx);
  const size_t out_bytes = partial ? sizeof(typename C::Acc) : sizeof(typename C::Aff);
  if (n == 0) {  // msm.nim:117: the sum starts from infG1 / infG2 = (0,0); XYZZ infinity is all-zero as well
    memset(out, 0, out_bytes);
    return G16_OK;
  }
  const void* d_s = scalars;
  const void* d_p = points;
  int32_t rc;
  if (!on_device) {
    if ((rc = ensure(ctx, ctx->stage_s, n * 32))) return rc;
    if ((rc = ensure(ctx, ctx->stage_p, n * sizeof(typename C::Aff)))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage_s.p, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage_p.p, points, n * sizeof(typename C::Aff), hipMemcpyHostToDevice,
                               ctx->stream));
    d_s = ctx->stage_s.p;
    d_p = ctx->stage_p.p;
  }
  if ((rc = ensure(ctx, ctx->stage_o, 512))) return rc;
  auto* d_aff = partial ? nullptr : (typename C::Aff*)ctx->stage_o.p;
  auto* d_acc = partial ? (typename C::Acc*)ctx->stage_o.p : nullptr;
  rc = sizeof(typename C::Aff) == 64 ? g16_msm_device_g1(ctx, d_s, flags, d_p, n, d_aff, d_acc, 0)
                                     : g16_msm_device_g2(ctx, d_s, flags, d_p, n, d_aff, d_acc, 0);
  if (rc) return rc;
  HIPCHK(ctx, hipMemcpyAsync(out, ctx->stage_o.p, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return G16_OK;
}

// ---- registered point sets (ProverPoints are constant per circuit: zkey_types.nim:36-41) ----------------
static int32_t points_register(g16_ctx* ctx, int group, const void* points, size_t n, bool on_device,
                               g16_points** out) {
  if (!ctx) return G16_EINVAL;
  if (!out || (n && !points) || n >= (size_t(1) << 26)) {
    ctx->err = "bad argument";
    return G16_EINVAL;
  }
  *out = nullptr;
  CTX_ENTER(ctx);
  g16_points* h = new (std::nothrow) g16_points();
  if (!h) return G16_ENOMEM;
  h->device = ctx->device;
  h->group = group;
  h->n = n;
  h->c = g16_pick_window_g1(n);
  h->nwin = 254 / h->c + 1;
  h->mtab = g16_pick_mtab(h->c);
  const size_t psz = group == 1 ? 64 : 128;
  // Two multiplier tables per window double the set's HBM footprint (~10 GB for a 2^20 key, ~40 GB at 2^22): a set
  // that does not fit that way -- table indices beyond 31 bits, or no room in HBM -- falls back to one table per window
  // and the plain bucket set (the rounds 1-3 layout) instead of failing.
  if ((size_t)h->mtab * h->nwin * n >= (size_t(1) << 31)) h->mtab = 1;
  if ((size_t)h->mtab * h->nwin * n >= (size_t(1) << 31)) {
    delete h;
    ctx->err = "point set too large for 31-bit table indices";
    return G16_EINVAL;
  }
  if (n) {
    hipError_t e = hipMalloc(&h->d_tables, (size_t)h->mtab * h->nwin * n * psz);   // packed reduced-radix entries: 64 / 128 B
    if (e != hipSuccess && h->mtab == 2) {
      (void)hipGetLastError();
      h->mtab = 1;
      e = hipMalloc(&h->d_tables, (size_t)h->nwin * n * psz);
    }
    if (e != hipSuccess) {
      delete h;
      ctx->err = "hipMalloc(tables) failed";
      return G16_ENOMEM;
    }
    const void* d_src = points;
    int32_t rc = G16_OK;
    if (!on_device) {
      rc = ensure(ctx, ctx->stage_p, n * psz);
      if (!rc && hipMemcpyAsync(ctx->stage_p.p, points, n * psz, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        rc = G16_EHIP;
      d_src = ctx->stage_p.p;
    }
    if (!rc)
      rc = group == 1 ? g16_precompute_device_g1(ctx, d_src, n, h->c, h->mtab, h->d_tables)
                      : g16_precompute_device_g2(ctx, d_src, n, h->c, h->mtab, h->d_tables);
    // which points are (0,0): snarkjs keys hold the point at infinity for every wire absent from a matrix
    uint32_t n_inf = 0;
    if (!rc && hipMalloc((void**)&h->d_live, ((n + 31) / 32 + 1) * 4) != hipSuccess) rc = G16_ENOMEM;
    if (!rc) rc = ensure(ctx, ctx->stage_o, 2048);
    if (!rc) {
      uint32_t* d_cnt = (uint32_t*)ctx->stage_o.p;
      if (hipMemsetAsync(d_cnt, 0, 4, ctx->stream) != hipSuccess) rc = G16_EHIP;
      if (!rc) rc = group == 1 ? g16_live_bitmap_device_g1(ctx, d_src, n, h->d_live, d_cnt)
                               : g16_live_bitmap_device_g2(ctx, d_src, n, h->d_live, d_cnt);
      if (!rc && hipMemcpyAsync(&n_inf, d_cnt, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = G16_EHIP;
    }
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = G16_EHIP;
    h->n_inf = n_inf;
    if (rc) {
      if (h->d_live) (void)hipFree(h->d_live);
      (void)hipFree(h->d_tables);
      delete h;
      if (ctx->err.empty()) ctx->err = "point registration failed";
      return rc;
    }
  }
  *out = h;
  return G16_OK;
}
extern "C" int32_t g16_points_register_g1(g16_ctx* ctx, const void* points, size_t n, g16_points** out) {
  return points_register(ctx, 1, points, n, false, out);
}
extern "C" int32_t g16_points_register_g2(g16_ctx* ctx, const void* points, size_t n, g16_points** out) {
  return points_register(ctx, 2, points, n, false, out);
}
extern "C" int32_t g16_points_register_g1_dev(g16_ctx* ctx, const void* d_points, size_t n, g16_points** out) {
  return points_register(ctx, 1, d_points, n, true, out);
}
extern "C" int32_t g16_points_register_g2_dev(g16_ctx* ctx, const void* d_points, size_t n, g16_points** out) {
  return points_register(ctx, 2, d_points, n, true, out);
}
extern "C" void g16_points_release(g16_points* h) {
  if (!h) return;
  if (h->d_tables) {
    // any context of the device may still have an MSM against these tables in flight, and the context that
    // registered them may already be gone: wait for the whole device, never for one context's stream
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(h->d_tables);
    if (h->d_live) (void)hipFree(h->d_live);
  }
  delete h;
}
extern "C" size_t g16_points_inf_count(const g16_points* h) { return h ? h->n_inf : 0; }
const uint32_t* g16_points_live_if_sparse(const g16_points* p) {
  return p && p->d_live && p->n && p->n_inf * 100 >= (size_t)g16_env().inf_compact_pct * p->n && p->n_inf ? p->d_live
                                                                                                        : nullptr;
}
extern "C" size_t g16_points_count(const g16_points* h) { return h ? h->n : 0; }
extern "C" int32_t g16_points_info(const g16_points* h, uint32_t* window_bits, uint32_t* ntables) {
  if (!h) return G16_EINVAL;
  if (window_bits) *window_bits = h->c;
  if (ntables) *ntables = h->nwin * h->mtab;
  return G16_OK;
}

// on-curve check of a host point array (mkG1 / mkG2 asserts of the reference's loaders, curves.nim:95-107):
// *first_bad = index of the first point off the curve, or SIZE_MAX if every point is on it ((0,0) = infinity ok)
static int32_t points_check(g16_ctx* ctx, int group, const void* points, size_t n, size_t* first_bad) {
  if (!ctx) return G16_EINVAL;
  if (!first_bad || (n && !points) || n >= (size_t(1) << 31)) {
    ctx->err = "bad argument";
    return G16_EINVAL;
  }
  CTX_ENTER(ctx);
  const size_t psz = group == 1 ? 64 : 128;
  int32_t rc;
  if ((rc = ensure(ctx, ctx->stage_p, n * psz + psz))) return rc;
  if ((rc = ensure(ctx, ctx->stage_o, 2048))) return rc;
  uint32_t* d_bad = (uint32_t*)ctx->stage_o.p;
  HIPCHK(ctx, hipMemsetAsync(d_bad, 0xff, 4, ctx->stream));
  if (n) HIPCHK(ctx, hipMemcpyAsync(ctx->stage_p.p, points, n * psz, hipMemcpyHostToDevice, ctx->stream));
  rc = group == 1 ? g16_on_curve_device_g1(ctx, ctx->stage_p.p, n, d_bad)
                  : g16_on_curve_device_g2(ctx, ctx->stage_p.p, n, d_bad);
  if (rc) return rc;
  uint32_t bad = 0;
  HIPCHK(ctx, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  *first_bad = bad == 0xffffffffu ? (size_t)-1 : (size_t)bad;
  return G16_OK;
}
extern "C" int32_t g16_points_check_g1(g16_ctx* ctx, const void* points, size_t n, size_t* first_bad) {
  return points_check(ctx, 1, points, n, first_bad);
}
extern "C" int32_t g16_points_check_g2(g16_ctx* ctx, const void* points, size_t n, size_t* first_bad) {
  return points_check(ctx, 2, points, n, first_bad);
}

// out[i] = scalars[i] * generator  (`y ** gen1` / `y ** gen2`, fake_setup.nim:258-261); host pointers
static int32_t fixed_base(g16_ctx* ctx, int group, const void* scalars, uint32_t flags, size_t n, void* out) {
  if (!ctx) return G16_EINVAL;
  if (n && (!scalars || !out)) {
    ctx->err = "null pointer argument";
    return G16_EINVAL;
  }
  CTX_ENTER(ctx);
  const size_t psz = group == 1 ? 64 : 128;
  int32_t rc;
  g16_ctx::Buf& tb = ctx->fb_table[group - 1];
  if ((rc = ensure(ctx, tb, 32 * 255 * psz))) return rc;
  if ((rc = ensure(ctx, ctx->stage_s, n * 32 + 32))) return rc;
  if ((rc = ensure(ctx, ctx->stage_p, n * psz + psz))) return rc;
  if (n) HIPCHK(ctx, hipMemcpyAsync(ctx->stage_s.p, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
  const uint32_t mont = (flags & G16_SCALARS_MONT) ? 1u : 0u;
  rc = group == 1 ? g16_fixed_base_device_g1(ctx, tb.p, ctx->fb_ready[0], ctx->stage_s.p, mont, n, ctx->stage_p.p)
                  : g16_fixed_base_device_g2(ctx, tb.p, ctx->fb_ready[1], ctx->stage_s.p, mont, n, ctx->stage_p.p);
  if (rc) return rc;
  ctx->fb_ready[group - 1] = true;
  if (n) HIPCHK(ctx, hipMemcpyAsync(out, ctx->stage_p.p, n * psz, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return G16_OK;
}
extern "C" int32_t g16_fixed_base_g1(g16_ctx* ctx, const void* scalars, uint32_t flags, size_t n, void* out) {
  return fixed_base(ctx, 1, scalars, flags, n, out);
}
extern "C" int32_t g16_fixed_base_g2(g16_ctx* ctx, const void* scalars, uint32_t flags, size_t n, void* out) {
  return fixed_base(ctx, 2, scalars, flags, n, out);
}

// MSM against a registered set; flags: G16_SCALARS_MONT | G16_SCALARS_DEVICE | G16_OUT_PARTIAL
extern "C" int32_t g16_msm_points(g16_ctx* ctx, const g16_points* pts, const void* scalars, uint32_t flags,
                                  void* out) {
  if (!ctx) return G16_EINVAL;
  if (!pts || !out || pts->device != ctx->device || (pts->n && !scalars)) {
    ctx->err = "bad argument (null pointer or point set of another device)";
    return G16_EINVAL;
  }
  CTX_ENTER(ctx);
  const bool partial = (flags & G16_OUT_PARTIAL) != 0;
  const size_t psz = pts->group == 1 ? 64 : 128;
  const size_t out_bytes = partial ? 2 * psz : psz;
  const size_t n = pts->n;
  if (n == 0) {
    memset(out, 0, out_bytes);
    return G16_OK;
  }
  int32_t rc;
  const void* d_s = scalars;
  if (!(flags & G16_SCALARS_DEVICE)) {
    if ((rc = ensure(ctx, ctx->stage_s, n * 32))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->stage_s.p, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    d_s = ctx->stage_s.p;
  }
  if ((rc = ensure(ctx, ctx->stage_o, 512))) return rc;
  void* d_aff = partial ? nullptr : ctx->stage_o.p;
  void* d_acc = partial ? ctx->stage_o.p : nullptr;
  const uint32_t* live = g16_points_live_if_sparse(pts);
  rc = pts->group == 1 ? g16_msm_device_g1(ctx, d_s, flags, pts->d_tables, n, d_aff, d_acc, pts->cfg(), live)
                       : g16_msm_device_g2(ctx, d_s, flags, pts->d_tables, n, d_aff, d_acc, pts->cfg(), live);
  if (rc) return rc;
  HIPCHK(ctx, hipMemcpyAsync(out, ctx->stage_o.p, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return G16_OK;
}

extern "C" int32_t g16_msm_g1(g16_ctx* ctx, const void* s, uint32_t f, const void* p, size_t n, void* out) {
  return msm_entry<G1>(ctx, s, f, p, n, out, false, false, "g1");
}
extern "C" int32_t g16_msm_g2(g16_ctx* ctx, const void* s, uint32_t f, const void* p, size_t n, void* out) {
  return msm_entry<G2>(ctx, s, f, p, n, out, false, false, "g2");
}
extern "C" int32_t g16_msm_g1_dev(g16_ctx* ctx, const void* s, uint32_t f, const void* p, size_t n, void* out) {
  return msm_entry<G1>(ctx, s, f, p, n, out, true, false, "g1");
}
extern "C" int32_t g16_msm_g2_dev(g16_ctx* ctx, const void* s, uint32_t f, const void* p, size_t n, void* out) {
  return msm_entry<G2>(ctx, s, f, p, n, out, true, false, "g2");
}
extern "C" int32_t g16_msm_g1_partial_dev(g16_ctx* ctx, const void* s, uint32_t f, const void* p, size_t n,
                                          void* out) {
  return msm_entry<G1>(ctx, s, f, p, n, out, true, true, "g1");
}
extern "C" int32_t g16_msm_g2_partial_dev(g16_ctx* ctx, const void* s, uint32_t f, const void* p, size_t n,
                                          void* out) {
  return msm_entry<G2>(ctx, s, f, p, n, out, true, true, "g2");
}

template <class C>
static int32_t sum_partials(g16_ctx* ctx, const void* xyzz, size_t count, void* out) {
  if (!ctx) return G16_EINVAL;
  if (!out || (count && !xyzz) || count > 4096) {
    ctx->err = "bad argument";
    return G16_EINVAL;
  }
  CTX_ENTER(ctx);
  int32_t rc;
  size_t bytes = count * sizeof(typename C::Acc);
  if ((rc = ensure(ctx, ctx->stage_p, bytes + 256))) return rc;
  if ((rc = ensure(ctx, ctx->stage_o, 512))) return rc;
  if (count) HIPCHK(ctx, hipMemcpyAsync(ctx->stage_p.p, xyzz, bytes, hipMemcpyHostToDevice, ctx->stream));
  rc = sizeof(typename C::Aff) == 64 ? g16_sum_partials_device_g1(ctx, ctx->stage_p.p, (uint32_t)count, ctx->stage_o.p)
                                     : g16_sum_partials_device_g2(ctx, ctx->stage_p.p, (uint32_t)count, ctx->stage_o.p);
  if (rc) return rc;
  HIPCHK(ctx, hipMemcpyAsync(out, ctx->stage_o.p, sizeof(typename C::Aff), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return G16_OK;
}
extern "C" int32_t g16_g1_sum_partials(g16_ctx* ctx, const void* x, size_t count, void* out) {
  return sum_partials<G1>(ctx, x, count, out);
}
extern "C" int32_t g16_g2_sum_partials(g16_ctx* ctx, const void* x, size_t count, void* out) {
  return sum_partials<G2>(ctx, x, count, out);
}

// ---- NTT ------------------------------------------------------------------------------------------
extern "C" int32_t g16_ntt_fr_dev(g16_ctx* ctx, const void* d_src, void* d_dst, uint32_t log2n, int32_t inverse) {
  if (!ctx) return G16_EINVAL;
  if (!d_src || !d_dst || log2n > 28) {
    ctx->err = "bad argument (null pointer or log2n > 28)";
    return G16_EINVAL;
  }
  CTX_ENTER(ctx);
  return g16_ntt_device(ctx, d_src, d_dst, log2n, inverse ? 1 : 0);
}

extern "C" int32_t g16_ntt_fr(g16_ctx* ctx, const void* src, void* dst, uint32_t log2n, int32_t inverse) {
  if (!ctx) return G16_EINVAL;
  if (!src || !dst || log2n > 28) {
    ctx->err = "bad argument (null pointer or log2n > 28)";
    return G16_EINVAL;
  }
  CTX_ENTER(ctx);
  const size_t bytes = (size_t(1) << log2n) * 32;
  int32_t rc;
  if ((rc = ensure(ctx, ctx->stage_s, bytes))) return rc;
  if ((rc = ensure(ctx, ctx->stage_p, bytes))) return rc;
  HIPCHK(ctx, hipMemcpyAsync(ctx->stage_s.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = g16_ntt_device(ctx, ctx->stage_s.p, ctx->stage_p.p, log2n, inverse ? 1 : 0))) return rc;
  HIPCHK(ctx, hipMemcpyAsync(dst, ctx->stage_p.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return G16_OK;
}

// ---- selftest -------------------------------------------------------------------------------------
__global__ void selftest_kernel(uint32_t* out) {
  if (threadIdx.x != 0) return;
  uint32_t ok = 1;
  // Montgomery one * one == one ; from_mont(one) == 1 ; gen1 = (1,2) on y^2 = x^3 + 3
  u256 one = Fp::one();
  ok &= Fp::eq(Fp::mul(one, one), one);
  u256 s1 = Fr::from_mont(Fr::one());
  ok &= (s1.v[0] == 1);
  for (int i = 1; i < 8; ++i) ok &= (s1.v[i] == 0);
  u256 x = one, y = Fp::dbl(one);
  u256 three = Fp::add(Fp::dbl(one), one);
  ok &= Fp::eq(Fp::sqr(y), Fp::add(Fp::mul(Fp::sqr(x), x), three));
  // 5*G via madd chain == to_affine(dbl(dbl(G)) + G)
  g1_aff g{x, y};
  g1_acc a = G1::acc_inf();
  for (int i = 0; i < 5; ++i) G1::madd(a, g);
  g1_acc b = G1::dbl(G1::dbl(G1::from_affine(g)));
  G1::madd(b, g);
  g1_aff pa = G1::to_affine(a), pb = G1::to_affine(b);
  ok &= Fp::eq(pa.x, pb.x) && Fp::eq(pa.y, pb.y);
  // inverse
  ok &= Fp::eq(Fp::mul(three, Fp::inv(three)), one);
  out[0] = ok;
  // inputs / expected values of the host-driven known-answer checks below (out + 16 words onwards):
  //   points {G, 2G, G}, scalars {1, 2, 3} (standard form)  ->  MSM = 8G ;  NTT input e_0 -> all ones
  g1_aff* pts = reinterpret_cast<g1_aff*>(out + 16);
  g1_aff g2x = G1::to_affine(G1::dbl(G1::from_affine(g)));
  pts[0] = g;
  pts[1] = g2x;
  pts[2] = g;
  pts[3] = G1::to_affine(G1::dbl(G1::dbl(G1::dbl(G1::from_affine(g)))));   // 8G: the expected MSM value
  u256* sc = reinterpret_cast<u256*>(out + 16 + 64);
  for (int i = 0; i < 3; ++i) {
    sc[i] = Fr::zero();
    sc[i].v[0] = (uint32_t)i + 1;
  }
  u256* nt = reinterpret_cast<u256*>(out + 16 + 64 + 24);   // 8 Fr: e_0 in Montgomery form
  for (int i = 0; i < 8; ++i) nt[i] = i == 0 ? Fr::one() : Fr::zero();
  nt[8] = Fr::one();                                          // what every output element must equal
}

extern "C" int32_t g16_selftest(g16_ctx* ctx) {
  if (!ctx) return G16_EINVAL;
  static_assert(sizeof(u256) == 32 && sizeof(g1_aff) == 64 && sizeof(g2_aff) == 128, "layout");
  CTX_ENTER(ctx);
  int32_t rc;
  g16_ctx::Buf st;   // own scratch: the MSM / NTT below use the context's staging buffers
  if ((rc = ensure(ctx, st, 4096))) return rc;
  uint32_t* d = (uint32_t*)st.p;
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, ctx->stream, d);
  uint32_t ok = 0;
  struct {
    g1_aff msm, want;
    u256 ntt[8], one;
  } h;
  auto finish = [&](int32_t code, const char* msg) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(st.p);
    if (code) ctx->err = msg;
    return code;
  };
  // known-answer MSM through the whole Pippenger pipeline (1*G + 2*(2G) + 3*G = 8G) and NTT (e_0 -> all ones)
  const g1_aff* d_pts = reinterpret_cast<const g1_aff*>(d + 16);
  const u256* d_sc = reinterpret_cast<const u256*>(d + 16 + 64);
  u256* d_nt = reinterpret_cast<u256*>(d + 16 + 64 + 24);
  g1_aff* d_res = reinterpret_cast<g1_aff*>(d + 16 + 64 + 24 + 72);
  u256* d_nout = reinterpret_cast<u256*>(d + 16 + 64 + 24 + 72 + 16);
  if ((rc = g16_msm_device_g1(ctx, d_sc, G16_SCALARS_STD, d_pts, 3, d_res, nullptr, 0))) return finish(rc, "self-test MSM failed to launch");
  if ((rc = g16_ntt_device(ctx, d_nt, d_nout, 3, 0))) return finish(rc, "self-test NTT failed to launch");
  if (hipMemcpyAsync(&ok, d, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipMemcpyAsync(&h.msm, d_res, 64, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipMemcpyAsync(&h.want, d_pts + 3, 64, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipMemcpyAsync(h.ntt, d_nout, 256, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipMemcpyAsync(&h.one, d_nt + 8, 32, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess)
    return finish(G16_EHIP, "self-test copy failed");
  if (ok != 1) return finish(G16_ESELFTEST, "device arithmetic self-test failed");
  if (memcmp(&h.msm, &h.want, 64) != 0) return finish(G16_ESELFTEST, "known-answer MSM self-test failed");
  for (int i = 0; i < 8; ++i)
    if (memcmp(&h.ntt[i], &h.one, 32) != 0) return finish(G16_ESELFTEST, "known-answer NTT self-test failed");
  return finish(G16_OK, "");
}
