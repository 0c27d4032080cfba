// Proving-key residency and the per-proof launch sequence: the GPU counterpart of
// generateProofWithMask (reference groth16/prover.nim:215-304).
//   buildABC                      prover.nim:56-73     -> spmv_binned kernel (spmv.hip; rows binned once per key)
//   computeSnarkjsScalarCoeffs /  prover.nim:158-181   -> g16_quotient_device (ntt.hip)
//   computeQuotientPointwise      prover.nim:118-148
//   5 x msmMultiThreaded          prover.nim:282-302   -> msm_device on the registered point tables
//   mask algebra (`**`, `+=`)     prover.nim:279-302   -> O(1) curve operations on the host, as in the
//                                                         reference (curves.nim:136-214); not a hot loop
#include <algorithm>
#include <cstddef>
#include <new>

#include "g16_internal.hpp"
#include "ec.cuh"
#include "host_ff64.hpp"

using namespace g16;

// Immutable after g16_pkey_create and tied to a DEVICE, not to the creating context: the in-flight proofs of one
// GPU (one g16_ctx each: private streams and workspaces) all prove against ONE resident key, and the key may be
// destroyed before or after any of those contexts (ProverPoints are per-circuit constants, zkey_types.nim:36-41).
struct g16_pkey {
  int device = 0;
  uint32_t nvars = 0, npubs = 0, log2n = 0, flavour = 1;
  // this key holds the index ranges [lo, hi) of each point set (msm.nim:105-115 chunk rule over shard_count ranks)
  uint32_t shard_index = 0, shard_count = 1;
  size_t w_lo = 0, w_hi = 0;   // A1 / B1 / B2 / C1 : range of wires (C1 is stored wire-aligned, see below)
  size_t h_lo = 0, h_hi = 0;   // H1                : range of domain indices
  g16_points *A1 = nullptr, *B1 = nullptr, *B2 = nullptr, *C1 = nullptr, *H1 = nullptr;
  // the A and B matrices (zkey section 4 / ZKey.coeffs, zkey_types.nim:48-59), row-binned for buildABC (spmv.hip)
  g16_spmat* abc = nullptr;
  size_t ncoeffs = 0;
  g1_aff alpha1, beta1, delta1;
  g2_aff beta2, delta2;
  // Points at infinity.  snarkjs keys hold (0,0) in pointsA1 / pointsB1 / pointsB2 for every wire absent from the
  // matrix (the loaders accept them, curves.nim:95-107; the MSMs sum over them, msm.nim:128-158).  In the shared
  // witness sort such an entry still occupies a loop trip of a wave whose other lanes run a full addition, so a set
  // with >= G16_INF_COMPACT % of them gets entry lists of its own that leave them out: liveA = A1's bitmap, liveB =
  // the union of B1's and B2's (one sort serves both).  nullptr = dense: the set rides on the shared sort.
  const uint32_t* liveA = nullptr;
  uint32_t* liveB = nullptr;      // owned (the union), or nullptr
  size_t deadB = 0;               // wires whose B1 AND B2 points are both (0,0)
};

// ---- host-side O(1) curve helpers (the reference does these on the host too: curves.nim:136-214) ------
// 64-bit-limb host field (host_ff64.hpp) under the same curve templates; 4-bit fixed windows.
using HG1 = Curve<HFp>;
using HG2 = Curve<HFp2>;
template <class HC, class DevAff>
static DevAff host_mul(const u256& k_std, const DevAff& p_dev) {
  static_assert(sizeof(DevAff) == sizeof(typename HC::Aff), "layout");
  typename HC::Aff p;
  memcpy(&p, &p_dev, sizeof p);
  typename HC::Acc tab[16];
  tab[0] = HC::acc_inf();
  tab[1] = HC::from_affine(p);
  for (int i = 2; i < 16; ++i) {
    tab[i] = tab[i - 1];
    HC::madd(tab[i], p);
  }
  typename HC::Acc acc = HC::acc_inf();
  for (int i = 7; i >= 0; --i)
    for (int nib = 7; nib >= 0; --nib) {
      if (!HC::is_inf(acc))
        for (int d = 0; d < 4; ++d) acc = HC::dbl(acc);
      uint32_t w = (k_std.v[i] >> (4 * nib)) & 15u;
      if (w) HC::add(acc, tab[w]);
    }
  typename HC::Aff r = HC::to_affine(acc);
  DevAff out;
  memcpy(&out, &r, sizeof out);
  return out;
}
template <class HC, class DevAff>
static DevAff host_add(const DevAff& a_dev, const DevAff& b_dev) {
  typename HC::Aff a, b;
  memcpy(&a, &a_dev, sizeof a);
  memcpy(&b, &b_dev, sizeof b);
  typename HC::Acc acc = HC::from_affine(a);
  HC::madd(acc, b);
  typename HC::Aff r = HC::to_affine(acc);
  DevAff out;
  memcpy(&out, &r, sizeof out);
  return out;
}
// k1 * p1 + k2 * p2 with ONE doubling chain (Shamir's trick, 2-bit joint windows: 16-entry table i*p1 + j*p2):
// 254 doublings + <= 127 additions instead of two separate 4-bit-window multiplications (512 + 156)
template <class HC, class DevAff>
static DevAff host_mul2(const u256& k1_std, const DevAff& p1_dev, const u256& k2_std, const DevAff& p2_dev) {
  typename HC::Aff p1, p2;
  memcpy(&p1, &p1_dev, sizeof p1);
  memcpy(&p2, &p2_dev, sizeof p2);
  typename HC::Acc tab[16];   // tab[4 i + j] = i*p1 + j*p2
  tab[0] = HC::acc_inf();
  for (int j = 1; j < 4; ++j) {
    tab[j] = tab[j - 1];
    HC::madd(tab[j], p2);
  }
  for (int i = 1; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      tab[4 * i + j] = tab[4 * (i - 1) + j];
      HC::madd(tab[4 * i + j], p1);
    }
  typename HC::Acc acc = HC::acc_inf();
  for (int limb = 7; limb >= 0; --limb)
    for (int pos = 15; pos >= 0; --pos) {
      if (!HC::is_inf(acc)) acc = HC::dbl(HC::dbl(acc));
      const uint32_t w = 4 * ((k1_std.v[limb] >> (2 * pos)) & 3u) + ((k2_std.v[limb] >> (2 * pos)) & 3u);
      if (w) HC::add(acc, tab[w]);
    }
  typename HC::Aff r = HC::to_affine(acc);
  DevAff out;
  memcpy(&out, &r, sizeof out);
  return out;
}

extern "C" void g16_pkey_destroy(g16_pkey* k) {
  if (!k) return;
  for (g16_points* p : {k->A1, k->B1, k->B2, k->C1, k->H1}) g16_points_release(p);
  // like g16_points_release: wait for the device, not for a context (the creating one may be gone already)
  (void)hipSetDevice(k->device);
  (void)hipDeviceSynchronize();
  g16_spmat_destroy(k->abc);
  if (k->liveB) (void)hipFree(k->liveB);
  delete k;
}

// points at infinity per set: out[0..4] = A1, B1, B2, C1 (without the public wires this library pads it with), H1;
// out[5] = wires whose B1 and B2 points are both (0,0); out[6] = 1 if A1, out[7] = 1 if B1/B2 use compacted entry lists
extern "C" int32_t g16_pkey_inf_counts(const g16_pkey* k, size_t out[8]) {
  if (!k || !out) return G16_EINVAL;
  // public wires 0 .. npubs of this shard's wire range: their C1 slots are this library's padding, not key points
  const size_t pub_end = std::min<size_t>(k->w_hi, (size_t)k->npubs + 1);
  const size_t pad = pub_end > k->w_lo ? pub_end - k->w_lo : 0;
  out[0] = k->A1->n_inf, out[1] = k->B1->n_inf, out[2] = k->B2->n_inf, out[3] = k->C1->n_inf - pad, out[4] = k->H1->n_inf;
  out[5] = k->deadB, out[6] = k->liveA ? 1 : 0, out[7] = k->liveB ? 1 : 0;
  return G16_OK;
}

extern "C" int32_t g16_pkey_abc_info(const g16_pkey* k, size_t out[11]) {
  if (!k || !out) return G16_EINVAL;
  out[0] = k->ncoeffs;
  g16_spmat_info(k->abc, out + 1);
  return G16_OK;
}

// where a key's A / B coefficients come from: an array of g16_coeff (values c R), or the .zkey file's section 4 as it
// lies on disk (44-byte entries, values c R^2: files/zkey.nim:169-192, io.nim:134-139)
struct CoeffSource {
  const unsigned char* base = nullptr;   // first entry
  size_t count = 0, stride = 0, value_off = 0;
  bool values_r2 = false;
};
static inline uint32_t rd32(const unsigned char* p) {
  uint32_t v;
  memcpy(&v, p, 4);   // section 4 is not 4-byte aligned past its first entry
  return v;
}

static int32_t pkey_create(g16_ctx* ctx, const g16_pkey_desc* d, const CoeffSource& cs, g16_pkey** out) {
  const size_t n = size_t(1) << d->log2_domain;
  // shape rules of generateProofWithMask (prover.nim:236, 270-276)
  if (d->log2_domain > 27 || d->nvars == 0 || d->npubs + 1 > d->nvars || d->flavour > 1 || !d->pointsA1 ||
      !d->pointsB1 || !d->pointsB2 || !d->pointsH1 || (d->nvars - d->npubs - 1 > 0 && !d->pointsC1) ||
      !d->alpha1 || !d->beta1 || !d->delta1 || !d->beta2 || !d->delta2) {
    ctx->err = "bad proving-key description";
    return G16_EINVAL;
  }
  CTX_ENTER(ctx);
  g16_pkey* k = new (std::nothrow) g16_pkey();
  if (!k) return G16_ENOMEM;
  k->device = ctx->device;
  k->nvars = d->nvars;
  k->npubs = d->npubs;
  k->log2n = d->log2_domain;
  k->flavour = d->flavour;
  memcpy(&k->alpha1, d->alpha1, 64);
  memcpy(&k->beta1, d->beta1, 64);
  memcpy(&k->delta1, d->delta1, 64);
  memcpy(&k->beta2, d->beta2, 128);
  memcpy(&k->delta2, d->delta2, 128);
  int32_t rc;
#define TRY(x)               \
  if ((rc = (x)) != G16_OK) { \
    g16_pkey_destroy(k);     \
    return rc;               \
  }
  // contiguous index ranges per rank: b = (N*(k+1)) div ntasks   (msm.nim:107-115)
  k->shard_count = d->shard_count ? d->shard_count : 1;
  k->shard_index = d->shard_index;
  if (k->shard_index >= k->shard_count) {
    ctx->err = "shard_index >= shard_count";
    g16_pkey_destroy(k);
    return G16_EINVAL;
  }
  auto range = [&](size_t N, size_t& lo, size_t& hi) {
    lo = (N * k->shard_index) / k->shard_count;
    hi = (N * (k->shard_index + 1)) / k->shard_count;
  };
  range(d->nvars, k->w_lo, k->w_hi);
  range(n, k->h_lo, k->h_hi);
  // pointsC1 covers wires npubs+1 .. nvars-1 (zkey_types.nim:40; zs = witness[npubs+1..], prover.nim:262-264).
  // It is stored wire-aligned -- infinity for the npubs+1 public wires -- so that MSM(witness, C1') ==
  // MSM(zs, C1) and all four witness MSMs share ONE bucket arrangement of the witness.
  std::vector<unsigned char> c1pad((k->w_hi - k->w_lo) * 64, 0);
  for (size_t wI = k->w_lo; wI < k->w_hi; ++wI)
    if (wI > d->npubs) memcpy(&c1pad[(wI - k->w_lo) * 64], (const char*)d->pointsC1 + 64 * (wI - d->npubs - 1), 64);
  TRY(g16_points_register_g1(ctx, (const char*)d->pointsA1 + 64 * k->w_lo, k->w_hi - k->w_lo, &k->A1));
  TRY(g16_points_register_g1(ctx, (const char*)d->pointsB1 + 64 * k->w_lo, k->w_hi - k->w_lo, &k->B1));
  TRY(g16_points_register_g2(ctx, (const char*)d->pointsB2 + 128 * k->w_lo, k->w_hi - k->w_lo, &k->B2));
  TRY(g16_points_register_g1(ctx, c1pad.data(), k->w_hi - k->w_lo, &k->C1));
  TRY(g16_points_register_g1(ctx, (const char*)d->pointsH1 + 64 * k->h_lo, k->h_hi - k->h_lo, &k->H1));
  {   // sparse sets get their own entry lists (see g16_pkey)
    const size_t nw = k->w_hi - k->w_lo;
    k->liveA = g16_points_live_if_sparse(k->A1);
    if (nw && k->B1->d_live && k->B2->d_live) {
      uint32_t* d_cnt = nullptr;
      uint32_t dead = 0;
      TRY(ensure(ctx, ctx->stage_o, 2048));
      d_cnt = (uint32_t*)ctx->stage_o.p;
      if (hipMalloc((void**)&k->liveB, ((nw + 31) / 32 + 1) * 4) != hipSuccess) TRY(G16_ENOMEM);
      if (hipMemsetAsync(d_cnt, 0, 4, ctx->stream) != hipSuccess) TRY(G16_EHIP);
      TRY(g16_bitmap_or_device(ctx, k->liveB, k->B1->d_live, k->B2->d_live, nw, d_cnt));
      if (hipMemcpyAsync(&dead, d_cnt, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess)
        TRY(G16_EHIP);
      k->deadB = dead;
      if (!dead || (size_t)dead * 100 < (size_t)g16_env().inf_compact_pct * nw) {   // dense: share the witness sort
        (void)hipFree(k->liveB);
        k->liveB = nullptr;
      }
    }
  }
  // the A and B entries in row order, rows binned by length (sum order is irrelevant mod r)
  std::vector<uint32_t> vrow;
  try {
    vrow.resize(cs.count ? cs.count : 1);
  } catch (const std::bad_alloc&) {
    ctx->err = "out of host memory";
    g16_pkey_destroy(k);
    return G16_ENOMEM;
  }
  for (size_t e = 0; e < cs.count; ++e) {
    const unsigned char* ent = cs.base + e * cs.stride;
    const uint32_t matrix = rd32(ent), row = rd32(ent + 4), col = rd32(ent + 8);
    if (matrix > 1 || row >= n || col >= d->nvars) {
      // MatrixC entries make the reference's buildABC raise (prover.nim:67)
      ctx->err = "coefficient entry out of range (matrix must be 0=A or 1=B)";
      g16_pkey_destroy(k);
      return G16_EINVAL;
    }
    vrow[e] = 2 * row + matrix;
  }
  k->ncoeffs = cs.count;
  TRY(g16_spmat_create(ctx, 2, (uint32_t)n, cs.count, vrow.data(), 4, (const uint32_t*)(cs.count ? cs.base + 8 : nullptr),
                       cs.stride, cs.count ? cs.base + cs.value_off : nullptr, cs.stride, &k->abc, cs.values_r2));
#undef TRY
  *out = k;
  return G16_OK;
}

extern "C" int32_t g16_pkey_create(g16_ctx* ctx, const g16_pkey_desc* d, g16_pkey** out) {
  if (!ctx) return G16_EINVAL;
  if (!d || !out || (d->ncoeffs && !d->coeffs)) {
    ctx->err = "null argument";
    return G16_EINVAL;
  }
  *out = nullptr;
  static_assert(sizeof(g16_coeff) == 48 && offsetof(g16_coeff, value) == 16, "g16_coeff layout");
  CoeffSource cs;
  cs.base = (const unsigned char*)d->coeffs, cs.count = d->ncoeffs, cs.stride = sizeof(g16_coeff), cs.value_off = 16;
  return pkey_create(ctx, d, cs, out);
}

// the key's coefficients straight from the .zkey file: section 4 as it lies on disk -- u32 count, then count entries of
// { u32 matrix, u32 row, u32 col, 32-byte value in DOUBLE Montgomery form } (files/zkey.nim:169-192; io.nim:134-139
// unmarshalFrWTF).  No host arithmetic: the values go to the device as they are.
extern "C" int32_t g16_pkey_create_zkey(g16_ctx* ctx, const g16_pkey_desc* d, const void* section4, size_t section4_bytes,
                                        g16_pkey** out) {
  if (!ctx) return G16_EINVAL;
  if (!d || !out || !section4 || section4_bytes < 4 || d->coeffs || d->ncoeffs) {
    ctx->err = "bad argument (section 4 of the .zkey in, desc.coeffs = NULL, desc.ncoeffs = 0)";
    return G16_EINVAL;
  }
  *out = nullptr;
  const unsigned char* p = (const unsigned char*)section4;
  const size_t count = rd32(p);
  if (section4_bytes != 4 + count * 44) {
    ctx->err = "unexpected length of the coefficient section (4 + 44 * count bytes)";   // zkey.nim:176 asserts the same
    return G16_EINVAL;
  }
  CoeffSource cs;
  cs.base = p + 4, cs.count = count, cs.stride = 44, cs.value_off = 12, cs.values_r2 = true;
  return pkey_create(ctx, d, cs, out);
}

// Az | Bz | Cz for a witness (device buffers); exposed for tests of the buildABC kernel
// need_cz = false: only Az | Bz (Montgomery); the quotient forms Cz = Az * Bz while its first pass loads (ntt.cuh)
static int32_t build_abc_device(g16_ctx* ctx, const g16_pkey* k, const u256* d_wit, uint32_t wit_mont, u256* d_abc,
                                bool need_cz = true) {
  return g16_spmat_apply(ctx, k->abc, d_wit, wit_mont, d_abc, need_cz);
}

extern "C" int32_t g16_build_abc(g16_ctx* ctx, const g16_pkey* k, const void* witness, uint32_t flags, void* out_abc) {
  if (!ctx) return G16_EINVAL;
  if (!k || !witness || !out_abc || k->device != ctx->device) {
    ctx->err = "bad argument";
    return G16_EINVAL;
  }
  CTX_ENTER(ctx);
  const size_t n = size_t(1) << k->log2n;
  int32_t rc;
  if ((rc = ensure(ctx, ctx->prove, ((size_t)k->nvars + 4 * n) * 32))) return rc;
  u256* d_w = (u256*)ctx->prove.p;
  u256* d_abc = d_w + k->nvars;
  HIPCHK(ctx, hipMemcpyAsync(d_w, witness, (size_t)k->nvars * 32, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = build_abc_device(ctx, k, d_w, (flags & G16_SCALARS_MONT) ? 1u : 0u, d_abc))) return rc;
  HIPCHK(ctx, hipMemcpyAsync(out_abc, d_abc, 3 * n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return G16_OK;
}

// record of the five MSM partials of one rank: XYZZ accumulators (Montgomery), fixed order
//   [0,128) A1 | [128,256) B1 | [256,512) B2 (G2) | [512,640) H1 | [640,768) C1
static constexpr size_t PART_A = 0, PART_B1 = 128, PART_B2 = 256, PART_H = 512, PART_C = 640, PART_BYTES = 768;

static int32_t prove_partials_impl(g16_ctx* ctx, const g16_pkey* k, const void* witness, uint32_t flags,
                                   void* out_partials);

extern "C" int32_t g16_prove_partials(g16_ctx* ctx, const g16_pkey* k, const void* witness, uint32_t flags,
                                      void* out_partials) {
  if (!ctx) return G16_EINVAL;
  if (!k || !witness || !out_partials || k->device != ctx->device) {
    ctx->err = "bad argument";
    return G16_EINVAL;
  }
  CTX_ENTER(ctx);
  const int32_t rc = prove_partials_impl(ctx, k, witness, flags, out_partials);
  // an error exit may leave work queued on the lane streams (e.g. a failed allocation after the witness MSMs were
  // launched): drain all of them before returning, so that the caller -- or the next call's ensure() -- can never
  // free a buffer a lane kernel is still reading
  if (rc != G16_OK) ctx_quiesce(ctx);
  return rc;
}

// ---- the pieces of a proof's launch sequence ---------------------------------------------------------------
struct ProveBufs {
  u256 *d_w, *d_abc, *d_qs;
  char* slots;
};
static int32_t prove_bufs(g16_ctx* ctx, const g16_pkey* k, ProveBufs& b) {
  const size_t n = size_t(1) << k->log2n;
  int32_t rc;
  if ((rc = ensure(ctx, ctx->prove, ((size_t)k->nvars + 4 * n) * 32))) return rc;
  if ((rc = ensure(ctx, ctx->stage_o, 2048))) return rc;
  b.d_w = (u256*)ctx->prove.p;
  b.d_abc = b.d_w + k->nvars;
  b.d_qs = b.d_abc + 3 * n;
  b.slots = (char*)ctx->stage_o.p;
  return G16_OK;
}

// witness -> HBM (main stream); ev_a marks its arrival
static int32_t upload_witness(g16_ctx* ctx, const g16_pkey* k, const void* witness, uint32_t flags, const ProveBufs& b) {
  HIPCHK(ctx, hipMemcpyAsync(b.d_w, witness, (size_t)k->nvars * 32,
                             (flags & G16_SCALARS_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                             ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(b.slots, 0, PART_BYTES, ctx->stream));   // empty range -> XYZZ infinity (all zero)
  HIPCHK(ctx, hipEventRecord(ctx->ev_a, ctx->stream));                // witness resident
  return G16_OK;
}

// C1 and H1 enter the proof only as their sum (pi_c = ... + H + C, prover.nim:301-302), and both point sets are
// registered with the same window, i.e. over the same bucket set: the H accumulation then STARTS from C1's bucket sums
// instead of from infinity, and the pair needs one bucket reduction (reduce1 / reduce2 / fold) instead of two.  The
// C1 slot of the record stays at infinity.  G16_CHAIN_CH=0 restores two separate MSMs.
static bool chain_c_into_h(const g16_pkey* k) {
  return g16_env().chain_ch && k->w_hi > k->w_lo && k->h_hi > k->h_lo && k->C1->cfg() == k->H1->cfg();
}

// The four MSMs that consume the witness (A1, B1, B2, C1: prover.nim:282, 288, 294, 302) on the lane streams.  The
// witness' signed-digit bucket arrangement is computed once (lane 0) and shared; the four accumulate / reduce pipelines
// run on four streams so that their latency-bound tails overlap the other pipelines' accumulation.  Nothing is waited
// for here.  Round 4 measured four other schedules against this one, same box, same session (profiles/r04_ab_*.txt):
//  * A1, B1, C1 as ONE batched launch sequence on one stream (every stage kernel takes blockIdx.y = MSM: three-wide
//    tails, 15 launches and 2 streams less): 110.2 vs 115.2 proofs/s.  Kept as G16_G1_BATCH=1.
//  * one accumulate stream per context with the tails on the lanes (a context then offers one accumulate kernel at a
//    time): 110.7 vs 117.4, and 33 / 10 proofs/s with 4 / 5 proofs in flight (the runtime's cross-stream waits stall).
//  * one accumulate stream for ALL in-flight proofs of the device: 8-12 proofs/s (stalls of 10-45 ms at the waits).
//  * an admission gate (the whole front of a proof -- upload, both sorts, buildABC, quotient -- first, then at most n
//    proofs past the gate, 4-6 in flight): 119.2-120.2 vs 119.5 -- the share of wall time without a resident
//    accumulate kernel falls from 12 % to 8 % (tools/overlap.py) and the throughput does not move.
// The step is bound by the instructions of ALL its kernels; what these schedules rearrange is latency.
// `after` (optional): an event the accumulations wait for in addition to the sort -- the quotient's last kernel
// (G16_QUOTIENT_FIRST / G16_LANES_AFTER_QUOTIENT experiments).
// phase 1: the bucket arrangements of the witness (lane 0; lane 1 for the live pairs of B1 / B2)
static int32_t launch_witness_sorts(g16_ctx* ctx, const g16_pkey* k, uint32_t flags) {
  const uint32_t wflags = (flags & G16_SCALARS_MONT) ? G16_SCALARS_MONT : 0u;
  const size_t nw = k->w_hi - k->w_lo;
  if (!nw) return G16_OK;
  int32_t rc;
  g16_ctx::MsmLane* L = ctx->lane;
  ProveBufs b;
  if ((rc = prove_bufs(ctx, k, b))) return rc;
  const u256* d_wr = b.d_w + k->w_lo;
  HIPCHK(ctx, hipStreamWaitEvent(L[0].stream, ctx->ev_a, 0));
  for (auto& srt : ctx->sort) srt.narrow_tail = true;   // proofs overlap their MSM tails with other work (msm_stage.cuh)
  if ((rc = g16_msm_sort(ctx, L[0].stream, d_wr, wflags, nw, k->A1->cfg(), ctx->sort[0]))) return rc;
  if (k->liveA)   // A1 with many (0,0) points: its own arrangement, behind the shared one
    if ((rc = g16_msm_sort(ctx, L[0].stream, d_wr, wflags, nw, k->A1->cfg(), ctx->sort[2], k->liveA))) return rc;
  HIPCHK(ctx, hipEventRecord(ctx->ev_b, L[0].stream));
  // B1 / B2 with many (0,0) points: their own arrangement of the witness (live pairs only), built on B2's lane
  // while lane 0 arranges the full witness
  if (k->liveB) {
    HIPCHK(ctx, hipStreamWaitEvent(L[1].stream, ctx->ev_a, 0));
    if ((rc = g16_msm_sort(ctx, L[1].stream, d_wr, wflags, nw, k->B2->cfg(), ctx->sort[3], k->liveB))) return rc;
    HIPCHK(ctx, hipEventRecord(ctx->ev_b2, L[1].stream));
  }
  return G16_OK;
}
// phase 2: accumulate + reduce A1, B1, B2, C1 against them
constexpr size_t G2_FIRST_MAX = size_t(1) << 18;
static int32_t launch_witness_msms(g16_ctx* ctx, const g16_pkey* k, const ProveBufs& b, hipEvent_t after) {
  int32_t rc;
  const size_t nw = k->w_hi - k->w_lo;
  if (!nw) return G16_OK;
  g16_ctx::MsmLane* L = ctx->lane;
  const bool batch = g16_env().g1_batch != 0;
  const bool chain = chain_c_into_h(k);
  // lanes of the three G1 MSMs (A1, B1, C1); G16_G1_LANES (read once per process, g16_env) reassigns them
  const int la = batch ? 0 : g16_env().g1_lanes[0], lb = batch ? 0 : g16_env().g1_lanes[1],
            lc = batch ? 0 : g16_env().g1_lanes[2];
  const int nlanes = batch ? 2 : 4;
  const g16_ctx::MsmSort* sortA = k->liveA ? &ctx->sort[2] : &ctx->sort[0];
  const g16_ctx::MsmSort* sortB = k->liveB ? &ctx->sort[3] : &ctx->sort[0];
  // workspaces: the accumulate buffers of lanes 1, 0, 2, 3 serve B2, A1, B1, C1 in every mode
  const g16_msm_run runB2{sortB, &L[1].acc, k->B2->d_tables, nullptr, b.slots + PART_B2, nullptr};
  g16_msm_run runs[3] = {{sortA, &L[0].acc, k->A1->d_tables, nullptr, b.slots + PART_A, nullptr},
                         {sortB, &L[2].acc, k->B1->d_tables, nullptr, b.slots + PART_B1, nullptr},
                         {&ctx->sort[0], &L[3].acc, k->C1->d_tables, nullptr, b.slots + PART_C, nullptr}};
  for (int i = 1; i < nlanes; ++i)   // (lane 1 sorted for itself when B is sparse)
    HIPCHK(ctx, hipStreamWaitEvent(L[i].stream, i == 1 && k->liveB ? ctx->ev_b2 : ctx->ev_b, 0));
  if (k->liveB) HIPCHK(ctx, hipStreamWaitEvent(L[lb].stream, ctx->ev_b2, 0));
  if (k->liveA && la != 0) HIPCHK(ctx, hipStreamWaitEvent(L[la].stream, ctx->ev_b, 0));
  if (after)
    for (int i = 0; i < nlanes; ++i) HIPCHK(ctx, hipStreamWaitEvent(L[i].stream, after, 0));
  // Small witness ranges (the shards of a proof spread over GPUs): the four accumulations together do not fill the
  // GPU, every kernel is a latency chain and B2's -- G2 additions, ~4 x the wave time of G1's -- is the longest: its
  // accumulation goes first, next to C1's only (the H accumulation continues C1's bucket sums: the second longest
  // chain); A1 and B1 then run under B2's reduce / fold tail.
  const bool g2_first = g16_env().g2_first >= 0 ? g16_env().g2_first != 0 : nw <= G2_FIRST_MAX;
  if ((rc = g16_msm_batch(ctx, L[1].stream, 2, &runB2, 1, 1, g2_first ? ctx->ev_g2 : nullptr))) return rc;
  if (g2_first) {
    HIPCHK(ctx, hipStreamWaitEvent(L[la].stream, ctx->ev_g2, 0));
    HIPCHK(ctx, hipStreamWaitEvent(L[lb].stream, ctx->ev_g2, 0));
    if (!chain || g16_env().g2_first == 2) HIPCHK(ctx, hipStreamWaitEvent(L[lc].stream, ctx->ev_g2, 0));
  }
  if (batch) {
    if ((rc = g16_msm_batch(ctx, L[0].stream, 1, runs, 3, chain ? 2 : 3, chain ? ctx->ev_c : nullptr))) return rc;
  } else {
    if ((rc = g16_msm_batch(ctx, L[la].stream, 1, &runs[0], 1, 1, nullptr))) return rc;
    if ((rc = g16_msm_batch(ctx, L[lb].stream, 1, &runs[1], 1, 1, nullptr))) return rc;
    if ((rc = g16_msm_batch(ctx, L[lc].stream, 1, &runs[2], 1, chain ? 0 : 1, chain ? ctx->ev_c : nullptr))) return rc;
  }
  for (int i = 0; i < nlanes; ++i) HIPCHK(ctx, hipEventRecord(L[i].done, L[i].stream));
  return G16_OK;
}

// the H MSM over this key's domain range (prover.nim:301), then join the lanes and hand out the five partials.
// d_qs_slice: the H scalars of [h_lo, h_hi), Montgomery.
static int32_t launch_h_sort(g16_ctx* ctx, const g16_pkey* k, const u256* d_qs_slice) {
  const size_t nh = k->h_hi - k->h_lo;
  ctx->sort[1].narrow_tail = true;
  return nh ? g16_msm_sort(ctx, ctx->stream, d_qs_slice, G16_SCALARS_MONT, nh, k->H1->cfg(), ctx->sort[1]) : G16_OK;
}
static int32_t launch_h_and_collect(g16_ctx* ctx, const g16_pkey* k, const u256* d_qs_slice, uint32_t flags,
                                    const ProveBufs& b, void* out_partials, bool sorted = false) {
  int32_t rc;
  hipStream_t M = ctx->stream;
  const size_t nw = k->w_hi - k->w_lo, nh = k->h_hi - k->h_lo;
  const int nlanes = g16_env().g1_batch ? 2 : 4;
  if (nh) {
    if (!sorted && (rc = launch_h_sort(ctx, k, d_qs_slice))) return rc;
    const bool chain = chain_c_into_h(k);
    const g16_msm_run run{&ctx->sort[1], &ctx->lane[4].acc, k->H1->d_tables, nullptr, b.slots + PART_H,
                          chain ? g16_msm_partial_ptr(ctx->lane[3].acc) : nullptr};
    // G16_CU_SPLIT: the main stream owns a few CUs per XCD only; the H accumulation then runs on the spare lane (the
    // large partition), ordered behind the H sort and joined again below
    hipStream_t HS = g16_env().cu_split ? ctx->lane[4].stream : M;
    if (HS != M) {
      HIPCHK(ctx, hipEventRecord(ctx->ev_q, M));
      HIPCHK(ctx, hipStreamWaitEvent(HS, ctx->ev_q, 0));
    }
    if (chain) HIPCHK(ctx, hipStreamWaitEvent(HS, ctx->ev_c, 0));   // C1's bucket sums are final
    if ((rc = g16_msm_batch(ctx, HS, 1, &run, 1, 1, nullptr))) return rc;
    if (HS != M) {
      HIPCHK(ctx, hipEventRecord(ctx->lane[4].done, HS));
      HIPCHK(ctx, hipStreamWaitEvent(M, ctx->lane[4].done, 0));
    }
  }
  if (nw)
    for (int i = 0; i < nlanes; ++i) HIPCHK(ctx, hipStreamWaitEvent(M, ctx->lane[i].done, 0));
  HIPCHK(ctx, hipMemcpyAsync(out_partials, b.slots, PART_BYTES,
                             (flags & G16_OUT_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
  // G16_NO_HOST_SYNC (device output only): the record is complete in stream order; the caller's next operation on
  // the context's stream (an all-gather enqueued on it, g16_prove_combine) is ordered behind it without a host wait
  if (!((flags & G16_NO_HOST_SYNC) && (flags & G16_OUT_DEVICE))) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return G16_OK;
}

static int32_t prove_partials_impl(g16_ctx* ctx, const g16_pkey* k, const void* witness, uint32_t flags,
                                   void* out_partials) {
  const size_t n = size_t(1) << k->log2n;
  const uint32_t wit_mont = (flags & G16_SCALARS_MONT) ? 1u : 0u;
  ProveBufs b;
  int32_t rc;
  if ((rc = prove_bufs(ctx, k, b))) return rc;
  if ((rc = upload_witness(ctx, k, witness, flags, b))) return rc;
  // Launch order.  Rounds 1-4 enqueued the four witness MSMs first and buildABC + quotient + H behind them on the main
  // stream: with the thread-per-row buildABC of those rounds the other order lost (r02: 107.6-108.5 vs 107.9-110.4
  // proofs/s, 13.1 vs 12.0 ms single proof).  Since round 5 the head of the longest dependency chain -- buildABC (one
  // row-balanced launch) -> quotient (Cz formed on the fly) -> sort(qs) -> H MSM -- goes to the GPU BEFORE the ~60
  // launches of the witness lanes: two sessions, same box, identical proof bytes: single proof 10.61 -> 10.34 and
  // 10.67 -> 10.46 ms, proofs/s 121.5 -> 122.2 and 119.9 -> 120.3 (profiles/r05_ab_quotient_first*.txt).  Holding the
  // lanes back until the quotient is done (G16_LANES_AFTER_QUOTIENT=1) still loses (118.5, 11.4 ms).
  // G16_QUOTIENT_FIRST=0 restores the old order.  Replicated on every rank of a sharded proof unless the caller uses the
  // task-parallel pair g16_prove_partials_begin / _end.
  const int fly = k->log2n >= 1 && g16_env().cz_on_the_fly ? 1 : 0;   // Cz formed by the quotient's first pass
  if (!g16_env().quotient_first) {
    if ((rc = launch_witness_sorts(ctx, k, flags))) return rc;
    if ((rc = launch_witness_msms(ctx, k, b, nullptr))) return rc;
    if ((rc = build_abc_device(ctx, k, b.d_w, wit_mont, b.d_abc, fly == 0))) return rc;
    if ((rc = g16_quotient_device(ctx, b.d_abc, b.d_abc + n, b.d_abc + 2 * n, k->log2n, (int)k->flavour, b.d_qs, fly))) return rc;
    return launch_h_and_collect(ctx, k, b.d_qs + k->h_lo, flags, b, out_partials);
  }
  if ((rc = build_abc_device(ctx, k, b.d_w, wit_mont, b.d_abc, fly == 0))) return rc;
  if ((rc = g16_quotient_device(ctx, b.d_abc, b.d_abc + n, b.d_abc + 2 * n, k->log2n, (int)k->flavour, b.d_qs, fly))) return rc;
  // ... and the bucket arrangement of the H scalars too: its dozen short kernels would otherwise queue, one after the
  // other, behind the GPU-filling accumulate waves of the four witness lanes (measured: 6 ms for a 0.5-ms sort)
  if ((rc = launch_h_sort(ctx, k, b.d_qs + k->h_lo))) return rc;
  hipEvent_t after = nullptr;
  if (g16_env().lanes_after_quotient) {
    HIPCHK(ctx, hipEventRecord(ctx->ev_q, ctx->stream));
    after = ctx->ev_q;
  }
  if ((rc = launch_witness_sorts(ctx, k, flags))) return rc;
  if ((rc = launch_witness_msms(ctx, k, b, after))) return rc;
  return launch_h_and_collect(ctx, k, b.d_qs + k->h_lo, flags, b, out_partials, true);
}

// ---- sharded proof with a task-parallel quotient ------------------------------------------------------------
// The reference runs the three coset pipelines of computeSnarkjsScalarCoeffs as three tasks (prover.nim:167-169).
// Across GPUs each pipeline lives on ONE rank: _begin launches this rank's witness MSMs and computes the pipelines
// named by task_mask (bit 0: A, bit 1: B, bit 2: C) into d_task_out (n Fr per set bit, ascending); the caller
// scatters the [h_lo, h_hi) slices of the three coset vectors to their ranks (nim_groth16_amd/distributed.py: three
// scatters of 32 n / G bytes per destination) while the MSM lanes keep computing; _end forms this rank's H scalars
// A1*B1 - C1 from the received slices (prover.nim:175-176), runs the H MSM over them and hands out the partials.
extern "C" int32_t g16_prove_partials_begin(g16_ctx* ctx, const g16_pkey* k, const void* witness, uint32_t flags,
                                            uint32_t task_mask, void* d_task_out) {
  if (!ctx) return G16_EINVAL;
  if (!k || !witness || k->device != ctx->device || task_mask > 7 || (task_mask && !d_task_out)) {
    ctx->err = "bad argument";
    return G16_EINVAL;
  }
  if (k->flavour != G16_FLAVOUR_SNARKJS) {
    ctx->err = "the task-parallel quotient serves snarkjs-flavour keys (JensGroth needs a 7th transform of the whole "
               "vector: use g16_prove_partials)";
    return G16_EINVAL;
  }
  CTX_ENTER(ctx);
  const size_t n = size_t(1) << k->log2n;
  ProveBufs b;
  int32_t rc = prove_bufs(ctx, k, b);
  if (!rc) rc = upload_witness(ctx, k, witness, flags, b);
  if (!rc && task_mask) {   // this rank's coset pipelines go to the GPU first: every other rank waits for their slices
    rc = build_abc_device(ctx, k, b.d_w, (flags & G16_SCALARS_MONT) ? 1u : 0u, b.d_abc, (task_mask & 4u) != 0);
    u256* out = (u256*)d_task_out;
    for (int v = 0; v < 3 && !rc; ++v)
      if (task_mask & (1u << v)) {
        rc = g16_coset_pipeline_device(ctx, b.d_abc + v * n, k->log2n, out);
        out += n;
      }
  }
  hipEvent_t after = nullptr;
  if (!rc && task_mask && g16_env().quotient_first && g16_env().lanes_after_quotient) {
    if (hipEventRecord(ctx->ev_q, ctx->stream) != hipSuccess) rc = G16_EHIP;
    after = ctx->ev_q;
  }
  if (!rc) rc = launch_witness_sorts(ctx, k, flags);
  if (!rc) rc = launch_witness_msms(ctx, k, b, after);
  if (!rc && !(flags & G16_NO_HOST_SYNC) && hipStreamSynchronize(ctx->stream) != hipSuccess) {   // the task outputs are complete; the lanes run on
    ctx->err = "hipStreamSynchronize failed";
    rc = G16_EHIP;
  }
  if (rc != G16_OK) {
    ctx_quiesce(ctx);
    return rc;
  }
  ctx->shard_begun = k;
  return G16_OK;
}

extern "C" int32_t g16_prove_partials_end(g16_ctx* ctx, const g16_pkey* k, const void* d_a1, const void* d_b1,
                                          const void* d_c1, uint32_t flags, void* out_partials) {
  if (!ctx) return G16_EINVAL;
  const size_t nh = k ? k->h_hi - k->h_lo : 0;
  if (!k || !out_partials || k->device != ctx->device || (nh && (!d_a1 || !d_b1 || !d_c1))) {
    ctx->err = "bad argument";
    return G16_EINVAL;
  }
  if (ctx->shard_begun != k) {
    ctx->err = "g16_prove_partials_end without a matching g16_prove_partials_begin on this context";
    return G16_EINVAL;
  }
  CTX_ENTER_KEEP(ctx);
  ctx->shard_begun = nullptr;
  ProveBufs b;
  int32_t rc = prove_bufs(ctx, k, b);
  if (!rc) rc = g16_abc_pointwise_device(ctx, d_a1, d_b1, d_c1, nh, b.d_qs);
  if (!rc) rc = launch_h_and_collect(ctx, k, b.d_qs, flags, b, out_partials);
  if (rc != G16_OK) ctx_quiesce(ctx);
  return rc;
}

// one workgroup per MSM: res = sum over ranks of that MSM's partial, then canonical affine
// (`res += sync pending[k]`, msm.nim:117-119, across GPUs instead of threads)
static __global__ void prove_combine_kernel(const unsigned char* __restrict__ gathered, uint32_t count,
                                            unsigned char* __restrict__ out) {
  if (threadIdx.x != 0) return;
  const uint32_t b = blockIdx.x;
  if (b == 2) {
    g2_acc r = G2::acc_inf();
    for (uint32_t i = 0; i < count; ++i) G2::add(r, *(const g2_acc*)(gathered + (size_t)i * PART_BYTES + PART_B2));
    *(g2_aff*)(out + 128) = G2::to_affine(r);
  } else {
    const size_t src = b == 0 ? PART_A : b == 1 ? PART_B1 : b == 3 ? PART_H : PART_C;
    const size_t dst = b == 0 ? 0 : b == 1 ? 64 : b == 3 ? 256 : 320;
    g1_acc r = G1::acc_inf();
    for (uint32_t i = 0; i < count; ++i) G1::add(r, *(const g1_acc*)(gathered + (size_t)i * PART_BYTES + src));
    *(g1_aff*)(out + dst) = G1::to_affine(r);
  }
}

extern "C" int32_t g16_prove_combine(g16_ctx* ctx, const g16_pkey* k, const void* partials, size_t count,
                                     uint32_t flags, const void* mask_r, const void* mask_s, g16_proof* out) {
  if (!ctx) return G16_EINVAL;
  if (!k || !partials || !out || k->device != ctx->device || count == 0 || count > 1024) {
    ctx->err = "bad argument";
    return G16_EINVAL;
  }
  CTX_ENTER(ctx);
  int32_t rc;
  if ((rc = ensure(ctx, ctx->stage_p, count * PART_BYTES))) return rc;
  if ((rc = ensure(ctx, ctx->stage_o, 2048))) return rc;
  HIPCHK(ctx, hipMemcpyAsync(ctx->stage_p.p, partials, count * PART_BYTES,
                             (flags & G16_SCALARS_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                             ctx->stream));
  unsigned char* d_res = (unsigned char*)ctx->stage_o.p + 1024;
  KLAUNCH(ctx, "prove_combine", prove_combine_kernel, 5, 64, 0, (const unsigned char*)ctx->stage_p.p, (uint32_t)count,
          d_res);
  struct {
    g1_aff a, b1;
    g2_aff b2;
    g1_aff h, c;
  } res;
  static_assert(sizeof(res) == 384, "slot layout");

  // Everything that depends on the mask and the key alone is computed while the GPU works -- in g16_prove that is the
  // whole proof, which is enqueued without a host wait (prover.nim:267-268, 279-302 regrouped):
  //   pi_a = (alpha1 + r delta1) + A                      pi_b = (beta2 + s delta2) + B2
  //   pi_c = s pi_a + r rho - rs delta1 + H + C           with rho = beta1 + s delta1 + B1
  //        = (s alpha1 + r beta1 + rs delta1) + (s A + r B1) + H + C
  // The same group elements as the reference's order of operations, hence the same canonical affine bytes.
  u256 r = Fr::zero(), s = Fr::zero();
  if (mask_r) memcpy(&r, mask_r, 32);
  if (mask_s) memcpy(&s, mask_s, 32);
  const u256 r_std = Fr::from_mont(r), s_std = Fr::from_mont(s);
  const u256 rs_std = Fr::from_mont(Fr::mul(r, s));
  const g1_aff a_pre = host_add<HG1>(k->alpha1, host_mul<HG1>(r_std, k->delta1));
  const g2_aff b_pre = host_add<HG2>(k->beta2, host_mul<HG2>(s_std, k->delta2));
  const g1_aff c_pre = host_add<HG1>(host_mul2<HG1>(s_std, k->alpha1, r_std, k->beta1), host_mul<HG1>(rs_std, k->delta1));
  // (the copy into pageable host memory makes the host wait for the stream: it comes AFTER the host arithmetic above)
  HIPCHK(ctx, hipMemcpyAsync(&res, d_res, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));

  // what needs the MSM results: three additions and ONE joint double-scalar multiplication
  g1_aff pi_a = host_add<HG1>(a_pre, res.a);
  g2_aff pi_b = host_add<HG2>(b_pre, res.b2);
  g1_aff pi_c = host_add<HG1>(c_pre, host_mul2<HG1>(s_std, res.a, r_std, res.b1));
  pi_c = host_add<HG1>(pi_c, res.h);
  pi_c = host_add<HG1>(pi_c, res.c);
  memcpy(out->pi_a, &pi_a, 64);
  memcpy(out->pi_b, &pi_b, 128);
  memcpy(out->pi_c, &pi_c, 64);
  return G16_OK;
}

extern "C" int32_t g16_prove(g16_ctx* ctx, const g16_pkey* k, const void* witness, uint32_t flags, const void* mask_r,
                             const void* mask_s, g16_proof* out) {
  if (!ctx) return G16_EINVAL;
  if (!k || !witness || !out || k->device != ctx->device) {
    ctx->err = "bad argument";
    return G16_EINVAL;
  }
  if (k->shard_count != 1) {
    ctx->err = "g16_prove needs an unsharded key; use g16_prove_partials + g16_prove_combine";
    return G16_EINVAL;
  }
  CTX_ENTER(ctx);   // before the first allocation: a fresh host thread's current device is 0, not the context's
  int32_t rc;
  if ((rc = ensure(ctx, ctx->stage_o, 2048))) return rc;
  if ((rc = ensure(ctx, ctx->stage_s, PART_BYTES))) return rc;
  // partials stay in HBM (stage_s is free again once the witness has been copied into the prove buffer)
  unsigned char* d_part = (unsigned char*)ctx->stage_s.p;
  // no host wait here: the combine's copy and kernel are ordered behind the record on the main stream, and its
  // mask-only host arithmetic (~0.3 ms) then overlaps the whole proof instead of following it
  if ((rc = g16_prove_partials(ctx, k, witness, flags | G16_OUT_DEVICE | G16_NO_HOST_SYNC, d_part))) return rc;
  rc = g16_prove_combine(ctx, k, d_part, 1, G16_SCALARS_DEVICE, mask_r, mask_s, out);
  if (rc != G16_OK) ctx_quiesce(ctx);   // a failed combine may leave the lanes running
  return rc;
}

// quotient alone, host pointers (replaces computeSnarkjsScalarCoeffs / computeQuotientPointwise)
extern "C" int32_t g16_quotient(g16_ctx* ctx, const void* Az, const void* Bz, const void* Cz, uint32_t log2n,
                                uint32_t flavour, void* out) {
  if (!ctx) return G16_EINVAL;
  if (!Az || !Bz || !Cz || !out || log2n > 27 || flavour > 1) {
    ctx->err = "bad argument";
    return G16_EINVAL;
  }
  CTX_ENTER(ctx);
  const size_t n = size_t(1) << log2n;
  int32_t rc;
  if ((rc = ensure(ctx, ctx->prove, 4 * n * 32))) return rc;
  u256* d = (u256*)ctx->prove.p;
  HIPCHK(ctx, hipMemcpyAsync(d, Az, n * 32, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d + n, Bz, n * 32, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d + 2 * n, Cz, n * 32, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = g16_quotient_device(ctx, d, d + n, d + 2 * n, log2n, (int)flavour, d + 3 * n))) return rc;
  HIPCHK(ctx, hipMemcpyAsync(out, d + 3 * n, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return G16_OK;
}
extern "C" int32_t g16_quotient_dev(g16_ctx* ctx, const void* d_Az, const void* d_Bz, const void* d_Cz, uint32_t log2n,
                                    uint32_t flavour, void* d_out) {
  if (!ctx) return G16_EINVAL;
  if (!d_Az || !d_Bz || !d_Cz || !d_out || log2n > 27 || flavour > 1) {
    ctx->err = "bad argument";
    return G16_EINVAL;
  }
  CTX_ENTER(ctx);
  return g16_quotient_device(ctx, d_Az, d_Bz, d_Cz, log2n, (int)flavour, d_out);
}
