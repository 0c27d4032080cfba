// Sparse matrix x vector over Fr, row-balanced: the GPU counterpart of buildABC (reference groth16/prover.nim:56-73)
// and of the sparse column dot products of the fake setup (fake_setup.nim:159-187, 254-256).
//
// The reference walks ZKey.coeffs (files/zkey.nim:169-192: one 44-byte entry per non-zero of A and B) sequentially
// and adds value * witness[col] into Az[row] / Bz[row].  A thread per row (rounds 1-4) is fine for rows of one or two
// terms; a circom circuit (Poseidon, Merkle paths: config 5 of BASELINE.json) has rows of 3-30+ terms next to rows of
// one, and a wave is then as slow as its longest row while its lanes read 32-byte values at strides of a kilobyte.
//
// Layout, built once per key (the matrices are per-circuit constants like the point sets):
//   * a VIRTUAL ROW per (row, matrix): v = 2 row + matrix for a key's A and B (v = row for one matrix); entries in
//     virtual-row order (ptr[v], ptr[v + 1]) with col / val arrays parallel to it.  A and B of one constraint are
//     balanced separately: x5 = x4 * lc has one A term and twenty B terms.
//   * virtual rows sorted into nine BINS by their length L, every wave running ONE code path at ONE trip count:
//       bin 0: L <= 1 (incl. the empty rows of the padded domain)   one lane, one Montgomery product
//       bin 1: L == 2                                               one lane, one fused two-term dot product
//       bin 2: L == 3, 4                                            one lane, one fused four-term dot product
//       bin 2 + g, g = 1..6: 4 * 2^(g-1) < L <= 4 * 2^g              a GROUP of 2^g lanes, four terms per lane
//       (the last bin takes all longer rows, looping).  Lane j of a group takes entries j, j + G, j + 2G, j + 3G:
//     every load instruction of a group reads G consecutive entries.
//   * a lane's (up to) four products are ONE Montgomery dot product (Fr::mul4: 4 x 64 multiply-adds + one 64-mad
//     reduction instead of four); the group then adds its lanes' sums with log2 G cross-lane steps (ds_bpermute, no
//     LDS memory) and lane 0 writes the sum.
//   * buildABC's pointwise Cz = Az * Bz (prover.nim:69-72) is NOT part of this kernel: inside a proof the quotient's
//     first NTT pass forms it while loading (ntt.cuh `mul_src`), so Cz never exists in HBM; g16_build_abc and the
//     task-parallel quotient of a sharded proof run the streaming kernel abc_pointwise_cz instead.
//   * a standard-form witness (raw .wtns values) is multiplied as it is.  With a value dictionary the kernel then reads
//     a second table holding v R^2 (the double-Montgomery form zkey files store, io.nim:134-139): (v R^2) w / R = v w R,
//     the Montgomery sum directly; without one the sums come out in standard form and abc_pointwise_cz converts them,
//     once per row, not once per entry.
//   * value DICTIONARY: circuit coefficients come from a small set (+-1, MDS entries, round constants); when a key's
//     non-zeros hold <= 65536 distinct values the entry stream is (col, value index) = 8 bytes instead of 36 and the
//     value table stays in L2.
#include <algorithm>
#include <new>
#include <unordered_map>

#include "g16_internal.hpp"
#include "ff.cuh"
#include "spmv_params.hpp"

using namespace g16;

namespace {

struct SpmvBins {
  uint32_t row_off[NBINS + 1];   // bin b: rows[row_off[b] .. row_off[b + 1])
  uint32_t blk_off[NBINS + 1];   // ... served by workgroups blk_off[b] .. blk_off[b + 1), 256 >> bin_glog(b) rows each
};

__device__ __forceinline__ u256 shfl_xor_u256(const u256& a, int off) {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = (uint32_t)__shfl_xor((int)a.v[i], off, 64);
  return r;
}

template <bool DICT>
__device__ __forceinline__ void load_entry(uint32_t idx, const uint32_t* __restrict__ col, const u256* __restrict__ val,
                                           const uint32_t* __restrict__ vidx, const u256* __restrict__ x, u256& v, u256& w) {
  v = DICT ? val[vidx[idx]] : val[idx];
  w = x[col[idx]];
}

// sum over the entries [s, e) this lane owns (lane, lane + G, ...) of val * x[col]; K = the bin's terms per trip
template <bool DICT, int K>
__device__ __forceinline__ u256 lane_dot(uint32_t s, uint32_t e, uint32_t lane, uint32_t g,
                                         const uint32_t* __restrict__ col, const u256* __restrict__ val,
                                         const uint32_t* __restrict__ vidx, const u256* __restrict__ x) {
  const uint32_t G = 1u << g;
  u256 a = Fr::zero();
  for (uint32_t i = s + lane; i < e; i += K * G) {
    u256 v[K], w[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const uint32_t idx = i + j * G;
      if (j == 0 || idx < e) {
        load_entry<DICT>(idx, col, val, vidx, x, v[j], w[j]);
      } else {
        v[j] = Fr::zero();
        w[j] = Fr::zero();
      }
    }
    u256 t;
    if constexpr (K == 1) t = Fr::mul(v[0], w[0]);
    else if constexpr (K == 2) t = Fr::mul2(v[0], w[0], v[1], w[1]);
    else t = Fr::mul4(v[0], w[0], v[1], w[1], v[2], w[2], v[3], w[3]);
    a = Fr::add(a, t);
  }
  return a;
}

// y[(v % NMAT) * n + v / NMAT] = sum over the entries of virtual row v of val * x[col]
template <int NMAT, bool DICT>
__global__ void __launch_bounds__(BLOCK) spmv_binned(SpmvBins bins, const uint32_t* __restrict__ ptr,
                                                     const uint32_t* __restrict__ col, const u256* __restrict__ val,
                                                     const uint32_t* __restrict__ vidx, const u256* __restrict__ x,
                                                     const uint32_t* __restrict__ rows, uint32_t n,
                                                     u256* __restrict__ out) {
  uint32_t b = 0;
  while (blockIdx.x >= bins.blk_off[b + 1]) ++b;   // wave-uniform: <= 8 steps
  const uint32_t g = bin_glog(b), G = 1u << g;
  const uint32_t slot = (blockIdx.x - bins.blk_off[b]) * (BLOCK >> g) + (threadIdx.x >> g);
  const uint32_t lane = threadIdx.x & (G - 1);
  const bool live = slot < bins.row_off[b + 1] - bins.row_off[b];
  const uint32_t v = live ? rows[bins.row_off[b] + slot] : 0u;
  const uint32_t s = live ? ptr[v] : 0u;
  const uint32_t e = live ? ptr[v + 1] : 0u;
  u256 a;
  if (b == 0) a = lane_dot<DICT, 1>(s, e, lane, g, col, val, vidx, x);
  else if (b == 1) a = lane_dot<DICT, 2>(s, e, lane, g, col, val, vidx, x);
  else a = lane_dot<DICT, 4>(s, e, lane, g, col, val, vidx, x);
  for (uint32_t off = G >> 1; off; off >>= 1) a = Fr::add(a, shfl_xor_u256(a, (int)off));   // every lane takes part
  if (live && lane == 0) out[(size_t)(v % NMAT) * n + v / NMAT] = a;
}

// Cz = Az * Bz (prover.nim:69-72); sums_mont == 0: the sums are in standard form (a .wtns witness on plain values): Az,
// Bz to Montgomery form first; write_cz == 0: only that conversion
__global__ void __launch_bounds__(BLOCK) abc_pointwise_cz(u256* __restrict__ abc, uint32_t sums_mont, uint32_t write_cz,
                                                          uint32_t n) {
  const uint32_t r = blockIdx.x * BLOCK + threadIdx.x;
  if (r >= n) return;
  u256 a = abc[r], b = abc[(size_t)n + r];
  if (!sums_mont) {
    a = Fr::to_mont(a);
    b = Fr::to_mont(b);
    abc[r] = a;
    abc[(size_t)n + r] = b;
  }
  if (write_cz) abc[2 * (size_t)n + r] = Fr::mul(a, b);
}

// out[i] = in[i] * R (c R -> c R^2) or in[i] / R (c R^2 -> c R); in == out is allowed
__global__ void __launch_bounds__(BLOCK) values_rescale(const u256* in, u256* out, size_t nd, uint32_t divide) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < nd) out[i] = divide ? Fr::from_mont(in[i]) : Fr::to_mont(in[i]);
}

struct Key256 {
  uint64_t w[4];
  bool operator==(const Key256& o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2] && w[3] == o.w[3]; }
};
struct Key256Hash {
  size_t operator()(const Key256& k) const {
    uint64_t h = k.w[0] * 0x9E3779B97F4A7C15ull;
    h ^= (k.w[1] + 0xBF58476D1CE4E5B9ull) * 0x94D049BB133111EBull;
    h ^= (k.w[2] >> 7) ^ (k.w[3] * 0xD6E8FEB86659FD93ull);
    return (size_t)(h ^ (h >> 29));
  }
};

}  // namespace

struct g16_spmat {
  int device = 0;
  uint32_t nmat = 1, nrows = 0;
  size_t nnz = 0, ndict = 0;   // ndict > 0: d_val holds the dictionary, d_vidx the per-entry indices
  uint32_t *d_ptr = nullptr, *d_col = nullptr, *d_vidx = nullptr, *d_rows = nullptr;
  u256* d_val = nullptr;
  u256* d_val2 = nullptr;      // dictionary only: the values times R (for standard-form x: the sum is Montgomery)
  SpmvBins bins;
};

void g16_spmat_destroy(g16_spmat* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  for (void* p : {(void*)m->d_ptr, (void*)m->d_col, (void*)m->d_vidx, (void*)m->d_rows, (void*)m->d_val, (void*)m->d_val2})
    if (p) (void)hipFree(p);
  delete m;
}

void g16_spmat_info(const g16_spmat* m, size_t out[1 + 9]) {
  out[0] = m ? m->ndict : 0;
  for (int b = 0; b < NBINS; ++b) out[1 + b] = m ? m->bins.row_off[b + 1] - m->bins.row_off[b] : 0;
}

// vrow[i] = nmat * row + matrix of entry i (< nmat * nrows, checked by the caller), col[i], val = 32 bytes at
// val_base + i * val_stride.  The caller has made ctx's device current (CTX_ENTER).
static int32_t spmat_create(g16_ctx* ctx, uint32_t nmat, uint32_t nrows, size_t nnz, const uint32_t* vrow,
                            size_t vrow_stride, const uint32_t* col, size_t col_stride, const void* val_base,
                            size_t val_stride, g16_spmat** out, bool values_r2);
// values_r2: the 32-byte values are c R^2 (the double-Montgomery form of a .zkey's section 4) instead of c R
int32_t g16_spmat_create(g16_ctx* ctx, uint32_t nmat, uint32_t nrows, size_t nnz, const uint32_t* vrow,
                         size_t vrow_stride, const uint32_t* col, size_t col_stride, const void* val_base,
                         size_t val_stride, g16_spmat** out, bool values_r2) {
  *out = nullptr;
  try {   // the host-side arrangement allocates O(nnz) memory: no exception may cross the C ABI
    return spmat_create(ctx, nmat, nrows, nnz, vrow, vrow_stride, col, col_stride, val_base, val_stride, out, values_r2);
  } catch (const std::bad_alloc&) {
    ctx->err = "out of host memory while arranging the sparse matrix";
    return G16_ENOMEM;
  }
}
static int32_t spmat_create(g16_ctx* ctx, uint32_t nmat, uint32_t nrows, size_t nnz, const uint32_t* vrow,
                            size_t vrow_stride, const uint32_t* col, size_t col_stride, const void* val_base,
                            size_t val_stride, g16_spmat** out, bool values_r2) {
  if (nnz >= (size_t(1) << 32) || (size_t)nmat * nrows >= (size_t(1) << 32) - 1) {
    ctx->err = "sparse matrix too large (entry offsets are 32-bit)";
    return G16_EINVAL;
  }
  auto at32 = [](const uint32_t* p, size_t stride, size_t i) {
    uint32_t v;
    memcpy(&v, (const char*)p + i * stride, 4);   // a .zkey's 44-byte entries are not 4-byte aligned
    return v;
  };
  const size_t nv = (size_t)nmat * nrows;
  std::vector<uint32_t> ptr(nv + 1, 0);
  for (size_t i = 0; i < nnz; ++i) ptr[at32(vrow, vrow_stride, i) + 1]++;
  for (size_t v = 0; v < nv; ++v) ptr[v + 1] += ptr[v];
  std::vector<uint32_t> cols(nnz ? nnz : 1), order(nnz ? nnz : 1);
  {
    std::vector<uint32_t> cur(ptr.begin(), ptr.end() - 1);
    for (size_t i = 0; i < nnz; ++i) {
      const uint32_t p = cur[at32(vrow, vrow_stride, i)]++;
      cols[p] = at32(col, col_stride, i);
      order[p] = (uint32_t)i;
    }
  }
  // dictionary of the values (<= 65536 distinct ones and at least 8 entries per distinct value), else plain values
  constexpr size_t DICT_MAX = 65536;
  std::vector<uint32_t> vidx;
  std::vector<u256> vals;
  bool dict = nnz >= 1024 && g16_env().abc_dict != 0;
  if (dict) {
    std::unordered_map<Key256, uint32_t, Key256Hash> map;
    map.reserve(4096);
    vidx.resize(nnz);
    for (size_t p = 0; p < nnz && dict; ++p) {
      Key256 k;
      memcpy(&k, (const char*)val_base + (size_t)order[p] * val_stride, 32);
      auto it = map.find(k);
      if (it == map.end()) {
        if (map.size() >= DICT_MAX || map.size() * 8 > nnz) {
          dict = false;
          break;
        }
        it = map.emplace(k, (uint32_t)map.size()).first;
        vals.resize(map.size());
        memcpy(&vals[it->second], &k, 32);
      }
      vidx[p] = it->second;
    }
  }
  if (!dict) {
    vidx.clear();
    vals.resize(nnz ? nnz : 1);
    for (size_t p = 0; p < nnz; ++p) memcpy(&vals[p], (const char*)val_base + (size_t)order[p] * val_stride, 32);
  }
  // bins by the length of a virtual row
  std::vector<uint8_t> bin(nv ? nv : 1);
  std::vector<uint32_t> rows(nv ? nv : 1);
  g16_spmat* m = new (std::nothrow) g16_spmat();   // (no allocation that can throw follows)
  if (!m) return G16_ENOMEM;
  m->device = ctx->device;
  m->nmat = nmat, m->nrows = nrows, m->nnz = nnz, m->ndict = dict ? vals.size() : 0;
  uint32_t cnt[NBINS] = {0};
  for (size_t v = 0; v < nv; ++v) {
    const uint32_t b = bin_of(ptr[v + 1] - ptr[v]);
    bin[v] = (uint8_t)b;
    cnt[b]++;
  }
  m->bins.row_off[0] = m->bins.blk_off[0] = 0;
  for (int b = 0; b < NBINS; ++b) {
    m->bins.row_off[b + 1] = m->bins.row_off[b] + cnt[b];
    const uint32_t per = BLOCK >> bin_glog(b);
    m->bins.blk_off[b + 1] = m->bins.blk_off[b] + (cnt[b] + per - 1) / per;
  }
  {
    uint32_t cur[NBINS];
    for (int b = 0; b < NBINS; ++b) cur[b] = m->bins.row_off[b];
    for (size_t v = 0; v < nv; ++v) rows[cur[bin[v]]++] = (uint32_t)v;
  }
  auto up = [&](void** dst, const void* src, size_t bytes) -> int32_t {
    if (hipMalloc(dst, bytes ? bytes : 4) != hipSuccess) {
      ctx->err = "hipMalloc(sparse matrix) failed";
      return G16_ENOMEM;
    }
    if (bytes && hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) {
      ctx->err = "hipMemcpy(sparse matrix) failed";
      return G16_EHIP;
    }
    return G16_OK;
  };
  int32_t rc = up((void**)&m->d_ptr, ptr.data(), ptr.size() * 4);
  if (!rc) rc = up((void**)&m->d_col, cols.data(), nnz * 4);
  if (!rc) rc = up((void**)&m->d_val, vals.data(), (dict ? vals.size() : nnz) * 32);
  if (!rc && dict) rc = up((void**)&m->d_vidx, vidx.data(), nnz * 4);
  // d_val holds what the host handed over: c R, or (values_r2) c R^2.  The kernel wants c R in d_val and, with a
  // dictionary, c R^2 in d_val2: one small launch rescales in the direction that is missing.
  const size_t nvals = dict ? vals.size() : nnz;
  auto rescale = [&](const u256* in, u256* o, uint32_t divide) {
    if (!nvals) return;
    hipLaunchKernelGGL(values_rescale, dim3((uint32_t)((nvals + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, ctx->stream, in, o,
                       nvals, divide);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
      ctx->err = "coefficient rescaling failed";
      rc = G16_EHIP;
    }
  };
  if (!rc && dict) {
    if (hipMalloc((void**)&m->d_val2, vals.size() * 32) != hipSuccess) {
      ctx->err = "hipMalloc(sparse matrix) failed";
      rc = G16_ENOMEM;
    } else if (values_r2) {   // the file's bytes ARE the second table
      if (hipMemcpy(m->d_val2, m->d_val, vals.size() * 32, hipMemcpyDeviceToDevice) != hipSuccess) rc = G16_EHIP;
      if (!rc) rescale(m->d_val2, m->d_val, 1);
    } else {
      rescale(m->d_val, m->d_val2, 0);
    }
  } else if (!rc && values_r2) {
    rescale(m->d_val, m->d_val, 1);   // plain values: c R^2 -> c R in place
  }
  if (!rc) rc = up((void**)&m->d_rows, rows.data(), nv * 4);
  if (rc) {
    g16_spmat_destroy(m);
    return rc;
  }
  *out = m;
  return G16_OK;
}

// out = (nmat == 2 ? Az | Bz | Cz : y), device pointers, on the context's main stream.  x_mont = 0: x in standard form
// (nmat == 2 only).  need_cz = false: the caller forms Cz itself (the quotient's first pass): Cz is not written.
int32_t g16_spmat_apply(g16_ctx* ctx, const g16_spmat* m, const void* d_x, uint32_t x_mont, void* d_out, bool need_cz) {
  const uint32_t nblk = m->bins.blk_off[NBINS];
  if (!nblk) return G16_OK;
  // standard-form x: the dictionary's second table (v R) makes the sums Montgomery; plain values leave them standard
  const u256* val = (m->ndict && !x_mont) ? m->d_val2 : m->d_val;
  const bool sums_mont = x_mont || m->ndict;
#define SPMV_LAUNCH(NM, DI)                                                                                          \
  KLAUNCH(ctx, NM == 2 ? "abc_spmv" : "spmv", (spmv_binned<NM, DI>), nblk, BLOCK, 0, m->bins, m->d_ptr, m->d_col,     \
          val, m->d_vidx, (const u256*)d_x, m->d_rows, m->nrows, (u256*)d_out)
  if (m->nmat == 2) {
    if (m->ndict) SPMV_LAUNCH(2, true);
    else SPMV_LAUNCH(2, false);
    if (need_cz || !sums_mont)
      KLAUNCH(ctx, "abc_cz", abc_pointwise_cz, (m->nrows + BLOCK - 1) / BLOCK, BLOCK, 0, (u256*)d_out,
              sums_mont ? 1u : 0u, need_cz ? 1u : 0u, m->nrows);
  } else {
    if (m->ndict) SPMV_LAUNCH(1, true);
    else SPMV_LAUNCH(1, false);
  }
#undef SPMV_LAUNCH
  HIPCHK(ctx, hipGetLastError());
  return G16_OK;
}

// y = M x from triplets, host pointers: see include/g16hip.h
extern "C" int32_t g16_spmv_fr(g16_ctx* ctx, const uint32_t* row, const uint32_t* col, const void* val, size_t nnz,
                               const void* x, size_t ncols, size_t nrows, void* y) {
  if (!ctx) return G16_EINVAL;
  if ((nnz && (!row || !col || !val)) || (ncols && !x) || (nrows && !y) || nrows >= (size_t(1) << 31) ||
      ncols >= (size_t(1) << 32)) {
    ctx->err = "bad argument";
    return G16_EINVAL;
  }
  for (size_t i = 0; i < nnz; ++i)
    if (row[i] >= nrows || col[i] >= ncols) {
      ctx->err = "sparse entry out of range";
      return G16_EINVAL;
    }
  if (!nrows) return G16_OK;
  CTX_ENTER(ctx);
  g16_spmat* m = nullptr;
  int32_t rc = g16_spmat_create(ctx, 1, (uint32_t)nrows, nnz, row, 4, col, 4, val, 32, &m);
  if (rc) return rc;
  void *d_x = nullptr, *d_y = nullptr;
  auto done = [&](int32_t code) {
    (void)hipStreamSynchronize(ctx->stream);
    if (d_x) (void)hipFree(d_x);
    if (d_y) (void)hipFree(d_y);
    g16_spmat_destroy(m);
    return code;
  };
  if (hipMalloc(&d_x, (ncols ? ncols : 1) * 32) != hipSuccess || hipMalloc(&d_y, nrows * 32) != hipSuccess) {
    ctx->err = "hipMalloc(spmv vectors) failed";
    return done(G16_ENOMEM);
  }
  if (ncols && hipMemcpyAsync(d_x, x, ncols * 32, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
    ctx->err = "hipMemcpy(x) failed";
    return done(G16_EHIP);
  }
  if ((rc = g16_spmat_apply(ctx, m, d_x, 1, d_y))) return done(rc);
  if (hipMemcpyAsync(y, d_y, nrows * 32, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) {
    ctx->err = "spmv: copy back failed";
    return done(G16_EHIP);
  }
  return done(G16_OK);
}
