// Test-only: compiles the *device* field/curve headers with g++ so their formulas can be checked
// against the oracle on a machine without a GPU.  Not part of the product library.
#include "../../nim_groth16_amd/csrc/ec29.cuh"
#include "../../nim_groth16_amd/csrc/pairing.cuh"
#include "../../nim_groth16_amd/csrc/msm_params.hpp"
#include "../../nim_groth16_amd/csrc/spmv_params.hpp"
#include <cstring>
#include <vector>
using namespace g16;

template <class T> static T ld(const void* p) { T t; std::memcpy(&t, p, sizeof(T)); return t; }
template <class T> static void st(void* p, const T& t) { std::memcpy(p, &t, sizeof(T)); }

// the reduced-radix accumulate path (ec29.cuh): op 0: acc += q_i ; op 1: acc += -(-q_i) through the sign flag
template <class C>
static void sum29(int op, const void* pts, int n, void* out) {
  using E = Ec29<C>;
  typename E::Acc acc = E::acc_inf();
  const char* p = (const char*)pts;
  for (int i = 0; i < n; ++i) {
    typename C::Aff q = ld<typename C::Aff>(p + sizeof(typename C::Aff) * i);
    if (op == 1) q = C::neg(q);
    const typename E::Tab t = E::tab_from_std(q);   // through the packed table entry
    E::madd(acc, &t, op == 1 ? 1u : 0u);
  }
  st(out, C::to_affine(E::to_std(acc)));
}

// op 2: general XYZZ += XYZZ additions in the reduced-radix field: ((q0 + q1) + (q2 + q3)) + ... pairwise tree, so that
// both operands are genuine accumulators
template <class C>
static void tree29(const void* pts, int n, void* out) {
  using E = Ec29<C>;
  std::vector<typename E::Acc> v;
  const char* p = (const char*)pts;
  for (int i = 0; i < n; ++i) {
    typename E::Acc a = E::acc_inf();
    const typename E::Tab t = E::tab_from_std(ld<typename C::Aff>(p + sizeof(typename C::Aff) * i));
    E::madd(a, &t, 0u);
    v.push_back(a);
  }
  while (v.size() > 1) {
    std::vector<typename E::Acc> w;
    for (size_t i = 0; i + 1 < v.size(); i += 2) {
      typename E::Acc a = v[i];
      E::add(a, v[i + 1]);
      w.push_back(a);
    }
    if (v.size() & 1) w.push_back(v.back());
    v.swap(w);
  }
  st(out, C::to_affine(v.empty() ? C::acc_inf() : E::to_std(v[0])));
}

extern "C" {
// bs_block (msm_params.hpp): every (partition, slice) exactly once, and all slices of a partition on one XCD (same
// block index modulo 8) whenever the partition count is a multiple of 8.  -> 0 if so
uint32_t shim_bs_block_check(uint32_t nparts) {
  std::vector<uint8_t> seen((size_t)nparts * BS_SPLIT, 0);
  std::vector<int> xcd(nparts, -1);
  for (uint32_t b = 0; b < nparts * BS_SPLIT; ++b) {
    uint32_t part = ~0u, q = ~0u;
    bs_block(b, nparts, part, q);
    if (part >= nparts || q >= (uint32_t)BS_SPLIT || seen[(size_t)part * BS_SPLIT + q]++) return 1 + b;
    if ((nparts & 7u) == 0) {
      if (xcd[part] < 0) xcd[part] = (int)(b & 7u);
      else if (xcd[part] != (int)(b & 7u)) return 1 + b;
    }
  }
  return 0;
}

// the row bins of the row-balanced sparse kernel (spmv_params.hpp): -> 0, or 1 + the first row length that breaks a rule
uint32_t shim_spmv_bins_check(uint32_t max_len) {
  uint32_t prev = 0;
  for (uint32_t L = 0; L <= max_len; ++L) {
    const uint32_t b = bin_of(L);
    if (b >= (uint32_t)NBINS || b < prev) return 1 + L;                       // monotone in L
    prev = b;
    const uint32_t lanes = 1u << bin_glog(b), per_trip = bin_terms(b) * lanes;
    if (lanes > 64 || (BLOCK % lanes) != 0) return 1 + L;
    if (b + 1 < (uint32_t)NBINS && L > per_trip) return 1 + L;               // one trip everywhere but in the last bin
    if (b >= 3 && L <= per_trip / 2) return 1 + L;                           // groups are at least half full
    if (b == 0 && L > 1) return 1 + L;
    if (b == 1 && L != 2) return 1 + L;
    if (b == 2 && (L < 3 || L > 4)) return 1 + L;
  }
  return 0;
}

// the class bucket set of registered point sets with two multiplier tables (msm_params.hpp): every digit magnitude
// t in [1, 2^(c-1)] must land in a bucket whose weight (from the class layout, restated here) times 2^s is t, every
// bucket must be hit once or twice, and none may lie outside the set.  -> 0, or the first offending t
uint32_t shim_class_buckets_check(uint32_t c) {
  const uint32_t h = 1u << (c - 1), nb = msm_table_buckets(c, 2);
  std::vector<uint8_t> hits(nb, 0);
  auto weight = [&](uint32_t id) -> uint64_t {
    if (id < h / 2) return 2ull * id + 1;
    if (id < h / 2 + h / 8) return 4ull * (2ull * (id - h / 2) + 1);
    if (id < h / 2 + h / 8 + h / 32) return 16ull * (2ull * (id - h / 2 - h / 8) + 1);
    return 64ull * (id - h / 2 - h / 8 - h / 32 + 1);
  };
  for (uint32_t t = 1; t <= h; ++t) {
    uint32_t s = 9;
    const uint32_t id = msm_class_bucket(t, c, s);
    if (id >= nb || s > 1 || (weight(id) << s) != t) return t;
    if (++hits[id] > 2) return t;
  }
  for (uint32_t id = 0; id < nb; ++id)
    if (!hits[id]) return 0x80000000u | id;
  return nb == h / 2 + h / 8 + h / 32 + h / 64 && nb % (MSM_CLASS_SLICES) == 0 && nb / MSM_CLASS_SLICES == (1u << (c - 7)) ? 0 : 1;
}

// op: 0 add 1 sub 2 mul 3 sqr 4 neg 5 dbl 6 div2 7 inv 8 from_mont 9 to_mont ; field: 0 Fp 1 Fr
void shim_field_op(int field, int op, const void* a, const void* b, void* r) {
  u256 x = ld<u256>(a), y = ld<u256>(b), z;
  auto run = [&](auto F) {
    using FF = decltype(F);
    switch (op) {
      case 0: z = FF::add(x, y); break;
      case 1: z = FF::sub(x, y); break;
      case 2: z = FF::mul(x, y); break;
      case 3: z = FF::sqr(x); break;
      case 4: z = FF::neg(x); break;
      case 5: z = FF::dbl(x); break;
      case 6: z = FF::div2(x); break;
      case 7: z = FF::inv(x); break;
      case 8: z = FF::from_mont(x); break;
      default: z = FF::to_mont(x); break;
    }
  };
  if (field == 0) run(Fp{}); else run(Fr{});
  st(r, z);
}
// op: 0 add 1 sub 2 mul 3 sqr 4 neg 7 inv
void shim_fp2_op(int op, const void* a, const void* b, void* r) {
  fp2_t x = ld<fp2_t>(a), y = ld<fp2_t>(b), z;
  switch (op) {
    case 0: z = Fp2::add(x, y); break;
    case 1: z = Fp2::sub(x, y); break;
    case 2: z = Fp2::mul(x, y); break;
    case 3: z = Fp2::sqr(x); break;
    case 4: z = Fp2::neg(x); break;
    default: z = Fp2::inv(x); break;
  }
  st(r, z);
}
// sums n affine points with madd (op 0), or with XYZZ+XYZZ add of from_affine (op 1), or as
// k*P via mul_small on the first point (op 2, k = n); returns the canonical affine result
void shim_g1_sum(int op, const void* pts, int n, void* out) {
  g1_acc acc = G1::acc_inf();
  const char* p = (const char*)pts;
  if (op == 2) {
    acc = G1::mul_small(G1::from_affine(ld<g1_aff>(p)), (uint32_t)n);
  } else {
    for (int i = 0; i < n; ++i) {
      g1_aff q = ld<g1_aff>(p + 64 * i);
      if (op == 0) G1::madd(acc, q); else G1::add(acc, G1::from_affine(q));
    }
  }
  st(out, G1::to_affine(acc));
}
void shim_g1_sum29(int op, const void* pts, int n, void* out) { if (op == 2) tree29<G1>(pts, n, out); else sum29<G1>(op, pts, n, out); }
void shim_g2_sum29(int op, const void* pts, int n, void* out) { if (op == 2) tree29<G2>(pts, n, out); else sum29<G2>(op, pts, n, out); }
// op 0: to_std(mul(from_std a, from_std b))  1: to_std(sqr(from_std a))  2: to_std(from_std a)
// 3: to_std(dot2(a,b,a,b)) = 2ab
void shim_f29_op(int op, const void* a, const void* b, void* r) {
  fe29 x = Fp29::from_std(ld<u256>(a)), y = Fp29::from_std(ld<u256>(b)), z;
  switch (op) {
    case 0: z = Fp29::mul(x, y); break;
    case 1: z = Fp29::sqr(x); break;
    case 2: z = x; break;
    default: z = Fp29::dot2(x, y, x, y); break;
  }
  st(r, Fp29::to_std(z));
}
void shim_g2_sum(int op, const void* pts, int n, void* out) {
  g2_acc acc = G2::acc_inf();
  const char* p = (const char*)pts;
  if (op == 2) {
    acc = G2::mul_small(G2::from_affine(ld<g2_aff>(p)), (uint32_t)n);
  } else {
    for (int i = 0; i < n; ++i) {
      g2_aff q = ld<g2_aff>(p + 128 * i);
      if (op == 0) G2::madd(acc, q); else G2::add(acc, G2::from_affine(q));
    }
  }
  st(out, G2::to_affine(acc));
}
// op 0: Miller loop f_{6x^2,Q}(P)   1: full pairing   2: final exponentiation of the Fp12 element at `p`
// out: 6 x Fp2 (384 bytes, Montgomery form), coefficient k belongs to w^k
void shim_pairing(int op, const void* p, const void* q, void* out) {
  fp12_t f;
  if (op == 2) {
    Pairing::final_exp(f, ld<fp12_t>(p));
  } else {
    Pairing::miller(f, ld<g1_aff>(p), ld<g2_aff>(q));
    if (op == 1) { fp12_t t = f; Pairing::final_exp(f, t); }
  }
  st(out, f);
}
}
