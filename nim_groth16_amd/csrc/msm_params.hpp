// Parameters of one MSM launch sequence (shared by host and device code).
#pragma once
#include <stdint.h>

namespace g16 {
struct MsmParams {
  uint32_t n;            // number of (scalar, point) pairs
  uint32_t c;            // window bits
  uint32_t nwin;         // number of windows  = 254 / c + 1
  uint32_t nbuckets;     // tables: 1 << (c-1) (ONE bucket set shared by all windows); else nwin << (c-1)
  uint32_t seg;          // L: max entries per accumulate task
  uint32_t scalars_mont; // 1: scalars are Montgomery Fr limbs (Nim seq[Fr]); 0: canonical LE (.wtns)
  uint32_t tables;       // 1: points array holds nwin tables [w][i] = 2^(c w) P_i (registered point set);
                         //    the 2^(c w) factor lives in the table, so every window uses the same buckets
  uint32_t max_extra;    // capacity of the extra-segment list
  uint32_t mtab;         // multiplier tables per window (registered sets): 1, or 2 = {1, 2} -- the points array then
                         //    holds tables [m][w][i] = 2^(c w + m) P_i and the bucket set is the CLASS set of msm.cuh
                         //    (msm_class_bucket): 0.67 x the buckets for the same windows
};
// number of buckets of a merged (registered) bucket set
inline uint32_t msm_table_buckets(uint32_t c, uint32_t mtab) {
  const uint32_t h = 1u << (c - 1);
  return mtab == 2 ? h / 2 + h / 8 + h / 32 + h / 64 : h;   // classes b = 4^z u (z = 0, 1, 2; u odd) + the 64 x class
}
constexpr uint32_t MSM_CLASS_SLICES = 32 + 8 + 2 + 1;   // slices of 2^(c-7) buckets: class 0, 1, 2, X
#if defined(__HIPCC__)
#define G16_MSMP_HD __host__ __device__ __forceinline__
#else
#define G16_MSMP_HD inline
#endif
// (partition, slice) of workgroup `bid` of bucket_hist / bucket_place; see msm.cuh
constexpr int BS_SPLIT = 8;
G16_MSMP_HD void bs_block(uint32_t bid, uint32_t nparts, uint32_t& part, uint32_t& q) {
  if (BS_SPLIT == 8 && (nparts & 7u) == 0) {
    const uint32_t j = bid >> 3;
    part = ((j >> 3) << 3) | (bid & 7u);
    q = j & 7u;
  } else {
    part = bid / BS_SPLIT;
    q = bid % BS_SPLIT;
  }
}
// digit magnitude t in [1, 2^(c-1)] -> (class bucket, table selector s): t = 2^s * weight(bucket); see msm.cuh
G16_MSMP_HD uint32_t msm_class_bucket(uint32_t t, uint32_t c, uint32_t& s) {
  const uint32_t tz = (uint32_t)__builtin_ctz(t), h = 1u << (c - 1);
  if (tz >= 6) {
    s = 0;
    return h / 2 + h / 8 + h / 32 + (t >> 6) - 1;
  }
  s = tz & 1u;
  const uint32_t z = tz >> 1, v = (t >> (tz + 1));          // t = 2^tz (2 v + 1)
  const uint32_t base = z == 0 ? 0u : z == 1 ? h / 2 : h / 2 + h / 8;
  return base + v;
}
}  // namespace g16
