// Row bins of the row-balanced sparse kernel (spmv.hip), shared by host and device code (and by the CPU test shim).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define G16_SPMV_HD __host__ __device__ constexpr
#else
#define G16_SPMV_HD constexpr
#endif

namespace g16 {
constexpr int NBINS = 9;
constexpr uint32_t BLOCK = 256;
// bin b of a virtual row with L terms:  0: L <= 1   1: L == 2   2: L <= 4   2 + g (g = 1..6): 4 * 2^(g-1) < L <= 4 * 2^g
// (the last bin also takes every longer row).  Lanes per virtual row of bin b = 2^bin_glog(b); terms per lane and trip
// = bin_terms(b); a row of bin b is done in ceil(L / (bin_terms(b) << bin_glog(b))) trips: ONE everywhere but in the last bin.
G16_SPMV_HD uint32_t bin_glog(uint32_t b) { return b < 3 ? 0u : b - 2; }
G16_SPMV_HD uint32_t bin_terms(uint32_t b) { return b == 0 ? 1u : b == 1 ? 2u : 4u; }
inline uint32_t bin_of(uint32_t L) {
  if (L <= 1) return 0;
  if (L == 2) return 1;
  uint32_t b = 2;
  while (b < (uint32_t)NBINS - 1 && (4u << bin_glog(b)) < L) ++b;
  return b;
}
}  // namespace g16
