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
};
}  // namespace g16
