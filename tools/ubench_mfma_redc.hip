// Micro-benchmark for ONE question (optional #8): half of a Montgomery multiplication in the 9 x 29-bit field
// is the product m * p against the CONSTANT modulus -- a contraction with a fixed Toeplitz matrix.  Could the matrix
// cores take that half?
//
// To let MFMA do it the reduction must be de-interleaved: m = (T mod 2^261) * (-1/p) mod 2^261 first (another constant
// product), then (T + m p) / 2^261 -- two constant products of 261-bit operands per multiplication.  In 8-bit digits
// (v_mfma_i32_*_i8) an operand is 33 digits, a product 66 columns; with the data as the B operand (one element per
// column n = lane % 32, its digits along K) and the Toeplitz digits of the constant as A, a wave's 64 elements need, per
// constant product, 2 halves x 3 row blocks of 32 output columns x 2 K-steps of 32 digits = 12 v_mfma_i32_32x32x32_i8,
// i.e. 24 per Montgomery multiplication -- before any digit conversion (29-bit limbs -> 8-bit digits in the MFMA operand
// layout, permlane32 swaps to feed the upper half-wave) and before carry recovery from 66 int32 columns.
// This program times ONLY those 24 MFMA instructions per wave (a LOWER bound of the matrix-core path) against what they
// would replace: 81 + 9 v_mad_u64_u32 of the interleaved VALU reduction (+ the 45 the de-interleaved form adds are NOT
// charged to the VALU side).  Cycles are real shader cycles (s_memtime), at 1 and at 4 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma_redc.hip -o tools/ubench_mfma_redc && ./tools/ubench_mfma_redc
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                             \
  do {                                                                                       \
    hipError_t e = (x);                                                                      \
    if (e != hipSuccess) {                                                                   \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);  \
      exit(1);                                                                               \
    }                                                                                        \
  } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// MODE 0: 24 MFMA per "multiplication" (6 accumulator blocks x 2 K-steps x 2 constant products)
// MODE 1: 90 v_mad_u64_u32 in 9 column chains (the interleaved reduction's m * p + the 9 m_i = t_i * pinv products)
template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned* out, int iters, unsigned long long* stamps) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned acc = 0;
  if constexpr (MODE == 0) {
    v4i a = {(int)threadIdx.x, 3, 5, 7}, b = {11, (int)threadIdx.x, 13, 17};
    v16i c[3];
    for (int j = 0; j < 3; ++j)
      for (int i = 0; i < 16; ++i) c[j][i] = i + j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int prod = 0; prod < 2; ++prod)        // * (-1/p) mod 2^261, then * p
#pragma unroll
        for (int half = 0; half < 2; ++half)      // elements of lanes 0-31, then of lanes 32-63
#pragma unroll
          for (int blk = 0; blk < 3; ++blk) {     // 66 output columns = 3 blocks of 32 rows
            c[blk] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[blk], 0, 0, 0);
            c[blk] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, c[blk], 0, 0, 0);   // second K-step (digits 32..)
          }
      a[0] ^= c[0][0];                            // a dependency per "multiplication", as a real chain has
    }
    for (int j = 0; j < 3; ++j)
      for (int i = 0; i < 16; ++i) acc += (unsigned)c[j][i];
  } else {
    unsigned x[9], p[9];
    unsigned long long col[9];
    for (int i = 0; i < 9; ++i) {
      x[i] = threadIdx.x * 2654435761u + i;
      p[i] = 0x1234567u * (i + 1);
      col[i] = i;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 10; ++j)              // 9 products of the column + the m_i product
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(col[i]) : "v"(x[(i + j) % 9]), "v"(p[j % 9]) : "vcc");
      x[0] ^= (unsigned)col[8];
    }
    for (int i = 0; i < 9; ++i) acc += (unsigned)col[i];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int MODE>
static double run(int waves_per_simd, int iters) {
  int dev = 0;
  hipDeviceProp_t pr;
  CHECK(hipGetDeviceProperties(&pr, dev));
  const int cus = pr.multiProcessorCount;
  const int blocks = cus * waves_per_simd;        // 256 threads = 4 waves = one wave per SIMD of a CU
  unsigned* out;
  unsigned long long* st;
  CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  CHECK(hipMalloc(&st, (size_t)blocks * 8));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, st);
  CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, st);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(blocks);
  CHECK(hipMemcpy(h.data(), st, (size_t)blocks * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  CHECK(hipFree(out));
  CHECK(hipFree(st));
  // cycles one wave needs per "multiplication" while `waves_per_simd` waves share its SIMD -> SIMD cycles per
  // multiplication of 64 elements = that / waves_per_simd
  return (double)h[blocks / 2] / iters / waves_per_simd;
}

int main() {
  const int iters = 2000;
  printf("constant products of one Montgomery multiplication in the 9 x 29-bit field, 64 elements (one wave):\n");
  for (int w : {1, 2, 4}) {
    const double mf = run<0>(w, iters), va = run<1>(w, iters);
    printf("  %d wave(s)/SIMD: 24 x v_mfma_i32_32x32x32_i8 = %7.1f SIMD cycles   90 x v_mad_u64_u32 = %7.1f SIMD cycles   "
           "MFMA / VALU = %.2f\n", w, mf, va, mf / va);
  }
  printf("(MFMA side: the matrix instructions ALONE -- no digit conversion, no half-wave swaps, no carry recovery;\n"
         " a matrix-core reduction pays for itself only if this ratio is below ~0.75, VERDICT r04 #8 asked for 1.3 x)\n");
  return 0;
}
