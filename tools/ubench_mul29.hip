// Micro-benchmark: 254-bit Montgomery multiplication, 8x32-bit limbs (ff.cuh, carry chains) vs
// 9x29-bit limbs with carry-free 64-bit column sums (ff29.cuh).  Dependent chain per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../nim_groth16_amd/csrc/ff.cuh"
#include "../nim_groth16_amd/csrc/ff29.cuh"
using namespace g16;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k32(u256* io, int iters) {
  u256 x = io[threadIdx.x & 63], y = io[64 + (threadIdx.x & 63)];
  x.v[0] ^= blockIdx.x; 
  for (int i = 0; i < iters; ++i) { x = Fp::mul(x, y); y = Fp::mul(y, x); }
  if (x.v[0] == 0x12345u) io[0] = y;
}
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k29(fe29* io, int iters) {
  fe29 x = io[threadIdx.x & 63], y = io[64 + (threadIdx.x & 63)];
  x.v[0] ^= blockIdx.x & 0xff;
  for (int i = 0; i < iters; ++i) { x = Fp29::mul(x, y); y = Fp29::mul(y, x); }
  if (x.v[0] == 0x12345u) io[0] = y;
}
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k29d(fe29* io, int iters) {   // dot product (ab+cd)/R
  fe29 x = io[threadIdx.x & 63], y = io[64 + (threadIdx.x & 63)], z = io[(threadIdx.x + 7) & 63], w = io[(threadIdx.x + 9) & 63];
  x.v[0] ^= blockIdx.x & 0xff;
  for (int i = 0; i < iters; ++i) { x = Fp29::dot2(x, y, z, w); z = Fp29::dot2(z, w, x, y); }   // all operands loop-variant
  if (x.v[0] == 0x12345u) io[0] = z;
}
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k29s(fe29* io, int iters) {   // squaring
  fe29 x = io[threadIdx.x & 63], y = io[64 + (threadIdx.x & 63)];
  x.v[0] ^= blockIdx.x & 0xff;
  for (int i = 0; i < iters; ++i) { x = Fp29::sqr(y); y = Fp29::sqr(x); }
  if (x.v[0] == 0x12345u) io[0] = y;
}
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k32d(u256* io, int iters) {
  u256 x = io[threadIdx.x & 63], y = io[64 + (threadIdx.x & 63)], z = io[(threadIdx.x + 7) & 63], w = io[(threadIdx.x + 9) & 63];
  x.v[0] ^= blockIdx.x;
  for (int i = 0; i < iters; ++i) { x = Fp::mul2(x, y, z, w); z = Fp::mul2(z, w, x, y); }
  if (x.v[0] == 0x12345u) io[0] = y;
}

template <class F>
double timeit(F launch) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch(); CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  void* buf; CHECK(hipMalloc(&buf, 128 * 64)); CHECK(hipMemset(buf, 0x5a, 128 * 64));
  // keep the 29-bit limbs in range: 0x1a5a5a5a & mask
  {
    uint32_t h[128 * 9]; for (int i = 0; i < 128 * 9; ++i) h[i] = (0x0a5a5a5au + 977u * i) & 0x0fffffffu;
    for (int i = 0; i < 128; ++i) h[i * 9 + 8] &= 0x1fffff;
    CHECK(hipMemcpy(buf, h, sizeof(h), hipMemcpyHostToDevice));
  }
  const int iters = 2000, blocks = 4096;
  const double muls = 2.0 * iters * blocks * 256;
  auto rep = [&](const char* name, double ms, double per) {
    printf("%-28s %8.3f ms  %7.2f Gmul/s  (%.0f SIMD-cycles/wave-mul @2.4GHz)\n", name, ms, per * muls / ms / 1e6,
           ms * 1e-3 * 2.4e9 * 1024 / (per * muls / 64));
  };
  rep("8x32 mul   (4 waves/SIMD)", timeit([&] { k32<4><<<blocks, 256>>>((u256*)buf, iters); }), 1);
  rep("9x29 mul   (4 waves/SIMD)", timeit([&] { k29<4><<<blocks, 256>>>((fe29*)buf, iters); }), 1);
  rep("8x32 mul   (2 waves/SIMD)", timeit([&] { k32<2><<<blocks, 256>>>((u256*)buf, iters); }), 1);
  rep("9x29 mul   (2 waves/SIMD)", timeit([&] { k29<2><<<blocks, 256>>>((fe29*)buf, iters); }), 1);
  rep("8x32 mul2  (4 waves/SIMD)", timeit([&] { k32d<4><<<blocks, 256>>>((u256*)buf, iters); }), 1);
  rep("9x29 dot2  (4 waves/SIMD)", timeit([&] { k29d<4><<<blocks, 256>>>((fe29*)buf, iters); }), 1);
  rep("9x29 sqr   (4 waves/SIMD)", timeit([&] { k29s<4><<<blocks, 256>>>((fe29*)buf, iters); }), 1);
  return 0;
}
