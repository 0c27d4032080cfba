// Micro-benchmark for the batched-affine question (DESIGN §5 "Not pursued"): what does ONE field
// inversion cost a wave, in units of the 9x29 multiplication the accumulate kernels are built from?
// Every lane inverts its own (different) value -- data-dependent, so the 64 lanes of a wave diverge and the wave pays
// for the slowest path of every step -- with (a) the binary extended Euclid of rounds 1-2 (Fp::inv_eea) and (b) the
// batched division-step inversion of round 3 (Fp::inv, ff.cuh), and, for comparison, runs a dependent chain of
// Fp29::mul.  Batched-affine bucket accumulation replaces the mixed
// XYZZ addition (6M + 2S + one 2-term dot) by the affine one (5M + 1S incl. Montgomery's trick) PLUS 1/K of an
// inversion per addition, K = additions sharing one inversion; this prints the K at which that breaks even.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_inv.hip -o tools/ubench_inv && ./tools/ubench_inv
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../nim_groth16_amd/csrc/ff29.cuh"
using namespace g16;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int WAVES, bool EEA>
__global__ void __launch_bounds__(256, WAVES) k_inv(u256* io, int iters) {
  u256 x = io[(blockIdx.x * 256 + threadIdx.x) & 4095];
  const u256 c = io[(threadIdx.x * 7 + 1) & 4095];
  for (int i = 0; i < iters; ++i) x = Fp::add(EEA ? Fp::inv_eea(x) : Fp::inv(x), c);   // dependent chain, lane-varying values
  if (x.v[0] == 0x12345u && x.v[1] == 0x77u) io[0] = x;
}
// correctness on the device: x * inv(x) == 1 and both algorithms agree
__global__ void k_check(const u256* io, unsigned* bad) {
  const u256 x = io[blockIdx.x * 64 + threadIdx.x];
  const u256 a = Fp::inv(x), b = Fp::inv_eea(x);
  if (!Fp::eq(a, b) || !Fp::eq(Fp::mul(x, a), Fp::one())) atomicAdd(bad, 1u);
}
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_mul29(fe29* io, int iters) {
  fe29 x = io[threadIdx.x & 63], y = io[64 + (threadIdx.x & 63)];
  x.v[0] ^= blockIdx.x & 0xff;
  for (int i = 0; i < iters; ++i) { x = Fp29::mul(x, y); y = Fp29::mul(y, x); }
  if (x.v[0] == 0x12345u) io[0] = y;
}

template <class F>
static double time_ms(F launch) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  launch();
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

int main() {
  std::vector<u256> h(4096);
  unsigned long long s = 88172645463325252ull;
  for (auto& v : h) {
    for (int j = 0; j < 8; ++j) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v.v[j] = (uint32_t)s; }
    v.v[7] &= 0x0fffffffu;   // < p
  }
  u256* d;
  CHECK(hipMalloc(&d, h.size() * 32));
  CHECK(hipMemcpy(d, h.data(), h.size() * 32, hipMemcpyHostToDevice));
  fe29* d29;
  CHECK(hipMalloc(&d29, 128 * sizeof(fe29)));
  CHECK(hipMemset(d29, 0x11, 128 * sizeof(fe29)));
  unsigned* d_bad;
  CHECK(hipMalloc(&d_bad, 4));
  CHECK(hipMemset(d_bad, 0, 4));
  hipLaunchKernelGGL(k_check, dim3(64), dim3(64), 0, 0, d, d_bad);
  unsigned bad = 1;
  CHECK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
  printf("4096 values: inv(x) == inv_eea(x) and x * inv(x) == 1 on the device: %s\n", bad ? "FAILED" : "ok");
  if (bad) return 1;
  const int blocks = 4096;
  for (int waves : {2, 4}) {
    const int it_inv = 8, it_mul = 512;
    double t_eea = waves == 2 ? time_ms([&] { hipLaunchKernelGGL((k_inv<2, true>), dim3(blocks), dim3(256), 0, 0, d, it_inv); })
                              : time_ms([&] { hipLaunchKernelGGL((k_inv<4, true>), dim3(blocks), dim3(256), 0, 0, d, it_inv); });
    double t_new = waves == 2 ? time_ms([&] { hipLaunchKernelGGL((k_inv<2, false>), dim3(blocks), dim3(256), 0, 0, d, it_inv); })
                              : time_ms([&] { hipLaunchKernelGGL((k_inv<4, false>), dim3(blocks), dim3(256), 0, 0, d, it_inv); });
    double t_mul = waves == 2 ? time_ms([&] { hipLaunchKernelGGL(k_mul29<2>, dim3(blocks), dim3(256), 0, 0, d29, it_mul); })
                              : time_ms([&] { hipLaunchKernelGGL(k_mul29<4>, dim3(blocks), dim3(256), 0, 0, d29, it_mul); });
    const double per_mul = t_mul / (2.0 * it_mul);
    for (int alg = 0; alg < 2; ++alg) {
      const double per_inv = (alg ? t_new : t_eea) / it_inv, ratio = per_inv / per_mul;
      // mixed XYZZ addition ~ 9.06 M (1468 multiply-adds / 162); affine with Montgomery's trick 5M + 1S ~ 5.78 M
      printf("%d waves/SIMD, %s: %.3f ms per inversion step, %.4f ms per multiplication step => one inversion = %.1f "
             "multiplications of wave time; batched affine (5.78 M + inv/K) beats XYZZ (9.06 M) only for K > %.1f additions "
             "per inversion; at K = 16: %.2f M, K = 32: %.2f M, K = 64: %.2f M per addition\n",
             waves, alg ? "division steps (Fp::inv)    " : "binary Euclid (Fp::inv_eea)", per_inv, per_mul, ratio,
             ratio / (9.06 - 5.78), 5.78 + ratio / 16, 5.78 + ratio / 32, 5.78 + ratio / 64);
    }
  }
  return 0;
}
