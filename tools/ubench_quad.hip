// Micro-benchmark for the latency chains of the MSM tails: what does ONE dependent group addition / doubling cost a
// wave that runs alone, one lane per point (Curve::add, ec.cuh) against a cooperating quad (add_quad / dbl_quad,
// msm.cuh), G1 and G2, as a hot loop (the code stays in the instruction cache)?  The reduce2 / fold kernels execute most
// of their code ONCE, so the difference between these figures and the kernels' time per operation is instruction fetch.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_quad.hip -o tools/ubench_quad && ./tools/ubench_quad
// The operands are random field elements taken as XYZZ coordinates: the formulas do not use the curve equation, and the
// timing does not depend on the values.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../nim_groth16_amd/csrc/g16_internal.hpp"
#include "../nim_groth16_amd/csrc/msm.cuh"
using namespace g16;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
const G16Env& g16_env() { static G16Env e; return e; }

// MODE 0: Curve::add, 1: add_quad, 2: Curve::dbl, 3: dbl_quad
template <class C, int MODE>
__global__ void __launch_bounds__(64) k_chain(const typename C::Acc* in, typename C::Acc* out, int iters) {
  typename C::Acc a = in[0];
  const typename C::Acc b = in[1];
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) C::add(a, b);
    if (MODE == 1) a = add_quad<C>(a, b);
    if (MODE == 2) a = C::dbl(a);
    if (MODE == 3) a = dbl_quad<C>(a);
  }
  if (threadIdx.x == 0) out[0] = a;
}
// the two paths agree (as points: compare x zz' == x' zz, y zzz' == y' zzz)
template <class C>
__global__ void k_check(const typename C::Acc* in, unsigned* bad) {
  using F = typename C::Field;
  typename C::Acc a = in[0], q = in[0];
  const typename C::Acc b = in[1];
  for (int i = 0; i < 5; ++i) {
    C::add(a, b);
    q = add_quad<C>(q, b);
    a = C::dbl(a);
    q = dbl_quad<C>(q);
  }
  const bool ok = F::is_zero(F::sub(F::mul(a.x, q.zz), F::mul(q.x, a.zz))) &&
                  F::is_zero(F::sub(F::mul(a.y, q.zzz), F::mul(q.y, a.zzz)));
  if (!ok) atomicAdd(bad, 1u);
}

template <class C>
static void run(const char* name) {
  using Acc = typename C::Acc;
  std::vector<unsigned char> h(2 * sizeof(Acc));
  unsigned long long s = 88172645463325252ull;
  for (size_t i = 0; i < h.size(); ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    h[i] = (unsigned char)(s >> 11);
  }
  for (size_t i = 31; i < h.size(); i += 32) h[i] &= 0x1f;   // every 256-bit word below the modulus
  Acc *d_in, *d_out;
  unsigned* d_bad;
  CHECK(hipMalloc(&d_in, h.size()));
  CHECK(hipMalloc(&d_out, sizeof(Acc)));
  CHECK(hipMalloc(&d_bad, 4));
  CHECK(hipMemset(d_bad, 0, 4));
  CHECK(hipMemcpy(d_in, h.data(), h.size(), hipMemcpyHostToDevice));
  k_check<C><<<1, 64>>>(d_in, d_bad);
  unsigned bad = 0;
  CHECK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
  printf("%s: quad and one-lane results agree: %s\n", name, bad ? "NO" : "yes");
  const int iters = 400;
  auto time_us = [&](auto kernel) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    kernel<<<1, 64>>>(d_in, d_out, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    kernel<<<1, 64>>>(d_in, d_out, iters);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / iters;
  };
  const double t_add = time_us(k_chain<C, 0>), t_addq = time_us(k_chain<C, 1>), t_dbl = time_us(k_chain<C, 2>),
               t_dblq = time_us(k_chain<C, 3>);
  printf("%s: addition %.2f us one lane, %.2f us quad (%.2fx) | doubling %.2f us one lane, %.2f us quad (%.2fx)\n", name,
         t_add, t_addq, t_add / t_addq, t_dbl, t_dblq, t_dbl / t_dblq);
}

int main() {
  run<G1>("G1");
  run<G2>("G2");
  return 0;
}
