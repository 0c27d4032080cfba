// Micro-benchmark: batched-AFFINE G2 additions with real memory traffic, against the mixed XYZZ addition of
// msm_accum<G2> (DESIGN §5 "Not pursued").
//
// Every lane adds K independent pairs of affine G2 points (gathered at random from a 1.7-GB table, like the bucket
// entries of a 2^20 MSM) sharing ONE inversion (Montgomery's trick):
//   forward : d_i = x2_i - x1_i ;  prefix_i = d_0 ... d_(i-1)  -> scratch (HBM, [i][lane]: coalesced) ;  acc *= d_i
//   invert  : acc^-1 in Fp2 (one Fp inversion by division steps, ff.cuh)
//   backward: re-gather both points ;  1/d_i = acc^-1 * prefix_i ;  acc^-1 *= d_i ;
//             lambda = (y2 - y1)/d_i ;  x3 = lambda^2 - x1 - x2 ;  y3 = lambda (x1 - x3) - y1  -> out (128 B)
// = 5 Fp2 products + 1 Fp2 square per addition + 1/K inversion, and ~640 B of HBM traffic instead of one 128-B gather.
// A second kernel computes the same sums with one inversion per pair; the outputs must agree bit for bit.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_affine_g2.hip -o tools/ubench_affine_g2
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define G16_F29_SERIAL
#define G16_F29_PAIR
#include "../nim_groth16_amd/csrc/ec29.cuh"
using namespace g16;
using E = Ec29<G2>;
using F = Fp29;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

struct alignas(16) f2pack {
  u256 c0, c1;
};
__device__ __forceinline__ f2e29 f2unpack(const f2pack& p) { return f2e29{F::relimb(p.c0), F::relimb(p.c1)}; }
__device__ __forceinline__ f2pack f2packc(const f2e29& a) {   // any normalized value < 16p -> canonical, packed
  return f2pack{F::relimb(F::canon<16>(a.c0)), F::relimb(F::canon<16>(a.c1))};
}
template <uint32_t M = 4>
__device__ __forceinline__ f2e29 f2sub(const f2e29& a, const f2e29& b) {   // a - b + M p, normalized; b normalized, < M p
  return f2e29{F::norm(F::subk<M, 1>(a.c0, b.c0)), F::norm(F::subk<M, 1>(a.c1, b.c1))};
}
__device__ __forceinline__ f2e29 f2canon(const f2e29& a) { return f2e29{F::canon<16>(a.c0), F::canon<16>(a.c1)}; }
// 1/a in Fp2 through the 8x32 field's division-step inversion: 1/(a0 + a1 u) = (a0 - a1 u)/(a0^2 + a1^2)
__device__ __forceinline__ f2e29 f2inv(const f2e29& a) {
  const fp2_t s = E::f2_to_std(f2canon(a));
  return E::f2_from_std(Fp2::inv(s));
}
__device__ __forceinline__ void affine_finish(const g2_tab29& p1, const g2_tab29& p2, const f2e29& dinv, g2_tab29& out) {
  const f2e29 x1{F::relimb(p1.x0), F::relimb(p1.x1)}, y1{F::relimb(p1.y0), F::relimb(p1.y1)};
  const f2e29 x2{F::relimb(p2.x0), F::relimb(p2.x1)}, y2{F::relimb(p2.y0), F::relimb(p2.y1)};
  // bounds (multiples of p, all values normalized): y2 - y1 < 5; lam < 2; l2 < 2; x3 < 10; x1 - x3 < 17; y3 < 6
  const f2e29 lam = E::f2mul<8>(f2sub(y2, y1), dinv);
  const f2e29 l2 = E::f2sqr<4>(lam);
  const f2e29 x3 = f2sub(f2sub(l2, x1), x2);
  const f2e29 y3 = f2sub(E::f2mul<4>(lam, f2sub<16>(x1, x3)), y1);
  const f2pack px = f2packc(x3), py = f2packc(y3);
  out = g2_tab29{px.c0, px.c1, py.c0, py.c1};
}

template <int K, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_batched(const g2_tab29* __restrict__ tab, const uint2* __restrict__ pairs,
                                                        f2pack* __restrict__ scratch, g2_tab29* __restrict__ out,
                                                        uint32_t nthreads) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nthreads) return;
  const uint2* my = pairs + (size_t)t * K;
  f2e29 acc = E::f2_one();
#pragma unroll 1
  for (int i = 0; i < K; ++i) {
    const uint2 pr = my[i];
    const f2e29 x1{F::relimb(tab[pr.x].x0), F::relimb(tab[pr.x].x1)}, x2{F::relimb(tab[pr.y].x0), F::relimb(tab[pr.y].x1)};
    scratch[(size_t)i * nthreads + t] = f2packc(acc);
    acc = E::f2mul<4>(acc, f2sub(x2, x1));
  }
  f2e29 inv = f2inv(acc);
#pragma unroll 1
  for (int i = K - 1; i >= 0; --i) {
    const uint2 pr = my[i];
    const g2_tab29 p1 = tab[pr.x], p2 = tab[pr.y];
    const f2e29 d = f2sub(f2e29{F::relimb(p2.x0), F::relimb(p2.x1)}, f2e29{F::relimb(p1.x0), F::relimb(p1.x1)});
    const f2e29 dinv = E::f2mul<4>(inv, f2unpack(scratch[(size_t)i * nthreads + t]));
    inv = E::f2mul<4>(inv, d);
    affine_finish(p1, p2, dinv, out[(size_t)t * K + i]);
  }
}
// the same sums, one inversion per pair (the check)
template <int K>
__global__ void __launch_bounds__(256) k_direct(const g2_tab29* __restrict__ tab, const uint2* __restrict__ pairs,
                                                g2_tab29* __restrict__ out, uint32_t npairs) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= npairs) return;
  const uint2 pr = pairs[p];
  const g2_tab29 p1 = tab[pr.x], p2 = tab[pr.y];
  const f2e29 d = f2sub(f2e29{F::relimb(p2.x0), F::relimb(p2.x1)}, f2e29{F::relimb(p1.x0), F::relimb(p1.x1)});
  affine_finish(p1, p2, f2inv(d), out[p]);
}
// the XYZZ side with the same gathers: every lane accumulates K table points into one XYZZ accumulator (Ec29<G2>::madd)
template <int K, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_xyzz(const g2_tab29* __restrict__ tab, const uint2* __restrict__ pairs,
                                                     E::Acc* __restrict__ out, uint32_t nthreads) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nthreads) return;
  const uint2* my = pairs + (size_t)t * K;
  E::Acc acc = E::acc_inf();
#pragma unroll 1
  for (int i = 0; i < K; ++i) {
    E::madd(acc, tab + my[i].x, 0);
    E::madd(acc, tab + my[i].y, my[i].y & 1);
  }
  out[t] = acc;
}

// pseudo-random canonical field elements (< 2^252 < p): the formulas do not care whether a point is on the curve, but all
// x coordinates must differ (x1 == x2 is the exceptional case a real kernel would have to route around)
__global__ void k_fill(uint32_t* words, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t x = i * 2654435761u + 0x9e3779b9u;
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  words[i] = (i & 7) == 7 ? (x & 0x0fffffffu) : x;
}

template <class L>
static double time_ms(L launch) {
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

template <int K, int WAVES>
static void run(const g2_tab29* d_tab, uint32_t ntab, size_t npairs_total) {
  const uint32_t nthreads = (uint32_t)(npairs_total / K), npairs = nthreads * K;
  std::vector<uint2> h(npairs);
  unsigned long long s = 0x9E3779B97F4A7C15ull ^ K;
  for (auto& p : h) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    p.x = (uint32_t)(s % ntab);
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    p.y = (uint32_t)(s % ntab);
    if (p.y == p.x) p.y = (p.x + 1) % ntab;
  }
  uint2* d_pairs;
  f2pack* d_scr;
  g2_tab29 *d_out, *d_ref;
  E::Acc* d_acc;
  CHECK(hipMalloc(&d_pairs, (size_t)npairs * 8));
  CHECK(hipMalloc(&d_scr, (size_t)npairs * sizeof(f2pack)));
  CHECK(hipMalloc(&d_out, (size_t)npairs * sizeof(g2_tab29)));
  CHECK(hipMalloc(&d_ref, (size_t)npairs * sizeof(g2_tab29)));
  CHECK(hipMalloc(&d_acc, (size_t)nthreads * sizeof(E::Acc)));
  CHECK(hipMemcpy(d_pairs, h.data(), (size_t)npairs * 8, hipMemcpyHostToDevice));
  const dim3 grid((nthreads + 255) / 256), gridp((npairs + 255) / 256);
  const double tb = time_ms([&] { hipLaunchKernelGGL((k_batched<K, WAVES>), grid, dim3(256), 0, 0, d_tab, d_pairs, d_scr, d_out, nthreads); });
  const double tx = time_ms([&] { hipLaunchKernelGGL((k_xyzz<K, WAVES>), grid, dim3(256), 0, 0, d_tab, d_pairs, d_acc, nthreads); });
  hipLaunchKernelGGL((k_direct<K>), gridp, dim3(256), 0, 0, d_tab, d_pairs, d_ref, npairs);
  std::vector<unsigned char> a((size_t)npairs * 128), b((size_t)npairs * 128);
  CHECK(hipMemcpy(a.data(), d_out, a.size(), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(b.data(), d_ref, b.size(), hipMemcpyDeviceToHost));
  const bool same = a == b;
  // traffic of the batched kernel per addition: 2 x 64 B (x, forward) + 64 B prefix out + 64 B prefix in + 2 x 128 B + 128 B out
  const double bytes = (double)npairs * (128 + 64 + 64 + 256 + 128 + 8 + 8);
  printf("K = %2d, %d waves/SIMD, %u lanes, %u additions: batched affine %.3f ms = %.1f ps/addition (%.2f TB/s of HBM "
         "traffic by construction), XYZZ madd on the same gathers %.3f ms = %.1f ps/addition (2 per pair) -> affine/XYZZ = "
         "%.2f ; batched == one-inversion-per-pair: %s\n",
         K, WAVES, nthreads, npairs, tb, tb * 1e9 / npairs, bytes / (tb * 1e-3) / 1e12, tx, tx * 1e9 / (2.0 * npairs),
         (tb / npairs) / (tx / (2.0 * npairs)), same ? "yes" : "NO");
  CHECK(hipFree(d_pairs)); CHECK(hipFree(d_scr)); CHECK(hipFree(d_out)); CHECK(hipFree(d_ref)); CHECK(hipFree(d_acc));
  if (!same) exit(1);
}

int main() {
  const uint32_t ntab = 13u << 20;   // the 13 window tables of a 2^20 G2 set: 1.74 GB
  g2_tab29* d_tab;
  CHECK(hipMalloc(&d_tab, (size_t)ntab * sizeof(g2_tab29)));
  hipLaunchKernelGGL(k_fill, dim3((ntab * 32 + 255) / 256), dim3(256), 0, 0, (uint32_t*)d_tab, ntab * 32);
  CHECK(hipDeviceSynchronize());
  const size_t total = (size_t)6 << 20;   // additions per launch (a 2^20 G2 MSM has 13.6 M)
  run<8, 2>(d_tab, ntab, total);
  run<16, 2>(d_tab, ntab, total);
  run<32, 2>(d_tab, ntab, total);
  run<32, 1>(d_tab, ntab, total);
  run<64, 1>(d_tab, ntab, total);
  return 0;
}
