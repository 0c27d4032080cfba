// Micro-benchmark: issue rate of the integer / fp64 instructions that a 254-bit
// Montgomery multiplier can be built from on gfx950.  Reports wave-instructions
// per second per SIMD relative to v_add_u32, both with 8 waves/SIMD (throughput)
// and 1 wave/SIMD (single-wave issue).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int CHAINS = 8;
constexpr int UNROLL = 4;   // x CHAINS instructions per loop iteration

enum Op { ADD_U32, MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, MAD_U32_U24, MUL_HI_U32_U24,
          FMA_F64, ADDC_PAIR, FMA_F32, MAD_U32_U16, LSHL_ADD_U64, MUL_U32_U24, MAD_U64_DEP, LSHR_B64, ALIGNBIT, AND_B32, MAD_U64_SGPR, ADD3_U32, LSHL_OR, MAD_NOP, MAD_NOP_DEP, MAD_BLOCK8, MAD_BLOCK8_DEP, AND_BLOCK8 };

// Every wave stamps the shader clock (s_memtime: one tick per shader cycle) and the constant 100 MHz counter
// (s_memrealtime) around its loop; wave 0 of each workgroup stores them.  No clock is assumed anywhere:
//   in-kernel clock                      = median over workgroups of d_memtime / d_memrealtime * 100 MHz
//   wall time of the launch              = (latest end stamp - earliest start stamp) of s_memrealtime
//   cycles per wave-instruction per SIMD = wall * clock * 1024 SIMDs / wave-instructions issued      -- REAL cycles
// (occupancy-independent: the 8-waves case need not have all its workgroups resident at once)
// (round 1 converted event time to cycles at an assumed 2.4 GHz, which the chip does not hold under this load).
template <int OP>
__global__ void __launch_bounds__(256) k(unsigned* out, int iters, unsigned seed, unsigned long long* stamps) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned a[CHAINS], b[CHAINS];
  unsigned long long w[CHAINS];
  double d[CHAINS];
  for (int i = 0; i < CHAINS; ++i) {
    a[i] = seed * (i + 3) + threadIdx.x;
    b[i] = seed ^ (0x9e3779b9u * (i + 1));
    w[i] = ((unsigned long long)a[i] << 32) | b[i];
    d[i] = 1.0 + 1e-9 * (double)(a[i] & 0xffff);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) {
        if constexpr (OP == ADD_U32)
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        else if constexpr (OP == MAD_U64_U32)
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
        else if constexpr (OP == MUL_LO_U32)
          asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        else if constexpr (OP == MUL_HI_U32)
          asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        else if constexpr (OP == MAD_U32_U24)
          asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        else if constexpr (OP == MUL_HI_U32_U24)
          asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        else if constexpr (OP == MUL_U32_U24)
          asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        else if constexpr (OP == FMA_F64)
          asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) % CHAINS]));
        else if constexpr (OP == FMA_F32) {
          float f = __uint_as_float(a[i]);
          asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f) : "v"(b[i]));
          a[i] = __float_as_uint(f);
        } else if constexpr (OP == ADDC_PAIR)
          asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc"
                       : "+v"(a[i]), "+v"(b[i]) : "v"(seed) : "vcc");
        else if constexpr (OP == MAD_U32_U16)
          asm volatile("v_mad_u32_u16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        else if constexpr (OP == LSHL_ADD_U64)
          asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i]) : "v"(w[(i + 1) % CHAINS]));
        else if constexpr (OP == LSHR_B64)
          asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(w[i]));
        else if constexpr (OP == ALIGNBIT)
          asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(a[i]) : "v"(b[i]));
        else if constexpr (OP == AND_B32)
          asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        else if constexpr (OP == ADD3_U32)
          asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
        else if constexpr (OP == LSHL_OR)
          asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b[i]));
        else if constexpr (OP == MAD_U64_SGPR)
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "s"(seed) : "vcc");
        else if constexpr (OP == MAD_BLOCK8) {  // 8 multiply-adds in ONE asm statement: hipcc pads once per statement
          if (i == 0)
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %16, %0\n\tv_mad_u64_u32 %1, vcc, %9, %17, %1\n\t"
                         "v_mad_u64_u32 %2, vcc, %10, %18, %2\n\tv_mad_u64_u32 %3, vcc, %11, %19, %3\n\t"
                         "v_mad_u64_u32 %4, vcc, %12, %20, %4\n\tv_mad_u64_u32 %5, vcc, %13, %21, %5\n\t"
                         "v_mad_u64_u32 %6, vcc, %14, %22, %6\n\tv_mad_u64_u32 %7, vcc, %15, %23, %7"
                         : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7])
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
                           "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "vcc");
        } else if constexpr (OP == MAD_BLOCK8_DEP) {   // the same as ONE dependent chain (a column of a product)
          if (i == 0)
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %9, %0\n\tv_mad_u64_u32 %0, vcc, %2, %10, %0\n\t"
                         "v_mad_u64_u32 %0, vcc, %3, %11, %0\n\tv_mad_u64_u32 %0, vcc, %4, %12, %0\n\t"
                         "v_mad_u64_u32 %0, vcc, %5, %13, %0\n\tv_mad_u64_u32 %0, vcc, %6, %14, %0\n\t"
                         "v_mad_u64_u32 %0, vcc, %7, %15, %0\n\tv_mad_u64_u32 %0, vcc, %8, %16, %0"
                         : "+v"(w[0])
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
                           "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "vcc");
        } else if constexpr (OP == AND_BLOCK8) {
          if (i == 0)
            asm volatile("v_and_b32 %0, %0, %8\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\t"
                         "v_and_b32 %4, %4, %8\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                         : "v"(b[0]));
        } else if constexpr (OP == MAD_NOP)       // what hipcc emits behind every pinned multiply-add: mad + s_nop
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\ts_nop 0" : "+v"(w[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
        else if constexpr (OP == MAD_NOP_DEP)
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\ts_nop 0" : "+v"(w[0]) : "v"(a[i]), "v"(b[i]) : "vcc");
        else if constexpr (OP == MAD_U64_DEP)   // single dependent chain through chain 0
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[0]) : "v"(a[i]), "v"(b[i]) : "vcc");
      }
    }
  }
  unsigned r = 0;
  for (int i = 0; i < CHAINS; ++i) r ^= a[i] ^ b[i] ^ (unsigned)w[i] ^ (unsigned)(w[i] >> 32) ^ (unsigned)__double_as_longlong(d[i]);
  if (r == 0x12345678u) out[0] = r;   // keep live
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (stamps && threadIdx.x == 0) {   // stamp buffer of its own: nothing else reads it
    stamps[4 * blockIdx.x] = t1 - t0;
    stamps[4 * blockIdx.x + 1] = r1 - r0;
    stamps[4 * blockIdx.x + 2] = r0;     // absolute 100 MHz stamps: the launch's wall time as seen from inside
    stamps[4 * blockIdx.x + 3] = r1;
  }
}

static unsigned long long* g_stamps = nullptr;
static std::vector<unsigned long long> h_stamps;
static double median(std::vector<double> v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

template <int OP>
double run(const char* name, int blocks, int iters, unsigned* out, int instr_per = 1) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<OP><<<blocks, 256>>>(out, 16, 7u, nullptr);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  double cyc = 0, clk = 0;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0));
    k<OP><<<blocks, 256>>>(out, iters, 7u + rep, g_stamps);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) {
      best = ms;
      CHECK(hipMemcpy(h_stamps.data(), g_stamps, (size_t)blocks * 32, hipMemcpyDeviceToHost));
      std::vector<double> ck;
      unsigned long long first = ~0ull, last = 0;
      for (int b = 0; b < blocks; ++b) {
        ck.push_back((double)h_stamps[4 * b] / (double)h_stamps[4 * b + 1] * 0.1);   // GHz
        first = std::min(first, h_stamps[4 * b + 2]);
        last = std::max(last, h_stamps[4 * b + 3]);
      }
      clk = median(ck);
      cyc = (double)(last - first) * 1e-8 * clk * 1e9;    // wall cycles of the launch at the in-kernel clock
    }
  }
  const double per_wave = (double)iters * UNROLL * CHAINS * instr_per;
  double waves = (double)blocks * 4;
  double winstr = waves * per_wave;
  double rate = winstr / (best * 1e-3);      // wave-instr / s chip-wide
  printf("%-16s blocks=%5d  %8.3f ms  %10.3f Gwinstr/s  => %6.2f cycles/winstr/SIMD (s_memtime; in-kernel clock %.3f GHz)\n",
         name, blocks, best, rate * 1e-9, cyc * 1024.0 / winstr, clk);
  return rate;
}

int main() {
  unsigned* out; CHECK(hipMalloc(&out, 4));
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  CHECK(hipMalloc(&g_stamps, 4096 * 32));
  h_stamps.resize(4096 * 4);
  for (int pass = 0; pass < 2; ++pass) {
    // pass 0: 8 waves/SIMD (2048 blocks of 4 waves on 256 CUs = 8 blocks/CU);  pass 1: 1 wave/SIMD
    int blocks = pass == 0 ? 256 * 8 : 256;
    int iters = pass == 0 ? 4000 : 16000;
    printf("---- %s ----\n", pass == 0 ? "8 waves/SIMD" : "1 wave/SIMD");
    run<ADD_U32>("v_add_u32", blocks, iters, out);
    run<FMA_F32>("v_fma_f32", blocks, iters, out);
    run<MAD_U64_U32>("v_mad_u64_u32", blocks, iters, out);
    run<MAD_U64_DEP>("v_mad_u64 dep", blocks, iters, out);
    run<MUL_LO_U32>("v_mul_lo_u32", blocks, iters, out);
    run<MUL_HI_U32>("v_mul_hi_u32", blocks, iters, out);
    run<MAD_U32_U24>("v_mad_u32_u24", blocks, iters, out);
    run<MUL_U32_U24>("v_mul_u32_u24", blocks, iters, out);
    run<MUL_HI_U32_U24>("v_mul_hi_u32_u24", blocks, iters, out);
    run<MAD_U32_U16>("v_mad_u32_u16", blocks, iters, out);
    run<FMA_F64>("v_fma_f64", blocks, iters, out);
    run<ADDC_PAIR>("add_co+addc", blocks, iters, out, 2);
    run<LSHL_ADD_U64>("v_lshl_add_u64", blocks, iters, out);
    run<LSHR_B64>("v_lshrrev_b64", blocks, iters, out);
    run<ALIGNBIT>("v_alignbit_b32", blocks, iters, out);
    run<AND_B32>("v_and_b32", blocks, iters, out);
    run<ADD3_U32>("v_add3_u32", blocks, iters, out);
    run<LSHL_OR>("v_lshl_or_b32", blocks, iters, out);
    run<MAD_U64_SGPR>("v_mad_u64 sgpr", blocks, iters, out);
    run<MAD_NOP>("mad + s_nop", blocks, iters, out);          // counted as ONE instruction: the price of the pair
    run<MAD_NOP_DEP>("mad + s_nop dep", blocks, iters, out);
    run<MAD_BLOCK8>("mad x8 / asm", blocks, iters, out);       // hipcc pads every asm STATEMENT with one s_nop: the rows
    run<MAD_BLOCK8_DEP>("mad x8 chain/asm", blocks, iters, out);   // above carry one s_nop per instruction, these 1/8
    run<AND_BLOCK8>("and x8 / asm", blocks, iters, out);
  }
  {   // the accumulate kernels' occupancy: 4 waves per SIMD
    printf("---- 4 waves/SIMD ----\n");
    run<MAD_U64_U32>("v_mad_u64_u32", 1024, 8000, out);
    run<MAD_U64_DEP>("v_mad_u64 dep", 1024, 8000, out);
    run<MAD_NOP>("mad + s_nop", 1024, 8000, out);
    run<MAD_NOP_DEP>("mad + s_nop dep", 1024, 8000, out);
    run<MAD_BLOCK8>("mad x8 / asm", 1024, 8000, out);
    run<MAD_BLOCK8_DEP>("mad x8 chain/asm", 1024, 8000, out);
    run<AND_B32>("v_and_b32", 1024, 8000, out);
    run<AND_BLOCK8>("and x8 / asm", 1024, 8000, out);
    printf("---- 2 waves/SIMD ----\n");
    run<MAD_U64_U32>("v_mad_u64_u32", 512, 8000, out);
    run<MAD_U64_DEP>("v_mad_u64 dep", 512, 8000, out);
    run<MAD_NOP>("mad + s_nop", 512, 8000, out);
    run<MAD_NOP_DEP>("mad + s_nop dep", 512, 8000, out);
    run<MAD_BLOCK8>("mad x8 / asm", 512, 8000, out);
    run<MAD_BLOCK8_DEP>("mad x8 chain/asm", 512, 8000, out);
  }
  return 0;
}
