// Verifier side of the C ABI: batched Groth16 verification and the raw pairing (verifier.nim:31-52,
// curves.nim:218-221).  One lane per pairing / per proof; see pairing.cuh.  Not a hot path of the prover --
// it exists so that a user of the reference's verifyProof finds it here, running on the same device.
#include "g16_internal.hpp"
#include "pairing.cuh"

using namespace g16;

struct g16_vkey {
  int device = 0;
  uint32_t npubs = 0;
  g2_aff gamma2, delta2;
  void* d_ic = nullptr;   // (npubs + 1) G1 points
  void* d_ab = nullptr;   // Miller value of (alpha1, beta2): 384 B
};

namespace {

constexpr int PBLOCK = 64;

// k * P by double-and-add over the 254 bits of a canonical scalar
template <class C>
__device__ typename C::Acc scalar_mul(const typename C::Aff& p, u256 k) {
  typename C::Acc acc = C::acc_inf();
#pragma unroll 1
  for (int j = 7; j >= 0; --j) {
    uint32_t limb = k.v[7];
#pragma unroll
    for (int q = 7; q > 0; --q) k.v[q] = k.v[q - 1];   // rotate: static register indices
    k.v[0] = limb;
#pragma unroll 1
    for (int b = 31; b >= 0; --b) {
      acc = C::dbl(acc);
      if ((limb >> b) & 1) C::madd(acc, p);
    }
  }
  return acc;
}

__device__ bool on_curve_g1(const g1_aff& p) {
  if (G1::is_inf(p)) return true;
  u256 three = Fp::add(Fp::dbl(Fp::one()), Fp::one());
  return Fp::eq(Fp::sqr(p.y), Fp::add(Fp::mul(Fp::sqr(p.x), p.x), three));
}
__device__ bool on_curve_g2(const g2_aff& p, const fp2_t& b) {
  if (G2::is_inf(p)) return true;
  return Fp2::eq(Fp2::sqr(p.y), Fp2::add(Fp2::mul(Fp2::sqr(p.x), p.x), b));
}

// partial[j * (npubs+1) + i] = publicIO[j][i] * IC[i]
__global__ void __launch_bounds__(PBLOCK) verify_pub_terms(const u256* __restrict__ pub, uint32_t mont,
                                                           const g1_aff* __restrict__ ic, uint32_t nio,
                                                           uint32_t total, g1_acc* __restrict__ partial,
                                                           int32_t* __restrict__ status) {
  uint32_t t = blockIdx.x * PBLOCK + threadIdx.x;
  if (t >= total) return;
  u256 s = pub[t];
  if (!Fr::is_canonical(s)) atomicMin(&status[t / nio], -6);   // one public input, one accepted encoding
  if (mont) s = Fr::from_mont(s);
  partial[t] = scalar_mul<G1>(ic[t % nio], s);
}

// three Miller loops per proof: (-A, B), (C, delta), (vk_x, gamma); status < 0 on malformed input
__global__ void __launch_bounds__(PBLOCK) verify_miller(const g16_proof* __restrict__ proofs, uint32_t count,
                                                        const g1_acc* __restrict__ partial, uint32_t nio,
                                                        g2_aff gamma2, g2_aff delta2, fp2_t twist_b,
                                                        uint32_t check_subgroup, fp12_t* __restrict__ mil,
                                                        int32_t* __restrict__ status) {
  uint32_t t = blockIdx.x * PBLOCK + threadIdx.x;
  if (t >= 3 * count) return;
  const uint32_t j = t / 3, which = t % 3;
  const g16_proof& pr = proofs[j];
  g1_aff P;
  g2_aff Q;
  if (which == 0) {
    P = *reinterpret_cast<const g1_aff*>(pr.pi_a);
    Q = *reinterpret_cast<const g2_aff*>(pr.pi_b);
    // every coordinate must be the canonical residue (< p): the field arithmetic below would silently reduce a
    // larger limb pattern, so without this one proof would have several accepted byte encodings
    if (!Fp::is_canonical(P.x) || !Fp::is_canonical(P.y) || !Fp::is_canonical(Q.x.c0) || !Fp::is_canonical(Q.x.c1) ||
        !Fp::is_canonical(Q.y.c0) || !Fp::is_canonical(Q.y.c1))
      atomicMin(&status[j], -5);
    if (!on_curve_g1(P)) atomicMin(&status[j], -1);
    if (!on_curve_g2(Q, twist_b)) {
      atomicMin(&status[j], -2);
    } else if (check_subgroup) {   // [r]Q == infinity (the reference only asserts the curve equation)
      u256 r{{FrParams::P0, FrParams::P1, FrParams::P2, FrParams::P3, FrParams::P4, FrParams::P5, FrParams::P6,
              FrParams::P7}};
      if (!G2::is_inf(scalar_mul<G2>(Q, r))) atomicMin(&status[j], -4);
    }
    P = G1::neg(P);
  } else if (which == 1) {
    P = *reinterpret_cast<const g1_aff*>(pr.pi_c);
    Q = delta2;
    if (!Fp::is_canonical(P.x) || !Fp::is_canonical(P.y)) atomicMin(&status[j], -5);
    if (!on_curve_g1(P)) atomicMin(&status[j], -3);
  } else {
    g1_acc acc = G1::acc_inf();
    for (uint32_t i = 0; i < nio; ++i) G1::add(acc, partial[(size_t)j * nio + i]);
    P = G1::to_affine(acc);
    Q = gamma2;
  }
  Pairing::miller(mil[t], P, Q);
}

// status[j] = (m0 m1 m2 ab)^((p^12-1)/r) == 1, unless already negative
__global__ void __launch_bounds__(PBLOCK) verify_final(const fp12_t* __restrict__ mil, const fp12_t* __restrict__ ab,
                                                       uint32_t count, int32_t* __restrict__ status) {
  uint32_t j = blockIdx.x * PBLOCK + threadIdx.x;
  if (j >= count || status[j] < 0) return;
  fp12_t f, t;
  Pairing::mul(t, mil[3 * j], mil[3 * j + 1]);
  Pairing::mul(f, t, mil[3 * j + 2]);
  Pairing::mul(t, f, *ab);
  Pairing::final_exp(f, t);
  status[j] = Pairing::is_one(f) ? 1 : 0;
}

__global__ void __launch_bounds__(PBLOCK) pairing_kernel(const g1_aff* __restrict__ p, const g2_aff* __restrict__ q,
                                                         uint32_t n, int do_final, fp12_t* __restrict__ out) {
  uint32_t t = blockIdx.x * PBLOCK + threadIdx.x;
  if (t >= n) return;
  fp12_t f;
  Pairing::miller(f, p[t], q[t]);
  if (do_final) {
    fp12_t g = f;
    Pairing::final_exp(f, g);
  }
  out[t] = f;
}

u256 std_fp(uint32_t a7, uint32_t a6, uint32_t a5, uint32_t a4, uint32_t a3, uint32_t a2, uint32_t a1, uint32_t a0) {
  u256 v;
  v.v[0] = a0; v.v[1] = a1; v.v[2] = a2; v.v[3] = a3; v.v[4] = a4; v.v[5] = a5; v.v[6] = a6; v.v[7] = a7;
  return Fp::to_mont(v);   // standard form -> Montgomery
}
fp2_t twist_b() {   // twistCoeffB = 3/(9+u) (curves.nim:75-77)
  fp2_t b;
  b.c0 = std_fp(0x2b149d40u, 0xceb8aaaeu, 0x81be1899u, 0x1be06ac3u, 0xb5b4c5e5u, 0x59dbefa3u, 0x3267e6dcu, 0x24a138e5u);
  b.c1 = std_fp(0x009713b0u, 0x3af0fed4u, 0xcd2cafadu, 0xeed8fdf4u, 0xa74fa084u, 0xe52d1852u, 0xe4a2bd06u, 0x85c315d2u);
  return b;
}

}  // namespace

extern "C" int32_t g16_pairing(g16_ctx* ctx, const void* g1_points, const void* g2_points, size_t n, void* out_gt) {
  if (!ctx) return G16_EINVAL;
  if ((n && (!g1_points || !g2_points || !out_gt)) || n >= (size_t(1) << 24)) {
    ctx->err = "g16_pairing: bad arguments";
    return G16_EINVAL;
  }
  if (!n) return G16_OK;
  CTX_ENTER(ctx);
  int32_t rc;
  const size_t o_q = n * 64, o_out = o_q + n * 128;
  if ((rc = ensure(ctx, ctx->stage_p, o_out + n * sizeof(fp12_t)))) return rc;
  char* ws = (char*)ctx->stage_p.p;
  HIPCHK(ctx, hipMemcpyAsync(ws, g1_points, n * 64, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ws + o_q, g2_points, n * 128, hipMemcpyHostToDevice, ctx->stream));
  KLAUNCH(ctx, "pairing", pairing_kernel, (uint32_t)((n + PBLOCK - 1) / PBLOCK), PBLOCK, 0, (const g1_aff*)ws,
          (const g2_aff*)(ws + o_q), (uint32_t)n, 1, (fp12_t*)(ws + o_out));
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out_gt, ws + o_out, n * sizeof(fp12_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return G16_OK;
}

extern "C" int32_t g16_vkey_create(g16_ctx* ctx, const g16_vkey_desc* d, g16_vkey** out) {
  if (!ctx) return G16_EINVAL;
  if (!d || !out || !d->alpha1 || !d->beta2 || !d->gamma2 || !d->delta2 || !d->pointsIC || d->npubs >= (1u << 20)) {
    ctx->err = "g16_vkey_create: bad descriptor";
    return G16_EINVAL;
  }
  CTX_ENTER(ctx);
  g16_vkey* k = new (std::nothrow) g16_vkey();
  if (!k) return G16_ENOMEM;
  k->device = ctx->device;
  k->npubs = d->npubs;
  memcpy(&k->gamma2, d->gamma2, 128);
  memcpy(&k->delta2, d->delta2, 128);
  const size_t nio = (size_t)d->npubs + 1;
  auto fail = [&](int32_t rc) {
    g16_vkey_destroy(k);
    return rc;
  };
  if (hipMalloc(&k->d_ic, nio * 64) != hipSuccess || hipMalloc(&k->d_ab, sizeof(fp12_t) + 64 + 128) != hipSuccess) {
    ctx->err = "g16_vkey_create: hipMalloc failed";
    return fail(G16_ENOMEM);
  }
  char* ab = (char*)k->d_ab;
  if (hipMemcpyAsync(k->d_ic, d->pointsIC, nio * 64, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipMemcpyAsync(ab + sizeof(fp12_t), d->alpha1, 64, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipMemcpyAsync(ab + sizeof(fp12_t) + 64, d->beta2, 128, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
    ctx->err = "g16_vkey_create: upload failed";
    return fail(G16_EHIP);
  }
  // vkey.spec.alphaBeta (zkey_types.nim:62-73) is kept as its Miller value; the final exponentiation is shared
  hipLaunchKernelGGL(pairing_kernel, dim3(1), dim3(PBLOCK), 0, ctx->stream, (const g1_aff*)(ab + sizeof(fp12_t)),
                     (const g2_aff*)(ab + sizeof(fp12_t) + 64), 1u, 0, (fp12_t*)ab);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
    ctx->err = "g16_vkey_create: kernel failed";
    return fail(G16_EHIP);
  }
  *out = k;
  return G16_OK;
}

extern "C" void g16_vkey_destroy(g16_vkey* k) {
  if (!k) return;
  (void)hipSetDevice(k->device);   // device-bound like g16_pkey: valid before and after any context
  (void)hipDeviceSynchronize();
  if (k->d_ic) (void)hipFree(k->d_ic);
  if (k->d_ab) (void)hipFree(k->d_ab);
  delete k;
}

extern "C" int32_t g16_verify(g16_ctx* ctx, const g16_vkey* key, const g16_proof* proofs, const void* public_io,
                              uint32_t flags, size_t count, int32_t* status) {
  if (!ctx) return G16_EINVAL;
  if (!key || key->device != ctx->device || (count && (!proofs || !public_io || !status)) || count >= (size_t(1) << 22)) {
    ctx->err = "g16_verify: bad arguments";
    return G16_EINVAL;
  }
  if (!count) return G16_OK;
  CTX_ENTER(ctx);
  const size_t nio = (size_t)key->npubs + 1, total = count * nio;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t r = o;
    o += (bytes + 255) & ~size_t(255);
    return r;
  };
  const size_t o_pr = take(count * sizeof(g16_proof)), o_pub = take(total * 32), o_part = take(total * sizeof(g1_acc)),
               o_mil = take(3 * count * sizeof(fp12_t)), o_st = take(count * 4);
  int32_t rc;
  if ((rc = ensure(ctx, ctx->stage_p, o))) return rc;
  char* ws = (char*)ctx->stage_p.p;
  HIPCHK(ctx, hipMemcpyAsync(ws + o_pr, proofs, count * sizeof(g16_proof), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ws + o_pub, public_io, total * 32, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ws + o_st, 0, count * 4, ctx->stream));
  KLAUNCH(ctx, "verify_pub_terms", verify_pub_terms, (uint32_t)((total + PBLOCK - 1) / PBLOCK), PBLOCK, 0,
          (const u256*)(ws + o_pub), (flags & G16_SCALARS_MONT) ? 1u : 0u, (const g1_aff*)key->d_ic, (uint32_t)nio,
          (uint32_t)total, (g1_acc*)(ws + o_part), (int32_t*)(ws + o_st));
  KLAUNCH(ctx, "verify_miller", verify_miller, (uint32_t)((3 * count + PBLOCK - 1) / PBLOCK), PBLOCK, 0,
          (const g16_proof*)(ws + o_pr), (uint32_t)count, (const g1_acc*)(ws + o_part), (uint32_t)nio, key->gamma2,
          key->delta2, twist_b(), (flags & G16_VERIFY_SUBGROUP) ? 1u : 0u, (fp12_t*)(ws + o_mil),
          (int32_t*)(ws + o_st));
  KLAUNCH(ctx, "verify_final", verify_final, (uint32_t)((count + PBLOCK - 1) / PBLOCK), PBLOCK, 0,
          (const fp12_t*)(ws + o_mil), (const fp12_t*)key->d_ab, (uint32_t)count, (int32_t*)(ws + o_st));
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(status, ws + o_st, count * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return G16_OK;
}
