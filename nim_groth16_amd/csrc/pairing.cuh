// BN254 ate pairing for the verifier (reference: `pairing`, bn128/curves.nim:218-221, used by verifyProof,
// verifier.nim:31-52; the reference gets it from constantine).  NOT a hot path: one lane per pairing, written
// for small code (out-of-line tower functions, rolled loops), batched over proofs by the kernels in pairing.hip.
//
//   Fp12 = Fp2[w]/(w^6 - xi), xi = 9 + u  (the tower of export_sage.nim:84-97 flattened: v = w^2),
//   element = g[0..5], value sum g[k] w^k;  the untwist is (x', y') -> (x' w^2, y' w^3).
//   Miller loop: f_{t-1,Q}(P), t - 1 = 6 x^2 (127 bits), affine steps with the line
//       l = yP - (lambda xP) w + (lambda xT - yT) w^3          (same normalisation as the oracle's _line, so the
//   Miller value is bit-comparable), final exponentiation = exactly (p^12 - 1)/r:
//       easy part (p^6 - 1)(p^2 + 1), hard part (p^4 - p^2 + 1)/r = l0 + l1 p + l2 p^2 + p^3 with
//       l2 = 6x^2 + 1, l1 = -36x^3 - 18x^2 - 12x + 1, l0 = -36x^3 - 30x^2 - 18x - 2   (x = 4965661367192848881).
#pragma once
#include "ec.cuh"

namespace g16 {

#include "pairing_consts.inc"

struct fp12_t {
  fp2_t g[6];
};
struct fp6_t {
  fp2_t c[3];
};

struct Pairing {
  static constexpr uint64_t BN_X = 4965661367192848881ull;

  static FF_HD fp2_t mul_xi(const fp2_t& a) {  // (a0 + a1 u)(9 + u)
    u256 n0 = Fp::add(Fp::mul_small(a.c0, 8), a.c0), n1 = Fp::add(Fp::mul_small(a.c1, 8), a.c1);
    return fp2_t{Fp::sub(n0, a.c1), Fp::add(n1, a.c0)};
  }
  static FF_HD_COLD fp2_t f2mul(const fp2_t& a, const fp2_t& b) { return Fp2::mul(a, b); }
  static FF_HD_COLD fp2_t f2sqr(const fp2_t& a) { return Fp2::sqr(a); }
  static FF_HD_COLD fp2_t f2inv(const fp2_t& a) { return Fp2::inv(a); }

  static FF_HD fp12_t one() {
    fp12_t r;
    for (int k = 0; k < 6; ++k) r.g[k] = Fp2::zero();
    r.g[0] = Fp2::one();
    return r;
  }
  static FF_HD bool is_one(const fp12_t& a) {
    bool ok = Fp2::eq(a.g[0], Fp2::one());
    for (int k = 1; k < 6; ++k) ok = ok && Fp2::is_zero(a.g[k]);
    return ok;
  }
  // schoolbook product over w, w^6 = xi
  static FF_HD_COLD void mul(fp12_t& r, const fp12_t& a, const fp12_t& b) {
    fp2_t lo[6], hi[6];
    for (int k = 0; k < 6; ++k) lo[k] = hi[k] = Fp2::zero();
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {
#pragma unroll 1
      for (int j = 0; j < 6; ++j) {
        fp2_t t = f2mul(a.g[i], b.g[j]);
        int k = i + j;
        if (k < 6) lo[k] = Fp2::add(lo[k], t);
        else hi[k - 6] = Fp2::add(hi[k - 6], t);
      }
    }
#pragma unroll 1
    for (int k = 0; k < 6; ++k) r.g[k] = Fp2::add(lo[k], mul_xi(hi[k]));
  }
  // square: 15 doubled cross products + 6 squares instead of 36 products
  static FF_HD_COLD void sqr(fp12_t& r, const fp12_t& a) {
    fp2_t lo[6], hi[6];
    for (int k = 0; k < 6; ++k) lo[k] = hi[k] = Fp2::zero();
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {
#pragma unroll 1
      for (int j = i; j < 6; ++j) {
        fp2_t t = i == j ? f2sqr(a.g[i]) : Fp2::dbl(f2mul(a.g[i], a.g[j]));
        int k = i + j;
        if (k < 6) lo[k] = Fp2::add(lo[k], t);
        else hi[k - 6] = Fp2::add(hi[k - 6], t);
      }
    }
#pragma unroll 1
    for (int k = 0; k < 6; ++k) r.g[k] = Fp2::add(lo[k], mul_xi(hi[k]));
  }
  // r = f * (l0 + l1 w + l3 w^3), l0 in Fp
  static FF_HD_COLD void mul_line(fp12_t& f, const u256& l0, const fp2_t& l1, const fp2_t& l3) {
    fp12_t r;
#pragma unroll 1
    for (int k = 0; k < 6; ++k) {
      fp2_t t{Fp::mul(f.g[k].c0, l0), Fp::mul(f.g[k].c1, l0)};
      int k1 = k - 1, k3 = k - 3;
      fp2_t a = f2mul(f.g[k1 < 0 ? k1 + 6 : k1], l1);
      if (k1 < 0) a = mul_xi(a);
      fp2_t b = f2mul(f.g[k3 < 0 ? k3 + 6 : k3], l3);
      if (k3 < 0) b = mul_xi(b);
      r.g[k] = Fp2::add(t, Fp2::add(a, b));
    }
    f = r;
  }
  // x -> x^(p^6): w -> -w
  static FF_HD void conj(fp12_t& a) {
    a.g[1] = Fp2::neg(a.g[1]);
    a.g[3] = Fp2::neg(a.g[3]);
    a.g[5] = Fp2::neg(a.g[5]);
  }
  // x -> x^(p^n), n = 1, 2, 3
  static FF_HD_COLD void frobenius(fp12_t& a, int n) {
#pragma unroll 1
    for (int k = 0; k < 6; ++k) {
      fp2_t t = a.g[k];
      if (n & 1) t.c1 = Fp::neg(t.c1);
      a.g[k] = f2mul(t, PAIRING_GAMMA[n - 1][k]);
    }
  }

  // ---- Fp6 = Fp2[v]/(v^3 - xi) (only for the one inversion of the final exponentiation) ----
  static FF_HD_COLD void f6mul(fp6_t& r, const fp6_t& a, const fp6_t& b) {
    fp2_t t00 = f2mul(a.c[0], b.c[0]), t11 = f2mul(a.c[1], b.c[1]), t22 = f2mul(a.c[2], b.c[2]);
    fp2_t t12 = Fp2::add(f2mul(a.c[1], b.c[2]), f2mul(a.c[2], b.c[1]));
    fp2_t t01 = Fp2::add(f2mul(a.c[0], b.c[1]), f2mul(a.c[1], b.c[0]));
    fp2_t t02 = Fp2::add(f2mul(a.c[0], b.c[2]), f2mul(a.c[2], b.c[0]));
    r.c[0] = Fp2::add(t00, mul_xi(t12));
    r.c[1] = Fp2::add(t01, mul_xi(t22));
    r.c[2] = Fp2::add(t02, t11);
  }
  static FF_HD_COLD void f6inv(fp6_t& r, const fp6_t& a) {
    fp2_t t0 = Fp2::sub(f2sqr(a.c[0]), mul_xi(f2mul(a.c[1], a.c[2])));
    fp2_t t1 = Fp2::sub(mul_xi(f2sqr(a.c[2])), f2mul(a.c[0], a.c[1]));
    fp2_t t2 = Fp2::sub(f2sqr(a.c[1]), f2mul(a.c[0], a.c[2]));
    fp2_t d = Fp2::add(f2mul(a.c[0], t0), mul_xi(Fp2::add(f2mul(a.c[2], t1), f2mul(a.c[1], t2))));
    fp2_t di = f2inv(d);
    r.c[0] = f2mul(t0, di);
    r.c[1] = f2mul(t1, di);
    r.c[2] = f2mul(t2, di);
  }
  // f = A + B w (A = g0,g2,g4; B = g1,g3,g5 over v = w^2):  1/f = (A - B w) / (A^2 - v B^2)
  static FF_HD_COLD void inv(fp12_t& r, const fp12_t& f) {
    fp6_t A{{f.g[0], f.g[2], f.g[4]}}, B{{f.g[1], f.g[3], f.g[5]}}, A2, B2, D, Di, RA, RB;
    f6mul(A2, A, A);
    f6mul(B2, B, B);
    fp6_t vB2{{mul_xi(B2.c[2]), B2.c[0], B2.c[1]}};
    for (int i = 0; i < 3; ++i) D.c[i] = Fp2::sub(A2.c[i], vB2.c[i]);
    f6inv(Di, D);
    f6mul(RA, A, Di);
    f6mul(RB, B, Di);
    for (int i = 0; i < 3; ++i) {
      r.g[2 * i] = RA.c[i];
      r.g[2 * i + 1] = Fp2::neg(RB.c[i]);
    }
  }
  // f^e for a 64-bit exponent
  static FF_HD_COLD void pow_u64(fp12_t& r, const fp12_t& f, uint64_t e) {
    fp12_t acc = one();
#pragma unroll 1
    for (int b = 63; b >= 0; --b) {
      sqr(acc, acc);
      if ((e >> b) & 1) {
        fp12_t t = acc;
        mul(acc, t, f);
      }
    }
    r = acc;
  }

  // Miller loop f_{6x^2,Q}(P); P, Q affine in Montgomery form; (0,0) on either side gives 1
  static FF_HD_COLD void miller(fp12_t& f, const g1_aff& P, const g2_aff& Q) {
    f = one();
    if (G1::is_inf(P) || G2::is_inf(Q)) return;
    // 6x^2 = 0x6f4d8248eeb859fbf83e9682e87cfd46 (127 bits)
    const uint64_t hi = 0x6f4d8248eeb859fbull, lo = 0xf83e9682e87cfd46ull;
    const u256 nxP = Fp::neg(P.x);
    fp2_t tx = Q.x, ty = Q.y;
    bool t_inf = false;
#pragma unroll 1
    for (int b = 125; b >= 0; --b) {
      const bool bit = b >= 64 ? (hi >> (b - 64)) & 1 : (lo >> b) & 1;
      if (t_inf) {  // only possible past the last addition of a point of order r: f is squared, T stays infinity
        sqr(f, f);
        continue;
      }
      // doubling step
      fp2_t lam = f2mul(Fp2::mul_small(f2sqr(tx), 3), f2inv(Fp2::dbl(ty)));
      sqr(f, f);
      mul_line(f, P.y, fp2_t{Fp::mul(lam.c0, nxP), Fp::mul(lam.c1, nxP)}, Fp2::sub(f2mul(lam, tx), ty));
      fp2_t x3 = Fp2::sub(f2sqr(lam), Fp2::dbl(tx));
      ty = Fp2::sub(f2mul(lam, Fp2::sub(tx, x3)), ty);
      tx = x3;
      if (bit) {
        if (Fp2::eq(tx, Q.x)) {  // T == -Q: vertical line, killed by the final exponentiation
          t_inf = true;
          continue;
        }
        lam = f2mul(Fp2::sub(Q.y, ty), f2inv(Fp2::sub(Q.x, tx)));
        mul_line(f, P.y, fp2_t{Fp::mul(lam.c0, nxP), Fp::mul(lam.c1, nxP)}, Fp2::sub(f2mul(lam, tx), ty));
        x3 = Fp2::sub(Fp2::sub(f2sqr(lam), tx), Q.x);
        ty = Fp2::sub(f2mul(lam, Fp2::sub(tx, x3)), ty);
        tx = x3;
      }
    }
  }

  // a^6, a^12, a^18, a^30, a^36
  struct Pows {
    fp12_t p6, p12, p18, p30, p36;
  };
  static FF_HD_COLD void small_pows(Pows& o, const fp12_t& a) {
    fp12_t a2, a3;
    sqr(a2, a);
    mul(a3, a2, a);
    sqr(o.p6, a3);
    sqr(o.p12, o.p6);
    mul(o.p18, o.p12, o.p6);
    mul(o.p30, o.p18, o.p12);
    sqr(o.p36, o.p18);
  }
  // f^((p^12 - 1)/r)
  static FF_HD_COLD void final_exp(fp12_t& r, const fp12_t& f0) {
    // easy part: f^((p^6 - 1)(p^2 + 1))
    fp12_t fi, f, t;
    inv(fi, f0);
    f = f0;
    conj(f);
    mul(t, f, fi);
    f = t;
    frobenius(t, 2);
    {
      fp12_t u = f;
      mul(f, t, u);
    }
    // hard part (f is now in the cyclotomic subgroup: inverse = conjugate)
    fp12_t fx, fx2, fx3;
    pow_u64(fx, f, BN_X);
    pow_u64(fx2, fx, BN_X);
    pow_u64(fx3, fx2, BN_X);
    Pows P1, P2, P3;
    small_pows(P1, fx);
    small_pows(P2, fx2);
    small_pows(P3, fx3);
    fp12_t e0, e1, e2, f2;
    // l0: conj(fx3^36 * fx2^30 * fx^18 * f^2)
    sqr(f2, f);
    mul(t, P3.p36, P2.p30);
    mul(e0, t, P1.p18);
    mul(t, e0, f2);
    e0 = t;
    conj(e0);
    // l1: conj(fx3^36 * fx2^18 * fx^12) * f, then ^p
    mul(t, P3.p36, P2.p18);
    mul(e1, t, P1.p12);
    conj(e1);
    mul(t, e1, f);
    e1 = t;
    frobenius(e1, 1);
    // l2: fx2^6 * f, then ^(p^2)
    mul(e2, P2.p6, f);
    frobenius(e2, 2);
    // l3 = 1: f^(p^3)
    fp12_t e3 = f;
    frobenius(e3, 3);
    mul(t, e0, e1);
    mul(e0, t, e2);
    mul(r, e0, e3);
  }
};

}  // namespace g16
