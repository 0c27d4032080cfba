// BN254 G1/G2 group arithmetic for the MSM kernels.
//
// Affine points use the reference's layout and convention: {x, y} in Montgomery form, point at
// infinity := (0,0)  (reference groth16/bn128/curves.nim:33-50).  Accumulators are extended
// Jacobian "XYZZ" (X, Y, ZZ, ZZZ) with x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity := ZZ == 0.
// XYZZ + affine costs 8M + 2S (vs 11M+5S for Jacobian+affine) and needs no inversion.
//
// All formulas are exact: the unequal-add formula is only used after checking P != +-Q; P == Q
// goes to the doubling formula and P == -Q to infinity.  (A Pippenger bucket can legitimately
// receive P twice or P and -P: snarkjs keys repeat points and signed digits negate them.)
#pragma once
#include "ff.cuh"

namespace g16 {

template <class F>
struct Affine {
  typename F::T x, y;
};
template <class F>
struct Xyzz {
  typename F::T x, y, zz, zzz;
};

template <class F>
struct Curve {
  using Field = F;
  using E = typename F::T;
  using Aff = Affine<F>;
  using Acc = Xyzz<F>;

  static FF_HD bool is_inf(const Aff& p) { return F::is_zero(p.x) && F::is_zero(p.y); }
  static FF_HD bool is_inf(const Acc& p) { return F::is_zero(p.zz); }
  static FF_HD Aff aff_inf() { return Aff{F::zero(), F::zero()}; }
  static FF_HD Acc acc_inf() { return Acc{F::zero(), F::zero(), F::zero(), F::zero()}; }
  static FF_HD Acc from_affine(const Aff& p) {
    if (is_inf(p)) return acc_inf();
    return Acc{p.x, p.y, F::one(), F::one()};
  }
  static FF_HD Aff neg(const Aff& p) { return Aff{p.x, F::neg(p.y)}; }
  static FF_HD Acc neg(const Acc& p) { return Acc{p.x, F::neg(p.y), p.zz, p.zzz}; }

  // 2*P for affine P != inf  (mdbl-2008-s-1)
  static FF_HD Acc dbl_affine(const Aff& p) {
    E u = F::dbl(p.y);
    E v = F::sqr(u);
    E w = F::mul(u, v);
    E s = F::mul(p.x, v);
    E xx = F::sqr(p.x);
    E m = F::add(F::dbl(xx), xx);
    E x3 = F::sub(F::sqr(m), F::dbl(s));
    E y3 = F::mulsub(m, F::sub(s, x3), w, p.y);
    return Acc{x3, y3, v, w};  // y == 0 would give zz = 0 = infinity (cannot happen: odd order)
  }
  // 2*P  (dbl-2008-s-1)
  static FF_HD Acc dbl(const Acc& p) {
    if (is_inf(p)) return p;
    E u = F::dbl(p.y);
    E v = F::sqr(u);
    E w = F::mul(u, v);
    E s = F::mul(p.x, v);
    E xx = F::sqr(p.x);
    E m = F::add(F::dbl(xx), xx);
    E x3 = F::sub(F::sqr(m), F::dbl(s));
    E y3 = F::mulsub(m, F::sub(s, x3), w, p.y);
    return Acc{x3, y3, F::mul(v, p.zz), F::mul(w, p.zzz)};
  }
  // acc += q (affine)   (madd-2008-s)
  static FF_HD void madd(Acc& acc, const Aff& q) {
    if (is_inf(q)) return;
    if (is_inf(acc)) {
      acc = Acc{q.x, q.y, F::one(), F::one()};
      return;
    }
    E u2 = F::mul(q.x, acc.zz);
    E s2 = F::mul(q.y, acc.zzz);
    E p = F::sub(u2, acc.x);
    E r = F::sub(s2, acc.y);
    if (F::is_zero(p)) {
      if (F::is_zero(r)) acc = dbl_affine(q);
      else acc = acc_inf();
      return;
    }
    E pp = F::sqr(p);
    E ppp = F::mul(p, pp);
    E qq = F::mul(acc.x, pp);
    E x3 = F::sub(F::sub(F::sqr(r), ppp), F::dbl(qq));
    E y3 = F::mulsub(r, F::sub(qq, x3), acc.y, ppp);
    acc.x = x3;
    acc.y = y3;
    acc.zz = F::mul(acc.zz, pp);
    acc.zzz = F::mul(acc.zzz, ppp);
  }
  // acc += q (XYZZ)   (add-2008-s)
  static FF_HD void add(Acc& acc, const Acc& q) {
    if (is_inf(q)) return;
    if (is_inf(acc)) {
      acc = q;
      return;
    }
    E u1 = F::mul(acc.x, q.zz);
    E u2 = F::mul(q.x, acc.zz);
    E s1 = F::mul(acc.y, q.zzz);
    E s2 = F::mul(q.y, acc.zzz);
    E p = F::sub(u2, u1);
    E r = F::sub(s2, s1);
    if (F::is_zero(p)) {
      if (F::is_zero(r)) acc = dbl(acc);
      else acc = acc_inf();
      return;
    }
    E pp = F::sqr(p);
    E ppp = F::mul(p, pp);
    E qq = F::mul(u1, pp);
    E x3 = F::sub(F::sub(F::sqr(r), ppp), F::dbl(qq));
    E y3 = F::mulsub(r, F::sub(qq, x3), s1, ppp);
    acc.x = x3;
    acc.y = y3;
    acc.zz = F::mul(F::mul(acc.zz, q.zz), pp);
    acc.zzz = F::mul(F::mul(acc.zzz, q.zzz), ppp);
  }
  // canonical affine form (one field inversion); infinity -> (0,0)  (msm.nim:53-54 `prj.affine`)
  static FF_HD Aff to_affine(const Acc& p) {
    if (is_inf(p)) return aff_inf();
    E t = F::inv(F::mul(p.zz, p.zzz));
    E izz = F::mul(t, p.zzz);
    E izzz = F::mul(t, p.zz);
    return Aff{F::mul(p.x, izz), F::mul(p.y, izzz)};
  }
  // k*P for a small unsigned k (vartime double-and-add)
  static FF_HD Acc mul_small(const Acc& p, uint32_t k) {
    Acc r = acc_inf();
    for (int b = 31; b >= 0; --b) {
      r = dbl(r);
      if ((k >> b) & 1) add(r, p);
    }
    return r;
  }
};

using G1 = Curve<Fp>;
using G2 = Curve<Fp2>;
using g1_aff = Affine<Fp>;
using g2_aff = Affine<Fp2>;
using g1_acc = Xyzz<Fp>;
using g2_acc = Xyzz<Fp2>;

static_assert(sizeof(u256) == 32, "Fr/Fp element must be 32 bytes (SURVEY 8a)");
static_assert(sizeof(g1_aff) == 64, "G1 affine must be 64 bytes");
static_assert(sizeof(g2_aff) == 128, "G2 affine must be 128 bytes");

}  // namespace g16
