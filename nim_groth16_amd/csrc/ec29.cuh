// Bucket accumulation arithmetic on the reduced-radix field of ff29.cuh: XYZZ accumulator += affine table point
// (madd-2008-s: 6M + 2S + one 2-term dot product in G1) and XYZZ += XYZZ (add-2008-s, for the bucket partials of
// msm_heavy / msm_reduce1), for G1 (Fp) and G2 (Fp2 = Fp[u]/(u^2+1)).
//
// The operation sequences below are mirrored line by line by g1_madd / g2_madd / g1_add / g2_add in
// tools/ff29_model.py, which
// proves on worst-case bounds that no column or limb overflows and that the loop invariant
//     G1: X < 10p, Y,ZZ,ZZZ <= 2p        G2 (per component): X < 12p, Y,ZZ,ZZZ <= 3p      (all normalized)
// is preserved.  Table points are canonical (< p).  Exactly as in ec.cuh, P + P and P - P are detected and
// handled exactly (through the 8x32 formulas of ec.cuh: that path is taken ~never and only has to be right).
#pragma once
#include "ec.cuh"
#include "ff29.cuh"

namespace g16 {

struct alignas(8) g1_aff29 {
  fe29 x, y;
};
struct g1_acc29 {
  fe29 x, y, zz, zzz;
};
struct f2e29 {
  fe29 c0, c1;
};
struct alignas(16) g2_aff29 {
  f2e29 x, y;
};
struct g2_acc29 {
  f2e29 x, y, zz, zzz;
};
// Table entries in HBM: the same canonical integers x*2^261 mod p, packed as plain 256-bit little-endian words
// (64 B per G1 point, 128 B per G2 point): one aligned 64 / 128-byte gather per bucket entry.  (Stored as 9
// 32-bit limbs an entry is 72 / 144 B and straddles sectors: rocprofv3 FETCH_SIZE showed 1.96 GB per 2^20 G1
// launch, 2x the bytes asked for.)  Unpacking is bit slicing: ~2 VALU instructions per limb.
struct alignas(16) g1_tab29 {
  u256 x, y;
};
struct alignas(16) g2_tab29 {
  u256 x0, x1, y0, y1;
};
static_assert(sizeof(g1_tab29) == 64 && sizeof(g2_tab29) == 128, "packed table entry sizes");

template <class C>
struct Ec29;

// ---------------------------------------------------------------------------------------------------------
template <>
struct Ec29<G1> {
  using F = Fp29;
  using Aff = g1_aff29;
  using Acc = g1_acc29;

  using Tab = g1_tab29;
  static FF_HD Aff from_std(const g1_aff& p) { return Aff{F::from_std(p.x), F::from_std(p.y)}; }
  static FF_HD Tab pack(const Aff& p) { return Tab{F::relimb(p.x), F::relimb(p.y)}; }   // p canonical
  static FF_HD Aff unpack(const Tab& t) { return Aff{F::relimb(t.x), F::relimb(t.y)}; }
  static FF_HD Tab tab_from_std(const g1_aff& p) { return pack(from_std(p)); }
  static FF_HD bool is_inf(const Aff& p) { return F::limbs_zero(p.x) && F::limbs_zero(p.y); }
  static FF_HD bool is_inf(const Acc& a) { return F::limbs_zero(a.zz); }
  static FF_HD Acc acc_inf() { return Acc{F::zero(), F::zero(), F::zero(), F::zero()}; }
  static FF_HD g1_acc to_std(const Acc& a) {
    if (is_inf(a)) return G1::acc_inf();
    return g1_acc{F::to_std(a.x), F::to_std(a.y), F::to_std(a.zz), F::to_std(a.zzz)};
  }
  static FF_HD Acc acc_from_std(const g1_acc& a) {
    if (G1::is_inf(a)) return acc_inf();
    return Acc{F::from_std(a.x), F::from_std(a.y), F::from_std(a.zz), F::from_std(a.zzz)};
  }
  // the exceptional cases of the addition (q == +-acc as group elements); by value: a by-reference call would
  // pin the accumulator to scratch memory
  static FF_HD_COLD Acc exceptional(const Tab* tp, uint32_t neg, bool same) {
    if (!same) return acc_inf();
    const Aff q = unpack(*tp);   // re-read: keeping the entry live across the hot path would cost registers
    g1_aff s{F::to_std(q.x), F::to_std(q.y)};
    if (neg) s = G1::neg(s);
    return acc_from_std(G1::dbl_affine(s));
  }

  static FF_HD bool is_inf(const Tab& t) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= t.x.v[i] | t.y.v[i];
    return o == 0;
  }
  // acc += (neg ? -q : q), q given as a packed table entry
  static FF_HD void madd(Acc& acc, const Tab* tp, uint32_t neg) {
    const Tab t = *tp;
    if (is_inf(t)) return;
    fe29 qx = F::relimb(t.x), qy = F::relimb(t.y);
    if (neg) qy = F::norm(F::negk<2, 1>(qy));
    if (is_inf(acc)) {
      acc = Acc{qx, qy, F::one(), F::one()};
      return;
    }
    fe29 u2 = F::mul(qx, acc.zz);
    fe29 s2 = F::mul(qy, acc.zzz);
    fe29 nP = F::norm(F::subk<16, 1>(u2, acc.x));
    fe29 nR = F::norm(F::subk<4, 1>(s2, acc.y));
    if (F::maybe_zero(nP)) {
      if (F::is_zero_exact<16>(nP)) {
        acc = exceptional(tp, neg, F::is_zero_exact<4>(nR));
        return;
      }
    }
    fe29 pp = F::sqr(nP);
    fe29 ppp = F::mul(nP, pp);
    fe29 qq = F::mul(acc.x, pp);
    acc.zz = F::mul(acc.zz, pp);
    acc.zzz = F::mul(acc.zzz, ppp);
    fe29 r2 = F::sqr(nR);
    fe29 x3 = F::norm(F::subk2<8, 3>(r2, ppp, qq));
    fe29 d = F::subk<16, 1>(qq, x3);
    fe29 yn = F::negk<4, 1>(acc.y);
    acc.y = F::dot2(nR, d, yn, ppp);
    acc.x = x3;
  }
  // the exceptional cases of a general addition: through the 8x32 formulas
  static FF_HD_COLD Acc add_exceptional(Acc a, Acc b) {
    g1_acc sa = to_std(a);
    G1::add(sa, to_std(b));
    return acc_from_std(sa);
  }
  // acc += q, both XYZZ (add-2008-s): 10 mul + 2 sqr + one 2-term dot; mirrored by g1_add in tools/ff29_model.py
  static FF_HD void add(Acc& acc, const Acc& q) {
    if (is_inf(q)) return;
    if (is_inf(acc)) {
      acc = q;
      return;
    }
    fe29 u1 = F::mul(acc.x, q.zz);
    fe29 u2 = F::mul(q.x, acc.zz);
    fe29 s1 = F::mul(acc.y, q.zzz);
    fe29 s2 = F::mul(q.y, acc.zzz);
    fe29 nP = F::norm(F::subk<4, 1>(u2, u1));
    fe29 nR = F::norm(F::subk<4, 1>(s2, s1));
    if (F::maybe_zero(nP)) {
      if (F::is_zero_exact<4>(nP)) {
        acc = add_exceptional(acc, q);
        return;
      }
    }
    fe29 pp = F::sqr(nP);
    fe29 ppp = F::mul(nP, pp);
    fe29 qq = F::mul(u1, pp);
    acc.zz = F::mul(F::mul(acc.zz, q.zz), pp);
    acc.zzz = F::mul(F::mul(acc.zzz, q.zzz), ppp);
    fe29 r2 = F::sqr(nR);
    fe29 x3 = F::norm(F::subk2<8, 3>(r2, ppp, qq));
    fe29 d = F::subk<16, 1>(qq, x3);
    fe29 s1n = F::negk<4, 1>(s1);
    acc.y = F::dot2(nR, d, s1n, ppp);
    acc.x = x3;
  }
};

// ---------------------------------------------------------------------------------------------------------
template <>
struct Ec29<G2> {
  using F = Fp29;
  using Aff = g2_aff29;
  using Acc = g2_acc29;

  static FF_HD f2e29 f2_from_std(const fp2_t& a) { return f2e29{F::from_std(a.c0), F::from_std(a.c1)}; }
  static FF_HD fp2_t f2_to_std(const f2e29& a) { return fp2_t{F::to_std(a.c0), F::to_std(a.c1)}; }
  static FF_HD bool f2_limbs_zero(const f2e29& a) { return F::limbs_zero(a.c0) && F::limbs_zero(a.c1); }
  static FF_HD f2e29 f2_zero() { return f2e29{F::zero(), F::zero()}; }
  static FF_HD f2e29 f2_one() { return f2e29{F::one(), F::zero()}; }

  // (a0 b0 - a1 b1, a0 b1 + a1 b0): two 2-term dot products; all four inputs normalized,
  // kform<KM,1> covers a.c1
  template <uint32_t KM>
  static FF_HD f2e29 f2mul(const f2e29& a, const f2e29& b) {
    fe29 n1 = F::negk<KM, 1>(a.c1);
#if defined(G16_F29_PAIR)
    f2e29 r;
    F::dot_pair<2>(r.c0, r.c1, a.c0, b.c0, n1, b.c1, n1, n1, n1, n1, a.c0, b.c1, a.c1, b.c0, n1, n1, n1, n1);
    return r;
#else
    return f2e29{F::dot2(a.c0, b.c0, n1, b.c1), F::dot2(a.c0, b.c1, a.c1, b.c0)};
#endif
  }
  // ((a0+a1)(a0-a1), 2 a0 a1)
  template <uint32_t KM>
  static FF_HD f2e29 f2sqr(const f2e29& a) {
    fe29 s = F::add(a.c0, a.c1);
    fe29 d = F::subk<KM, 1>(a.c0, a.c1);
#if defined(G16_F29_PAIR)
    fe29 t = F::add(a.c0, a.c0);
    f2e29 r;
    F::dot_pair<1>(r.c0, r.c1, s, d, s, s, s, s, s, s, t, a.c1, s, s, s, s, s, s);
    return r;
#else
    return f2e29{F::mul(s, d), F::mul(F::add(a.c0, a.c0), a.c1)};
#endif
  }

  using Tab = g2_tab29;
  static FF_HD Aff from_std(const g2_aff& p) { return Aff{f2_from_std(p.x), f2_from_std(p.y)}; }
  static FF_HD Tab pack(const Aff& p) {   // p canonical
    return Tab{F::relimb(p.x.c0), F::relimb(p.x.c1), F::relimb(p.y.c0), F::relimb(p.y.c1)};
  }
  static FF_HD Aff unpack(const Tab& t) {
    return Aff{f2e29{F::relimb(t.x0), F::relimb(t.x1)}, f2e29{F::relimb(t.y0), F::relimb(t.y1)}};
  }
  static FF_HD Tab tab_from_std(const g2_aff& p) { return pack(from_std(p)); }
  static FF_HD bool is_inf(const Aff& p) { return f2_limbs_zero(p.x) && f2_limbs_zero(p.y); }
  static FF_HD bool is_inf(const Acc& a) { return f2_limbs_zero(a.zz); }
  static FF_HD Acc acc_inf() { return Acc{f2_zero(), f2_zero(), f2_zero(), f2_zero()}; }
  static FF_HD g2_acc to_std(const Acc& a) {
    if (is_inf(a)) return G2::acc_inf();
    return g2_acc{f2_to_std(a.x), f2_to_std(a.y), f2_to_std(a.zz), f2_to_std(a.zzz)};
  }
  static FF_HD Acc acc_from_std(const g2_acc& a) {
    if (G2::is_inf(a)) return acc_inf();
    return Acc{f2_from_std(a.x), f2_from_std(a.y), f2_from_std(a.zz), f2_from_std(a.zzz)};
  }
  static FF_HD_COLD Acc exceptional(const Tab* tp, uint32_t neg, bool same) {
    if (!same) return acc_inf();
    const Aff q = unpack(*tp);
    g2_aff s{f2_to_std(q.x), f2_to_std(q.y)};
    if (neg) s = G2::neg(s);
    return acc_from_std(G2::dbl_affine(s));
  }

  static FF_HD bool is_inf(const Tab& t) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= t.x0.v[i] | t.x1.v[i] | t.y0.v[i] | t.y1.v[i];
    return o == 0;
  }
  // acc += (neg ? -q : q), q given as a packed table entry: each coordinate is unpacked where it is consumed
  // (unpacking all 36 limbs up front costs registers the G2 kernel does not have)
  static FF_HD void madd(Acc& acc, const Tab* tp, uint32_t neg) {
    const Tab t = *tp;
    if (is_inf(t)) return;
    if (is_inf(acc)) {
      Aff q = unpack(t);
      if (neg) q.y = f2e29{F::norm(F::negk<2, 1>(q.y.c0)), F::norm(F::negk<2, 1>(q.y.c1))};
      acc = Acc{q.x, q.y, f2_one(), f2_one()};
      return;
    }
    f2e29 u2 = f2mul<4>(f2e29{F::relimb(t.x0), F::relimb(t.x1)}, acc.zz);
    f2e29 qy{F::relimb(t.y0), F::relimb(t.y1)};
    if (neg) qy = f2e29{F::norm(F::negk<2, 1>(qy.c0)), F::norm(F::negk<2, 1>(qy.c1))};
    f2e29 s2 = f2mul<4>(qy, acc.zzz);
    f2e29 nP{F::norm(F::subk<16, 1>(u2.c0, acc.x.c0)), F::norm(F::subk<16, 1>(u2.c1, acc.x.c1))};
    f2e29 nR{F::norm(F::subk<4, 1>(s2.c0, acc.y.c0)), F::norm(F::subk<4, 1>(s2.c1, acc.y.c1))};
    if (F::maybe_zero(nP.c0) && F::maybe_zero(nP.c1)) {
      if (F::is_zero_exact<16>(nP.c0) && F::is_zero_exact<16>(nP.c1)) {
        acc = exceptional(tp, neg, F::is_zero_exact<4>(nR.c0) && F::is_zero_exact<4>(nR.c1));
        return;
      }
    }
    // (ordered for short live ranges: pp dies after zz, acc.x after qq)
    f2e29 pp = f2sqr<32>(nP);
    f2e29 ppp = f2mul<32>(nP, pp);
    f2e29 qq = f2mul<16>(acc.x, pp);
    acc.zz = f2mul<4>(acc.zz, pp);
    acc.zzz = f2mul<4>(acc.zzz, ppp);
    f2e29 r2 = f2sqr<8>(nR);
    f2e29 x3{F::norm(F::subk2<8, 3>(r2.c0, ppp.c0, qq.c0)), F::norm(F::subk2<8, 3>(r2.c1, ppp.c1, qq.c1))};
    f2e29 d{F::norm(F::subk<16, 1>(qq.c0, x3.c0)), F::norm(F::subk<16, 1>(qq.c1, x3.c1))};
    // Y3 = R*D - Y*PPP:  c0 = r0 d0 - r1 d1 - y0 t0 + y1 t1,  c1 = r0 d1 + r1 d0 - y0 t1 - y1 t0
    fe29 nr1 = F::norm(F::negk<8, 1>(nR.c1));
    fe29 ny0 = F::norm(F::negk<4, 1>(acc.y.c0));
    fe29 ny1 = F::norm(F::negk<4, 1>(acc.y.c1));
#if defined(G16_F29_PAIR)
    f2e29 y3;
    F::dot_pair<4>(y3.c0, y3.c1, nR.c0, d.c0, nr1, d.c1, ny0, ppp.c0, acc.y.c1, ppp.c1,
                   nR.c0, d.c1, nR.c1, d.c0, ny0, ppp.c1, ny1, ppp.c0);
#else
    f2e29 y3{F::dot4(nR.c0, d.c0, nr1, d.c1, ny0, ppp.c0, acc.y.c1, ppp.c1),
             F::dot4(nR.c0, d.c1, nR.c1, d.c0, ny0, ppp.c1, ny1, ppp.c0)};
#endif
    acc.x = x3;
    acc.y = y3;
  }
  static FF_HD_COLD Acc add_exceptional(Acc a, Acc b) {
    g2_acc sa = to_std(a);
    G2::add(sa, to_std(b));
    return acc_from_std(sa);
  }
  // acc += q, both XYZZ; mirrored by g2_add in tools/ff29_model.py
  static FF_HD void add(Acc& acc, const Acc& q) {
    if (is_inf(q)) return;
    if (is_inf(acc)) {
      acc = q;
      return;
    }
    f2e29 u1 = f2mul<16>(acc.x, q.zz);
    f2e29 u2 = f2mul<16>(q.x, acc.zz);
    f2e29 s1 = f2mul<4>(acc.y, q.zzz);
    f2e29 s2 = f2mul<4>(q.y, acc.zzz);
    f2e29 nP{F::norm(F::subk<4, 1>(u2.c0, u1.c0)), F::norm(F::subk<4, 1>(u2.c1, u1.c1))};
    f2e29 nR{F::norm(F::subk<4, 1>(s2.c0, s1.c0)), F::norm(F::subk<4, 1>(s2.c1, s1.c1))};
    if (F::maybe_zero(nP.c0) && F::maybe_zero(nP.c1)) {
      if (F::is_zero_exact<4>(nP.c0) && F::is_zero_exact<4>(nP.c1)) {
        acc = add_exceptional(acc, q);
        return;
      }
    }
    f2e29 pp = f2sqr<8>(nP);
    f2e29 ppp = f2mul<8>(nP, pp);
    f2e29 qq = f2mul<4>(u1, pp);
    acc.zz = f2mul<4>(f2mul<4>(acc.zz, q.zz), pp);
    acc.zzz = f2mul<4>(f2mul<4>(acc.zzz, q.zzz), ppp);
    f2e29 r2 = f2sqr<8>(nR);
    f2e29 x3{F::norm(F::subk2<8, 3>(r2.c0, ppp.c0, qq.c0)), F::norm(F::subk2<8, 3>(r2.c1, ppp.c1, qq.c1))};
    f2e29 d{F::norm(F::subk<16, 1>(qq.c0, x3.c0)), F::norm(F::subk<16, 1>(qq.c1, x3.c1))};
    fe29 nr1 = F::norm(F::negk<8, 1>(nR.c1));
    fe29 ns0 = F::norm(F::negk<4, 1>(s1.c0));
    fe29 ns1 = F::norm(F::negk<4, 1>(s1.c1));
    f2e29 y3{F::dot4(nR.c0, d.c0, nr1, d.c1, ns0, ppp.c0, s1.c1, ppp.c1),
             F::dot4(nR.c0, d.c1, nR.c1, d.c0, ns0, ppp.c1, ns1, ppp.c0)};
    acc.x = x3;
    acc.y = y3;
  }
};

}  // namespace g16
