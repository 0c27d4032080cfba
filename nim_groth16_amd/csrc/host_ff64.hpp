// Host-only BN254 base field on 4 x u64 limbs (unsigned __int128 CIOS Montgomery), with the same static
// interface as g16::Field so that the curve templates of ec.cuh can be instantiated for the O(1) mask algebra
// of the prover (reference prover.nim:279-302: five scalar multiplications and ~10 additions per proof, done on
// the host there as well, curves.nim:136-214).  Same 32-byte element layout as the device type.
#pragma once
#include <stdint.h>
#include <string.h>

namespace g16 {

struct alignas(16) h256 {
  uint64_t v[4];
};

struct HFp {
  using T = h256;
  typedef unsigned __int128 u128;
  static constexpr uint64_t P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL,
                                    0x30644e72e131a029ULL};
  static constexpr uint64_t INV = 0x87d20782e4866389ULL;  // -p^-1 mod 2^64
  static T zero() { return T{{0, 0, 0, 0}}; }
  static T one() {  // R mod p
    return T{{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}};
  }
  static bool is_zero(const T& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
  static bool eq(const T& a, const T& b) {
    return ((a.v[0] ^ b.v[0]) | (a.v[1] ^ b.v[1]) | (a.v[2] ^ b.v[2]) | (a.v[3] ^ b.v[3])) == 0;
  }
  static bool geq_p(const uint64_t* a) {
    for (int i = 3; i >= 0; --i) {
      if (a[i] > P[i]) return true;
      if (a[i] < P[i]) return false;
    }
    return true;
  }
  static void sub_p(uint64_t* a) {
    u128 b = 0;
    for (int i = 0; i < 4; ++i) {
      u128 t = (u128)a[i] - P[i] - (uint64_t)b;
      a[i] = (uint64_t)t;
      b = (t >> 64) & 1;
    }
  }
  static T add(const T& a, const T& b) {
    T r;
    u128 c = 0;
    for (int i = 0; i < 4; ++i) {
      c += (u128)a.v[i] + b.v[i];
      r.v[i] = (uint64_t)c;
      c >>= 64;
    }
    if (geq_p(r.v)) sub_p(r.v);
    return r;
  }
  static T sub(const T& a, const T& b) {
    T r;
    u128 bw = 0;
    for (int i = 0; i < 4; ++i) {
      u128 d = (u128)a.v[i] - b.v[i] - (uint64_t)bw;
      r.v[i] = (uint64_t)d;
      bw = (d >> 64) & 1;
    }
    if (bw) {
      u128 c = 0;
      for (int i = 0; i < 4; ++i) {
        c += (u128)r.v[i] + P[i];
        r.v[i] = (uint64_t)c;
        c >>= 64;
      }
    }
    return r;
  }
  static T neg(const T& a) { return is_zero(a) ? a : sub(zero(), a); }
  static T dbl(const T& a) { return add(a, a); }
  static T mul(const T& a, const T& b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
      u128 c = 0;
      for (int j = 0; j < 4; ++j) {
        c += (u128)a.v[j] * b.v[i] + t[j];
        t[j] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[4] = (uint64_t)c;
      t[5] = (uint64_t)(c >> 64);
      const uint64_t m = t[0] * INV;
      c = ((u128)m * P[0] + t[0]) >> 64;
      for (int j = 1; j < 4; ++j) {
        c += (u128)m * P[j] + t[j];
        t[j - 1] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[3] = (uint64_t)c;
      t[4] = t[5] + (uint64_t)(c >> 64);
    }
    if (t[4] || geq_p(t)) sub_p(t);
    T r;
    memcpy(r.v, t, 32);
    return r;
  }
  static T sqr(const T& a) { return mul(a, a); }
  static T mulsub(const T& a, const T& b, const T& c, const T& d) { return sub(mul(a, b), mul(c, d)); }
  static T inv(const T& a) {  // a^(p-2)
    uint64_t e[4] = {P[0] - 2, P[1], P[2], P[3]};
    T acc = one(), base = a;
    for (int i = 0; i < 256; ++i) {
      if ((e[i >> 6] >> (i & 63)) & 1) acc = mul(acc, base);
      base = sqr(base);
    }
    return acc;
  }
  static T mul_small(const T& a, uint32_t k) {
    if (k == 2) return dbl(a);
    if (k == 3) return add(dbl(a), a);
    if (k == 4) return dbl(dbl(a));
    if (k == 8) return dbl(dbl(dbl(a)));
    return a;
  }
};

struct hfp2_t {
  h256 c0, c1;
};
struct HFp2 {
  using T = hfp2_t;
  static T zero() { return T{HFp::zero(), HFp::zero()}; }
  static T one() { return T{HFp::one(), HFp::zero()}; }
  static bool is_zero(const T& a) { return HFp::is_zero(a.c0) && HFp::is_zero(a.c1); }
  static bool eq(const T& a, const T& b) { return HFp::eq(a.c0, b.c0) && HFp::eq(a.c1, b.c1); }
  static T add(const T& a, const T& b) { return T{HFp::add(a.c0, b.c0), HFp::add(a.c1, b.c1)}; }
  static T sub(const T& a, const T& b) { return T{HFp::sub(a.c0, b.c0), HFp::sub(a.c1, b.c1)}; }
  static T neg(const T& a) { return T{HFp::neg(a.c0), HFp::neg(a.c1)}; }
  static T dbl(const T& a) { return T{HFp::dbl(a.c0), HFp::dbl(a.c1)}; }
  static T mul(const T& a, const T& b) {
    h256 v0 = HFp::mul(a.c0, b.c0), v1 = HFp::mul(a.c1, b.c1);
    h256 s = HFp::mul(HFp::add(a.c0, a.c1), HFp::add(b.c0, b.c1));
    return T{HFp::sub(v0, v1), HFp::sub(HFp::sub(s, v0), v1)};
  }
  static T mulsub(const T& a, const T& b, const T& c, const T& d) { return sub(mul(a, b), mul(c, d)); }
  static T sqr(const T& a) {
    h256 t = HFp::mul(a.c0, a.c1);
    return T{HFp::mul(HFp::add(a.c0, a.c1), HFp::sub(a.c0, a.c1)), HFp::dbl(t)};
  }
  static T inv(const T& a) {
    h256 d = HFp::inv(HFp::add(HFp::sqr(a.c0), HFp::sqr(a.c1)));
    return T{HFp::mul(a.c0, d), HFp::neg(HFp::mul(a.c1, d))};
  }
  static T mul_small(const T& a, uint32_t k) { return T{HFp::mul_small(a.c0, k), HFp::mul_small(a.c1, k)}; }
};

}  // namespace g16
