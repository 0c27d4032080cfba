#!/usr/bin/env python3
"""Interval model of the 9x29-bit lazy Montgomery arithmetic in nim_groth16_amd/csrc/ff29.cuh / ec29.cuh.

Every element carries concrete limbs AND worst-case bounds (per-limb upper bounds + a value upper bound).
Each operation asserts, on the BOUNDS (not on the sample values), that
  * no 64-bit column accumulator overflows,   * no 32-bit limb overflows or goes negative,
and that the concrete result is congruent to the exact field result.  Running one mixed addition from the
loop-invariant bounds and getting bounds back inside the invariant is therefore a proof for all inputs.
Run: python3 tools/ff29_model.py   (also used by tests/test_device_headers_cpu.py)
"""
import random

p = 21888242871839275222246405745257275088696311157297823662689037894645226208583
B, L = 29, 9
MASK = (1 << B) - 1
R = 1 << (B * L)
RINV = pow(R, -1, p)
TOPSH = B * (L - 1)


def limbs(x):
    return [(x >> (B * i)) & MASK for i in range(L - 1)] + [x >> TOPSH]


def val(v):
    return sum(x << (B * i) for i, x in enumerate(v))


PL = limbs(p)
N0 = (-pow(p, -1, 1 << B)) % (1 << B)


class E:
    """element: concrete limbs v, limb bounds lb (inclusive max), value bound vb (inclusive max)"""

    def __init__(self, v, lb, vb):
        self.v, self.lb, self.vb = list(v), list(lb), vb
        assert len(v) == L
        for x, m in zip(v, lb):
            assert 0 <= x <= m < (1 << 32), ("limb out of bound / u32 overflow", x, m)
        assert val(v) <= vb

    @staticmethod
    def normalized(x, vmax):
        """normalized limbs, value in [0, vmax]"""
        assert 0 <= x <= vmax
        return E(limbs(x), [MASK] * (L - 1) + [vmax >> TOPSH], vmax)

    def __repr__(self):
        return f"E(vb={self.vb / p:.2f}p, lb=[{', '.join(f'{m.bit_length()}' for m in self.lb)}])"


def kform(mult, lift):
    """limbs of mult*p with `lift` units borrowed from each higher limb into the one below"""
    k = limbs(mult * p)
    for i in range(L - 1):
        k[i] += lift << B
        k[i + 1] -= lift
    assert val(k) == mult * p and all(0 <= x < (1 << 32) for x in k)
    return k


def kfor(b, extra=None):
    """smallest (mult, lift) whose borrow form covers the limb bounds of b (plus optionally those of `extra`)"""
    need = list(b.lb)
    if extra is not None:
        need = [x + y for x, y in zip(need, extra)]
    lift = 0
    while (lift << B) - lift < max(need[:L - 1]):   # K_i >= lift*2^29 - lift for i < L-1 (k_i >= 0)
        lift += 1
    mult = 1
    while True:
        k = limbs(mult * p)
        kk = [k[0] + (lift << B)] + [k[i] + (lift << B) - lift for i in range(1, L - 1)] + [k[L - 1] - lift]
        if all(x >= n for x, n in zip(kk, need)):
            return mult, lift
        mult *= 2


USED_K = set()
maxcol = 0


def mont(*pairs):
    """(sum of a*b over the pairs) / R mod p, product scanning with one 64-bit column accumulator"""
    global maxcol
    m = [0] * L
    mb = [MASK] * L
    r, acc, accb = [0] * L, 0, 0
    for k in range(2 * L - 1):
        lo, hi = max(0, k - L + 1), min(k, L - 1)
        for i in range(lo, hi + 1):
            for a, b in pairs:
                acc += a.v[i] * b.v[k - i]
                accb += a.lb[i] * b.lb[k - i]
        for i in range(lo, hi + 1):
            if k < L and i == k:
                continue
            acc += m[i] * PL[k - i]
            accb += mb[i] * PL[k - i]
        if k < L:
            m[k] = ((acc & 0xffffffff) * N0) & MASK
            acc += m[k] * PL[0]
            accb += mb[k] * PL[0]
            assert acc & MASK == 0
        assert accb < (1 << 64), ("column overflow", k, accb.bit_length())
        maxcol = max(maxcol, accb)
        if k >= L:
            r[k - L] = acc & MASK
        acc >>= B
        accb >>= B
    r[L - 1] = acc
    vb = sum(a.vb * b.vb for a, b in pairs) // R + p
    assert vb >> TOPSH < (1 << 32)
    out = E(r, [MASK] * (L - 1) + [min(accb, vb >> TOPSH)], vb)
    assert val(r) % p == sum(val(a.v) * val(b.v) for a, b in pairs) * RINV % p
    return out


def sqr(a):
    """dedicated squaring: cross products once against the doubled operand (limbs of 2a must stay < 2^32)"""
    global maxcol
    a2 = [2 * x for x in a.lb]
    assert all(x < (1 << 32) for x in a2)
    # column bound of the a*a part: sum_{i<j} a_i*(2a_j) + a_(k/2)^2  -- identical value to the full product
    m = [0] * L
    mb = [MASK] * L
    r, acc, accb = [0] * L, 0, 0
    for k in range(2 * L - 1):
        lo, hi = max(0, k - L + 1), min(k, L - 1)
        for i in range(lo, hi + 1):
            j = k - i
            if i < j:
                acc += a.v[i] * (2 * a.v[j])
                accb += a.lb[i] * a2[j]
            elif i == j:
                acc += a.v[i] * a.v[i]
                accb += a.lb[i] * a.lb[i]
        for i in range(lo, hi + 1):
            if k < L and i == k:
                continue
            acc += m[i] * PL[k - i]
            accb += mb[i] * PL[k - i]
        if k < L:
            m[k] = ((acc & 0xffffffff) * N0) & MASK
            acc += m[k] * PL[0]
            accb += mb[k] * PL[0]
        assert accb < (1 << 64), ("column overflow", k)
        maxcol = max(maxcol, accb)
        if k >= L:
            r[k - L] = acc & MASK
        acc >>= B
        accb >>= B
    r[L - 1] = acc
    vb = a.vb * a.vb // R + p
    out = E(r, [MASK] * (L - 1) + [min(accb, vb >> TOPSH)], vb)
    assert val(r) % p == val(a.v) ** 2 * RINV % p
    return out


def add(a, b):
    return E([x + y for x, y in zip(a.v, b.v)], [x + y for x, y in zip(a.lb, b.lb)], a.vb + b.vb)


def subk(a, b, K=None, twice=None):
    """a + K - b [- 2*twice] limbwise; K (mult, lift) must cover b's (and twice's) limb bounds"""
    need = list(b.lb)
    if twice is not None:
        need = [x + 2 * y for x, y in zip(need, twice.lb)]
    if K is None:
        K = kfor(b, [2 * y for y in twice.lb] if twice is not None else None)
    USED_K.add(K)
    k = kform(*K)
    assert all(x >= n for x, n in zip(k, need)), ("K does not cover subtrahend", K)
    v = [x + kk - y - (2 * t if twice is not None else 0)
         for x, kk, y, t in zip(a.v, k, b.v, twice.v if twice is not None else [0] * L)]
    lbs = [x + kk for x, kk in zip(a.lb, k)]
    return E(v, lbs, a.vb + K[0] * p)


def negk(b, K=None):
    return subk(E([0] * L, [0] * L, 0), b, K)


def norm(a):
    c, cb, v, lb = 0, 0, [], []
    for i in range(L - 1):
        t, tb = a.v[i] + c, a.lb[i] + cb
        assert tb < (1 << 32)
        v.append(t & MASK)
        lb.append(MASK)
        c, cb = t >> B, tb >> B
    t, tb = a.v[L - 1] + c, a.lb[L - 1] + cb
    assert tb < (1 << 32)
    v.append(t)
    lb.append(min(tb, a.vb >> TOPSH))
    return E(v, lb, a.vb)


def fval(a):
    return val(a.v) * RINV % p      # the field element represented (Montgomery radix 2^261)


# ---------------------------------------------------------------------------------------------------------
# G1 mixed addition, XYZZ += affine (madd-2008-s), exactly the op sequence of ec29.cuh
# ---------------------------------------------------------------------------------------------------------
G1_INV = dict(X=10, Y=2, ZZ=2, ZZZ=2)     # loop invariant: normalized limbs, value <= k*p


def g1_madd(X, Y, ZZ, ZZZ, qx, qy, log=None):
    u2 = mont((qx, ZZ))
    s2 = mont((qy, ZZZ))
    nP = norm(subk(u2, X, (16, 1)))
    nR = norm(subk(s2, Y, (4, 1)))
    PP = sqr(nP)
    PPP = mont((nP, PP))
    Q = mont((X, PP))
    R2 = sqr(nR)
    X3 = norm(subk(R2, PPP, (8, 3), twice=Q))
    D = subk(Q, X3, (16, 1))
    Yn = negk(Y, (4, 1))
    Y3 = mont((nR, D), (Yn, PPP))
    ZZ3 = mont((ZZ, PP))
    ZZZ3 = mont((ZZZ, PPP))
    if log is not None:
        log.update(u2=u2, nP=nP, nR=nR, PP=PP, PPP=PPP, Q=Q, R2=R2, X3=X3, D=D, Y3=Y3, ZZ3=ZZ3, ZZZ3=ZZZ3)
    return X3, Y3, ZZ3, ZZZ3


def check_g1(rnd):
    inv = G1_INV
    for trial in range(200):
        def pick(k):
            x = rnd.choice([0, k * p, rnd.randrange(k * p + 1)])
            return E.normalized(x, k * p)
        X, Y, ZZ, ZZZ = pick(inv["X"]), pick(inv["Y"]), pick(inv["ZZ"]), pick(inv["ZZZ"])
        qx, qy = E.normalized(rnd.randrange(p), p - 1), E.normalized(rnd.randrange(2 * p + 1), 2 * p)
        log = {}
        X3, Y3, ZZ3, ZZZ3 = g1_madd(X, Y, ZZ, ZZZ, qx, qy, log)
        for nm, o in (("X", X3), ("Y", Y3), ("ZZ", ZZ3), ("ZZZ", ZZZ3)):
            assert o.vb <= inv[nm] * p, (nm, o.vb / p)
            assert all(m <= MASK for m in o.lb[:L - 1]), nm
        # exact formulas
        x, y, zz, zzz, ax, ay = map(fval, (X, Y, ZZ, ZZZ, qx, qy))
        P_ = (ax * zz - x) % p
        R_ = (ay * zzz - y) % p
        pp, ppp = P_ * P_ % p, P_ * P_ * P_ % p
        q = x * pp % p
        x3 = (R_ * R_ - ppp - 2 * q) % p
        y3 = (R_ * (q - x3) - y * ppp) % p
        assert (fval(X3), fval(Y3), fval(ZZ3), fval(ZZZ3)) == (x3, y3, zz * pp % p, zzz * ppp % p)
    return log


# ---------------------------------------------------------------------------------------------------------
# G2: Fp2 = Fp[u]/(u^2+1); an element is a pair (c0, c1) of E
# ---------------------------------------------------------------------------------------------------------
G2_INV = dict(X=12, Y=3, ZZ=3, ZZZ=3)


def f2mul(a, b, Kneg=None):
    """(a0 b0 - a1 b1, a0 b1 + a1 b0); all four inputs normalized"""
    n1 = negk(a[1], Kneg)
    return mont((a[0], b[0]), (n1, b[1])), mont((a[0], b[1]), (a[1], b[0]))


def f2sqr(a, Kneg=None):
    """((a0+a1)(a0-a1), 2 a0 a1) as two dot-free products"""
    s = add(a[0], a[1])
    d = subk(a[0], a[1], Kneg)
    return mont((s, d)), mont((add(a[0], a[0]), a[1]))


def f2val(a):
    return (fval(a[0]), fval(a[1]))


def g2_madd(X, Y, ZZ, ZZZ, qx, qy, log=None):
    u2 = f2mul(qx, ZZ, (4, 1))            # negates qx.c1 < 2p (canonical point coordinates, possibly negated y)
    s2 = f2mul(qy, ZZZ, (4, 1))
    nP = tuple(norm(subk(u2[i], X[i], (16, 1))) for i in range(2))
    nR = tuple(norm(subk(s2[i], Y[i], (4, 1))) for i in range(2))
    PP = f2sqr(nP, (32, 1))
    PPP = f2mul(nP, PP, (32, 1))
    Q = f2mul(X, PP, (16, 1))
    R2 = f2sqr(nR, (8, 1))
    X3 = tuple(norm(subk(R2[i], PPP[i], (8, 3), twice=Q[i])) for i in range(2))
    D = tuple(norm(subk(Q[i], X3[i], (16, 1))) for i in range(2))
    # Y3 = R*D - Y*PPP:  c0 = r0 d0 - r1 d1 - y0 t0 + y1 t1 ;  c1 = r0 d1 + r1 d0 - y0 t1 - y1 t0
    nr1 = norm(negk(nR[1], (8, 1)))
    ny0 = norm(negk(Y[0], (4, 1)))
    ny1 = norm(negk(Y[1], (4, 1)))
    Y3 = (mont((nR[0], D[0]), (nr1, D[1]), (ny0, PPP[0]), (Y[1], PPP[1])),
          mont((nR[0], D[1]), (nR[1], D[0]), (ny0, PPP[1]), (ny1, PPP[0])))
    ZZ3 = f2mul(ZZ, PP, (4, 1))
    ZZZ3 = f2mul(ZZZ, PPP, (4, 1))
    if log is not None:
        log.update(u2=u2, nP=nP, nR=nR, PP=PP, PPP=PPP, Q=Q, R2=R2, X3=X3, D=D, Y3=Y3, ZZ3=ZZ3, ZZZ3=ZZZ3)
    return X3, Y3, ZZ3, ZZZ3


def check_g2(rnd):
    inv = G2_INV

    def m2(a, b):
        return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def s2_(a, b):
        return ((a[0] - b[0]) % p, (a[1] - b[1]) % p)

    for trial in range(200):
        def pick(k):
            return tuple(E.normalized(rnd.choice([0, k * p, rnd.randrange(k * p + 1)]), k * p) for _ in range(2))
        X, Y, ZZ, ZZZ = pick(inv["X"]), pick(inv["Y"]), pick(inv["ZZ"]), pick(inv["ZZZ"])
        qx = tuple(E.normalized(rnd.randrange(p), p - 1) for _ in range(2))
        qy = tuple(E.normalized(rnd.randrange(2 * p + 1), 2 * p) for _ in range(2))
        log = {}
        X3, Y3, ZZ3, ZZZ3 = g2_madd(X, Y, ZZ, ZZZ, qx, qy, log)
        for nm, o in (("X", X3), ("Y", Y3), ("ZZ", ZZ3), ("ZZZ", ZZZ3)):
            for c in o:
                assert c.vb <= inv[nm] * p, (nm, c.vb / p)
                assert all(m <= MASK for m in c.lb[:L - 1]), nm
        x, y, zz, zzz, ax, ay = map(f2val, (X, Y, ZZ, ZZZ, qx, qy))
        P_ = s2_(m2(ax, zz), x)
        R_ = s2_(m2(ay, zzz), y)
        pp = m2(P_, P_)
        ppp = m2(P_, pp)
        q = m2(x, pp)
        rr = m2(R_, R_)
        x3 = ((rr[0] - ppp[0] - 2 * q[0]) % p, (rr[1] - ppp[1] - 2 * q[1]) % p)
        y3 = s2_(m2(R_, s2_(q, x3)), m2(y, ppp))
        assert (f2val(X3), f2val(Y3), f2val(ZZ3), f2val(ZZZ3)) == (x3, y3, m2(zz, pp), m2(zzz, ppp))
    return log


# ---------------------------------------------------------------------------------------------------------
# general additions XYZZ += XYZZ (add-2008-s) used by the bucket-reduction kernels; both operands satisfy the
# same invariants as above and so does the result
# ---------------------------------------------------------------------------------------------------------
def g1_add(A, Bq, log=None):
    X1, Y1, ZZ1, ZZZ1 = A
    X2, Y2, ZZ2, ZZZ2 = Bq
    u1 = mont((X1, ZZ2))
    u2 = mont((X2, ZZ1))
    s1 = mont((Y1, ZZZ2))
    s2 = mont((Y2, ZZZ1))
    nP = norm(subk(u2, u1, (4, 1)))
    nR = norm(subk(s2, s1, (4, 1)))
    PP = sqr(nP)
    PPP = mont((nP, PP))
    Q = mont((u1, PP))
    R2 = sqr(nR)
    X3 = norm(subk(R2, PPP, (8, 3), twice=Q))
    D = subk(Q, X3, (16, 1))
    s1n = negk(s1, (4, 1))
    Y3 = mont((nR, D), (s1n, PPP))
    ZZ3 = mont((mont((ZZ1, ZZ2)), PP))
    ZZZ3 = mont((mont((ZZZ1, ZZZ2)), PPP))
    if log is not None:
        log.update(u1=u1, nP=nP, nR=nR, PP=PP, PPP=PPP, Q=Q, X3=X3, Y3=Y3, ZZ3=ZZ3, ZZZ3=ZZZ3)
    return X3, Y3, ZZ3, ZZZ3


def g2_add(A, Bq, log=None):
    X1, Y1, ZZ1, ZZZ1 = A
    X2, Y2, ZZ2, ZZZ2 = Bq
    u1 = f2mul(X1, ZZ2, (16, 1))
    u2 = f2mul(X2, ZZ1, (16, 1))
    s1 = f2mul(Y1, ZZZ2, (4, 1))
    s2 = f2mul(Y2, ZZZ1, (4, 1))
    nP = tuple(norm(subk(u2[i], u1[i], (4, 1))) for i in range(2))
    nR = tuple(norm(subk(s2[i], s1[i], (4, 1))) for i in range(2))
    PP = f2sqr(nP, (8, 1))
    PPP = f2mul(nP, PP, (8, 1))
    Q = f2mul(u1, PP, (4, 1))
    R2 = f2sqr(nR, (8, 1))
    X3 = tuple(norm(subk(R2[i], PPP[i], (8, 3), twice=Q[i])) for i in range(2))
    D = tuple(norm(subk(Q[i], X3[i], (16, 1))) for i in range(2))
    nr1 = norm(negk(nR[1], (8, 1)))
    ns0 = norm(negk(s1[0], (4, 1)))
    ns1 = norm(negk(s1[1], (4, 1)))
    Y3 = (mont((nR[0], D[0]), (nr1, D[1]), (ns0, PPP[0]), (s1[1], PPP[1])),
          mont((nR[0], D[1]), (nR[1], D[0]), (ns0, PPP[1]), (ns1, PPP[0])))
    ZZ3 = f2mul(f2mul(ZZ1, ZZ2, (4, 1)), PP, (4, 1))
    ZZZ3 = f2mul(f2mul(ZZZ1, ZZZ2, (4, 1)), PPP, (4, 1))
    if log is not None:
        log.update(u1=u1, nP=nP, nR=nR, PP=PP, PPP=PPP, Q=Q, X3=X3, Y3=Y3, ZZ3=ZZ3, ZZZ3=ZZZ3)
    return X3, Y3, ZZ3, ZZZ3


def check_adds(rnd):
    def pick(k):
        return E.normalized(rnd.choice([0, k * p, rnd.randrange(k * p + 1)]), k * p)

    def m2(a, b):
        return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def s2_(a, b):
        return ((a[0] - b[0]) % p, (a[1] - b[1]) % p)

    logs = []
    for group, inv, fn in ((1, G1_INV, g1_add), (2, G2_INV, g2_add)):
        for trial in range(100):
            if group == 1:
                A = tuple(pick(inv[n]) for n in ("X", "Y", "ZZ", "ZZZ"))
                Bq = tuple(pick(inv[n]) for n in ("X", "Y", "ZZ", "ZZZ"))
            else:
                A = tuple((pick(inv[n]), pick(inv[n])) for n in ("X", "Y", "ZZ", "ZZZ"))
                Bq = tuple((pick(inv[n]), pick(inv[n])) for n in ("X", "Y", "ZZ", "ZZZ"))
            log = {}
            out = fn(A, Bq, log)
            for nm, o in zip(("X", "Y", "ZZ", "ZZZ"), out):
                for c in (o if group == 2 else (o,)):
                    assert c.vb <= inv[nm] * p, (group, nm, c.vb / p)
                    assert all(m <= MASK for m in c.lb[:L - 1])
            if group == 1:
                x1, y1, z1, t1 = map(fval, A)
                x2, y2, z2, t2 = map(fval, Bq)
                U1, U2, S1, S2 = x1 * z2 % p, x2 * z1 % p, y1 * t2 % p, y2 * t1 % p
                P_, R_ = (U2 - U1) % p, (S2 - S1) % p
                pp, ppp = P_ * P_ % p, P_ ** 3 % p
                q = U1 * pp % p
                x3 = (R_ * R_ - ppp - 2 * q) % p
                exp = (x3, (R_ * (q - x3) - S1 * ppp) % p, z1 * z2 * pp % p, t1 * t2 * ppp % p)
                assert tuple(map(fval, out)) == exp
            else:
                x1, y1, z1, t1 = map(f2val, A)
                x2, y2, z2, t2 = map(f2val, Bq)
                U1, U2, S1, S2 = m2(x1, z2), m2(x2, z1), m2(y1, t2), m2(y2, t1)
                P_, R_ = s2_(U2, U1), s2_(S2, S1)
                pp = m2(P_, P_)
                ppp = m2(P_, pp)
                q = m2(U1, pp)
                rr = m2(R_, R_)
                x3 = ((rr[0] - ppp[0] - 2 * q[0]) % p, (rr[1] - ppp[1] - 2 * q[1]) % p)
                exp = (x3, s2_(m2(R_, s2_(q, x3)), m2(S1, ppp)), m2(m2(z1, z2), pp), m2(m2(t1, t2), ppp))
                assert tuple(map(f2val, out)) == exp
        logs.append(log)
    return logs


def main():
    rnd = random.Random(7)
    # plain products and squares at the documented input bound (13p x 13p)
    for _ in range(500):
        x, y = rnd.randrange(13 * p), rnd.randrange(13 * p)
        a, b = E.normalized(x, 13 * p), E.normalized(y, 13 * p)
        assert mont((a, b)).vb < 2.1 * p and sqr(a).vb < 2.1 * p
        assert fval(sqr(a)) == x * x * RINV * RINV % p
    lg = check_g1(rnd)
    print("G1 madd: invariant", G1_INV, "holds;", {k: f"{v.vb / p:.2f}p" for k, v in lg.items()})
    lg = check_g2(rnd)
    print("G2 madd: invariant", G2_INV, "holds;", {k: f"{max(c.vb for c in v) / p:.2f}p" for k, v in lg.items()})
    for nm, lg in zip(("G1 add", "G2 add"), check_adds(rnd)):
        print(nm + ": invariant holds;", {k: f"{max(c.vb for c in (v if isinstance(v, tuple) else (v,))) / p:.2f}p"
                                          for k, v in lg.items()})
    print("largest column bound: 2^%.3f" % (maxcol.bit_length() - 1 + (maxcol / (1 << (maxcol.bit_length() - 1)) - 1)))
    print("K forms used (mult, lift):", sorted(USED_K))


if __name__ == "__main__":
    main()
