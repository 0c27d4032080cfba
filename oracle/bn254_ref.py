"""
ORACLE (test infrastructure only) -- pure-Python big-int restatement of the
nim-groth16 hot path (BN254 MSM + NTT + the Groth16 prover algebra around it).

NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product path (nim_groth16_amd/) never does.

Parity status: the reference (Nim + constantine) cannot be built or run in this image and
ships no golden vectors; its only check is prove -> verifyProof == true on a toy circuit
(tests/groth16/testProver.nim:59-73).  This restatement is pinned by
  (1) every constant the reference embeds (fields.nim:36-50, io.nim:87-92, ntt.nim:95,
      domain.nim:26, curves.nim:75-77,112-124)  -> tests/test_oracle_constants.py
  (2) the reference's own fixture: toy R1CS + witness, fake setup -> prove -> verify,
      both flavours (tests/groth16/testProver.nim:17-73) -> tests/test_oracle_prover.py
  (3) canonicity: every output on the path is a canonical residue or a canonical affine
      point, so any correct implementation is byte-identical to constantine's.
Byte-level golden vectors from the reference itself: none exist ("parity unpinned" at the
byte level by the reference; pinned at the level the reference pins itself).

All values here are *standard-form* Python ints; Montgomery (R = 2^256, io.nim:60-92) is
applied only in the (de)serialisation helpers at the bottom, which define the byte layout
at the C-ABI boundary.
"""
from __future__ import annotations

# ----------------------------------------------------------------------------------------
# fields  (groth16/bn128/fields.nim:36-50)
# ----------------------------------------------------------------------------------------
P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47   # primeP
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001   # primeR
MONT = 1 << 256                                                          # io.nim:60-92 (R=2^256)

FP_MONT_R = MONT % P
FR_MONT_R = MONT % R
FP_INV_MONT_R = pow(MONT, -1, P)
FR_INV_MONT_R = pow(MONT, -1, R)

ONE_HALF_FR = 0x183227397098d014dc2822db40c0ac2e9419f4243cdcb848a1f0fac9f8000001  # ntt.nim:95
GEN28 = 0x2a3c09f0a58a7e8500e0a7eb8ef62abc402d111e41112ed49bd61b6e725b19f0        # domain.nim:26


def inv_fr(x: int) -> int:
    return pow(x, -1, R)


def inv_fp(x: int) -> int:
    return pow(x, -1, P)


def small_pow_fr(base: int, expo: int) -> int:
    """fields.nim:139-153 (negative exponents invert the base)."""
    if expo < 0:
        return pow(inv_fr(base), -expo, R)
    return pow(base, expo, R)


def batch_inverse_fr(xs):
    """fields.nim:163-174 (Montgomery's trick)."""
    n = len(xs)
    assert n > 0
    us = [1] * (n + 1)
    a = 1
    for i in range(n):
        a = a * xs[i] % R
        us[i + 1] = a
    vs = [0] * n
    vs[n - 1] = inv_fr(us[n])
    for i in range(n - 2, -1, -1):
        vs[i] = vs[i + 1] * xs[i + 1] % R
    return [us[i] * vs[i] % R for i in range(n)]


# Fp2 = Fp[u]/(u^2+1), element = (c0, c1)   (fields.nim:27-32, coords:[i,u])
def fp2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def fp2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def fp2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def fp2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def fp2_sqr(a):
    return ((a[0] + a[1]) * (a[0] - a[1]) % P, 2 * a[0] * a[1] % P)


def fp2_scal(a, k):
    return (a[0] * k % P, a[1] * k % P)


def fp2_inv(a):
    d = inv_fp((a[0] * a[0] + a[1] * a[1]) % P)
    return (a[0] * d % P, (-a[1]) * d % P)


def fp2_is_zero(a):
    return a[0] == 0 and a[1] == 0


FP2_ZERO = (0, 0)
FP2_ONE = (1, 0)


class _Fp:
    """Field-ops bundle for the base field (G1 coordinates)."""
    zero = 0
    one = 1
    add = staticmethod(lambda a, b: (a + b) % P)
    sub = staticmethod(lambda a, b: (a - b) % P)
    neg = staticmethod(lambda a: (-a) % P)
    mul = staticmethod(lambda a, b: a * b % P)
    sqr = staticmethod(lambda a: a * a % P)
    inv = staticmethod(inv_fp)
    is_zero = staticmethod(lambda a: a == 0)
    small = staticmethod(lambda a, k: a * k % P)


class _Fp2:
    """Field-ops bundle for Fp2 (G2 coordinates)."""
    zero = FP2_ZERO
    one = FP2_ONE
    add = staticmethod(fp2_add)
    sub = staticmethod(fp2_sub)
    neg = staticmethod(fp2_neg)
    mul = staticmethod(fp2_mul)
    sqr = staticmethod(fp2_sqr)
    inv = staticmethod(fp2_inv)
    is_zero = staticmethod(fp2_is_zero)
    small = staticmethod(fp2_scal)


# ----------------------------------------------------------------------------------------
# curves  (groth16/bn128/curves.nim)
#   G1: y^2 = x^3 + 3 over Fp ; G2: y^2 = x^3 + 3/(9+u) over Fp2 ; infinity := (0,0)
# ----------------------------------------------------------------------------------------
TWIST_B = (0x2b149d40ceb8aaae81be18991be06ac3b5b4c5e559dbefa33267e6dc24a138e5,
           0x009713b03af0fed4cd2cafadeed8fdf4a74fa084e52d1852e4a2bd0685c315d2)    # curves.nim:75-77
GEN1 = (1, 2)                                                                      # curves.nim:112-113
GEN2 = ((0x1adcd0ed10df9cb87040f46655e3808f98aa68a570acf5b0bde23fab1f149701,
         0x09e847e9f05a6082c3cd2a1d0a3a82e6fbfbe620f7f31269fa15d21c1c13b23b),
        (0x056c01168a5319461f7ca7aa19d4fcfd1c7cdf52dbfc4cbee6f915250b7f6fc8,
         0x0efe500a2d02dd77f5f401329f30895df553b878fc3c0dadaaa86456a623235c))      # curves.nim:115-121
INF_G1 = (0, 0)                                                                    # curves.nim:49
INF_G2 = (FP2_ZERO, FP2_ZERO)                                                      # curves.nim:50


class Curve:
    """Short-Weierstrass a=0 curve over field bundle F, affine infinity = (0,0)."""

    def __init__(self, F, b, name):
        self.F, self.b, self.name = F, b, name
        self.inf = (F.zero, F.zero)

    def is_inf(self, p):
        return self.F.is_zero(p[0]) and self.F.is_zero(p[1])

    def is_on_curve(self, p):
        """curves.nim:54-67 / :79-91 -- infinity is on the curve by definition."""
        F = self.F
        if self.is_inf(p):
            return True
        x, y = p
        return F.is_zero(F.sub(F.add(F.mul(F.sqr(x), x), self.b), F.sqr(y)))

    def neg(self, p):
        if self.is_inf(p):
            return p
        return (p[0], self.F.neg(p[1]))

    # Jacobian (X,Y,Z), infinity Z=0 -- the oracle's own choice of coordinates; results are
    # compared only after conversion to the canonical affine form.
    def to_jac(self, p):
        F = self.F
        if self.is_inf(p):
            return (F.one, F.one, F.zero)
        return (p[0], p[1], F.one)

    def to_affine(self, j):
        F = self.F
        X, Y, Z = j
        if F.is_zero(Z):
            return self.inf
        zi = F.inv(Z)
        zi2 = F.sqr(zi)
        return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))

    def jdbl(self, j):
        F = self.F
        X, Y, Z = j
        if F.is_zero(Z) or F.is_zero(Y):
            return (F.one, F.one, F.zero)
        A = F.sqr(X)
        B = F.sqr(Y)
        C = F.sqr(B)
        D = F.small(F.sub(F.sub(F.sqr(F.add(X, B)), A), C), 2)
        E = F.small(A, 3)
        Fv = F.sqr(E)
        X3 = F.sub(Fv, F.small(D, 2))
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), F.small(C, 8))
        Z3 = F.small(F.mul(Y, Z), 2)
        return (X3, Y3, Z3)

    def jadd(self, p, q):
        F = self.F
        X1, Y1, Z1 = p
        X2, Y2, Z2 = q
        if F.is_zero(Z1):
            return q
        if F.is_zero(Z2):
            return p
        Z1Z1 = F.sqr(Z1)
        Z2Z2 = F.sqr(Z2)
        U1 = F.mul(X1, Z2Z2)
        U2 = F.mul(X2, Z1Z1)
        S1 = F.mul(Y1, F.mul(Z2, Z2Z2))
        S2 = F.mul(Y2, F.mul(Z1, Z1Z1))
        H = F.sub(U2, U1)
        Rr = F.sub(S2, S1)
        if F.is_zero(H):
            if F.is_zero(Rr):
                return self.jdbl(p)
            return (F.one, F.one, F.zero)
        HH = F.sqr(H)
        HHH = F.mul(H, HH)
        V = F.mul(U1, HH)
        X3 = F.sub(F.sub(F.sqr(Rr), HHH), F.small(V, 2))
        Y3 = F.sub(F.mul(Rr, F.sub(V, X3)), F.mul(S1, HHH))
        Z3 = F.mul(F.mul(Z1, Z2), H)
        return (X3, Y3, Z3)

    def add(self, p, q):
        """curves.nim:136-154 addG1/addG2 (affine in, affine out)."""
        return self.to_affine(self.jadd(self.to_jac(p), self.to_jac(q)))

    def jmul(self, j, k):
        F = self.F
        acc = (F.one, F.one, F.zero)
        if k == 0:
            return acc
        for bit in bin(k)[2:]:
            acc = self.jdbl(acc)
            if bit == '1':
                acc = self.jadd(acc, j)
        return acc

    def mul(self, k, p):
        """curves.nim:182-214 `**` : scalar (standard-form Fr value or bigint) times affine point."""
        return self.to_affine(self.jmul(self.to_jac(p), k))

    def msm_naive(self, coeffs, points):
        """msm.nim:162-198 msmNaiveG1/G2 -- the in-tree *definition* of the MSM."""
        assert len(coeffs) == len(points), "incompatible sequence lengths"
        s = (self.F.one, self.F.one, self.F.zero)
        for c, pt in zip(coeffs, points):
            s = self.jadd(s, self.jmul(self.to_jac(pt), c % R))
        return self.to_affine(s)

    def msm_pippenger(self, coeffs, points, c=8):
        """Independent second algorithm (unsigned c-bit windows) used to cross-check msm_naive
        and to reach a few thousand points in pure Python."""
        assert len(coeffs) == len(points), "incompatible sequence lengths"
        F = self.F
        inf = (F.one, F.one, F.zero)
        nwin = (254 + c - 1) // c
        jp = [self.to_jac(p) for p in points]
        total = inf
        for w in range(nwin - 1, -1, -1):
            for _ in range(c):
                total = self.jdbl(total)
            buckets = [inf] * (1 << c)
            for s, p in zip(coeffs, jp):
                d = ((s % R) >> (w * c)) & ((1 << c) - 1)
                if d:
                    buckets[d] = self.jadd(buckets[d], p)
            run = inf
            acc = inf
            for d in range((1 << c) - 1, 0, -1):
                run = self.jadd(run, buckets[d])
                acc = self.jadd(acc, run)
            total = self.jadd(total, acc)
        return self.to_affine(total)


G1 = Curve(_Fp, 3, "G1")
G2 = Curve(_Fp2, TWIST_B, "G2")


def msm_multithreaded(curve, nthreads_hint, coeffs, points, ncpu=8):
    """msm.nim:89-158 msmMultiThreadedG1/G2: same chunking rule, partials summed from infinity.
    (Sequential here -- the chunking cannot change the canonical result; restated so that the
    product's host mirror can be compared structurally.)"""
    N = len(coeffs)
    assert N == len(points), "incompatible sequence lengths"
    target = ncpu if nthreads_hint <= 0 else min(nthreads_hint, 256)
    nthreads = max(1, min(N // 128, target))
    ntasks = nthreads
    res = curve.inf
    a = 0
    for k in range(ntasks):
        b = (N * (k + 1)) // ntasks if k < ntasks - 1 else N
        res = curve.add(res, curve.msm_pippenger(coeffs[a:b], points[a:b]) if b - a > 64
                        else curve.msm_naive(coeffs[a:b], points[a:b]))
        a = b
    return res


# ----------------------------------------------------------------------------------------
# domain + NTT  (groth16/math/domain.nim, groth16/math/ntt.nim)
# ----------------------------------------------------------------------------------------
def ceiling_log2(x: int) -> int:
    """misc.nim:43-47"""
    if x == 0:
        return -1
    return (x - 1).bit_length()


class Domain:
    """domain.nim:15-46"""

    def __init__(self, size: int):
        log2 = ceiling_log2(size)
        assert (1 << log2) == size, "domain must have a power-of-two size"
        gen = small_pow_fr(GEN28, 1 << (28 - log2))
        assert pow(gen, size, R) == 1, "domain generator sanity check /A"
        assert size == 1 or pow(gen, size // 2, R) != 1, "domain generator sanity check /B"
        self.domainSize = size
        self.logDomainSize = log2
        self.domainGen = gen
        self.invDomainGen = inv_fr(gen)
        self.invDomainSize = inv_fr(size % R)


def enumerate_domain(D: Domain):
    """domain.nim:50-56"""
    xs, g = [], 1
    for _ in range(D.domainSize):
        xs.append(g)
        g = g * D.domainGen % R
    return xs


def _forward_worker(m, src_stride, gpows, src, src_ofs, buf, buf_ofs, tgt, tgt_ofs):
    """ntt.nim:17-50 -- transliterated recursion (DIT, natural order in/out)."""
    if m == 0:
        tgt[tgt_ofs] = src[src_ofs]
    elif m == 1:
        tgt[tgt_ofs] = (src[src_ofs] + src[src_ofs + src_stride]) % R
        tgt[tgt_ofs + 1] = (src[src_ofs] - src[src_ofs + src_stride]) % R
    else:
        N = 1 << m
        half = 1 << (m - 1)
        _forward_worker(m - 1, src_stride << 1, gpows, src, src_ofs, buf, buf_ofs + N, buf, buf_ofs)
        _forward_worker(m - 1, src_stride << 1, gpows, src, src_ofs + src_stride, buf, buf_ofs + N, buf, buf_ofs + half)
        for j in range(half):
            y = gpows[j * src_stride] * buf[buf_ofs + j + half] % R
            tgt[tgt_ofs + j] = (buf[buf_ofs + j] + y) % R
            tgt[tgt_ofs + j + half] = (buf[buf_ofs + j] - y) % R


def forward_ntt(src, D: Domain):
    """ntt.nim:55-77  y_k = sum_i x_i g^(ik), unscaled."""
    assert D.domainSize == (1 << D.logDomainSize), "domain must have a power-of-two size"
    assert D.domainSize == len(src), "input must have the same size as the domain"
    N = D.domainSize
    buf = [0] * (2 * N)
    tgt = [0] * N
    gpows, x = [], 1
    for _ in range(N // 2):
        gpows.append(x)
        x = x * D.domainGen % R
    _forward_worker(D.logDomainSize, 1, gpows, list(src), 0, buf, 0, tgt, 0)
    return tgt


def extend_and_forward_ntt(src, D: Domain):
    """ntt.nim:81-91"""
    n, N = len(src), D.domainSize
    assert n <= N
    return forward_ntt(list(src) + [0] * (N - n), D)


def _div2(x):
    return x * ONE_HALF_FR % R


def _inverse_worker(m, tgt_stride, gpows, src, src_ofs, buf, buf_ofs, tgt, tgt_ofs):
    """ntt.nim:97-134 -- transliterated recursion (DIF with per-level halving)."""
    if m == 0:
        tgt[tgt_ofs] = src[src_ofs]
    elif m == 1:
        tgt[tgt_ofs] = _div2((src[src_ofs] + src[src_ofs + 1]) % R)
        tgt[tgt_ofs + tgt_stride] = _div2((src[src_ofs] - src[src_ofs + 1]) % R)
    else:
        N = 1 << m
        half = 1 << (m - 1)
        for j in range(half):
            a, b = src[src_ofs + j], src[src_ofs + j + half]
            buf[buf_ofs + j] = _div2((a + b) % R)
            buf[buf_ofs + j + half] = (a - b) * gpows[j * tgt_stride] % R
        _inverse_worker(m - 1, tgt_stride << 1, gpows, buf, buf_ofs, buf, buf_ofs + N, tgt, tgt_ofs)
        _inverse_worker(m - 1, tgt_stride << 1, gpows, buf, buf_ofs + half, buf, buf_ofs + N, tgt, tgt_ofs + tgt_stride)


def inverse_ntt(src, D: Domain):
    """ntt.nim:139-161  exact inverse of forward_ntt (1/N folded into the butterflies)."""
    assert D.domainSize == (1 << D.logDomainSize), "domain must have a power-of-two size"
    assert D.domainSize == len(src), "input must have the same size as the domain"
    N = D.domainSize
    buf = [0] * (2 * N)
    tgt = [0] * N
    gpows, x = [], ONE_HALF_FR
    ginv = inv_fr(D.domainGen)
    for _ in range(N // 2):
        gpows.append(x)
        x = x * ginv % R
    _inverse_worker(D.logDomainSize, 1, gpows, list(src), 0, buf, 0, tgt, 0)
    return tgt


def naive_dft(src, D: Domain):
    """Definition used to pin forward_ntt's convention (natural order, w = domainGen)."""
    n = D.domainSize
    return [sum(src[i] * pow(D.domainGen, i * k, R) for i in range(n)) % R for k in range(n)]


# ----------------------------------------------------------------------------------------
# polynomial helpers needed by the fake setup  (groth16/math/poly.nim)
# ----------------------------------------------------------------------------------------
def eval_lagrange_poly_at(D: Domain, k: int, zeta: int) -> int:
    """poly.nim:242-250"""
    omega_k = small_pow_fr(D.domainGen, k)
    denom = (zeta - omega_k) % R
    if denom == 0:
        raise AssertionError("point should be outside the domain")
    return omega_k * ((pow(zeta, D.domainSize, R) - 1) % R) % R * D.invDomainSize % R * inv_fr(denom) % R


# ----------------------------------------------------------------------------------------
# quotient  (groth16/prover.nim:96-181)
# ----------------------------------------------------------------------------------------
def multiply_by_powers(xs, eta):
    """prover.nim:96-106 (the n=1 out-of-range quirk is not reproduced; n>=2 on the path)."""
    n = len(xs)
    assert n >= 1
    ys, s = [], 1
    for i in range(n):
        ys.append(s * xs[i] % R)
        s = s * eta % R
    return ys


def shift_eval_domain(values, D: Domain, eta):
    """prover.nim:109-113"""
    cs = inverse_ntt(values, D)
    ds = multiply_by_powers(cs, eta)
    return forward_ntt(ds, D)


def compute_snarkjs_scalar_coeffs(Az, Bz, Cz):
    """prover.nim:158-181"""
    n = len(Az)
    assert len(Bz) == n and len(Cz) == n
    D = Domain(n)
    eta = Domain(2 * n).domainGen
    A1 = shift_eval_domain(Az, D, eta)
    B1 = shift_eval_domain(Bz, D, eta)
    C1 = shift_eval_domain(Cz, D, eta)
    return [(A1[j] * B1[j] - C1[j]) % R for j in range(n)]


def compute_quotient_pointwise(Az, Bz, Cz):
    """prover.nim:118-148 (JensGroth flavour): coefficients of Q = (A*B-C)/Z."""
    n = len(Az)
    assert len(Bz) == n and len(Cz) == n
    D = Domain(n)
    eta = Domain(2 * n).domainGen
    invZ1 = inv_fr((pow(eta, n, R) - 1) % R)
    A1 = shift_eval_domain(Az, D, eta)
    B1 = shift_eval_domain(Bz, D, eta)
    C1 = shift_eval_domain(Cz, D, eta)
    ys = [(A1[j] * B1[j] - C1[j]) % R * invZ1 % R for j in range(n)]
    Q1 = inverse_ntt(ys, D)
    return multiply_by_powers(Q1, inv_fr(eta))


# ----------------------------------------------------------------------------------------
# data model  (groth16/zkey_types.nim, files/r1cs.nim, files/witness.nim)
# ----------------------------------------------------------------------------------------
JENS_GROTH, SNARKJS = "JensGroth", "Snarkjs"     # zkey_types.nim:10-12
MATRIX_A, MATRIX_B, MATRIX_C = 0, 1, 2           # zkey_types.nim:43-46


class R1CS:
    """files/r1cs.nim:62-80; a constraint is (A,B,C), each a list of (wireIdx, value)."""

    def __init__(self, nWires, nPubOut, nPubIn, nPrivIn, constraints):
        self.nWires, self.nPubOut, self.nPubIn, self.nPrivIn = nWires, nPubOut, nPubIn, nPrivIn
        self.constraints = constraints


class ZKey:
    """zkey_types.nim:14-59 (header + spec points + verifier/prover points + coeffs)."""

    def __init__(self):
        self.flavour = SNARKJS
        self.nvars = self.npubs = self.domainSize = self.logDomainSize = 0
        self.alpha1 = self.beta1 = self.delta1 = INF_G1
        self.beta2 = self.gamma2 = self.delta2 = INF_G2
        self.pointsIC = []
        self.pointsA1, self.pointsB1, self.pointsB2, self.pointsC1, self.pointsH1 = [], [], [], [], []
        self.coeffs = []          # list of (matrix, row, col, value)


def r1cs_to_coeffs(r1cs: R1CS):
    """fake_setup.nim:46-65 (incl. the snarkjs dummy rows for the public IO)."""
    coeffs = []
    n = len(r1cs.constraints)
    p = r1cs.nPubIn + r1cs.nPubOut
    for i, (A, B, _C) in enumerate(r1cs.constraints):
        for (w, v) in A:
            coeffs.append((MATRIX_A, i, w, v % R))
        for (w, v) in B:
            coeffs.append((MATRIX_B, i, w, v % R))
    for i in range(n, n + p + 1):
        coeffs.append((MATRIX_A, i, i - n, 1))
    return coeffs


def build_abc(coeffs, domain_size, witness):
    """prover.nim:56-73"""
    Az = [0] * domain_size
    Bz = [0] * domain_size
    for (m, row, col, v) in coeffs:
        if m == MATRIX_A:
            Az[row] = (Az[row] + v * witness[col]) % R
        elif m == MATRIX_B:
            Bz[row] = (Bz[row] + v * witness[col]) % R
        else:
            raise AssertionError("fatal error")
    Cz = [Az[i] * Bz[i] % R for i in range(domain_size)]
    return Az, Bz, Cz


class ToxicWaste:
    """fake_setup.nim:23-29"""

    def __init__(self, alpha, beta, gamma, delta, tau):
        self.alpha, self.beta, self.gamma, self.delta, self.tau = alpha, beta, gamma, delta, tau


def fake_circuit_setup(r1cs: R1CS, toxic: ToxicWaste, flavour=SNARKJS, batch_g1=None, batch_g2=None) -> ZKey:
    """fake_setup.nim:201-326.  batch_g1/batch_g2(list of scalars) -> list of affine points let the C
    oracle's fixed-base multiplier be plugged in for mid-size circuits (same canonical points)."""
    batch_g1 = batch_g1 or (lambda ks: [G1.mul(k % R, GEN1) for k in ks])
    batch_g2 = batch_g2 or (lambda ks: [G2.mul(k % R, GEN2) for k in ks])
    mul_g1 = lambda k: batch_g1([k])[0]     # noqa: E731
    mul_g2 = lambda k: batch_g2([k])[0]     # noqa: E731
    neqs = len(r1cs.constraints)
    npub = r1cs.nPubIn + r1cs.nPubOut
    logDom = ceiling_log2(neqs + npub + 1)
    dom = 1 << logDom
    nvars = r1cs.nWires
    zk = ZKey()
    zk.flavour, zk.nvars, zk.npubs, zk.domainSize, zk.logDomainSize = flavour, nvars, npub, dom, logDom
    zk.alpha1, zk.beta1, zk.delta1 = mul_g1(toxic.alpha), mul_g1(toxic.beta), mul_g1(toxic.delta)
    zk.beta2, zk.gamma2, zk.delta2 = mul_g2(toxic.beta), mul_g2(toxic.gamma), mul_g2(toxic.delta)

    # sparse columns (fake_setup.nim:159-187)
    colA = [dict() for _ in range(nvars)]
    colB = [dict() for _ in range(nvars)]
    colC = [dict() for _ in range(nvars)]

    def ins(col, i, y):
        col[i] = (col.get(i, 0) + y) % R
    for i, (A, B, C) in enumerate(r1cs.constraints):
        for (w, v) in A:
            ins(colA[w], i, v)
        for (w, v) in B:
            ins(colB[w], i, v)
        for (w, v) in C:
            ins(colC[w], i, v)
    for i in range(neqs, neqs + npub + 1):
        ins(colA[i - neqs], i, 1)

    D = Domain(dom)
    # L_k(tau) for all k at once (same values as evalLagrangePolyAt per k, fake_setup.nim:251)
    tau = toxic.tau % R
    ztau = (pow(tau, dom, R) - 1) % R
    omegas = enumerate_domain(D)
    dinv = batch_inverse_fr([(tau - w) % R for w in omegas])
    lag = [omegas[k] * ztau % R * D.invDomainSize % R * dinv[k] % R for k in range(dom)]

    def dot(col):
        return sum(x * lag[i] for i, x in col.items()) % R
    tausA = [dot(c) for c in colA]
    tausB = [dot(c) for c in colB]
    tausC = [dot(c) for c in colC]
    zk.pointsA1 = batch_g1(tausA)
    zk.pointsB1 = batch_g1(tausB)
    zk.pointsB2 = batch_g2(tausB)
    gamma_inv, delta_inv = inv_fr(toxic.gamma), inv_fr(toxic.delta)
    # fake_setup.nim:273-277: inv * (beta*A_j + alpha*B1_j + C_j)  == ((beta*a + alpha*b + c) * inv) * G
    comb = [(toxic.beta * tausA[j] + toxic.alpha * tausB[j] + tausC[j]) % R for j in range(nvars)]
    zk.pointsIC = batch_g1([gamma_inv * comb[j] % R for j in range(npub + 1)])
    zk.pointsC1 = batch_g1([delta_inv * comb[j] % R for j in range(npub + 1, nvars)])
    if flavour == JENS_GROTH:
        # fake_setup.nim:290-292  [delta^-1 tau^i Z(tau)]
        zk.pointsH1 = batch_g1([delta_inv * pow(tau, i, R) % R * ztau % R for i in range(dom)])
    else:
        # fake_setup.nim:299-302  [delta^-1 L_{2i+1}(tau)] on the doubled domain
        D2 = Domain(2 * dom)
        zk.pointsH1 = batch_g1([delta_inv * eval_lagrange_poly_at(D2, 2 * i + 1, tau) % R for i in range(dom)])
    zk.coeffs = r1cs_to_coeffs(r1cs)
    return zk


# ----------------------------------------------------------------------------------------
# prover  (groth16/prover.nim:215-304)
# ----------------------------------------------------------------------------------------
class Proof:
    def __init__(self, publicIO, pi_a, pi_b, pi_c):
        self.publicIO, self.pi_a, self.pi_b, self.pi_c, self.curve = publicIO, pi_a, pi_b, pi_c, "bn128"


def generate_proof_with_mask(zk: ZKey, witness, r: int, s: int, msm_g1=None, msm_g2=None, quotient=None) -> Proof:
    """prover.nim:215-304.  msm_g1/msm_g2/quotient hooks let tests swap in another MSM/NTT
    implementation for those calls while keeping this orchestration as the oracle."""
    msm_g1 = msm_g1 or G1.msm_naive
    msm_g2 = msm_g2 or G2.msm_naive
    nvars, npubs = zk.nvars, zk.npubs
    assert nvars == len(witness), "wrong witness length"
    pubIO = [witness[i] for i in range(npubs + 1)]
    Az, Bz, Cz = build_abc(zk.coeffs, zk.domainSize, witness)
    if quotient is not None:
        qs = quotient(Az, Bz, Cz, zk.flavour)
    elif zk.flavour == JENS_GROTH:
        qs = compute_quotient_pointwise(Az, Bz, Cz)
    else:
        qs = compute_snarkjs_scalar_coeffs(Az, Bz, Cz)
    zs = [witness[j] for j in range(npubs + 1, nvars)]
    assert len(witness) == len(zk.pointsA1) == len(zk.pointsB1) == len(zk.pointsB2)
    assert zk.domainSize == len(qs) == len(zk.pointsH1)
    assert nvars - npubs - 1 == len(zs) == len(zk.pointsC1)

    pi_a = zk.alpha1
    pi_a = G1.add(pi_a, G1.mul(r, zk.delta1))
    pi_a = G1.add(pi_a, msm_g1(witness, zk.pointsA1))

    rho = zk.beta1
    rho = G1.add(rho, G1.mul(s, zk.delta1))
    rho = G1.add(rho, msm_g1(witness, zk.pointsB1))

    pi_b = zk.beta2
    pi_b = G2.add(pi_b, G2.mul(s, zk.delta2))
    pi_b = G2.add(pi_b, msm_g2(witness, zk.pointsB2))

    pi_c = G1.mul(s, pi_a)
    pi_c = G1.add(pi_c, G1.mul(r, rho))
    pi_c = G1.add(pi_c, G1.mul((-(r * s)) % R, zk.delta1))
    pi_c = G1.add(pi_c, msm_g1(qs, zk.pointsH1))
    pi_c = G1.add(pi_c, msm_g1(zs, zk.pointsC1))
    return Proof(pubIO, pi_a, pi_b, pi_c)


# ----------------------------------------------------------------------------------------
# verifier  (groth16/verifier.nim:31-52) with a self-contained ate pairing
#   Fp12 = Fp[w]/(w^12 - 18 w^6 + 82)  (u = w^6 - 9, so that (9+u) = w^6; export_sage.nim:84-97)
# ----------------------------------------------------------------------------------------
def _f12_mul(a, b):
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                t[i + j] += ai * bj
    for k in range(22, 11, -1):          # w^12 = 18 w^6 - 82
        c = t[k]
        if c:
            t[k - 6] += 18 * c
            t[k - 12] -= 82 * c
    return [x % P for x in t[:12]]


def _f12_one():
    return [1] + [0] * 11


def _f12_pow(a, e):
    r = _f12_one()
    for bit in bin(e)[2:]:
        r = _f12_mul(r, r)
        if bit == '1':
            r = _f12_mul(r, a)
    return r


def _emb(a2, k):
    """Fp2 element a+bu placed at w^k:  (a-9b) w^k + b w^(k+6)."""
    out = [0] * 12
    out[k] = (a2[0] - 9 * a2[1]) % P
    out[k + 6] = a2[1] % P
    return out


def _line(T, lam, Pt):
    """Line through twist point T with twist-slope lam, evaluated at P in G1 (untwist x=x'w^2, y=y'w^3):
       l = yP - (lam*xP) w + (lam*xT - yT) w^3."""
    xP, yP = Pt
    l = [0] * 12
    l[0] = yP % P
    a = _emb(fp2_scal(lam, (-xP) % P), 1)
    b = _emb(fp2_sub(fp2_mul(lam, T[0]), T[1]), 3)
    return [(l[i] + a[i] + b[i]) % P for i in range(12)]


ATE_LOOP = P - R        # t-1, with #E(Fp) = p+1-t = r


def miller_loop(Pt, Q):
    """f_{t-1,Q}(P) for the ate pairing; P in G1 (affine), Q in G2 (affine, on the twist)."""
    if G1.is_inf(Pt) or G2.is_inf(Q):
        return _f12_one()
    f = _f12_one()
    T = Q
    for bit in bin(ATE_LOOP)[3:]:
        lam = fp2_mul(fp2_scal(fp2_sqr(T[0]), 3), fp2_inv(fp2_scal(T[1], 2)))
        f = _f12_mul(_f12_mul(f, f), _line(T, lam, Pt))
        x3 = fp2_sub(fp2_sqr(lam), fp2_scal(T[0], 2))
        T = (x3, fp2_sub(fp2_mul(lam, fp2_sub(T[0], x3)), T[1]))
        if bit == '1':
            if T[0] == Q[0]:
                # T == -Q can only occur at the very end for points of order r; vertical line lies in a
                # proper subfield and is killed by the final exponentiation.
                assert fp2_is_zero(fp2_add(T[1], Q[1]))
                T = INF_G2
                continue
            lam = fp2_mul(fp2_sub(Q[1], T[1]), fp2_inv(fp2_sub(Q[0], T[0])))
            f = _f12_mul(f, _line(T, lam, Pt))
            x3 = fp2_sub(fp2_sub(fp2_sqr(lam), T[0]), Q[0])
            T = (x3, fp2_sub(fp2_mul(lam, fp2_sub(T[0], x3)), T[1]))
    return f


FINAL_EXP = (P ** 12 - 1) // R


def final_exp(f):
    return _f12_pow(f, FINAL_EXP)


def pairing(Pt, Q):
    """curves.nim:218-221 (any non-degenerate bilinear pairing yields the same verify() verdict)."""
    return final_exp(miller_loop(Pt, Q))


def verify_proof(zk: ZKey, proof: Proof) -> bool:
    """verifier.nim:31-52:  e(-A,B) * e(alpha,beta) * e(C,delta) * e(sum pub_i IC_i, gamma) == 1"""
    assert proof.curve == "bn128"
    assert G1.is_on_curve(proof.pi_a), "pi_a is not in G1"
    assert G2.is_on_curve(proof.pi_b), "pi_b is not in G2"
    assert G1.is_on_curve(proof.pi_c), "pi_c is not in G1"
    pub = G1.msm_naive(proof.publicIO, zk.pointsIC)
    f = miller_loop(G1.neg(proof.pi_a), proof.pi_b)
    f = _f12_mul(f, miller_loop(zk.alpha1, zk.beta2))
    f = _f12_mul(f, miller_loop(proof.pi_c, zk.delta2))
    f = _f12_mul(f, miller_loop(pub, zk.gamma2))
    return final_exp(f) == _f12_one()


# ----------------------------------------------------------------------------------------
# byte layouts at the C-ABI boundary (constantine in-memory = 4 x u64 LE limbs, Montgomery
# R=2^256; io.nim:60-153)
# ----------------------------------------------------------------------------------------
def fr_to_mont_bytes(x: int) -> bytes:
    return (x % R * FR_MONT_R % R).to_bytes(32, "little")


def fr_from_mont_bytes(b: bytes) -> int:
    return int.from_bytes(b, "little") * FR_INV_MONT_R % R


def fr_to_std_bytes(x: int) -> bytes:
    """.wtns layout: canonical little-endian (witness.nim:14,57-60)."""
    return (x % R).to_bytes(32, "little")


def fp_to_mont_bytes(x: int) -> bytes:
    return (x % P * FP_MONT_R % P).to_bytes(32, "little")


def fp_from_mont_bytes(b: bytes) -> int:
    return int.from_bytes(b, "little") * FP_INV_MONT_R % P


def g1_to_bytes(p) -> bytes:
    return fp_to_mont_bytes(p[0]) + fp_to_mont_bytes(p[1])


def g1_from_bytes(b: bytes):
    return (fp_from_mont_bytes(b[0:32]), fp_from_mont_bytes(b[32:64]))


def g2_to_bytes(p) -> bytes:
    return (fp_to_mont_bytes(p[0][0]) + fp_to_mont_bytes(p[0][1]) +
            fp_to_mont_bytes(p[1][0]) + fp_to_mont_bytes(p[1][1]))


def g2_from_bytes(b: bytes):
    return ((fp_from_mont_bytes(b[0:32]), fp_from_mont_bytes(b[32:64])),
            (fp_from_mont_bytes(b[64:96]), fp_from_mont_bytes(b[96:128])))


# ----------------------------------------------------------------------------------------
# seeded inputs shared by oracle, tests and bench (SplitMix64)
# ----------------------------------------------------------------------------------------
class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def fr(self) -> int:
        """4 x u64 little-endian limbs -> 256-bit integer reduced mod r."""
        v = 0
        for i in range(4):
            v |= self.next() << (64 * i)
        return v % R


# the reference's own fixture (tests/groth16/testProver.nim:17-55)
def toy_r1cs() -> R1CS:
    m1 = R - 1
    eq1 = ([], [], [(1, m1), (2, 1), (7, 1)])
    eq2 = ([(3, 1)], [(4, 1)], [(6, 1)])
    eq3 = ([(5, 1)], [(6, 1)], [(7, 1)])
    return R1CS(nWires=8, nPubOut=1, nPubIn=1, nPrivIn=3, constraints=[eq1, eq2, eq3])


TOY_WITNESS = [1, 2023, 1022, 7, 11, 13, 7 * 11, 7 * 11 * 13]
