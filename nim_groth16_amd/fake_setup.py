"""Host mirror of groth16/fake_setup.nim: a "fake" circuit-specific trusted setup from explicit toxic
waste.  The O(n) scalar bookkeeping (Lagrange values at tau, the combinations of fake_setup.nim:263-277) is host
code as in the reference; every `y ** gen1` / `y ** gen2` (fake_setup.nim:258-261, 273-277, 290-302) -- the
expensive part: ~4N+n G1 and N G2 scalar multiplications -- runs on the GPU (g16_fixed_base_g1/g2), and so do the
sparse column dot products (fake_setup.nim:159-187, 254-256: ~10^7 multiplications for a Poseidon-shaped circuit of
2^20 constraints) through the transposed matrices (g16_spmv_fr)."""
from __future__ import annotations

from dataclasses import dataclass

from . import bn128 as F
from ._lib import default_context
from .zkey_types import (GrothHeader, JensGroth, MatrixA, MatrixB, ProverPoints, Snarkjs, SpecPoints, ZKey)

R = F.primeR


@dataclass
class ToxicWaste:                    # fake_setup.nim:23-29
    alpha: int
    beta: int
    gamma: int
    delta: int
    tau: int


@dataclass
class R1CS:                          # files/r1cs.nim:62-80; constraint = (A, B, C), each [(wireIdx, value)]
    nWires: int
    nPubOut: int
    nPubIn: int
    nPrivIn: int
    constraints: list
    wireToLabel: list = None


def r1csToCoeffs(r1cs: R1CS):
    """fake_setup.nim:46-65 (incl. snarkjs' dummy A rows for the public IO)."""
    coeffs = []
    n = len(r1cs.constraints)
    p = r1cs.nPubIn + r1cs.nPubOut
    for i, (A, B, _C) in enumerate(r1cs.constraints):
        for (w, v) in A:
            coeffs.append((MatrixA, i, w, F.frToMontBytes(v)))
        for (w, v) in B:
            coeffs.append((MatrixB, i, w, F.frToMontBytes(v)))
    one = F.frToMontBytes(1)
    for i in range(n, n + p + 1):
        coeffs.append((MatrixA, i, i - n, one))
    return coeffs


def r1csToCoeffArray(r1cs):
    """r1csToCoeffs for a synthetic.SparseR1CS, vectorised: -> zkey_types.CoeffArray (the g16_coeff records of
    include/g16hip.h: all A entries, all B entries, then the dummy A rows; buildABC sums per row, so the order of
    ZKey.coeffs is immaterial, prover.nim:56-73)."""
    import numpy as np
    from .zkey_types import COEFF_DTYPE, CoeffArray
    n, p = r1cs.nConstraints, r1cs.nPubIn + r1cs.nPubOut
    table = np.frombuffer(F.frSeqToMontBytes(r1cs.values + [1]), dtype=np.uint8).reshape(-1, 32)
    na, nb = len(r1cs.A[0]), len(r1cs.B[0])
    arr = np.zeros(na + nb + p + 1, dtype=COEFF_DTYPE)
    for lo, (rows, wires, vi), m in ((0, r1cs.A, MatrixA), (na, r1cs.B, MatrixB)):
        hi = lo + len(rows)
        arr["m"][lo:hi], arr["r"][lo:hi], arr["c"][lo:hi], arr["v"][lo:hi] = m, rows, wires, table[vi]
    arr["m"][na + nb:] = MatrixA
    arr["r"][na + nb:] = np.arange(n, n + p + 1)
    arr["c"][na + nb:] = np.arange(p + 1)
    arr["v"][na + nb:] = table[len(r1cs.values)]
    return CoeffArray(arr)


def _triplets(r1cs):
    """-> (values, A, B, C): the distinct coefficient values (ints) and per matrix (constraint, wire, value index) as
    numpy arrays, from the list form (files/r1cs.nim:62-80) or from a synthetic.SparseR1CS"""
    import numpy as np
    if hasattr(r1cs, "values"):
        return r1cs.values, r1cs.A, r1cs.B, r1cs.C
    values, index = [], {}
    mats = ([[], [], []], [[], [], []], [[], [], []])
    for i, con in enumerate(r1cs.constraints):
        for k in range(3):
            for (w, v) in con[k]:
                v %= R
                j = index.get(v)
                if j is None:
                    j = index[v] = len(values)
                    values.append(v)
                mats[k][0].append(i), mats[k][1].append(w), mats[k][2].append(j)
    return (values,) + tuple(tuple(np.array(x, dtype=np.uint32) for x in m) for m in mats)


def columnDots(r1cs, lag, ctx):
    """The sparse column dot products of fake_setup.nim:159-187, 254-256: taus_M[wire] = sum over the entries of
    column `wire` of matrix M of value * L_row(tau), for M = A (incl. the dummy rows of the public IO, :59-63), B, C
    -- on the GPU: the transposed matrices through the row-balanced kernel of buildABC (g16_spmv_fr)."""
    import numpy as np
    values, A, B, C = _triplets(r1cs)
    neqs = r1cs.nConstraints if hasattr(r1cs, "nConstraints") else len(r1cs.constraints)
    npub = r1cs.nPubIn + r1cs.nPubOut
    table = np.frombuffer(F.frSeqToMontBytes(list(values) + [1]), dtype=np.uint8).reshape(-1, 32)
    lagb = F.frSeqToMontBytes(lag)
    out = []
    for k, (rows, wires, vi) in enumerate((A, B, C)):
        rows, wires, vi = (np.asarray(x, dtype=np.uint32) for x in (rows, wires, vi))
        if k == 0:
            rows = np.concatenate([rows, np.arange(neqs, neqs + npub + 1, dtype=np.uint32)])
            wires = np.concatenate([wires, np.arange(npub + 1, dtype=np.uint32)])
            vi = np.concatenate([vi, np.full(npub + 1, len(values), dtype=np.uint32)])
        out.append(ctx.spmv(wires, rows, table[vi], lagb, r1cs.nWires))
    return out            # Montgomery Fr bytes, nWires elements each


_lag_cache = {}


def lagrangeTaus(logDom: int, tau: int):
    """[L_k(tau) for k < 2^logDom]  == evalLagrangePolyAt (math/poly.nim:242-250) for every k.  (The last two results
    are kept: bench.py builds two keys over one domain from the same toxic waste.)"""
    key = (logDom, tau % R)
    if key not in _lag_cache:
        while len(_lag_cache) >= 2:
            _lag_cache.pop(next(iter(_lag_cache)))
        _lag_cache[key] = _lagrangeTaus(logDom, tau)
    return _lag_cache[key]


def _lagrangeTaus(logDom: int, tau: int):
    dom = 1 << logDom
    g = F.domainGen(logDom)
    ztau = (pow(tau, dom, R) - 1) % R
    invN = F.invFr(dom % R)
    omegas, denoms, w = [], [], 1
    for _ in range(dom):
        omegas.append(w)
        d = (tau - w) % R
        if d == 0:
            raise AssertionError("point should be outside the domain")
        denoms.append(d)
        w = w * g % R
    dinv = F.batchInverseFr(denoms)
    c = ztau * invN % R
    return [omegas[k] * c % R * dinv[k] % R for k in range(dom)]


def fakeCircuitSetup(r1cs: R1CS, toxic: ToxicWaste, flavour=Snarkjs, ctx=None) -> ZKey:
    """fake_setup.nim:201-326"""
    ctx = ctx or default_context()
    neqs = r1cs.nConstraints if hasattr(r1cs, "nConstraints") else len(r1cs.constraints)
    npub = r1cs.nPubIn + r1cs.nPubOut
    logDom = F.ceilingLog2(neqs + npub + 1)
    dom = 1 << logDom
    nvars = r1cs.nWires
    tau = toxic.tau % R
    lag = lagrangeTaus(logDom, tau)

    bytesA, bytesB, bytesC = columnDots(r1cs, lag, ctx)
    tausA, tausB, tausC = (F.frSeqFromMontBytes(x) for x in (bytesA, bytesB, bytesC))

    gammaInv, deltaInv = F.invFr(toxic.gamma), F.invFr(toxic.delta)
    comb = [(toxic.beta * tausA[j] + toxic.alpha * tausB[j] + tausC[j]) % R for j in range(nvars)]
    if flavour == JensGroth:
        ztau = (pow(tau, dom, R) - 1) % R
        hs, tp = [], 1
        for _ in range(dom):                       # fake_setup.nim:290-292
            hs.append(deltaInv * tp % R * ztau % R)
            tp = tp * tau % R
    else:                                          # fake_setup.nim:299-302: delta^-1 L_{2i+1}(tau), doubled domain
        lag2 = lagrangeTaus(logDom + 1, tau)
        hs = [deltaInv * lag2[2 * i + 1] % R for i in range(dom)]

    enc = F.frSeqToMontBytes
    g1 = lambda xs: ctx.fixed_base(1, enc(xs))     # noqa: E731   `y ** gen1`
    g2 = lambda xs: ctx.fixed_base(2, enc(xs))     # noqa: E731   `y ** gen2`
    spec1 = g1([toxic.alpha, toxic.beta, toxic.delta])
    spec2 = g2([toxic.beta, toxic.gamma, toxic.delta])
    zkey = ZKey()
    zkey.header = GrothHeader("bn128", flavour, nvars, npub, dom, logDom)
    zkey.specPoints = SpecPoints(alpha1=spec1[0:64], beta1=spec1[64:128], delta1=spec1[128:192],
                                 beta2=spec2[0:128], gamma2=spec2[128:256], delta2=spec2[256:384])
    zkey.pointsIC = g1([gammaInv * comb[j] % R for j in range(npub + 1)])
    zkey.pPoints = ProverPoints(
        pointsA1=ctx.fixed_base(1, bytesA), pointsB1=ctx.fixed_base(1, bytesB), pointsB2=ctx.fixed_base(2, bytesB),
        pointsC1=g1([deltaInv * comb[j] % R for j in range(npub + 1, nvars)]),
        pointsH1=g1(hs))
    zkey.coeffs = r1csToCoeffArray(r1cs) if hasattr(r1cs, "values") else r1csToCoeffs(r1cs)
    return zkey
