"""Host mirror of groth16/fake_setup.nim: a "fake" circuit-specific trusted setup from explicit toxic
waste.  The O(n) scalar bookkeeping (Lagrange values at tau, sparse column dot products) is host code as in
the reference; every `y ** gen1` / `y ** gen2` (fake_setup.nim:258-261, 273-277, 290-302) -- the expensive
part: ~4N+n G1 and N G2 scalar multiplications -- runs on the GPU (g16_fixed_base_g1/g2)."""
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


def lagrangeTaus(logDom: int, tau: int):
    """[L_k(tau) for k < 2^logDom]  == evalLagrangePolyAt (math/poly.nim:242-250) for every k."""
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
    neqs = len(r1cs.constraints)
    npub = r1cs.nPubIn + r1cs.nPubOut
    logDom = F.ceilingLog2(neqs + npub + 1)
    dom = 1 << logDom
    nvars = r1cs.nWires
    tau = toxic.tau % R
    lag = lagrangeTaus(logDom, tau)

    # sparse column dot products with the Lagrange values (fake_setup.nim:159-187, 254-256)
    tausA, tausB, tausC = [0] * nvars, [0] * nvars, [0] * nvars
    for i, (A, B, C) in enumerate(r1cs.constraints):
        li = lag[i]
        for (w, v) in A:
            tausA[w] = (tausA[w] + v * li) % R
        for (w, v) in B:
            tausB[w] = (tausB[w] + v * li) % R
        for (w, v) in C:
            tausC[w] = (tausC[w] + v * li) % R
    for i in range(neqs, neqs + npub + 1):
        tausA[i - neqs] = (tausA[i - neqs] + lag[i]) % R

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
        pointsA1=g1(tausA), pointsB1=g1(tausB), pointsB2=g2(tausB),
        pointsC1=g1([deltaInv * comb[j] % R for j in range(npub + 1, nvars)]),
        pointsH1=g1(hs))
    zkey.coeffs = r1csToCoeffs(r1cs)
    return zkey
