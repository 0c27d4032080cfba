"""Host mirror of groth16/verifier.nim and of the VKey part of groth16/zkey_types.nim.

    verifyProof(vkey, prf)   -- verifier.nim:31-52:  e(-A,B) e(alpha,beta) e(C,delta) e(sum pub_i IC_i, gamma) == 1
    extractVKey(zkey)        -- zkey_types.nim:69-73

The pairing product runs on the GPU (csrc/pairing.cuh, one lane per pairing, batched over proofs with
verifyProofs); like the reference, malformed points raise AssertionError ("pi_a is not in G1", ...).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence

from ._lib import Context, VerifyingKey, default_context
from .prover import Proof
from .zkey_types import SpecPoints, ZKey


@dataclass
class VKey:                          # zkey_types.nim:62-66
    curve: str
    spec: SpecPoints
    pointsIC: bytes                  # VerifierPoints (zkey_types.nim:33-34): npubs + 1 G1 points

    @property
    def npubs(self) -> int:
        return len(self.pointsIC) // 64 - 1


def extractVKey(zkey: ZKey) -> VKey:
    """zkey_types.nim:69-73"""
    return VKey(curve=zkey.header.curve, spec=zkey.specPoints, pointsIC=zkey.pointsIC)


def loadVerifyingKey(vkey: VKey, ctx: Context = None) -> VerifyingKey:
    """Uploads the key once and precomputes the (alpha, beta) Miller value (the reference's spec.alphaBeta)."""
    ctx = ctx or default_context()
    s = vkey.spec
    return VerifyingKey(ctx, vkey.npubs, s.alpha1, s.beta2, s.gamma2, s.delta2, vkey.pointsIC)


_ASSERTS = {-1: "pi_a is not in G1", -2: "pi_b is not in G2", -3: "pi_c is not in G1",
            -4: "pi_b is not in the order-r subgroup of G2",
            -5: "a proof coordinate is not a canonical field element (>= p)",
            -6: "a public input is not a canonical scalar (>= r)"}


def verifyProofs(vkey, proofs: Sequence[Proof], ctx: Context = None, subgroup: bool = False) -> List[bool]:
    """Batch form: one GPU launch sequence for all proofs.  `vkey` is a VKey or an already loaded VerifyingKey."""
    dev = vkey if isinstance(vkey, VerifyingKey) else loadVerifyingKey(vkey, ctx)
    for prf in proofs:
        assert prf.curve == "bn128"                                   # verifier.nim:33
        assert len(prf.publicIO) == 32 * (dev.npubs + 1), "publicIO must hold npubs + 1 values (incl. the leading 1)"
    st = dev.verify([(p.pi_a, p.pi_b, p.pi_c) for p in proofs], b"".join(p.publicIO for p in proofs),
                    mont=True, subgroup=subgroup)
    for code in st:
        assert code >= 0, _ASSERTS.get(code, f"malformed proof ({code})")   # verifier.nim:35-37
    return [code == 1 for code in st]


def verifyProof(vkey, prf: Proof, ctx: Context = None, subgroup: bool = False) -> bool:
    """verifier.nim:31-52"""
    return verifyProofs(vkey, [prf], ctx, subgroup)[0]
