"""groth16/files/export_json.nim: snarkjs-compatible proof.json / public.json (decimal strings, projective
third coordinate "1")."""
from __future__ import annotations

from .. import bn128 as F
from ..prover import Proof


def _q(x: int) -> str:
    return '"' + str(x) + '"'


def _fp(b: bytes) -> int:
    return F.fpFromMontBytes(b)


def exportPublicIO(fpath: str, prf: Proof) -> None:
    """export_json.nim:25-44 (publicIO[0] is the constant 1 and is skipped; the reference writes invalid JSON
    when there are no public signals -- here an empty list is written)."""
    vals = F.frSeqFromMontBytes(prf.publicIO)
    assert len(vals) > 0 and vals[0] == 1
    with open(fpath, "w") as f:
        if len(vals) == 1:
            f.write("[ ]\n")
            return
        for i in range(1, len(vals)):
            f.write(("[ " if i == 1 else ", ") + _q(vals[i]) + "\n")
        f.write("] \n")


def _writeG1(f, p: bytes) -> None:                                       # export_json.nim:55-59
    f.write("    [ " + _q(_fp(p[0:32])) + "\n")
    f.write("    , " + _q(_fp(p[32:64])) + "\n")
    f.write("    , " + _q(1) + "\n")
    f.write("    ]\n")


def _writeFp2(f, c: str, z) -> None:                                     # export_json.nim:48-53
    f.write("    " + c + " [ " + _q(z[0]) + "\n")
    f.write("      , " + _q(z[1]) + "\n")
    f.write("      ]\n")


def _writeG2(f, p: bytes) -> None:                                       # export_json.nim:61-65
    _writeFp2(f, "[", (_fp(p[0:32]), _fp(p[32:64])))
    _writeFp2(f, ",", (_fp(p[64:96]), _fp(p[96:128])))
    _writeFp2(f, ",", (1, 0))
    f.write("    ]\n")


def exportProof(fpath: str, prf: Proof) -> None:
    """export_json.nim:70-80"""
    with open(fpath, "w") as f:
        f.write('{ "protocol": "groth16"\n')
        f.write(', "curve":    "bn128"\n')
        f.write(', "pi_a":\n')
        _writeG1(f, prf.pi_a)
        f.write(', "pi_b":\n')
        _writeG2(f, prf.pi_b)
        f.write(', "pi_c":\n')
        _writeG1(f, prf.pi_c)
        f.write("}\n")


def exportVKey(fpath: str, vkey) -> None:
    """snarkjs `verification_key.json` for a VKey (verifier.py).  NOT in the reference (it only exports keys to Sage,
    files/export_sage.nim:36-60; snarkjs users run `snarkjs zkey export verificationkey`): needed so that proofs of
    fake-setup keys can be checked with `snarkjs groth16 verify vkey.json public.json proof.json`.  `vk_alphabeta_12`
    is omitted: snarkjs recomputes the pairing from vk_alpha_1 / vk_beta_2 and does not read it when verifying."""
    import json

    def g1(p):
        return [str(_fp(p[0:32])), str(_fp(p[32:64])), "1"]

    def g2(p):
        return [[str(_fp(p[0:32])), str(_fp(p[32:64]))], [str(_fp(p[64:96])), str(_fp(p[96:128]))], ["1", "0"]]
    s = vkey.spec
    ic = vkey.pointsIC
    doc = {"protocol": "groth16", "curve": vkey.curve, "nPublic": vkey.npubs,
           "vk_alpha_1": g1(s.alpha1), "vk_beta_2": g2(s.beta2), "vk_gamma_2": g2(s.gamma2), "vk_delta_2": g2(s.delta2),
           "IC": [g1(ic[i:i + 64]) for i in range(0, len(ic), 64)]}
    with open(fpath, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
