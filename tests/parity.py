"""Full-proof parity check shared by the GPU tests and by bench.py's correctness gate (TEST INFRASTRUCTURE: the
oracle is the checker here, never the thing measured or shipped).

The C oracle recomputes what generateProofWithMask computes (reference groth16/prover.nim:215-304) -- buildABC
(:56-73), the H scalars (:158-181 or :118-148), the five MSMs (:282-302) -- and the Python oracle finishes the O(1)
mask algebra (:279-302); the GPU proof must equal the result bit for bit and satisfy the pairing equation
(verifier.nim:31-52, the reference's own test, testProver.nim:59-73)."""
import time

from oracle import bn254_ref as o


def oracle_zkey_shell(zkey):
    """oracle ZKey carrying only what the mask algebra and the verifier need (no big point lists)"""
    hdr, sp = zkey.header, zkey.specPoints
    oz = o.ZKey()
    oz.flavour = o.SNARKJS if hdr.flavour == 1 else o.JENS_GROTH
    oz.nvars, oz.npubs, oz.domainSize = hdr.nvars, hdr.npubs, hdr.domainSize
    oz.alpha1, oz.beta1, oz.delta1 = (o.g1_from_bytes(x) for x in (sp.alpha1, sp.beta1, sp.delta1))
    oz.beta2, oz.gamma2, oz.delta2 = (o.g2_from_bytes(x) for x in (sp.beta2, sp.gamma2, sp.delta2))
    oz.pointsIC = [o.g1_from_bytes(zkey.pointsIC[i:i + 64]) for i in range(0, len(zkey.pointsIC), 64)]
    oz.pointsA1 = oz.pointsB1 = oz.pointsB2 = [None] * hdr.nvars
    oz.pointsC1, oz.pointsH1, oz.coeffs = [None] * (hdr.nvars - hdr.npubs - 1), [None] * hdr.domainSize, []
    return oz


def oracle_proof(orc, zkey, wit_ints, wbytes_mont, r, s):
    """-> (oracle Proof, oracle ZKey shell, seconds of CPU work in the C oracle).  wit_ints: the witness as Python
    ints (only the public prefix is read); wbytes_mont: the same witness as Montgomery Fr bytes."""
    from nim_groth16_amd.zkey_types import packCoeffs
    pts, hdr = zkey.pPoints, zkey.header
    log2n = hdr.logDomainSize
    packed = packCoeffs(zkey.coeffs)
    t0 = time.perf_counter()
    Az, Bz, Cz = orc.build_abc(packed, wbytes_mont, log2n)
    quot = orc.quotient_snarkjs if hdr.flavour == 1 else orc.quotient_jensgroth
    qs = quot(Az, Bz, Cz, log2n, parallel=True)
    mA = orc.msm(1, wbytes_mont, pts.pointsA1)
    mB1 = orc.msm(1, wbytes_mont, pts.pointsB1)
    mB2 = orc.msm(2, wbytes_mont, pts.pointsB2)
    mH = orc.msm(1, qs, pts.pointsH1)
    mC = orc.msm(1, wbytes_mont[32 * (hdr.npubs + 1):], pts.pointsC1)
    cpu_s = time.perf_counter() - t0
    it = iter([o.g1_from_bytes(mA), o.g1_from_bytes(mB1), o.g2_from_bytes(mB2), o.g1_from_bytes(mH),
               o.g1_from_bytes(mC)])
    oz = oracle_zkey_shell(zkey)
    ref = o.generate_proof_with_mask(oz, wit_ints, r, s, msm_g1=lambda c_, p_: next(it),
                                     msm_g2=lambda c_, p_: next(it), quotient=lambda *a: [0] * hdr.domainSize)
    return ref, oz, cpu_s


def check_gpu_proof(orc, zkey, wit_ints, wbytes_mont, r, s, proof, ctx=None):
    """proof = (pi_a, pi_b, pi_c) bytes from the GPU.  Raises AssertionError on any mismatch; returns the CPU
    seconds the C oracle took (bench.py's cpu_baseline leg)."""
    ref, oz, cpu_s = oracle_proof(orc, zkey, wit_ints, wbytes_mont, r, s)
    got = (o.g1_from_bytes(proof[0]), o.g2_from_bytes(proof[1]), o.g1_from_bytes(proof[2]))
    assert got == (ref.pi_a, ref.pi_b, ref.pi_c), "GPU proof differs from the CPU oracle's proof"
    assert o.verify_proof(oz, ref), "proof does not satisfy the pairing equation"
    if ctx is not None:
        from nim_groth16_amd import Proof, extractVKey, verifyProof
        pio = wbytes_mont[:32 * (zkey.header.npubs + 1)]      # Proof.publicIO = witness[0..npubs] (prover.nim:238-240)
        assert verifyProof(extractVKey(zkey), Proof(pio, proof[0], proof[1], proof[2]), ctx), \
            "the GPU verifier rejects the proof"
    return cpu_s
