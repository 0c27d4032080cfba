"""nim_groth16_amd -- MI355X-native MSM / NTT hot path of a Groth16 (BN254) prover.

Host-side mirror of the reference's Nim interface for that path (codex-storage/nim-groth16:
groth16/bn128/msm.nim, groth16/math/ntt.nim, groth16/math/domain.nim) over the C ABI of
libg16hip.so (include/g16hip.h).  All arithmetic runs in hand-written HIP kernels; there is no CPU
fallback -- importing works anywhere, but every compute call raises without the HIP library + a GPU.
"""
from ._lib import (G16Error, Context, DeviceGroup, GroupKey, ProvingKey, PointSet, VerifyingKey,  # noqa: F401
                   lib_path, load_library)
from .msm import (msmMultiThreadedG1, msmMultiThreadedG2, msmG1, msmG2)  # noqa: F401
from .ntt import (Domain, createDomain, forwardNTT, inverseNTT, extendAndForwardNTT,  # noqa: F401
                  polyForwardNTT, polyInverseNTT)
from .prover import (ABC, Proof, Mask, Witness, buildABC, computeQuotientPointwise,  # noqa: F401
                     computeSnarkjsScalarCoeffs, generateProof, generateProofWithMask,
                     generateProofWithTrivialMask, loadGroupKey, loadProvingKey)
from .verifier import VKey, extractVKey, loadVerifyingKey, verifyProof, verifyProofs  # noqa: F401
from .zkey_types import ZKey, GrothHeader, SpecPoints, ProverPoints, JensGroth, Snarkjs  # noqa: F401
