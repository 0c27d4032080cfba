## groth16/gpu/g16hip.nim -- Nim binding of libg16hip.so (include/g16hip.h) for codex-storage/nim-groth16.
##
## Keeps the reference's proc names and signatures (groth16/bn128/msm.nim:89,128,202-203; groth16/math/ntt.nim:55,139;
## groth16/prover.nim:215; groth16/verifier.nim:31) so that prover.nim compiles against it unchanged.
## Assembled from INTEGRATION.md sections 2 and "Verifier".  NOT compile-tested: this image has no `nim` / `nimble`
## and constantine is not vendored (SURVEY.md section 8c); the same boundary is exercised through ctypes
## (nim_groth16_amd/_lib.py), plain C (examples/c_abi_demo.c) and C++ (tools/g16prove.cpp) by the GPU tests.
##
## Build: nim c --passC:-I<repo>/include --passL:-L<repo>/nim_groth16_amd/csrc --passL:-lg16hip ...

# groth16/gpu/g16hip.nim -- binds libg16hip.so; keeps the signatures of bn128/msm.nim and math/ntt.nim
import groth16/bn128
import groth16/math/domain
import groth16/zkey_types
import groth16/files/witness

{.passL: "-lg16hip".}
type
  G16Ctx   {.importc: "g16_ctx",    header: "g16hip.h", incompleteStruct.} = object
  G16PKey  {.importc: "g16_pkey",   header: "g16hip.h", incompleteStruct.} = object
  G16Coeff {.importc: "g16_coeff",  header: "g16hip.h".} = object
    matrix, row, col, reserved: uint32
    value: array[32, byte]
  G16PKeyDesc {.importc: "g16_pkey_desc", header: "g16hip.h".} = object
    nvars, npubs, log2_domain, flavour: uint32
    pointsA1, pointsB1, pointsB2, pointsC1, pointsH1: pointer
    coeffs: ptr G16Coeff
    ncoeffs: csize_t
    alpha1, beta1, delta1, beta2, delta2: pointer
    shard_index, shard_count: uint32
  G16Proof {.importc: "g16_proof", header: "g16hip.h".} = object
    pi_a: array[64, byte]; pi_b: array[128, byte]; pi_c: array[64, byte]

const G16_SCALARS_MONT = 1'u32

proc g16_ctx_create(device: int32, ctx: ptr ptr G16Ctx): int32 {.importc, header: "g16hip.h".}
proc g16_selftest(ctx: ptr G16Ctx): int32 {.importc, header: "g16hip.h".}
proc g16_last_error(ctx: ptr G16Ctx): cstring {.importc, header: "g16hip.h".}
proc g16_msm_g1(ctx: ptr G16Ctx, scalars: pointer, flags: uint32, points: pointer, n: csize_t, res: pointer): int32 {.importc, header: "g16hip.h".}
proc g16_msm_g2(ctx: ptr G16Ctx, scalars: pointer, flags: uint32, points: pointer, n: csize_t, res: pointer): int32 {.importc, header: "g16hip.h".}
proc g16_ntt_fr(ctx: ptr G16Ctx, src, dst: pointer, log2n: uint32, inverse: int32): int32 {.importc, header: "g16hip.h".}
proc g16_quotient(ctx: ptr G16Ctx, az, bz, cz: pointer, log2n, flavour: uint32, res: pointer): int32 {.importc, header: "g16hip.h".}
proc g16_pkey_create(ctx: ptr G16Ctx, desc: ptr G16PKeyDesc, key: ptr ptr G16PKey): int32 {.importc, header: "g16hip.h".}
proc g16_prove(ctx: ptr G16Ctx, key: ptr G16PKey, witness: pointer, flags: uint32, r, s: pointer, res: ptr G16Proof): int32 {.importc, header: "g16hip.h".}
# points at infinity per ProverPoints array (A1, B1, B2, C1, H1, B1-and-B2) and whether A1 / B1+B2 run on compacted
# entry lists: snarkjs keys hold (0,0) for every wire absent from a matrix (curves.nim:95-107 accepts them)
proc g16_pkey_inf_counts(key: ptr G16PKey, res: ptr array[8, csize_t]): int32 {.importc, header: "g16hip.h".}

var gctx: ptr G16Ctx

proc check(rc: int32) =
  if rc != 0: raise newException(AssertionDefect, "g16hip: " & $g16_last_error(gctx))

proc initG16Hip*(device = 0) =
  doAssert sizeof(Fr) == 32 and sizeof(G1) == 64 and sizeof(G2) == 128, "constantine layout changed"
  check g16_ctx_create(int32(device), addr gctx)
  check g16_selftest(gctx)

# --- drop-ins for groth16/bn128/msm.nim:89,128,202-203 ---------------------------------------------------
proc msmMultiThreadedG1*(nthreads_hint: int, coeffs: seq[Fr], points: seq[G1]): G1 =
  assert coeffs.len == points.len, "incompatible sequence lengths"          # msm.nim:97
  if coeffs.len == 0: return infG1                                            # msm.nim:117
  check g16_msm_g1(gctx, unsafeAddr coeffs[0], G16_SCALARS_MONT, unsafeAddr points[0], csize_t(coeffs.len), addr result)

proc msmMultiThreadedG2*(nthreads_hint: int, coeffs: seq[Fr], points: seq[G2]): G2 =
  assert coeffs.len == points.len, "incompatible sequence lengths"          # msm.nim:131
  if coeffs.len == 0: return infG2
  check g16_msm_g2(gctx, unsafeAddr coeffs[0], G16_SCALARS_MONT, unsafeAddr points[0], csize_t(coeffs.len), addr result)

proc msmG1*(coeffs: seq[Fr], points: seq[G1]): G1 = msmMultiThreadedG1(0, coeffs, points)
proc msmG2*(coeffs: seq[Fr], points: seq[G2]): G2 = msmMultiThreadedG2(0, coeffs, points)

# --- drop-ins for groth16/math/ntt.nim:55,139 ---------------------------------------------------------------
proc forwardNTT*(src: seq[Fr], D: Domain): seq[Fr] =
  assert D.domainSize == (1 shl D.logDomainSize) and D.domainSize == src.len   # ntt.nim:56-57
  result = newSeq[Fr](src.len)
  check g16_ntt_fr(gctx, unsafeAddr src[0], addr result[0], uint32(D.logDomainSize), 0)

proc inverseNTT*(src: seq[Fr], D: Domain): seq[Fr] =
  assert D.domainSize == (1 shl D.logDomainSize) and D.domainSize == src.len   # ntt.nim:140-141
  result = newSeq[Fr](src.len)
  check g16_ntt_fr(gctx, unsafeAddr src[0], addr result[0], uint32(D.logDomainSize), 1)

# --- drop-in for computeSnarkjsScalarCoeffs / computeQuotientPointwise (prover.nim:118-181) ------------------
proc computeQuotientGpu*(valuesAz, valuesBz, valuesCz: seq[Fr], flavour: Flavour): seq[Fr] =
  let n = valuesAz.len
  result = newSeq[Fr](n)
  check g16_quotient(gctx, unsafeAddr valuesAz[0], unsafeAddr valuesBz[0], unsafeAddr valuesCz[0],
                     uint32(createDomain(n).logDomainSize), uint32(ord(flavour)), addr result[0])

var gkey: ptr G16PKey
proc loadKeyGpu*(zkey: ZKey) =
  var cs = newSeq[G16Coeff](zkey.coeffs.len)
  for i, c in zkey.coeffs:                                  # zkey_types.nim:48-52
    cs[i] = G16Coeff(matrix: uint32(ord(c.matrix)), row: uint32(c.row), col: uint32(c.col))
    copyMem(addr cs[i].value, unsafeAddr c.coeff, 32)
  var d = G16PKeyDesc(nvars: uint32(zkey.header.nvars), npubs: uint32(zkey.header.npubs),
    log2_domain: uint32(zkey.header.logDomainSize), flavour: uint32(ord(zkey.header.flavour)),
    pointsA1: unsafeAddr zkey.pPoints.pointsA1[0], pointsB1: unsafeAddr zkey.pPoints.pointsB1[0],
    pointsB2: unsafeAddr zkey.pPoints.pointsB2[0], pointsC1: unsafeAddr zkey.pPoints.pointsC1[0],
    pointsH1: unsafeAddr zkey.pPoints.pointsH1[0], coeffs: addr cs[0], ncoeffs: csize_t(cs.len),
    alpha1: unsafeAddr zkey.specPoints.alpha1, beta1: unsafeAddr zkey.specPoints.beta1,
    delta1: unsafeAddr zkey.specPoints.delta1, beta2: unsafeAddr zkey.specPoints.beta2,
    delta2: unsafeAddr zkey.specPoints.delta2, shard_index: 0, shard_count: 1)
  check g16_pkey_create(gctx, addr d, addr gkey)

proc generateProofWithMask*(nthreads: int, printTimings: bool, zkey: ZKey, wtns: Witness, mask: Mask): Proof =
  assert zkey.header.curve == wtns.curve and zkey.header.nvars == wtns.values.len   # prover.nim:224,236
  var p: G16Proof
  check g16_prove(gctx, gkey, unsafeAddr wtns.values[0], G16_SCALARS_MONT, unsafeAddr mask.r, unsafeAddr mask.s, addr p)
  result = Proof(curve: "bn128", publicIO: wtns.values[0..zkey.header.npubs])        # prover.nim:238-240
  copyMem(addr result.pi_a, addr p.pi_a, 64); copyMem(addr result.pi_b, addr p.pi_b, 128); copyMem(addr result.pi_c, addr p.pi_c, 64)

# --- one proof over several GPUs of the node: the device group (include/g16hip.h "device group") ------------------------
# The reference shards every MSM over Taskpool threads (msm.nim:96-122) and runs the three coset pipelines as three tasks
# (prover.nim:165-173); a group does the same over GPUs, one host thread per device INSIDE the library.  The Nim host only
# names the devices:   initG16Hip([0, 1, 2, 3, 4, 5, 6, 7]);  loadKeyGpu(zkey);  generateProofWithMask(...)
type
  G16Group    {.importc: "g16_group",      header: "g16hip.h", incompleteStruct.} = object
  G16GroupKey {.importc: "g16_group_pkey", header: "g16hip.h", incompleteStruct.} = object
proc g16_group_create(devices: ptr int32, ndev: int32, grp: ptr ptr G16Group): int32 {.importc, header: "g16hip.h".}
proc g16_group_last_error(grp: ptr G16Group): cstring {.importc, header: "g16hip.h".}
proc g16_group_pkey_create(grp: ptr G16Group, desc: ptr G16PKeyDesc, key: ptr ptr G16GroupKey): int32 {.importc, header: "g16hip.h".}
proc g16_group_prove(grp: ptr G16Group, key: ptr G16GroupKey, witness: pointer, flags: uint32, r, s: pointer,
                     res: ptr G16Proof): int32 {.importc, header: "g16hip.h".}

var ggroup: ptr G16Group
var ggroupKey: ptr G16GroupKey

proc checkGroup(rc: int32) =
  if rc != 0: raise newException(AssertionDefect, "g16hip group: " & $g16_group_last_error(ggroup))

proc initG16Hip*(devices: openArray[int]) =
  ## several GPUs: MSMs, NTTs and the verifier of single calls run on devices[0]; proofs are sharded over all of them
  initG16Hip(devices[0])
  var ds = newSeq[int32](devices.len)
  for i, d in devices: ds[i] = int32(d)
  if g16_group_create(addr ds[0], int32(ds.len), addr ggroup) != 0:
    raise newException(AssertionDefect, "g16hip: g16_group_create failed")

proc loadKeyGroup*(zkey: ZKey) =
  ## like loadKeyGpu, sharded: member g keeps index range g of every ProverPoints array (msm.nim:105-115)
  var cs = newSeq[G16Coeff](zkey.coeffs.len)
  for i, c in zkey.coeffs:
    cs[i] = G16Coeff(matrix: uint32(ord(c.matrix)), row: uint32(c.row), col: uint32(c.col))
    copyMem(addr cs[i].value, unsafeAddr c.coeff, 32)
  var d = G16PKeyDesc(nvars: uint32(zkey.header.nvars), npubs: uint32(zkey.header.npubs),
    log2_domain: uint32(zkey.header.logDomainSize), flavour: uint32(ord(zkey.header.flavour)),
    pointsA1: unsafeAddr zkey.pPoints.pointsA1[0], pointsB1: unsafeAddr zkey.pPoints.pointsB1[0],
    pointsB2: unsafeAddr zkey.pPoints.pointsB2[0], pointsC1: unsafeAddr zkey.pPoints.pointsC1[0],
    pointsH1: unsafeAddr zkey.pPoints.pointsH1[0], coeffs: addr cs[0], ncoeffs: csize_t(cs.len),
    alpha1: unsafeAddr zkey.specPoints.alpha1, beta1: unsafeAddr zkey.specPoints.beta1,
    delta1: unsafeAddr zkey.specPoints.delta1, beta2: unsafeAddr zkey.specPoints.beta2,
    delta2: unsafeAddr zkey.specPoints.delta2, shard_index: 0, shard_count: 1)
  checkGroup g16_group_pkey_create(ggroup, addr d, addr ggroupKey)

proc generateProofWithMaskGroup*(nthreads: int, printTimings: bool, zkey: ZKey, wtns: Witness, mask: Mask): Proof =
  ## generateProofWithMask (prover.nim:215-304) over the device group: bit-identical to the one-GPU proof
  assert zkey.header.curve == wtns.curve and zkey.header.nvars == wtns.values.len   # prover.nim:224,236
  var p: G16Proof
  checkGroup g16_group_prove(ggroup, ggroupKey, unsafeAddr wtns.values[0], G16_SCALARS_MONT, unsafeAddr mask.r,
                             unsafeAddr mask.s, addr p)
  result = Proof(curve: "bn128", publicIO: wtns.values[0..zkey.header.npubs])        # prover.nim:238-240
  copyMem(addr result.pi_a, addr p.pi_a, 64); copyMem(addr result.pi_b, addr p.pi_b, 128); copyMem(addr result.pi_c, addr p.pi_c, 64)

# continues groth16/gpu/g16hip.nim above: same `header:` style, every type it names is declared here
type
  G16VKey {.importc: "g16_vkey", header: "g16hip.h", incompleteStruct.} = object
  G16VKeyDesc {.importc: "g16_vkey_desc", header: "g16hip.h".} = object
    npubs: uint32
    alpha1, beta2, gamma2, delta2, pointsIC: pointer

proc g16_vkey_create(ctx: ptr G16Ctx, desc: ptr G16VKeyDesc, key: ptr ptr G16VKey): int32 {.importc, header: "g16hip.h".}
proc g16_vkey_destroy(key: ptr G16VKey) {.importc, header: "g16hip.h".}
proc g16_verify(ctx: ptr G16Ctx, key: ptr G16VKey, proofs: ptr G16Proof, publicIO: pointer, flags: uint32,
                count: csize_t, status: ptr int32): int32 {.importc, header: "g16hip.h".}

proc verifyProof*(vkey: VKey, prf: Proof): bool =
  assert prf.curve == "bn128"                                             # verifier.nim:33
  var d = G16VKeyDesc(npubs: uint32(vkey.vpoints.pointsIC.len - 1), alpha1: unsafeAddr vkey.spec.alpha1,
    beta2: unsafeAddr vkey.spec.beta2, gamma2: unsafeAddr vkey.spec.gamma2, delta2: unsafeAddr vkey.spec.delta2,
    pointsIC: unsafeAddr vkey.vpoints.pointsIC[0])
  var k: ptr G16VKey
  check g16_vkey_create(gctx, addr d, addr k)            # keep `k` per circuit: it caches the (alpha,beta) Miller value
  defer: g16_vkey_destroy(k)
  var p: G16Proof; var st: int32
  copyMem(addr p.pi_a, unsafeAddr prf.pi_a, 64); copyMem(addr p.pi_b, unsafeAddr prf.pi_b, 128); copyMem(addr p.pi_c, unsafeAddr prf.pi_c, 64)
  check g16_verify(gctx, k, addr p, unsafeAddr prf.publicIO[0], G16_SCALARS_MONT, 1, addr st)
  assert st != -1, "pi_a is not in G1"; assert st != -2, "pi_b is not in G2"; assert st != -3, "pi_c is not in G1"   # verifier.nim:35-37
  assert st != -5 and st != -6, "non-canonical field element in the proof / public input"
  result = st == 1
