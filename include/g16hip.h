/*
 * g16hip -- C ABI of the MI355X-native MSM / NTT hot path of a Groth16 (BN254) prover.
 *
 * Drop-in boundary for codex-storage/nim-groth16 (reference paths relative to /root/reference):
 * every entry point names the reference proc it replaces.  The reference has no FFI layer of its
 * own (it is pure Nim calling constantine); INTEGRATION.md shows the `importc` shims that give
 * these functions the reference's Nim signatures.
 *
 * Data layouts (identical to the reference's in-memory types):
 *   Fr, Fp   : 32 bytes, 4 x u64 little-endian limbs, Montgomery form R = 2^256, canonical (< modulus)
 *              (groth16/bn128/fields.nim:23-25, io.nim:60-92)
 *   Fp2      : 64 bytes  = { c0, c1 }                       (fields.nim:27-32)
 *   G1 affine: 64 bytes  = { x, y : Fp  }, infinity = (0,0)  (curves.nim:33,49)
 *   G2 affine: 128 bytes = { x, y : Fp2 }, infinity = (0,0)  (curves.nim:34,50)
 *   "std" scalars: 32 bytes canonical little-endian, NOT Montgomery (the .wtns layout,
 *              files/witness.nim:14,57-60)
 *
 * Error convention: every function returns 0 on success or a negative G16_E* code; nothing
 * throws or aborts across the boundary (the reference uses `assert`, msm.nim:97, ntt.nim:56-57).
 * A g16_ctx is used by one host thread at a time (any thread: every call makes the context's device the calling
 * thread's current HIP device); distinct contexts are independent.
 * Registered point sets, proving keys and verification keys (g16_points / g16_pkey / g16_vkey) are immutable and
 * belong to the DEVICE of the context that created them: any context of that device may use them, concurrently
 * (the in-flight proofs of one GPU share one resident key), and they may be released before or after any context.
 * Inputs are read-only and not retained after return, except explicitly registered point sets.
 * G16_* environment knobs are read once per process, at the first g16_ctx_create.
 * There is NO CPU fallback: without a usable HIP device every call fails with G16_ENODEV.
 */
#ifndef G16HIP_H
#define G16HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define G16_OK 0
#define G16_EINVAL (-1)  /* bad argument (null pointer, size mismatch, log2n out of range ...) */
#define G16_ENODEV (-2)  /* no HIP device / device init failed */
#define G16_EHIP (-3)    /* a HIP runtime call or kernel failed; see g16_last_error */
#define G16_ENOMEM (-4)  /* device or host allocation failed */
#define G16_ESELFTEST (-5)

typedef struct g16_ctx g16_ctx;
typedef struct g16_points g16_points; /* device-resident point set (ProverPoints member, zkey_types.nim:36-41) */

/* flags for the scalar argument of the MSM calls */
#define G16_SCALARS_MONT 1u /* Nim seq[Fr] limbs (Montgomery) -- what msm.nim:42-44 `toBig` consumes   */
#define G16_SCALARS_STD 0u  /* canonical little-endian (raw .wtns values)                              */
#define G16_SCALARS_DEVICE 2u /* g16_msm_points only: the scalar pointer is a device (HBM) pointer       */
#define G16_OUT_PARTIAL 4u    /* g16_msm_points only: write the 128/256-byte XYZZ partial, not the affine */
#define G16_OUT_DEVICE 8u     /* g16_prove_partials only: the output pointer is a device pointer           */
#define G16_NO_HOST_SYNC 32u  /* g16_prove_partials_begin/_end: do not block the host (see there)             */

/* ---- context ------------------------------------------------------------------------------------ */
int32_t g16_ctx_create(int32_t device, g16_ctx** out);
void g16_ctx_destroy(g16_ctx* ctx);
const char* g16_last_error(const g16_ctx* ctx);
/* run all work of this context on an existing HIP stream (e.g. torch's current stream); NULL = own stream */
int32_t g16_ctx_set_stream(g16_ctx* ctx, void* hip_stream);
int32_t g16_ctx_synchronize(g16_ctx* ctx);
/* waits for EVERYTHING the context has queued (its main stream and its MSM lanes) and forgets a pending
 * g16_prove_partials_begin (a later _end returns G16_EINVAL): what a caller does when its exchange between _begin and
 * _end failed, before it reuses or frees the buffers it handed in */
int32_t g16_ctx_cancel(g16_ctx* ctx);
/* layout / constant / arithmetic self-check on the device (sizeof, Montgomery one == frMontR io.nim:91,
 * gen1 on curve, small known-answer MSM and NTT).  The Nim shim calls it once at start-up. */
int32_t g16_selftest(g16_ctx* ctx);

/* ---- MSM: replaces msmMultiThreadedG1/G2 (groth16/bn128/msm.nim:89-158) and msmG1/msmG2 (:202-203),
 *      i.e. constantine's multiScalarMul_vartime + prj.affine (msm.nim:49-54, 76-81) ----------------- */
/* host pointers in, affine point out (64 / 128 bytes, host).  n == 0 -> infinity (0,0). */
int32_t g16_msm_g1(g16_ctx* ctx, const void* scalars, uint32_t scalar_flags, const void* points, size_t n,
                   void* out_affine);
int32_t g16_msm_g2(g16_ctx* ctx, const void* scalars, uint32_t scalar_flags, const void* points, size_t n,
                   void* out_affine);
/* same with scalars and points already resident in device memory (HBM); result to host memory */
int32_t g16_msm_g1_dev(g16_ctx* ctx, const void* d_scalars, uint32_t scalar_flags, const void* d_points, size_t n,
                       void* out_affine);
int32_t g16_msm_g2_dev(g16_ctx* ctx, const void* d_scalars, uint32_t scalar_flags, const void* d_points, size_t n,
                       void* out_affine);
/* partial result for multi-GPU sharding (msm.nim:105-119 chunks): 128 / 256 byte XYZZ accumulator
 * (X, Y, ZZ, ZZZ), Montgomery; combine with g16_g1_sum_partials / g16_g2_sum_partials. */
int32_t g16_msm_g1_partial_dev(g16_ctx* ctx, const void* d_scalars, uint32_t scalar_flags, const void* d_points,
                               size_t n, void* out_xyzz);
int32_t g16_msm_g2_partial_dev(g16_ctx* ctx, const void* d_scalars, uint32_t scalar_flags, const void* d_points,
                               size_t n, void* out_xyzz);
/* res = sum of `count` XYZZ partials (host memory, e.g. gathered over RCCL) -> affine; replaces the
 * `res += sync pending[k]` loop (msm.nim:117-119, 151-153) */
int32_t g16_g1_sum_partials(g16_ctx* ctx, const void* xyzz, size_t count, void* out_affine);
int32_t g16_g2_sum_partials(g16_ctx* ctx, const void* xyzz, size_t count, void* out_affine);

/* ---- registered point sets -------------------------------------------------------------------------
 * The five ProverPoints arrays (pointsA1/B1/B2/C1/H1, groth16/zkey_types.nim:36-41) are constant per
 * circuit and are loaded once (files/zkey.nim:201-224).  Registering a set copies it to HBM and
 * precomputes the window tables 2^(c w) * P_i, so that no MSM against it contains a doubling chain.
 * Because the factor lives in the table, all windows share one bucket set and the window size c is chosen
 * by a cost model (c = 20 for 2^20 points: 13 tables, 832 MiB per 2^20 G1 points, 1.6 GiB for G2). */
int32_t g16_points_register_g1(g16_ctx* ctx, const void* points, size_t n, g16_points** out);
int32_t g16_points_register_g2(g16_ctx* ctx, const void* points, size_t n, g16_points** out);
int32_t g16_points_register_g1_dev(g16_ctx* ctx, const void* d_points, size_t n, g16_points** out);
int32_t g16_points_register_g2_dev(g16_ctx* ctx, const void* d_points, size_t n, g16_points** out);
void g16_points_release(g16_points* pts);
size_t g16_points_count(const g16_points* pts);
/* points at infinity (0,0) in the set (legal inputs: curves.nim:95-107; snarkjs keys hold one for every wire absent
 * from a matrix).  A set with at least G16_INF_COMPACT percent of them (environment, default 10) gets bucket entry
 * lists that leave them out, instead of paying a loop trip per (0,0) entry. */
size_t g16_points_inf_count(const g16_points* pts);
/* window size c and number of tables chosen for this set: W = 254 / c + 1 windows x 1 or 2 multiplier tables
 * ([m][w][i] = 2^(c w + m) P_i).  An MSM against it performs count * W bucket additions + 2 reduction additions per
 * bucket: 2^(c-1) buckets with one table per window, 0.67 * 2^(c-1) with two (the class bucket set, msm.cuh) */
int32_t g16_points_info(const g16_points* pts, uint32_t* window_bits, uint32_t* ntables);
/* sum_i scalars[i] * P_i over the whole registered set (scalars: g16_points_count elements);
 * flags = G16_SCALARS_MONT/STD | G16_SCALARS_DEVICE | G16_OUT_PARTIAL.  This is the call a prover makes
 * per proof for msmMultiThreadedG1/G2 (groth16/prover.nim:282,288,294,301,302). */
int32_t g16_msm_points(g16_ctx* ctx, const g16_points* pts, const void* scalars, uint32_t flags, void* out);

/* ---- curve membership of a point array: replaces the per-point asserts of mkG1 / mkG2 that the reference runs
 * while loading a .zkey (groth16/bn128/curves.nim:95-107 via io.nim:240-250).  *first_bad = index of the first
 * point that is neither (0,0) nor on y^2 = x^3 + b, or (size_t)-1 when all are fine.  Like the reference this is
 * NOT a subgroup check.  Host pointers. */
int32_t g16_points_check_g1(g16_ctx* ctx, const void* points, size_t n, size_t* first_bad);
int32_t g16_points_check_g2(g16_ctx* ctx, const void* points, size_t n, size_t* first_bad);

/* ---- fixed-base multiples of the generators: out[i] = scalars[i] * gen1 (resp. gen2) --------------------
 * replaces the `y ** gen1` / `y ** gen2` comprehensions of the fake trusted setup
 * (groth16/fake_setup.nim:258-261, 273-277, 290-302; generators: curves.nim:112-124).  Used to build
 * proving keys for the synthetic benchmark circuits on the GPU box.  Host pointers. */
int32_t g16_fixed_base_g1(g16_ctx* ctx, const void* scalars, uint32_t scalar_flags, size_t n, void* out_points);
int32_t g16_fixed_base_g2(g16_ctx* ctx, const void* scalars, uint32_t scalar_flags, size_t n, void* out_points);

/* ---- NTT: replaces forwardNTT / inverseNTT (groth16/math/ntt.nim:55-77, 139-161) ------------------- */
/* natural order in and out; forward unscaled, inverse includes 1/n; omega = gen28^(2^(28-log2n))
 * (math/domain.nim:26-33).  src/dst: n = 2^log2n Fr elements (Montgomery), host memory. 0 <= log2n <= 28 */
int32_t g16_ntt_fr(g16_ctx* ctx, const void* src, void* dst, uint32_t log2n, int32_t inverse);
int32_t g16_ntt_fr_dev(g16_ctx* ctx, const void* d_src, void* d_dst, uint32_t log2n, int32_t inverse);

/* ---- quotient: replaces computeSnarkjsScalarCoeffs (flavour 1, groth16/prover.nim:158-181) and
 *      computeQuotientPointwise (flavour 0, prover.nim:118-148) -------------------------------------------
 * Az, Bz, Cz, out: n = 2^log2n Fr elements (Montgomery).  Snarkjs: out[j] = A1[j]*B1[j] - C1[j] on the coset
 * eta*H, eta = w_(2n); JensGroth: the n coefficients of (A*B - C)/Z.  log2n <= 27. */
#define G16_FLAVOUR_JENSGROTH 0u /* zkey_types.nim:11 */
#define G16_FLAVOUR_SNARKJS 1u   /* zkey_types.nim:12 */
int32_t g16_quotient(g16_ctx* ctx, const void* Az, const void* Bz, const void* Cz, uint32_t log2n, uint32_t flavour,
                     void* out);
int32_t g16_quotient_dev(g16_ctx* ctx, const void* d_Az, const void* d_Bz, const void* d_Cz, uint32_t log2n,
                         uint32_t flavour, void* d_out);

/* ---- proving key + proof: replaces generateProofWithMask (groth16/prover.nim:215-304) -----------------
 * g16_pkey_create uploads a ZKey (zkey_types.nim:54-59) once: the five ProverPoints arrays become
 * registered point sets, ZKey.coeffs becomes a device CSR for buildABC (prover.nim:56-73). */
typedef struct g16_pkey g16_pkey;
typedef struct {          /* one Coeff (zkey_types.nim:48-52) */
  uint32_t matrix;        /* 0 = MatrixA, 1 = MatrixB (MatrixC is rejected, as prover.nim:67 raises) */
  uint32_t row;           /* 0 .. domainSize-1 */
  uint32_t col;           /* 0 .. nvars-1 */
  uint32_t reserved;
  uint8_t value[32];      /* Fr, Montgomery */
} g16_coeff;
typedef struct {
  uint32_t nvars, npubs;  /* GrothHeader (zkey_types.nim:14-22) */
  uint32_t log2_domain;   /* logDomainSize */
  uint32_t flavour;       /* G16_FLAVOUR_* */
  const void *pointsA1, *pointsB1; /* nvars G1 points each          (zkey_types.nim:37-38) */
  const void* pointsB2;            /* nvars G2 points               (:39) */
  const void* pointsC1;            /* nvars - npubs - 1 G1 points   (:40) */
  const void* pointsH1;            /* domainSize G1 points          (:41) */
  const g16_coeff* coeffs;
  size_t ncoeffs;
  const void *alpha1, *beta1, *delta1; /* SpecPoints G1 (zkey_types.nim:24-31) */
  const void *beta2, *delta2;          /* SpecPoints G2 */
  /* multi-GPU: this key keeps only index range [N*i/count, N*(i+1)/count) of each point set -- the contiguous
   * chunks of msmMultiThreadedG1/G2 (msm.nim:105-115) with one chunk per GPU.  0/0 or 0/1 = whole key. */
  uint32_t shard_index, shard_count;
} g16_pkey_desc;
typedef struct { /* Proof (prover.nim:37-43) minus publicIO (= witness[0..npubs], which the caller already has) */
  uint8_t pi_a[64];
  uint8_t pi_b[128];
  uint8_t pi_c[64];
} g16_proof;
int32_t g16_pkey_create(g16_ctx* ctx, const g16_pkey_desc* desc, g16_pkey** out);
/* The same with ZKey.coeffs taken straight from the .zkey FILE: `section4` = section 4 as it lies on disk -- u32 count,
 * then count x { u32 matrix, u32 row, u32 col, 32-byte value in DOUBLE Montgomery form c R^2 } (files/zkey.nim:169-192;
 * the reference un-Montgomerys each value twice while loading, io.nim:134-139 unmarshalFrWTF) -- and desc->coeffs = NULL,
 * desc->ncoeffs = 0.  No host arithmetic and no 48-byte records: with the point sections of the file (plain Montgomery
 * bytes = the in-memory layout) and a raw .wtns witness (G16_SCALARS_STD) a memory-mapped .zkey / .wtns pair is proved
 * without being parsed.  c R^2 is also exactly what buildABC multiplies a raw .wtns value with: (c R^2) w / R = c w R. */
int32_t g16_pkey_create_zkey(g16_ctx* ctx, const g16_pkey_desc* desc, const void* section4, size_t section4_bytes,
                             g16_pkey** out);
void g16_pkey_destroy(g16_pkey* key);
/* points at infinity per ProverPoints array of this key (this shard): out[0..4] = A1, B1, B2, C1, H1; out[5] = wires
 * whose B1 AND B2 points are both (0,0); out[6] / out[7] = 1 if A1 / B1+B2 run on compacted entry lists (their own
 * arrangement of the witness without those wires; sets below the G16_INF_COMPACT threshold share one arrangement). */
int32_t g16_pkey_inf_counts(const g16_pkey* key, size_t out[8]);
/* witness: nvars Fr values (flags: G16_SCALARS_MONT for Nim seq[Fr], G16_SCALARS_STD for raw .wtns,
 * | G16_SCALARS_DEVICE); mask_r / mask_s: Fr Montgomery (Mask, prover.nim:210-213), NULL = zero
 * (generateProofWithTrivialMask, prover.nim:308-310). */
int32_t g16_prove(g16_ctx* ctx, const g16_pkey* key, const void* witness, uint32_t flags, const void* mask_r,
                  const void* mask_s, g16_proof* out);
/* The two halves of g16_prove, for a proof sharded over several GPUs (one g16_ctx + one sharded key per GPU):
 *  g16_prove_partials: buildABC + quotient (replicated) and the five MSMs over this key's index ranges;
 *      writes G16_PARTIALS_BYTES = 768 bytes: XYZZ accumulators A1 | B1 | B2 (256 B) | H1 | C1.  The record is
 *      opaque: accumulators are not canonical (the addition order inside a bucket is not fixed), and when the C1 and
 *      H1 sets share their bucket set the H1 slot holds H1 + C1 and the C1 slot infinity (pi_c needs only the sum,
 *      prover.nim:301-302).  Only g16_prove_combine gives records a meaning; it accepts any mix of them.
 *      flags: G16_SCALARS_MONT/STD | G16_SCALARS_DEVICE (witness in HBM) | G16_OUT_DEVICE (output in HBM).
 *  g16_prove_combine: `count` gathered records (rank order; e.g. from an RCCL all-gather) are summed per MSM
 *      -- the `res += sync pending[k]` of msm.nim:117-119 across GPUs -- and the mask algebra of
 *      prover.nim:279-302 yields the proof.  flags: G16_SCALARS_DEVICE if `partials` is a device pointer. */
#define G16_PARTIALS_BYTES 768
int32_t g16_prove_partials(g16_ctx* ctx, const g16_pkey* key, const void* witness, uint32_t flags, void* out_partials);
int32_t g16_prove_combine(g16_ctx* ctx, const g16_pkey* key, const void* partials, size_t count, uint32_t flags,
                          const void* mask_r, const void* mask_s, g16_proof* out);
/* Sharded proof with a TASK-PARALLEL quotient (snarkjs-flavour keys): the three coset pipelines that
 * computeSnarkjsScalarCoeffs runs as three Taskpool tasks (prover.nim:167-169) live on three different ranks instead
 * of being replicated on all of them.
 *  g16_prove_partials_begin: uploads the witness, launches this key's four witness MSMs (they keep running after
 *      the call returns) and computes the coset vectors named by task_mask (bit 0: A, 1: B, 2: C) -- buildABC +
 *      shiftEvalDomain (prover.nim:56-73, 109-113) -- into d_task_out: domainSize Fr per set bit, in ascending bit
 *      order, device memory.  flags as for g16_prove_partials.  task_mask = 0: this rank owns no pipeline.
 *  (caller: every rank receives the [h_lo, h_hi) slices of the three coset vectors, h = domainSize * rank / count,
 *      from the ranks that own them -- three scatters; 32 * domainSize / count bytes per slice)
 *  g16_prove_partials_end: forms this rank's H scalars A1*B1 - C1 from the received slices (device pointers,
 *      h_hi - h_lo Fr each; prover.nim:175-176), runs the H MSM over them, joins the witness MSMs and writes the
 *      768-byte record exactly like g16_prove_partials.  Then g16_prove_combine as usual.
 *  Between _begin and _end the witness MSMs are in flight on this context's workspaces.  Only g16_ctx_synchronize
 *  and the g16_profile_* calls leave them alone; ANY other call on the context first waits for them and cancels
 *  the pending proof (a later _end returns G16_EINVAL) -- so a caller whose exchange failed may simply go on, and
 *  one that wants to overlap two sharded proofs uses two contexts (nim_groth16_amd/distributed.py does).
 *  flags for _begin additionally: G16_NO_HOST_SYNC -- return without waiting for the coset vectors; the caller
 *  orders its exchange behind them through the context's stream (g16_ctx_set_stream: e.g. torch's current stream,
 *  on which the collective is then enqueued).
 *  G16_NO_HOST_SYNC, the rules: (1) _begin honours it wherever the witness lives.  A HOST witness must then be in
 *  pinned memory (otherwise the copy blocks anyway) and must stay untouched until the context's stream has passed the
 *  copy -- e.g. until the matching _end or g16_prove_combine has returned, or an event recorded on that stream after
 *  _begin has completed; the library does not wait for it.  (2) g16_prove_partials and _end honour it only together
 *  with G16_OUT_DEVICE (a record written to host memory is not complete in stream order for the host); without
 *  G16_OUT_DEVICE the flag is ignored and the call waits as usual. */
int32_t g16_prove_partials_begin(g16_ctx* ctx, const g16_pkey* key, const void* witness, uint32_t flags,
                                 uint32_t task_mask, void* d_task_out);
int32_t g16_prove_partials_end(g16_ctx* ctx, const g16_pkey* key, const void* d_a1_slice, const void* d_b1_slice,
                               const void* d_c1_slice, uint32_t flags, void* out_partials);
/* buildABC alone (prover.nim:56-73): out_abc = Az | Bz | Cz, 3 * domainSize Fr (Montgomery), host memory */
int32_t g16_build_abc(g16_ctx* ctx, const g16_pkey* key, const void* witness, uint32_t flags, void* out_abc);
/* shape of the key's A / B matrices as buildABC sees them (ZKey.coeffs, zkey_types.nim:48-59; files/zkey.nim:169-192):
 * out[0] = ncoeffs; out[1] = distinct coefficient values if the key runs on a value dictionary (entries are then
 * 8 bytes: column + value index), else 0 (36 bytes: column + value); out[2 + b], b = 0..8 = rows of A or of B (2 per
 * domain row) by their number of terms L: L <= 1, L = 2, L <= 4 (one lane each), then L <= 8, 16, 32, 64, 128 and
 * longer: groups of 2, 4, 8, 16, 32, 64 lanes with four terms per lane */
int32_t g16_pkey_abc_info(const g16_pkey* key, size_t out[11]);
/* y = M x over Fr (Montgomery in and out) for a sparse M given as nnz triplets (row[i], col[i], val[i] = 32 bytes);
 * entries of one row add up.  The same row-balanced kernel as buildABC.  Replaces the sparse column dot products of
 * the fake setup -- `for each coefficient: taus[wire] += value * L_row(tau)` (fake_setup.nim:159-187, 254-256): pass
 * row = wire, col = constraint, x = the Lagrange values.  Host pointers; x: ncols elements, y: nrows elements. */
int32_t g16_spmv_fr(g16_ctx* ctx, const uint32_t* row, const uint32_t* col, const void* val, size_t nnz, const void* x,
                    size_t ncols, size_t nrows, void* y);

/* ---- device group: one proof over several GPUs from ONE host process -------------------------------------------------
 * The reference is one process whose MSMs run as Taskpool tasks over contiguous index ranges, partial sums added in task
 * order (groth16/bn128/msm.nim:96-122), and whose three coset pipelines are three tasks (prover.nim:165-173).  A group is
 * that shape with GPUs in place of threads: member g = a context on devices[g] + shard g of ndev of the key, one host
 * thread per member INSIDE the library; per proof the members exchange 3 x 32 n / ndev bytes of coset slices (peer copies
 * between their HBMs) and member 0 adds the ndev 768-byte records in member order.  The caller needs no Python, no
 * torch.distributed and no RCCL; it calls g16_group_prove where it called generateProofWithMask.
 *   devices: HIP device ordinals, one per member (a device may repeat: several members then share it)
 *   g16_group_pkey_create: desc = the WHOLE key (shard_count 0 or 1); member g keeps index range g of every point set
 *   g16_group_prove: witness = nvars Fr in HOST memory (flags: G16_SCALARS_MONT or G16_SCALARS_STD), masks as g16_prove;
 *       the proof is bit-identical to g16_prove's on the unsharded key
 * A group is used by one host thread at a time; its keys may be destroyed before or after it.  (Python ranks over RCCL: nim_groth16_amd/distributed.py does the same
 * exchange with one process per GPU.) */
typedef struct g16_group g16_group;
typedef struct g16_group_pkey g16_group_pkey;
int32_t g16_group_create(const int32_t* devices, int32_t ndev, g16_group** out);
void g16_group_destroy(g16_group* group);
int32_t g16_group_size(const g16_group* group);
const char* g16_group_last_error(const g16_group* group);
int32_t g16_group_pkey_create(g16_group* group, const g16_pkey_desc* desc, g16_group_pkey** out);
void g16_group_pkey_destroy(g16_group_pkey* key);
int32_t g16_group_prove(g16_group* group, const g16_group_pkey* key, const void* witness, uint32_t flags,
                        const void* mask_r, const void* mask_s, g16_proof* out);

/* ---- verifier ------------------------------------------------------------------------
 * Replaces verifyProof (groth16/verifier.nim:31-52), extractVKey / VKey (groth16/zkey_types.nim:62-73) and the
 * `pairing` of bn128/curves.nim:218-221, on the device, batched over proofs.
 *
 * g16_vkey_desc: npubs = number of public inputs; pointsIC holds npubs + 1 G1 points (the first one belongs to
 * the constant 1).  alpha1: 64 B; beta2, gamma2, delta2: 128 B each (Montgomery affine, as everywhere).
 * g16_verify: `public_io` = count x (npubs + 1) scalars, each row starting with the constant 1 exactly like
 * Proof.publicIO (prover.nim:238-240); flags: G16_SCALARS_MONT / G16_SCALARS_STD for the scalars, and
 * G16_VERIFY_SUBGROUP to also require [r]pi_b = infinity (the reference only asserts the curve equations).
 * status[j]: 1 = proof j verifies, 0 = pairing equation fails, -1 / -2 / -3 = pi_a / pi_b / pi_c is not on its
 * curve (the reference's three asserts), -4 = pi_b is not in the order-r subgroup, -5 = a proof coordinate is not
 * the canonical residue (limbs >= p), -6 = a public input is not canonical (>= r): each proof / input has exactly
 * one accepted byte encoding.  (The smallest code wins when several apply.)
 * g16_pairing: out_gt[i] = e(P_i, Q_i) as 6 x Fp2 = 384 bytes, coefficient k of w^k in
 * Fp12 = Fp2[w]/(w^6 - (9+u)) (the tower of files/export_sage.nim:84-97 flattened), Montgomery form;
 * e is the ate pairing f_{t-1,Q}(P)^((p^12-1)/r).  NOT byte-compatible with the GT values of the reference's
 * `pairing` (constantine's optimal ate, curves.nim:218-221): the two differ by a fixed exponent, so products /
 * equality-to-one tests (all verifyProof needs) agree, the 384 bytes do not. */
typedef struct g16_vkey g16_vkey;
typedef struct {
  uint32_t npubs;
  const void* alpha1;
  const void* beta2;
  const void* gamma2;
  const void* delta2;
  const void* pointsIC;
} g16_vkey_desc;
#define G16_VERIFY_SUBGROUP 16u
#define G16_GT_BYTES 384
int32_t g16_vkey_create(g16_ctx* ctx, const g16_vkey_desc* desc, g16_vkey** out);
void g16_vkey_destroy(g16_vkey* key);
int32_t g16_verify(g16_ctx* ctx, const g16_vkey* key, const g16_proof* proofs, const void* public_io, uint32_t flags,
                   size_t count, int32_t* status);
int32_t g16_pairing(g16_ctx* ctx, const void* g1_points, const void* g2_points, size_t n, void* out_gt);

/* ---- profiling ---------------------------------------------------------------------------------- */
/* on = 1: every kernel launch is bracketed by HIP events on the stream it is launched on; on = 2: only the
 * bucket-accumulation kernels (cheap enough to leave on inside a timed region); on = 0: off */
int32_t g16_profile_enable(g16_ctx* ctx, int32_t on);
int32_t g16_profile_reset(g16_ctx* ctx);
/* writes a JSON object {"kernel": {"calls": k, "total_ms": t}, ...} into buf (NUL-terminated) */
int32_t g16_profile_report(g16_ctx* ctx, char* buf, size_t buflen);
/* The shader clock (GHz) the chip sustained while the bucket-accumulation kernels launched since the last call ran
 * (profiling on): sum of s_memtime deltas / sum of s_memrealtime deltas over thread 0 of every workgroup.  0 if none
 * ran.  Waits for the context's work; resets the sums. */
int32_t g16_profile_clock(g16_ctx* ctx, double* ghz);

#ifdef __cplusplus
}
#endif
#endif /* G16HIP_H */
