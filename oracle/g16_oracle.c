/*
 * ORACLE (test infrastructure only) -- plain-C CPU restatement of the nim-groth16 MSM / NTT hot path.
 *
 * NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * liboracle.so; nim_groth16_amd/ never does.  It shares no source with the HIP library: 4 x u64 limbs
 * and unsigned __int128 CIOS Montgomery here, 8 x u32 product-scanning there; Jacobian coordinates and
 * unsigned windows here, XYZZ and signed windows there.
 *
 * Structure follows the reference so that it can stand in as the CPU baseline ("port"):
 *   orc_msm_g1/g2        msmMultiThreadedG1/G2      groth16/bn128/msm.nim:89-158 (thread rule :98-100,
 *                                                   contiguous chunks :105-115, partials summed from inf :117-119)
 *   msm_chunk_*          msmConstantineG1/G2        msm.nim:35-83 (Pippenger bucket method per chunk)
 *   orc_msm_naive_g1/g2  msmNaiveG1/G2              msm.nim:162-198 (the in-tree definition)
 *   orc_ntt_forward      forwardNTT(+_worker)       groth16/math/ntt.nim:17-77   (recursion transliterated)
 *   orc_ntt_inverse      inverseNTT(+_worker)       ntt.nim:97-161
 *   orc_quotient_snarkjs computeSnarkjsScalarCoeffs groth16/prover.nim:158-181 (3 parallel shiftEvalDomain tasks)
 *   orc_fixed_base_g1/g2 `y ** gen1` / `y ** gen2`  groth16/fake_setup.nim:258-261 (input generator)
 *
 * Parity status: pinned against oracle/bn254_ref.py (pure-Python ints) in tests/test_oracle_c.py, which in
 * turn is pinned by the reference's embedded constants and its toy-circuit prove->verify fixture.  The
 * reference itself cannot be built here (Nim + un-vendored constantine): no reference-executed vectors exist.
 *
 * All byte layouts = the C-ABI layouts: Fr/Fp 32 B little-endian Montgomery (R = 2^256), G1 64 B, G2 128 B,
 * infinity = all-zero (groth16/bn128/curves.nim:49-50).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static long sysconf_nproc(void) { long n = sysconf(_SC_NPROCESSORS_ONLN); return n > 0 ? n : 1; }

typedef uint64_t u64;
typedef unsigned __int128 u128;

typedef struct { u64 v[4]; } fe;                 /* field element (Fp or Fr), Montgomery */
typedef struct { u64 p[4]; u64 inv; fe one; fe r2; } modulus;

static const modulus MP = { /* fields.nim:36 */
  {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
  0x87d20782e4866389ULL,
  {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}},
  {{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}}};
static const modulus MR = { /* fields.nim:37 */
  {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
  0xc2e1f593efffffffULL,
  {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}},
  {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}}};

static inline int fe_is_zero(const fe* a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static inline int fe_eq(const fe* a, const fe* b) {
  return ((a->v[0] ^ b->v[0]) | (a->v[1] ^ b->v[1]) | (a->v[2] ^ b->v[2]) | (a->v[3] ^ b->v[3])) == 0;
}
static inline int geq_p(const u64 a[4], const modulus* M) {
  for (int i = 3; i >= 0; --i) {
    if (a[i] > M->p[i]) return 1;
    if (a[i] < M->p[i]) return 0;
  }
  return 1;
}
static inline void sub_p(u64 a[4], const modulus* M) {
  u128 b = 0;
  for (int i = 0; i < 4; ++i) {
    u128 t = (u128)a[i] - M->p[i] - (u64)b;
    a[i] = (u64)t;
    b = (t >> 64) & 1;
  }
}
static inline void fe_add(fe* r, const fe* a, const fe* b, const modulus* M) {
  u128 c = 0;
  for (int i = 0; i < 4; ++i) {
    c += (u128)a->v[i] + b->v[i];
    r->v[i] = (u64)c;
    c >>= 64;
  }
  if (geq_p(r->v, M)) sub_p(r->v, M);
}
static inline void fe_sub(fe* r, const fe* a, const fe* b, const modulus* M) {
  u128 bw = 0;
  u64 t[4];
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a->v[i] - b->v[i] - (u64)bw;
    t[i] = (u64)d;
    bw = (d >> 64) & 1;
  }
  if (bw) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) {
      c += (u128)t[i] + M->p[i];
      t[i] = (u64)c;
      c >>= 64;
    }
  }
  memcpy(r->v, t, 32);
}
static inline void fe_neg(fe* r, const fe* a, const modulus* M) {
  fe z = {{0, 0, 0, 0}};
  fe_sub(r, &z, a, M);
}
/* CIOS Montgomery product */
static inline void fe_mul(fe* r, const fe* a, const fe* b, const modulus* M) {
  u64 t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) {
      c += (u128)a->v[j] * b->v[i] + t[j];
      t[j] = (u64)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (u64)c;
    t[5] = (u64)(c >> 64);
    u64 m = t[0] * M->inv;
    c = (u128)m * M->p[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; ++j) {
      c += (u128)m * M->p[j] + t[j];
      t[j - 1] = (u64)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (u64)c;
    t[4] = t[5] + (u64)(c >> 64);
  }
  if (t[4] || geq_p(t, M)) sub_p(t, M);
  memcpy(r->v, t, 32);
}
static inline void fe_sqr(fe* r, const fe* a, const modulus* M) { fe_mul(r, a, a, M); }
static void fe_from_mont(fe* r, const fe* a, const modulus* M) {
  fe one = {{1, 0, 0, 0}};
  fe_mul(r, a, &one, M);
}
static void fe_to_mont(fe* r, const fe* a, const modulus* M) { fe_mul(r, a, &M->r2, M); }
static void fe_pow(fe* r, const fe* a, const u64 e[4], const modulus* M) {
  fe acc = M->one, base = *a;
  for (int i = 0; i < 256; ++i) {
    if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(&acc, &acc, &base, M);
    fe_sqr(&base, &base, M);
  }
  *r = acc;
}
static void fe_inv(fe* r, const fe* a, const modulus* M) {
  u64 e[4] = {M->p[0] - 2, M->p[1], M->p[2], M->p[3]};
  fe_pow(r, a, e, M);
}
static void fe_div2(fe* r, const fe* a, const modulus* M) { /* constantine div2, ntt.nim:111-112,121 */
  u64 t[5];
  u128 c = 0;
  int odd = a->v[0] & 1;
  for (int i = 0; i < 4; ++i) {
    c += (u128)a->v[i] + (odd ? M->p[i] : 0);
    t[i] = (u64)c;
    c >>= 64;
  }
  t[4] = (u64)c;
  for (int i = 0; i < 4; ++i) r->v[i] = (t[i] >> 1) | (t[i + 1] << 63);
}

/* ---- Fp2 (fields.nim:27-32) ---- */
typedef struct { fe c0, c1; } fe2;
static inline int fe2_is_zero(const fe2* a) { return fe_is_zero(&a->c0) && fe_is_zero(&a->c1); }
static inline int fe2_eq(const fe2* a, const fe2* b) { return fe_eq(&a->c0, &b->c0) && fe_eq(&a->c1, &b->c1); }
static inline void fe2_add(fe2* r, const fe2* a, const fe2* b) { fe_add(&r->c0, &a->c0, &b->c0, &MP); fe_add(&r->c1, &a->c1, &b->c1, &MP); }
static inline void fe2_sub(fe2* r, const fe2* a, const fe2* b) { fe_sub(&r->c0, &a->c0, &b->c0, &MP); fe_sub(&r->c1, &a->c1, &b->c1, &MP); }
static inline void fe2_neg(fe2* r, const fe2* a) { fe_neg(&r->c0, &a->c0, &MP); fe_neg(&r->c1, &a->c1, &MP); }
static inline void fe2_mul(fe2* r, const fe2* a, const fe2* b) { /* schoolbook: 4 products */
  fe t0, t1, t2, t3;
  fe_mul(&t0, &a->c0, &b->c0, &MP);
  fe_mul(&t1, &a->c1, &b->c1, &MP);
  fe_mul(&t2, &a->c0, &b->c1, &MP);
  fe_mul(&t3, &a->c1, &b->c0, &MP);
  fe_sub(&r->c0, &t0, &t1, &MP);
  fe_add(&r->c1, &t2, &t3, &MP);
}
static inline void fe2_sqr(fe2* r, const fe2* a) { fe2_mul(r, a, a); }
static void fe2_inv(fe2* r, const fe2* a) {
  fe n, t, d;
  fe_sqr(&n, &a->c0, &MP);
  fe_sqr(&t, &a->c1, &MP);
  fe_add(&n, &n, &t, &MP);
  fe_inv(&d, &n, &MP);
  fe_mul(&r->c0, &a->c0, &d, &MP);
  fe_mul(&t, &a->c1, &d, &MP);
  fe_neg(&r->c1, &t, &MP);
}

/* ---- curve code, instantiated for Fp (G1) and Fp2 (G2) ---- */
#define FE fe
#define FN(x) g1_##x
#define F_ADD(r, a, b) fe_add(r, a, b, &MP)
#define F_SUB(r, a, b) fe_sub(r, a, b, &MP)
#define F_MUL(r, a, b) fe_mul(r, a, b, &MP)
#define F_SQR(r, a) fe_sqr(r, a, &MP)
#define F_NEG(r, a) fe_neg(r, a, &MP)
#define F_INV(r, a) fe_inv(r, a, &MP)
#define F_ISZERO(a) fe_is_zero(a)
#define F_EQ(a, b) fe_eq(a, b)
#define F_SETONE(r) (*(r) = MP.one)
#include "g16_oracle_ec.inc"
#undef FE
#undef FN
#undef F_ADD
#undef F_SUB
#undef F_MUL
#undef F_SQR
#undef F_NEG
#undef F_INV
#undef F_ISZERO
#undef F_EQ
#undef F_SETONE

#define FE fe2
#define FN(x) g2_##x
#define F_ADD(r, a, b) fe2_add(r, a, b)
#define F_SUB(r, a, b) fe2_sub(r, a, b)
#define F_MUL(r, a, b) fe2_mul(r, a, b)
#define F_SQR(r, a) fe2_sqr(r, a)
#define F_NEG(r, a) fe2_neg(r, a)
#define F_INV(r, a) fe2_inv(r, a)
#define F_ISZERO(a) fe2_is_zero(a)
#define F_EQ(a, b) fe2_eq(a, b)
#define F_SETONE(r) do { (r)->c0 = MP.one; memset(&(r)->c1, 0, sizeof(fe)); } while (0)
#include "g16_oracle_ec.inc"


/* group generators (curves.nim:112-124), standard form -> Montgomery */
static void g1_generator(g1_aff* g) {
  g->x = MP.one;
  fe_add(&g->y, &MP.one, &MP.one, &MP);
}
static void g2_generator(g2_aff* g) {
  static const fe XI = {{0xbde23fab1f149701ULL, 0x98aa68a570acf5b0ULL, 0x7040f46655e3808fULL, 0x1adcd0ed10df9cb8ULL}}, XU = {{0xfa15d21c1c13b23bULL, 0xfbfbe620f7f31269ULL, 0xc3cd2a1d0a3a82e6ULL, 0x09e847e9f05a6082ULL}};
  static const fe YI = {{0xe6f915250b7f6fc8ULL, 0x1c7cdf52dbfc4cbeULL, 0x1f7ca7aa19d4fcfdULL, 0x056c01168a531946ULL}}, YU = {{0xaaa86456a623235cULL, 0xf553b878fc3c0dadULL, 0xf5f401329f30895dULL, 0x0efe500a2d02dd77ULL}};
  fe_to_mont(&g->x.c0, &XI, &MP); fe_to_mont(&g->x.c1, &XU, &MP);
  fe_to_mont(&g->y.c0, &YI, &MP); fe_to_mont(&g->y.c1, &YU, &MP);
}

/* ================= exported API ================= */
int orc_cores(void) { return (int)sysconf_nproc(); }

/* field ops on byte buffers (tests): field 0 = Fp, 1 = Fr ; op 0 add 1 sub 2 mul 7 inv 8 from_mont 9 to_mont 6 div2 */
void orc_field_op(int field, int op, const void* a, const void* b, void* r) {
  const modulus* M = field ? &MR : &MP;
  fe x, y, z;
  memcpy(&x, a, 32);
  memcpy(&y, b, 32);
  switch (op) {
    case 0: fe_add(&z, &x, &y, M); break;
    case 1: fe_sub(&z, &x, &y, M); break;
    case 2: fe_mul(&z, &x, &y, M); break;
    case 6: fe_div2(&z, &x, M); break;
    case 7: fe_inv(&z, &x, M); break;
    case 8: fe_from_mont(&z, &x, M); break;
    default: fe_to_mont(&z, &x, M); break;
  }
  memcpy(r, &z, 32);
}

void orc_msm_naive_g1(const void* scalars, int mont, const void* points, size_t n, void* out) { g1_msm_naive(scalars, mont, points, n, out); }
void orc_msm_naive_g2(const void* scalars, int mont, const void* points, size_t n, void* out) { g2_msm_naive(scalars, mont, points, n, out); }
void orc_msm_g1(int nthreads_hint, const void* scalars, int mont, const void* points, size_t n, void* out) { g1_msm_mt(nthreads_hint, scalars, mont, points, n, out); }
void orc_msm_g2(int nthreads_hint, const void* scalars, int mont, const void* points, size_t n, void* out) { g2_msm_mt(nthreads_hint, scalars, mont, points, n, out); }
void orc_fixed_base_g1(int nthreads, const void* scalars, int mont, size_t n, void* out_points) { g1_fixed_base(nthreads, scalars, mont, n, out_points); }
void orc_fixed_base_g2(int nthreads, const void* scalars, int mont, size_t n, void* out_points) { g2_fixed_base(nthreads, scalars, mont, n, out_points); }
void orc_g1_add(const void* a, const void* b, void* r) { g1_add_affine_bytes(a, b, r); }
void orc_g2_add(const void* a, const void* b, void* r) { g2_add_affine_bytes(a, b, r); }
void orc_g1_mul(const void* scalar, int mont, const void* p, void* r) { g1_mul_bytes(scalar, mont, p, r); }
void orc_g2_mul(const void* scalar, int mont, const void* p, void* r) { g2_mul_bytes(scalar, mont, p, r); }

/* ---- domain + NTT (math/domain.nim, math/ntt.nim) ---- */
static const fe GEN28_STD = {{0x9bd61b6e725b19f0ULL, 0x402d111e41112ed4ULL, 0x00e0a7eb8ef62abcULL, 0x2a3c09f0a58a7e85ULL}}; /* domain.nim:26 */
static const fe ONE_HALF_STD = {{0xa1f0fac9f8000001ULL, 0x9419f4243cdcb848ULL, 0xdc2822db40c0ac2eULL, 0x183227397098d014ULL}}; /* ntt.nim:95 */

static void domain_gen(fe* g, int log2n) { /* domain.nim:31-33 */
  fe w;
  fe_to_mont(&w, &GEN28_STD, &MR);
  for (int i = log2n; i < 28; ++i) fe_sqr(&w, &w, &MR);
  *g = w;
}
void orc_domain_gen(int log2n, void* out) {
  fe g;
  domain_gen(&g, log2n);
  memcpy(out, &g, 32);
}

static void fwd_worker(int m, size_t src_stride, const fe* gpows, const fe* src, size_t src_ofs, fe* buf, size_t buf_ofs,
                       fe* tgt, size_t tgt_ofs) { /* ntt.nim:17-50 */
  if (m == 0) {
    tgt[tgt_ofs] = src[src_ofs];
  } else if (m == 1) {
    fe a = src[src_ofs], b = src[src_ofs + src_stride];
    fe_add(&tgt[tgt_ofs], &a, &b, &MR);
    fe_sub(&tgt[tgt_ofs + 1], &a, &b, &MR);
  } else {
    size_t N = (size_t)1 << m, half = N >> 1;
    fwd_worker(m - 1, src_stride << 1, gpows, src, src_ofs, buf, buf_ofs + N, buf, buf_ofs);
    fwd_worker(m - 1, src_stride << 1, gpows, src, src_ofs + src_stride, buf, buf_ofs + N, buf, buf_ofs + half);
    for (size_t j = 0; j < half; ++j) {
      fe y, e = buf[buf_ofs + j];
      fe_mul(&y, &gpows[j * src_stride], &buf[buf_ofs + j + half], &MR);
      fe_add(&tgt[tgt_ofs + j], &e, &y, &MR);
      fe_sub(&tgt[tgt_ofs + j + half], &e, &y, &MR);
    }
  }
}
int orc_ntt_forward(const void* src_, void* dst_, int log2n) { /* ntt.nim:55-77 */
  size_t n = (size_t)1 << log2n, half = n >> 1;
  const fe* src = (const fe*)src_;
  fe* buf = (fe*)malloc(2 * n * sizeof(fe));
  fe* tgt = (fe*)malloc(n * sizeof(fe));
  fe* gp = (fe*)malloc((half ? half : 1) * sizeof(fe));
  if (!buf || !tgt || !gp) return -1;
  fe g, x = MR.one;
  domain_gen(&g, log2n);
  for (size_t i = 0; i < half; ++i) {
    gp[i] = x;
    fe_mul(&x, &x, &g, &MR);
  }
  fwd_worker(log2n, 1, gp, src, 0, buf, 0, tgt, 0);
  memcpy(dst_, tgt, n * sizeof(fe));
  free(buf); free(tgt); free(gp);
  return 0;
}
static void inv_worker(int m, size_t tgt_stride, const fe* gpows, const fe* src, size_t src_ofs, fe* buf, size_t buf_ofs,
                       fe* tgt, size_t tgt_ofs) { /* ntt.nim:97-134 */
  if (m == 0) {
    tgt[tgt_ofs] = src[src_ofs];
  } else if (m == 1) {
    fe a = src[src_ofs], b = src[src_ofs + 1], s, d;
    fe_add(&s, &a, &b, &MR);
    fe_sub(&d, &a, &b, &MR);
    fe_div2(&tgt[tgt_ofs], &s, &MR);
    fe_div2(&tgt[tgt_ofs + tgt_stride], &d, &MR);
  } else {
    size_t N = (size_t)1 << m, half = N >> 1;
    for (size_t j = 0; j < half; ++j) {
      fe a = src[src_ofs + j], b = src[src_ofs + j + half], s, d;
      fe_add(&s, &a, &b, &MR);
      fe_sub(&d, &a, &b, &MR);
      fe_div2(&buf[buf_ofs + j], &s, &MR);
      fe_mul(&buf[buf_ofs + j + half], &d, &gpows[j * tgt_stride], &MR);
    }
    inv_worker(m - 1, tgt_stride << 1, gpows, buf, buf_ofs, buf, buf_ofs + N, tgt, tgt_ofs);
    inv_worker(m - 1, tgt_stride << 1, gpows, buf, buf_ofs + half, buf, buf_ofs + N, tgt, tgt_ofs + tgt_stride);
  }
}
int orc_ntt_inverse(const void* src_, void* dst_, int log2n) { /* ntt.nim:139-161 */
  size_t n = (size_t)1 << log2n, half = n >> 1;
  const fe* src = (const fe*)src_;
  fe* buf = (fe*)malloc(2 * n * sizeof(fe));
  fe* tgt = (fe*)malloc(n * sizeof(fe));
  fe* gp = (fe*)malloc((half ? half : 1) * sizeof(fe));
  if (!buf || !tgt || !gp) return -1;
  fe g, ginv, x;
  domain_gen(&g, log2n);
  fe_inv(&ginv, &g, &MR);
  fe_to_mont(&x, &ONE_HALF_STD, &MR);
  for (size_t i = 0; i < half; ++i) {
    gp[i] = x;
    fe_mul(&x, &x, &ginv, &MR);
  }
  inv_worker(log2n, 1, gp, src, 0, buf, 0, tgt, 0);
  memcpy(dst_, tgt, n * sizeof(fe));
  free(buf); free(tgt); free(gp);
  return 0;
}

/* shiftEvalDomain (prover.nim:109-113) : values -> iNTT -> * eta^i -> NTT, in place */
typedef struct { fe* v; int log2n; int rc; } shift_task;
static void* shift_eval_thread(void* arg) {
  shift_task* t = (shift_task*)arg;
  size_t n = (size_t)1 << t->log2n;
  fe* tmp = (fe*)malloc(n * sizeof(fe));
  if (!tmp) { t->rc = -1; return NULL; }
  t->rc = orc_ntt_inverse(t->v, tmp, t->log2n);
  fe eta, pw = MR.one;
  domain_gen(&eta, t->log2n + 1); /* prover.nim:163 eta = createDomain(2n).domainGen */
  for (size_t i = 0; i < n; ++i) {  /* multiplyByPowers, prover.nim:96-106 */
    fe_mul(&tmp[i], &tmp[i], &pw, &MR);
    fe_mul(&pw, &pw, &eta, &MR);
  }
  if (!t->rc) t->rc = orc_ntt_forward(tmp, t->v, t->log2n);
  free(tmp);
  return NULL;
}
/* the three shiftEvalDomain tasks both quotient flavours start with (prover.nim:132-138, 167-173); v[k] malloc'ed */
static int shift_eval_abc(const void* Az, const void* Bz, const void* Cz, int log2n, int parallel, fe* v[3]);

/* computeQuotientPointwise (prover.nim:118-148), JensGroth flavour: coefficients of Q = (A*B - C) / Z */
int orc_quotient_jensgroth(const void* Az, const void* Bz, const void* Cz, int log2n, void* out, int parallel) {
  size_t n = (size_t)1 << log2n;
  fe* v[3];
  int rc = shift_eval_abc(Az, Bz, Cz, log2n, parallel, v);
  if (rc) return rc;
  /* invZ1 = 1 / (eta^n - 1), eta = createDomain(2n).domainGen   (prover.nim:127-128) */
  fe eta, etan, zm1, invZ1, eta_inv;
  domain_gen(&eta, log2n + 1);
  etan = eta;
  for (int k = 0; k < log2n; ++k) fe_sqr(&etan, &etan, &MR);   /* smallPowFr(eta, n), n = 2^log2n */
  fe_sub(&zm1, &etan, &MR.one, &MR);
  fe_inv(&invZ1, &zm1, &MR);
  fe* ys = (fe*)malloc(n * sizeof(fe));
  if (!ys) return -1;
  for (size_t j = 0; j < n; ++j) {   /* prover.nim:141 */
    fe p, d;
    fe_mul(&p, &v[0][j], &v[1][j], &MR);
    fe_sub(&d, &p, &v[2][j], &MR);
    fe_mul(&ys[j], &d, &invZ1, &MR);
  }
  fe* o = (fe*)out;
  rc = orc_ntt_inverse(ys, o, log2n);   /* prover.nim:142 */
  fe_inv(&eta_inv, &eta, &MR);
  fe pw = MR.one;
  for (size_t i = 0; i < n; ++i) {      /* multiplyByPowers(Q1.coeffs, invFr(eta)), prover.nim:143 */
    fe_mul(&o[i], &o[i], &pw, &MR);
    fe_mul(&pw, &pw, &eta_inv, &MR);
  }
  free(ys);
  for (int k = 0; k < 3; ++k) free(v[k]);
  return rc;
}

static int shift_eval_abc(const void* Az, const void* Bz, const void* Cz, int log2n, int parallel, fe* v[3]) {
  size_t n = (size_t)1 << log2n;
  const void* in[3] = {Az, Bz, Cz};
  shift_task t[3];
  pthread_t th[3];
  for (int k = 0; k < 3; ++k) {
    v[k] = (fe*)malloc(n * sizeof(fe));
    if (!v[k]) return -1;
    memcpy(v[k], in[k], n * sizeof(fe));
    t[k].v = v[k];
    t[k].log2n = log2n;
    t[k].rc = 0;
  }
  if (parallel) {
    for (int k = 0; k < 3; ++k) pthread_create(&th[k], NULL, shift_eval_thread, &t[k]);
    for (int k = 0; k < 3; ++k) pthread_join(th[k], NULL);
  } else {
    for (int k = 0; k < 3; ++k) shift_eval_thread(&t[k]);
  }
  return t[0].rc | t[1].rc | t[2].rc;
}

/* computeSnarkjsScalarCoeffs (prover.nim:158-181): 3 tasks, then A1*B1 - C1 */
int orc_quotient_snarkjs(const void* Az, const void* Bz, const void* Cz, int log2n, void* out, int parallel) {
  size_t n = (size_t)1 << log2n;
  fe* v[3];
  const void* in[3] = {Az, Bz, Cz};
  shift_task t[3];
  pthread_t th[3];
  for (int k = 0; k < 3; ++k) {
    v[k] = (fe*)malloc(n * sizeof(fe));
    if (!v[k]) return -1;
    memcpy(v[k], in[k], n * sizeof(fe));
    t[k].v = v[k];
    t[k].log2n = log2n;
    t[k].rc = 0;
  }
  if (parallel) {
    for (int k = 0; k < 3; ++k) pthread_create(&th[k], NULL, shift_eval_thread, &t[k]);
    for (int k = 0; k < 3; ++k) pthread_join(th[k], NULL);
  } else {
    for (int k = 0; k < 3; ++k) shift_eval_thread(&t[k]);
  }
  fe* o = (fe*)out;
  for (size_t j = 0; j < n; ++j) {
    fe p;
    fe_mul(&p, &v[0][j], &v[1][j], &MR);
    fe_sub(&o[j], &p, &v[2][j], &MR);
  }
  for (int k = 0; k < 3; ++k) free(v[k]);
  return t[0].rc | t[1].rc | t[2].rc;
}

/* buildABC (prover.nim:56-73).  coeffs: packed 48-byte entries {u32 matrix, u32 row, u32 col, u32 pad, Fr value
 * (Montgomery)} -- the g16_coeff layout of include/g16hip.h; out = Az | Bz | Cz (3n Fr). */
int orc_build_abc(const void* coeffs, size_t ncoeffs, const void* witness, int log2n, void* out) {
  size_t n = (size_t)1 << log2n;
  fe* A = (fe*)out;
  fe* B = A + n;
  fe* C = B + n;
  memset(out, 0, 3 * n * sizeof(fe));
  const unsigned char* p = (const unsigned char*)coeffs;
  const fe* w = (const fe*)witness;
  for (size_t e = 0; e < ncoeffs; ++e, p += 48) {
    uint32_t m, row, col;
    fe v, t;
    memcpy(&m, p, 4); memcpy(&row, p + 4, 4); memcpy(&col, p + 8, 4); memcpy(&v, p + 16, 32);
    if (m > 1 || row >= n) return -1;   /* MatrixC raises in the reference (prover.nim:67) */
    fe_mul(&t, &v, &w[col], &MR);
    fe* dst = m == 0 ? &A[row] : &B[row];
    fe_add(dst, dst, &t, &MR);
  }
  for (size_t i = 0; i < n; ++i) fe_mul(&C[i], &A[i], &B[i], &MR);
  return 0;
}
