// g16prove -- native host side above the C ABI: the `-p` (prove) path of the reference CLI
// (cli/cli_main.nim:162-231: parseZKey, parseWitness, generateProof, exportProof, exportPublicIO, verifyProof)
// as a plain C++ program linked against libg16hip.so.  No Python, no torch.
//
//   g++ -O2 -std=c++17 -Iinclude tools/g16prove.cpp -Lnim_groth16_amd/csrc -lg16hip
//       -Wl,-rpath,$PWD/nim_groth16_amd/csrc -o g16prove
//   ./g16prove -z circuit.zkey -w witness.wtns -o proof.json -i public.json [-n] [-y] [-t]
//
// File formats restated from the reference (groth16/files/container.nim:6-20, zkey.nim:6-91, witness.nim:5-15):
// the point sections of a .zkey are little-endian Montgomery with R = 2^256 -- byte for byte the layout the
// library takes -- so they go from the memory map to the GPU unparsed; .wtns values are canonical little-endian
// and are passed with G16_SCALARS_STD.  Only the section-4 coefficients need arithmetic on the host (they are
// doubly Montgomery-encoded, bn128/io.nim:134-139): one Montgomery reduction each.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "g16hip.h"

namespace {

[[noreturn]] void die(const std::string& msg) {
  fprintf(stderr, "g16prove: %s\n", msg.c_str());
  exit(1);
}

// ---- 256-bit helpers on 4 x u64 little-endian limbs -----------------------------------------------------
using u128 = unsigned __int128;
struct U256 {
  uint64_t v[4];
};
const U256 PRIME_R = {{0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};
const U256 PRIME_P = {{0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};
constexpr uint64_t R_NINV = 0xc2e1f593efffffffull;  // -r^-1 mod 2^64
constexpr uint64_t P_NINV = 0x87d20782e4866389ull;  // -p^-1 mod 2^64

bool geq(const U256& a, const U256& b) {
  for (int i = 3; i >= 0; --i)
    if (a.v[i] != b.v[i]) return a.v[i] > b.v[i];
  return true;
}
U256 sub(const U256& a, const U256& b) {
  U256 r;
  u128 br = 0;
  for (int i = 0; i < 4; ++i) {
    u128 t = (u128)a.v[i] - b.v[i] - br;
    r.v[i] = (uint64_t)t;
    br = (t >> 64) & 1;
  }
  return r;
}
// x * 2^-256 mod m (one Montgomery reduction of a 256-bit value)
U256 mont_reduce(const U256& x, const U256& m, uint64_t ninv) {
  uint64_t t[9] = {x.v[0], x.v[1], x.v[2], x.v[3], 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    uint64_t q = t[i] * ninv;
    u128 c = 0;
    for (int j = 0; j < 4; ++j) {
      c += (u128)q * m.v[j] + t[i + j];
      t[i + j] = (uint64_t)c;
      c >>= 64;
    }
    for (int j = i + 4; c && j < 9; ++j) {
      c += t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
  }
  U256 r = {{t[4], t[5], t[6], t[7]}};
  if (t[8] || geq(r, m)) r = sub(r, m);
  return r;
}
U256 load(const uint8_t* p) {
  U256 r;
  memcpy(r.v, p, 32);
  return r;
}
std::string decimal(U256 x) {  // canonical value -> decimal string
  std::string s;
  bool zero = false;
  while (!zero) {
    u128 rem = 0;
    zero = true;
    for (int i = 3; i >= 0; --i) {
      u128 cur = (rem << 64) | x.v[i];
      x.v[i] = (uint64_t)(cur / 1000000000ull);
      rem = cur % 1000000000ull;
      if (x.v[i]) zero = false;
    }
    char buf[16];
    snprintf(buf, sizeof buf, zero ? "%llu" : "%09llu", (unsigned long long)rem);
    s = std::string(buf) + s;
  }
  return s;
}
std::string fp_dec(const uint8_t* mont) { return decimal(mont_reduce(load(mont), PRIME_P, P_NINV)); }

// ---- container (groth16/files/container.nim:75-93) over a read-only memory map ----------------------------
struct Section {
  const uint8_t* p;
  size_t len;
};
struct Container {
  std::map<uint32_t, Section> sec;
  Container(const char* path, const char magic[4], uint32_t version) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) die(std::string("cannot open ") + path);
    struct stat st;
    fstat(fd, &st);
    const uint8_t* b = (const uint8_t*)mmap(nullptr, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (b == MAP_FAILED) die("mmap failed");
    size_t n = st.st_size;
    if (n < 12 || memcmp(b, magic, 4)) die(std::string("not a `") + std::string(magic, 4) + "` file");
    uint32_t ver, nsec;
    memcpy(&ver, b + 4, 4);
    memcpy(&nsec, b + 8, 4);
    if (ver != version) die("unexpected container version");
    size_t pos = 12;
    for (uint32_t i = 0; i < nsec; ++i) {
      if (pos + 12 > n) die("truncated file");
      uint32_t id;
      uint64_t len;
      memcpy(&id, b + pos, 4);
      memcpy(&len, b + pos + 4, 8);
      pos += 12;
      if (pos + len > n) die("truncated section");
      if (!sec.count(id)) sec[id] = Section{b + pos, (size_t)len};
      pos += len;
    }
  }
  Section get(uint32_t id) const {
    auto it = sec.find(id);
    if (it == sec.end()) die("missing section " + std::to_string(id));
    return it->second;
  }
};
uint32_t u32(const uint8_t* p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}
void expect_prime(const uint8_t*& p, const U256& want, const char* what) {
  if (u32(p) != 32) die("expecting 256 bit primes");
  if (memcmp(p + 4, want.v, 32)) die(std::string("expecting the alt-bn128 curve (") + what + ")");
  p += 36;
}

void write_g1(FILE* f, const uint8_t* p) {  // export_json.nim:55-59
  fprintf(f, "    [ \"%s\"\n    , \"%s\"\n    , \"1\"\n    ]\n", fp_dec(p).c_str(), fp_dec(p + 32).c_str());
}
void write_fp2(FILE* f, const char* lead, const std::string& a, const std::string& b) {  // export_json.nim:48-53
  fprintf(f, "    %s [ \"%s\"\n      , \"%s\"\n      ]\n", lead, a.c_str(), b.c_str());
}
void write_g2(FILE* f, const uint8_t* p) {  // export_json.nim:61-65
  write_fp2(f, "[", fp_dec(p), fp_dec(p + 32));
  write_fp2(f, ",", fp_dec(p + 64), fp_dec(p + 96));
  write_fp2(f, ",", "1", "0");
  fprintf(f, "    ]\n");
}

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

int main(int argc, char** argv) {
  const char *zpath = nullptr, *wpath = nullptr, *opath = "proof.json", *ipath = "public.json";
  bool nomask = false, verify = false, timing = false;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() -> const char* {
      if (i + 1 >= argc) die("missing value after " + a);
      return argv[++i];
    };
    if (a == "-z" || a == "--zkey") zpath = next();
    else if (a == "-w" || a == "--wtns") wpath = next();
    else if (a == "-o" || a == "--output") opath = next();
    else if (a == "-i" || a == "--io") ipath = next();
    else if (a == "-n" || a == "--nomask") nomask = true;   // cli_main.nim -n
    else if (a == "-y" || a == "--verify") verify = true;   // cli_main.nim -y
    else if (a == "-t" || a == "--time") timing = true;     // cli_main.nim -t
    else die("unknown option " + a + "\nusage: g16prove -z circuit.zkey -w witness.wtns [-o proof.json] [-i public.json] [-n] [-y] [-t]");
  }
  if (!zpath || !wpath) die("usage: g16prove -z circuit.zkey -w witness.wtns [-o proof.json] [-i public.json] [-n] [-y] [-t]");

  const double t0 = now();
  // ---- .zkey (zkey.nim:104-224) ----
  Container zk(zpath, "zkey", 1);
  if (zk.get(1).len != 4 || u32(zk.get(1).p) != 1) die("expecting `.zkey` file for a Groth16 prover");
  Section s2 = zk.get(2);
  if (s2.len != 2 * 4 + 64 + 3 * 4 + 3 * 64 + 3 * 128) die("unexpected header section length");
  const uint8_t* p = s2.p;
  expect_prime(p, PRIME_P, "base field");
  expect_prime(p, PRIME_R, "scalar field");
  const uint32_t nvars = u32(p), npubs = u32(p + 4), domsiz = u32(p + 8);
  p += 12;
  uint32_t log2n = 0;
  while ((1u << log2n) < domsiz) ++log2n;
  if ((1u << log2n) != domsiz) die("domain size should be a power of two");
  const uint8_t *alpha1 = p, *beta1 = p + 64, *beta2 = p + 128, *gamma2 = p + 256, *delta1 = p + 384, *delta2 = p + 448;
  auto points = [&](uint32_t id, size_t psz, size_t n) {
    Section s = zk.get(id);
    if (s.len != psz * n) die("unexpected length of section " + std::to_string(id));
    return s.p;
  };
  const uint8_t* ic = points(3, 64, (size_t)npubs + 1);
  Section s4 = zk.get(4);
  const uint32_t ncoeffs = u32(s4.p);
  if (s4.len != 4 + (size_t)ncoeffs * 44) die("unexpected coefficient section length");
  std::vector<g16_coeff> coeffs(ncoeffs);
  for (uint32_t i = 0; i < ncoeffs; ++i) {
    const uint8_t* e = s4.p + 4 + (size_t)i * 44;
    coeffs[i].matrix = u32(e);
    coeffs[i].row = u32(e + 4);
    coeffs[i].col = u32(e + 8);
    coeffs[i].reserved = 0;
    if (coeffs[i].matrix > 2 || coeffs[i].row >= domsiz || coeffs[i].col >= nvars) die("coefficient out of range");
    U256 v = mont_reduce(load(e + 12), PRIME_R, R_NINV);  // c*R^2 -> c*R   (unmarshalFrWTF, io.nim:134-139)
    memcpy(coeffs[i].value, v.v, 32);
  }
  // ---- .wtns (witness.nim:36-60) ----
  Container wt(wpath, "wtns", 2);
  Section w1 = wt.get(1);
  if (w1.len != 4 + 32 + 4) die("unexpected witness header length");
  const uint8_t* wp = w1.p;
  expect_prime(wp, PRIME_R, "witness field");
  if (u32(wp) != nvars) die("wrong witness length");  // prover.nim:236
  Section w2 = wt.get(2);
  if (w2.len != (size_t)nvars * 32) die("unexpected witness section length");
  const double t1 = now();

  g16_ctx* ctx = nullptr;
  if (g16_ctx_create(0, &ctx) != G16_OK) die("no usable GPU (there is no CPU fallback)");
  auto chk = [&](int32_t rc, const char* what) {
    if (rc != G16_OK) die(std::string(what) + " failed: " + g16_last_error(ctx));
  };
  chk(g16_selftest(ctx), "g16_selftest");
  g16_pkey_desc d;
  memset(&d, 0, sizeof d);
  d.nvars = nvars, d.npubs = npubs, d.log2_domain = log2n, d.flavour = G16_FLAVOUR_SNARKJS;  // zkey.nim:129
  d.pointsA1 = points(5, 64, nvars), d.pointsB1 = points(6, 64, nvars), d.pointsB2 = points(7, 128, nvars);
  d.pointsC1 = points(8, 64, (size_t)nvars - npubs - 1), d.pointsH1 = points(9, 64, domsiz);
  d.coeffs = coeffs.data(), d.ncoeffs = ncoeffs;
  d.alpha1 = alpha1, d.beta1 = beta1, d.delta1 = delta1, d.beta2 = beta2, d.delta2 = delta2;
  d.shard_index = 0, d.shard_count = 1;
  g16_pkey* key = nullptr;
  chk(g16_pkey_create(ctx, &d, &key), "g16_pkey_create");
  const double t2 = now();

  // mask (prover.nim:312-319): r, s uniform in Fr unless -n; passed in Montgomery form
  uint8_t rmask[32], smask[32];
  const uint8_t *rp = nullptr, *sp = nullptr;
  if (!nomask) {
    std::random_device rd;
    for (uint8_t* m : {rmask, smask}) {
      U256 x;
      do {
        for (int i = 0; i < 4; ++i) x.v[i] = ((uint64_t)rd() << 32) | rd();
        x.v[3] &= 0x3fffffffffffffffull;
      } while (geq(x, PRIME_R));
      // a uniform residue is a uniform Montgomery residue: use the limbs as they are
      memcpy(m, x.v, 32);
    }
    rp = rmask, sp = smask;
  }
  g16_proof proof;
  chk(g16_prove(ctx, key, w2.p, G16_SCALARS_STD, rp, sp, &proof), "g16_prove");
  const double t3 = now();

  FILE* f = fopen(opath, "w");  // export_json.nim:70-80
  if (!f) die(std::string("cannot write ") + opath);
  fprintf(f, "{ \"protocol\": \"groth16\"\n, \"curve\":    \"bn128\"\n, \"pi_a\":\n");
  write_g1(f, proof.pi_a);
  fprintf(f, ", \"pi_b\":\n");
  write_g2(f, proof.pi_b);
  fprintf(f, ", \"pi_c\":\n");
  write_g1(f, proof.pi_c);
  fprintf(f, "}\n");
  fclose(f);
  f = fopen(ipath, "w");  // export_json.nim:25-44 (the constant 1 is skipped)
  if (!f) die(std::string("cannot write ") + ipath);
  if (npubs == 0) fprintf(f, "[ ]\n");
  for (uint32_t i = 1; i <= npubs; ++i) fprintf(f, "%s\"%s\"\n", i == 1 ? "[ " : ", ", decimal(load(w2.p + 32 * i)).c_str());
  if (npubs) fprintf(f, "] \n");
  fclose(f);

  if (verify) {  // verifier.nim:31-52 on the GPU
    g16_vkey_desc vd;
    vd.npubs = npubs, vd.alpha1 = alpha1, vd.beta2 = beta2, vd.gamma2 = gamma2, vd.delta2 = delta2, vd.pointsIC = ic;
    g16_vkey* vk = nullptr;
    chk(g16_vkey_create(ctx, &vd, &vk), "g16_vkey_create");
    int32_t st = 0;
    chk(g16_verify(ctx, vk, &proof, w2.p, G16_SCALARS_STD, 1, &st), "g16_verify");
    g16_vkey_destroy(vk);
    printf("verification %s\n", st == 1 ? "succeeded" : "FAILED");
    if (st != 1) return 1;
  }
  if (timing)
    printf("parsing %.3fs | key upload + tables %.3fs | proof %.3fs\n", t1 - t0, t2 - t1, t3 - t2);
  g16_pkey_destroy(key);
  g16_ctx_destroy(ctx);
  return 0;
}
