// .zkey / .wtns readers over read-only memory maps, shared by the native tools (g16prove.cpp, ab_prove.cpp).
// File formats restated from the reference (groth16/files/container.nim:6-20, zkey.nim:6-91, witness.nim:5-15):
// the point sections of a .zkey are little-endian Montgomery with R = 2^256 -- byte for byte the layout the
// library takes -- so they go from the memory map to the GPU unparsed; .wtns values are canonical little-endian
// and are passed with G16_SCALARS_STD.  The section-4 coefficients (doubly Montgomery-encoded, bn128/io.nim:134-139) go
// in unparsed too since g16_pkey_create_zkey; for g16_pkey_create (and the builds before round 5 that ab_prove dlopen()s)
// convert_coeffs() turns them into g16_coeff records with one Montgomery reduction each.
#pragma once
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


#ifndef G16_TOOL_NAME
#define G16_TOOL_NAME "g16"
#endif

namespace {   // (internal linkage: every tool that includes this header gets its own copy)

[[noreturn]] void die(const std::string& msg) {
  fprintf(stderr, "%s: %s\n", G16_TOOL_NAME, msg.c_str());
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
[[maybe_unused]] std::string decimal(U256 x) {  // canonical value -> decimal string
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
[[maybe_unused]] std::string fp_dec(const uint8_t* mont) { return decimal(mont_reduce(load(mont), PRIME_P, P_NINV)); }

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
    if (fstat(fd, &st) != 0 || st.st_size <= 0) die(std::string("cannot stat ") + path);
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
      if (n - pos < 12) die("truncated file");
      uint32_t id;
      uint64_t len;
      memcpy(&id, b + pos, 4);
      memcpy(&len, b + pos + 4, 8);
      pos += 12;
      if (len > n - pos) die("truncated section");   // (not pos + len > n: a crafted 64-bit length would wrap)
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

[[maybe_unused]] void write_g1(FILE* f, const uint8_t* p) {  // export_json.nim:55-59
  fprintf(f, "    [ \"%s\"\n    , \"%s\"\n    , \"1\"\n    ]\n", fp_dec(p).c_str(), fp_dec(p + 32).c_str());
}
[[maybe_unused]] void write_fp2(FILE* f, const char* lead, const std::string& a, const std::string& b) {  // export_json.nim:48-53
  fprintf(f, "    %s [ \"%s\"\n      , \"%s\"\n      ]\n", lead, a.c_str(), b.c_str());
}
[[maybe_unused]] void write_g2(FILE* f, const uint8_t* p) {  // export_json.nim:61-65
  write_fp2(f, "[", fp_dec(p), fp_dec(p + 32));
  write_fp2(f, ",", fp_dec(p + 64), fp_dec(p + 96));
  write_fp2(f, ",", "1", "0");
  fprintf(f, "    ]\n");
}

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }


// ---- parsed .zkey / .wtns: pointers into the memory maps + the reduced section-4 coefficients -------------------
struct ZkeyFile {
  Container zk;
  uint32_t nvars = 0, npubs = 0, domsiz = 0, log2n = 0;
  const uint8_t *alpha1 = nullptr, *beta1 = nullptr, *beta2 = nullptr, *gamma2 = nullptr, *delta1 = nullptr,
                *delta2 = nullptr, *ic = nullptr;
  mutable std::vector<g16_coeff> coeffs;   // only for callers of the older g16_pkey_create (ab_prove of earlier builds)
  const uint8_t* section4 = nullptr;       // the coefficient section as it lies in the file (g16_pkey_create_zkey)
  size_t section4_len = 0;
  explicit ZkeyFile(const char* path) : zk(path, "zkey", 1) {   // zkey.nim:104-224
    if (zk.get(1).len != 4 || u32(zk.get(1).p) != 1) die("expecting `.zkey` file for a Groth16 prover");
    Section s2 = zk.get(2);
    if (s2.len != 2 * 4 + 64 + 3 * 4 + 3 * 64 + 3 * 128) die("unexpected header section length");
    const uint8_t* p = s2.p;
    expect_prime(p, PRIME_P, "base field");
    expect_prime(p, PRIME_R, "scalar field");
    nvars = u32(p), npubs = u32(p + 4), domsiz = u32(p + 8);
    p += 12;
    while ((1u << log2n) < domsiz) ++log2n;
    if ((1u << log2n) != domsiz) die("domain size should be a power of two");
    alpha1 = p, beta1 = p + 64, beta2 = p + 128, gamma2 = p + 256, delta1 = p + 384, delta2 = p + 448;
    ic = points(3, 64, (size_t)npubs + 1);
    Section s4 = zk.get(4);
    if (s4.len < 4) die("coefficient section too short");
    const uint32_t ncoeffs = u32(s4.p);
    if (s4.len != 4 + (size_t)ncoeffs * 44) die("unexpected coefficient section length");
    section4 = s4.p, section4_len = s4.len;   // handed to the library unparsed: no arithmetic on the host
  }
  // the coefficients as g16_coeff records (values c R): what g16_pkey_create of the builds before g16_pkey_create_zkey takes
  void convert_coeffs() const {
    const uint32_t ncoeffs = u32(section4);
    coeffs.resize(ncoeffs);
    for (uint32_t i = 0; i < ncoeffs; ++i) {
      const uint8_t* e = section4 + 4 + (size_t)i * 44;
      coeffs[i].matrix = u32(e);
      coeffs[i].row = u32(e + 4);
      coeffs[i].col = u32(e + 8);
      coeffs[i].reserved = 0;
      if (coeffs[i].matrix > 2 || coeffs[i].row >= domsiz || coeffs[i].col >= nvars) die("coefficient out of range");
      U256 v = mont_reduce(load(e + 12), PRIME_R, R_NINV);  // c*R^2 -> c*R   (unmarshalFrWTF, io.nim:134-139)
      memcpy(coeffs[i].value, v.v, 32);
    }
  }
  const uint8_t* points(uint32_t id, size_t psz, size_t n) const {
    Section s = zk.get(id);
    if (s.len != psz * n) die("unexpected length of section " + std::to_string(id));
    return s.p;
  }
  // the proving-key description of include/g16hip.h: the point sections go to the GPU as they lie in the file.
  // raw_coeffs = true: desc.coeffs stays NULL and the caller passes section4 / section4_len to g16_pkey_create_zkey
  g16_pkey_desc desc(bool raw_coeffs = false) const {
    g16_pkey_desc d;
    memset(&d, 0, sizeof d);
    d.nvars = nvars, d.npubs = npubs, d.log2_domain = log2n, d.flavour = G16_FLAVOUR_SNARKJS;  // zkey.nim:129
    d.pointsA1 = points(5, 64, nvars), d.pointsB1 = points(6, 64, nvars), d.pointsB2 = points(7, 128, nvars);
    d.pointsC1 = points(8, 64, (size_t)nvars - npubs - 1), d.pointsH1 = points(9, 64, domsiz);
    if (!raw_coeffs) {
      if (coeffs.empty()) convert_coeffs();
      d.coeffs = coeffs.data(), d.ncoeffs = coeffs.size();
    }
    d.alpha1 = alpha1, d.beta1 = beta1, d.delta1 = delta1, d.beta2 = beta2, d.delta2 = delta2;
    d.shard_index = 0, d.shard_count = 1;
    return d;
  }
};
struct WtnsFile {   // witness.nim:36-60
  Container wt;
  const uint8_t* values = nullptr;   // nvars x 32 bytes, canonical little-endian
  WtnsFile(const char* path, uint32_t nvars) : wt(path, "wtns", 2) {
    Section w1 = wt.get(1);
    if (w1.len != 4 + 32 + 4) die("unexpected witness header length");
    const uint8_t* wp = w1.p;
    expect_prime(wp, PRIME_R, "witness field");
    if (u32(wp) != nvars) die("wrong witness length");  // prover.nim:236
    Section w2 = wt.get(2);
    if (w2.len != (size_t)nvars * 32) die("unexpected witness section length");
    values = w2.p;
  }
};

}  // namespace
