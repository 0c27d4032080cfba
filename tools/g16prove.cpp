// g16prove -- native host side above the C ABI: the `-p` (prove) path of the reference CLI
// (cli/cli_main.nim:162-231: parseZKey, parseWitness, generateProof, exportProof, exportPublicIO, verifyProof)
// as a plain C++ program linked against libg16hip.so.  No Python, no torch.
//
//   g++ -O2 -std=c++17 -Iinclude tools/g16prove.cpp -Lnim_groth16_amd/csrc -lg16hip
//       -Wl,-rpath,$PWD/nim_groth16_amd/csrc -o g16prove
//   ./g16prove -z circuit.zkey -w witness.wtns -o proof.json -i public.json [-n] [-y] [-t] [--gpus 0,1,2,3]
//
// --gpus d0,d1,...: the proof sharded over those devices through the device group of the C ABI (g16_group_*: one host
// thread per device inside the library; the reference's Taskpool shape, msm.nim:96-122).  A device may repeat.
//
// The file readers live in tools/g16_files.hpp.
#define G16_TOOL_NAME "g16prove"
#include "g16_files.hpp"

int main(int argc, char** argv) {
  const char *zpath = nullptr, *wpath = nullptr, *opath = "proof.json", *ipath = "public.json";
  bool nomask = false, verify = false, timing = false;
  std::vector<int32_t> gpus;
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
    else if (a == "--gpus") {
      for (const char* q = next(); *q;) {
        char* end = nullptr;
        const long d = strtol(q, &end, 10);
        if (end == q || d < 0) die("--gpus takes a comma-separated list of device ordinals");
        gpus.push_back((int32_t)d);
        q = *end == ',' ? end + 1 : end;
      }
    }
    else die("unknown option " + a + "\nusage: g16prove -z circuit.zkey -w witness.wtns [-o proof.json] [-i public.json] [-n] [-y] [-t] [--gpus 0,1,...]");
  }
  if (!zpath || !wpath) die("usage: g16prove -z circuit.zkey -w witness.wtns [-o proof.json] [-i public.json] [-n] [-y] [-t]");

  const double t0 = now();
  ZkeyFile zf(zpath);
  WtnsFile wf(wpath, zf.nvars);
  const uint32_t npubs = zf.npubs;
  const uint8_t* wvals = wf.values;
  const double t1 = now();

  g16_ctx* ctx = nullptr;
  if (g16_ctx_create(gpus.empty() ? 0 : gpus[0], &ctx) != G16_OK) die("no usable GPU (there is no CPU fallback)");
  auto chk = [&](int32_t rc, const char* what) {
    if (rc != G16_OK) die(std::string(what) + " failed: " + g16_last_error(ctx));
  };
  chk(g16_selftest(ctx), "g16_selftest");
  g16_pkey_desc d = zf.desc(gpus.empty());   // one GPU: the coefficient section goes to the library as it lies in the file
  g16_pkey* key = nullptr;
  g16_group* grp = nullptr;
  g16_group_pkey* gkey = nullptr;
  if (gpus.empty()) {
    chk(g16_pkey_create_zkey(ctx, &d, zf.section4, zf.section4_len, &key), "g16_pkey_create_zkey");
  } else {
    if (g16_group_create(gpus.data(), (int32_t)gpus.size(), &grp) != G16_OK) die("g16_group_create failed");
    if (g16_group_pkey_create(grp, &d, &gkey) != G16_OK)
      die(std::string("g16_group_pkey_create failed: ") + g16_group_last_error(grp));
  }
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
  if (grp) {
    if (g16_group_prove(grp, gkey, wvals, G16_SCALARS_STD, rp, sp, &proof) != G16_OK)
      die(std::string("g16_group_prove failed: ") + g16_group_last_error(grp));
  } else {
    chk(g16_prove(ctx, key, wvals, G16_SCALARS_STD, rp, sp, &proof), "g16_prove");
  }
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
  for (uint32_t i = 1; i <= npubs; ++i) fprintf(f, "%s\"%s\"\n", i == 1 ? "[ " : ", ", decimal(load(wvals + 32 * i)).c_str());
  if (npubs) fprintf(f, "] \n");
  fclose(f);

  if (verify) {  // verifier.nim:31-52 on the GPU
    g16_vkey_desc vd;
    vd.npubs = npubs, vd.alpha1 = zf.alpha1, vd.beta2 = zf.beta2, vd.gamma2 = zf.gamma2, vd.delta2 = zf.delta2, vd.pointsIC = zf.ic;
    g16_vkey* vk = nullptr;
    chk(g16_vkey_create(ctx, &vd, &vk), "g16_vkey_create");
    int32_t st = 0;
    chk(g16_verify(ctx, vk, &proof, wvals, G16_SCALARS_STD, 1, &st), "g16_verify");
    g16_vkey_destroy(vk);
    printf("verification %s\n", st == 1 ? "succeeded" : "FAILED");
    if (st != 1) return 1;
  }
  if (timing)
    printf("parsing %.3fs | key upload + tables %.3fs | proof %.3fs\n", t1 - t0, t2 - t1, t3 - t2);
  g16_pkey_destroy(key);
  g16_group_pkey_destroy(gkey);
  g16_group_destroy(grp);
  g16_ctx_destroy(ctx);
  return 0;
}
