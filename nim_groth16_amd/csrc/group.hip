// A device GROUP behind the C ABI: one proof sharded over several GPUs of a node from ONE host process, one host thread
// per device inside the library -- the shape of the reference, which is one process with Taskpool threads
// (groth16/bn128/msm.nim:96-122: contiguous index ranges of every MSM, one task each, partial sums added in task order;
// prover.nim:165-173: the three coset pipelines as three tasks).  A Nim host calls g16_group_prove like it calls
// generateProofWithMask; no Python, no torch.distributed, no RCCL: the exchange is 3 x 32 n / G bytes of coset slices per
// device (hipMemcpyPeerAsync between the owners' and the receivers' HBM) and G 768-byte records to the host.
//
//   member g: a g16_ctx on devices[g] + the key shard g of G (g16_pkey_desc.shard_index / shard_count)
//   prove:    every member  g16_prove_partials_begin  (its witness MSMs keep running; its coset pipelines complete)
//             every member  copies its [h_lo, h_hi) slices from the pipelines' owners (peer copies on its own stream)
//             every member  g16_prove_partials_end    -> record g
//             member 0      g16_prove_combine over the G records in member order
// JensGroth keys (a seventh transform over the whole vector) take g16_prove_partials on every member instead.
// The same device may appear several times in `devices` (tests: every "device" = 0 on a one-GPU box).
#include <thread>

#include "g16_internal.hpp"

namespace {
struct Member {
  int device = 0;
  g16_ctx* ctx = nullptr;
};
// coset pipeline v (0: A, 1: B, 2: C) lives on member v % G: prover.nim:167-169's three tasks dealt round robin
inline uint32_t task_owner(uint32_t v, uint32_t G) { return v % G; }
}  // namespace

struct g16_group {
  std::vector<Member> m;
  std::string err;
};

struct g16_group_pkey {
  g16_group* grp = nullptr;          // identity check in g16_group_prove only: the key may outlive the group
  std::vector<int> device;           // devices[g], for the key's own teardown
  uint32_t nvars = 0, log2n = 0, flavour = 1;
  std::vector<g16_pkey*> key;        // shard g on member g
  std::vector<void*> task_out;       // member g: its owned coset vectors (owned x n Fr), HBM of devices[g]
  std::vector<void*> slices;         // member g: 3 x (h_hi - h_lo) Fr, HBM of devices[g]
  std::vector<size_t> h_lo, h_hi;
};

extern "C" const char* g16_group_last_error(const g16_group* g) { return g ? g->err.c_str() : "null group"; }
extern "C" int32_t g16_group_size(const g16_group* g) { return g ? (int32_t)g->m.size() : 0; }

extern "C" void g16_group_destroy(g16_group* g) {
  if (!g) return;
  for (auto& mb : g->m) g16_ctx_destroy(mb.ctx);
  delete g;
}

extern "C" int32_t g16_group_create(const int32_t* devices, int32_t ndev, g16_group** out) {
  if (!out) return G16_EINVAL;
  *out = nullptr;
  if (!devices || ndev < 1 || ndev > 64) return G16_EINVAL;
  g16_group* g = new (std::nothrow) g16_group();
  if (!g) return G16_ENOMEM;
  for (int32_t i = 0; i < ndev; ++i) {
    Member mb;
    mb.device = devices[i];
    const int32_t rc = g16_ctx_create(devices[i], &mb.ctx);
    if (rc != G16_OK) {
      g16_group_destroy(g);
      return rc;
    }
    g->m.push_back(mb);
  }
  // direct peer copies where the hardware offers them (xGMI); without it hipMemcpyPeerAsync stages through the host
  for (auto& a : g->m)
    for (auto& b : g->m)
      if (a.device != b.device) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a.device, b.device) == hipSuccess && can && hipSetDevice(a.device) == hipSuccess) {
          (void)hipDeviceEnablePeerAccess(b.device, 0);
          (void)hipGetLastError();   // "already enabled" is fine
        }
      }
  *out = g;
  return G16_OK;
}

extern "C" void g16_group_pkey_destroy(g16_group_pkey* k) {
  if (!k) return;
  for (size_t i = 0; i < k->key.size(); ++i) {
    g16_pkey_destroy(k->key[i]);
    if (i < k->device.size()) (void)hipSetDevice(k->device[i]);
    if (i < k->task_out.size() && k->task_out[i]) (void)hipFree(k->task_out[i]);
    if (i < k->slices.size() && k->slices[i]) (void)hipFree(k->slices[i]);
  }
  delete k;
}

// run fn(g) for every member on its own host thread; the first failure wins
template <class F>
static int32_t for_members(g16_group* g, F fn) {
  const size_t G = g->m.size();
  std::vector<int32_t> rc(G, G16_OK);
  if (G == 1) {
    rc[0] = fn(0);
  } else {
    std::vector<std::thread> th;
    th.reserve(G);
    try {   // nothing throws across the C ABI: a thread that cannot be started is a failed member
      for (size_t i = 0; i < G; ++i) th.emplace_back([&, i] { rc[i] = fn(i); });
    } catch (...) {
      for (size_t i = th.size(); i < G; ++i) rc[i] = G16_ENOMEM;
    }
    for (auto& t : th) t.join();
  }
  for (size_t i = 0; i < G; ++i)
    if (rc[i] != G16_OK) {
      g->err = "member " + std::to_string(i) + " (device " + std::to_string(g->m[i].device) + "): " +
               g16_last_error(g->m[i].ctx);
      return rc[i];
    }
  return G16_OK;
}

extern "C" int32_t g16_group_pkey_create(g16_group* g, const g16_pkey_desc* desc, g16_group_pkey** out) {
  if (!g) return G16_EINVAL;
  if (!desc || !out) {
    g->err = "null argument";
    return G16_EINVAL;
  }
  *out = nullptr;
  if (desc->shard_count > 1) {
    g->err = "the group shards the key itself: pass the whole key (shard_count 0 or 1)";
    return G16_EINVAL;
  }
  const size_t G = g->m.size();
  g16_group_pkey* k = new (std::nothrow) g16_group_pkey();
  if (!k) return G16_ENOMEM;
  k->grp = g;
  for (auto& mb : g->m) k->device.push_back(mb.device);
  k->nvars = desc->nvars, k->log2n = desc->log2_domain, k->flavour = desc->flavour;
  k->key.assign(G, nullptr);
  k->task_out.assign(G, nullptr);
  k->slices.assign(G, nullptr);
  k->h_lo.assign(G, 0);
  k->h_hi.assign(G, 0);
  const size_t n = size_t(1) << desc->log2_domain;
  // table precomputation of the shards runs concurrently, one thread per member
  int32_t rc = for_members(g, [&](size_t i) -> int32_t {
    g16_pkey_desc d = *desc;
    d.shard_index = (uint32_t)i;
    d.shard_count = (uint32_t)G;
    int32_t r = g16_pkey_create(g->m[i].ctx, &d, &k->key[i]);
    if (r != G16_OK) return r;
    k->h_lo[i] = (n * i) / G;               // msm.nim:107-115: b = (N * (k + 1)) div ntasks
    k->h_hi[i] = (n * (i + 1)) / G;
    if (desc->flavour != G16_FLAVOUR_SNARKJS) return G16_OK;
    size_t owned = 0;
    for (uint32_t v = 0; v < 3; ++v) owned += task_owner(v, (uint32_t)G) == i;
    if (hipSetDevice(g->m[i].device) != hipSuccess) return G16_EHIP;
    if (owned && hipMalloc(&k->task_out[i], owned * n * 32) != hipSuccess) return G16_ENOMEM;
    const size_t nh = k->h_hi[i] - k->h_lo[i];
    if (nh && hipMalloc(&k->slices[i], 3 * nh * 32) != hipSuccess) return G16_ENOMEM;
    return G16_OK;
  });
  if (rc != G16_OK) {
    g16_group_pkey_destroy(k);
    return rc;
  }
  *out = k;
  return G16_OK;
}

extern "C" int32_t g16_group_prove(g16_group* g, const g16_group_pkey* k, const void* witness, uint32_t flags,
                                   const void* mask_r, const void* mask_s, g16_proof* out) {
  if (!g) return G16_EINVAL;
  if (!k || k->grp != g || !witness || !out || (flags & (G16_SCALARS_DEVICE | G16_OUT_DEVICE | G16_NO_HOST_SYNC))) {
    g->err = "bad argument (the witness is a host pointer; flags: G16_SCALARS_MONT or G16_SCALARS_STD)";
    return G16_EINVAL;
  }
  const size_t G = g->m.size();
  const size_t n = size_t(1) << k->log2n;
  const uint32_t wflags = flags & G16_SCALARS_MONT;
  std::vector<unsigned char> records;
  try {
    records.resize(G * G16_PARTIALS_BYTES);
  } catch (...) {
    g->err = "out of host memory";
    return G16_ENOMEM;
  }
  int32_t rc;
  if (k->flavour != G16_FLAVOUR_SNARKJS) {
    rc = for_members(g, [&](size_t i) {
      return g16_prove_partials(g->m[i].ctx, k->key[i], witness, wflags, records.data() + i * G16_PARTIALS_BYTES);
    });
  } else {
    // phase 1: witness MSMs launched, owned coset vectors complete when the call returns
    rc = for_members(g, [&](size_t i) {
      uint32_t mask = 0;
      for (uint32_t v = 0; v < 3; ++v)
        if (task_owner(v, (uint32_t)G) == i) mask |= 1u << v;
      return g16_prove_partials_begin(g->m[i].ctx, k->key[i], witness, wflags, mask, k->task_out[i]);
    });
    // phase 2: every member fetches its slices from the owners' HBM and finishes
    if (rc == G16_OK)
      rc = for_members(g, [&](size_t i) -> int32_t {
        g16_ctx* ctx = g->m[i].ctx;
        const size_t nh = k->h_hi[i] - k->h_lo[i];
        char* sl = (char*)k->slices[i];
        if (hipSetDevice(g->m[i].device) != hipSuccess) return G16_EHIP;
        for (uint32_t v = 0; v < 3 && nh; ++v) {
          const size_t o = task_owner(v, (uint32_t)G);
          size_t idx = 0;                               // position of v among the owner's pipelines
          for (uint32_t u = 0; u < v; ++u) idx += task_owner(u, (uint32_t)G) == o;
          const char* src = (const char*)k->task_out[o] + (idx * n + k->h_lo[i]) * 32;
          const hipError_t e = g->m[o].device == g->m[i].device
                                   ? hipMemcpyAsync(sl + v * nh * 32, src, nh * 32, hipMemcpyDeviceToDevice, ctx->stream)
                                   : hipMemcpyPeerAsync(sl + v * nh * 32, g->m[i].device, src, g->m[o].device, nh * 32,
                                                        ctx->stream);
          if (e != hipSuccess) {
            ctx->err = std::string("coset slice copy: ") + hipGetErrorString(e);
            return G16_EHIP;
          }
        }
        return g16_prove_partials_end(ctx, k->key[i], sl, sl + nh * 32, sl + 2 * nh * 32, 0,
                                      records.data() + i * G16_PARTIALS_BYTES);
      });
  }
  if (rc != G16_OK) {
    for (auto& mb : g->m) (void)g16_ctx_cancel(mb.ctx);   // no member keeps lanes running or a proof pending
    return rc;
  }
  rc = g16_prove_combine(g->m[0].ctx, k->key[0], records.data(), G, 0, mask_r, mask_s, out);
  if (rc != G16_OK) g->err = std::string("combine: ") + g16_last_error(g->m[0].ctx);
  return rc;
}
