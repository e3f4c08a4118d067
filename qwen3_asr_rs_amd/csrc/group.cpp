// Native multi-GPU entry points: one process, one host thread per GPU.
//
//   q3a_group_create      packs the weight arena ONCE on the host, uploads it to the first GPU, and ships it to every other
//                         GPU with ONE ncclBroadcast over xGMI (RCCL, communicators from ncclCommInitAll); every GPU then
//                         gets an engine created from its device copy (q3a_engine_create_from_arena).
//   q3a_group_transcribe  partitions the B independent utterances contiguously over the GPUs (the same split as
//                         qwen3_asr_rs_amd/distributed.py::partition) and runs q3a_transcribe_batch on every GPU from its own
//                         host thread.  No collective on the data path: utterances are independent
//                         (AsrInference::transcribe is a pure function of weights and samples, src/inference.rs:89).
//
// The reference is single-process / single-device (src/main.rs:51-65); a Rust host binds these exactly like the engine
// entry points.  RCCL is loaded with dlopen at group creation so that libq3asr_hip.so itself keeps no
// link-time dependency on librccl (a 1-GPU group never touches it unless Q3A_GROUP_FORCE_RCCL=1 asks for a 1-rank
// communicator -- the test hook that exercises the RCCL calls on a single-GPU box).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/q3asr.h"
#include "model.h"

using namespace q3a;

namespace q3a {
void set_thread_error(const std::string& msg);  // engine.cpp
}

namespace {

// the slice of rccl.h this file needs (ABI of RCCL 2.x as shipped in /opt/rocm/include/rccl/rccl.h)
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;     // ncclSuccess == 0
typedef int ncclDataType_t;   // ncclUint8 == 1
struct Rccl {
  void* so = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  void load() {
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (so) break;
    }
    if (!so) fail(std::string("q3a_group_create: cannot load librccl.so (") + dlerror() + ")");
    auto sym = [&](const char* n) {
      void* p = dlsym(so, n);
      if (!p) fail(std::string("q3a_group_create: librccl.so lacks ") + n);
      return p;
    };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    Broadcast = reinterpret_cast<decltype(Broadcast)>(sym("ncclBroadcast"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
  }
  void check(ncclResult_t r, const char* what) const {
    if (r != 0) fail(std::string("RCCL ") + what + ": " + (GetErrorString ? GetErrorString(r) : "error"));
  }
};

#define GHIP(expr)                                                                                          \
  do {                                                                                                      \
    hipError_t _e = (expr);                                                                                 \
    if (_e != hipSuccess) fail(std::string("HIP error: ") + hipGetErrorString(_e) + " (" #expr ")");        \
  } while (0)

}  // namespace

struct q3a_group {
  std::vector<int> devices;
  std::vector<void*> arenas;          // one device copy of the weight arena per GPU (owned)
  std::vector<q3a_engine*> engines;
  uint64_t arena_bytes = 0;
  bool used_rccl = false;
  double startup_s[4] = {0, 0, 0, 0};  // pack (read + lay out the checkpoint), upload (H2D to the first GPU), broadcast, engines
  std::string err;
  ~q3a_group() {
    for (auto* e : engines)
      if (e) q3a_engine_destroy(e);
    for (size_t i = 0; i < arenas.size(); ++i)
      if (arenas[i]) { (void)hipSetDevice(devices[i]); (void)hipFree(arenas[i]); }
  }
};

extern "C" {

void q3a_group_partition(int32_t n_items, int32_t world_size, int32_t rank, int32_t* begin, int32_t* end) {
  const int base = n_items / world_size, rem = n_items % world_size;  // distributed.py::partition
  const int b = rank * base + (rank < rem ? rank : rem);
  if (begin) *begin = b;
  if (end) *end = b + base + (rank < rem ? 1 : 0);
}

int32_t q3a_group_create(const char* model_dir, int32_t n_gpus, const int32_t* devices, const q3a_opts* opts, q3a_group** out) {
  std::unique_ptr<q3a_group> g(new q3a_group());
  try {
    if (!model_dir || !out || n_gpus < 1) fail("q3a_group_create: bad argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) fail("no HIP device available: libq3asr_hip has no CPU fallback");
    for (int i = 0; i < n_gpus; ++i) {
      const int dv = devices ? devices[i] : i;
      if (dv < 0 || dv >= n_dev) fail("q3a_group_create: device index out of range (" + std::to_string(dv) + " of " + std::to_string(n_dev) + ")");
      for (int j : g->devices)
        if (j == dv) fail("q3a_group_create: a GPU may appear only once in a group");
      g->devices.push_back(dv);
    }
    uint64_t bytes = 0;
    if (q3a_arena_bytes(model_dir, &bytes) != 0) fail(q3a_last_error(nullptr));
    g->arena_bytes = bytes;
    g->arenas.assign(n_gpus, nullptr);
    g->engines.assign(n_gpus, nullptr);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    // RAII for everything start-up creates besides the group's own members: an exception below (HIP, RCCL, a bad checkpoint)
    // leaves no stream, communicator or pinned allocation behind
    struct Stream {
      int dev = 0;
      hipStream_t s = nullptr;
      ~Stream() { if (s) { (void)hipSetDevice(dev); (void)hipStreamDestroy(s); } }
    };
    {  // the checkpoint is read ONCE (weights.rs:10-120) and packed straight into pinned host memory: the upload is one
       // asynchronous DMA at PCIe rate instead of a pageable copy staged through the driver's bounce buffers.  A host that
       // refuses to pin the whole arena (1.6 / 4.1 GB; ulimit -l, cgroup limits) falls back to pageable memory + a blocking copy.
      GHIP(hipSetDevice(g->devices[0]));
      struct Pinned { void* p = nullptr; ~Pinned() { if (p) (void)hipHostFree(p); } } pinned;
      std::vector<uint8_t> pageable;
      void* host = nullptr;
      if (hipHostMalloc(&pinned.p, bytes, hipHostMallocDefault) == hipSuccess && pinned.p) {
        host = pinned.p;
      } else {
        (void)hipGetLastError();  // clear the sticky error of the failed allocation
        pinned.p = nullptr;
        pageable.resize(bytes);
        host = pageable.data();
      }
      const auto t0 = now();
      if (q3a_arena_pack(model_dir, host, bytes) != 0) fail(q3a_last_error(nullptr));
      const auto t1 = now();
      for (int i = 0; i < n_gpus; ++i) {
        GHIP(hipSetDevice(g->devices[i]));
        GHIP(hipMalloc(&g->arenas[i], bytes));
      }
      GHIP(hipSetDevice(g->devices[0]));
      if (pinned.p) {
        Stream up{g->devices[0]};
        GHIP(hipStreamCreateWithFlags(&up.s, hipStreamNonBlocking));
        GHIP(hipMemcpyAsync(g->arenas[0], host, bytes, hipMemcpyHostToDevice, up.s));
        GHIP(hipStreamSynchronize(up.s));
      } else {
        GHIP(hipMemcpy(g->arenas[0], host, bytes, hipMemcpyHostToDevice));
      }
      g->startup_s[0] = secs(t0, t1);
      g->startup_s[1] = secs(t1, now());
    }
    const char* force = getenv("Q3A_GROUP_FORCE_RCCL");
    if (n_gpus > 1 || (force && atoi(force) != 0)) {
      // ONE broadcast of the whole arena (1.56 GB at 0.6B, 4.08 GB at 1.7B): xGMI is point-to-point, one large message
      // per link is the cheapest shape; steady state has no collective at all
      const auto tb = now();
      Rccl r;
      r.load();
      struct Comms {  // ncclCommInitAll creates all of them or none; destroyed on every way out
        const Rccl& r;
        std::vector<ncclComm_t> c;
        ~Comms() { for (auto x : c) if (x) (void)r.CommDestroy(x); }
      } comms{r, std::vector<ncclComm_t>(n_gpus, nullptr)};
      std::vector<Stream> streams(n_gpus);
      r.check(r.CommInitAll(comms.c.data(), n_gpus, g->devices.data()), "ncclCommInitAll");
      for (int i = 0; i < n_gpus; ++i) {
        streams[i].dev = g->devices[i];
        GHIP(hipSetDevice(g->devices[i]));
        GHIP(hipStreamCreateWithFlags(&streams[i].s, hipStreamNonBlocking));
      }
      r.check(r.GroupStart(), "ncclGroupStart");
      ncclResult_t first_bad = 0;
      for (int i = 0; i < n_gpus; ++i) {
        hipError_t he = hipSetDevice(g->devices[i]);
        const ncclResult_t br = he == hipSuccess ? r.Broadcast(g->arenas[i], g->arenas[i], bytes, /*ncclUint8*/ 1, /*root*/ 0, comms.c[i], streams[i].s) : 1;
        if (br != 0 && first_bad == 0) first_bad = br;  // (the group must still be closed before anything throws)
      }
      const ncclResult_t ge = r.GroupEnd();
      r.check(first_bad, "ncclBroadcast");
      r.check(ge, "ncclGroupEnd");
      for (int i = 0; i < n_gpus; ++i) {
        GHIP(hipSetDevice(g->devices[i]));
        GHIP(hipStreamSynchronize(streams[i].s));
      }
      for (auto& c : comms.c) {  // checked destroy on the normal path (the guard only covers what is left)
        ncclComm_t x = c;
        c = nullptr;
        r.check(r.CommDestroy(x), "ncclCommDestroy");
      }
      g->used_rccl = true;
      g->startup_s[2] = secs(tb, now());
    }
    const auto te = now();
    for (int i = 0; i < n_gpus; ++i) {
      if (q3a_engine_create_from_arena(model_dir, g->devices[i], g->arenas[i], bytes, opts, &g->engines[i]) != 0)
        fail(std::string("q3a_group_create: GPU ") + std::to_string(g->devices[i]) + ": " + q3a_last_error(nullptr));
    }
    g->startup_s[3] = secs(te, now());
    *out = g.release();
    return 0;
  } catch (const std::exception& ex) {
    set_thread_error(ex.what());
    if (out) *out = nullptr;
    return 1;
  }
}

void q3a_group_destroy(q3a_group* g) { delete g; }
int32_t q3a_group_size(const q3a_group* g) { return g ? (int32_t)g->engines.size() : 0; }
int32_t q3a_group_used_rccl(const q3a_group* g) { return g && g->used_rccl ? 1 : 0; }
int32_t q3a_group_startup_seconds(const q3a_group* g, double* out4) {
  if (!g || !out4) return 1;
  for (int i = 0; i < 4; ++i) out4[i] = g->startup_s[i];
  return 0;
}
const char* q3a_group_last_error(const q3a_group* g) { return g ? g->err.c_str() : q3a_last_error(nullptr); }
q3a_engine* q3a_group_engine(q3a_group* g, int32_t rank) {
  return (g && rank >= 0 && rank < (int32_t)g->engines.size()) ? g->engines[rank] : nullptr;
}

int32_t q3a_group_transcribe_ptrs(q3a_group* g, const float* const* pcm16k, const int64_t* n_samples, int32_t B,
                                  const int32_t* lang_prefix_ids, int32_t n_prefix, int32_t max_new, int32_t fixed_new_tokens,
                                  int32_t* out_ids, int32_t stride, int32_t* out_lens) {
  if (!g) return 1;
  if (!pcm16k || !n_samples || B < 1 || !out_ids || !out_lens || stride < 1) { g->err = "q3a_group_transcribe: bad argument"; return 1; }
  const int G = (int)g->engines.size();
  std::vector<std::string> errs(G);
  std::vector<std::thread> th;
  for (int r = 0; r < G; ++r) {
    int32_t b0 = 0, b1 = 0;
    q3a_group_partition(B, G, r, &b0, &b1);
    if (b1 <= b0) continue;  // fewer utterances than GPUs
    th.emplace_back([=, &errs] {
      q3a_engine* e = g->engines[r];
      if (q3a_transcribe_batch_ptrs(e, pcm16k + b0, n_samples + b0, b1 - b0, lang_prefix_ids, n_prefix, max_new, fixed_new_tokens,
                                    out_ids + (size_t)b0 * stride, stride, out_lens + b0) != 0)
        errs[r] = std::string("GPU ") + std::to_string(g->devices[r]) + ": " + q3a_last_error(e);
    });
  }
  for (auto& t : th) t.join();
  for (auto& m : errs)
    if (!m.empty()) { g->err = m; return 1; }
  return 0;
}

int32_t q3a_group_transcribe(q3a_group* g, const float* pcm16k, const int64_t* n_samples, int32_t B,
                             const int32_t* lang_prefix_ids, int32_t n_prefix, int32_t max_new, int32_t fixed_new_tokens,
                             int32_t* out_ids, int32_t stride, int32_t* out_lens) {
  if (!g) return 1;
  if (!pcm16k || !n_samples || B < 1) { g->err = "q3a_group_transcribe: bad argument"; return 1; }
  std::vector<const float*> ptrs((size_t)B);
  int64_t off = 0;
  for (int b = 0; b < B; ++b) { ptrs[b] = pcm16k + off; off += n_samples[b]; }
  return q3a_group_transcribe_ptrs(g, ptrs.data(), n_samples, B, lang_prefix_ids, n_prefix, max_new, fixed_new_tokens, out_ids, stride, out_lens);
}

}  // extern "C"
