// Microbenchmark (not product code): what does an in-kernel barrier among the workgroups of ONE XCD cost, next to the
// dependent kernel boundary it would replace?  ("qkv GEMV + decode attention fused per kv-head group --
// a 32-workgroup cluster barrier, not a grid barrier".)  256 workgroups, one per CU; group g = blockIdx.x % 8 = the 32
// workgroups the dispatcher places on XCD g.  Variants:
//   same-XCD group, workgroup-scope atomics (resolved in that XCD's L2)     -- the cheapest thing the hardware offers
//   same-XCD group, agent-scope atomics + release/acquire fences             -- what the memory model asks for
//   32 consecutive workgroups (spread over all XCDs), agent scope            -- placement control
//   all 256 workgroups, agent scope                                          -- grid barrier
// each with and without a 2 KiB hand-off (every workgroup publishes 64 B before arriving and reads all 32 slices after).
// Spins are bounded: a missing arrival sets a flag instead of hanging the box.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_out/cluster_barrier tools/cluster_barrier.hip && gpurun_out/cluster_barrier
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_empty() {}

// MODE 0: groups of b % 8 (same XCD); 1: groups of b / 32 (all XCDs); 2: one group of 256.  AGENT: scope of atomics + fences.
template <int MODE, bool AGENT, bool PAYLOAD>
__global__ __launch_bounds__(256) void k_barrier(unsigned* cnt, float* slices, float* sink, unsigned epoch, int* failed) {
  const int b = blockIdx.x;
  const int g = MODE == 0 ? (b & 7) : MODE == 1 ? (b >> 5) : 0;
  const int members = MODE == 2 ? 256 : 32;
  const int slot = MODE == 0 ? (b >> 3) : MODE == 1 ? (b & 31) : b;
  unsigned* c = cnt + g * 64;  // one counter per 256 B
  float* mine = slices + ((size_t)g * 256 + slot) * 16;
  if (PAYLOAD && threadIdx.x < 16) mine[threadIdx.x] = (float)(epoch + threadIdx.x);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (AGENT) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    const unsigned target = (epoch + 1) * members;
    int spins = 0;
    for (;;) {
      const unsigned v = AGENT ? __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                               : __hip_atomic_fetch_add(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // RMW: resolved in L2
      if (v >= target) break;
      if (++spins > 20000) { *failed = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (AGENT) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  __syncthreads();
  if (PAYLOAD) {
    const float* grp = slices + (size_t)g * 256 * 16;
    float acc = 0.f;
    for (int i = threadIdx.x; i < members * 16; i += 256) acc += __builtin_nontemporal_load(grp + i);
    if (acc == -1.f) sink[b] = acc;  // (never true: keeps the loads)
  }
}

template <class F>
static int time_graph(const char* name, int n, hipStream_t s, F&& enqueue, double base_us, double* out_us) {
  hipGraph_t g;
  hipGraphExec_t ge;
  CHK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < n; ++i) enqueue(i);
  CHK(hipStreamEndCapture(s, &g));
  CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  CHK(hipEventRecord(a, s));
  CHK(hipGraphLaunch(ge, s));  // ONE replay: the counters are monotonic in the launch index
  CHK(hipEventRecord(b, s));
  CHK(hipStreamSynchronize(s));
  float ms;
  CHK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1e3 / n;
  if (base_us > 0) printf("%-72s %7.3f us/launch   barrier = %+6.3f us over the empty kernel\n", name, us, us - base_us);
  else printf("%-72s %7.3f us/launch\n", name, us);
  if (out_us) *out_us = us;
  CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
  return 0;
}

int main() {
  hipStream_t s;
  CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned* cnt; float *slices, *sink; int* failed;
  CHK(hipMalloc(&cnt, 8 * 64 * 4)); CHK(hipMalloc(&slices, 8 * 256 * 16 * 4)); CHK(hipMalloc(&sink, 256 * 4)); CHK(hipMalloc(&failed, 4));
  CHK(hipMemset(slices, 0, 8 * 256 * 16 * 4)); CHK(hipMemset(failed, 0, 4));
  const int n = 500;
  double base = 0;
  if (time_graph("empty <<<256,256>>> (the dependent kernel boundary)", n, s, [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); }, 0, &base)) return 1;
#define RUN(MODE, AGENT, PAYLOAD, label)                                                                              \
  do {                                                                                                                \
    CHK(hipMemsetAsync(cnt, 0, 8 * 64 * 4, s));                                                                       \
    CHK(hipStreamSynchronize(s));                                                                                     \
    if (time_graph(label, n, s, [&](int i) {                                                                          \
          hipLaunchKernelGGL((k_barrier<MODE, AGENT, PAYLOAD>), dim3(256), dim3(256), 0, s, cnt, slices, sink, (unsigned)i, failed); \
        }, base, nullptr)) return 1;                                                                                  \
    int f = 0;                                                                                                        \
    CHK(hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost));                                                             \
    if (f) { printf("  ^ a spin ran out (barrier never completed): placement assumption wrong or scope too weak\n"); CHK(hipMemset(failed, 0, 4)); } \
  } while (0)
  RUN(0, false, false, "32 WGs of one XCD (b % 8), workgroup-scope atomics");
  RUN(0, false, true, "32 WGs of one XCD, workgroup scope, + 2 KiB hand-off");
  RUN(0, true, false, "32 WGs of one XCD (b % 8), agent-scope atomics + fences");
  RUN(0, true, true, "32 WGs of one XCD, agent scope, + 2 KiB hand-off");
  RUN(1, true, false, "32 consecutive WGs (all XCDs), agent scope");
  RUN(1, true, true, "32 consecutive WGs (all XCDs), agent scope, + 2 KiB hand-off");
  RUN(2, true, false, "all 256 WGs, agent scope (grid barrier)");
  RUN(2, true, true, "all 256 WGs, agent scope, + 16 KiB hand-off");
  return 0;
}
