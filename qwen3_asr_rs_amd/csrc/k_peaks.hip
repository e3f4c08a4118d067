// Measured denominators for the roofline fractions ("Peaks to divide by: measure on the box, don't trust
// datasheets: (i) bf16 MFMA peak via a tuned square GEMM microbench; (ii) HBM via device memcpy / triad").
//
// q3a_measure_peaks runs, on the caller's device and in the caller's process (bench.py calls it next to the timed workload):
//   * hbm_read   a read-only stream over 2 GiB (8x the 256 MB Infinity Cache, so every sweep comes from HBM): 16 B per lane per
//                load, eight loads in flight per lane, non-temporal -- the access pattern of the decode-step weight streams; best of
//                a grid-stride and a contiguous-span form at 4 / 8 / 16 workgroups per CU;
//   * hbm_copy   1 GiB -> 1 GiB (bytes counted both ways), the "device memcpy" number;
//   * hbm_triad  a = b + s * c over three 680 MiB arrays of fp32 (bytes counted three ways);
//   * mfma_bf16  the product's own 256 x 256 x 64 bf16 GEMM (k_gemm256.hip) on 8192^3 with a bf16 output: 2 M N K / time -- on constant
//                operands (the ceiling: nothing toggles, the clock stays high) and on random ones (what real data draws).
// Each is the BEST of `reps` timed launches between two HIP events after one warm-up launch.  None of this is on the product
// path; nothing here is used to compute a transcript.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <string>

#include "../../include/q3asr.h"
#include "dev.h"
#include "kernels.h"

namespace q3a {
namespace {

constexpr int PK_UNROLL = 8;

// bf16 values with a random sign and random mantissa, magnitudes in [0.5, 1) (integer hash of the element index)
__global__ __launch_bounds__(256) void peak_fill_random_kernel(uint16_t* __restrict__ p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned h = (unsigned)i * 2654435761u ^ (unsigned)(i >> 32) ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = (uint16_t)(0x3f00u | (h & 0x7fu) | ((h & 0x80u) << 8));
  }
}

// grid-stride over 16-byte words; every lane keeps PK_UNROLL independent loads in flight
__global__ __launch_bounds__(256) void peak_read_kernel(const uint4* __restrict__ src, size_t n16, unsigned* __restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  unsigned acc = 0;
  for (; i + (PK_UNROLL - 1) * stride < n16; i += PK_UNROLL * stride) {
    uint4 v[PK_UNROLL];
#pragma unroll
    for (int u = 0; u < PK_UNROLL; ++u) v[u] = ld_stream16(src + i + (size_t)u * stride);
#pragma unroll
    for (int u = 0; u < PK_UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  for (; i < n16; i += stride) { const uint4 v = ld_stream16(src + i); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x9e3779b9u) sink[0] = acc;  // (keeps the loads alive; practically never taken)
}

// the same bytes, but a workgroup walks CONTIGUOUS 32 KiB spans (8 x 4 KiB per lane round): the shape of the decode weight streams
// (a wave reads whole rows), with better DRAM page locality than the grid-stride form
__global__ __launch_bounds__(256) void peak_read_span_kernel(const uint4* __restrict__ src, size_t n16, unsigned* __restrict__ sink) {
  constexpr size_t SPAN = (size_t)PK_UNROLL * 256;  // 16-byte words per workgroup round
  unsigned acc = 0;
  for (size_t base = (size_t)blockIdx.x * SPAN; base + SPAN <= n16; base += (size_t)gridDim.x * SPAN) {
    uint4 v[PK_UNROLL];
#pragma unroll
    for (int u = 0; u < PK_UNROLL; ++u) v[u] = ld_stream16(src + base + (size_t)u * 256 + threadIdx.x);
#pragma unroll
    for (int u = 0; u < PK_UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x9e3779b9u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void peak_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (PK_UNROLL - 1) * stride < n16; i += PK_UNROLL * stride) {
    uint4 v[PK_UNROLL];
#pragma unroll
    for (int u = 0; u < PK_UNROLL; ++u) v[u] = ld_stream16(src + i + (size_t)u * stride);
#pragma unroll
    for (int u = 0; u < PK_UNROLL; ++u) __builtin_nontemporal_store(u32x4_t{v[u].x, v[u].y, v[u].z, v[u].w}, reinterpret_cast<u32x4_t*>(dst + i + (size_t)u * stride));
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
__global__ __launch_bounds__(256) void peak_triad_kernel(f32x4_t* __restrict__ a, const f32x4_t* __restrict__ b, const f32x4_t* __restrict__ c,
                                                         float s, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  constexpr int U = PK_UNROLL / 2;
  for (; i + (U - 1) * stride < n16; i += U * stride) {
    f32x4_t vb[U], vc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      vb[u] = __builtin_nontemporal_load(b + i + (size_t)u * stride);
      vc[u] = __builtin_nontemporal_load(c + i + (size_t)u * stride);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(vb[u] + vc[u] * s, a + i + (size_t)u * stride);
  }
  for (; i < n16; i += stride) a[i] = b[i] + c[i] * s;
}

struct Scratch {  // RAII: nothing leaks when a step fails
  void* p = nullptr;
  ~Scratch() { if (p) (void)hipFree(p); }
};
struct Events {
  hipEvent_t a = nullptr, b = nullptr;
  ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};

}  // namespace
}  // namespace q3a

extern "C" int32_t q3a_measure_peaks(int32_t device, int32_t reps, q3a_peaks* out) {
  using namespace q3a;
  static thread_local std::string err;
  if (!out) return 1;
  *out = q3a_peaks{};
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0 || device < 0 || device >= n_dev) return 1;
  if (hipSetDevice(device) != hipSuccess) return 1;
  if (reps < 1) reps = 5;
  int n_cu = 256;
  (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device);
  const size_t total = (size_t)2 << 30;  // 2 GiB of scratch, reused by every stream test and by the GEMM
  Scratch buf;
  if (hipMalloc(&buf.p, total) != hipSuccess) return 1;
  if (hipMemset(buf.p, 0x3c, total) != hipSuccess) return 1;  // 0x3c3c: a small finite bf16 / a finite fp32 pattern
  Events ev;
  if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) return 1;
  hipStream_t s = nullptr;
  auto best_ms = [&](auto&& launch) -> float {
    launch();  // warm-up (clocks, TLB)
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
      (void)hipEventRecord(ev.a, s);
      launch();
      (void)hipEventRecord(ev.b, s);
      if (hipEventSynchronize(ev.b) != hipSuccess) return -1.f;
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev.a, ev.b) != hipSuccess) return -1.f;
      best = std::min(best, ms);
    }
    return best;
  };
  const dim3 grid(n_cu * 8), block(256);
  unsigned* sink = reinterpret_cast<unsigned*>(buf.p);
  {  // read: the best of two access shapes x three grid sizes (a peak is the best the chip does, not the best of one guess)
    const size_t n16 = total / 16;
    float ms = 1e30f;
    for (int per_cu : {4, 8, 16}) {
      const dim3 g(n_cu * per_cu);
      const float a = best_ms([&] { hipLaunchKernelGGL(peak_read_kernel, g, block, 0, s, reinterpret_cast<const uint4*>(buf.p), n16, sink); });
      const float b = best_ms([&] { hipLaunchKernelGGL(peak_read_span_kernel, g, block, 0, s, reinterpret_cast<const uint4*>(buf.p), n16, sink); });
      if (a <= 0.f || b <= 0.f) return 1;
      ms = std::min(ms, std::min(a, b));
    }
    out->hbm_read_gbps = (double)total / (ms * 1e-3) / 1e9;
    out->hbm_read_bytes = (double)total;
  }
  {  // copy
    const size_t half = total / 2, n16 = half / 16;
    const uint4* src = reinterpret_cast<const uint4*>(buf.p);
    uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<char*>(buf.p) + half);
    const float ms = best_ms([&] { hipLaunchKernelGGL(peak_copy_kernel, grid, block, 0, s, src, dst, n16); });
    if (ms <= 0.f) return 1;
    out->hbm_copy_gbps = 2.0 * (double)half / (ms * 1e-3) / 1e9;
  }
  {  // triad
    const size_t third = (total / 3) & ~(size_t)4095, n16 = third / 16;
    char* base = reinterpret_cast<char*>(buf.p);
    const float ms = best_ms([&] {
      hipLaunchKernelGGL(peak_triad_kernel, grid, block, 0, s, reinterpret_cast<f32x4_t*>(base), reinterpret_cast<const f32x4_t*>(base + third),
                         reinterpret_cast<const f32x4_t*>(base + 2 * third), 0.5f, n16);
    });
    if (ms <= 0.f) return 1;
    out->hbm_triad_gbps = 3.0 * (double)third / (ms * 1e-3) / 1e9;
  }
  {  // square bf16 GEMM through the product's 256 x 256 tiles
    const int M = 8192, N = 8192, K = 8192;
    if (hipMemset(buf.p, 0x3c, total) != hipSuccess) return 1;  // the triad left fp32 sums behind: back to small finite bf16
    uint16_t* X = reinterpret_cast<uint16_t*>(buf.p);
    uint16_t* W = X + (size_t)M * K;
    uint16_t* Y = W + (size_t)N * K;  // 3 x 128 MiB
    GemmEpilogue ep;
    ep.out16 = Y; ep.ldo = N;
    const char* kerr = nullptr;
    const float ms = best_ms([&] { if (const char* e = launch_gemm256(X, K, W, M, N, K, ep, false, s)) kerr = e; });
    if (kerr || ms <= 0.f) return 1;
    out->mfma_bf16_tflops = 2.0 * M * (double)N * K / (ms * 1e-3) / 1e12;
    out->gemm_m = M; out->gemm_n = N; out->gemm_k = K;
    // the same launch on random operands: the matrix pipes' power follows the bits that toggle, and under it the chip clocks down
    // (k_gemm256's probe: 1.10-1.16 PFLOP/s on random data where the constant-operand run above reaches 1.4-1.65) -- the
    // denominator to hold real-data GEMM launches against
    hipLaunchKernelGGL(peak_fill_random_kernel, dim3(n_cu * 8), dim3(256), 0, s, X, (size_t)M * K + (size_t)N * K, 0x9e3779b9u);
    const float msr = best_ms([&] { if (const char* e = launch_gemm256(X, K, W, M, N, K, ep, false, s)) kerr = e; });
    if (kerr || msr <= 0.f) return 1;
    out->mfma_bf16_tflops_random = 2.0 * M * (double)N * K / (msr * 1e-3) / 1e12;
  }
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return 1;
  out->n_cu = n_cu;
  out->reps = reps;
  return 0;
}
