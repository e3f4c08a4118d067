// Race screen for the LDS-DMA rings of k_gemm16.hip: the same GEMM launched over and over on one stream while a second
// stream saturates HBM (a copy kernel), every result compared bit for bit with the first one.  A read of a stage whose
// DMA has not landed shows up as a mismatch count > 0 under memory load and 0 on an idle chip.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I qwen3_asr_rs_amd/csrc -o build/tools/gemm16_race tools/gemm16_race.hip
//   build/tools/gemm16_race [ring]     (default: the two-stage kernels; "ring": knob gemm16_ring = 1)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include "k_gemm16.hip"
#include "k_gemm256.hip"
namespace q3a {
Knobs& knobs() { static Knobs k; return k; }
const char* launch_qknorm_rope_kv(const RopeKvArgs&, int, bool, hipStream_t) { return "not linked"; }
}
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define KCHK(x) do { const char* m_ = (x); if (m_) { printf("launch error: %s (line %d)\n", m_, __LINE__); exit(1); } } while (0)
__global__ void hog_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void diff_kernel(const uint32_t* a, const uint32_t* b, size_t n, unsigned* count) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (a[i] != b[i]) atomicAdd(count, 1u);
}
int main(int argc, char** argv) {
  hipStream_t s, h;
  CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CHK(hipStreamCreateWithFlags(&h, hipStreamNonBlocking));
  q3a::knobs().gemm256_min_tiles = 1 << 30;
  const size_t HOG = (size_t)1 << 30;
  uint4 *hs, *hd;
  CHK(hipMalloc(&hs, HOG)); CHK(hipMalloc(&hd, HOG)); CHK(hipMemset(hs, 1, HOG));
  struct Shape { const char* name; int M, N, K; bool glu; };
  const Shape shapes[] = {{"405 x 4096 x 1024 (64x64 tiles, ring of 4)", 405, 4096, 1024, false}, {"405 x 6144 x 1024 GLU (64x64, ring of 3)", 405, 6144, 1024, true},
                          {"390 x 2688 x 896 (32x64 tiles, ring of 4)", 390, 2688, 896, false}, {"390 x 896 x 896 (K split, ring of 4 x 128)", 390, 896, 896, false},
                          {"405 x 1024 x 2048 (K split, two stages of 256)", 405, 1024, 2048, false}};
  const bool ring_on = argc > 1 && !strcmp(argv[1], "ring");
  q3a::knobs().gemm16_ring = ring_on ? 1 : 0;
  printf("%s\n", ring_on ? "rings of 3-4 LDS stages (gemm16_ring = 1)" : "two LDS stages everywhere (default)");
  for (const Shape& sh : shapes) {
    const int N_out = sh.glu ? sh.N / 2 : sh.N;
    uint16_t *X, *W; float *Y0, *Y; unsigned* cnt;
    CHK(hipMalloc(&X, (size_t)sh.M * sh.K * 2)); CHK(hipMalloc(&W, (size_t)sh.N * sh.K * 2));
    CHK(hipMalloc(&Y0, (size_t)sh.M * N_out * 4)); CHK(hipMalloc(&Y, (size_t)sh.M * N_out * 4)); CHK(hipMalloc(&cnt, 4));
    std::vector<uint16_t> hx((size_t)sh.M * sh.K), hw((size_t)sh.N * sh.K);
    uint32_t r = 12345u;
    union FU { float f; uint32_t u; };
    auto rnd = [&]() { r = r * 1664525u + 1013904223u; FU x; x.f = ((r >> 8) & 0xffff) / 65536.0f - 0.5f; return (uint16_t)(x.u >> 16); };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd();
    CHK(hipMemcpy(X, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CHK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    q3a::GemmEpilogue ep; ep.out = Y0; ep.ldo = N_out;
    KCHK(q3a::launch_gemm16_small(X, sh.K, W, sh.M, sh.N, sh.K, ep, sh.glu, s));
    CHK(hipStreamSynchronize(s));
    for (int load = 0; load < 2; ++load) {
      CHK(hipMemset(cnt, 0, 4));
      const int REPS = 400;
      ep.out = Y;
      for (int i = 0; i < REPS; ++i) {
        if (load && i % 4 == 0) hipLaunchKernelGGL(hog_kernel, dim3(2048), dim3(256), 0, h, hs, hd, HOG / 16 / 8);
        KCHK(q3a::launch_gemm16_small(X, sh.K, W, sh.M, sh.N, sh.K, ep, sh.glu, s));
        hipLaunchKernelGGL(diff_kernel, dim3(256), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(Y0), reinterpret_cast<const uint32_t*>(Y), (size_t)sh.M * N_out, cnt);
      }
      CHK(hipStreamSynchronize(s)); CHK(hipStreamSynchronize(h));
      unsigned c = 0; CHK(hipMemcpy(&c, cnt, 4, hipMemcpyDeviceToHost));
      printf("  %-52s %s: %u mismatching words over %d launches\n", sh.name, load ? "under HBM load" : "idle chip     ", c, REPS);
    }
    CHK(hipFree(X)); CHK(hipFree(W)); CHK(hipFree(Y0)); CHK(hipFree(Y)); CHK(hipFree(cnt));
  }
  return 0;
}
