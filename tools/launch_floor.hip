// Microbenchmark (not product code): what does one dependent kernel boundary cost on this box, and how does a
// weight-streaming GEMV-shaped kernel's time split into fixed cost + bytes / bandwidth?
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/launch_floor tools/launch_floor.hip && gpurun_out/launch_floor
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "k_gemv.hip"  // the product GEMV kernels, timed in the same harness (compile with -I qwen3_asr_rs_amd/csrc)
#include "k_dattn.hip"
#include "k_skinny.hip"

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__global__ void k_empty() {}
__global__ void k_touch(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
struct BigArgs { const float* x; float* y; const uint16_t* w; int n, k; int pad[30]; };
__global__ void k_bigargs(BigArgs a) { if (threadIdx.x == 0 && blockIdx.x == 0) a.y[0] = a.x[0] + a.pad[29]; }

// y[n] = sum_k x[k] * W[n][k]: 4 waves per block, PR rows per wave, K = 1024 (x in registers), chain through x<->y
template <int PR, bool NT>
__global__ __launch_bounds__(256) void k_gemv(const float* __restrict__ x, const uint16_t* __restrict__ W, float* __restrict__ y, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = blockIdx.x * 4 + wave;
  u32x4_t w[PR][2];
#pragma unroll
  for (int i = 0; i < PR; ++i)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const u32x4_t* p = reinterpret_cast<const u32x4_t*>(W + (size_t)(g * PR + i) * 1024 + lane * 8 + it * 512);
      w[i][it] = NT ? __builtin_nontemporal_load(p) : *p;
    }
  float xs[2][8];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const float4 a = *reinterpret_cast<const float4*>(x + lane * 8 + it * 512), b = *reinterpret_cast<const float4*>(x + lane * 8 + it * 512 + 4);
    xs[it][0] = a.x; xs[it][1] = a.y; xs[it][2] = a.z; xs[it][3] = a.w; xs[it][4] = b.x; xs[it][5] = b.y; xs[it][6] = b.z; xs[it][7] = b.w;
  }
#pragma unroll
  for (int i = 0; i < PR; ++i) {
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const unsigned q[4] = {w[i][it].x, w[i][it].y, w[i][it].z, w[i][it].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s += __uint_as_float(q[e] << 16) * xs[it][2 * e];
        s += __uint_as_float(q[e] & 0xffff0000u) * xs[it][2 * e + 1];
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) y[(g * PR + i) & 1023] = s * 1e-3f;  // keep the chain going: the next launch reads y as its x
  }
}

// reads `bytes` once, 16 B per lane, 16-KiB chunk c from workgroup (c + 1) % gridDim.x
template <bool NT>
__global__ __launch_bounds__(256) void k_prefetch(const uint16_t* __restrict__ w, size_t bytes, int* sink) {
  const size_t chunk = ((size_t)blockIdx.x + gridDim.x - 1) % gridDim.x;
  const char* base = reinterpret_cast<const char*>(w) + chunk * 16384;
  unsigned acc = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const size_t off = chunk * 16384 + (size_t)j * 4096 + threadIdx.x * 16;
    if (off < bytes) {
      const u32x4_t* p = reinterpret_cast<const u32x4_t*>(base + j * 4096 + threadIdx.x * 16);
      const u32x4_t v = NT ? __builtin_nontemporal_load(p) : *p;
      acc ^= v.x ^ v.w;
    }
  }
  if (acc == 0x7fc12345u) *sink = 1;
}

template <class F>
static int time_graph(const char* name, int n, hipStream_t s, F&& enqueue, double bytes_per_launch = 0) {
  hipGraph_t g;
  hipGraphExec_t ge;
  CHK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < n; ++i) enqueue(i);
  CHK(hipStreamEndCapture(s, &g));
  CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  CHK(hipGraphLaunch(ge, s));
  CHK(hipStreamSynchronize(s));
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CHK(hipEventRecord(a, s));
    CHK(hipGraphLaunch(ge, s));
    CHK(hipEventRecord(b, s));
    CHK(hipStreamSynchronize(s));
    float ms;
    CHK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  const double us = best * 1e3 / n;
  if (bytes_per_launch > 0) printf("%-44s %8.3f us/launch  %8.1f GB/s\n", name, us, bytes_per_launch / us * 1e-3);
  else printf("%-44s %8.3f us/launch\n", name, us);
  CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
  return 0;
}

int main() {
  hipStream_t s;
  CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float *x, *y;
  uint16_t* W;
  const size_t WB = (size_t)1 << 30;  // 1 GiB of weights: successive launches walk through it (no cache reuse)
  CHK(hipMalloc(&x, 4096 * 4)); CHK(hipMalloc(&y, 4096 * 4)); CHK(hipMalloc(&W, WB));
  CHK(hipMemset(x, 0, 4096 * 4)); CHK(hipMemset(y, 0, 4096 * 4)); CHK(hipMemset(W, 0x3c, WB));
  const int n = 1000;
  if (time_graph("empty <<<1,64>>>", n, s, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); })) return 1;
  if (time_graph("empty <<<256,256>>>", n, s, [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); })) return 1;
  if (time_graph("empty <<<1024,256>>>", n, s, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s); })) return 1;
  if (time_graph("touch one float <<<256,256>>>", n, s, [&](int) { hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, s, x); })) return 1;
  BigArgs ba{}; ba.x = x; ba.y = y;
  if (time_graph("160-byte kernarg, 1 load+store <<<256,256>>>", n, s, [&](int) { hipLaunchKernelGGL(k_bigargs, dim3(256), dim3(256), 0, s, ba); })) return 1;
  for (int N : {1024, 2048, 4096, 6144, 8192, 16384, 32768}) {
    const double bytes = (double)N * 1024 * 2;
    const size_t stride = (size_t)N * 1024;  // elements
    const int slots = (int)(WB / 2 / stride);
    char name[96];
    auto run = [&](auto kern, int pr, const char* tag) {
      snprintf(name, sizeof name, "gemv N=%d K=1024 PR=%d %s (%.1f MB)", N, pr, tag, bytes * 1e-6);
      return time_graph(name, n, s, [&](int i) {
        const float* xi = (i & 1) ? y : x;
        float* yi = (i & 1) ? x : y;
        hipLaunchKernelGGL(kern, dim3(N / (4 * pr)), dim3(256), 0, s, xi, W + (size_t)(i % slots) * stride, yi, N);
      }, bytes);
    };
    if (run(k_gemv<1, false>, 1, "plain")) return 1;
    if (run(k_gemv<2, false>, 2, "plain")) return 1;
    if (run(k_gemv<2, true>, 2, "nt")) return 1;
    if (run(k_gemv<4, true>, 4, "nt")) return 1;
  }
  // the product GEMV (k_gemv.hip) on the decode shapes of the 0.6B model, same chain structure
  {
    float* rmsw; float* big;
    CHK(hipMalloc(&rmsw, 8192 * 4)); CHK(hipMalloc(&big, 8192 * 4));
    CHK(hipMemset(rmsw, 0, 8192 * 4)); CHK(hipMemset(big, 0, 8192 * 4));
    struct Cfg { const char* name; int N, K, mode; bool rms; };
    const Cfg cfgs[] = {{"qkv   N=4096 K=1024 rms", 4096, 1024, 0, true}, {"qkv   N=4096 K=1024 no-rms", 4096, 1024, 0, false},
                        {"gateup N=6144 K=1024 rms GLU", 6144, 1024, 2, true}, {"gateup N=6144 K=1024 no-rms GLU", 6144, 1024, 2, false},
                        {"plain N=6144 K=1024 rms", 6144, 1024, 0, true}, {"plain N=6144 K=1024 no-rms", 6144, 1024, 0, false}, {"o    N=1024 K=2048 resid", 1024, 2048, 1, false},
                        {"down  N=1024 K=3072 resid", 1024, 3072, 1, false}};
    for (const Cfg& c : cfgs) {
      const double bytes = (double)c.N * c.K * 2;
      const size_t stride = (size_t)c.N * c.K;
      const int slots = (int)(WB / 2 / stride);
      char name[96];
      snprintf(name, sizeof name, "product gemv %s (%.1f MB)", c.name, bytes * 1e-6);
      if (time_graph(name, n, s, [&](int i) {
            q3a::GemvArgs a{};
            a.x = (i & 1) ? big : x; a.ldx = c.K; a.rms_w = c.rms ? rmsw : nullptr; a.eps = 1e-6f;
            a.W = W + (size_t)(i % slots) * stride; a.N = c.N; a.K = c.K; a.mode = c.mode;
            a.out = (i & 1) ? x : big; a.ldo = c.N; a.resid = a.out;
            const char* e = q3a::launch_gemv(a, 1, s);
            if (e) printf("launch_gemv: %s\n", e);
          }, bytes)) return 1;
    }
  }
  // o_proj with the flash-decoding merge fused in (16 heads x nsplit partials), chained through the partial buffers
  for (int nsplit : {4, 8}) {
    float *pa, *pb, *pm, *pl;
    const int heads = 16;
    CHK(hipMalloc(&pa, (size_t)heads * nsplit * 128 * 4)); CHK(hipMalloc(&pb, (size_t)heads * nsplit * 128 * 4));
    CHK(hipMalloc(&pm, heads * nsplit * 4)); CHK(hipMalloc(&pl, heads * nsplit * 4));
    CHK(hipMemset(pa, 0, (size_t)heads * nsplit * 128 * 4)); CHK(hipMemset(pb, 0, (size_t)heads * nsplit * 128 * 4));
    std::vector<float> ones(heads * nsplit, 1.0f);
    CHK(hipMemcpy(pl, ones.data(), ones.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemset(pm, 0, heads * nsplit * 4));
    const int N = 1024, K = 2048;
    const size_t stride = (size_t)N * K;
    const int slots = (int)(WB / 2 / stride);
    char name[96];
    snprintf(name, sizeof name, "product gemv o N=1024 K=2048 merge nsplit=%d", nsplit);
    if (time_graph(name, n, s, [&](int i) {
          q3a::GemvArgs a{};
          a.ldx = K; a.W = W + (size_t)(i % slots) * stride; a.N = N; a.K = K; a.mode = 1;
          a.attn_pm = pm; a.attn_pl = pl; a.attn_po = (i & 1) ? pb : pa; a.attn_nsplit = nsplit; a.attn_heads = heads;
          a.out = (i & 1) ? pa : pb; a.ldo = N; a.resid = a.out;
          const char* e = q3a::launch_gemv(a, 1, s);
          if (e) printf("launch_gemv: %s\n", e);
        }, (double)N * K * 2)) return 1;
  }
  // skinny MFMA GEMM (k_skinny.hip), 32 sequences, the decode shapes of the 0.6B model
  {
    float *xf, *yf, *rmsw;
    uint16_t* xh;
    CHK(hipMalloc(&xf, 32 * 4096 * 4)); CHK(hipMalloc(&yf, 32 * 6144 * 4)); CHK(hipMalloc(&xh, 32 * 4096 * 2)); CHK(hipMalloc(&rmsw, 4096 * 4));
    CHK(hipMemset(xf, 0, 32 * 4096 * 4)); CHK(hipMemset(yf, 0, 32 * 6144 * 4)); CHK(hipMemset(xh, 0, 32 * 4096 * 2)); CHK(hipMemset(rmsw, 0, 4096 * 4));
    if (const char* e = q3a::skinny_init()) printf("skinny_init: %s\n", e);
    struct Cfg { const char* name; int N, K, mode, xmode; };
    const Cfg cfgs[] = {{"qkv  N=4096 K=1024 fp32 x + fused norm", 4096, 1024, 0, 1}, {"qkv  N=4096 K=1024 fp32 x", 4096, 1024, 0, 0},
                        {"qkv  N=4096 K=1024 bf16 x", 4096, 1024, 0, 2}, {"gateup N=6144 K=1024 GLU fused norm", 6144, 1024, 2, 1},
                        {"o    N=1024 K=2048 bf16 x resid", 1024, 2048, 1, 2}, {"down N=1024 K=3072 bf16 x resid", 1024, 3072, 1, 2},
                        {"down N=1024 K=3072 fp32 x resid", 1024, 3072, 1, 0},
                        {"o    N=1024 K=2048 bf16 x fragment order", 1024, 2048, 1, 3}, {"down N=1024 K=3072 bf16 x fragment order", 1024, 3072, 1, 3},
                        {"qkv  N=4096 K=1024 bf16 x fragment order", 4096, 1024, 0, 3}};
    for (const Cfg& c : cfgs) {
      const double bytes = (double)c.N * c.K * 2;
      const size_t stride = (size_t)c.N * c.K;
      const int slots = (int)(WB / 2 / stride);
      char name[96];
      snprintf(name, sizeof name, "skinny S=32 %s (%.1f MB)", c.name, bytes * 1e-6);
      if (time_graph(name, n, s, [&](int i) {
            q3a::SkinnyArgs a{};
            a.x = xf; a.x16 = c.xmode >= 2 ? xh : nullptr; a.x16_frag = c.xmode == 3; a.ldx = c.K; a.S = 32; a.rms_w = c.xmode == 1 ? rmsw : nullptr; a.eps = 1e-6f;
            a.W = W + (size_t)(i % slots) * stride; a.N = c.N; a.K = c.K; a.mode = c.mode; a.out = yf; a.ldo = c.mode == 2 ? c.N / 2 : c.N; a.resid = yf;
            const char* e = q3a::launch_skinny(a, false, s);
            if (e) printf("launch_skinny: %s\n", e);
          }, bytes)) return 1;
    }
  }
  // decode attention (k_dattn.hip): 16 q heads / 8 kv heads, context 450 of 512, cold bf16 cache (cycled through W)
  {
    const int n_q = 16, n_kv = 8, max_ctx = 512, nsplit = 4;
    float *qkv, *cs, *sn, *nw, *pm, *pl, *po;
    int* pos;
    CHK(hipMalloc(&qkv, 4096 * 4)); CHK(hipMalloc(&cs, 512 * 64 * 4)); CHK(hipMalloc(&sn, 512 * 64 * 4)); CHK(hipMalloc(&nw, 128 * 4));
    CHK(hipMalloc(&pm, n_q * nsplit * 4)); CHK(hipMalloc(&pl, n_q * nsplit * 4)); CHK(hipMalloc(&po, n_q * nsplit * 128 * 4));
    CHK(hipMalloc(&pos, 4));
    CHK(hipMemset(qkv, 0, 4096 * 4)); CHK(hipMemset(cs, 0, 512 * 64 * 4)); CHK(hipMemset(sn, 0, 512 * 64 * 4)); CHK(hipMemset(nw, 0, 128 * 4));
    const int p450 = 450;
    CHK(hipMemcpy(pos, &p450, 4, hipMemcpyHostToDevice));
    const size_t layer_elems = (size_t)n_kv * max_ctx * 128;  // bf16 elements of one layer's K (or V)
    const int layers = 200;
    if (time_graph("product decode_attn ctx=450 nsplit=4 (cold KV)", n, s, [&](int i) {
          q3a::DecodeAttnArgs a{};
          a.qkv = qkv; a.pos = pos; a.q_norm = nw; a.k_norm = nw; a.eps = 1e-6f; a.rope_cur = cs;
          a.kcache = W + (size_t)(i % layers) * layer_elems; a.vcache = W + (size_t)(layers + i % layers) * layer_elems;
          a.pm = pm; a.pl = pl; a.po = po; a.nsplit = nsplit; a.n_q = n_q; a.n_kv = n_kv; a.max_ctx = max_ctx; a.scale_div = 11.3137f;
          const char* e = q3a::launch_decode_attn(a, 1, false, s);
          if (e) printf("launch_decode_attn: %s\n", e);
        })) return 1;
    // pair: qkv GEMV (optionally prefetching the layer's cache) followed by the attention that reads it
    for (int pf = 0; pf < 1; ++pf) {
      float* xin; int* sink;
      CHK(hipMalloc(&xin, 4096 * 4)); CHK(hipMalloc(&sink, 4)); CHK(hipMemset(xin, 0, 4096 * 4));
      const size_t wstride = (size_t)4096 * 1024;
      if (time_graph(pf ? "pair qkv-GEMV(+KV prefetch) + decode_attn (cold KV)" : "pair qkv-GEMV + decode_attn (cold KV)", n / 2, s, [&](int i) {
            void* kc = W + (size_t)(i % layers) * layer_elems;
            void* vc = W + (size_t)(layers + i % layers) * layer_elems;
            q3a::GemvArgs g{};
            g.x = xin; g.ldx = 1024; g.rms_w = nw; g.eps = 1e-6f; g.W = W + (size_t)(2 * layers) * layer_elems + (size_t)(i % 64) * wstride;
            g.N = 4096; g.K = 1024; g.mode = 0; g.out = qkv; g.ldo = 4096;
            const char* e = q3a::launch_gemv(g, 1, s);
            if (e) printf("launch_gemv: %s\n", e);
            q3a::DecodeAttnArgs a{};
            a.qkv = qkv; a.pos = pos; a.q_norm = nw; a.k_norm = nw; a.eps = 1e-6f; a.rope_cur = cs;
            a.kcache = kc; a.vcache = vc;
            a.pm = pm; a.pl = pl; a.po = po; a.nsplit = nsplit; a.n_q = n_q; a.n_kv = n_kv; a.max_ctx = max_ctx; a.scale_div = 11.3137f;
            e = q3a::launch_decode_attn(a, 1, false, s);
            if (e) printf("launch_decode_attn: %s\n", e);
          })) return 1;
    }
    if (time_graph("product decode_attn ctx=450 nsplit=4 (warm KV)", n, s, [&](int i) {
          q3a::DecodeAttnArgs a{};
          a.qkv = qkv; a.pos = pos; a.q_norm = nw; a.k_norm = nw; a.eps = 1e-6f; a.rope_cur = cs;
          a.kcache = W; a.vcache = W + layer_elems;
          a.pm = pm; a.pl = pl; a.po = po; a.nsplit = nsplit; a.n_q = n_q; a.n_kv = n_kv; a.max_ctx = max_ctx; a.scale_div = 11.3137f;
          const char* e = q3a::launch_decode_attn(a, 1, false, s);
          if (e) printf("launch_decode_attn: %s\n", e);
        })) return 1;
  }
  // does a weight matrix that another kernel has just pulled through the Infinity Cache stream faster?  The prefetch
  // reads chunk c from workgroup c+1 (another XCD than the GEMV workgroup that consumes it: L2 misses, MALL hits)
  {
    const int N = 6144, K = 1024;
    const size_t stride = (size_t)N * K;
    const int slots = (int)(WB / 2 / stride);
    float* rmsw; float* big; int* sink;
    CHK(hipMalloc(&rmsw, 8192 * 4)); CHK(hipMalloc(&big, 8192 * 4)); CHK(hipMalloc(&sink, 4));
    CHK(hipMemset(rmsw, 0, 8192 * 4)); CHK(hipMemset(big, 0, 8192 * 4));
    auto gemv = [&](int i, const uint16_t* w) {
      q3a::GemvArgs a{};
      a.x = (i & 1) ? big : x; a.ldx = K; a.rms_w = rmsw; a.eps = 1e-6f; a.W = w; a.N = N; a.K = K; a.mode = 2;
      a.out = (i & 1) ? x : big; a.ldo = N / 2;
      const char* e = q3a::launch_gemv(a, 1, s);
      if (e) printf("launch_gemv: %s\n", e);
    };
    for (int nt = 0; nt < 2; ++nt) {
      auto pf = [&](const uint16_t* w) {
        if (nt) hipLaunchKernelGGL((k_prefetch<true>), dim3(768), dim3(256), 0, s, w, stride * 2, sink);
        else hipLaunchKernelGGL((k_prefetch<false>), dim3(768), dim3(256), 0, s, w, stride * 2, sink);
      };
      if (time_graph(nt ? "prefetch kernel alone 12.6 MB (nt loads)" : "prefetch kernel alone 12.6 MB (plain loads)", n, s,
                     [&](int i) { pf(W + (size_t)(i % slots) * stride); }, stride * 2.0)) return 1;
      if (time_graph(nt ? "prefetch(nt) + gateup GEMV on the same matrix" : "prefetch(plain) + gateup GEMV on the same matrix", n / 2, s,
                     [&](int i) { const uint16_t* w = W + (size_t)(i % slots) * stride; pf(w); gemv(i, w); }, stride * 2.0)) return 1;
    }
    if (time_graph("gateup GEMV twice on the same matrix (2nd: L2/MALL warm)", n / 2, s,
                   [&](int i) { const uint16_t* w = W + (size_t)(i % slots) * stride; gemv(i, w); gemv(i + 1, w); }, stride * 4.0)) return 1;
  }
  // eager (no graph) chain for comparison
  {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
      CHK(hipEventRecord(a, s));
      for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s);
      CHK(hipEventRecord(b, s));
      CHK(hipStreamSynchronize(s));
      float ms;
      CHK(hipEventElapsedTime(&ms, a, b));
      if (rep) printf("%-44s %8.3f us/launch\n", "eager empty <<<256,256>>>", ms * 1e3 / n);
    }
  }
  return 0;
}
