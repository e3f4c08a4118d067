// Prototype / microbenchmark (not product code): how long does ONE batch-1 decoder layer take as TWO launches with the
// hand-offs kept inside each XCD, next to the same arithmetic as five dependent launches?
//
// Background (profiles/r2_cluster_barrier.txt): a dependent kernel boundary costs 1.55 us, an in-kernel
// barrier among the 32 workgroups of one XCD 0.3 us (+0.4 with a 2 KiB hand-off), and the product's batch-1 layer is five
// launches = 20.2 us.  With 8 kv heads = 8 XCDs every all-to-all edge of the layer can be cut at an XCD boundary if o_proj
// and down are computed as per-XCD PARTIAL sums that the next launch adds up (fixed order: deterministic):
//
//   launch A (XCD g = kv head g):  x = resid + sum of the 8 down partials of the previous layer; RMSNorm;
//       the 512 qkv rows of head g (16 per workgroup)  | in-XCD barrier |  4 key-split workgroups: attention of head g
//       | in-XCD barrier |  o_proj partial: W_o[:, 256 g .. +256] . ctx_g  (32 output rows per workgroup) -> P_o[g][H]
//   launch B (XCD g = slice g of the MLP):  x = resid + sum of the 8 o_proj partials; RMSNorm;
//       gate/up rows of act[I/8 * g .. +I/8] (12 act values per workgroup)  | in-XCD barrier |
//       down partial: W_down[:, I/8 * g .. +I/8] . act_g  -> P_d[g][H]
//   The weight rows of the phase AFTER a barrier are requested BEFORE it (they do not depend on it).
//
// This file measures the structure with the real byte streams (0.6B shapes: H 1024, 16 q heads, 8 kv heads, I 3072, 450 keys,
// distinct weights per layer so nothing is cache-resident) and arithmetic of the right size; the attention is simplified (no
// RoPE / QK-norm, one softmax pass) and results are only checked for finiteness -- it answers "what does the launch
// structure cost", not "is this the product kernel".
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/tools/layer2_proto tools/layer2_proto.hip && build/tools/layer2_proto
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <cmath>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int H = 1024, NQ = 16, NKV = 8, QKV = (NQ + 2 * NKV) * 128, QD = NQ * 128, I = 3072, CTX = 450, MAXC = 512, LAYERS = 28;
constexpr int NSPLIT = 4;

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float dot8(const uint4& w, const float* x) {
  return bf_lo(w.x) * x[0] + bf_hi(w.x) * x[1] + bf_lo(w.y) * x[2] + bf_hi(w.y) * x[3] + bf_lo(w.z) * x[4] + bf_hi(w.z) * x[5] +
         bf_lo(w.w) * x[6] + bf_hi(w.w) * x[7];
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float half_sum(float v) {  // over the 32 lanes of a half wave
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct Layer {
  const uint16_t *wqkv, *wo, *wgu, *wd;  // [QKV][H], [H][QD], [2I][H] (gate rows then up rows), [H][I]
  const uint16_t *kc, *vc;               // [NKV][MAXC][128]
};
struct Bufs {
  float* x;        // [H] residual at the layer input
  float* p_o;      // [8][H] o_proj partials
  float* p_d;      // [8][H] down partials
  float* qkv;      // [QKV]
  float* ctx;      // [QD]
  float* act;      // [I]
  float* part;     // [NKV][2][NSPLIT][130] attention partials (m, l, o[128])
  unsigned* sync;  // [8][64]
  int* failed;
  long long* stamps;  // [2 workgroups][2 launches][8] wall_clock64 (100 MHz) of layer 5: phase boundaries of workgroups 0 and 255
};

#define STAMP(kind, i) do { if (epoch == 5 && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 255)) \
    b.stamps[((blockIdx.x == 0 ? 0 : 1) * 2 + (kind)) * 8 + (i)] = wall_clock64(); } while (0)

// ---- in-XCD barrier (tools/cluster_barrier.hip; poll = agent-scope load, arrive = L2-resolved RMW, bounded) ----
__device__ __forceinline__ void xcd_barrier(unsigned* cnt, unsigned target, int* failed) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    int spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > 20000) { *failed = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// x = resid + sum of 8 partials (np = 0: resid only); returns 1/rms; this lane's 8 * KI elements of x end up in xs (LDS)
__device__ __forceinline__ float load_x(const float* resid, const float* parts, int np, float* xs) {
  float ss = 0.f;
  for (int k = threadIdx.x; k < H; k += blockDim.x) {
    float v = resid[k];
    for (int p = 0; p < np; ++p) v += parts[p * H + k];
    xs[k] = v;
    ss += v * v;
  }
  __shared__ float red[8];
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
  return 1.0f / sqrtf(t / H + 1e-6f);
}

// rows [row0, row0 + n) of W[.][H] against xs, one row per wave at a time; out[r] = dot * rstd
__device__ __forceinline__ void gemv_rows(const uint16_t* W, int row0, int n, const float* xs, float rstd, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int r = wave; r < n; r += nw) {
    const uint4* w = reinterpret_cast<const uint4*>(W + (size_t)(row0 + r) * H);
    float a = 0.f;
    for (int c = lane; c < H / 8; c += 64) a += dot8(w[c], xs + c * 8);
    a = wave_sum(a);
    if (lane == 0) out[row0 + r] = a * rstd;
  }
}

// attention of kv head g, key split sp (128 keys), both q heads; writes (m, l, o[128]) partials
__device__ __forceinline__ void attn_split(const Layer& L, const float* qkv, int g, int sp, float* part, float* lds /* >= 2*130*8 + 512 */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & 15, kq = lane >> 4;
  const uint16_t* kc = L.kc + (size_t)g * MAXC * 128;
  const uint16_t* vc = L.vc + (size_t)g * MAXC * 128;
  uint4 kr[4], vr[4];
  int keys[4];
  for (int i = 0; i < 4; ++i) {
    keys[i] = sp * 128 + wave * 16 + kq + i * 4;
    const int kk = keys[i] < MAXC ? keys[i] : MAXC - 1;
    kr[i] = *reinterpret_cast<const uint4*>(kc + (size_t)kk * 128 + sub * 8);
    vr[i] = *reinterpret_cast<const uint4*>(vc + (size_t)kk * 128 + sub * 8);
  }
  float* wm = lds;              // [8 waves][2]
  float* wl = lds + 16;         // [8][2]
  float* wo = lds + 32;         // [8][2][128]
  for (int h = 0; h < 2; ++h) {
    const float* q = qkv + (g * 2 + h) * 128 + sub * 8;
    float sc[4], mx = -INFINITY;
    for (int i = 0; i < 4; ++i) {
      float p = dot8(kr[i], q);
      for (int o = 8; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
      sc[i] = keys[i] < CTX ? p * 0.0883883f : -INFINITY;
      mx = fmaxf(mx, sc[i]);
    }
    for (int o = 32; o >= 16; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float l = 0.f, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
      const float p = mx == -INFINITY ? 0.f : __expf(sc[i] - mx);
      l += p;
      const uint32_t vw[4] = {vr[i].x, vr[i].y, vr[i].z, vr[i].w};
      for (int e = 0; e < 4; ++e) { acc[2 * e] += p * bf_lo(vw[e]); acc[2 * e + 1] += p * bf_hi(vw[e]); }
    }
    for (int o = 32; o >= 16; o >>= 1) {
      l += __shfl_xor(l, o, 64);
      for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
    }
    if (lane == 0) { wm[wave * 2 + h] = mx; wl[wave * 2 + h] = l; }
    if (kq == 0) for (int e = 0; e < 8; ++e) wo[(wave * 2 + h) * 128 + sub * 8 + e] = acc[e];
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    const int h = threadIdx.x >> 7, d = threadIdx.x & 127;
    float M = -INFINITY;
    for (int w = 0; w < 8; ++w) M = fmaxf(M, wm[w * 2 + h]);
    float Ls = 0.f, o = 0.f;
    for (int w = 0; w < 8; ++w) {
      const float f = wm[w * 2 + h] == -INFINITY ? 0.f : __expf(wm[w * 2 + h] - M);
      Ls += wl[w * 2 + h] * f;
      o += wo[(w * 2 + h) * 128 + d] * f;
    }
    float* pp = part + ((size_t)(g * 2 + h) * NSPLIT + sp) * 130;
    if (d == 0) { pp[0] = M; pp[1] = Ls; }
    pp[2 + d] = o;
  }
}

// merged attention output of kv head g (256 floats) from the split partials, into LDS
__device__ __forceinline__ void merge_ctx(const float* part, int g, float* ctx_s) {
  if (threadIdx.x < 256) {
    const int h = threadIdx.x >> 7, d = threadIdx.x & 127;
    const float* pp = part + (size_t)(g * 2 + h) * NSPLIT * 130;
    float M = -INFINITY;
    for (int s = 0; s < NSPLIT; ++s) M = fmaxf(M, pp[s * 130]);
    float Ls = 0.f, o = 0.f;
    for (int s = 0; s < NSPLIT; ++s) {
      const float f = pp[s * 130] == -INFINITY ? 0.f : __expf(pp[s * 130] - M);
      Ls += pp[s * 130 + 1] * f;
      o += pp[s * 130 + 2 + d] * f;
    }
    ctx_s[threadIdx.x] = o / Ls;
  }
  __syncthreads();
}

// partial GEMV: out[r] = W[r][c0 .. c0+ncol) . v for the workgroup's rows [r0, r0+32); chunks were requested up front
template <int NCH>  // 16-B chunks per thread: 32 rows * ncol * 2 B / 16 / 512
struct PartRows {
  uint4 w[NCH];
  __device__ __forceinline__ void request(const uint16_t* W, int ld, int r0, int c0, int ncol) {
    const int cpr = ncol / 8;
    for (int j = 0; j < NCH; ++j) {
      const int c = threadIdx.x + j * 512, r = c / cpr, cc = c % cpr;
      w[j] = *reinterpret_cast<const uint4*>(W + (size_t)(r0 + r) * ld + c0 + cc * 8);
    }
  }
  __device__ __forceinline__ void finish(const float* v_s, int ncol, float* acc_s /* [32] zeroed */, float* out, int r0) {
    const int cpr = ncol / 8;
    for (int j = 0; j < NCH; ++j) {
      const int c = threadIdx.x + j * 512, r = c / cpr, cc = c % cpr;
      atomicAdd(&acc_s[r], dot8(w[j], v_s + cc * 8));  // (prototype: order not fixed; the product would reduce in a fixed tree)
    }
    __syncthreads();
    if (threadIdx.x < 32) out[r0 + threadIdx.x] = acc_s[threadIdx.x];
  }
};

// ------------------------------------------------------------------------------------------------------------------
// launch A and B of the two-launch layer
__global__ __launch_bounds__(512) void k_layer_a(Layer L, Bufs b, const float* resid, int np_in, float* x_out, unsigned epoch) {
  __shared__ float xs[H];
  __shared__ float lds[32 + 8 * 2 * 128];
  __shared__ float ctx_s[256];
  __shared__ float acc_s[32];
  const int g = blockIdx.x & 7, slot = blockIdx.x >> 3;
  unsigned* cnt = b.sync + g * 64;
  STAMP(0, 0);
  PartRows<2> orows;                                        // 32 rows x 256 cols x 2 B = 16 KiB per workgroup
  orows.request(L.wo, QD, slot * 32, g * 256, 256);         // does not depend on anything: in flight across both barriers
  const float rstd = load_x(resid, b.p_d, np_in, xs);
  STAMP(0, 1);
  if (slot == 0) for (int k = threadIdx.x; k < H / 8; k += 512) x_out[g * (H / 8) + k] = xs[g * (H / 8) + k];  // new residual, one slice per XCD
  if (threadIdx.x < 32) acc_s[threadIdx.x] = 0.f;
  // 512 rows of kv head g: q rows [256 g, +256), k rows [2048 + 128 g, +128), v rows [3072 + 128 g, +128); 16 per workgroup
  {
    const int r = slot * 16;
    const int row0 = r < 256 ? g * 256 + r : r < 384 ? NQ * 128 + g * 128 + (r - 256) : (NQ + NKV) * 128 + g * 128 + (r - 384);
    gemv_rows(L.wqkv, row0, 16, xs, rstd, b.qkv);
  }
  STAMP(0, 2);
  xcd_barrier(cnt, (2 * epoch + 1) * 32, b.failed);
  STAMP(0, 3);
  if (slot < NSPLIT) attn_split(L, b.qkv, g, slot, b.part, lds);
  STAMP(0, 4);
  xcd_barrier(cnt, (2 * epoch + 2) * 32, b.failed);
  STAMP(0, 5);
  merge_ctx(b.part, g, ctx_s);
  orows.finish(ctx_s, 256, acc_s, b.p_o + g * H, slot * 32);
  STAMP(0, 6);
}

__global__ __launch_bounds__(512) void k_layer_b(Layer L, Bufs b, const float* resid, float* x_out, unsigned epoch) {
  __shared__ float xs[H];
  __shared__ float act_s[I / 8];
  __shared__ float gu_s[24];
  __shared__ float acc_s[32];
  const int g = blockIdx.x & 7, slot = blockIdx.x >> 3;
  unsigned* cnt = b.sync + g * 64 + 16;
  STAMP(1, 0);
  PartRows<3> drows;                                        // 32 rows x 384 cols x 2 B = 24 KiB per workgroup
  drows.request(L.wd, I, slot * 32, g * (I / 8), I / 8);
  const float rstd = load_x(resid, b.p_o, 8, xs);
  STAMP(1, 1);
  if (slot == 0) for (int k = threadIdx.x; k < H / 8; k += 512) x_out[g * (H / 8) + k] = xs[g * (H / 8) + k];
  if (threadIdx.x < 32) acc_s[threadIdx.x] = 0.f;
  // 12 act values per workgroup: gate rows j, up rows I + j
  const int j0 = g * (I / 8) + slot * 12;
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < 24; r += 8) {
      const int row = r < 12 ? j0 + r : I + j0 + (r - 12);
      const uint4* w = reinterpret_cast<const uint4*>(L.wgu + (size_t)row * H);
      float a = 0.f;
      for (int c = lane; c < H / 8; c += 64) a += dot8(w[c], xs + c * 8);
      a = wave_sum(a);
      if (lane == 0) gu_s[r] = a * rstd;
    }
  }
  __syncthreads();
  if (threadIdx.x < 12) {
    const float gt = gu_s[threadIdx.x], up = gu_s[12 + threadIdx.x];
    b.act[j0 + threadIdx.x] = gt / (1.0f + __expf(-gt)) * up;
  }
  STAMP(1, 2);
  xcd_barrier(cnt, (epoch + 1) * 32, b.failed);
  STAMP(1, 3);
  for (int k = threadIdx.x; k < I / 8; k += 512) act_s[k] = b.act[g * (I / 8) + k];
  __syncthreads();
  drows.finish(act_s, I / 8, acc_s, b.p_d + g * H, slot * 32);
  STAMP(1, 4);
}

// ------------------------------------------------------------------------------------------------------------------
// the same arithmetic as five dependent launches (whole-row GEMVs, no partials)
__global__ __launch_bounds__(512) void k5_qkv(Layer L, Bufs b) {
  __shared__ float xs[H];
  const float rstd = load_x(b.x, nullptr, 0, xs);
  gemv_rows(L.wqkv, blockIdx.x * 16, 16, xs, rstd, b.qkv);
}
__global__ __launch_bounds__(512) void k5_attn(Layer L, Bufs b) {
  __shared__ float lds[32 + 8 * 2 * 128];
  attn_split(L, b.qkv, blockIdx.x & 7, blockIdx.x >> 3, b.part, lds);
}
__global__ __launch_bounds__(512) void k5_o(Layer L, Bufs b) {  // 256 workgroups x 4 rows, K = 2048; merges the splits itself
  __shared__ float ctx[QD];
  for (int t = threadIdx.x; t < QD; t += 512) {
    const int hh = t >> 7, d = t & 127;
    const float* pp = b.part + (size_t)hh * NSPLIT * 130;
    float M = -INFINITY;
    for (int s = 0; s < NSPLIT; ++s) M = fmaxf(M, pp[s * 130]);
    float Ls = 0.f, o = 0.f;
    for (int s = 0; s < NSPLIT; ++s) { const float f = __expf(pp[s * 130] - M); Ls += pp[s * 130 + 1] * f; o += pp[s * 130 + 2 + d] * f; }
    ctx[t] = o / Ls;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < 4) {
    const int row = blockIdx.x * 4 + wave;
    const uint4* w = reinterpret_cast<const uint4*>(L.wo + (size_t)row * QD);
    float a = 0.f;
    for (int c = lane; c < QD / 8; c += 64) a += dot8(w[c], ctx + c * 8);
    a = wave_sum(a);
    if (lane == 0) b.x[row] += a;
  }
}
__global__ __launch_bounds__(512) void k5_gu(Layer L, Bufs b) {  // 256 workgroups x 12 act values
  __shared__ float xs[H];
  __shared__ float gu_s[24];
  const float rstd = load_x(b.x, nullptr, 0, xs);
  const int j0 = blockIdx.x * 12, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < 24; r += 8) {
    const int row = r < 12 ? j0 + r : I + j0 + (r - 12);
    const uint4* w = reinterpret_cast<const uint4*>(L.wgu + (size_t)row * H);
    float a = 0.f;
    for (int c = lane; c < H / 8; c += 64) a += dot8(w[c], xs + c * 8);
    a = wave_sum(a);
    if (lane == 0) gu_s[r] = a * rstd;
  }
  __syncthreads();
  if (threadIdx.x < 12) { const float gt = gu_s[threadIdx.x]; b.act[j0 + threadIdx.x] = gt / (1.0f + __expf(-gt)) * gu_s[12 + threadIdx.x]; }
}
__global__ __launch_bounds__(512) void k5_down(Layer L, Bufs b) {  // 256 workgroups x 4 rows, K = 3072
  __shared__ float as[I];
  for (int t = threadIdx.x; t < I; t += 512) as[t] = b.act[t];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < 4) {
    const int row = blockIdx.x * 4 + wave;
    const uint4* w = reinterpret_cast<const uint4*>(L.wd + (size_t)row * I);
    float a = 0.f;
    for (int c = lane; c < I / 8; c += 64) a += dot8(w[c], as + c * 8);
    a = wave_sum(a);
    if (lane == 0) b.x[row] += a;
  }
}

static uint16_t frand_bf16(uint32_t& st, float scale) {
  st = st * 1664525u + 1013904223u;
  const float f = (((st >> 8) & 0xffff) / 65536.0f - 0.5f) * scale;
  uint32_t u;
  memcpy(&u, &f, 4);
  return (uint16_t)(u >> 16);
}

int main() {
  hipStream_t s;
  CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const size_t per_layer = (size_t)QKV * H + (size_t)H * QD + (size_t)2 * I * H + (size_t)H * I;  // bf16 elements
  const size_t kv_layer = (size_t)NKV * MAXC * 128;
  uint16_t *W, *KV;
  CHK(hipMalloc(&W, per_layer * LAYERS * 2));
  CHK(hipMalloc(&KV, kv_layer * 2 * LAYERS * 2));
  {
    std::vector<uint16_t> h(per_layer);
    uint32_t st = 7;
    for (auto& v : h) v = frand_bf16(st, 0.06f);
    for (int l = 0; l < LAYERS; ++l) CHK(hipMemcpy(W + per_layer * l, h.data(), per_layer * 2, hipMemcpyHostToDevice));
    std::vector<uint16_t> k(kv_layer * 2);
    for (auto& v : k) v = frand_bf16(st, 1.0f);
    for (int l = 0; l < LAYERS; ++l) CHK(hipMemcpy(KV + kv_layer * 2 * l, k.data(), kv_layer * 4, hipMemcpyHostToDevice));
  }
  Bufs b{};
  float* xa; float* xb;
  CHK(hipMalloc(&xa, H * 4)); CHK(hipMalloc(&xb, H * 4)); CHK(hipMalloc(&b.x, H * 4));
  CHK(hipMalloc(&b.p_o, 8 * H * 4)); CHK(hipMalloc(&b.p_d, 8 * H * 4)); CHK(hipMalloc(&b.qkv, QKV * 4)); CHK(hipMalloc(&b.ctx, QD * 4));
  CHK(hipMalloc(&b.act, I * 4)); CHK(hipMalloc(&b.part, NQ * NSPLIT * 130 * 4)); CHK(hipMalloc(&b.sync, 8 * 64 * 4)); CHK(hipMalloc(&b.failed, 4)); CHK(hipMalloc(&b.stamps, 32 * 8)); CHK(hipMemset(b.stamps, 0, 32 * 8));
  std::vector<float> x0(H);
  for (int i = 0; i < H; ++i) x0[i] = sinf(0.37f * i);
  auto layer = [&](int l) {
    Layer L;
    const uint16_t* w = W + per_layer * l;
    L.wqkv = w; L.wo = w + (size_t)QKV * H; L.wgu = L.wo + (size_t)H * QD; L.wd = L.wgu + (size_t)2 * I * H;
    L.kc = KV + kv_layer * 2 * l; L.vc = L.kc + kv_layer;
    return L;
  };
  auto timed = [&](const char* name, auto&& enqueue, float* check) -> int {
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipMemset(b.sync, 0, 8 * 64 * 4)); CHK(hipMemset(b.failed, 0, 4));
    CHK(hipMemset(b.p_d, 0, 8 * H * 4)); CHK(hipMemset(b.p_o, 0, 8 * H * 4));
    CHK(hipMemcpy(xa, x0.data(), H * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(b.x, x0.data(), H * 4, hipMemcpyHostToDevice));
    CHK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    enqueue();
    CHK(hipStreamEndCapture(s, &g));
    CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0, s)); CHK(hipGraphLaunch(ge, s)); CHK(hipEventRecord(e1, s));
    CHK(hipStreamSynchronize(s));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    int f = 0; CHK(hipMemcpy(&f, b.failed, 4, hipMemcpyDeviceToHost));
    std::vector<float> out(H);
    CHK(hipMemcpy(out.data(), check, H * 4, hipMemcpyDeviceToHost));
    double nrm = 0; bool fin = true;
    for (float v : out) { nrm += (double)v * v; fin = fin && std::isfinite(v); }
    printf("%-60s %8.2f us per layer   (|x| = %.4g, finite %d%s)\n", name, ms * 1e3 / LAYERS, sqrt(nrm), (int)fin, f ? ", A WAIT RAN OUT" : "");
    CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
    return 0;
  };
  // one graph = one token = 28 layers (epoch = layer index: the counters are monotonic inside the graph, reset per run)
  if (timed("five launches per layer (qkv, attention, o, gate/up, down)", [&] {
        for (int l = 0; l < LAYERS; ++l) {
          const Layer L = layer(l);
          hipLaunchKernelGGL(k5_qkv, dim3(QKV / 16), dim3(512), 0, s, L, b);
          hipLaunchKernelGGL(k5_attn, dim3(NKV * NSPLIT), dim3(512), 0, s, L, b);
          hipLaunchKernelGGL(k5_o, dim3(H / 4), dim3(512), 0, s, L, b);
          hipLaunchKernelGGL(k5_gu, dim3(I / 12), dim3(512), 0, s, L, b);
          hipLaunchKernelGGL(k5_down, dim3(H / 4), dim3(512), 0, s, L, b);
        }
      }, b.x)) return 1;
  if (timed("two launches per layer, hand-offs inside each XCD", [&] {
        for (int l = 0; l < LAYERS; ++l) {
          const Layer L = layer(l);
          hipLaunchKernelGGL(k_layer_a, dim3(256), dim3(512), 0, s, L, b, (const float*)xa, l == 0 ? 0 : 8, xb, (unsigned)l);
          hipLaunchKernelGGL(k_layer_b, dim3(256), dim3(512), 0, s, L, b, (const float*)xb, xa, (unsigned)l);
        }
      }, xa)) return 1;
  {  // phase boundaries of layer 5 (10 ns ticks): launch A = start, x ready, rows done, barrier 1, attention, barrier 2, o partial
    long long st[32];
    CHK(hipMemcpy(st, b.stamps, 32 * 8, hipMemcpyDeviceToHost));
    const char* names[2] = {"launch A", "launch B"};
    for (int wg = 0; wg < 2; ++wg)
      for (int k = 0; k < 2; ++k) {
        printf("workgroup %3d %s:", wg ? 255 : 0, names[k]);
        const int n = k == 0 ? 7 : 5;
        for (int i = 1; i < n; ++i) printf(" %6.2f", (st[(wg * 2 + k) * 8 + i] - st[(wg * 2 + k) * 8 + i - 1]) * 0.01);
        printf("  us between stamps; launch A of wg 0 -> launch B of wg 0 start: %.2f us\n", (st[(0 * 2 + 1) * 8] - st[(0 * 2 + 0) * 8]) * 0.01);
      }
  }
  return 0;
}
