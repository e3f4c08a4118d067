// Skinny MFMA GEMM for the batched decode step on gfx950: out[s][n] = sum_k x[s][k] * W[n][k] with
// 4 < S <= 32 sequences.  Replaces Linear::forward (src/layers.rs:74-80) for one token of each of S
// independent utterances; every weight byte is still streamed from HBM exactly once per step, now amortised
// over up to 32 sequences.
//
// The step is latency-bound (a CU sustains only ~16 GB/s when it waits for one round trip per 8 KB), so the
// kernel is shaped to put a workgroup's whole weight slab in flight at once: one workgroup = 16 weight rows
// (32 in GLU mode: a 16-row gate tile + its 16-row up tile), 8 waves split K eight ways, and every wave issues
// all loads of four 32-wide k-steps before its first MFMA.  Per k-step
//   A = W fragment : lane (row l&15, k-chunk l>>4) loads its 16 B from the row-major bf16 matrix
//                    (4 lanes cover 64 contiguous bytes of a row)
//   B = x fragment : lane (sequence l&15 [+16], same k-chunk) loads 8 fp32 activations (L2-resident) and rounds
//                    them to bf16 (default) or splits them into bf16 hi + lo (precise mode: two MFMAs)
//   v_mfma_f32_16x16x32_bf16 -> D[row][sequence], sequences 0-15 and 16-31 share the A fragment.
// The eight K-slices are summed through LDS in a fixed order (deterministic), then bias / residual /
// SiLU(gate)*up are applied.
#include "dev.h"
#include "kernels.h"

namespace q3a {
namespace {

constexpr int SK_WAVES = 8;

__device__ __forceinline__ bf16x8_t pack8(const float4& a, const float4& b) {
  uint4 p;
  p.x = pack_bf16x2(a.x, a.y); p.y = pack_bf16x2(a.z, a.w); p.z = pack_bf16x2(b.x, b.y); p.w = pack_bf16x2(b.z, b.w);
  return *reinterpret_cast<const bf16x8_t*>(&p);
}
__device__ __forceinline__ void split8(const float4& a, const float4& b, bf16x8_t& hi, bf16x8_t& lo) {
  uint4 h, l;
  h.x = pack_bf16x2(a.x, a.y); h.y = pack_bf16x2(a.z, a.w); h.z = pack_bf16x2(b.x, b.y); h.w = pack_bf16x2(b.z, b.w);
  l.x = pack_bf16x2(a.x - bf16lo(h.x), a.y - bf16hi(h.x)); l.y = pack_bf16x2(a.z - bf16lo(h.y), a.w - bf16hi(h.y));
  l.z = pack_bf16x2(b.x - bf16lo(h.z), b.y - bf16hi(h.z)); l.w = pack_bf16x2(b.z - bf16lo(h.w), b.w - bf16hi(h.w));
  hi = *reinterpret_cast<const bf16x8_t*>(&h);
  lo = *reinterpret_cast<const bf16x8_t*>(&l);
}

// TILES = 16-row weight tiles per workgroup (1, or 2 = gate + up); SH = 16-sequence halves (1: S <= 16, 2: S <= 32)
template <bool SPLIT, int TILES, int SH>
__global__ __launch_bounds__(SK_WAVES * 64) void skinny_kernel(SkinnyArgs a) {
  __shared__ float part[SK_WAVES][TILES][SH][16][17];  // [k-slice][tile][seq half][row][sequence] (+1 pad)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kc = lane >> 4;  // row / sequence inside the fragment, k-chunk (8 elements)
  const int n0 = blockIdx.x * 16 * TILES;
  const int K = a.K;
  const int steps = K / 32, per = (steps + SK_WAVES - 1) / SK_WAVES;
  const int ks0 = wave * per, ks1 = min(steps, ks0 + per);
  // out-of-range rows / sequences are clamped to valid memory: their products are discarded by the epilogue
  const uint16_t* wrow[TILES];
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
    const int row = n0 + t * 16 + l15;
    wrow[t] = a.W + (size_t)(row < a.N ? row : a.N - 1) * K + kc * 8;
  }
  const float* xrow[SH];
#pragma unroll
  for (int h = 0; h < SH; ++h) {
    const int s = h * 16 + l15;
    xrow[h] = a.x + (size_t)(s < a.S ? s : a.S - 1) * a.ldx + kc * 8;
  }
  f32x4_t acc[TILES][SH];
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int h = 0; h < SH; ++h) acc[t][h] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  constexpr int UNR = 4;
  for (int kb = ks0; kb < ks1; kb += UNR) {
    uint4 wv[UNR][TILES];
    float4 x0[UNR][SH], x1[UNR][SH];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const bool live = kb + u < ks1;
      const int ks = live ? kb + u : ks1 - 1;  // clamp the address, zero the weight of the tail
#pragma unroll
      for (int t = 0; t < TILES; ++t) {
        wv[u][t] = *reinterpret_cast<const uint4*>(wrow[t] + ks * 32);
        if (!live) wv[u][t] = make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int h = 0; h < SH; ++h) {
        x0[u][h] = *reinterpret_cast<const float4*>(xrow[h] + ks * 32);
        x1[u][h] = *reinterpret_cast<const float4*>(xrow[h] + ks * 32 + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int h = 0; h < SH; ++h) {
        bf16x8_t hi, lo;
        if (SPLIT) split8(x0[u][h], x1[u][h], hi, lo);
        else hi = pack8(x0[u][h], x1[u][h]);
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
          const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(&wv[u][t]);
          acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, hi, acc[t][h], 0, 0, 0);
          if (SPLIT) acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, lo, acc[t][h], 0, 0, 0);
        }
      }
  }
  // D[row i][sequence j] of v_mfma_f32_16x16x32: j = lane&15, i = (lane>>4)*4 + r
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int h = 0; h < SH; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[wave][t][h][kc * 4 + r][l15] = acc[t][h][r];
  __syncthreads();
  // ---- fixed-order reduction of the K-slices + epilogue: thread -> (row i, sequence s) ----
  const int i = tid >> 5, s = tid & 31;  // 16 rows x 32 sequences = 512 threads
  if (s >= a.S || (SH == 1 && s >= 16)) return;
  const int sh = s >> 4, sj = s & 15;
  float v[TILES];
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
    v[t] = 0.f;
#pragma unroll
    for (int w = 0; w < SK_WAVES; ++w) v[t] += part[w][t][SH == 1 ? 0 : sh][i][sj];
  }
  if (TILES == 1) {
    const int n = n0 + i;
    if (n >= a.N) return;
    float y = v[0];
    if (a.bias) y += a.bias[n];
    if (a.mode == 1) y += a.resid[(size_t)s * a.ldo + n];
    a.out[(size_t)s * a.ldo + n] = y;
  } else {  // rows n0..n0+15 = gate, n0+16..n0+31 = up of logical rows n0/2 .. n0/2+15
    if (n0 + 16 + i >= a.N) return;
    float g = v[0], u = v[TILES - 1];
    if (a.bias) { g += a.bias[n0 + i]; u += a.bias[n0 + 16 + i]; }
    a.out[(size_t)s * a.ldo + (n0 >> 1) + i] = silu_f(g) * u;
  }
}

template <bool SPLIT>
void launch_s(const SkinnyArgs& a, hipStream_t s) {
  const dim3 block(SK_WAVES * 64);
  if (a.mode == 2) {
    const dim3 grid((a.N + 31) / 32);
    if (a.S <= 16) hipLaunchKernelGGL((skinny_kernel<SPLIT, 2, 1>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((skinny_kernel<SPLIT, 2, 2>), grid, block, 0, s, a);
  } else {
    const dim3 grid((a.N + 15) / 16);
    if (a.S <= 16) hipLaunchKernelGGL((skinny_kernel<SPLIT, 1, 1>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((skinny_kernel<SPLIT, 1, 2>), grid, block, 0, s, a);
  }
}

}  // namespace

const char* launch_skinny(const SkinnyArgs& a, bool split, hipStream_t s) {
  if (a.S <= 0) return nullptr;
  if (a.S > 32) return "skinny gemm: at most 32 sequences";
  if (a.K % 32 != 0 || a.ldx % 4 != 0) return "skinny gemm: K must be a multiple of 32, ldx of 4";
  if (a.mode == 2 && a.N % 32 != 0) return "skinny gemm: GLU needs N % 32 == 0";
  if (split) launch_s<true>(a, s);
  else launch_s<false>(a, s);
  return nullptr;
}

}  // namespace q3a
