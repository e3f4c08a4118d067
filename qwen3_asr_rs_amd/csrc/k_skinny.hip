// Skinny MFMA GEMM for the batched decode step on gfx950: out[s][n] = sum_k x[s][k] * W[n][k] with
// 4 < S <= 32 sequences.  Replaces Linear::forward (src/layers.rs:74-80) for one token of each of S
// independent utterances; every weight byte is still streamed from HBM exactly once per step, now amortised
// over up to 32 sequences.
//
// One workgroup = one tile of 32 weight rows; its 4 waves split K four ways.  Per 16-wide k-step a wave issues
//   A = W fragment : lane (row l&31, k-chunk l>>5) loads its 16 B straight from the row-major bf16 matrix
//   B = x fragment : lane (sequence l&31, same k-chunk) loads 8 fp32 activations (L2-resident) and rounds them
//                    to bf16 (default) or splits them into bf16 hi + lo (precise mode: two MFMAs)
//   v_mfma_f32_32x32x16_bf16 -> D[row][sequence]
// then the four K-slices are summed through LDS and the epilogue (bias / residual / SiLU(gate)*up on the
// [16 gate | 16 up] row blocks) is applied by all waves on a quarter of the tile each.
#include "dev.h"
#include "kernels.h"

namespace q3a {
namespace {

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

template <bool SPLIT>
__global__ __launch_bounds__(256) void skinny_kernel(SkinnyArgs a) {
  __shared__ float part[4][32][33];  // [k-slice][row][sequence] (+1 pad)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int n0 = blockIdx.x * 32;
  const int K = a.K;
  const int steps = K / 16, per = (steps + 3) / 4;
  const int ks0 = wave * per, ks1 = min(steps, ks0 + per);
  const int row = n0 + l31;
  // out-of-range rows / sequences are clamped to valid memory: their products are discarded by the epilogue
  const uint16_t* wrow = a.W + (size_t)(row < a.N ? row : a.N - 1) * K + half * 8;
  const float* xrow = a.x + (size_t)(l31 < a.S ? l31 : a.S - 1) * a.ldx + half * 8;

  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int ks = ks0; ks < ks1; ++ks) {
    const uint4 wv = *reinterpret_cast<const uint4*>(wrow + ks * 16);
    const float4 x0 = *reinterpret_cast<const float4*>(xrow + ks * 16);
    const float4 x1 = *reinterpret_cast<const float4*>(xrow + ks * 16 + 4);
    const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(&wv);
    if (SPLIT) {
      bf16x8_t hi, lo;
      split8(x0, x1, hi, lo);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, hi, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, lo, acc, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, pack8(x0, x1), acc, 0, 0, 0);
    }
  }
  // D[row i][sequence j]: j = lane&31, i = (r&3) + 8*(r>>2) + 4*half
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * half][l31] = acc[r];
  __syncthreads();
  // ---- reduce the four K-slices; thread -> (sequence s = tid & 31, rows i0 .. i0+3) ----
  const int s = tid & 31, i0 = (tid >> 5) * 4;
  if (s >= a.S) return;
  if (a.mode != 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = i0 + e, n = n0 + i;
      if (n >= a.N) continue;
      float v = part[0][i][s] + part[1][i][s] + part[2][i][s] + part[3][i][s];
      if (a.bias) v += a.bias[n];
      if (a.mode == 1) v += a.resid[(size_t)s * a.ldo + n];
      a.out[(size_t)s * a.ldo + n] = v;
    }
  } else if (i0 < 16) {  // rows 0..15 = gate, 16..31 = up of logical rows (n0/2) .. (n0/2)+15
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = i0 + e;
      if (n0 + 16 + i >= a.N) continue;
      float g = part[0][i][s] + part[1][i][s] + part[2][i][s] + part[3][i][s];
      float u = part[0][i + 16][s] + part[1][i + 16][s] + part[2][i + 16][s] + part[3][i + 16][s];
      if (a.bias) { g += a.bias[n0 + i]; u += a.bias[n0 + 16 + i]; }
      a.out[(size_t)s * a.ldo + (n0 >> 1) + i] = silu_f(g) * u;
    }
  }
}

}  // namespace

const char* launch_skinny(const SkinnyArgs& a, bool split, hipStream_t s) {
  if (a.S <= 0) return nullptr;
  if (a.S > 32) return "skinny gemm: at most 32 sequences";
  if (a.K % 16 != 0 || a.ldx % 4 != 0) return "skinny gemm: K must be a multiple of 16, ldx of 4";
  if (a.mode == 2 && a.N % 32 != 0) return "skinny gemm: GLU needs N % 32 == 0";
  const int blocks = (a.N + 31) / 32;
  if (split) hipLaunchKernelGGL(skinny_kernel<true>, dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(skinny_kernel<false>, dim3(blocks), dim3(256), 0, s, a);
  return nullptr;
}

}  // namespace q3a
