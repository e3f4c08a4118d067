// LayerNorm (src/layers.rs:25-28 -> src/tensor.rs:388-402, eps 1e-5, affine) and RMSNorm
// (src/layers.rs:48-54: x * 1/sqrt(mean(x^2)+eps) * w in fp32).  One wave per row, the row lives in
// registers (float4 per lane per 256 columns), statistics via wavefront shuffles; HBM-bound.
#include "dev.h"
#include "kernels.h"

namespace q3a {
namespace {

constexpr int MAX_V = 8;  // float4 per lane -> D <= 2048

template <bool LN>
__global__ __launch_bounds__(256) void norm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ b, float* __restrict__ y, int rows, int D,
                                                   float eps, uint16_t* __restrict__ y16) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * D;
  float* yr = y + (size_t)row * D;
  const int nv = D / 4;  // float4 count
  // Every load of the row, of the weight and of the bias is requested before anything is used, from clamped (always valid)
  // addresses; lanes beyond the row are voided arithmetically.  With `if (c < nv) { load; use }` hipcc put every load into an
  // exec-masked block of its own with s_waitcnt vmcnt(0) behind it: 8 dependent round trips per row where one is enough -- 5 us per
  // launch at one clip, where a wave has its SIMD to itself and nothing hides them (profiles/r4_isa_wait_audit.txt).
  const int nit = (nv + 63) >> 6;  // wave-uniform: iterations that hold a column at all
  float4 v[MAX_V], wv[MAX_V], bv[MAX_V];
#pragma unroll
  for (int i = 0; i < MAX_V; ++i) {
    v[i] = wv[i] = bv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < nit) {
      const int c = lane + i * 64, cc = c < nv ? c : nv - 1;
      v[i] = reinterpret_cast<const float4*>(xr)[cc];
      wv[i] = reinterpret_cast<const float4*>(w)[cc];
      if (LN) bv[i] = reinterpret_cast<const float4*>(b)[cc];
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAX_V; ++i) {
    if (lane + i * 64 >= nv) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);  // a select, not a branch
    sum += LN ? (v[i].x + v[i].y + v[i].z + v[i].w) : (v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w);
  }
  sum = wave_sum(sum);
  float mean = 0.f, rstd;
  if (LN) {
    mean = sum / (float)D;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_V; ++i) {
      const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
      const float q = a * a + bb * bb + cc * cc + dd * dd;
      var += (lane + i * 64 < nv) ? q : 0.f;
    }
    var = wave_sum(var) / (float)D;
    rstd = 1.0f / sqrtf(var + eps);
  } else {
    rstd = 1.0f / sqrtf(sum / (float)D + eps);  // sqrt().reciprocal(), src/tensor.rs:323-326
  }
#pragma unroll
  for (int i = 0; i < MAX_V; ++i) {
    const int c = lane + i * 64;
    float4 o;
    if (LN) {
      o.x = (v[i].x - mean) * rstd * wv[i].x + bv[i].x;
      o.y = (v[i].y - mean) * rstd * wv[i].y + bv[i].y;
      o.z = (v[i].z - mean) * rstd * wv[i].z + bv[i].z;
      o.w = (v[i].w - mean) * rstd * wv[i].w + bv[i].w;
    } else {
      o.x = (v[i].x * rstd) * wv[i].x;
      o.y = (v[i].y * rstd) * wv[i].y;
      o.z = (v[i].z * rstd) * wv[i].z;
      o.w = (v[i].w * rstd) * wv[i].w;
    }
    if (c < nv) {
      if (y16) {  // default mode: bf16 copy for the next GEMM's LDS-DMA
        uint2 pk;
        pk.x = pack_bf16x2(o.x, o.y);
        pk.y = pack_bf16x2(o.z, o.w);
        reinterpret_cast<uint2*>(y16 + (size_t)row * D)[c] = pk;
      } else {
        reinterpret_cast<float4*>(yr)[c] = o;
      }
    }
  }
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ row_idx, int D,
                                   float* __restrict__ dst) {
  const int r = blockIdx.x;
  const float4* s = reinterpret_cast<const float4*>(src + (size_t)row_idx[r] * D);
  float4* d = reinterpret_cast<float4*>(dst + (size_t)r * D);
  for (int i = threadIdx.x; i < D / 4; i += blockDim.x) d[i] = s[i];
}

__global__ void scatter_rows_kernel(const float* __restrict__ src, const int* __restrict__ dst_row, int D,
                                    float* __restrict__ dst) {
  const int r = blockIdx.x;
  const int o = dst_row[r];
  if (o < 0) return;
  const float4* s = reinterpret_cast<const float4*>(src + (size_t)r * D);
  float4* d = reinterpret_cast<float4*>(dst + (size_t)o * D);
  for (int i = threadIdx.x; i < D / 4; i += blockDim.x) d[i] = s[i];
}

}  // namespace

const char* launch_layernorm(const float* x, const float* w, const float* b, float* y, int rows, int D, float eps,
                             hipStream_t s, uint16_t* y16) {
  if (rows <= 0) return nullptr;
  if (D % 4 != 0 || D > MAX_V * 256) return "layernorm: D must be a multiple of 4 and <= 2048";
  hipLaunchKernelGGL(norm_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, b, y, rows, D, eps, y16);
  return nullptr;
}
const char* launch_rmsnorm(const float* x, const float* w, float* y, int rows, int D, float eps, hipStream_t s, uint16_t* y16) {
  if (rows <= 0) return nullptr;
  if (D % 4 != 0 || D > MAX_V * 256) return "rmsnorm: D must be a multiple of 4 and <= 2048";
  hipLaunchKernelGGL(norm_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, (const float*)nullptr, y, rows, D, eps, y16);
  return nullptr;
}
const char* launch_gather_rows(const float* src, const int* row_idx, int n, int D, float* dst, hipStream_t s) {
  if (n <= 0) return nullptr;
  if (D % 4 != 0) return "gather_rows: D must be a multiple of 4";
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(256), 0, s, src, row_idx, D, dst);
  return nullptr;
}

const char* launch_scatter_rows(const float* src, const int* dst_row, int n, int D, float* dst, hipStream_t s) {
  if (n <= 0) return nullptr;
  if (D % 4 != 0) return "scatter_rows: D must be a multiple of 4";
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(n), dim3(256), 0, s, src, dst_row, D, dst);
  return nullptr;
}

}  // namespace q3a
