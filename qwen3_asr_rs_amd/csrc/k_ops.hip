// Kernels of the op-level veneer (include/q3asr_ops.h): an eager fp32 tensor library behind the reference's
// `struct Tensor` seam (src/tensor.rs:145-488).  One generic kernel per op class; correctness first -- the fast path is
// the engine (k_gemm256 / k_fattn / k_gemv ...).  LayerNorm reuses the engine's kernel (k_norm.hip).
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev.h"
#include "ops.h"

namespace q3a {
namespace ops {
namespace {

__device__ __forceinline__ double load_as_double(const void* p, int dt, long i) {
  switch (dt) {
    case DT_F32: return (double)reinterpret_cast<const float*>(p)[i];
    case DT_BF16: return (double)bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(p)[i]);
    case DT_F16: return (double)__half2float(reinterpret_cast<const __half*>(p)[i]);
    case DT_I64: return (double)reinterpret_cast<const long long*>(p)[i];
    case DT_I32: return (double)reinterpret_cast<const int*>(p)[i];
    default: return reinterpret_cast<const uint8_t*>(p)[i] ? 1.0 : 0.0;
  }
}
__device__ __forceinline__ void store_from_double(void* p, int dt, long i, double v) {
  switch (dt) {
    case DT_F32: reinterpret_cast<float*>(p)[i] = (float)v; break;
    case DT_BF16: reinterpret_cast<uint16_t*>(p)[i] = (uint16_t)f32_to_bf16_bits((float)v); break;
    case DT_F16: reinterpret_cast<__half*>(p)[i] = __float2half((float)v); break;
    case DT_I64: reinterpret_cast<long long*>(p)[i] = (long long)v; break;
    case DT_I32: reinterpret_cast<int*>(p)[i] = (int)v; break;
    default: reinterpret_cast<uint8_t*>(p)[i] = v != 0.0 ? 1 : 0; break;
  }
}

__device__ __forceinline__ long view_index(const View& v, long lin) {  // row-major linear index over v.shape -> element offset
  long off = v.offset;
#pragma unroll 1
  for (int d = v.nd - 1; d >= 0; --d) {
    const long q = lin / v.shape[d];
    off += (lin - q * v.shape[d]) * v.stride[d];
    lin = q;
  }
  return off;
}

// dst[dv(i)] = convert(src[sv(i)]) over the common shape; same-dtype copies move the bits (exact for I64)
__global__ void copy_view_kernel(void* dst, int ddt, View dv, const void* src, int sdt, View sv, long n, int esize) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long di = view_index(dv, i), si = view_index(sv, i);
    if (ddt == sdt) {
      switch (esize) {
        case 8: reinterpret_cast<uint64_t*>(dst)[di] = reinterpret_cast<const uint64_t*>(src)[si]; break;
        case 4: reinterpret_cast<uint32_t*>(dst)[di] = reinterpret_cast<const uint32_t*>(src)[si]; break;
        case 2: reinterpret_cast<uint16_t*>(dst)[di] = reinterpret_cast<const uint16_t*>(src)[si]; break;
        default: reinterpret_cast<uint8_t*>(dst)[di] = reinterpret_cast<const uint8_t*>(src)[si]; break;
      }
    } else {
      store_from_double(dst, ddt, di, load_as_double(src, sdt, si));
    }
  }
}

__global__ void fill_kernel(void* dst, int dt, View dv, long n, double val) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    store_from_double(dst, dt, view_index(dv, i), val);
}
__global__ void arange_kernel(void* dst, int dt, long n, double start, double step) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    store_from_double(dst, dt, i, start + step * (double)i);
}

__global__ void binary_kernel(float* out, const float* a, View av, const float* b, View bv, long n, int op) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = a[view_index(av, i)], y = b[view_index(bv, i)];
    float r;
    switch (op) {
      case B_ADD: r = x + y; break;
      case B_SUB: r = x - y; break;
      case B_MUL: r = x * y; break;
      case B_DIV: r = x / y; break;
      default: r = (x != x || y != y) ? (x + y) : fmaxf(x, y); break;  // torch.maximum propagates NaN
    }
    out[i] = r;
  }
}

__global__ void unary_kernel(float* out, const float* in, long n, int op, float p) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = in[i];
    float r;
    switch (op) {
      case U_NEG: r = -x; break;
      case U_ABS: r = fabsf(x); break;
      case U_SQUARE: r = x * x; break;
      case U_SQRT: r = sqrtf(x); break;
      case U_RSQRT: r = 1.0f / sqrtf(x); break;  // tensor.rs:323-326: sqrt, then reciprocal
      case U_LOG10: r = log10f(x); break;
      case U_SIN: r = sinf(x); break;
      case U_COS: r = cosf(x); break;
      case U_EXP: r = expf(x); break;
      case U_GELU: r = gelu_erf(x); break;
      case U_SILU: r = silu_f(x); break;
      case U_CLAMP_MIN: r = (x != x) ? x : fmaxf(x, p); break;
      case U_ADD_S: r = x + p; break;
      case U_SUB_S: r = x - p; break;
      case U_MUL_S: r = x * p; break;
      case U_DIV_S: r = x / p; break;
      default: r = powf(x, p); break;
    }
    out[i] = r;
  }
}

__global__ void complex_abs_kernel(float* out, const float2* in, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = hypotf(in[i].x, in[i].y);
}

// C[b][M][N] = A[b][M][K] . B[b][K][N]  (row-major, contiguous; batch strides may be 0 = broadcast) on the f32-input matrix
// core: v_mfma_f32_32x32x2_f32 is exact f32 -- D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)), one rounding per product, k ascending --
// so every output is the same plain k-ordered fmaf chain a CPU reference (and the VALU kernel this replaces) computes, at
// the f32 vector rate without occupying the VALU.  64 x 64 tile per workgroup, 4 waves of 32 x 32, K in steps of 32 through
// LDS.  The tile loader is a functor so that the convolution below runs on the same body as an implicit GEMM.
constexpr int MT = 64, MK = 32;
// tb != 0: B is given transposed, row-major [N][K] (Linear::forward's `weight.tr()`, src/layers.rs:74-80: the weight is read in
// place instead of being copied into a [K][N] matrix first)
struct DenseTiles {
  const float* a; const float* b; int M, N, K, tb;
  __device__ __forceinline__ float A(int m, int k) const { return (m < M && k < K) ? a[(long)m * K + k] : 0.f; }
  __device__ __forceinline__ float B(int k, int n) const { return (k < K && n < N) ? (tb ? b[(long)n * K + k] : b[(long)k * N + n]) : 0.f; }
  __device__ __forceinline__ float init(int) const { return 0.f; }
  __device__ __forceinline__ void store(float* c, int m, int n, float v) const { if (m < M && n < N) c[(long)m * N + n] = v; }
};
template <class T>
__device__ __forceinline__ void mfma_tile_f32(const T& t, float* c, int K, bool b_k_fast) {
  __shared__ float as[MT][MK + 1], bs[MK][MT + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * MT, n0 = blockIdx.y * MT;  // rows on x: the long dimension (implicit-GEMM convolutions)
  f32x16_t acc;
  {
    const float v0 = t.init(n0 + wn * 32 + (lane & 31));  // (bias of the output column: the accumulator starts there)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = v0;
  }
  for (int k0 = 0; k0 < K; k0 += MK) {
#pragma unroll
    for (int i = 0; i < MT * MK / 256; ++i) {
      const int idx = tid + i * 256;
      as[idx / MK][idx % MK] = t.A(m0 + idx / MK, k0 + idx % MK);  // consecutive threads along k
      if (b_k_fast) bs[idx % MK][idx / MK] = t.B(k0 + idx % MK, n0 + idx / MK);   // B stored [N][K]: consecutive threads along k
      else bs[idx / MT][idx % MT] = t.B(k0 + idx / MT, n0 + idx % MT);            // B stored [K][N]: consecutive threads along n
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < MK; kk += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[wm * 32 + (lane & 31)][kk + (lane >> 5)], bs[kk + (lane >> 5)][wn * 32 + (lane & 31)], acc, 0, 0, 0);
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 16; ++r)  // C/D layout of the 32 x 32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    t.store(c, m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), n0 + wn * 32 + (lane & 31), acc[r]);
}
__global__ __launch_bounds__(256) void matmul_kernel(float* C, const float* A, const float* B, int M, int N, int K, long sa, long sb, int tb) {
  const int bz = blockIdx.z;
  const DenseTiles t{A + (long)bz * sa, B + (long)bz * sb, M, N, K, tb};
  mfma_tile_f32(t, C + (long)bz * M * N, K, tb != 0);
}

// one workgroup per row of D contiguous floats
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* out, const float* in, int D) {
  __shared__ float red[4];
  const float* x = in + (long)blockIdx.x * D;
  float* y = out + (long)blockIdx.x * D;
  const int tid = threadIdx.x;
  float m = -INFINITY;
  for (int i = tid; i < D; i += 256) m = fmaxf(m, x[i]);
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int i = tid; i < D; i += 256) s += expf(x[i] - m);
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  s = (red[0] + red[1]) + (red[2] + red[3]);
  for (int i = tid; i < D; i += 256) y[i] = expf(x[i] - m) / s;
}

// generic LayerNorm over the last dimension (any D; the engine's kernel of k_norm.hip serves D % 4 == 0, D <= 2048):
// two-pass mean / variance in fp32 like ATen's CPU kernel, one workgroup per row
__global__ __launch_bounds__(256) void layernorm_rows_kernel(float* out, const float* in, const float* w, const float* b, int D, float eps) {
  __shared__ float red[4];
  const float* x = in + (long)blockIdx.x * D;
  float* y = out + (long)blockIdx.x * D;
  const int tid = threadIdx.x;
  float s = 0.f;
  for (int i = tid; i < D; i += 256) s += x[i];
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)D;
  __syncthreads();
  float v = 0.f;
  for (int i = tid; i < D; i += 256) { const float dlt = x[i] - mean; v += dlt * dlt; }
  v = wave_sum(v);
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  const float rstd = 1.0f / sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)D + eps);
  for (int i = tid; i < D; i += 256) y[i] = (x[i] - mean) * rstd * (w ? w[i] : 1.f) + (b ? b[i] : 0.f);
}

__global__ __launch_bounds__(256) void mean_rows_kernel(float* out, const float* in, int D) {
  __shared__ float red[4];
  const float* x = in + (long)blockIdx.x * D;
  float s = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) s += x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)D;
}

// row max / argmax (first index on ties, NaN treated as larger than everything like torch): one workgroup per row
__global__ __launch_bounds__(256) void argmax_rows_kernel(long long* idx_out, float* val_out, const float* in, int D) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  const float* x = in + (long)blockIdx.x * D;
  float best = -INFINITY;
  int bidx = 0x7fffffff;
  for (int i = threadIdx.x; i < D; i += 256) {
    const float v = x[i];
    if (bidx == 0x7fffffff || v > best) { best = v; bidx = i; }  // i increases: strict > keeps the first index
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bidx, o, 64);
    if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
  }
  if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = bidx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < bidx)) { best = bv[w]; bidx = bi[w]; }
    if (idx_out) idx_out[blockIdx.x] = bidx == 0x7fffffff ? 0 : bidx;
    if (val_out) val_out[blockIdx.x] = best;
  }
}

__global__ void triu_kernel(float* out, const float* in, long n, int R, int Cc, long diag) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long c = i % Cc, r = (i / Cc) % R;
    out[i] = (c - r >= diag) ? in[i] : 0.f;
  }
}

__global__ void embedding_kernel(float* out, const float* w, const long long* idx, long n_idx, int D) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_idx * D; i += (long)gridDim.x * blockDim.x) {
    const long r = i / D;
    out[i] = w[(long)idx[r] * D + (i - r * D)];
  }
}

__global__ void reflect_pad_kernel(float* out, const float* in, long rows, long n, long pl, long pr) {
  const long no = n + pl + pr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * no; i += (long)gridDim.x * blockDim.x) {
    const long r = i / no, j = i - r * no;
    long src = j - pl;
    if (src < 0) src = -src;
    if (src >= n) src = 2 * (n - 1) - src;
    out[i] = in[r * n + src];
  }
}

// NCHW convolution, groups == 1, as an implicit GEMM on the tile body above: row m = (image, oh, ow), column = output
// channel, k = (ci, kh, kw) ascending -- the summation order of a direct loop nest; padded taps contribute exact zeros and
// the accumulator starts at the bias.
struct ConvTiles {
  const float* in; const float* w; const float* bias; ConvDims d; int M, K;
  __device__ __forceinline__ float A(int m, int k) const {
    if (m >= M || k >= K) return 0.f;
    const int ow = m % d.OW, oh = (m / d.OW) % d.OH, n = m / (d.OW * d.OH);
    const int kw = k % d.KW, kh = (k / d.KW) % d.KH, ci = k / (d.KW * d.KH);
    const int ih = oh * d.sh - d.ph + kh * d.dh, iw = ow * d.sw - d.pw + kw * d.dw;
    if (ih < 0 || ih >= d.H || iw < 0 || iw >= d.W) return 0.f;
    return in[(((long)n * d.Ci + ci) * d.H + ih) * d.W + iw];
  }
  __device__ __forceinline__ float B(int k, int co) const { return (k < K && co < d.Co) ? w[(long)co * K + k] : 0.f; }
  __device__ __forceinline__ float init(int co) const { return (bias && co < d.Co) ? bias[co] : 0.f; }
  __device__ __forceinline__ void store(float* out, int m, int co, float v) const {
    if (m >= M || co >= d.Co) return;
    const int ow = m % d.OW, oh = (m / d.OW) % d.OH, n = m / (d.OW * d.OH);
    out[(((long)n * d.Co + co) * d.OH + oh) * d.OW + ow] = v;
  }
};
__global__ __launch_bounds__(256) void conv2d_kernel(float* out, const float* in, const float* w, const float* bias, ConvDims d) {
  const ConvTiles t{in, w, bias, d, d.N * d.OH * d.OW, d.Ci * d.KH * d.KW};
  mfma_tile_f32(t, out, t.K, true);
}

// STFT without centering: frame f = x[f*hop .. f*hop + n_fft) * window; out[k][f] = sum_t frame[t] * exp(-2 pi i k t / n_fft)
// (the DFT matrix cos/sin [n_freq][n_fft] is made on the host in f64).  Output (n_freq, n_frames) complex64.
__global__ __launch_bounds__(256) void stft_kernel(float2* out, const float* x, const float* win, const float* ct, const float* st,
                                                   int n_fft, int hop, int n_frames, int n_freq, float scale) {
  extern __shared__ float frame[];
  const int f = blockIdx.x;
  for (int t = threadIdx.x; t < n_fft; t += 256) frame[t] = x[(long)f * hop + t] * win[t];
  __syncthreads();
  for (int k = threadIdx.x; k < n_freq; k += 256) {
    float re = 0.f, im = 0.f;
    const float* c = ct + (long)k * n_fft;
    const float* s = st + (long)k * n_fft;
    for (int t = 0; t < n_fft; ++t) {
      re = fmaf(frame[t], c[t], re);
      im = fmaf(frame[t], s[t], im);
    }
    out[(long)k * n_frames + f] = make_float2(re * scale, -im * scale);
  }
}

inline int grid_for(long n) {
  long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

void k_copy_view(void* dst, int ddt, const View& dv, const void* src, int sdt, const View& sv, long n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(copy_view_kernel, dim3(grid_for(n)), dim3(256), 0, s, dst, ddt, dv, src, sdt, sv, n, dtype_size(ddt));
}
void k_fill(void* dst, int dt, const View& dv, long n, double val, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, s, dst, dt, dv, n, val);
}
void k_arange(void* dst, int dt, long n, double start, double step, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(arange_kernel, dim3(grid_for(n)), dim3(256), 0, s, dst, dt, n, start, step);
}
void k_binary(float* out, const float* a, const View& av, const float* b, const View& bv, long n, int op, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(binary_kernel, dim3(grid_for(n)), dim3(256), 0, s, out, a, av, b, bv, n, op);
}
void k_unary(float* out, const float* in, long n, int op, float p, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(unary_kernel, dim3(grid_for(n)), dim3(256), 0, s, out, in, n, op, p);
}
void k_complex_abs(float* out, const void* in, long n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(complex_abs_kernel, dim3(grid_for(n)), dim3(256), 0, s, out, reinterpret_cast<const float2*>(in), n);
}
void k_matmul(float* C, const float* A, const float* B, int batch, int M, int N, int K, long sa, long sb, bool b_transposed, hipStream_t s) {
  if (batch <= 0 || M <= 0 || N <= 0) return;
  hipLaunchKernelGGL(matmul_kernel, dim3((M + MT - 1) / MT, (N + MT - 1) / MT, batch), dim3(256), 0, s, C, A, B, M, N, K, sa, sb, b_transposed ? 1 : 0);
}
void k_softmax_rows(float* out, const float* in, long rows, int D, hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, out, in, D);
}
void k_layernorm_rows(float* out, const float* in, const float* w, const float* b, long rows, int D, float eps, hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(layernorm_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, out, in, w, b, D, eps);
}
void k_mean_rows(float* out, const float* in, long rows, int D, hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(mean_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, out, in, D);
}
void k_argmax_rows(long long* idx_out, float* val_out, const float* in, long rows, int D, hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, idx_out, val_out, in, D);
}
void k_triu(float* out, const float* in, long n, int R, int C, long diag, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(triu_kernel, dim3(grid_for(n)), dim3(256), 0, s, out, in, n, R, C, diag);
}
void k_embedding(float* out, const float* w, const long long* idx, long n_idx, int D, hipStream_t s) {
  if (n_idx <= 0) return;
  hipLaunchKernelGGL(embedding_kernel, dim3(grid_for(n_idx * D)), dim3(256), 0, s, out, w, idx, n_idx, D);
}
void k_reflect_pad(float* out, const float* in, long rows, long n, long pl, long pr, hipStream_t s) {
  hipLaunchKernelGGL(reflect_pad_kernel, dim3(grid_for(rows * (n + pl + pr))), dim3(256), 0, s, out, in, rows, n, pl, pr);
}
void k_conv2d(float* out, const float* in, const float* w, const float* bias, const ConvDims& d, hipStream_t s) {
  const long rows = (long)d.N * d.OH * d.OW;
  if (rows <= 0 || d.Co <= 0) return;
  hipLaunchKernelGGL(conv2d_kernel, dim3((unsigned)((rows + MT - 1) / MT), (d.Co + MT - 1) / MT), dim3(256), 0, s, out, in, w, bias, d);
}
void k_stft(void* out, const float* x, const float* win, const float* ct, const float* st, int n_fft, int hop, int n_frames,
            int n_freq, float scale, hipStream_t s) {
  if (n_frames <= 0) return;
  hipLaunchKernelGGL(stft_kernel, dim3(n_frames), dim3(256), n_fft * sizeof(float), s, reinterpret_cast<float2*>(out), x, win, ct, st,
                     n_fft, hop, n_frames, n_freq, scale);
}

}  // namespace ops
}  // namespace q3a
