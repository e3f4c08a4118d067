// bf16-MFMA GEMM for gfx950:  Y[M][N] = act( X[M][K] . W[N][K]^T + bias ) (+ residual), fp32 accumulate.
//
//   * X: fp32 activations in HBM (dense rows, or gathered on the fly as the im2col view of an NHWC
//     feature map for the 3x3/stride-2/pad-1 convolutions) -- converted to bf16 while being staged
//     into LDS.  SPLIT=true stages a second tile with the rounding residual (x - bf16(x)) and issues
//     two MFMAs per fragment: activations then carry ~16 mantissa bits, weights are exact bf16, so the
//     product is fp32-accurate ("precise" mode of the engine).
//   * W: bf16 [N][K] row-major (K contiguous = the layout nn.Linear checkpoints already have).
//   * 256 threads = 4 waves (2x2); each wave owns a (BM/2)x(BN/2) tile built from
//     v_mfma_f32_16x16x32_bf16 fragments; register prefetch of the next K tile overlaps the MFMAs of
//     the current one.  Three tile shapes, picked by how many workgroups the problem yields:
//     128x128x64 (big M: conv stem, batched encoder), 64x64x128 and 32x32x128 (M ~ 400 prefill /
//     single-clip encoder shapes, where a deep K tile amortises the global-load latency per barrier).
//   * LDS rows are BK+8 bf16 wide so the 16 rows a lane group reads with ds_read_b128 fall on
//     distinct 16-B bank slots.
//   * blockIdx -> tile uses the bijective XCD remap: consecutive tile ids (which share the W column
//     panel) run on the same XCD/L2.
//
// Reference ops replaced: Linear::forward (src/layers.rs:74-80), Conv2d::forward (src/layers.rs:109-118),
// the gelu/+bias/+residual/SiLU*up element-wise passes around them (src/layers.rs:192-195,230-242,
// 396-400,442-463) and the permute+contiguous+reshape of src/audio_encoder.rs:132-133.
#include "dev.h"
#include "kernels.h"

namespace q3a {

namespace {


struct DenseA {
  const float* x;
  int lda;
  __device__ __forceinline__ void init_row(int m, int M, int& state0, int& state1, int& state2) const {
    state0 = (m < M) ? m : -1;
    state1 = 0;
    state2 = 0;
  }
  __device__ __forceinline__ const float* row_ptr(int s0, int, int, int k0) const {
    return s0 < 0 ? nullptr : x + (size_t)s0 * lda + k0;
  }
};

// im2col view of an NHWC fp32 feature map [img][H][W][C] for a 3x3 stride-2 pad-1 convolution.
// GEMM row m = (img, oh, ow); GEMM column k = (kh, kw, c) with c fastest.
struct ConvA {
  const float* x;
  int H, W, C, OH, OW;
  __device__ __forceinline__ void init_row(int m, int M, int& img, int& oh, int& ow) const {
    if (m < M) {
      img = m / (OH * OW);
      int r = m - img * OH * OW;
      oh = r / OW;
      ow = r - oh * OW;
    } else {
      img = -1;
      oh = ow = 0;
    }
  }
  __device__ __forceinline__ const float* row_ptr(int img, int oh, int ow, int k0) const {
    if (img < 0) return nullptr;
    int tap = k0 / C, c0 = k0 - tap * C;
    int kh = tap / 3, kw = tap - kh * 3;
    int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
    if (ih < 0 || ih >= H || iw < 0 || iw >= W) return nullptr;
    return x + (((size_t)img * H + ih) * W + iw) * C + c0;
  }
};

template <int BM, int BN, int BK, bool SPLIT, bool GLU, class ALoader>
__global__ __launch_bounds__(256) void gemm_kernel(ALoader A, const uint16_t* __restrict__ Wt, int M, int N, int K,
                                                   GemmEpilogue ep) {
  constexpr int LDS_STRIDE = BK + 8;          // bf16 elements per LDS row
  constexpr int TPR_A = BK / 4, RP_A = 256 / TPR_A;  // threads per A row (float4 each), rows per pass
  constexpr int TPR_W = BK / 8, RP_W = 256 / TPR_W;  // threads per W row (16 B each), rows per pass
  constexpr int A_LOADS = BM / RP_A;          // float4 loads per thread per K tile
  constexpr int W_LOADS = BN / RP_W;          // 16-B loads per thread per K tile
  constexpr int MI = BM / 32, NI = BN / 32;   // 16x16 fragments per wave
  static_assert(A_LOADS >= 1 && W_LOADS >= 1 && MI >= 1 && NI >= 1, "tile too small for 256 threads");
  __shared__ __attribute__((aligned(16))) uint16_t As[(SPLIT ? 2 : 1) * BM * LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) uint16_t Ws[BN * LDS_STRIDE];

  // ---- tile assignment (bijective XCD remap, cdna guide T1) ----
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = bid % tiles_m, tn = bid / tiles_m;  // tiles sharing a W panel are consecutive
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // ---- per-thread staging assignment ----
  int a_s0[A_LOADS], a_s1[A_LOADS], a_s2[A_LOADS];
  const int a_c4 = tid % TPR_A;  // which float4 of the BK-float row
  const int a_r = tid / TPR_A;
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) A.init_row(m0 + a_r + i * RP_A, M, a_s0[i], a_s1[i], a_s2[i]);
  const int w_chunk = tid % TPR_W, w_r = tid / TPR_W;
  const uint16_t* w_ptr[W_LOADS];
#pragma unroll
  for (int i = 0; i < W_LOADS; ++i) {
    int n = n0 + w_r + i * RP_W;
    w_ptr[i] = (n < N) ? Wt + (size_t)n * K + w_chunk * 8 : nullptr;
  }

  float4 a_reg[A_LOADS];
  uint4 w_reg[W_LOADS];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      const float* p = A.row_ptr(a_s0[i], a_s1[i], a_s2[i], k0);
      a_reg[i] = p ? *reinterpret_cast<const float4*>(p + a_c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < W_LOADS; ++i)
      w_reg[i] = w_ptr[i] ? *reinterpret_cast<const uint4*>(w_ptr[i] + k0) : make_uint4(0u, 0u, 0u, 0u);
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      const int row = a_r + i * RP_A;
      const float4 v = a_reg[i];
      uint2 hi;
      hi.x = pack_bf16x2(v.x, v.y);
      hi.y = pack_bf16x2(v.z, v.w);
      *reinterpret_cast<uint2*>(&As[row * LDS_STRIDE + a_c4 * 4]) = hi;
      if (SPLIT) {
        uint2 lo;
        lo.x = pack_bf16x2(v.x - bf16lo(hi.x), v.y - bf16hi(hi.x));
        lo.y = pack_bf16x2(v.z - bf16lo(hi.y), v.w - bf16hi(hi.y));
        *reinterpret_cast<uint2*>(&As[(BM + row) * LDS_STRIDE + a_c4 * 4]) = lo;
      }
    }
#pragma unroll
    for (int i = 0; i < W_LOADS; ++i) {
      const int row = w_r + i * RP_W;
      *reinterpret_cast<uint4*>(&Ws[row * LDS_STRIDE + w_chunk * 8]) = w_reg[i];
    }
  };

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int frag_row = lane & 15, frag_k = (lane >> 4) * 8;
  const int KT = K / BK;
  load_tile(0);
  for (int kt = 0; kt < KT; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < KT) load_tile((kt + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      const int kk = ks * 32 + frag_k;
      bf16x8_t bfrag[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j)
        bfrag[j] = *reinterpret_cast<const bf16x8_t*>(&Ws[(wc * (BN / 2) + j * 16 + frag_row) * LDS_STRIDE + kk]);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int arow = wr * (BM / 2) + i * 16 + frag_row;
        const bf16x8_t ah = *reinterpret_cast<const bf16x8_t*>(&As[arow * LDS_STRIDE + kk]);
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bfrag[j], acc[i][j], 0, 0, 0);
        if (SPLIT) {
          const bf16x8_t al = *reinterpret_cast<const bf16x8_t*>(&As[(BM + arow) * LDS_STRIDE + kk]);
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bfrag[j], acc[i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue.  C/D layout of v_mfma_f32_16x16x32: col = lane&15, row = (lane>>4)*4 + reg ----
  const int col_in = lane & 15, row_in = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wr * (BM / 2) + i * 16 + row_in + r;
      if (m >= M) continue;
      int orow = ep.rowmap ? ep.rowmap[m] : m;
      if (orow < 0) continue;
      if (!GLU) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int n = n0 + wc * (BN / 2) + j * 16 + col_in;
          if (n >= N) continue;
          float v = acc[i][j][r];
          if (ep.bias) v += ep.bias[n];
          if (ep.addend) v += ep.addend[(size_t)(m % ep.addend_period) * ep.ldo + n];
          if (ep.act == 1) v = gelu_erf(v);
          if (ep.resid) v += ep.resid[(size_t)orow * ep.ldo + n];
          ep.out[(size_t)orow * ep.ldo + n] = v;
        }
      } else {
        // columns come in [16 gate | 16 up] blocks: fragment j (even) = gate, j+1 = up, same lane/reg
#pragma unroll
        for (int j = 0; j < NI; j += 2) {
          const int nb = n0 + wc * (BN / 2) + j * 16;  // multiple of 32
          if (nb + 16 + col_in >= N) continue;
          float g = acc[i][j][r], u = acc[i][j + 1][r];
          if (ep.bias) {
            g += ep.bias[nb + col_in];
            u += ep.bias[nb + 16 + col_in];
          }
          ep.out[(size_t)orow * ep.ldo + (nb >> 1) + col_in] = silu_f(g) * u;
        }
      }
    }
  }
}

template <int BM, int BN, int BK, bool SPLIT, bool GLU, class ALoader>
void launch_one(const ALoader& A, const uint16_t* W, int M, int N, int K, const GemmEpilogue& ep, hipStream_t s) {
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, SPLIT, GLU, ALoader>), dim3(tiles), dim3(256), 0, s, A, W, M, N, K, ep);
}

inline long tiles_of(int M, int N, int bm, int bn) { return (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn); }

template <bool GLU, class ALoader>
const char* launch_sized(const ALoader& A, const uint16_t* W, int M, int N, int K, const GemmEpilogue& ep, bool split,
                         hipStream_t s) {
  // enough workgroups to fill 256 CUs 1.5x over decides the tile; deep-K small tiles otherwise (latency-bound shapes)
  if (tiles_of(M, N, 128, 128) >= 384 && K % 64 == 0) {
    if (split) launch_one<128, 128, 64, true, GLU>(A, W, M, N, K, ep, s);
    else launch_one<128, 128, 64, false, GLU>(A, W, M, N, K, ep, s);
  } else if (tiles_of(M, N, 64, 64) >= 384 && K % 128 == 0) {
    if (split) launch_one<64, 64, 128, true, GLU>(A, W, M, N, K, ep, s);
    else launch_one<64, 64, 128, false, GLU>(A, W, M, N, K, ep, s);
  } else if (K % 128 == 0) {
    constexpr int BN_S = GLU ? 64 : 32;  // the GLU epilogue pairs two 16-column fragments per wave
    if (split) launch_one<32, BN_S, 128, true, GLU>(A, W, M, N, K, ep, s);
    else launch_one<32, BN_S, 128, false, GLU>(A, W, M, N, K, ep, s);
  } else if (K % 32 == 0) {
    if (split) launch_one<64, 64, 32, true, GLU>(A, W, M, N, K, ep, s);
    else launch_one<64, 64, 32, false, GLU>(A, W, M, N, K, ep, s);
  } else {
    return "gemm: K must be a multiple of 32";
  }
  return nullptr;
}

}  // namespace

const char* launch_gemm(const float* X, int lda, const uint16_t* W, int M, int N, int K, const GemmEpilogue& ep,
                        bool glu, bool split, hipStream_t s) {
  if (M <= 0) return nullptr;
  if (lda % 4 != 0) return "gemm: lda must be a multiple of 4";
  if (glu && N % 32 != 0) return "gemm: GLU needs N % 32 == 0";
  DenseA A{X, lda};
  return glu ? launch_sized<true>(A, W, M, N, K, ep, split, s) : launch_sized<false>(A, W, M, N, K, ep, split, s);
}

const char* launch_conv3x3s2_gemm(const float* X, int imgs, int H, int Wd, int C, const uint16_t* Wt, int Cout,
                                  const GemmEpilogue& ep, bool split, hipStream_t s) {
  if (C % 32 != 0) return "conv gemm: C must be a multiple of 32";
  ConvA A{X, H, Wd, C, (H - 1) / 2 + 1, (Wd - 1) / 2 + 1};
  const int M = imgs * A.OH * A.OW;
  if (M <= 0) return nullptr;
  // a K tile must not straddle two filter taps: BK has to divide C
  if (C % 128 == 0) return launch_sized<false>(A, Wt, M, Cout, 9 * C, ep, split, s);
  if (C % 64 == 0 && tiles_of(M, Cout, 128, 128) >= 384) {
    if (split) launch_one<128, 128, 64, true, false>(A, Wt, M, Cout, 9 * C, ep, s);
    else launch_one<128, 128, 64, false, false>(A, Wt, M, Cout, 9 * C, ep, s);
    return nullptr;
  }
  if (tiles_of(M, Cout, 128, 128) >= 384) {  // C = 480 = 15 x 32: only BK = 32 divides it
    if (split) launch_one<128, 128, 32, true, false>(A, Wt, M, Cout, 9 * C, ep, s);
    else launch_one<128, 128, 32, false, false>(A, Wt, M, Cout, 9 * C, ep, s);
    return nullptr;
  }
  if (split) launch_one<64, 64, 32, true, false>(A, Wt, M, Cout, 9 * C, ep, s);
  else launch_one<64, 64, 32, false, false>(A, Wt, M, Cout, 9 * C, ep, s);
  return nullptr;
}

// ---- self-test support: naive fp32 reference of the same contraction (bf16 weights) ----
__global__ void gemm_ref_kernel(const float* X, const uint16_t* W, float* Y, int M, int N, int K) {
  int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N || m >= M) return;
  double acc = 0.0;
  for (int k = 0; k < K; ++k) acc += (double)X[(size_t)m * K + k] * (double)bf16_bits_to_f32(W[(size_t)n * K + k]);
  Y[(size_t)m * N + n] = (float)acc;
}
void launch_gemm_ref(const float* X, const uint16_t* W, float* Y, int M, int N, int K, hipStream_t s) {
  hipLaunchKernelGGL(gemm_ref_kernel, dim3((N + 127) / 128, M), dim3(128), 0, s, X, W, Y, M, N, K);
}

}  // namespace q3a
