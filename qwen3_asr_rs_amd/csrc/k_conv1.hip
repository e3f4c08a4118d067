// First conv of the audio stem (C_in = 1): 3x3, stride 2, pad 1, + bias + erf-GELU
// (src/audio_encoder.rs:127, Conv2d::forward src/layers.rs:109-118).  Memory/VALU bound (9 MACs per
// output) -- no MFMA.  Fuses the chunking of src/audio_encoder.rs:96-124: the kernel reads the
// utterance's (128, F) mel block directly, chunk c covers frames [c*100, c*100+100) and everything
// past the utterance's last frame reads as zero (the reference zero-pads the tail chunk).
// Output is NHWC fp32 [chunk][H/2][W/2][Cout] so that conv2's implicit-GEMM rows are contiguous.
#include "dev.h"
#include "kernels.h"

namespace q3a {
namespace {

constexpr int MAX_W = 128;     // chunk_frames <= 128
constexpr int MAX_COUT = 512;  // weights + bias staged in LDS

__global__ __launch_bounds__(256) void conv1_kernel(const float* __restrict__ mel, const int64_t* __restrict__ mel_off,
                                                    const int* __restrict__ n_frames, ChunkTable ct, int n_mels,
                                                    int chunk_frames, const float* __restrict__ w,
                                                    const float* __restrict__ bias, int Cout, float* __restrict__ out,
                                                    uint16_t* __restrict__ out16) {
  __shared__ float in_l[3][MAX_W + 2];     // three input rows, column index shifted by +1 (pad)
  __shared__ float w_l[MAX_COUT * 9];
  __shared__ float b_l[MAX_COUT];
  const int chunk = blockIdx.y, oh = blockIdx.x;
  const int OH = (n_mels - 1) / 2 + 1, OW = (chunk_frames - 1) / 2 + 1;
  const int utt = ct.chunk_utt[chunk], fr0 = ct.chunk_frame0[chunk];
  const int F = n_frames[utt];
  const float* m = mel + mel_off[utt];
  const int tid = threadIdx.x;
  for (int i = tid; i < 3 * (MAX_W + 2); i += 256) {
    const int kh = i / (MAX_W + 2), col = i % (MAX_W + 2);
    const int ih = oh * 2 - 1 + kh, iw = col - 1;
    float v = 0.f;
    if (ih >= 0 && ih < n_mels && iw >= 0 && iw < chunk_frames && fr0 + iw < F) v = m[(size_t)ih * F + fr0 + iw];
    in_l[kh][col] = v;
  }
  for (int i = tid; i < Cout * 9; i += 256) w_l[i] = w[i];
  for (int i = tid; i < Cout; i += 256) b_l[i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const size_t obase = ((size_t)chunk * OH + oh) * OW * Cout;
  const int total = OW * Cout;
  for (int e = tid; e < total; e += 256) {
    const int ow = e / Cout, co = e - ow * Cout;
    const float* wp = w_l + co * 9;
    float acc = b_l[co];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) acc += wp[kh * 3 + kw] * in_l[kh][ow * 2 + kw];  // (ow*2-1+kw)+1
    if (out16) out16[obase + e] = (uint16_t)f32_to_bf16_bits(gelu_fast(acc));  // default mode: bf16 map feeds conv2's LDS-DMA
    else out[obase + e] = gelu_erf(acc);
  }
}

// Default-mode variant (bf16 NHWC map for conv2's LDS-DMA loader).  The kernel is VALU-bound on the activation
// (9 MACs + ~20 instructions of GELU per output against 2 B written), so the mapping is chosen for issue slots:
// a thread owns one PAIR of output channels for the whole row -- its 18 weights and 2 biases stay in registers,
// every multiply-add and the GELU polynomial run as packed fp32 (v_pk_fma_f32: both channels per issue), the nine
// inputs of an output column are wave-uniform LDS broadcasts, and a lane stores both channels as one dword
// (a wave writes 240..256 contiguous bytes of an NHWC row).  ROWS output rows per block share the staged input rows.
constexpr int C1_ROWS = 2;
__global__ __launch_bounds__(256) void conv1_pair_kernel(const float* __restrict__ mel, const int64_t* __restrict__ mel_off,
                                                         const int* __restrict__ n_frames, ChunkTable ct, int n_mels,
                                                         int chunk_frames, const float* __restrict__ w,
                                                         const float* __restrict__ bias, int Cout,
                                                         uint32_t* __restrict__ out16x2) {
  constexpr int IN_ROWS = 2 * C1_ROWS + 1;
  __shared__ float in_l[IN_ROWS][MAX_W + 2];  // input rows oh0*2-1 .., column index shifted by +1 (pad)
  const int chunk = blockIdx.y, oh0 = blockIdx.x * C1_ROWS;
  const int OH = (n_mels - 1) / 2 + 1, OW = (chunk_frames - 1) / 2 + 1;
  const int utt = ct.chunk_utt[chunk], fr0 = ct.chunk_frame0[chunk];
  const int F = n_frames[utt];
  const float* m = mel + mel_off[utt];
  const int tid = threadIdx.x;
  for (int i = tid; i < IN_ROWS * (MAX_W + 2); i += 256) {
    const int kh = i / (MAX_W + 2), col = i % (MAX_W + 2);
    const int ih = oh0 * 2 - 1 + kh, iw = col - 1;
    float v = 0.f;
    if (ih >= 0 && ih < n_mels && iw >= 0 && iw < chunk_frames && fr0 + iw < F) v = m[(size_t)ih * F + fr0 + iw];
    in_l[kh][col] = v;
  }
  __syncthreads();
  const int half = Cout >> 1;
  for (int pair = tid; pair < half; pair += 256) {  // (one trip for Cout <= 512)
    f32x2_t wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = f32x2_t{w[(2 * pair) * 9 + k], w[(2 * pair + 1) * 9 + k]};
    const f32x2_t bv = bias ? f32x2_t{bias[2 * pair], bias[2 * pair + 1]} : f32x2_t{0.f, 0.f};
#pragma unroll
    for (int r = 0; r < C1_ROWS; ++r) {
      const int oh = oh0 + r;
      if (oh >= OH) break;
      uint32_t* orow = out16x2 + (((size_t)chunk * OH + oh) * OW * Cout >> 1) + pair;
      for (int ow = 0; ow < OW; ++ow) {
        f32x2_t acc = bv;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const float x = own_vgpr(in_l[2 * r + kh][ow * 2 + kw]);  // (not the high half of a ds_read2 pair: dev.h, op_sel hazard)
            acc += wv[kh * 3 + kw] * f32x2_t{x, x};
          }
        const f32x2_t g = gelu_fast2(acc);
        orow[(size_t)ow * half] = pack_bf16x2(g.x, g.y);
      }
    }
  }
}

}  // namespace

const char* launch_conv1(const float* mel, const int64_t* mel_off, const int* n_frames, const ChunkTable& ct,
                         int n_chunks, int n_mels, int chunk_frames, const float* w, const float* b, int Cout,
                         float* out, hipStream_t s, uint16_t* out16) {
  if (n_chunks <= 0) return nullptr;
  if (chunk_frames > MAX_W) return "conv1: chunk_frames > 128 unsupported";
  if (Cout > MAX_COUT) return "conv1: more than 512 channels unsupported";
  const int OH = (n_mels - 1) / 2 + 1;
  if (out16 && Cout % 2 == 0 && Cout >= 2) {
    hipLaunchKernelGGL(conv1_pair_kernel, dim3((OH + C1_ROWS - 1) / C1_ROWS, n_chunks), dim3(256), 0, s, mel, mel_off,
                       n_frames, ct, n_mels, chunk_frames, w, b, Cout, reinterpret_cast<uint32_t*>(out16));
    return nullptr;
  }
  hipLaunchKernelGGL(conv1_kernel, dim3(OH, n_chunks), dim3(256), 0, s, mel, mel_off, n_frames, ct, n_mels,
                     chunk_frames, w, b, Cout, out, out16);
  return nullptr;
}

}  // namespace q3a
