// Log-mel front end for gfx950 (replaces WhisperFeatureExtractor::extract, src/mel.rs:49-96).
//
// One workgroup = 32 STFT frames of one utterance.
//   1. stage the 32*160+240 reflect-padded samples in LDS (zero pad to a multiple of 160 first,
//      mel.rs:51-53, then reflection_pad1d(200,200), mel.rs:63-65); rows are skewed by one float per
//      160 samples so the 32 frame starts (stride 160) hit 32 different banks;
//   2. 400-point real DFT as an exact-f32 matrix product on v_mfma_f32_32x32x2_f32: [32 frames x 400]
//      . [400 x 402] with the periodic Hann window folded into the basis (n_fft = 400 is not a power
//      of two; ~1 GFLOP per 30 s clip on the 157 TFLOP/s f32 MFMA path).  Columns interleave Re/Im so
//      |X|^2 is one multiply and one lane shuffle away from the accumulator;
//   3. mel filterbank [128 x 201] . power[201 x 32] on the same MFMA, one 32-bin tile per wave;
//   4. log10(max(.,1e-10)) written to HBM plus a per-utterance running max (ordered-key atomicMax).
// A second tiny kernel applies max(x, gmax-8), (x+4)/4 (mel.rs:91-93).  The final STFT frame is never
// computed (mel.rs:83-84 drops it).
#include "dev.h"
#include "kernels.h"

namespace q3a {
namespace {

constexpr int N_FFT = 400, HOP = 160, N_FREQ = 201, DFT_COLS = 416, N_MELS = 128;
constexpr int FRAMES_PER_BLOCK = 32;
constexpr int SAMPLES_PER_BLOCK = (FRAMES_PER_BLOCK - 1) * HOP + N_FFT;  // 5360
constexpr int P_STRIDE = 209;                                             // odd -> conflict-free column reads
constexpr int K_MEL = 202;                                                // 201 padded to even

__device__ __forceinline__ unsigned float_key(float x) {
  unsigned u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(256) void mel_kernel(MelBatch mb, const float* __restrict__ dft,
                                                  const float* __restrict__ filt_t) {
  __shared__ float s_lds[SAMPLES_PER_BLOCK + SAMPLES_PER_BLOCK / HOP + 8];
  __shared__ float p_lds[FRAMES_PER_BLOCK * P_STRIDE];
  const int b = blockIdx.y;
  const int F = mb.n_frames[b];
  const int f0 = blockIdx.x * FRAMES_PER_BLOCK;
  if (f0 >= F) return;
  const int64_t n = mb.n_samples[b];
  const int64_t Lp = (int64_t)F * HOP;  // zero-padded length (multiple of hop)
  const float* pcm = mb.pcm + mb.pcm_off[b];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;

  for (int p = tid; p < SAMPLES_PER_BLOCK; p += 256) {
    int64_t i = (int64_t)f0 * HOP + p - N_FFT / 2;
    if (i < 0) i = -i;
    else if (i >= Lp) i = 2 * (Lp - 1) - i;
    float v = 0.f;
    if (i >= 0 && i < n) v = pcm[i];
    s_lds[p + p / HOP] = v;
  }
  for (int i = tid; i < FRAMES_PER_BLOCK * P_STRIDE; i += 256) p_lds[i] = 0.f;
  __syncthreads();

  // ---- DFT: D[frame][c] = sum_n s[frame*160+n] * dft[n][c],  13 column tiles of 32 over 4 waves ----
  const int a_base = l31 * (HOP + 1);
  for (int tile = wave; tile < DFT_COLS / 32; tile += 4) {
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* bcol = dft + tile * 32 + l31;
#pragma unroll 4
    for (int nn = 0; nn < N_FFT; nn += 2) {
      const int m = nn + half;
      const float a = s_lds[a_base + m + (m >= HOP) + (m >= 2 * HOP)];
      const float bv = bcol[(size_t)m * DFT_COLS];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
    }
    // C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int c = tile * 32 + l31, k = c >> 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float pw = acc[r] * acc[r];
      pw += __shfl_xor(pw, 1, 64);  // Re^2 + Im^2 (adjacent columns)
      const int frame = (r & 3) + 8 * (r >> 2) + 4 * half;
      if ((c & 1) == 0 && k < P_STRIDE - 1) p_lds[frame * P_STRIDE + k] = pw;
    }
  }
  __syncthreads();

  // ---- mel: D[m][frame] = sum_k filt_t[k][m] * P[frame][k]; wave w owns mel bins 32w..32w+31 ----
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* acol = filt_t + wave * 32 + l31;
  const float* brow = p_lds + l31 * P_STRIDE;
#pragma unroll 4
  for (int kk = 0; kk < K_MEL; kk += 2) {
    const int k = kk + half;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(acol[(size_t)k * N_MELS], brow[k], acc, 0, 0, 0);
  }
  const int f = f0 + l31;
  float vmax = -3.0e38f;
  float* out = mb.mel + mb.mel_off[b];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    const float lv = log10f(fmaxf(acc[r], 1e-10f));  // mel.rs:90
    if (f < F) {
      out[(size_t)m * F + f] = lv;
      vmax = fmaxf(vmax, lv);
    }
  }
  vmax = wave_max(vmax);
  if (lane == 0) atomicMax(mb.gmax_key + b, float_key(vmax));
}

__global__ void mel_finalize_kernel(MelBatch mb) {
  const int b = blockIdx.y;
  const int64_t total = (int64_t)mb.n_frames[b] * N_MELS;
  const float floor_v = key_float(mb.gmax_key[b]) - 8.0f;  // mel.rs:91-92
  float* m = mb.mel + mb.mel_off[b];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    m[i] = (fmaxf(m[i], floor_v) + 4.0f) / 4.0f;  // mel.rs:93
}

}  // namespace

const char* launch_mel(const MelBatch& mb, int B, int max_frames, const float* dft, const float* filt_t,
                       hipStream_t s) {
  if (B <= 0) return nullptr;
  hipError_t e = hipMemsetAsync(mb.gmax_key, 0, sizeof(unsigned) * B, s);
  if (e != hipSuccess) return "mel: memset failed";
  const int tiles = (max_frames + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK;
  hipLaunchKernelGGL(mel_kernel, dim3(tiles, B), dim3(256), 0, s, mb, dft, filt_t);
  int fb = (max_frames * N_MELS + 255) / 256;
  if (fb > 1024) fb = 1024;
  hipLaunchKernelGGL(mel_finalize_kernel, dim3(fb, B), dim3(256), 0, s, mb);
  return nullptr;
}

}  // namespace q3a
