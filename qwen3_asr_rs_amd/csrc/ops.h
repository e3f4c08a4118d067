// Shared declarations of the op-level veneer (ops.cpp: arrays / views / dispatch; k_ops.hip: kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace q3a {
namespace ops {

enum { DT_F32 = 0, DT_F16 = 1, DT_BF16 = 2, DT_I64 = 3, DT_I32 = 4, DT_BOOL = 5, DT_C64 = 6 };
inline int dtype_size(int dt) { return dt == DT_F32 || dt == DT_I32 ? 4 : (dt == DT_I64 || dt == DT_C64 ? 8 : (dt == DT_BOOL ? 1 : 2)); }

constexpr int MAXD = 8;
struct View {  // element strides; offset in elements from the storage base
  int nd;
  long shape[MAXD];
  long stride[MAXD];
  long offset;
};

enum { B_ADD = 0, B_SUB, B_MUL, B_DIV, B_MAX };
enum { U_NEG = 0, U_ABS, U_SQUARE, U_SQRT, U_RSQRT, U_LOG10, U_SIN, U_COS, U_EXP, U_GELU, U_SILU, U_CLAMP_MIN, U_ADD_S, U_SUB_S, U_MUL_S,
       U_DIV_S, U_POW_S };

struct ConvDims { int N, Ci, H, W, Co, KH, KW, OH, OW, sh, sw, ph, pw, dh, dw; };

void k_copy_view(void* dst, int ddt, const View& dv, const void* src, int sdt, const View& sv, long n, hipStream_t s);
void k_fill(void* dst, int dt, const View& dv, long n, double val, hipStream_t s);
void k_arange(void* dst, int dt, long n, double start, double step, hipStream_t s);
void k_binary(float* out, const float* a, const View& av, const float* b, const View& bv, long n, int op, hipStream_t s);
void k_unary(float* out, const float* in, long n, int op, float p, hipStream_t s);
void k_complex_abs(float* out, const void* in, long n, hipStream_t s);
void k_matmul(float* C, const float* A, const float* B, int batch, int M, int N, int K, long sa, long sb, bool b_transposed, hipStream_t s);
void k_softmax_rows(float* out, const float* in, long rows, int D, hipStream_t s);
void k_layernorm_rows(float* out, const float* in, const float* w, const float* b, long rows, int D, float eps, hipStream_t s);
void k_mean_rows(float* out, const float* in, long rows, int D, hipStream_t s);
void k_argmax_rows(long long* idx_out, float* val_out, const float* in, long rows, int D, hipStream_t s);
void k_triu(float* out, const float* in, long n, int R, int C, long diag, hipStream_t s);
void k_embedding(float* out, const float* w, const long long* idx, long n_idx, int D, hipStream_t s);
void k_reflect_pad(float* out, const float* in, long rows, long n, long pl, long pr, hipStream_t s);
void k_conv2d(float* out, const float* in, const float* w, const float* bias, const ConvDims& d, hipStream_t s);
void k_stft(void* out, const float* x, const float* win, const float* ct, const float* st, int n_fft, int hop, int n_frames, int n_freq,
            float scale, hipStream_t s);

}  // namespace ops
}  // namespace q3a
