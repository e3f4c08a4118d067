/* q3asr_ops.h -- op-level C ABI of libq3asr_hip.so: the compatibility veneer that lets the reference's
 * `struct Tensor` (src/tensor.rs:120-126) grow a third `#[cfg(feature = "hip")]` arm next to tch and MLX
 * (SURVEY.md section 8b / 8f-4).  Shaped like mlx-c as bound by src/backend/mlx/ffi.rs:60-574: opaque handles, every op
 * `int32_t q3a_op_xxx(q3a_array** out, inputs..., q3a_stream* s)` returning 0 or an error (q3a_ops_last_error()).
 *
 * This is NOT the fast path: the engine API of q3asr.h runs the fused kernels (one arena, bf16 MFMA GEMMs, hipGraph
 * decode).  The veneer is an eager fp32 tensor library on the same GPU -- value-semantic, one kernel (or one existing
 * engine kernel) per op -- so that an unchanged src/layers.rs / mel.rs / audio_encoder.rs / text_decoder.rs runs under
 * `feature = "hip"` and produces the numbers of the tch-CPU arm to fp32 rounding.  The Rust side (uncompiled here, no
 * toolchain) is under integration/rust/src/backend/hip/.
 *
 * Semantics follow the tch arm, method by method (the line of src/tensor.rs each entry replaces is quoted).
 *   - arrays are immutable values with shared, reference-counted storage; shape ops return views (tch semantics);
 *   - compute ops take F32 (index ops: I64) device arrays; host arrays (from_slice, before to_device) support only
 *     metadata, to_dtype, to_device and data extraction -- there is no CPU compute path in this library;
 *   - one in-order stream per device (s == NULL), like the MLX arm's global stream (src/backend/mlx/stream.rs:6).
 */
#ifndef Q3ASR_OPS_H
#define Q3ASR_OPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct q3a_array q3a_array;      /* refcounted tensor handle; Rust RAII: Drop -> q3a_array_free (backend/mlx/array.rs:8-29) */
typedef struct q3a_stream q3a_stream;    /* NULL = the default in-order stream of the array's device */

/* DType (src/tensor.rs:9-17).  Q3A_C64 exists only as the result of q3a_op_stft(return_complex) and the input of abs. */
enum { Q3A_F32 = 0, Q3A_F16 = 1, Q3A_BF16 = 2, Q3A_I64 = 3, Q3A_I32 = 4, Q3A_BOOL = 5, Q3A_C64 = 6 };
/* Device (src/tensor.rs:83-93): -1 = Cpu, i >= 0 = Gpu(i). */
#define Q3A_CPU (-1)

const char* q3a_ops_last_error(void);     /* message of the calling thread's last failed q3a_op_* / q3a_array_* call */
int32_t q3a_ops_synchronize(int32_t device);

/* ---- handles ------------------------------------------------------------------------------------------- */
void q3a_array_free(q3a_array* a);                                     /* Drop */
int32_t q3a_op_shallow_clone(q3a_array** out, const q3a_array* a);     /* tensor.rs:467 / Clone :134-141 */
int32_t q3a_array_ndim(const q3a_array* a);                            /* dim() :235 */
int32_t q3a_array_shape(const q3a_array* a, int64_t* shape, int32_t cap); /* size() :223 (size3 :227, size4 :231 on the Rust side); returns ndim */
int32_t q3a_array_dtype(const q3a_array* a);                           /* kind() :459 */
int32_t q3a_array_device(const q3a_array* a);                          /* device() :463 */
int64_t q3a_array_numel(const q3a_array* a);

/* ---- creation -------------------------------------------------------------------------------------------- */
int32_t q3a_op_from_slice_f32(q3a_array** out, const float* data, int64_t n);    /* :162 (host array, 1-D) */
int32_t q3a_op_from_slice_i64(q3a_array** out, const int64_t* data, int64_t n);  /* :166 */
/* weights.rs loader arm: raw little-endian bytes of `dtype` with `shape`, copied to `device` (Q3A_CPU keeps it on the host) */
int32_t q3a_op_from_bytes(q3a_array** out, const void* data, int32_t dtype, const int64_t* shape, int32_t ndim, int32_t device);
int32_t q3a_op_zeros(q3a_array** out, const int64_t* shape, int32_t ndim, int32_t dtype, int32_t device);               /* :170 */
int32_t q3a_op_ones(q3a_array** out, const int64_t* shape, int32_t ndim, int32_t dtype, int32_t device);                /* :175 */
int32_t q3a_op_full(q3a_array** out, const int64_t* shape, int32_t ndim, double val, int32_t dtype, int32_t device);    /* :180 */
int32_t q3a_op_arange(q3a_array** out, int64_t start, int64_t end, int32_t device);                                      /* :185 (I64) */
int32_t q3a_op_arange_f(q3a_array** out, double start, double end, double step, int32_t dtype, int32_t device);         /* :191 */
int32_t q3a_op_cat(q3a_array** out, const q3a_array* const* tensors, int32_t n, int64_t dim, q3a_stream* s);            /* :199 */
int32_t q3a_op_stack(q3a_array** out, const q3a_array* const* tensors, int32_t n, int64_t dim, q3a_stream* s);          /* :204 */
int32_t q3a_op_embedding(q3a_array** out, const q3a_array* weight, const q3a_array* indices, q3a_stream* s);            /* :209 */
int32_t q3a_op_hann_window(q3a_array** out, int64_t size, int32_t device);                                               /* :215 (periodic, F32) */

/* ---- shape (views; tch semantics) -------------------------------------------------------------------------- */
int32_t q3a_op_view(q3a_array** out, const q3a_array* a, const int64_t* shape, int32_t ndim);                 /* :239 (fails on non-viewable strides) */
int32_t q3a_op_reshape(q3a_array** out, const q3a_array* a, const int64_t* shape, int32_t ndim, q3a_stream* s); /* :243 (one -1 allowed; copies when needed) */
int32_t q3a_op_narrow(q3a_array** out, const q3a_array* a, int64_t dim, int64_t start, int64_t len);          /* :247 */
int32_t q3a_op_unsqueeze(q3a_array** out, const q3a_array* a, int64_t dim);                                   /* :251 */
int32_t q3a_op_squeeze_dim(q3a_array** out, const q3a_array* a, int64_t dim);                                 /* :255 */
int32_t q3a_op_transpose(q3a_array** out, const q3a_array* a, int64_t dim0, int64_t dim1);                    /* :259 */
int32_t q3a_op_permute(q3a_array** out, const q3a_array* a, const int64_t* dims, int32_t ndim);               /* :263 */
int32_t q3a_op_expand(q3a_array** out, const q3a_array* a, const int64_t* size, int32_t ndim);                /* :267 (-1 keeps a dimension) */
int32_t q3a_op_contiguous(q3a_array** out, const q3a_array* a, q3a_stream* s);                                /* :271 */
int32_t q3a_op_tr(q3a_array** out, const q3a_array* a);                                                       /* :275 (2-D transpose) */
int32_t q3a_op_get(q3a_array** out, const q3a_array* a, int64_t index);                                       /* :279 (select on dim 0) */
int32_t q3a_op_select(q3a_array** out, const q3a_array* a, int64_t dim, int64_t index);                       /* :283 */

/* ---- arithmetic / math (F32, broadcasting as torch) -------------------------------------------------------- */
int32_t q3a_op_matmul(q3a_array** out, const q3a_array* a, const q3a_array* b, q3a_stream* s);                /* :289 */
int32_t q3a_op_add(q3a_array** out, const q3a_array* a, const q3a_array* b, q3a_stream* s);                   /* operators :960-1161 */
int32_t q3a_op_sub(q3a_array** out, const q3a_array* a, const q3a_array* b, q3a_stream* s);
int32_t q3a_op_mul(q3a_array** out, const q3a_array* a, const q3a_array* b, q3a_stream* s);
int32_t q3a_op_div(q3a_array** out, const q3a_array* a, const q3a_array* b, q3a_stream* s);
int32_t q3a_op_maximum(q3a_array** out, const q3a_array* a, const q3a_array* b, q3a_stream* s);               /* :305 */
int32_t q3a_op_add_scalar(q3a_array** out, const q3a_array* a, double v, q3a_stream* s);                      /* T + f64 */
int32_t q3a_op_sub_scalar(q3a_array** out, const q3a_array* a, double v, q3a_stream* s);                      /* T - f64 */
int32_t q3a_op_mul_scalar(q3a_array** out, const q3a_array* a, double v, q3a_stream* s);                      /* T * f64 */
int32_t q3a_op_div_scalar(q3a_array** out, const q3a_array* a, double v, q3a_stream* s);                      /* T / f64 */
int32_t q3a_op_add_inplace(q3a_array* a, const q3a_array* b, q3a_stream* s);                                  /* += (writes through a's view) */
int32_t q3a_op_fill_inplace(q3a_array* a, double v, q3a_stream* s);                                           /* fill_ :382 */
int32_t q3a_op_pow_scalar(q3a_array** out, const q3a_array* a, double e, q3a_stream* s);                      /* :293 */
int32_t q3a_op_neg(q3a_array** out, const q3a_array* a, q3a_stream* s);                                       /* :297 */
int32_t q3a_op_clamp_min(q3a_array** out, const q3a_array* a, double min, q3a_stream* s);                     /* :301 */
int32_t q3a_op_abs(q3a_array** out, const q3a_array* a, q3a_stream* s);                                       /* :311 (C64 -> F32 magnitude) */
int32_t q3a_op_square(q3a_array** out, const q3a_array* a, q3a_stream* s);                                    /* :315 */
int32_t q3a_op_sqrt(q3a_array** out, const q3a_array* a, q3a_stream* s);                                      /* :319 */
int32_t q3a_op_rsqrt(q3a_array** out, const q3a_array* a, q3a_stream* s);                                     /* :323 (1 / sqrt(x): sqrt then reciprocal, as the tch arm) */
int32_t q3a_op_log10(q3a_array** out, const q3a_array* a, q3a_stream* s);                                     /* :328 */
int32_t q3a_op_sin(q3a_array** out, const q3a_array* a, q3a_stream* s);                                       /* :332 */
int32_t q3a_op_cos(q3a_array** out, const q3a_array* a, q3a_stream* s);                                       /* :336 */
int32_t q3a_op_exp(q3a_array** out, const q3a_array* a, q3a_stream* s);                                       /* :340 */
int32_t q3a_op_softmax(q3a_array** out, const q3a_array* a, int64_t dim, q3a_stream* s);                      /* :346 (computed in f32) */
int32_t q3a_op_gelu(q3a_array** out, const q3a_array* a, q3a_stream* s);                                      /* :350 gelu("none") = erf form */
int32_t q3a_op_silu(q3a_array** out, const q3a_array* a, q3a_stream* s);                                      /* :354 */
int32_t q3a_op_mean_dim(q3a_array** out, const q3a_array* a, const int64_t* dims, int32_t ndims, int32_t keepdim, q3a_stream* s); /* :360 */
int32_t q3a_op_max(q3a_array** out, const q3a_array* a, q3a_stream* s);                                       /* :364 (0-d global max) */
int32_t q3a_op_argmax(q3a_array** out, const q3a_array* a, int64_t dim, int32_t keepdim, q3a_stream* s);      /* :370 (I64, first index on ties) */
int32_t q3a_op_triu(q3a_array** out, const q3a_array* a, int64_t diagonal, q3a_stream* s);                    /* :374 */
int32_t q3a_op_slice_scatter(q3a_array** out, const q3a_array* a, const q3a_array* src, int64_t dim, int64_t start, int64_t end,
                             int64_t step, q3a_stream* s);                                                    /* :378 */
int32_t q3a_op_layer_norm(q3a_array** out, const q3a_array* a, const int64_t* normalized_shape, int32_t n, const q3a_array* weight,
                          const q3a_array* bias, double eps, q3a_stream* s);                                  /* :388 (weight / bias nullable) */
int32_t q3a_op_conv2d(q3a_array** out, const q3a_array* a, const q3a_array* weight, const q3a_array* bias, const int64_t* stride,
                      const int64_t* padding, const int64_t* dilation, int64_t groups, q3a_stream* s);        /* :406 (NCHW, groups == 1) */
int32_t q3a_op_reflection_pad1d(q3a_array** out, const q3a_array* a, const int64_t* pad, q3a_stream* s);      /* :423 (pad[2] on the last dim) */
int32_t q3a_op_stft(q3a_array** out, const q3a_array* a, int64_t n_fft, int64_t hop_length, int64_t win_length, const q3a_array* window,
                    int32_t normalized, int32_t onesided, int32_t return_complex, q3a_stream* s);             /* :427 (center = false; 1-D input) */

/* ---- type / device / extraction ---------------------------------------------------------------------------- */
int32_t q3a_op_to_dtype(q3a_array** out, const q3a_array* a, int32_t dtype, q3a_stream* s);                   /* :451 */
int32_t q3a_op_to_device(q3a_array** out, const q3a_array* a, int32_t device, q3a_stream* s);                 /* :455 */
int32_t q3a_array_int64_value(const q3a_array* a, const int64_t* indices, int32_t n, int64_t* value);         /* :473 (synchronises) */
int32_t q3a_array_f64_value(const q3a_array* a, const int64_t* indices, int32_t n, double* value);            /* :477 */
int32_t q3a_array_to_vec_f32(const q3a_array* a, float* dst, int64_t cap);                                    /* :481 (flattened, row-major) */

#ifdef __cplusplus
}
#endif
#endif /* Q3ASR_OPS_H */
