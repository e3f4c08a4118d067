// Device-side helpers shared by the gfx950 kernels (wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Phase stamps for tools/phase_probe.hip (probe builds define Q3A_STAMP; the product library never does): thread 0 of a
// workgroup records the 100 MHz wall clock at up to 8 phase boundaries of the kernel.
#ifdef Q3A_STAMP
#define Q3A_STAMP_FIELD unsigned long long* stamp = nullptr;  /* [workgroup][8], null = off */
#define Q3A_STAMP_AT(ptr, wg, slot) do { if ((ptr) && threadIdx.x == 0) (ptr)[(size_t)(wg) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define Q3A_STAMP_FIELD
#define Q3A_STAMP_AT(ptr, wg, slot) do { } while (0)
#endif

// hipcc loads kernel arguments lazily -- a scalar load right in front of each first use -- and on this chip a scalar load
// that misses (the kernarg segment of a fresh dispatch always does) is a 0.3-0.5 us round trip: the stamped timelines of
// tools/phase_probe.hip showed 1.4-2.2 us between a kernel's entry and its last load request, most of it three or four
// such round trips in series.  Q3A_ARG(x) makes x (a kernel-argument expression) resident in SGPRs at that point; a
// block of them at the top of a kernel turns the series into ONE clause that overlaps the address arithmetic.
#define Q3A_ARG(x) asm volatile("" ::"s"(x))

namespace q3a {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;  // MFMA 16x16x32 bf16 A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // MFMA 16x16 accumulator
typedef __attribute__((ext_vector_type(2))) float f32x2_t;    // operand of the packed VALU ops (v_pk_fma_f32)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // MFMA 32x32 accumulator

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t v) { return __uint_as_float(v << 16); }
// fp32 -> bf16, round-to-nearest-even: one v_cvt_pk_bf16_f32 per PAIR on gfx950 (the integer formulation --
// add 0x7fff + lsb, shift -- costs ~5 VALU instructions per value, and a wave64 VALU instruction holds its SIMD for
// 4 cycles: in the conversion-heavy kernels (skinny GEMM with fp32 x, attention P packing, bf16 epilogues) that was
// a large share of the issue slots)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) { return pack_bf16x2(f, 0.f) & 0xffffu; }
// two bf16 packed in one dword -> two floats
__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// 16 B of a weight stream that the kernel reads exactly once per launch: non-temporal load, so the stream does not
// displace the activations and partials the next kernels re-read from L2 (measured: decode 80.9 -> 76.9 ms per 100
// tokens at batch 1; -DQ3A_NT_STREAM=0 restores plain loads for A/B runs).
#ifndef Q3A_NT_STREAM
#define Q3A_NT_STREAM 1
#endif
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ uint4 ld_stream16(const void* p) {
#if Q3A_NT_STREAM
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
#else
  const u32x4_t v = *reinterpret_cast<const u32x4_t*>(p);
#endif
  return make_uint4(v.x, v.y, v.z, v.w);
}

// Cross-row exchanges on gfx950 without the LDS crossbar (ds_bpermute: address VALU + LDS op + wait per value):
// v_permlane16_swap / v_permlane32_swap with both operands equal leave lane l with its own value in one result and
// the value of lane l^16 / l^32 in the other, so an xor-16 / xor-32 all-reduce step is one swap + one op.
__device__ __forceinline__ float xor16_sum(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_max(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// DPP inside the 16-lane rows, permlane swaps across them: no LDS crossbar, every lane gets the result
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}
// sum over the aligned group of 8 lanes
__device__ __forceinline__ float row8_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return v;
}
// Every lane ends with the same bits: (r0 + r1) + (r2 + r3) up to commutation.
__device__ __forceinline__ float wave_sum_fast(float v) { return xor32_sum(xor16_sum(row16_sum(v))); }
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_fast(v); }
__device__ __forceinline__ float wave_max(float v) { return xor32_max(xor16_max(row16_max(v))); }

// erf-form GELU (reference: Tensor::gelu -> gelu("none"), src/tensor.rs:350-352)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// The same two activations for epilogues whose result is rounded to bf16 anyway (default mode): erff / expf + a
// division cost 30-50 VALU instructions per element -- more issue slots than the whole K loop of a short-K GEMM tile.
// erfc by Abramowitz & Stegun 7.1.28: erfc(z) = (1 + a1 z + ... + a6 z^6)^-16 (|error| <= 3e-7), z = |x| / sqrt 2 >= 0 -- six
// fused multiply-adds, four squarings and ONE hardware reciprocal (round 3 used 7.1.26: five FMAs + a reciprocal + an
// exponential, i.e. two quarter-rate transcendentals per element; the GELU of conv1 / conv2 / conv3 / fc1 is VALU time the
// MFMA pipe waits for).  The negative branch uses erfc directly, so there is no 1 - (1 - tiny) cancellation; for large |x| the
// 16th power overflows to +inf and the reciprocal returns 0.  |gelu_fast - gelu_erf| < 4e-7 * max(1, |x|) (checked over
// [-12, 12] in steps of 1.2e-5: 8.2e-7 at x = 4.03).
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  float p = 0.0000430638f;
  p = p * z + 0.0002765672f;
  p = p * z + 0.0001520143f;
  p = p * z + 0.0092705272f;
  p = p * z + 0.0422820123f;
  p = p * z + 0.0705230784f;
  p = p * z + 1.0f;
  p = p * p; p = p * p; p = p * p; p = p * p;
  const float pe = __builtin_amdgcn_rcpf(p);  // erfc(|x| / sqrt 2)
  return 0.5f * x * (x >= 0.f ? 2.0f - pe : pe);
}
// the same on a channel pair: every multiply / fma is one packed instruction for both values
__device__ __forceinline__ f32x2_t gelu_fast2(f32x2_t x) {
  const f32x2_t z = f32x2_t{fabsf(x.x), fabsf(x.y)} * f32x2_t{0.70710678118654752440f, 0.70710678118654752440f};
  f32x2_t p = f32x2_t{0.0000430638f, 0.0000430638f} * z + f32x2_t{0.0002765672f, 0.0002765672f};
  p = p * z + f32x2_t{0.0001520143f, 0.0001520143f};
  p = p * z + f32x2_t{0.0092705272f, 0.0092705272f};
  p = p * z + f32x2_t{0.0422820123f, 0.0422820123f};
  p = p * z + f32x2_t{0.0705230784f, 0.0705230784f};
  p = p * z + f32x2_t{1.0f, 1.0f};
  p = p * p; p = p * p; p = p * p; p = p * p;
  const f32x2_t pe = {__builtin_amdgcn_rcpf(p.x), __builtin_amdgcn_rcpf(p.y)};
  const f32x2_t hx = f32x2_t{0.5f, 0.5f} * x;
  return f32x2_t{hx.x * (x.x >= 0.f ? 2.0f - pe.x : pe.x), hx.y * (x.y >= 0.f ? 2.0f - pe.y : pe.y)};
}
__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// 1 / sqrt(v) for the RMSNorm scale of the decode-step kernels in the default mode: one v_rsq_f32 (1 ulp) instead of
// sqrt + an IEEE division (~20 dependent instructions on the tail of a 4 us kernel); the precise mode keeps 1 / sqrtf.
#ifndef Q3A_FAST_EPILOGUE
#define Q3A_FAST_EPILOGUE 1  // A/B builds: 0 restores the exact forms everywhere
#endif
__device__ __forceinline__ float rstd_of(float mean_sq_plus_eps, bool fast) {
  return (Q3A_FAST_EPILOGUE && fast) ? __builtin_amdgcn_rsqf(mean_sq_plus_eps) : 1.0f / sqrtf(mean_sq_plus_eps);
}
// SiLU (reference: Tensor::silu, src/tensor.rs:354-356)
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float silu_sel(float x, bool fast) { return (Q3A_FAST_EPILOGUE && fast) ? silu_fast(x) : silu_f(x); }

// The lane id from a statement hipcc may not merge with an earlier one: what is derived from it is recomputed where it is used instead
// of living in registers across a loop that does not need it (the epilogue constants and row-pointer inputs of k_gemm256.hip's walk)
__device__ __forceinline__ int lane_id_fresh() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

__device__ __forceinline__ float lane_bcast(float v, int lane_uniform) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_uniform));
}

// ---- gfx950 hazard: packed fp32 with op_sel (profiles/r4_pk_op_sel_hazard.txt) ---------------------------
// A v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 whose LOW result half reads the HIGH register of a source pair (an `op_sel:[..1..]`
// modifier: the operand swap / high-register broadcast hipcc emits for scalar code it SLP-packs, or for {x, x} splats of the second
// register of a 64-bit load) occasionally returns that half as if the product were 0, in lanes 48-63 only, while a second HIP queue
// keeps the GPU busy -- reproduced with a hand-written instruction (the same multiply with the operands swapped in registers and
// no op_sel never fails; round 4's in-engine variants of head_norm_rope, removed in round 6 -- the standalone sweep
// tools/pk_hazard.hip -> profiles/r5_pk_hazard.txt pins the failing set).  Consequences for this library:
//   * it is built with -fno-slp-vectorize (qwen3_asr_rs_amd/build.py), so packed fp32 only comes from explicit f32x2_t code;
//   * the build scans the ISA of every kernel and fails on any packed fp32 instruction with an op_sel bit set (build.py scan_isa);
//   * no arithmetic in inline asm as a way around the packing: hipcc's hazard recogniser does not look inside asm blocks (a
//     v_mul_f32 written in asm straight behind the v_rsq_f32 that produced its operand -- the gfx940+ trans-op forwarding
//     hazard -- read a stale value now and then: round 4, the decode attention's q/k norm, caught by the graph == eager id test).
// a value the compiler must keep in a VGPR of its own (not the high half of a 64-bit pair): {x, x} splats of it use op_sel_hi only
// (an empty asm block: no instruction is emitted)
__device__ __forceinline__ float own_vgpr(float x) {
  asm volatile("" : "+v"(x));
  return x;
}
// RoPE on the rotate_half partners (n1, n2) = dims (d, d + 64): x*cos + rotate_half(x)*sin, rotate_half = cat(-x2, x1)
// (src/layers.rs:361-375): two multiplies + two fused multiply-adds, scalar (see above)
__device__ __forceinline__ void rope_rotate(float n1, float n2, float c, float sn, float& x1, float& x2) {
  const float a = sn * n2, b = sn * n1;
  x1 = __builtin_fmaf(c, n1, -a);
  x2 = __builtin_fmaf(c, n2, b);
}

// One wave per 128-wide head vector; the lane owns dims (lane, lane+64) = the rotate_half partners.
// per-head RMSNorm (src/layers.rs:303-304,48-54) then RoPE (layers.rs:361-375)
// plain scalar C++; with the SLP vectoriser off it stays scalar, and the build's ISA scan checks that
__device__ __forceinline__ void head_norm_rope(float& x1, float& x2, const float* __restrict__ w, float eps,
                                               const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                               int pos, int lane) {
  const float ss = wave_sum_fast(x1 * x1 + x2 * x2);
  const float rstd = 1.0f / sqrtf(ss / 128.0f + eps);
  const float c = cos_t[(size_t)pos * 64 + lane], sn = sin_t[(size_t)pos * 64 + lane];
  const float n1 = (x1 * rstd) * w[lane], n2 = (x2 * rstd) * w[lane + 64];
  rope_rotate(n1, n2, c, sn, x1, x2);
}

// KV-cache element types: bf16 (default) or f32 (precise mode)
template <typename T> struct KvIo;
template <> struct KvIo<float> {
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float round(float v) { return v; }
};
template <> struct KvIo<uint16_t> {
  static __device__ __forceinline__ float load(const uint16_t* p) { return bf16_bits_to_f32(*p); }
  static __device__ __forceinline__ void store(uint16_t* p, float v) { *p = (uint16_t)f32_to_bf16_bits(v); }
  static __device__ __forceinline__ float round(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
};

}  // namespace q3a
