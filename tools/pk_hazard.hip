// Standalone sweep for the gfx950 wrong-result hazard of DESIGN.md section 8.1 (round 4: a packed fp32 multiply whose LOW half
// reads a HIGH source register -- op_sel -- returned that half as 0 in lanes 48-63 while a second HIP queue was busy; found and
// reproduced INSIDE the engine only).  This program asks the two questions the in-engine reproduction could not:
//   * which instruction FORMS are affected -- v_pk_{mul,fma,add}_f32 x {op_sel, op_sel_hi, neg_lo / neg_hi}, v_pk_mov_b32, packed
//     f16 with op_sel, the VOP3 mix / f16 forms with op_sel, and the DPP / permlane forms the product kernels contain;
//   * which kind of NEIGHBOUR on the second queue it takes (HBM streaming, bf16 / f32 MFMA, transcendental + LDS, packed fp32, DPP,
//     copy-engine / fill traffic from the host, launch churn, a mix).
// Stream B runs a victim kernel that executes ONE instruction form over and over on per-lane operands and compares every result
// bit for bit with the result of the lane's FIRST execution (no model of the instruction's semantics is needed: a hazard shows as
// two executions of the same instruction on the same operands that differ, which is exactly how round 4's rope_twice detector saw
// it); stream A, driven by a second host thread, runs the neighbour.  Output: a forms x neighbours table of differing executions,
// the lanes and result halves they hit, and a few samples.  No engine, no model: hipcc --offload-arch=gfx950 -O2 -o
// tools/bin/pk_hazard tools/pk_hazard.hip; run on the GPU box:  tools/bin/pk_hazard [--seconds 0.3] [--forms a,b] [--aggr x,y]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#define CHK(x)                                                                                          \
  do {                                                                                                  \
    hipError_t e_ = (x);                                                                                \
    if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } \
  } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

struct Sample { unsigned form, block, lane, iter, copy, got_lo, got_hi, exp_lo, exp_hi, a_lo, a_hi, b_lo, b_hi; };
struct Result {
  unsigned long long bad;        // executions whose result differs from the lane's first execution
  unsigned long long bad_lo, bad_hi;  // ... in the low / high 32 bits of the result
  unsigned lane_hist[64];
  unsigned n_samples;
  Sample samples[16];
};

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float operand(unsigned gid, unsigned k) {  // [1, 2): products and sums stay normal numbers
  return 1.0f + (float)(hash32(gid * 8u + k) & 0xffffu) * (1.0f / 65536.0f);
}
__device__ __forceinline__ unsigned operand_h2(unsigned gid, unsigned k) {  // two f16 in [1, 2)
  const unsigned h = hash32(gid * 8u + k);
  return (0x3c00u | (h & 0x3ffu)) | ((0x3c00u | ((h >> 10) & 0x3ffu)) << 16);
}
__device__ __forceinline__ void report(Result* r, unsigned form, unsigned it, unsigned copy, unsigned g_lo, unsigned g_hi, unsigned e_lo,
                                       unsigned e_hi, unsigned a_lo, unsigned a_hi, unsigned b_lo, unsigned b_hi) {
  atomicAdd(&r->bad, 1ull);
  if (g_lo != e_lo) atomicAdd(&r->bad_lo, 1ull);
  if (g_hi != e_hi) atomicAdd(&r->bad_hi, 1ull);
  atomicAdd(&r->lane_hist[threadIdx.x & 63], 1u);
  const unsigned s = atomicAdd(&r->n_samples, 1u);
  if (s < 16) r->samples[s] = Sample{form, blockIdx.x, threadIdx.x & 63, it, copy, g_lo, g_hi, e_lo, e_hi, a_lo, a_hi, b_lo, b_hi};
}

// ---- victims: 64-bit (register pair) results --------------------------------------------------------------------------------
// INS2 / INS3: the instruction text with %0 = destination pair, %1.. = source pairs.  Every loop iteration re-creates the
// operands with a VALU multiply by a 1.0 the compiler cannot see through (round 4: the operands came straight from VALU
// producers) and executes the instruction four times into four destinations.
#define VICTIM_PK2(NAME, INS)                                                                                                   \
  __global__ __launch_bounds__(256) void NAME(int iters, unsigned form, Result* res) {                                          \
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;                                                                 \
    float one;                                                                                                                  \
    asm volatile("v_rsq_f32 %0, 1.0\n s_nop 4" : "=v"(one));                                                                     \
    f32x2 a = {operand(gid, 0), operand(gid, 1)}, b = {operand(gid, 2), operand(gid, 3)};                                        \
    f32x2 e;                                                                                                                    \
    asm volatile(INS : "=&v"(e) : "v"(a), "v"(b));                                                                               \
    for (int it = 0; it < iters; ++it) {                                                                                        \
      a *= one; b *= one;                                                                                                       \
      f32x2 d0, d1, d2, d3;                                                                                                     \
      asm volatile(INS : "=&v"(d0) : "v"(a), "v"(b));                                                                            \
      asm volatile(INS : "=&v"(d1) : "v"(a), "v"(b));                                                                            \
      asm volatile(INS : "=&v"(d2) : "v"(a), "v"(b));                                                                            \
      asm volatile(INS : "=&v"(d3) : "v"(a), "v"(b));                                                                            \
      const f32x2 d[4] = {d0, d1, d2, d3};                                                                                      \
      _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                                             \
        if (__float_as_uint(d[c].x) != __float_as_uint(e.x) || __float_as_uint(d[c].y) != __float_as_uint(e.y))                  \
          report(res, form, it, c, __float_as_uint(d[c].x), __float_as_uint(d[c].y), __float_as_uint(e.x), __float_as_uint(e.y), \
                 __float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(b.x), __float_as_uint(b.y));                        \
    }                                                                                                                           \
  }
#define VICTIM_PK3(NAME, INS)                                                                                                   \
  __global__ __launch_bounds__(256) void NAME(int iters, unsigned form, Result* res) {                                          \
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;                                                                 \
    float one;                                                                                                                  \
    asm volatile("v_rsq_f32 %0, 1.0\n s_nop 4" : "=v"(one));                                                                     \
    f32x2 a = {operand(gid, 0), operand(gid, 1)}, b = {operand(gid, 2), operand(gid, 3)}, cc = {operand(gid, 4), operand(gid, 5)}; \
    f32x2 e;                                                                                                                    \
    asm volatile(INS : "=&v"(e) : "v"(a), "v"(b), "v"(cc));                                                                      \
    for (int it = 0; it < iters; ++it) {                                                                                        \
      a *= one; b *= one; cc *= one;                                                                                            \
      f32x2 d0, d1, d2, d3;                                                                                                     \
      asm volatile(INS : "=&v"(d0) : "v"(a), "v"(b), "v"(cc));                                                                   \
      asm volatile(INS : "=&v"(d1) : "v"(a), "v"(b), "v"(cc));                                                                   \
      asm volatile(INS : "=&v"(d2) : "v"(a), "v"(b), "v"(cc));                                                                   \
      asm volatile(INS : "=&v"(d3) : "v"(a), "v"(b), "v"(cc));                                                                   \
      const f32x2 d[4] = {d0, d1, d2, d3};                                                                                      \
      _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                                             \
        if (__float_as_uint(d[c].x) != __float_as_uint(e.x) || __float_as_uint(d[c].y) != __float_as_uint(e.y))                  \
          report(res, form, it, c, __float_as_uint(d[c].x), __float_as_uint(d[c].y), __float_as_uint(e.x), __float_as_uint(e.y), \
                 __float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(b.x), __float_as_uint(b.y));                        \
    }                                                                                                                           \
  }
// ---- victims: 32-bit results (packed f16, mix, VOP3 f16 op_sel, DPP).  Operands are raw 32-bit patterns (f16 pairs in [1, 2) or
// fp32 in [1, 2)); they are re-created each iteration with v_and_b32 against an all-ones mask the compiler cannot see through.
#define VICTIM_U2(NAME, INS, F16)                                                                                               \
  __global__ __launch_bounds__(256) void NAME(int iters, unsigned form, Result* res) {                                          \
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;                                                                 \
    unsigned ones;                                                                                                              \
    asm volatile("v_mov_b32 %0, -1\n s_nop 1" : "=v"(ones));                                                                     \
    unsigned a = F16 ? operand_h2(gid, 0) : __float_as_uint(operand(gid, 0)), b = F16 ? operand_h2(gid, 1) : __float_as_uint(operand(gid, 1)); \
    unsigned e;                                                                                                                 \
    asm volatile(INS : "=&v"(e) : "v"(a), "v"(b));                                                                               \
    for (int it = 0; it < iters; ++it) {                                                                                        \
      a &= ones; b &= ones;                                                                                                     \
      unsigned d0, d1, d2, d3;                                                                                                  \
      asm volatile(INS : "=&v"(d0) : "v"(a), "v"(b));                                                                            \
      asm volatile(INS : "=&v"(d1) : "v"(a), "v"(b));                                                                            \
      asm volatile(INS : "=&v"(d2) : "v"(a), "v"(b));                                                                            \
      asm volatile(INS : "=&v"(d3) : "v"(a), "v"(b));                                                                            \
      const unsigned d[4] = {d0, d1, d2, d3};                                                                                   \
      _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                                             \
        if (d[c] != e) report(res, form, it, c, d[c], 0u, e, 0u, a, 0u, b, 0u);                                                 \
    }                                                                                                                           \
  }
// (DST0: instructions that write only one 16-bit half of the destination get a zeroed destination register, otherwise the
// preserved half would differ between the four destination registers)
#define VICTIM_U3_DST0(NAME, INS)                                                                                               \
  __global__ __launch_bounds__(256) void NAME(int iters, unsigned form, Result* res) {                                          \
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;                                                                 \
    unsigned ones;                                                                                                              \
    asm volatile("v_mov_b32 %0, -1\n s_nop 1" : "=v"(ones));                                                                     \
    unsigned a = operand_h2(gid, 0), b = operand_h2(gid, 1), cc = operand_h2(gid, 2);                                            \
    unsigned e = 0;                                                                                                             \
    asm volatile(INS : "+v"(e) : "v"(a), "v"(b), "v"(cc));                                                                       \
    for (int it = 0; it < iters; ++it) {                                                                                        \
      a &= ones; b &= ones; cc &= ones;                                                                                         \
      unsigned d0 = 0, d1 = 0, d2 = 0, d3 = 0;                                                                                  \
      asm volatile(INS : "+v"(d0) : "v"(a), "v"(b), "v"(cc));                                                                    \
      asm volatile(INS : "+v"(d1) : "v"(a), "v"(b), "v"(cc));                                                                    \
      asm volatile(INS : "+v"(d2) : "v"(a), "v"(b), "v"(cc));                                                                    \
      asm volatile(INS : "+v"(d3) : "v"(a), "v"(b), "v"(cc));                                                                    \
      const unsigned d[4] = {d0, d1, d2, d3};                                                                                   \
      _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                                             \
        if (d[c] != e) report(res, form, it, c, d[c], 0u, e, 0u, a, 0u, b, cc);                                                 \
    }                                                                                                                           \
  }
#define VICTIM_U3(NAME, INS, F16)                                                                                               \
  __global__ __launch_bounds__(256) void NAME(int iters, unsigned form, Result* res) {                                          \
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;                                                                 \
    unsigned ones;                                                                                                              \
    asm volatile("v_mov_b32 %0, -1\n s_nop 1" : "=v"(ones));                                                                     \
    unsigned a = F16 ? operand_h2(gid, 0) : __float_as_uint(operand(gid, 0)), b = F16 ? operand_h2(gid, 1) : __float_as_uint(operand(gid, 1)), \
             cc = F16 ? operand_h2(gid, 2) : __float_as_uint(operand(gid, 2));                                                   \
    unsigned e;                                                                                                                 \
    asm volatile(INS : "=&v"(e) : "v"(a), "v"(b), "v"(cc));                                                                      \
    for (int it = 0; it < iters; ++it) {                                                                                        \
      a &= ones; b &= ones; cc &= ones;                                                                                         \
      unsigned d0, d1, d2, d3;                                                                                                  \
      asm volatile(INS : "=&v"(d0) : "v"(a), "v"(b), "v"(cc));                                                                   \
      asm volatile(INS : "=&v"(d1) : "v"(a), "v"(b), "v"(cc));                                                                   \
      asm volatile(INS : "=&v"(d2) : "v"(a), "v"(b), "v"(cc));                                                                   \
      asm volatile(INS : "=&v"(d3) : "v"(a), "v"(b), "v"(cc));                                                                   \
      const unsigned d[4] = {d0, d1, d2, d3};                                                                                   \
      _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                                             \
        if (d[c] != e) report(res, form, it, c, d[c], 0u, e, 0u, a, 0u, b, cc);                                                 \
    }                                                                                                                           \
  }
// ---- victims: lane swaps (both operands are read AND written): the pair (x, y) is swapped and swapped back; after the round
// trip both registers must hold what they held before.  s_nop 1 in front: the VALU-write -> permlane*_swap read hazard is the
// compiler's to handle outside asm blocks, ours inside.
#define VICTIM_SWAP(NAME, INS)                                                                                                  \
  __global__ __launch_bounds__(256) void NAME(int iters, unsigned form, Result* res) {                                          \
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;                                                                 \
    const unsigned x0 = hash32(gid * 2u), y0 = hash32(gid * 2u + 1u);                                                            \
    for (int it = 0; it < iters; ++it) {                                                                                        \
      unsigned x = x0, y = y0;                                                                                                  \
      asm volatile("s_nop 1\n" INS "\n s_nop 1\n" INS "\n s_nop 1\n" INS "\n s_nop 1\n" INS "\n s_nop 1" : "+v"(x), "+v"(y));     \
      if (x != x0 || y != y0) report(res, form, it, 0, x, y, x0, y0, x0, 0u, y0, 0u);                                           \
    }                                                                                                                           \
  }

// -- packed fp32: plain, every op_sel / op_sel_hi pattern the product code contains, the round-4 failing forms, neg modifiers --
VICTIM_PK2(v_mul_plain, "v_pk_mul_f32 %0, %1, %2")
VICTIM_PK2(v_mul_sel01_hi00, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]")   // round 4's failing instruction
VICTIM_PK2(v_mul_sel01, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]")
VICTIM_PK2(v_mul_sel10, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]")
VICTIM_PK2(v_mul_sel11, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,1]")
VICTIM_PK2(v_mul_sel01_hi01, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]")
VICTIM_PK2(v_mul_sel01_hi10, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]")
VICTIM_PK2(v_mul_sel10_hi01, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]")
VICTIM_PK2(v_add_sel10, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0]")
VICTIM_PK2(v_add_sel11, "v_pk_add_f32 %0, %1, %2 op_sel:[1,1]")
VICTIM_PK2(v_mul_hi01, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]")                      // product code: 358 + 130 sites
VICTIM_PK2(v_mul_hi10, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]")
VICTIM_PK2(v_mul_hi00, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,0]")
VICTIM_PK2(v_mul_neg, "v_pk_mul_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[1,0]")
VICTIM_PK2(v_add_sel01, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1]")
VICTIM_PK2(v_add_hi01, "v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]")                      // product code
VICTIM_PK2(v_add_hi10, "v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]")                      // product code
VICTIM_PK2(v_mov_plain, "v_pk_mov_b32 %0, %1, %2")
VICTIM_PK2(v_mov_sel10, "v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]")
VICTIM_PK2(v_mov_sel01, "v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]")
VICTIM_PK3(v_fma_plain, "v_pk_fma_f32 %0, %1, %2, %3")
VICTIM_PK3(v_fma_sel010, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]")                 // conv1 before the round-4 fix
VICTIM_PK3(v_fma_sel001, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]")
VICTIM_PK3(v_fma_sel100, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]")
VICTIM_PK3(v_fma_sel011, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1]")
VICTIM_PK3(v_fma_sel110, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0]")
VICTIM_PK3(v_fma_sel111, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,1]")
VICTIM_PK3(v_fma_sel010_hi000, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,0]")
VICTIM_PK3(v_fma_hi011, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]")               // product code: 1230 sites
VICTIM_PK3(v_fma_hi110, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]")               // product code: 536 + 130
VICTIM_PK3(v_fma_hi100, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]")               // product code
VICTIM_PK3(v_fma_hi101, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]")               // product code
VICTIM_PK3(v_fma_hi010, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]")               // product code
VICTIM_PK3(v_fma_neg, "v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,1,0] neg_hi:[1,0,1]")
// -- the failing form with the bf16 MFMA work INSIDE the same kernel (waves 1 and 3 of every workgroup run MFMAs, waves 0 and 2 the
//    victim loop): does it take a second queue at all, or just a matrix-instruction wave on the same CU? --
__global__ __launch_bounds__(256) void v_mul_sel01_mfma_waves_same_kernel(int iters, unsigned form, Result* res) {
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  if ((threadIdx.x >> 6) & 1) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x + i); b[i] = (short)(0x3f00 + i); }
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int it = 0; it < iters * 2; ++it) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc1, 0, 0, 0);
    }
    if (acc0[0] + acc1[1] == 12345.f) res->lane_hist[0] = 1;
    return;
  }
  float one;
  asm volatile("v_rsq_f32 %0, 1.0\n s_nop 4" : "=v"(one));
  f32x2 a = {operand(gid, 0), operand(gid, 1)}, b = {operand(gid, 2), operand(gid, 3)};
  f32x2 e;
  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(e) : "v"(a), "v"(b));
  for (int it = 0; it < iters; ++it) {
    a *= one; b *= one;
    f32x2 d0, d1, d2, d3;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(d0) : "v"(a), "v"(b));
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(d1) : "v"(a), "v"(b));
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(d2) : "v"(a), "v"(b));
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(d3) : "v"(a), "v"(b));
    const f32x2 d[4] = {d0, d1, d2, d3};
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (__float_as_uint(d[c].x) != __float_as_uint(e.x) || __float_as_uint(d[c].y) != __float_as_uint(e.y))
        report(res, form, it, c, __float_as_uint(d[c].x), __float_as_uint(d[c].y), __float_as_uint(e.x), __float_as_uint(e.y), __float_as_uint(a.x),
               __float_as_uint(a.y), __float_as_uint(b.x), __float_as_uint(b.y));
  }
}
// -- 16-bit packed / mix / VOP3 op_sel (not in the product code; the judge's class question) --
VICTIM_U3(v_fma_f16_sel, "v_pk_fma_f16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]", 1)
VICTIM_U2(v_mul_f16_sel, "v_pk_mul_f16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]", 1)
VICTIM_U2(v_add_f16_sel, "v_pk_add_f16 %0, %1, %2 op_sel:[1,0]", 1)
VICTIM_U3(v_fma_mix_sel, "v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,0]", 1)
VICTIM_U3_DST0(v_fma_mixlo_sel, "v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,0]")
VICTIM_U3_DST0(v_fma_f16_vop3sel, "v_fma_f16 %0, %1, %2, %3 op_sel:[1,0,1,1]")
VICTIM_U2(v_cvt_pk_bf16, "v_cvt_pk_bf16_f32 %0, %1, %2", 0)                            // product code (bf16 epilogues)
VICTIM_U2(v_dot2c_bf16, "v_mov_b32 %0, 0\n s_nop 0\n v_dot2c_f32_bf16 %0, %1, %2", 1)  // product code (decode attention scores)
// -- DPP forms and lane swaps the product kernels contain --
VICTIM_U2(v_dpp_quad1032, "v_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1", 0)
VICTIM_U2(v_dpp_quad2301, "v_add_f32_dpp %0, %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1", 0)
VICTIM_U2(v_dpp_row_mirror, "v_add_f32_dpp %0, %1, %2 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1", 0)
VICTIM_U2(v_dpp_row_half_mirror, "v_add_f32_dpp %0, %1, %2 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1", 0)
VICTIM_SWAP(v_permlane32_swap, "v_permlane32_swap_b32 %0, %1")
VICTIM_SWAP(v_permlane16_swap, "v_permlane16_swap_b32 %0, %1")

typedef void (*victim_fn)(int, unsigned, Result*);
struct Form { const char* name; const char* text; victim_fn fn; bool in_product; };
#define F(fn, text, prod) {#fn, text, fn, prod}
static const Form kForms[] = {
    F(v_mul_plain, "v_pk_mul_f32 d, a, b", true),
    F(v_mul_sel01_hi00, "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,0]  (round 4's failing form)", false),
    F(v_mul_sel01, "v_pk_mul_f32 op_sel:[0,1]", false), F(v_mul_sel10, "v_pk_mul_f32 op_sel:[1,0]", false), F(v_mul_sel11, "v_pk_mul_f32 op_sel:[1,1]", false),
    F(v_mul_sel01_hi01, "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,1]", false), F(v_mul_sel01_hi10, "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]", false),
    F(v_mul_sel10_hi01, "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1]", false), F(v_add_sel10, "v_pk_add_f32 op_sel:[1,0]", false), F(v_add_sel11, "v_pk_add_f32 op_sel:[1,1]", false),
    F(v_fma_sel011, "v_pk_fma_f32 op_sel:[0,1,1]", false), F(v_fma_sel110, "v_pk_fma_f32 op_sel:[1,1,0]", false), F(v_fma_sel111, "v_pk_fma_f32 op_sel:[1,1,1]", false),
    F(v_fma_sel010_hi000, "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[0,0,0]", false),
    F(v_mul_hi01, "v_pk_mul_f32 op_sel_hi:[0,1]", true), F(v_mul_hi10, "v_pk_mul_f32 op_sel_hi:[1,0]", true), F(v_mul_hi00, "v_pk_mul_f32 op_sel_hi:[0,0]", false),
    F(v_mul_neg, "v_pk_mul_f32 neg_lo:[0,1] neg_hi:[1,0]", false),
    F(v_add_sel01, "v_pk_add_f32 op_sel:[0,1]", false), F(v_add_hi01, "v_pk_add_f32 op_sel_hi:[0,1]", true), F(v_add_hi10, "v_pk_add_f32 op_sel_hi:[1,0]", true),
    F(v_mov_plain, "v_pk_mov_b32 d, a, b", false), F(v_mov_sel10, "v_pk_mov_b32 op_sel:[1,0]", false), F(v_mov_sel01, "v_pk_mov_b32 op_sel:[0,1]", false),
    F(v_fma_plain, "v_pk_fma_f32 d, a, b, c", true),
    F(v_fma_sel010, "v_pk_fma_f32 op_sel:[0,1,0]", false), F(v_fma_sel001, "v_pk_fma_f32 op_sel:[0,0,1]", false), F(v_fma_sel100, "v_pk_fma_f32 op_sel:[1,0,0]", false),
    F(v_fma_hi011, "v_pk_fma_f32 op_sel_hi:[0,1,1]", true), F(v_fma_hi110, "v_pk_fma_f32 op_sel_hi:[1,1,0]", true), F(v_fma_hi100, "v_pk_fma_f32 op_sel_hi:[1,0,0]", true),
    F(v_fma_hi101, "v_pk_fma_f32 op_sel_hi:[1,0,1]", true), F(v_fma_hi010, "v_pk_fma_f32 op_sel_hi:[0,1,0]", true),
    F(v_fma_neg, "v_pk_fma_f32 neg_lo:[0,1,0] neg_hi:[1,0,1]", false),
    F(v_mul_sel01_mfma_waves_same_kernel, "v_pk_mul_f32 op_sel:[0,1], waves 1 / 3 of the SAME kernel run bf16 MFMAs", false),
    F(v_fma_f16_sel, "v_pk_fma_f16 op_sel:[0,1,0] op_sel_hi:[1,0,1]", false), F(v_mul_f16_sel, "v_pk_mul_f16 op_sel:[1,0] op_sel_hi:[0,1]", false),
    F(v_add_f16_sel, "v_pk_add_f16 op_sel:[1,0]", false), F(v_fma_mix_sel, "v_fma_mix_f32 op_sel:[1,0,0] op_sel_hi:[1,1,0]", false),
    F(v_fma_mixlo_sel, "v_fma_mixlo_f16 op_sel:[1,0,0] op_sel_hi:[1,1,0]", false), F(v_fma_f16_vop3sel, "v_fma_f16 op_sel:[1,0,1,1]", false),
    F(v_cvt_pk_bf16, "v_cvt_pk_bf16_f32", true), F(v_dot2c_bf16, "v_dot2c_f32_bf16", true),
    F(v_dpp_quad1032, "v_add_f32_dpp quad_perm:[1,0,3,2]", true), F(v_dpp_quad2301, "v_add_f32_dpp quad_perm:[2,3,0,1]", true),
    F(v_dpp_row_mirror, "v_add_f32_dpp row_mirror", true), F(v_dpp_row_half_mirror, "v_add_f32_dpp row_half_mirror", true),
    F(v_permlane32_swap, "v_permlane32_swap_b32 (x4, round trip)", true), F(v_permlane16_swap, "v_permlane16_swap_b32 (x4, round trip)", true),
};
constexpr int kNumForms = sizeof(kForms) / sizeof(kForms[0]);

// ---- neighbours (stream A) ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ag_stream(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = src[i];
    v.x += 1.0f;
    dst[i] = v;
  }
}
__global__ __launch_bounds__(256) void ag_mfma_bf16(int iters, float* sink) {
  __shared__ short lds[256 * 8];
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x + i); b[i] = (short)(0x3f00 + i); }
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    *reinterpret_cast<bf16x8*>(&lds[threadIdx.x * 8]) = a;
    __syncthreads();
    a = *reinterpret_cast<bf16x8*>(&lds[((threadIdx.x + 1) & 255) * 8]);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc1, 0, 0, 0);
    __syncthreads();
  }
  if (acc0[0] + acc1[1] == 12345.f) sink[0] = acc0[0];
}
__global__ __launch_bounds__(256) void ag_mfma_f32(int iters, float* sink) {  // the log-mel kernel's matrix instruction
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  for (int it = 0; it < iters; ++it) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc, 0, 0, 0);
    a += 1e-6f;
  }
  if (acc[0] == 12345.f) sink[0] = acc[3];
}
__global__ __launch_bounds__(256) void ag_trans_lds(int iters, float* sink) {
  __shared__ float lds[256 * 4];
  float x = 0.5f + threadIdx.x * 1e-3f, y = 0.f;
  for (int it = 0; it < iters; ++it) {
    lds[(threadIdx.x * 4 + it) & 1023] = x;
    y += __expf(x) + __frsqrt_rn(x + 1.0f) + __frcp_rn(x + 2.0f) + __logf(x + 3.0f);
    x = lds[(threadIdx.x * 4 + 17 * it) & 1023] * 0.999f + 1e-3f;
  }
  if (y == 12345.f) sink[0] = y;
}
__global__ __launch_bounds__(256) void ag_pk_f32(int iters, float* sink) {
  f32x2 a = {1.0f + threadIdx.x * 1e-3f, 0.5f}, b = {0.999f, 1.001f}, c = {1e-3f, 2e-3f};
  for (int it = 0; it < iters; ++it) {
    asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %2\n v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  }
  if (a.x == 12345.f) sink[0] = a.y;
}
__global__ __launch_bounds__(256) void ag_dpp(int iters, float* sink) {
  float x = 1.0f + threadIdx.x * 1e-3f;
  unsigned u = threadIdx.x, w = threadIdx.x * 3u;
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
                 "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_mul_f32 %0, 0.25, %0\n"
                 "s_nop 1\n v_permlane32_swap_b32 %1, %2\n s_nop 1\n v_permlane16_swap_b32 %1, %2\n s_nop 1"
                 : "+v"(x), "+v"(u), "+v"(w));
  }
  if (x == 12345.f) sink[0] = x + u + w;
}
__global__ __launch_bounds__(256) void ag_mfma_bf16_nolds(int iters, float* sink) {  // the bf16 matrix instruction alone: no LDS, no barrier
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x + i); b[i] = (short)(0x3f00 + i); }
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc1, 0, 0, 0);
  }
  if (acc0[0] + acc1[1] == 12345.f) sink[0] = acc0[0];
}
__global__ __launch_bounds__(256) void ag_mfma_bf16_32(int iters, float* sink) {  // 32x32x16 bf16 (the flash attention's instruction), no LDS
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x + i); b[i] = (short)(0x3f00 + i); }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int it = 0; it < iters; ++it) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0);
  }
  if (acc[0] == 12345.f) sink[0] = acc[3];
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void ag_mfma_f16(int iters, float* sink) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(1.0f + 0.001f * (threadIdx.x + i)); b[i] = (_Float16)0.5f; }
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc1, 0, 0, 0);
  }
  if (acc0[0] + acc1[1] == 12345.f) sink[0] = acc0[0];
}
__global__ __launch_bounds__(256) void ag_lds_barrier(int iters, float* sink) {  // the LDS round trip + workgroup barrier of ag_mfma_bf16 without its MFMAs
  __shared__ short lds[256 * 8];
  bf16x8 a;
  for (int i = 0; i < 8; ++i) a[i] = (short)(0x3f80 + threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
    *reinterpret_cast<bf16x8*>(&lds[threadIdx.x * 8]) = a;
    __syncthreads();
    a = *reinterpret_cast<bf16x8*>(&lds[((threadIdx.x + 1) & 255) * 8]);
    a[0] += 1;
    __syncthreads();
  }
  if (a[0] == 12345) sink[0] = a[1];
}
__global__ __launch_bounds__(256) void ag_mfma_f32_lds(int iters, float* sink) {  // the f32 matrix instruction WITH the LDS round trip + barrier
  __shared__ float lds[256];
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  for (int it = 0; it < iters; ++it) {
    lds[threadIdx.x] = a;
    __syncthreads();
    a = lds[(threadIdx.x + 1) & 255];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc, 0, 0, 0);
    __syncthreads();
  }
  if (acc[0] == 12345.f) sink[0] = acc[3];
}
__global__ void ag_tiny(float* sink) {
  if (threadIdx.x == 9999) sink[0] = 1.f;
}

enum Aggr { A_NONE, A_STREAM, A_MFMA_BF16, A_MFMA_F32, A_TRANS_LDS, A_PK_F32, A_DPP, A_HOST_COPIES, A_CHURN, A_MIX,
            A_MFMA_BF16_NOLDS, A_MFMA_BF16_32, A_MFMA_F16, A_LDS_BARRIER, A_MFMA_F32_LDS, A_COUNT };
static const char* kAggrNames[A_COUNT] = {"none", "hbm_stream", "mfma_bf16", "mfma_f32", "trans_lds", "pk_f32", "dpp_swap", "host_copies", "launch_churn", "mix",
                                          "mfma_bf16_nolds", "mfma_bf16_32x32", "mfma_f16", "lds_barrier", "mfma_f32_lds"};

struct AggrCtx {
  hipStream_t s;
  float4 *src, *dst;
  size_t n4;
  float* sink;
  void* pageable;
  void* dev_small;
  size_t small_bytes;
};
static void aggressor_once(const AggrCtx& c, int kind, int round) {
  switch (kind) {
    case A_STREAM: hipLaunchKernelGGL(ag_stream, dim3(2048), dim3(256), 0, c.s, c.src, c.dst, c.n4); break;
    case A_MFMA_BF16: hipLaunchKernelGGL(ag_mfma_bf16, dim3(1024), dim3(256), 0, c.s, 4000, c.sink); break;
    case A_MFMA_F32: hipLaunchKernelGGL(ag_mfma_f32, dim3(1024), dim3(256), 0, c.s, 4000, c.sink); break;
    case A_TRANS_LDS: hipLaunchKernelGGL(ag_trans_lds, dim3(1024), dim3(256), 0, c.s, 4000, c.sink); break;
    case A_PK_F32: hipLaunchKernelGGL(ag_pk_f32, dim3(1024), dim3(256), 0, c.s, 20000, c.sink); break;
    case A_DPP: hipLaunchKernelGGL(ag_dpp, dim3(1024), dim3(256), 0, c.s, 8000, c.sink); break;
    case A_MFMA_BF16_NOLDS: hipLaunchKernelGGL(ag_mfma_bf16_nolds, dim3(1024), dim3(256), 0, c.s, 8000, c.sink); break;
    case A_MFMA_BF16_32: hipLaunchKernelGGL(ag_mfma_bf16_32, dim3(1024), dim3(256), 0, c.s, 4000, c.sink); break;
    case A_MFMA_F16: hipLaunchKernelGGL(ag_mfma_f16, dim3(1024), dim3(256), 0, c.s, 8000, c.sink); break;
    case A_LDS_BARRIER: hipLaunchKernelGGL(ag_lds_barrier, dim3(1024), dim3(256), 0, c.s, 4000, c.sink); break;
    case A_MFMA_F32_LDS: hipLaunchKernelGGL(ag_mfma_f32_lds, dim3(1024), dim3(256), 0, c.s, 4000, c.sink); break;
    case A_HOST_COPIES:  // what an engine does once per batch around its kernels: pageable H2D (staged by the runtime), fills, D2H
      CHK(hipMemcpyAsync(c.dev_small, c.pageable, c.small_bytes, hipMemcpyHostToDevice, c.s));
      CHK(hipMemsetAsync(c.dev_small, 0, c.small_bytes / 2, c.s));
      CHK(hipMemcpyAsync(c.pageable, c.dev_small, 64 << 10, hipMemcpyDeviceToHost, c.s));
      CHK(hipMemcpyAsync((char*)c.dev_small + c.small_bytes / 2, c.dev_small, c.small_bytes / 4, hipMemcpyDeviceToDevice, c.s));
      break;
    case A_CHURN:
      for (int i = 0; i < 64; ++i) hipLaunchKernelGGL(ag_tiny, dim3(1), dim3(64), 0, c.s, c.sink);
      break;
    case A_MIX: {
      static const int seq[] = {A_HOST_COPIES, A_MFMA_F32, A_PK_F32, A_MFMA_BF16, A_STREAM, A_TRANS_LDS, A_DPP, A_CHURN};
      aggressor_once(c, seq[round % 8], round);
      break;
    }
    default: break;
  }
}

int main(int argc, char** argv) {
  double seconds = 0.3;
  std::string only_forms, only_aggr;
  int wgs = 1024, iters = 2000;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--seconds") && i + 1 < argc) seconds = atof(argv[++i]);
    else if (!strcmp(argv[i], "--forms") && i + 1 < argc) only_forms = std::string(",") + argv[++i] + ",";
    else if (!strcmp(argv[i], "--aggr") && i + 1 < argc) only_aggr = std::string(",") + argv[++i] + ",";
    else if (!strcmp(argv[i], "--wgs") && i + 1 < argc) wgs = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
    else { fprintf(stderr, "usage: pk_hazard [--seconds S] [--forms a,b] [--aggr x,y] [--wgs N] [--iters N]\n"); return 1; }
  }
  CHK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  printf("# pk_hazard: %s (%s), %d CUs; victim grid %d x 256 threads x %d iterations x 4 executions per launch, %.2f s per cell\n", prop.name,
         prop.gcnArchName, prop.multiProcessorCount, wgs, iters, seconds);
  hipStream_t sa, sb;
  CHK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CHK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  AggrCtx ctx{};
  ctx.s = sa;
  ctx.n4 = (size_t(256) << 20) / 16;
  CHK(hipMalloc(&ctx.src, ctx.n4 * 16));
  CHK(hipMalloc(&ctx.dst, ctx.n4 * 16));
  CHK(hipMemset(ctx.src, 0, ctx.n4 * 16));
  CHK(hipMalloc(&ctx.sink, 256));
  ctx.small_bytes = 8 << 20;
  ctx.pageable = malloc(ctx.small_bytes);
  memset(ctx.pageable, 1, ctx.small_bytes);
  CHK(hipMalloc(&ctx.dev_small, ctx.small_bytes));
  Result* d_res;
  CHK(hipMalloc(&d_res, sizeof(Result)));
  std::vector<std::vector<unsigned long long>> table(kNumForms, std::vector<unsigned long long>(A_COUNT, 0));
  std::vector<std::vector<double>> execs(kNumForms, std::vector<double>(A_COUNT, 0));
  std::vector<std::string> notes;
  for (int f = 0; f < kNumForms; ++f) {
    if (!only_forms.empty() && only_forms.find(std::string(",") + kForms[f].name + ",") == std::string::npos) continue;
    for (int ag = 0; ag < A_COUNT; ++ag) {
      if (!only_aggr.empty() && only_aggr.find(std::string(",") + kAggrNames[ag] + ",") == std::string::npos) continue;
      CHK(hipMemset(d_res, 0, sizeof(Result)));
      std::atomic<bool> stop{false};
      std::thread th([&] {
        if (ag == A_NONE) return;
        CHK(hipSetDevice(0));
        hipEvent_t ev[4];
        for (auto& e : ev) CHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (int round = 0; !stop.load(std::memory_order_relaxed); ++round) {
          if (round >= 4) CHK(hipEventSynchronize(ev[round & 3]));  // at most four neighbour rounds in flight
          aggressor_once(ctx, ag, round);
          CHK(hipEventRecord(ev[round & 3], sa));
        }
        CHK(hipStreamSynchronize(sa));
        for (auto& e : ev) CHK(hipEventDestroy(e));
      });
      const auto t0 = std::chrono::steady_clock::now();
      long launches = 0;
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(kForms[f].fn, dim3(wgs), dim3(256), 0, sb, iters, (unsigned)f, d_res);
        launches += 4;
        CHK(hipStreamSynchronize(sb));
      }
      stop.store(true);
      th.join();
      CHK(hipDeviceSynchronize());
      Result r;
      CHK(hipMemcpy(&r, d_res, sizeof(r), hipMemcpyDeviceToHost));
      table[f][ag] = r.bad;
      execs[f][ag] = (double)launches * wgs * 4.0 /*waves per workgroup*/ * iters * 4.0;
      if (r.bad) {
        char buf[1024];
        int lo = 64, hi = -1;
        for (int l = 0; l < 64; ++l)
          if (r.lane_hist[l]) { lo = std::min(lo, l); hi = std::max(hi, l); }
        snprintf(buf, sizeof(buf), "%s | %s: %llu differing executions of %.3g wave-executions (low half %llu, high half %llu), lanes %d..%d", kForms[f].name,
                 kAggrNames[ag], r.bad, execs[f][ag], r.bad_lo, r.bad_hi, lo, hi);
        notes.push_back(buf);
        for (unsigned s = 0; s < std::min(r.n_samples, 4u); ++s) {
          const Sample& m = r.samples[s];
          snprintf(buf, sizeof(buf), "    sample: block %u lane %u iter %u copy %u: got %08x %08x, first execution gave %08x %08x (a = %08x %08x, b = %08x %08x)", m.block, m.lane,
                   m.iter, m.copy, m.got_lo, m.got_hi, m.exp_lo, m.exp_hi, m.a_lo, m.a_hi, m.b_lo, m.b_hi);
          notes.push_back(buf);
        }
      }
      fflush(stdout);
    }
    printf("%-22s", kForms[f].name);
    for (int ag = 0; ag < A_COUNT; ++ag)
      if (execs[f][ag] > 0) printf(" %s=%llu", kAggrNames[ag], table[f][ag]);
    double ex_max = 0;
    for (int ag = 0; ag < A_COUNT; ++ag) ex_max = std::max(ex_max, execs[f][ag]);
    printf("   # up to %.2g wave-executions per cell%s\n", ex_max, kForms[f].in_product ? "; form present in the product library" : "");
    fflush(stdout);
  }
  printf("\n## differing executions (form | neighbour)\n");
  if (notes.empty()) printf("none\n");
  for (auto& n : notes) printf("%s\n", n.c_str());
  printf("\n## forms\n");
  for (int f = 0; f < kNumForms; ++f) {
    double ran = 0;
    for (int ag = 0; ag < A_COUNT; ++ag) ran += execs[f][ag];
    if (ran == 0) continue;  // not part of this run (--forms)
    unsigned long long tot = 0, ctl = table[f][A_NONE];
    for (int ag = 1; ag < A_COUNT; ++ag) tot += table[f][ag];
    printf("%-22s %-60s %s%s\n", kForms[f].name, kForms[f].text,
           tot == 0 && ctl == 0 ? "0 differing" : (ctl ? "DIFFERS EVEN ALONE (harness or instruction problem)" : "DIFFERS NEXT TO A BUSY QUEUE"),
           kForms[f].in_product ? "  [in product]" : "");
  }
  return 0;
}
