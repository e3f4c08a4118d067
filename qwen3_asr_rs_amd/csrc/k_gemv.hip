// GEMV family for the single-token decode step on gfx950: y[b][n] = sum_k x[b][k] * W[n][k].
//
// Replaces Linear::forward at one token per sequence (src/layers.rs:74-80) with the surrounding
// RMSNorm (layers.rs:48-54), bias, residual add (layers.rs:442-463) and SiLU(gate)*up (layers.rs:396-400)
// fused, plus the lm_head matmul + argmax of the greedy loop (src/text_decoder.rs:111-112,
// src/inference.rs:161).  HBM-bound: every weight byte is streamed exactly once per token, 16 B per lane
// per load (a wave reads 1 KiB of one weight row per instruction); weights are bf16, x stays fp32, so
// the products are fp32-exact and only the summation order differs from the reference.
//
// Everything in these kernels is latency (a 4-12 MB matrix is 16-48 KB per CU), so the order of work is
//   1. issue the first PF weight loads of every row the wave owns -- they do not depend on x;
//   2. fetch x (L2-resident) while those are in flight.  With one sequence (NB == 1) each lane loads
//      exactly the x slice its weight columns need straight into registers: no LDS, no barrier, and
//      sum(x^2) for the RMSNorm is a plain wavefront reduction because a wave covers all of K.  For
//      NB in {2,4} x is staged through LDS once per workgroup;
//   3. FMA.  The RMSNorm scale is a scalar per row of x, so it multiplies the finished dot product
//      instead of every element (the reference's (x*rstd)*w differs by fp32 rounding only);
//   4. wavefront reduction on DPP (no LDS crossbar), fused epilogue.
// x can also be assembled on the fly from the flash-decoding partials of k_dattn.hip (o_proj input).
#include "dev.h"
#include "kernels.h"


namespace q3a {

int gemv_rows_per_wave(const GemvArgs& a) {  // physical weight rows per wave
  const int logical = (a.mode == 2) ? a.N / 2 : a.N;
  // tuning knobs for A/B runs (rows per wave for mid-size and small matrices); defaults are the measured best
  static const int mid = [] { const char* e = getenv("Q3A_GEMV_PR_MID"); return e ? atoi(e) : 2; }();
  static const int small = [] { const char* e = getenv("Q3A_GEMV_PR_SMALL"); return e ? atoi(e) : 1; }();
  if (logical >= 32768 && a.K <= 2048) return 4;
  if (logical >= 4096 || a.mode == 2) return (mid == 4 || mid == 2) ? mid : 2;
  return (small == 1 || small == 2 || small == 4) ? small : 1;
}
int gemv_blocks(const GemvArgs& a) {
  const int pr = gemv_rows_per_wave(a);
  const int rows_per_wave = (a.mode == 2) ? pr / 2 : pr;
  const int logical = (a.mode == 2) ? a.N / 2 : a.N;
  return (logical + 4 * rows_per_wave - 1) / (4 * rows_per_wave);
}

namespace {

// Flash-decoding merge (o_proj input).  attn_scales: every workgroup first turns the per-split softmax
// statistics (m, l) of k_dattn.hip into one scale factor per (sequence, head, split),
// f = exp(m - M) / L, in LDS; attn_combined8 then builds 8 consecutive elements (k .. k+7, inside one head)
// of the attention output as sum_split f * o_split -- loads that depend on nothing but the partial buffers.
constexpr int ATTN_F_MAX = 1024;  // NB * heads * nsplit entries (three LDS tables of this size)
struct AttnTables {
  float m[ATTN_F_MAX], l[ATTN_F_MAX], f[ATTN_F_MAX];
};
__device__ __forceinline__ void attn_scales(const GemvArgs& a, int nb, AttnTables& t) {
  const int ns = a.attn_nsplit, n = nb * a.attn_heads * ns;
  // one thread per (sequence, head, split): all statistics are fetched in ONE round of parallel loads
  for (int i = threadIdx.x; i < n; i += 256) {
    t.m[i] = a.attn_pm[i];
    t.l[i] = a.attn_pl[i];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    const int base = (i / ns) * ns;
    float M = -INFINITY;
    for (int sp = 0; sp < ns; ++sp) M = fmaxf(M, t.m[base + sp]);
    float L = 0.f;
    for (int sp = 0; sp < ns; ++sp) {
      const float m = t.m[base + sp];
      if (m != -INFINITY) L += t.l[base + sp] * expf(m - M);
    }
    const float mi = t.m[i];
    t.f[i] = (mi == -INFINITY) ? 0.f : expf(mi - M) / L;
  }
  __syncthreads();
}
__device__ __forceinline__ void attn_combined8(const GemvArgs& a, int b, int k, const AttnTables& t, float (&x)[8]) {
  const int h = k >> 7, d = k & 127;
  const int ns = a.attn_nsplit;
  const int base = (b * a.attn_heads + h) * ns;
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = 0.f;
  for (int sp0 = 0; sp0 < ns; sp0 += 4) {  // 4 splits (8 independent 16-B loads) in flight at a time
    float4 o0[4], o1[4];
    float f[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int sp = sp0 + j, spc = sp < ns ? sp : ns - 1;  // clamp the address, zero the weight
      f[j] = sp < ns ? t.f[base + spc] : 0.f;
      o0[j] = *reinterpret_cast<const float4*>(a.attn_po + (size_t)(base + spc) * 128 + d);
      o1[j] = *reinterpret_cast<const float4*>(a.attn_po + (size_t)(base + spc) * 128 + d + 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // empty splits carry f = 0 and zeroed partials
      x[0] += o0[j].x * f[j]; x[1] += o0[j].y * f[j]; x[2] += o0[j].z * f[j]; x[3] += o0[j].w * f[j];
      x[4] += o1[j].x * f[j]; x[5] += o1[j].y * f[j]; x[6] += o1[j].z * f[j]; x[7] += o1[j].w * f[j];
    }
  }
}

// Single-sequence form of the merge: no table, no barrier.  The 16 lanes that share a head read that head's (m, l)
// statistics themselves (same-address loads, one request) together with the partial outputs -- every load is
// independent, one L2 round trip -- and rescale online over chunks of 4 splits.
// FAST: hardware exponential (default mode); the precise mode keeps expf.
template <bool FAST, int CH>
__device__ __forceinline__ void attn_merge8_ch(const GemvArgs& a, int b, int k, float (&x)[8]) {
  auto ex = [](float v) { return FAST ? __expf(v) : expf(v); };
  const int h = k >> 7, d = k & 127;
  const int ns = a.attn_nsplit;
  const int base = (b * a.attn_heads + h) * ns;
  float M = -INFINITY, L = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = 0.f;
  // Chunks of CH splits: every load of a chunk is requested before anything is combined.  CH = 4 up to 512 keys, 8 beyond:
  // contexts up to 1024 keys cost ONE L2 round trip (with chunks of 4 the fifth split of a 513..640-key context was a
  // second one: 2.4 us of merge in this kernel, profiles/r3_phase_probe_decode_layers.txt).
  for (int sp0 = 0; sp0 < ns; sp0 += CH) {
    float m[CH], l[CH];
    float4 o0[CH], o1[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int sp = sp0 + j, spc = sp < ns ? sp : ns - 1;  // clamp the address, void the statistics
      m[j] = a.attn_pm[base + spc];
      l[j] = a.attn_pl[base + spc];
      if (sp >= ns) m[j] = -INFINITY;
      o0[j] = *reinterpret_cast<const float4*>(a.attn_po + (size_t)(base + spc) * 128 + d);
      o1[j] = *reinterpret_cast<const float4*>(a.attn_po + (size_t)(base + spc) * 128 + d + 4);
    }
    // every request above stays above: without the fence hipcc sinks the l / o loads below the `continue` that only needs m --
    // m first, a wait, then the rest: two dependent round trips where the source has one (ISA: global_load x5, s_waitcnt
    // vmcnt(0), branch, global_load x21)
#ifndef Q3A_NO_MERGE_FENCE  // A/B builds only
    asm volatile("" ::: "memory");
#endif
    float Mn = M;
#pragma unroll
    for (int j = 0; j < CH; ++j) Mn = fmaxf(Mn, m[j]);
    if (Mn == -INFINITY) continue;  // only empty splits so far
    if (sp0 > 0) {  // rescale what the earlier chunks accumulated (nothing in the common <= 8-split case)
      const float sc = (M == -INFINITY) ? 0.f : ex(M - Mn);
      L *= sc;
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] *= sc;
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const float f = ex(m[j] - Mn);  // empty split: exp(-inf) = 0 (Mn is finite here)
      L += l[j] * f;
      x[0] += o0[j].x * f; x[1] += o0[j].y * f; x[2] += o0[j].z * f; x[3] += o0[j].w * f;
      x[4] += o1[j].x * f; x[5] += o1[j].y * f; x[6] += o1[j].z * f; x[7] += o1[j].w * f;
    }
    M = Mn;
  }
  const float inv = 1.0f / L;  // split 0 always holds at least the current token
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] *= inv;
}
template <bool FAST>
__device__ __forceinline__ void attn_merge8(const GemvArgs& a, int b, int k, float (&x)[8]) {
  if (a.attn_nsplit <= 4) attn_merge8_ch<FAST, 4>(a, b, k, x);  // kernel-argument condition: uniform
  else attn_merge8_ch<FAST, 8>(a, b, k, x);
}

template <int PR>
__device__ __forceinline__ void gemv_rows(const GemvArgs& a, int g, int (&prow)[PR]) {
#pragma unroll
  for (int i = 0; i < PR; ++i) {
    if (a.mode == 2) {
      const int j = g * (PR / 2) + (i >> 1);  // logical row; W rows come in [16 gate | 16 up] blocks
      prow[i] = (j / 16) * 32 + (j % 16) + ((i & 1) ? 16 : 0);
    } else {
      prow[i] = g * PR + i;
    }
    if (prow[i] >= a.N) prow[i] = -1;
  }
}

// 8 weights x 8 activations; the multiply-adds run two wide (v_pk_fma_f32): 8 unpack + 4 packed FMA + 2 adds
__device__ __forceinline__ float dot8(const uint4& w, const float (&x)[8], float s) {
  f32x2_t a = f32x2_t{bf16lo(w.x), bf16hi(w.x)} * f32x2_t{x[0], x[1]};
  a += f32x2_t{bf16lo(w.y), bf16hi(w.y)} * f32x2_t{x[2], x[3]};
  a += f32x2_t{bf16lo(w.z), bf16hi(w.z)} * f32x2_t{x[4], x[5]};
  a += f32x2_t{bf16lo(w.w), bf16hi(w.w)} * f32x2_t{x[6], x[7]};
  return s + (a.x + a.y);
}

// epilogue shared by both variants; acc holds the finished (rstd-scaled) dot products, valid on every lane
template <int NB, int PR>
__device__ __forceinline__ void gemv_epilogue(const GemvArgs& a, int g, const int (&prow)[PR], float (&acc)[PR][NB],
                                              float (*am_v)[NB], int (*am_i)[NB]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (a.mode == 3) {  // logits + argmax partial (first-index tie-break: rows ascend with i, wave, block)
    float bv[NB];
    int bi[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { bv[b] = -INFINITY; bi[b] = 0x7fffffff; }
#pragma unroll
    for (int i = 0; i < PR; ++i) {
      if (prow[i] < 0) continue;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float v = acc[i][b];
        if (a.bias) v += a.bias[prow[i]];
        if (lane == 0 && a.out) a.out[(size_t)b * a.ldo + prow[i]] = v;
        if (v > bv[b]) { bv[b] = v; bi[b] = prow[i]; }
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int b = 0; b < NB; ++b) { am_v[wave][b] = bv[b]; am_i[wave][b] = bi[b]; }
    }
    __syncthreads();
    if (tid < NB) {
      float v = am_v[0][tid];
      int ix = am_i[0][tid];
      for (int w = 1; w < 4; ++w)
        if (am_v[w][tid] > v || (am_v[w][tid] == v && am_i[w][tid] < ix)) { v = am_v[w][tid]; ix = am_i[w][tid]; }
      a.part_val[(size_t)tid * a.part_stride + (g >> 2)] = v;
      a.part_idx[(size_t)tid * a.part_stride + (g >> 2)] = ix;
    }
    return;
  }
  if (lane != 0) return;
  if (a.mode != 2) {
#pragma unroll
    for (int i = 0; i < PR; ++i) {
      if (prow[i] < 0) continue;
      const int n = prow[i];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float v = acc[i][b];
        if (a.bias) v += a.bias[n];
        if (a.mode == 1) v += a.resid[(size_t)b * a.ldo + n];
        a.out[(size_t)b * a.ldo + n] = v;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i + 1 < PR; i += 2) {
      if (prow[i] < 0) continue;
      const int j = g * (PR / 2) + (i >> 1);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float gv = acc[i][b], uv = acc[i + 1][b];
        if (a.bias) { gv += a.bias[prow[i]]; uv += a.bias[prow[i + 1]]; }
        a.out[(size_t)b * a.ldo + j] = silu_sel(gv, a.fast_math != 0) * uv;
      }
    }
  }
}

// ---- NB == 1: x lives in registers (KI k-iterations of 512 columns, K <= 512*KI) ---------------------
// Straight-line by construction: RMS (norm fused) and ATTN (x = merged attention partials) are template flags and no
// load sits under a lane-divergent branch -- out-of-range rows / columns are clamped to valid addresses and voided
// arithmetically.  (With `row < N ? load : 0` guards hipcc put every weight load into its own exec-masked block with
// an s_waitcnt vmcnt(0) between them: half of the stream was only requested after the other half had landed.)
// MF (ATTN only, at most MF key splits, MF in {4, 8}): the merge's loads -- per-split statistics and partial rows -- are requested
// IN FRONT of the weight stream.  Loads return in order: behind the weights (MF = 0) the merge starts when the last weight chunk
// has landed and its exp / scale / LDS round trip / barrier sit exposed between the stream and the FMAs; in front, the partials are
// back after one L2 round trip and the merged vector is in LDS while the weights are still in flight.
// XL (round 6; !ATTN; used for the norm-fused projections at K <= 1024): x (and the norm weight) are fetched ONCE per workgroup -- KI / 2 16-byte loads per lane instead of
// 2 KI (4 KI with the norm weight) -- and handed to the four waves through LDS: every wave needs the whole vector, and in the
// register form each of them pulls it through the CU's texture-address path again (two thirds of a qkv workgroup's load
// instructions were x and norm-weight re-reads).  Same values, same arithmetic.
template <int PR, int KI, bool RMS, bool ATTN, int MF = 0, bool XL = false>
__global__ __launch_bounds__(256) void gemv1_kernel(GemvArgs a) {
  static_assert(!XL || !ATTN, "the merged attention vector already goes through LDS");
  __shared__ __attribute__((aligned(16))) float xl_x[XL ? KI * 512 : 4], xl_w[(XL && RMS) ? KI * 512 : 4];
  __shared__ float am_v[4][1];
  __shared__ int am_i[4][1];
  __shared__ __attribute__((aligned(16))) float x_s[ATTN ? KI * 512 : 4];  // only the attention merge goes through LDS
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  Q3A_STAMP_AT(a.stamp, blockIdx.x, 0);  // entry
  const int K = a.K;
  // (an XCD-contiguous block -> row remap, so that each output line is dirtied in one L2 only, measured 0.6 % slower)
  const int g = blockIdx.x * 4 + wave;
  int prow[PR];
  gemv_rows<PR>(a, g, prow);
  bool kin[KI];
  int kk[KI];
#pragma unroll
  for (int it = 0; it < KI; ++it) {
    const int k = lane * 8 + it * 512;
    kin[it] = k < K;
    kk[it] = kin[it] ? k : 0;
  }
  float4 xr[KI][2], nr[RMS ? KI : 1][2];
  uint4 wq[KI][PR];
  constexpr int XLN = (KI + 1) / 2;  // 16-byte pieces of x per lane in the XL form (256 lanes x 4 floats = 1024 columns per round)
  float4 xl_rx[XL ? XLN : 1], xl_rw[(XL && RMS) ? XLN : 1];
  auto request_x = [&]() {  // L2 hits
    if (ATTN) return;
    if constexpr (XL) {
#pragma unroll
      for (int j = 0; j < XLN; ++j) {
        const int k = (tid + j * 256) * 4, kc = k < K ? k : 0;
        xl_rx[j] = *reinterpret_cast<const float4*>(a.x + kc);
        if (RMS) xl_rw[j] = *reinterpret_cast<const float4*>(a.rms_w + kc);
      }
      return;
    }
#pragma unroll
    for (int it = 0; it < KI; ++it) {
      xr[it][0] = *reinterpret_cast<const float4*>(a.x + kk[it]);
      xr[it][1] = *reinterpret_cast<const float4*>(a.x + kk[it] + 4);
      if (RMS) {
        nr[it][0] = *reinterpret_cast<const float4*>(a.rms_w + kk[it]);
        nr[it][1] = *reinterpret_cast<const float4*>(a.rms_w + kk[it] + 4);
      }
    }
  };
  auto request_w = [&]() {  // the HBM stream: all KI * PR chunks of the wave's rows in flight at once
#pragma unroll
    for (int it = 0; it < KI; ++it)
#pragma unroll
      for (int i = 0; i < PR; ++i) {
        const int row = prow[i] >= 0 ? prow[i] : a.N - 1;
        wq[it][i] = ld_stream16(a.W + (size_t)row * K + kk[it]);
      }
  };
  // Loads return in order.  With a fused norm, x and the norm weight go first: they are back, squared and scaled long
  // before the first weight chunk lands (-0.2 us).  Without one the weights go first (x-first measured +0.2..0.4 us).
  constexpr int MCH = MF > 0 ? MF : 1;
  float mg_m[MCH], mg_l[MCH];
  float4 mg_o0[MCH], mg_o1[MCH];
  const int mg_k = lane * 8 + wave * 512;            // this wave merges elements mg_k .. mg_k + 7 (it == wave)
  const bool mg_on = wave < KI && mg_k < K;          // wave-uniform up to the last partial wave of a short K (clamped below)
  if constexpr (ATTN && MF > 0) {
    const int kc = mg_k < K ? mg_k : 0;
    const int h = kc >> 7, d = kc & 127, ns = a.attn_nsplit, base = h * ns;
    // A head's statistics are ns consecutive floats: 16-byte loads (4-byte aligned; entries past the head's own belong to the next
    // head or to the allocation's 256-byte rounding and are voided) instead of one 4-byte load per split -- and NO branch around
    // loads: a uniform `if` whose arms load made hipcc wait at the join before requesting anything else (tests/test_isa_hazards.py).
    typedef float __attribute__((ext_vector_type(4), aligned(4))) f4u_t;
#pragma unroll
    for (int j = 0; j < MCH; j += 4) {
      const f4u_t m4 = *reinterpret_cast<const f4u_t*>(a.attn_pm + base + j), l4 = *reinterpret_cast<const f4u_t*>(a.attn_pl + base + j);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mg_m[j + e] = j + e < ns ? m4[e] : -INFINITY;
        mg_l[j + e] = l4[e];
      }
    }
#pragma unroll
    for (int j = 0; j < MCH; ++j) {
      const int spc = j < ns ? j : ns - 1;
      mg_o0[j] = *reinterpret_cast<const float4*>(a.attn_po + (size_t)(base + spc) * 128 + d);
      mg_o1[j] = *reinterpret_cast<const float4*>(a.attn_po + (size_t)(base + spc) * 128 + d + 4);
    }
    __builtin_amdgcn_sched_barrier(0);  // the partials' requests stay in front of the weight stream
  }
  if (RMS || XL) { request_x(); request_w(); } else { request_w(); request_x(); }
  // epilogue operands (bias, residual) ride behind the stream instead of costing an L2 round trip at the very end
  float bv[PR], rv[PR];
#pragma unroll
  for (int i = 0; i < PR; ++i) {
    const int n = prow[i] >= 0 ? prow[i] : 0;
    bv[i] = a.bias ? a.bias[n] : 0.f;
    rv[i] = a.mode == 1 ? a.resid[n] : 0.f;
  }
  __builtin_amdgcn_sched_barrier(0);  // every request is issued before anything is waited for
  Q3A_STAMP_AT(a.stamp, blockIdx.x, 1);  // every load requested
  if constexpr (ATTN && MF > 0) {  // merge from the registers requested up front (same arithmetic as attn_merge8_ch<FAST, MF>, one chunk)
    if (mg_on) {
      const bool fast = a.attn_fast_exp != 0;
      float Mn = -INFINITY;
#pragma unroll
      for (int j = 0; j < MCH; ++j) Mn = fmaxf(Mn, mg_m[j]);
      float L = 0.f, v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
      for (int j = 0; j < MCH; ++j) {
        const float f = fast ? __expf(mg_m[j] - Mn) : expf(mg_m[j] - Mn);  // empty split: exp(-inf) = 0 (split 0 is never empty)
        L += mg_l[j] * f;
        v[0] += mg_o0[j].x * f; v[1] += mg_o0[j].y * f; v[2] += mg_o0[j].z * f; v[3] += mg_o0[j].w * f;
        v[4] += mg_o1[j].x * f; v[5] += mg_o1[j].y * f; v[6] += mg_o1[j].z * f; v[7] += mg_o1[j].w * f;
      }
      const float inv = 1.0f / L;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= inv;
      *reinterpret_cast<float4*>(x_s + mg_k) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(x_s + mg_k + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < KI; ++it) {
      xr[it][0] = *reinterpret_cast<const float4*>(x_s + kk[it]);
      xr[it][1] = *reinterpret_cast<const float4*>(x_s + kk[it] + 4);
    }
  } else if (ATTN) {  // each wave merges a quarter of the vector, once per block
#pragma unroll
    for (int it = 0; it < KI; ++it) {
      if ((it & 3) == wave && kin[it]) {
        float v[8];
        if (a.attn_fast_exp) attn_merge8<true>(a, 0, kk[it], v); else attn_merge8<false>(a, 0, kk[it], v);
        *reinterpret_cast<float4*>(x_s + kk[it]) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(x_s + kk[it] + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < KI; ++it) {
      xr[it][0] = *reinterpret_cast<const float4*>(x_s + kk[it]);
      xr[it][1] = *reinterpret_cast<const float4*>(x_s + kk[it] + 4);
    }
  }
  if constexpr (XL) {  // hand the vector to the four waves (the weights are still in flight)
#pragma unroll
    for (int j = 0; j < XLN; ++j) {
      const int k = (tid + j * 256) * 4;
      if (k < KI * 512) {
        *reinterpret_cast<float4*>(xl_x + k) = xl_rx[j];
        if (RMS) *reinterpret_cast<float4*>(xl_w + k) = xl_rw[j];
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < KI; ++it) {
      xr[it][0] = *reinterpret_cast<const float4*>(xl_x + kk[it]);
      xr[it][1] = *reinterpret_cast<const float4*>(xl_x + kk[it] + 4);
      if (RMS) {
        nr[it][0] = *reinterpret_cast<const float4*>(xl_w + kk[it]);
        nr[it][1] = *reinterpret_cast<const float4*>(xl_w + kk[it] + 4);
      }
    }
  }
  Q3A_STAMP_AT(a.stamp, blockIdx.x, 2);  // (ATTN: merged vector in LDS)
  float x[KI][8];
  float ss = 0.f;  // (packing sum(x^2) and x * w_norm two wide measured 0.6 % slower on the whole step)
#pragma unroll
  for (int it = 0; it < KI; ++it) {
    const float4 v0 = xr[it][0], v1 = xr[it][1];
    const float xv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) x[it][e] = (K == KI * 512 || kin[it]) ? xv[e] : 0.f;  // columns beyond K contribute nothing
    if (RMS) {
      const float4 w0 = nr[it][0], w1 = nr[it][1];
      const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) { ss += x[it][e] * x[it][e]; x[it][e] *= wv[e]; }
    }
  }
  float acc[PR][1];
#pragma unroll
  for (int i = 0; i < PR; ++i) acc[i][0] = 0.f;
#pragma unroll
  for (int it = 0; it < KI; ++it)
#pragma unroll
    for (int i = 0; i < PR; ++i) acc[i][0] = dot8(wq[it][i], x[it], acc[i][0]);
  if (wave == 0) asm volatile("" ::"v"(acc[0][0]));  // (the stamp below sits behind the FMAs of wave 0, i.e. behind its weights)
  Q3A_STAMP_AT(a.stamp, blockIdx.x, 3);  // weights landed, dot products done
  float rstd = 1.0f;
  if (RMS) rstd = rstd_of(wave_sum_fast(ss) / (float)K + a.eps, a.fast_math != 0);  // a wave covers all of K
#pragma unroll
  for (int i = 0; i < PR; ++i) acc[i][0] = wave_sum_fast(acc[i][0]) * rstd + bv[i];
  // epilogue (bias already added; acc is valid on every lane)
  if (a.mode == 3) {  // logits + argmax partial (first-index tie-break: rows ascend with i, wave, block)
    float best = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < PR; ++i) {
      if (prow[i] < 0) continue;
      if (lane == 0 && a.out) a.out[prow[i]] = acc[i][0];
      if (acc[i][0] > best) { best = acc[i][0]; bi = prow[i]; }
    }
    if (lane == 0) { am_v[wave][0] = best; am_i[wave][0] = bi; }
    __syncthreads();
    if (tid == 0) {
      float v = am_v[0][0];
      int ix = am_i[0][0];
      for (int w = 1; w < 4; ++w)
        if (am_v[w][0] > v || (am_v[w][0] == v && am_i[w][0] < ix)) { v = am_v[w][0]; ix = am_i[w][0]; }
      a.part_val[blockIdx.x] = v;
      a.part_idx[blockIdx.x] = ix;
    }
    return;
  }
  if (lane != 0) return;
  if (a.mode != 2) {
#pragma unroll
    for (int i = 0; i < PR; ++i)
      if (prow[i] >= 0) a.out[prow[i]] = acc[i][0] + rv[i];
  } else {
#pragma unroll
    for (int i = 0; i + 1 < PR; i += 2)
      if (prow[i] >= 0) a.out[g * (PR / 2) + (i >> 1)] = silu_sel(acc[i][0], a.fast_math != 0) * acc[i + 1][0];
  }
  Q3A_STAMP_AT(a.stamp, blockIdx.x, 4);  // reduced and stored
}

// ---- NB in {2, 4}: x staged through LDS once per workgroup ---------------------------------------------
template <int NB, int PR, int PF>
__global__ __launch_bounds__(256) void gemvn_kernel(GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [NB][K]
  __shared__ float red[NB][4];
  __shared__ float am_v[4][NB];
  __shared__ int am_i[4][NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K;
  const int g = blockIdx.x * 4 + wave;
  int prow[PR];
  gemv_rows<PR>(a, g, prow);
  uint4 wq[PF][PR];
#pragma unroll
  for (int it = 0; it < PF; ++it) {
    const int k = lane * 8 + it * 512;
#pragma unroll
    for (int i = 0; i < PR; ++i)  // unconditional (clamped) loads: no exec-masked blocks, no waits between requests
      wq[it][i] = ld_stream16(a.W + (size_t)(prow[i] >= 0 ? prow[i] : a.N - 1) * K + (k < K ? k : 0));
  }
  __shared__ AttnTables f_s;
  if (a.attn_po) attn_scales(a, NB, f_s);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float ss = 0.f;
    for (int k = tid * 8; k < K; k += 2048) {
      float x[8];
      if (a.attn_po) {
        attn_combined8(a, b, k, f_s, x);
      } else {
        const float4 v0 = *reinterpret_cast<const float4*>(a.x + (size_t)b * a.ldx + k);
        const float4 v1 = *reinterpret_cast<const float4*>(a.x + (size_t)b * a.ldx + k + 4);
        x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
      }
      if (a.rms_w) {
        const float4 w0 = *reinterpret_cast<const float4*>(a.rms_w + k);
        const float4 w1 = *reinterpret_cast<const float4*>(a.rms_w + k + 4);
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) { ss += x[e] * x[e]; x[e] *= wv[e]; }
      }
      *reinterpret_cast<float4*>(xs + (size_t)b * K + k) = make_float4(x[0], x[1], x[2], x[3]);
      *reinterpret_cast<float4*>(xs + (size_t)b * K + k + 4) = make_float4(x[4], x[5], x[6], x[7]);
    }
    if (a.rms_w) {
      ss = wave_sum_fast(ss);
      if (lane == 0) red[b][wave] = ss;
    }
  }
  __syncthreads();
  float acc[PR][NB];
#pragma unroll
  for (int i = 0; i < PR; ++i)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[i][b] = 0.f;
  auto fma_all = [&](const uint4 (&w)[PR], int k) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float4 xa = *reinterpret_cast<const float4*>(xs + (size_t)b * K + k);
      const float4 xb = *reinterpret_cast<const float4*>(xs + (size_t)b * K + k + 4);
      const float x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
      for (int i = 0; i < PR; ++i) acc[i][b] = dot8(w[i], x, acc[i][b]);
    }
  };
#pragma unroll
  for (int it = 0; it < PF; ++it) {
    const int k = lane * 8 + it * 512;
    if (k < K) fma_all(wq[it], k);
  }
  for (int k = lane * 8 + PF * 512; k < K; k += 512) {
    uint4 w[PR];
#pragma unroll
    for (int i = 0; i < PR; ++i)
      w[i] = ld_stream16(a.W + (size_t)(prow[i] >= 0 ? prow[i] : a.N - 1) * K + k);
    fma_all(w, k);
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float rstd = 1.0f;
    if (a.rms_w) rstd = rstd_of((red[b][0] + red[b][1] + red[b][2] + red[b][3]) / (float)K + a.eps, a.fast_math != 0);
#pragma unroll
    for (int i = 0; i < PR; ++i) acc[i][b] = wave_sum_fast(acc[i][b]) * rstd;
  }
  gemv_epilogue<NB, PR>(a, g, prow, acc, am_v, am_i);
}

template <int PR, int KI>
void launch1k(const GemvArgs& a, hipStream_t s) {
  const dim3 grid(gemv_blocks(a)), block(256);
  // up to 8 key splits (contexts up to 1024 keys): the merge's partials go in FRONT of the weight stream (round 6: -1.2 % per step at
  // 0.6B x 1 and x 2, neutral at 1.7B; profiles/r6_ab_merge_first.txt); beyond, the chunked merge behind it
  if (a.attn_po && KI <= 4 && a.attn_nsplit <= 4) hipLaunchKernelGGL((gemv1_kernel<PR, KI, false, true, 4>), grid, block, 0, s, a);
  else if (a.attn_po && KI <= 4 && a.attn_nsplit <= 8) hipLaunchKernelGGL((gemv1_kernel<PR, KI, false, true, 8>), grid, block, 0, s, a);
  else if (a.attn_po) hipLaunchKernelGGL((gemv1_kernel<PR, KI, false, true>), grid, block, 0, s, a);
  else if (a.rms_w) {
    // Norm-fused projections at K <= 1024 (qkv, gate / up at hidden 1024): x and the norm weight once per workgroup through LDS
    // (round 6: 609.7 -> 595.1 us per step at 0.6B, ids identical; profiles/r6_ab_gemv_x_lds.txt).  Not where it was measured to
    // lose: K = 2048 (+18 %: 32 KiB of LDS per workgroup halves the residency of the 1536-workgroup gate / up launch), the down
    // projection (+4 %: x in front of the weights delays them), the lm_head (+1.3 %: a barrier in each of 9496 workgroups).
    if constexpr (KI == 2) {
      if (a.mode != 3) { hipLaunchKernelGGL((gemv1_kernel<PR, KI, true, false, 0, true>), grid, block, 0, s, a); return; }
    }
    hipLaunchKernelGGL((gemv1_kernel<PR, KI, true, false>), grid, block, 0, s, a);
  } else {
    hipLaunchKernelGGL((gemv1_kernel<PR, KI, false, false>), grid, block, 0, s, a);
  }
}
template <int PR>
void launch1(const GemvArgs& a, hipStream_t s) {
  if (a.K <= 1024) launch1k<PR, 2>(a, s);
  else if (a.K <= 2048) launch1k<PR, 4>(a, s);
  else if constexpr (PR < 4) {  // gemv_rows_per_wave keeps 4 rows per wave to K <= 2048 (registers)
    if (a.K <= 3072) launch1k<PR, 6>(a, s); else launch1k<PR, 12>(a, s);
  }
}
template <int NB, int PR, int PF>
void launchn(const GemvArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((gemvn_kernel<NB, PR, PF>), dim3(gemv_blocks(a)), dim3(256), (size_t)NB * a.K * sizeof(float), s, a);
}

}  // namespace

const char* launch_gemv(const GemvArgs& a0, int NB, hipStream_t s) {
  if (a0.K % 8 != 0) return "gemv: K must be a multiple of 8";
  if (a0.K > 6144) return "gemv: K > 6144 unsupported";
  if (a0.mode == 2 && a0.N % 32 != 0) return "gemv: GLU needs N % 32 == 0";
  if (a0.mode == 3 && (!a0.part_val || !a0.part_idx || gemv_blocks(a0) > a0.part_stride))
    return "gemv: argmax partial buffer missing/too small";
  if (a0.attn_po && (a0.K % 128 != 0 || a0.K != a0.attn_heads * 128)) return "gemv: attention-partial input needs K = heads*128";
  if (a0.attn_po && a0.rms_w) return "gemv: attention-partial input cannot be combined with a fused RMSNorm";
  if (a0.attn_po && (NB < 4 ? NB : 4) * a0.attn_heads * a0.attn_nsplit > ATTN_F_MAX) return "gemv: too many attention splits for the LDS scale table";
  const int pr = gemv_rows_per_wave(a0);
  const int nb_cap = (int)((64 * 1024) / ((size_t)a0.K * 4));  // rows of x that fit in 64 KiB of LDS
  int done = 0;
  while (done < NB) {
    GemvArgs a = a0;
    if (a0.x) a.x = a0.x + (size_t)done * a0.ldx;
    if (a0.out) a.out = a0.out + (size_t)done * a0.ldo;
    if (a0.resid) a.resid = a0.resid + (size_t)done * a0.ldo;
    if (a0.part_val) { a.part_val = a0.part_val + (size_t)done * a0.part_stride; a.part_idx = a0.part_idx + (size_t)done * a0.part_stride; }
    if (a0.attn_po) {
      const size_t adv = (size_t)done * a0.attn_heads * a0.attn_nsplit;
      a.attn_pm = a0.attn_pm + adv; a.attn_pl = a0.attn_pl + adv; a.attn_po = a0.attn_po + adv * 128;
    }
    int nb = NB - done;
    if (nb >= 4 && nb_cap >= 4) nb = 4;
    else if (nb >= 2 && nb_cap >= 2) nb = 2;
    else nb = 1;
    if (nb == 1) {
      if (pr == 4) launch1<4>(a, s); else if (pr == 2) launch1<2>(a, s); else launch1<1>(a, s);
    } else if (nb == 2) {
      if (pr == 4) launchn<2, 4, 2>(a, s); else if (pr == 2) launchn<2, 2, 4>(a, s); else launchn<2, 1, 6>(a, s);
    } else {
      if (pr == 4) launchn<4, 4, 2>(a, s); else if (pr == 2) launchn<4, 2, 4>(a, s); else launchn<4, 1, 6>(a, s);
    }
    done += nb;
  }
  return nullptr;
}

}  // namespace q3a
