// Single-token (decode) attention for gfx950, flash-decoding style.
//
// Replaces, for the one new token of every sequence, TextAttention::forward's per-head RMSNorm on q/k
// (src/layers.rs:303-304), RoPE (layers.rs:307-308,361-375), the KV-cache append (layers.rs:311-319; the
// reference re-allocates the cache with `cat` every step), repeat_kv (layers.rs:321-324) and
// softmax(q K^T / sqrt(128)) V (layers.rs:327-335; the decode-step mask of text_decoder.rs:121-131 is
// all zeros).
//
// Grid = (kv head, sequence, key split of 128 keys).  One CU streams only ~25-50 GB/s, so the ~200 KB of
// K+V a kv head holds at context ~400 must be spread over several CUs: measured 12.8 us with one workgroup
// per kv head (8 or 16 waves alike) against 6.4 us with 128-key splits.  Every workgroup
//   * first issues its cache loads (they do not depend on the new token): each wave owns 16 keys and reads
//     them with fully coalesced 16-B-per-lane loads -- LPK lanes share one key (DPL head dims each), one
//     wave instruction covers KPI consecutive cache rows;
//   * meanwhile normalises/rotates q for its GROUP query heads (and, in the split that owns position
//     `pos`, the new k, and appends k/v to the cache);
//   * scores: per-lane partial dot over DPL dims + DPP reduction across the LPK lanes; P.V needs no
//     cross-lane traffic until one fold at the end;
//   * writes an unnormalised partial (m, l, o[128]) per (sequence, head, split).  The consumer (the
//     o_proj GEMV in k_gemv.hip, or attn_combine_kernel below) merges the splits -- no second launch on
//     the GEMV path.
#include <stdlib.h>

#include <type_traits>

#include "dev.h"
#include "kernels.h"

namespace q3a {
namespace {

template <typename KVT> struct Frag16;  // 16 bytes of a cached key/value row -> floats
template <> struct Frag16<uint16_t> {
  static constexpr int DPL = 8;
  static __device__ __forceinline__ void unpack(const uint4& r, float (&f)[8]) {
    f[0] = bf16lo(r.x); f[1] = bf16hi(r.x); f[2] = bf16lo(r.y); f[3] = bf16hi(r.y);
    f[4] = bf16lo(r.z); f[5] = bf16hi(r.z); f[6] = bf16lo(r.w); f[7] = bf16hi(r.w);
  }
};
template <> struct Frag16<float> {
  static constexpr int DPL = 4;
  static __device__ __forceinline__ void unpack(const uint4& r, float (&f)[4]) {
    f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
  }
};

#ifndef Q3A_DATTN_NT
#define Q3A_DATTN_NT 1  // non-temporal cache-row loads: neutral at batch 1, -2 % decode time at batch 32 (1.8 GB of KV per step)
#endif
#ifndef Q3A_DA_WAVES
#define Q3A_DA_WAVES 8
#endif
constexpr int DA_WAVES = Q3A_DA_WAVES;  // waves per workgroup; a split is always 128 keys

// (Round 2 built a fused qkv-projection + attention launch on top of this body -- qkv_attn_kernel, handed over inside each XCD --
// and round 5 a pair-split form of the batched kernel below; both were correct and slower, and were removed in round 6:
// docs/HISTORY.md, last present at commit caf7a05.)
template <int GROUP, typename KVT>
__device__ __forceinline__ void decode_attn_body(const DecodeAttnArgs& a, const int kvh, const int s, const int sp) {
  constexpr int DPL = Frag16<KVT>::DPL;   // head dims per lane: 8 (bf16) / 4 (f32)
  constexpr int LPK = 128 / DPL;          // lanes per key: 16 / 32
  constexpr int KPI = 64 / LPK;           // keys per load instruction: 4 / 2
  constexpr int SPLIT_KEYS = sizeof(KVT) == 2 ? DATTN_KEYS_PER_SPLIT_BF16 : DATTN_KEYS_PER_SPLIT_F32;
  constexpr int NI = (SPLIT_KEYS / DA_WAVES) / KPI;  // load instructions per wave: 4 / 8 at 16 keys per wave
  constexpr int KEYS_PER_WAVE = KPI * NI;     // 16 with 8 waves
  constexpr int KEYS_PER_SPLIT = KEYS_PER_WAVE * DA_WAVES;  // 128
  static_assert(KEYS_PER_SPLIT == (sizeof(KVT) == 2 ? DATTN_KEYS_PER_SPLIT_BF16 : DATTN_KEYS_PER_SPLIT_F32), "split size");
  __shared__ float q_s[GROUP][128];
  __shared__ __attribute__((aligned(16))) KVT k_s[128];  // the new token's k / v row in cache format
  __shared__ __attribute__((aligned(16))) KVT v_s[128];
  __shared__ float cm[DA_WAVES][GROUP], cl[DA_WAVES][GROUP];
  __shared__ float co[DA_WAVES][GROUP][128];
  // pos in front of the Q3A_ARG block: behind it (asm volatile) hipcc no longer proves the location clobber-free and turns
  // the scalar load into a vector load that it waits for at once
  const int pos = a.pos[s];
  // every argument in one scalar-load clause (dev.h Q3A_ARG) instead of three scalar round trips in series
  Q3A_ARG(a.qkv); Q3A_ARG(a.pos); Q3A_ARG(a.q_norm); Q3A_ARG(a.k_norm); Q3A_ARG(a.eps); Q3A_ARG(a.rope_cur); Q3A_ARG(a.kcache); Q3A_ARG(a.vcache);
  Q3A_ARG(a.pm); Q3A_ARG(a.pl); Q3A_ARG(a.po); Q3A_ARG(a.nsplit); Q3A_ARG(a.n_q); Q3A_ARG(a.n_kv); Q3A_ARG(a.max_ctx); Q3A_ARG(a.scale_div);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int stamp_wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  Q3A_STAMP_AT(a.stamp, stamp_wg, 0);  // entry
  const int sub = lane % LPK, kq = lane / LPK;
  const int key_lo = sp * KEYS_PER_SPLIT;
  const size_t pbase = ((size_t)s * a.n_q + (size_t)kvh * GROUP) * a.nsplit + sp;  // + g * nsplit
  const int qkv_dim = (a.n_q + 2 * a.n_kv) * 128;
  const float* row = a.qkv + (size_t)s * qkv_dim;
  KVT* kc = reinterpret_cast<KVT*>(a.kcache) + ((size_t)s * a.n_kv + kvh) * (size_t)a.max_ctx * 128;
  KVT* vc = reinterpret_cast<KVT*>(a.vcache) + ((size_t)s * a.n_kv + kvh) * (size_t)a.max_ctx * 128;

  // ---- one memory round trip: nothing below depends on a loaded value until everything has been requested.
  // 1. the new token's q/k/v rows, the norm weights and the RoPE row of this position (L2 hits; the RoPE row sits at a
  //    fixed address, written by the previous step's finalize) go first: loads return in order, so q is normalised,
  //    rotated and in LDS while the cache rows are still in flight
  float x1 = 0.f, x2 = 0.f, nw1 = 0.f, nw2 = 0.f, c = 0.f, sn = 0.f;
  auto load_rows = [&]() {
    if (wave < GROUP + 2) {  // waves 0..GROUP-1: the q heads, GROUP: k, GROUP+1: v
      const int r = wave < GROUP ? kvh * GROUP + wave : (wave == GROUP ? a.n_q + kvh : a.n_q + a.n_kv + kvh);
      x1 = row[r * 128 + lane];
      x2 = row[r * 128 + lane + 64];
    }
  };
  if (wave <= GROUP) {
    const float* nw = wave < GROUP ? a.q_norm : a.k_norm;
    nw1 = nw[lane];
    nw2 = nw[lane + 64];
    c = a.rope_cur[(size_t)s * 128 + lane];
    sn = a.rope_cur[(size_t)s * 128 + 64 + lane];
  }
  load_rows();
  // 2. the cache rows, unconditionally (rows at or beyond pos hold stale data and are masked below; the index is
  //    clamped to the allocation)
  const int key_base = key_lo + wave * KEYS_PER_WAVE + kq;
  uint4 kraw[NI], vraw[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int key = min(key_base + i * KPI, a.max_ctx - 1);
#if Q3A_DATTN_NT
    kraw[i] = ld_stream16(kc + (size_t)key * 128 + sub * DPL);
    vraw[i] = ld_stream16(vc + (size_t)key * 128 + sub * DPL);
#else
    kraw[i] = *reinterpret_cast<const uint4*>(kc + (size_t)key * 128 + sub * DPL);
    vraw[i] = *reinterpret_cast<const uint4*>(vc + (size_t)key * 128 + sub * DPL);
#endif
  }
  __builtin_amdgcn_sched_barrier(0);
  Q3A_STAMP_AT(a.stamp, stamp_wg, 1);  // every load requested
  if (key_lo > pos) {  // this split holds no key yet: statistics of an empty set, zeroed output
    if (tid < GROUP) { a.pm[pbase + (size_t)tid * a.nsplit] = -INFINITY; a.pl[pbase + (size_t)tid * a.nsplit] = 0.f; }
    if (tid < GROUP * 128) a.po[(pbase + (size_t)(tid >> 7) * a.nsplit) * 128 + (tid & 127)] = 0.f;
    return;
  }
  const bool owner = (pos - key_lo) < KEYS_PER_SPLIT;  // the new token lands in this split

  // ---- new token: q for the GROUP heads of this kv head; k/v only in the owning split ----
  if (wave <= GROUP) {  // per-head RMSNorm + RoPE (as dev.h head_norm_rope)
    const float ss = wave_sum_fast(x1 * x1 + x2 * x2);
    const float rstd = rstd_of(ss / 128.0f + a.eps, sizeof(KVT) == 2);  // bf16 cache = default mode: hardware rsq
    const float n1 = (x1 * rstd) * nw1, n2 = (x2 * rstd) * nw2;
    rope_rotate(n1, n2, c, sn, x1, x2);
  }
  if (wave < GROUP) {
    q_s[wave][lane] = x1;
    q_s[wave][lane + 64] = x2;
  } else if (owner && wave == GROUP) {
    KvIo<KVT>::store(kc + (size_t)pos * 128 + lane, x1);
    KvIo<KVT>::store(kc + (size_t)pos * 128 + lane + 64, x2);
    KvIo<KVT>::store(&k_s[lane], x1);
    KvIo<KVT>::store(&k_s[lane + 64], x2);
  } else if (owner && wave == GROUP + 1) {
    KvIo<KVT>::store(vc + (size_t)pos * 128 + lane, x1);
    KvIo<KVT>::store(vc + (size_t)pos * 128 + lane + 64, x2);
    KvIo<KVT>::store(&v_s[lane], x1);
    KvIo<KVT>::store(&v_s[lane + 64], x2);
  }
  __syncthreads();
  Q3A_STAMP_AT(a.stamp, stamp_wg, 2);  // q (and the new k / v) in LDS

  // ---- scores for this wave's 16 keys ----
  // This section is VALU work on the critical path (a wave64 VALU instruction occupies its SIMD for 4 cycles and two
  // waves share a SIMD): one reciprocal instead of a division per score, the hardware exponential in the default
  // mode (the precise mode keeps expf), permlane swaps instead of ds_bpermute for the cross-row steps.
  const float inv_scale = 1.0f / a.scale_div;
  auto expw = [](float x) { return sizeof(KVT) == 4 ? expf(x) : __expf(x); };
  // The multiply-adds run two lanes wide (v_pk_fma_f32 on float pairs).  The new token's own k / v row and the rows
  // beyond it are patched into the raw 16-B registers (4 selects) rather than into the unpacked floats (8).
  f32x2_t qf[GROUP][DPL / 2];
#pragma unroll
  for (int g = 0; g < GROUP; ++g)
#pragma unroll
    for (int e = 0; e < DPL / 2; ++e) qf[g][e] = f32x2_t{q_s[g][sub * DPL + 2 * e], q_s[g][sub * DPL + 2 * e + 1]};
  const uint4 k_new = *reinterpret_cast<const uint4*>(&k_s[sub * DPL]), v_new = *reinterpret_cast<const uint4*>(&v_s[sub * DPL]);
  float sc[NI][GROUP];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int key = key_base + i * KPI;
    if (key == pos) { kraw[i] = k_new; vraw[i] = v_new; }
    if (key > pos) vraw[i] = make_uint4(0u, 0u, 0u, 0u);  // stale cache row (possibly NaN bits): p is 0 there, keep 0 * v finite
    float kf[DPL];
    Frag16<KVT>::unpack(kraw[i], kf);
#pragma unroll
    for (int g = 0; g < GROUP; ++g) {
      f32x2_t p2 = qf[g][0] * f32x2_t{kf[0], kf[1]};
#pragma unroll
      for (int e = 1; e < DPL / 2; ++e) p2 += qf[g][e] * f32x2_t{kf[2 * e], kf[2 * e + 1]};
      float p = p2.x + p2.y;
      p = row16_sum(p);
      if (LPK == 32) p = xor16_sum(p);
      sc[i][g] = (key <= pos) ? p * inv_scale : -INFINITY;  // layers.rs:327-328 scales after the matmul
    }
  }
  f32x2_t acc[GROUP][DPL / 2];
  float mw[GROUP], lw[GROUP];
#pragma unroll
  for (int g = 0; g < GROUP; ++g) {
    float mx = sc[0][g];
#pragma unroll
    for (int i = 1; i < NI; ++i) mx = fmaxf(mx, sc[i][g]);
    if (LPK == 16) mx = xor16_max(mx);
    mx = xor32_max(mx);
    mw[g] = mx;  // -inf when none of this wave's keys exists yet
    lw[g] = 0.f;
#pragma unroll
    for (int e = 0; e < DPL / 2; ++e) acc[g][e] = f32x2_t{0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int key = key_base + i * KPI;
    float vf[DPL];
    Frag16<KVT>::unpack(vraw[i], vf);
#pragma unroll
    for (int g = 0; g < GROUP; ++g) {
      const float p = (key <= pos) ? expw(sc[i][g] - mw[g]) : 0.f;
      lw[g] += p;
#pragma unroll
      for (int e = 0; e < DPL / 2; ++e) acc[g][e] += f32x2_t{p, p} * f32x2_t{vf[2 * e], vf[2 * e + 1]};
    }
  }
  Q3A_STAMP_AT(a.stamp, stamp_wg, 3);  // scores + P.V of wave 0's keys (its cache rows have landed)
  // fold the KPI key columns of the wave (lanes with equal `sub`)
#pragma unroll
  for (int g = 0; g < GROUP; ++g) {
    if (LPK == 16) lw[g] = xor16_sum(lw[g]);
    lw[g] = xor32_sum(lw[g]);
    float af[DPL];
#pragma unroll
    for (int e = 0; e < DPL; ++e) {
      af[e] = (e & 1) ? acc[g][e >> 1].y : acc[g][e >> 1].x;
      if (LPK == 16) af[e] = xor16_sum(af[e]);
      af[e] = xor32_sum(af[e]);
    }
    if (lane == 0) { cm[wave][g] = mw[g]; cl[wave][g] = lw[g]; }
    if (kq == 0) {
#pragma unroll
      for (int e = 0; e < DPL; ++e) co[wave][g][sub * DPL + e] = af[e];
    }
  }
  __syncthreads();
  Q3A_STAMP_AT(a.stamp, stamp_wg, 4);  // wave partials in LDS
  // ---- merge the 8 waves, write the split's partial ----
  if (wave < GROUP) {
    const int g = wave;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < DA_WAVES; ++w) M = fmaxf(M, cm[w][g]);  // finite: key_lo <= pos
    float L = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
    for (int w = 0; w < DA_WAVES; ++w) {
      const float f = (cm[w][g] == -INFINITY) ? 0.f : expw(cm[w][g] - M);
      L += cl[w][g] * f;
      o0 += co[w][g][lane] * f;
      o1 += co[w][g][lane + 64] * f;
    }
    const size_t pi = pbase + (size_t)g * a.nsplit;
    if (lane == 0) { a.pm[pi] = M; a.pl[pi] = L; }
    a.po[pi * 128 + lane] = o0;
    a.po[pi * 128 + lane + 64] = o1;
  }
  Q3A_STAMP_AT(a.stamp, stamp_wg, 5);
}

template <int GROUP, typename KVT>
__global__ __launch_bounds__(DA_WAVES * 64) void decode_attn_kernel(DecodeAttnArgs a) {
  decode_attn_body<GROUP, KVT>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}


// ---- batched form: the batch alone fills the chip (S * n_kv >= ~CUs), so one workgroup walks ALL keys of its
// (sequence, kv head) in 128-key tiles with an online softmax and writes the normalised context itself.  No partials
// in HBM, no merge launch (at 32 sequences: 15.4 us attention + 4.8 us merge per layer before).  The tile body is the one
// above; what changes:
//   * the next tile's cache rows are requested before the current tile is consumed (register double buffer);
//   * bf16 cache: q is rounded to bf16 pairs once (the rounding the MFMA prefill attention applies to q as well) and the
//     scores use v_dot2c_f32_bf16 on the RAW cache dwords: 4 instructions per 8 dims and head instead of 8 unpack
//     shifts + 4 packed FMAs -- this kernel was thought VALU-bound at batch 32 (round 5: a form with 39 % fewer issues per tile bought 2 %);
//   * each wave keeps running (max, sum, acc) over its 16 keys of every tile; the cross-lane / cross-wave folds happen
//     once at the end.
template <bool B> struct BoolC { static constexpr bool value = B; };  // compile-time flag passed to a generic lambda
#ifndef Q3A_DATTN_RING
#define Q3A_DATTN_RING 2  // tiles in flight per wave (A/B: 2 = one tile ahead 15.3 us per layer at 32 x 500 keys, 3 = 16.6 us)
#endif
template <int GROUP, typename KVT, int TILE = 128, int RING_T = Q3A_DATTN_RING>
__global__ __launch_bounds__(DA_WAVES * 64) void decode_attn_batched_kernel(DecodeAttnArgs a) {
  constexpr int DPL = Frag16<KVT>::DPL;
  constexpr int LPK = 128 / DPL;
  constexpr int KPI = 64 / LPK;
  constexpr int NI = (TILE / DA_WAVES) / KPI;  // TILE keys per round of the workgroup
  constexpr int KEYS_PER_WAVE = KPI * NI;
  constexpr bool DOT2 = sizeof(KVT) == 2;
  __shared__ float q_s[GROUP][128];
  __shared__ __attribute__((aligned(16))) KVT k_s[128];
  __shared__ __attribute__((aligned(16))) KVT v_s[128];
  __shared__ float cm[DA_WAVES][GROUP], cl[DA_WAVES][GROUP];
  __shared__ float co[DA_WAVES][GROUP][128];
  const int pos = a.pos[blockIdx.y];  // (a scalar load as long as it sits in front of the Q3A_ARG block: see decode_attn_body)
  // every argument in one scalar-load clause (dev.h Q3A_ARG): the prologue had three scalar round trips in series in front
  // of the first cache-row request
  Q3A_ARG(a.qkv); Q3A_ARG(a.pos); Q3A_ARG(a.q_norm); Q3A_ARG(a.k_norm); Q3A_ARG(a.eps); Q3A_ARG(a.rope_cur); Q3A_ARG(a.kcache); Q3A_ARG(a.vcache);
  Q3A_ARG(a.n_q); Q3A_ARG(a.n_kv); Q3A_ARG(a.max_ctx); Q3A_ARG(a.scale_div); Q3A_ARG(a.out); Q3A_ARG(a.out16); Q3A_ARG(a.out_frag); Q3A_ARG(a.trim_prologue);
  const int kvh = blockIdx.x, s = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int stamp_wg = blockIdx.y * gridDim.x + blockIdx.x;
  Q3A_STAMP_AT(a.stamp, stamp_wg, 0);  // entry
  const int sub = lane % LPK, kq = lane / LPK;
  const int qkv_dim = (a.n_q + 2 * a.n_kv) * 128;
  const float* row = a.qkv + (size_t)s * qkv_dim;
  KVT* kc = reinterpret_cast<KVT*>(a.kcache) + ((size_t)s * a.n_kv + kvh) * (size_t)a.max_ctx * 128;
  KVT* vc = reinterpret_cast<KVT*>(a.vcache) + ((size_t)s * a.n_kv + kvh) * (size_t)a.max_ctx * 128;

  const int key_w = wave * KEYS_PER_WAVE + kq;  // this lane's first key inside a tile
  // Register ring of RING tiles: the rows of tiles t+1 .. t+RING-1 are in flight while tile t is consumed (one CU must
  // keep > 100 KB requested to stream its share of HBM bandwidth; with one tile ahead the loop ran at one memory round
  // trip per 128 keys: 15.3 us per layer at 32 sequences x 500 keys against ~10 us of HBM time)
  constexpr int RING = DOT2 ? RING_T : 1;  // (the fp32 cache of the precise mode has twice the registers per tile: no ring there)
  static_assert(RING >= 1 && RING <= 4 && NI >= 1, "ring depth / tile size");
  uint4 kr0[NI], vr0[NI], kr1[RING > 1 ? NI : 1], vr1[RING > 1 ? NI : 1], kr2[RING > 2 ? NI : 1], vr2[RING > 2 ? NI : 1], kr3[RING > 3 ? NI : 1], vr3[RING > 3 ? NI : 1];
  // The cache rows go FIRST (stamped timeline: 2.2 us passed between the entry and the last request of the first two tiles
  // with the new token's rows in front -- the memory pipe of a kernel that is bound by its 256 KB per CU sat idle that long).
  // max_ctx is a multiple of TILE (launcher), so only the TILE index is clamped (scalar): one base address per tile, the
  // NI rows of a lane are KPI * 256 B apart (immediate offsets).  Rows beyond pos hold stale data and are masked below.
  const int last_tile = a.max_ctx / TILE - 1;
  const size_t lane_off = (size_t)key_w * 128 + sub * DPL;
  // trim (round 5): a tile asks for row min(key, pos) instead of rows past the sequence's last key: those lanes then all hit the one line of row `pos` instead of streaming stale cache rows from
  // HBM (they are masked either way).  The last tile of a context is half empty on average: 11 % of the bytes at 405-505 keys
  // (PMC: 68.2 MB fetched per launch for 59.6 MB of live rows).
  auto load_tile = [&](int j, uint4 (&kr)[NI], uint4 (&vr)[NI], const bool trim = false, const int pos_row = 0) {
    const int t = j;
    if (!trim) {
      const size_t base = (size_t)min(t, last_tile) * (TILE * 128) + lane_off;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        kr[i] = ld_stream16(kc + base + (size_t)i * (KPI * 128));
        vr[i] = ld_stream16(vc + base + (size_t)i * (KPI * 128));
      }
    } else {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int key = min(t * TILE + key_w + i * KPI, pos_row);  // (pos < max_ctx: inside the allocation)
        const size_t off = (size_t)key * 128 + sub * DPL;
        kr[i] = ld_stream16(kc + off);
        vr[i] = ld_stream16(vc + off);
      }
    }
  };
#ifndef Q3A_DATTN_TRIM
#define Q3A_DATTN_TRIM 2  // (A/B builds: 0 = every tile requested whole, 1 = only the tiles requested from inside the loop are trimmed)
#endif
  constexpr bool TRIM = Q3A_DATTN_TRIM != 0;
  // The prologue's later tiles are trimmed only when the host says some sequence of the batch is shorter than the prologue
  // (a.trim_prologue: shortest prompt < 2 tiles).  Their addresses then need `pos`, a second, dependent scalar load: measured
  // +4 us per step at 32 x 30 s clips, where no sequence has a stale row there (1133-1136 vs 1129-1132 us), -4 % per step at 32 x
  // 7 s clips, where tile 1 is stale for everyone (962 vs 1003 us) -- profiles/r5_ab_dattn_trim.txt.
  const bool TRIM_PRO = Q3A_DATTN_TRIM >= 2 && a.trim_prologue != 0;
  // (tile 0 goes out whole, the moment the arguments are in: `pos` is a second, dependent scalar load -- a.pos is one of the arguments --
  // and waiting for it here would put its round trip in front of the first request.)
  load_tile(0, kr0, vr0);
  // the new token's q / k / v rows, norm weights and RoPE row go BETWEEN tile 0 and the rest of the ring: loads return in
  // order, so they are back right behind tile 0 -- when they are first needed -- and tile 0 is consumed (and the ring
  // refilled) while tile 1 is still in flight
  float x1 = 0.f, x2 = 0.f, nw1 = 0.f, nw2 = 0.f, c = 0.f, sn = 0.f;
  if (wave < GROUP + 2) {  // waves 0..GROUP-1: the q heads, GROUP: k, GROUP+1: v
    const int r = wave < GROUP ? kvh * GROUP + wave : (wave == GROUP ? a.n_q + kvh : a.n_q + a.n_kv + kvh);
    x1 = row[r * 128 + lane];
    x2 = row[r * 128 + lane + 64];
    if (wave <= GROUP) {
      const float* nw = wave < GROUP ? a.q_norm : a.k_norm;
      nw1 = nw[lane];
      nw2 = nw[lane + 64];
      c = a.rope_cur[(size_t)s * 128 + lane];
      sn = a.rope_cur[(size_t)s * 128 + 64 + lane];
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // (keeps the row requests in front of the following tiles)
  if constexpr (RING > 1) load_tile(1, kr1, vr1, TRIM_PRO, pos);
  if constexpr (RING > 2) load_tile(2, kr2, vr2, TRIM_PRO, pos);
  if constexpr (RING > 3) load_tile(3, kr3, vr3, TRIM_PRO, pos);
  __builtin_amdgcn_sched_barrier(0);
  Q3A_STAMP_AT(a.stamp, stamp_wg, 1);  // q/k/v row + first tiles requested
  const int n_tiles = pos / TILE + 1;  // tiles that hold at least one key <= pos

  if (wave <= GROUP) {  // per-head RMSNorm + RoPE (as dev.h head_norm_rope)
    const float ss = wave_sum_fast(x1 * x1 + x2 * x2);
    const float rstd = rstd_of(ss / 128.0f + a.eps, sizeof(KVT) == 2);  // bf16 cache = default mode: hardware rsq
    const float n1 = (x1 * rstd) * nw1, n2 = (x2 * rstd) * nw2;
    rope_rotate(n1, n2, c, sn, x1, x2);
  }
  if (wave < GROUP) {
    q_s[wave][lane] = x1;
    q_s[wave][lane + 64] = x2;
  } else if (wave == GROUP) {
    KvIo<KVT>::store(kc + (size_t)pos * 128 + lane, x1);
    KvIo<KVT>::store(kc + (size_t)pos * 128 + lane + 64, x2);
    KvIo<KVT>::store(&k_s[lane], x1);
    KvIo<KVT>::store(&k_s[lane + 64], x2);
  } else if (wave == GROUP + 1) {
    KvIo<KVT>::store(vc + (size_t)pos * 128 + lane, x1);
    KvIo<KVT>::store(vc + (size_t)pos * 128 + lane + 64, x2);
    KvIo<KVT>::store(&v_s[lane], x1);
    KvIo<KVT>::store(&v_s[lane + 64], x2);
  }
  __syncthreads();
  Q3A_STAMP_AT(a.stamp, stamp_wg, 2);  // q / new k / new v in LDS

  const float inv_scale = 1.0f / a.scale_div;
  auto expw = [](float x) { return sizeof(KVT) == 4 ? expf(x) : __expf(x); };
  f32x2_t qf[GROUP][DPL / 2];
  uint32_t qp[GROUP][DPL / 2];  // DOT2: q as bf16 pairs
#pragma unroll
  for (int g = 0; g < GROUP; ++g)
#pragma unroll
    for (int e = 0; e < DPL / 2; ++e) {
      qf[g][e] = f32x2_t{q_s[g][sub * DPL + 2 * e], q_s[g][sub * DPL + 2 * e + 1]};
      qp[g][e] = pack_bf16x2(qf[g][e].x, qf[g][e].y);
    }
  const uint4 k_new = *reinterpret_cast<const uint4*>(&k_s[sub * DPL]), v_new = *reinterpret_cast<const uint4*>(&v_s[sub * DPL]);
  f32x2_t acc[GROUP][DPL / 2];
  float mrun[GROUP], lrun[GROUP];
#pragma unroll
  for (int g = 0; g < GROUP; ++g) {
    mrun[g] = -INFINITY;
    lrun[g] = 0.f;
#pragma unroll
    for (int e = 0; e < DPL / 2; ++e) acc[g][e] = f32x2_t{0.f, 0.f};
  }

  // A tile is consumed in one of two forms (round 5, Q3A_DATTN_EDGE): every tile in front of the one that holds position `pos`
  // has 128 live keys -- no new-token row to patch in, no stale row to blank, no score to mask -- and takes the INNER form,
  // which has none of that (the patched form spent 64 of its ~430 VALU issues per tile on register copies and exec-masked
  // moves alone: the in-place patch of the ring registers made hipcc copy the whole tile twice).  The choice is scalar (`pos`
  // and the tile index are SGPRs), the arithmetic of an inner tile is the same instruction for instruction: bit-identical.
  auto tile_body = [&](auto edge_c, int j, const uint4 (&kraw)[NI], const uint4 (&vraw)[NI]) {
    constexpr bool EDGE = decltype(edge_c)::value;
    const int key_base = j * TILE + key_w;
    float sc[NI][GROUP];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int key = key_base + i * KPI;
#pragma unroll
      for (int g = 0; g < GROUP; ++g) {
        float p;
        if constexpr (DOT2) {
          const uint32_t kw[4] = {kraw[i].x, kraw[i].y, kraw[i].z, kraw[i].w};
          p = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            p = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, kw[e]), __builtin_bit_cast(bf16x2_t, qp[g][e]), p, false);
        } else {
          float kf[DPL];
          Frag16<KVT>::unpack(kraw[i], kf);
          f32x2_t p2 = qf[g][0] * f32x2_t{kf[0], kf[1]};
#pragma unroll
          for (int e = 1; e < DPL / 2; ++e) p2 += qf[g][e] * f32x2_t{kf[2 * e], kf[2 * e + 1]};
          p = p2.x + p2.y;
        }
        p = row16_sum(p);
        if (LPK == 32) p = xor16_sum(p);
        sc[i][g] = (!EDGE || key <= pos) ? p * inv_scale : -INFINITY;  // layers.rs:327-328 scales after the matmul
      }
    }
#pragma unroll
    for (int g = 0; g < GROUP; ++g) {
      float mx = sc[0][g];
#pragma unroll
      for (int i = 1; i < NI; ++i) mx = fmaxf(mx, sc[i][g]);
      if (LPK == 16) mx = xor16_max(mx);
      mx = xor32_max(mx);  // over the wave's 16 keys of this tile
      const float m_new = fmaxf(mrun[g], mx);
      if (m_new == -INFINITY) continue;  // wave-uniform: none of this wave's keys exists yet
      const float alpha = expw(mrun[g] - m_new);  // exp(-inf) = 0 on the first live tile
      mrun[g] = m_new;
      lrun[g] *= alpha;
#pragma unroll
      for (int e = 0; e < DPL / 2; ++e) acc[g][e] *= f32x2_t{alpha, alpha};
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int key = key_base + i * KPI;
      float vf[DPL];
      Frag16<KVT>::unpack(vraw[i], vf);
#pragma unroll
      for (int g = 0; g < GROUP; ++g) {
        const float p = (!EDGE || key <= pos) ? expw(sc[i][g] - mrun[g]) : 0.f;
        lrun[g] += p;
#pragma unroll
        for (int e = 0; e < DPL / 2; ++e)  // explicit FMA: with two forms of the tile hipcc left one of the 32 contractions as multiply + add across a branch
          acc[g][e] = __builtin_elementwise_fma(f32x2_t{p, p}, f32x2_t{vf[2 * e], vf[2 * e + 1]}, acc[g][e]);
      }
    }
  };
#ifndef Q3A_DATTN_EDGE
#define Q3A_DATTN_EDGE 1  // (A/B builds: 0 = every tile in the patched form, as before)
#endif
  const int edge_tile = pos / TILE;  // the tile that holds position `pos`: the last one with a live key
  auto consume = [&](int j, const uint4 (&kraw)[NI], const uint4 (&vraw)[NI]) {
    if (Q3A_DATTN_EDGE && j < edge_tile) {
      tile_body(BoolC<false>{}, j, kraw, vraw);
      return;
    }
    const int key_base = j * TILE + key_w;
    uint4 kp[NI], vp[NI];  // the tile with the new token's rows patched in and the rows past it blanked
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int key = key_base + i * KPI;
      kp[i] = (key == pos) ? k_new : kraw[i];
      vp[i] = (key == pos) ? v_new : vraw[i];
      if (key > pos) vp[i] = make_uint4(0u, 0u, 0u, 0u);  // stale cache row (possibly NaN bits): keep 0 * v finite
    }
    tile_body(BoolC<true>{}, j, kp, vp);
  };
  for (int t = 0; t < n_tiles; t += RING) {
    consume(t, kr0, vr0);
    if (t + RING < n_tiles) load_tile(t + RING, kr0, vr0, TRIM, pos);
    if constexpr (RING > 1) {
      if (t + 1 < n_tiles) {
        consume(t + 1, kr1, vr1);
        if (t + 1 + RING < n_tiles) load_tile(t + 1 + RING, kr1, vr1, TRIM, pos);
      }
    }
    if constexpr (RING > 2) {
      if (t + 2 < n_tiles) {
        consume(t + 2, kr2, vr2);
        if (t + 2 + RING < n_tiles) load_tile(t + 2 + RING, kr2, vr2, TRIM, pos);
      }
    }
    if constexpr (RING > 3) {
      if (t + 3 < n_tiles) {
        consume(t + 3, kr3, vr3);
        if (t + 3 + RING < n_tiles) load_tile(t + 3 + RING, kr3, vr3, TRIM, pos);
      }
    }
  }
  Q3A_STAMP_AT(a.stamp, stamp_wg, 3);  // every tile consumed
  // fold the KPI key columns of the wave (lanes with equal `sub`), then the waves through LDS
#pragma unroll
  for (int g = 0; g < GROUP; ++g) {
    float lw = lrun[g];
    if (LPK == 16) lw = xor16_sum(lw);
    lw = xor32_sum(lw);
    float af[DPL];
#pragma unroll
    for (int e = 0; e < DPL; ++e) {
      af[e] = (e & 1) ? acc[g][e >> 1].y : acc[g][e >> 1].x;
      if (LPK == 16) af[e] = xor16_sum(af[e]);
      af[e] = xor32_sum(af[e]);
    }
    if (lane == 0) { cm[wave][g] = mrun[g]; cl[wave][g] = lw; }
    if (kq == 0) {
#pragma unroll
      for (int e = 0; e < DPL; ++e) co[wave][g][sub * DPL + e] = af[e];
    }
  }
  __syncthreads();
  Q3A_STAMP_AT(a.stamp, stamp_wg, 4);  // wave partials in LDS
  float M = -INFINITY, L = 0.f, o0 = 0.f, o1 = 0.f;  // (waves < GROUP) this workgroup's unnormalised result for head g = wave
  if (wave < GROUP) {
    const int g = wave;
#pragma unroll
    for (int w = 0; w < DA_WAVES; ++w) M = fmaxf(M, cm[w][g]);  // finite: key 0 <= pos always exists
#pragma unroll
    for (int w = 0; w < DA_WAVES; ++w) {
      const float f = (cm[w][g] == -INFINITY) ? 0.f : expw(cm[w][g] - M);
      L += cl[w][g] * f;
      o0 += co[w][g][lane] * f;
      o1 += co[w][g][lane + 64] * f;
    }
  }
  if (wave < GROUP) {
    const int g = wave;
    o0 /= L;
    o1 /= L;
    const int head = kvh * GROUP + g;
    if (a.out16) {
      const int k0 = head * 128 + lane;
      if (a.out_frag) {
        a.out16[skinny_frag_index(s, k0)] = (uint16_t)f32_to_bf16_bits(o0);
        a.out16[skinny_frag_index(s, k0 + 64)] = (uint16_t)f32_to_bf16_bits(o1);
      } else {
        a.out16[(size_t)s * a.n_q * 128 + k0] = (uint16_t)f32_to_bf16_bits(o0);
        a.out16[(size_t)s * a.n_q * 128 + k0 + 64] = (uint16_t)f32_to_bf16_bits(o1);
      }
    } else {
      a.out[(size_t)s * a.n_q * 128 + head * 128 + lane] = o0;
      a.out[(size_t)s * a.n_q * 128 + head * 128 + lane + 64] = o1;
    }
  }
  Q3A_STAMP_AT(a.stamp, stamp_wg, 5);
}

// merge of the split partials into [S][n_q*128] (only the GEMM decode path needs it as a separate launch)
__global__ __launch_bounds__(128) void attn_combine_kernel(const float* __restrict__ pm, const float* __restrict__ pl,
                                                           const float* __restrict__ po, int nsplit, float* __restrict__ out,
                                                           uint16_t* __restrict__ out16, int n_q, int frag) {
  const size_t sh = blockIdx.x;  // (sequence, head) flattened
  const int d = threadIdx.x;
  float M = -INFINITY;
  for (int sp = 0; sp < nsplit; ++sp) M = fmaxf(M, pm[sh * nsplit + sp]);
  float L = 0.f, o = 0.f;
  for (int sp = 0; sp < nsplit; ++sp) {
    const float m = pm[sh * nsplit + sp];
    if (m == -INFINITY) continue;
    const float f = expf(m - M);
    L += pl[sh * nsplit + sp] * f;
    o += po[(sh * nsplit + sp) * 128 + d] * f;
  }
  if (out16) {  // feeds the bf16-x skinny GEMM (o_proj), optionally in its fragment order
    const int s = (int)(sh / n_q), k = (int)(sh % n_q) * 128 + d;
    out16[frag ? skinny_frag_index(s, k) : sh * 128 + d] = (uint16_t)f32_to_bf16_bits(o / L);
  } else {
    out[sh * 128 + d] = o / L;
  }
}

}  // namespace

const char* launch_decode_attn(const DecodeAttnArgs& a, int S, bool kv_f32, hipStream_t s) {
  if (S <= 0) return nullptr;
  // the caller launches as many splits as the caches HOLD keys for (every pos < nsplit * keys per split), not as many as they
  // have room for: keys beyond the last split are never looked at
  if (a.nsplit <= 0 || (a.nsplit - 1) * dattn_keys_per_split(kv_f32) >= a.max_ctx) return "decode_attn: 1 .. ceil(max_ctx / keys per split) key splits";
  const int group = a.n_q / a.n_kv;
  // (Round 6 measured an L2 warm-up duty here -- the CUs this launch leaves idle pulling the o_proj / down matrices and the next
  // layer's K / V rows into the XCD-local L2s: -1.4 % per step on one box, +1.2 % on another, every partial form slower; removed,
  // docs/HISTORY.md 3.2, commit 3ac6499 has the code.)
  dim3 grid(a.n_kv, S, a.nsplit), block(DA_WAVES * 64);
#define Q3A_DA(G)                                                                                   \
  do {                                                                                              \
    if (kv_f32) hipLaunchKernelGGL((decode_attn_kernel<G, float>), grid, block, 0, s, a);          \
    else hipLaunchKernelGGL((decode_attn_kernel<G, uint16_t>), grid, block, 0, s, a);              \
  } while (0)
  if (group == 1) Q3A_DA(1);
  else if (group == 2) Q3A_DA(2);
  else if (group == 4) Q3A_DA(4);
  else return "decode_attn: GQA group must be 1, 2 or 4";
#undef Q3A_DA
  return nullptr;
}

// (S * n_kv -- the workgroups of the batched kernel -- from which the engine's batched decode step uses it is the knob
// dattn_batched_min_wgs of kernels.h; below it the key-split kernel + merge keep more CUs busy.)

const char* launch_decode_attn_batched(const DecodeAttnArgs& a, int S, bool kv_f32, hipStream_t s) {
  if (S <= 0) return nullptr;
  if (!a.out && !a.out16) return "decode_attn_batched: no output buffer";
  if (a.out16 && a.out_frag && S > 32) return "decode_attn_batched: fragment order holds at most 32 sequences";
  if (a.max_ctx % 128 != 0) return "decode_attn_batched: max_ctx must be a multiple of 128 (whole key tiles)";
  const int group = a.n_q / a.n_kv;
  dim3 grid(a.n_kv, S), block(DA_WAVES * 64);
  // (Round 5 measured two more shapes of the ring here -- 64-key tiles x 4 in flight, 128-key tiles x 4 in flight -- behind
  // environment switches; both lost at 16 and 32 sequences (profiles/r5_ab_dattn_ring_16seq.txt) and their instantiations are gone:
  // the template still takes TILE and RING_T.)
#define Q3A_DAB(G)                                                                                          \
  do {                                                                                                      \
    if (kv_f32) hipLaunchKernelGGL((decode_attn_batched_kernel<G, float>), grid, block, 0, s, a);          \
    else hipLaunchKernelGGL((decode_attn_batched_kernel<G, uint16_t>), grid, block, 0, s, a);              \
  } while (0)
  if (group == 1) Q3A_DAB(1);
  else if (group == 2) Q3A_DAB(2);
  else if (group == 4) Q3A_DAB(4);
  else return "decode_attn_batched: GQA group must be 1, 2 or 4";
#undef Q3A_DAB
  return nullptr;
}

const char* launch_attn_combine(const float* pm, const float* pl, const float* po, int nsplit, int S, int n_q, float* out,
                                hipStream_t s, uint16_t* out16, bool frag) {
  if (S <= 0) return nullptr;
  if (frag && S > 32) return "attn_combine: fragment order holds at most 32 sequences";
  hipLaunchKernelGGL(attn_combine_kernel, dim3(S * n_q), dim3(128), 0, s, pm, pl, po, nsplit, out, out16, n_q, frag ? 1 : 0);
  return nullptr;
}

}  // namespace q3a
