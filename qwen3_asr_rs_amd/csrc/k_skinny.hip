// Skinny MFMA GEMM for the batched decode step on gfx950: out[s][n] = sum_k x[s][k] * W[n][k] with
// 4 < S <= 32 sequences.  Replaces Linear::forward (src/layers.rs:74-80) for one token of each of S
// independent utterances -- with the RMSNorm in front of it (layers.rs:48-54) and the SiLU(gate)*up behind it
// (layers.rs:396-400) fused where they apply; every weight byte is still streamed from HBM exactly once per step,
// now amortised over up to 32 sequences.
//
// The step is latency-bound (a CU sustains only ~16 GB/s when it waits for one round trip per 8 KB), so the
// kernel is shaped to put a workgroup's whole weight slab in flight at once: one workgroup = 16 weight rows
// (32 in GLU mode: a 16-row gate tile + its 16-row up tile), 8 waves split K eight ways, and every wave issues
// all loads of four 32-wide k-steps before its first MFMA.  Per k-step
//   A = W fragment : lane (row l&15, k-chunk l>>4) loads its 16 B from the row-major bf16 matrix
//                    (4 lanes cover 64 contiguous bytes of a row)
//   B = x fragment : lane (sequence l&15 [+16], same k-chunk) holds 8 activations as bf16.  Three sources (XMODE):
//                    0  fp32 x, rounded to bf16 (default) or split into bf16 hi + lo (precise mode: two MFMAs);
//                    1  fp32 x with the RMSNorm fused: x * w_norm enters the MFMA, sum(x^2) is accumulated on the
//                       side and 1/rms scales the finished dot product (a scalar per sequence) -- no norm launch;
//                    2  bf16 x written by the producing kernel (attention merge, SwiGLU epilogue): one 16-B load,
//                       no conversion, half the L2 bytes (x is re-read by every workgroup: at N = 1024 it outweighs
//                       the workgroup's own weight slab 4:1 in fp32)
//                    3  the RMSNorm-fused form of 2: the kernel that produced the residual row (an o/down skinny GEMM,
//                       the embedding kernels) also left bf16(x * w_norm) in fragment order and partial sums of x^2;
//                       numerically the same as mode 1, with lane-linear loads
//   v_mfma_f32_16x16x32_bf16 -> D[row][sequence], sequences 0-15 and 16-31 share the A fragment.
// The eight K-slices are summed through LDS in a fixed order (deterministic), then 1/rms, bias / residual /
// SiLU(gate)*up are applied.
#include <stdlib.h>

#include "dev.h"
#include "kernels.h"

namespace q3a {
namespace {

constexpr int SK_WAVES = 8;
typedef const void __attribute__((address_space(1)))* sk_gptr_t;
typedef void __attribute__((address_space(3)))* sk_lptr_t;
#ifndef Q3A_SK_EXP
#define Q3A_SK_EXP 0  // timing experiments of tools/launch_floor.hip only
#endif

__device__ __forceinline__ bf16x8_t pack8(const float4& a, const float4& b) {
  uint4 p;
  p.x = pack_bf16x2(a.x, a.y); p.y = pack_bf16x2(a.z, a.w); p.z = pack_bf16x2(b.x, b.y); p.w = pack_bf16x2(b.z, b.w);
  return *reinterpret_cast<const bf16x8_t*>(&p);
}
__device__ __forceinline__ void split8(const float4& a, const float4& b, bf16x8_t& hi, bf16x8_t& lo) {
  uint4 h, l;
  h.x = pack_bf16x2(a.x, a.y); h.y = pack_bf16x2(a.z, a.w); h.z = pack_bf16x2(b.x, b.y); h.w = pack_bf16x2(b.z, b.w);
  l.x = pack_bf16x2(a.x - bf16lo(h.x), a.y - bf16hi(h.x)); l.y = pack_bf16x2(a.z - bf16lo(h.y), a.w - bf16hi(h.y));
  l.z = pack_bf16x2(b.x - bf16lo(h.z), b.y - bf16hi(h.z)); l.w = pack_bf16x2(b.z - bf16lo(h.w), b.w - bf16hi(h.w));
  hi = *reinterpret_cast<const bf16x8_t*>(&h);
  lo = *reinterpret_cast<const bf16x8_t*>(&l);
}
__device__ __forceinline__ float4 mul4(const float4& a, const float4& b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float sq4(const float4& a) { return a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w; }

// TILES = 16-row weight tiles per workgroup (1, or 2 = gate + up); SH = 16-sequence halves (1: S <= 16, 2: S <= 32)
// UNR = k-steps whose loads a wave issues before its first MFMA: the launcher picks it to cover the wave's whole K
// slice (K/8: 4 steps at K = 1024, 12 at K = 3072), so a wave makes ONE memory round trip, not K/8/4 of them
// WLDS: the wave's weight slab goes HBM -> LDS by DMA (global_load_lds, non-temporal) in 8-row x 128-B pieces -- 8
// cache lines per wave instruction instead of the 16 of a direct fragment load, which the L1 retires twice as fast
// -- into a wave-private region (no barrier: the wave waits for its own DMA), swizzled like k_gemm16.hip so that the
// fragment reads (ds_read_b128, 16 rows x 16 B) are conflict-free.  The wave slice is consumed in passes of UNR steps (one pass when UNR == K/32/8).
// QS ("quarter" workgroups, hidden-wide outputs only): a workgroup owns 8 weight rows x ONE 16-sequence half.  At N = 1024
// the 16-row x 32-sequence shape gives 64 workgroups, each pulling its 64-96 KB weight slab AND the whole 128-192 KB
// activation matrix through one CU's L1 (the measured bound of these kernels) while 192 CUs idle; 256 quarter workgroups
// move half the bytes each.  The two halves of a row tile sit 8 apart in dispatch order (same XCD): the second weight
// read is an L2 hit, so the weight stream uses the default cache policy here instead of nt.
// PALIAS (round 5): the wave's partial tile of the cross-wave reduction lives in the wave's OWN weight region of the dynamic LDS (the
// wave has consumed its weights when it stores its partials; LDS operations of a wave are in order) instead of a static array.
// That takes 17-35 KB off the workgroup's LDS footprint: a 2-tile (gate + up) workgroup whose K slice is staged in two passes then
// needs 64.5 KiB, so TWO of them fit on a CU.  The gate/up projection at hidden 2048 has 384 workgroups: with one resident per CU
// it ran as a full round and a half-empty one (14.8 us for 50 MB at 16 sequences, profiles/r5_kernel_trace_1p7b_b16.txt).
// HP ("half pair" gate/up tiles, round 5): a 16-row MFMA fragment holds 8 GATE rows and the 8 UP rows of the same output columns
// (the packed matrix interleaves gate / up in 16-row blocks, so each half is 8 contiguous memory rows = one 8-row DMA piece), a
// workgroup owns TILES such fragments = 8 * TILES output columns, and SiLU(gate) * up pairs fragment rows i and i + 8.  That makes
// the gate/up projection divisible in units of 8 columns instead of 16: at hidden 2048 (inter 6144: 384 pair tiles for 256 CUs --
// a CU with two of them streams 2 x (128 KB of weights + the activations)) 256 workgroups x 3 half pairs give every CU 192 KB of
// weights and ONE pass over the activations.  Same K slices, same reduction order per output element as the pair form.
template <bool SPLIT, int TILES, int SH, int XMODE, int UNR, bool WLDS, bool QS = false, bool PALIAS = false, bool HP = false>
__global__ __launch_bounds__(SK_WAVES * 64) void skinny_kernel(SkinnyArgs a) {
  static_assert(!(SPLIT && XMODE >= 2), "the precise mode keeps fp32 activations");
  static_assert(!HP || (WLDS && PALIAS && !QS && !SPLIT), "half-pair tiles: LDS-staged weights, aliased partial tile");
  static_assert(!WLDS || UNR % 2 == 0, "the LDS image is made of 64-wide k columns");
  static_assert(!QS || (TILES == 1 && SH == 1 && XMODE == 2 && WLDS), "quarter workgroups: bf16 fragment-order x, LDS-staged weights");
  constexpr int PIECES = QS ? 1 : 2;  // 8-row x 128-B DMA pieces per (tile, 64-wide k column)
  constexpr int WREGION = TILES * (UNR / 2) * PIECES * 1024;  // bytes of a wave's weight region
  constexpr int PART_W = TILES * SH * 16 * 17;                // floats of a wave's partial tile
  static_assert(!PALIAS || (WLDS && PART_W * 4 <= WREGION), "the partial tile must fit the wave's weight region");
  extern __shared__ __attribute__((aligned(16))) unsigned char wlds[];  // WLDS: [wave][tile][UNR/2 columns][16 rows][128 B]
  __shared__ float part_static[PALIAS ? 1 : SK_WAVES * PART_W];  // [k-slice][tile][seq half][row][sequence] (+1 pad)
  auto part = [&](int w, int t, int h, int row, int col) -> float& {
    float* base = PALIAS ? reinterpret_cast<float*>(wlds + (size_t)w * WREGION) : part_static + w * PART_W;
    return base[((t * SH + h) * 16 + row) * 17 + col];
  };
  __shared__ float ssp[SK_WAVES][SH][16];              // XMODE 1: sum(x^2) of each sequence over the wave's K slice
  // every argument this instantiation touches, in one scalar-load clause (dev.h Q3A_ARG)
  Q3A_ARG(a.W); Q3A_ARG(a.N); Q3A_ARG(a.K); Q3A_ARG(a.S); Q3A_ARG(a.ldx); Q3A_ARG(a.bias); Q3A_ARG(a.mode); Q3A_ARG(a.out); Q3A_ARG(a.ldo);
  Q3A_ARG(a.resid); Q3A_ARG(a.eps); Q3A_ARG(a.fast_math);
  if (XMODE <= 1) { Q3A_ARG(a.x); Q3A_ARG(a.rms_w); }
  if (XMODE == 2) { Q3A_ARG(a.x16); Q3A_ARG(a.x16_frag); }
  if (XMODE == 3) { Q3A_ARG(a.xw16f); Q3A_ARG(a.ss_parts); Q3A_ARG(a.ss_nparts); }
  if (TILES == 2 || HP) { Q3A_ARG(a.out16); Q3A_ARG(a.out16_frag); }
  if (TILES == 1 && !HP) { Q3A_ARG(a.next_w); Q3A_ARG(a.next_xw16f); Q3A_ARG(a.next_ss); }
  if (QS) Q3A_ARG(a.qs_halves);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  Q3A_STAMP_AT(a.stamp, blockIdx.x, 0);  // entry
  const int l15 = lane & 15, kc = lane >> 4;  // row / sequence inside the fragment, k-chunk (8 elements)
  int n0 = blockIdx.x * 16 * TILES, hsel = 0, part_row = blockIdx.x;
  if (QS) {  // block b: XCD b & 7; consecutive same-XCD blocks alternate the sequence half
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    hsel = j % a.qs_halves;
    part_row = (j / a.qs_halves) * 8 + xcd;
    n0 = part_row * 8;
  }
  const int lrow = QS ? (l15 & 7) : l15;  // fragment row -> weight row of the tile (QS: rows 8..15 duplicate 0..7, discarded)
  // memory row of fragment row r16 of tile t.  Pair / plain form: consecutive rows.  HP: output columns c0 .. c0 + 7 of the tile;
  // their gate rows start at (c0 / 16) * 32 + c0 % 16 in the interleaved matrix, the up rows 16 further on.
  auto tile_row = [&](int t, int r16) -> int {
    if (!HP) return n0 + t * 16 + r16;
    const int c0 = (blockIdx.x * TILES + t) * 8;
    return ((c0 >> 4) << 5) + (c0 & 15) + (r16 >> 3) * 16 + (r16 & 7);
  };
  const int K = a.K;
  const int steps = K / 32, per = (steps + SK_WAVES - 1) / SK_WAVES;
  const int ks0 = wave * per, ks1 = min(steps, ks0 + per);
  // Measured (tools/launch_floor.hip, S = 32): these fragment loads touch 16 cache lines per wave instruction (16 rows
  // x 64 B) and the L1 retires them at ~16 B/clk per CU -- at N = 1024 (64 workgroups) that, not HBM, is the bound:
  // down-proj 12.2 us = 6.3 (W) + 7.2 (x) - overlap.  Pairing k-steps so that 4 lanes cover a whole 128-B line in two
  // back-to-back loads did not help (+0.3..0.9 us); a fragment-ordered activation layout is the next step.
  constexpr int kcw = 8;
  // out-of-range rows / sequences are clamped to valid memory: their products are discarded by the epilogue
  const uint16_t* wrow[TILES];
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
    const int row = tile_row(t, lrow);
    wrow[t] = a.W + (size_t)(row < a.N ? row : a.N - 1) * K + kc * kcw;
  }
  const float* xrow[SH];
  const uint16_t* xrow16[SH];
#pragma unroll
  for (int h = 0; h < SH; ++h) {
    const int s = (QS ? hsel : h) * 16 + l15;
    const size_t off = (size_t)(s < a.S ? s : a.S - 1) * a.ldx + kc * kcw;
    xrow[h] = a.x + off;
    xrow16[h] = a.x16 + off;
  }
  const float* nrow = a.rms_w + kc * kcw;
  f32x4_t acc[TILES][SH];
  float ss[SH];
#pragma unroll
  for (int h = 0; h < SH; ++h) ss[h] = 0.f;
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int h = 0; h < SH; ++h) acc[t][h] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // epilogue duty of this thread (row i, sequence s) and the operands it will add there: requested now, not after the
  // reduction barrier (an L2 round trip on the tail of a 5 us kernel)
  const int ep_i = QS ? (tid & 7) : (tid & 15), ep_s = QS ? hsel * 16 + ((tid >> 3) & 15) : (tid >> 4);
  const bool ep_live = QS ? (tid < 128 && ep_s < a.S) : (ep_s < a.S && (SH == 2 || ep_s < 16));
  float ep_resid = 0.f, ep_bias = 0.f, ep_nw = 0.f;
  if (TILES == 1 && !HP) {  // clamped addresses, no lane-divergent branch around the loads (the epilogue only uses them where live)
    const int en = min(n0 + ep_i, a.N - 1), es = min(ep_s, a.S - 1);
    if (a.mode == 1) ep_resid = a.resid[(size_t)es * a.ldo + en];
    if (a.bias) ep_bias = a.bias[en];
    if (a.mode == 1 && a.next_w) ep_nw = a.next_w[en];
  }
  // XMODE 3: this lane's share of the producer's sum(x^2) partial rows (p = wave*4 + kc, then every 32nd): the first SS_PRE
  // of them are requested here, in front of the weight stream, so their L2 round trip is not on the tail of the kernel
  constexpr int SS_PRE = 8;  // covers 256 partial rows (hidden 2048 in 8-column blocks); anything beyond is summed after the K loop
  float ssq[SH][SS_PRE];
  if (XMODE == 3) {
#pragma unroll
    for (int h = 0; h < SH; ++h)
#pragma unroll
      for (int j = 0; j < SS_PRE; ++j) {
        // unconditional (clamped) load, selected afterwards: with `p < n ? load : 0` hipcc puts every load into its own
        // exec-masked block and waits vmcnt(0) between them -- two L2 round trips in series in front of the weight stream
        const int p = wave * 4 + kc + j * SK_WAVES * 4, pc = min(p, a.ss_nparts - 1);
        const float v = a.ss_parts[(size_t)pc * 32 + (QS ? hsel : h) * 16 + l15];
        ssq[h][j] = p < a.ss_nparts ? v : 0.f;
      }
  }
  unsigned char* const wbase = wlds + (size_t)wave * WREGION;
  for (int kb = ks0; kb < ks1; kb += UNR) {
    uint4 wv[WLDS ? 1 : UNR][TILES];
    float4 x0[UNR][SH], x1[UNR][SH], w0[UNR], w1[UNR];
    uint4 xq[UNR][SH];
    if (WLDS) {  // the launcher guarantees (ks1 - ks0) % UNR == 0: whole passes.  A later pass refills the wave's private region
                 // after this wave's own fragment reads of the previous pass were consumed by its MFMAs (program order)
      const int rr = lane >> 3, p = lane & 7;
#pragma unroll
      for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int col = 0; col < UNR / 2; ++col)
#pragma unroll
          for (int g = 0; g < PIECES; ++g) {
            const int r16 = g * 8 + rr, row = tile_row(t, r16);
            const uint16_t* src = a.W + (size_t)(row < a.N ? row : a.N - 1) * K + (size_t)kb * 32 + col * 64 + (p ^ ((r16 >> 1) & 7)) * 8;
            __builtin_amdgcn_global_load_lds((sk_gptr_t)src, (sk_lptr_t)(wbase + ((t * (UNR / 2) + col) * PIECES + g) * 1024), 16, 0, QS ? 0 : 2);
          }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const bool live = kb + u < ks1;
      const int ks = live ? kb + u : ks1 - 1;  // clamp the address, zero the weight of the tail
      const int ko = ks * 32;  // element offset of this step inside the row
#pragma unroll
      for (int t = 0; t < TILES; ++t) {
        if (WLDS) continue;
#if (Q3A_SK_EXP & 2)
        wv[u][t] = make_uint4(ks, lane, 0u, 0u);
#else
        wv[u][t] = ld_stream16(wrow[t] + ko);
#endif
        if (!live) wv[u][t] = make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int h = 0; h < SH; ++h) {
        if (XMODE == 3) {  // pre-normalised x * w_norm, always in fragment order
          xq[u][h] = *reinterpret_cast<const uint4*>(a.xw16f + ((((size_t)ks * 2 + (QS ? hsel : h)) * 4 + kc) * 16 + l15) * 8);
        } else if (XMODE == 2) {
#if (Q3A_SK_EXP & 1)
          xq[u][h] = make_uint4(ks, lane, 0u, 0u);
#else
          // fragment order (kernels.h skinny_frag_index): lane-linear, 1 KiB contiguous per (k-step, sequence half)
          const uint16_t* xp = a.x16_frag ? a.x16 + ((((size_t)ks * 2 + (QS ? hsel : h)) * 4 + kc) * 16 + l15) * 8 : xrow16[h] + ko;
          xq[u][h] = *reinterpret_cast<const uint4*>(xp);
#endif
        } else {
          x0[u][h] = *reinterpret_cast<const float4*>(xrow[h] + ko);
          x1[u][h] = *reinterpret_cast<const float4*>(xrow[h] + ko + 4);
        }
      }
      if (XMODE == 1) {
        w0[u] = *reinterpret_cast<const float4*>(nrow + ko);
        w1[u] = *reinterpret_cast<const float4*>(nrow + ko + 4);
      }
    }
    Q3A_STAMP_AT(a.stamp, blockIdx.x, 1);  // every load of the pass requested
    if (WLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA is not in the compiler's load bookkeeping
    Q3A_STAMP_AT(a.stamp, blockIdx.x, 2);  // (WLDS) weights landed
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const bool live = kb + u < ks1;
#pragma unroll
      for (int h = 0; h < SH; ++h) {
        bf16x8_t hi, lo;
        if (XMODE >= 2) {
          hi = *reinterpret_cast<const bf16x8_t*>(&xq[u][h]);
        } else {
          float4 p0 = x0[u][h], p1 = x1[u][h];
          if (XMODE == 1) {
            if (live) ss[h] += sq4(p0) + sq4(p1);
            p0 = mul4(p0, w0[u]);
            p1 = mul4(p1, w1[u]);
          }
          if (SPLIT) split8(p0, p1, hi, lo);
          else hi = pack8(p0, p1);
        }
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
          const bf16x8_t wf = WLDS
              ? *reinterpret_cast<const bf16x8_t*>(wbase + (t * (UNR / 2) + (u >> 1)) * (PIECES * 1024) + lrow * 128 +
                                                   ((((u & 1) * 4 + kc) ^ ((lrow >> 1) & 7)) * 16))
              : *reinterpret_cast<const bf16x8_t*>(&wv[WLDS ? 0 : u][t]);
          acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, hi, acc[t][h], 0, 0, 0);
          if (SPLIT) acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, lo, acc[t][h], 0, 0, 0);
        }
      }
    }
  }
#if (Q3A_SK_EXP & 4)
  if (wave == 0 && n0 + kc * 4 < a.N) {  // experiment: no cross-wave reduction, wave 0 stores its partial
    for (int h = 0; h < SH; ++h)
      for (int r = 0; r < 4; ++r) a.out[(size_t)(h * 16 + l15) * a.ldo + n0 + kc * 4 + r] = acc[0][h][r];
  }
  return;
#endif
  Q3A_STAMP_AT(a.stamp, blockIdx.x, 3);  // MFMAs issued (their operands have arrived)
  // D[row i][sequence j] of v_mfma_f32_16x16x32: j = lane&15, i = (lane>>4)*4 + r
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int h = 0; h < SH; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) part(wave, t, h, kc * 4 + r, l15) = acc[t][h][r];
  if (XMODE == 3) {  // sum(x^2) comes from the producer's partials (requested before the K loop, see there)
#pragma unroll
    for (int h = 0; h < SH; ++h) {
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < SS_PRE; ++j) q += ssq[h][j];
      for (int p = wave * 4 + kc + SS_PRE * SK_WAVES * 4; p < a.ss_nparts; p += SK_WAVES * 4) q += a.ss_parts[(size_t)p * 32 + (QS ? hsel : h) * 16 + l15];
      ss[h] = q;
    }
  }
  if (XMODE == 1 || XMODE == 3) {
#pragma unroll
    for (int h = 0; h < SH; ++h) {
      float v = ss[h];
      v = xor32_sum(xor16_sum(v));  // the four k-chunk lanes of this sequence
      if (kc == 0) ssp[wave][h][l15] = v;
    }
  }
  __syncthreads();
  Q3A_STAMP_AT(a.stamp, blockIdx.x, 4);  // partial tiles of all waves in LDS
  // ---- fixed-order reduction of the K-slices + epilogue: thread -> (row i, sequence s) ----
  // 16 rows x 32 sequences = 512 threads; 16 consecutive lanes own the 16 consecutive output columns of one sequence
  // (64-B runs; with the sequence as the fast index every lane hit its own line: 4 KB stride)
  // (QS: 8 rows x 16 sequences = the first 128 threads, 8 consecutive lanes per sequence)
  const int i = ep_i, s = ep_s;
  const bool live_s = ep_live;  // (no early return: the row reduction below needs whole rows)
  const int sh = (QS || SH == 1 || !live_s) ? 0 : s >> 4, sj = s & 15;
  float v[TILES], vu[HP ? TILES : 1];  // (HP: v = gate row i, vu = up row i + 8 of every tile; threads with i >= 8 idle along)
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
    v[t] = 0.f;
#pragma unroll
    for (int w = 0; w < SK_WAVES; ++w) v[t] += part(w, t, sh, HP ? (i & 7) : i, sj);
    if (HP) {
      vu[t] = 0.f;
#pragma unroll
      for (int w = 0; w < SK_WAVES; ++w) vu[t] += part(w, t, sh, (i & 7) + 8, sj);
    }
  }
  if (XMODE == 1 || XMODE == 3) {
    float q = 0.f;
#pragma unroll
    for (int w = 0; w < SK_WAVES; ++w) q += ssp[w][sh][sj];
    const float rstd = rstd_of(q / (float)K + a.eps, a.fast_math != 0);
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      v[t] *= rstd;
      if (HP) vu[t] *= rstd;
    }
  }
  if (HP) {
    if (!live_s || i >= 8) return;
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      const int c = (blockIdx.x * TILES + t) * 8 + i;  // output column; its gate / up rows in the interleaved matrix:
      const int gr = ((c >> 4) << 5) + (c & 15), ur = gr + 16;
      if (ur >= a.N) continue;
      float g = v[t], u = vu[t];
      if (a.bias) { g += a.bias[gr]; u += a.bias[ur]; }
      const float y = silu_sel(g, a.fast_math != 0) * u;
      if (a.out16) a.out16[a.out16_frag ? skinny_frag_index(s, c) : (size_t)s * a.ldo + c] = (uint16_t)f32_to_bf16_bits(y);
      else a.out[(size_t)s * a.ldo + c] = y;
    }
    Q3A_STAMP_AT(a.stamp, blockIdx.x, 5);
  } else if (TILES == 1) {
    const int n = n0 + i;
    const bool ok = live_s && n < a.N;
    float y = 0.f;
    if (ok) {
      y = v[0];
      if (a.bias) y += ep_bias;
      if (a.mode == 1) y += ep_resid;
      a.out[(size_t)s * a.ldo + n] = y;
    }
    if (a.mode == 1 && a.next_w) {  // hand the new residual row to the next GEMM pre-normalised (kernels.h)
      if (ok) a.next_xw16f[skinny_frag_index(s, n)] = (uint16_t)f32_to_bf16_bits(y * ep_nw);
      const float q = QS ? row8_sum(y * y) : row16_sum(y * y);  // this block's 16 (8) columns of sequence s
      if (i == 0 && live_s) a.next_ss[(size_t)part_row * 32 + s] = q;
    }
    Q3A_STAMP_AT(a.stamp, blockIdx.x, 5);  // epilogue stores issued
  } else {  // rows n0..n0+15 = gate, n0+16..n0+31 = up of logical rows n0/2 .. n0/2+15
    if (!live_s || n0 + 16 + i >= a.N) return;
    float g = v[0], u = v[TILES - 1];
    if (a.bias) { g += a.bias[n0 + i]; u += a.bias[n0 + 16 + i]; }
    const float y = silu_sel(g, a.fast_math != 0) * u;
    if (a.out16) a.out16[a.out16_frag ? skinny_frag_index(s, (n0 >> 1) + i) : (size_t)s * a.ldo + (n0 >> 1) + i] = (uint16_t)f32_to_bf16_bits(y);
    else a.out[(size_t)s * a.ldo + (n0 >> 1) + i] = y;
    Q3A_STAMP_AT(a.stamp, blockIdx.x, 5);
  }
}

// dynamic LDS a kernel instance may use: the CU's 160 KiB minus its static arrays (part, ssp) and a 2 KiB margin
constexpr size_t sk_dyn_lds_max(int tiles, int sh) {
  return 160 * 1024 - sizeof(float) * (SK_WAVES * tiles * sh * 16 * 17 + SK_WAVES * sh * 16) - 2048;
}

template <bool SPLIT, int TILES, int SH, int XMODE, int UNR>
void launch_k(const SkinnyArgs& a, dim3 grid, hipStream_t s) {
  const dim3 block(SK_WAVES * 64);
  // weights through LDS when the wave's slice is exactly UNR steps and the image fits next to the reduction buffer
  const int steps = a.K / 32, per = (steps + SK_WAVES - 1) / SK_WAVES;
  const size_t wbytes = (size_t)SK_WAVES * TILES * (UNR / 2) * 2048;
  static const bool wlds_on = [] { const char* e = getenv("Q3A_SKINNY_WLDS"); return !e || atoi(e) != 0; }();  // A/B knob
  if constexpr (TILES == 2 && !SPLIT && XMODE == 3 && (UNR == 4 || UNR == 8)) {
    // more pair tiles than CUs and the output columns divide into 3 half pairs per workgroup with at most one workgroup per CU
    // (hidden 2048 / inter 6144: 256 workgroups): balanced half-pair form, the K slice in passes of 4 steps (96 KiB of LDS).
    // glu_hp3: 1 = when it balances (default), 0 = never (A/B), 2 = whenever the shape allows (tests at small shapes)
    const int n_cu_hp = a.n_cu > 0 ? a.n_cu : 256;  // the ENGINE's device (a function-local static cached whichever device the first caller had: ADVICE r5)
    const int inter = a.N / 2;
    const bool shape_ok = wlds_on && a.N % 32 == 0 && inter % 24 == 0 && per % 4 == 0 && steps % per == 0 && steps == per * SK_WAVES;
    const bool balances = (int)grid.x > n_cu_hp && inter / 24 <= n_cu_hp;
    if (shape_ok && (a.glu_hp3 == 2 || (a.glu_hp3 == 1 && balances))) {
      const size_t wb = (size_t)SK_WAVES * 3 * 2 * 2048;  // 8 waves x 3 tiles x 2 k columns (4 steps) x 16 rows x 128 B
      hipLaunchKernelGGL((skinny_kernel<false, 3, SH, 3, 4, true, false, true, true>), dim3(inter / 24), block, wb, s, a);
      return;
    }
  }
  if constexpr (TILES == 2 && UNR % 4 == 0 && UNR >= 8 && !SPLIT && XMODE >= 2) {
    // more workgroups than CUs (gate/up at hidden 2048: 384): two passes of UNR / 2 steps with the partial tile aliased into the
    // weight region -> 64.5 KiB per workgroup, two resident per CU, no half-empty second round (round 5: 1.428 vs 1.469 ms per step at
    // 1.7B x 16 against the single-pass form, whose knob is gone)
    const int n_cu = a.n_cu > 0 ? a.n_cu : 256;
    if (wlds_on && (int)grid.x > n_cu && per == UNR && steps % per == 0) {
      const size_t wb2 = (size_t)SK_WAVES * TILES * (UNR / 4) * 2048;
      hipLaunchKernelGGL((skinny_kernel<SPLIT, TILES, SH, XMODE, UNR / 2, true, false, true>), grid, block, wb2, s, a);
      return;
    }
  }
  if constexpr (UNR % 2 == 0 && !SPLIT) {
    // (K = 6144 at 1.7B: 24 steps per wave = two passes of 12; before, such shapes fell back to direct fragment loads:
    // down projection 14.8 us for 25 MB)
    if (wlds_on && per % UNR == 0 && steps % per == 0 && wbytes <= sk_dyn_lds_max(TILES, SH)) {
      // dynamic LDS above the 64 KiB default: the attribute is set by skinny_init() (not here: this may run under capture)
      hipLaunchKernelGGL((skinny_kernel<SPLIT, TILES, SH, XMODE, UNR, true>), grid, block, wbytes, s, a);
      return;
    }
  }
  hipLaunchKernelGGL((skinny_kernel<SPLIT, TILES, SH, XMODE, UNR, false>), grid, block, 0, s, a);
}
template <bool SPLIT, int XMODE, int UNR>
void launch_u(const SkinnyArgs& a, hipStream_t s) {
  if constexpr (!SPLIT && XMODE == 2 && UNR % 2 == 0) {
    if (a.qsplit) {  // validated by launch_skinny: mode 1, fragment-order x, N % 64 == 0, whole passes of UNR steps
      SkinnyArgs q = a;
      q.qs_halves = a.S > 16 ? 2 : 1;
      const dim3 grid((a.N / 8) * q.qs_halves), block(SK_WAVES * 64);
      hipLaunchKernelGGL((skinny_kernel<false, 1, 1, 2, UNR, true, true>), grid, block, (size_t)SK_WAVES * (UNR / 2) * 1024, s, q);
      return;
    }
  }
  if (a.mode == 2) {
    const dim3 grid((a.N + 31) / 32);
    if (a.S <= 16) launch_k<SPLIT, 2, 1, XMODE, UNR>(a, grid, s); else launch_k<SPLIT, 2, 2, XMODE, UNR>(a, grid, s);
  } else {
    const dim3 grid((a.N + 15) / 16);
    if (a.S <= 16) launch_k<SPLIT, 1, 1, XMODE, UNR>(a, grid, s); else launch_k<SPLIT, 1, 2, XMODE, UNR>(a, grid, s);
  }
}
template <bool SPLIT, int XMODE>
void launch_s(const SkinnyArgs& a, hipStream_t s) {
  const int per = (a.K / 32 + SK_WAVES - 1) / SK_WAVES;  // k-steps per wave
  // registers per step and lane: 4 (W) x tiles + 4 (bf16 x) or 8..16 (fp32 x [+ norm weight]) x sequence halves
  if constexpr (XMODE >= 2) {
    if (per <= 2) launch_u<SPLIT, XMODE, 2>(a, s); else if (per <= 4) launch_u<SPLIT, XMODE, 4>(a, s); else if (per <= 8) launch_u<SPLIT, XMODE, 8>(a, s); else launch_u<SPLIT, XMODE, 12>(a, s);
  } else {
    if (per <= 4) launch_u<SPLIT, XMODE, 4>(a, s); else launch_u<SPLIT, XMODE, 6>(a, s);
  }
}

template <int TILES, int SH, int XMODE, int UNR>
hipError_t allow_big_lds() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_kernel<false, TILES, SH, XMODE, UNR, true>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)sk_dyn_lds_max(TILES, SH));
}
template <int SH, int XMODE, int UNR>
hipError_t allow_glu_2pass() {  // the two-pass gate/up form: UNR = steps per pass
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_kernel<false, 2, SH, XMODE, UNR, true, false, true>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, SK_WAVES * 2 * (UNR / 2) * 2048);
}
template <int SH>
hipError_t allow_glu_hp3() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_kernel<false, 3, SH, 3, 4, true, false, true, true>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, SK_WAVES * 3 * 2 * 2048);
}
template <int XMODE, int UNR>
hipError_t allow_big_lds_shapes() {
  hipError_t e = allow_big_lds<1, 1, XMODE, UNR>();
  if (e == hipSuccess) e = allow_big_lds<1, 2, XMODE, UNR>();
  if (e == hipSuccess) e = allow_big_lds<2, 1, XMODE, UNR>();
  if (e == hipSuccess) e = allow_big_lds<2, 2, XMODE, UNR>();
  return e;
}

}  // namespace

// Once per device, outside any stream capture: the LDS-staged variants use up to 150 KiB of dynamic LDS.
const char* skinny_init() {
  hipError_t e = allow_big_lds_shapes<2, 2>();
  if (e == hipSuccess) e = allow_big_lds_shapes<3, 2>();
  if (e == hipSuccess) e = allow_big_lds_shapes<2, 4>();
  if (e == hipSuccess) e = allow_big_lds_shapes<2, 8>();
  if (e == hipSuccess) e = allow_big_lds_shapes<2, 12>();
  if (e == hipSuccess) e = allow_big_lds_shapes<3, 4>();
  if (e == hipSuccess) e = allow_big_lds_shapes<3, 8>();
  if (e == hipSuccess) e = allow_big_lds_shapes<3, 12>();
  if (e == hipSuccess) e = allow_big_lds_shapes<0, 4>();
  if (e == hipSuccess) e = allow_big_lds_shapes<0, 6>();
  if (e == hipSuccess) e = allow_big_lds_shapes<1, 4>();
  if (e == hipSuccess) e = allow_big_lds_shapes<1, 6>();
  if (e == hipSuccess) e = allow_glu_hp3<1>();
  if (e == hipSuccess) e = allow_glu_hp3<2>();
  if (e == hipSuccess) e = allow_glu_2pass<1, 2, 4>();
  if (e == hipSuccess) e = allow_glu_2pass<2, 2, 4>();
  if (e == hipSuccess) e = allow_glu_2pass<1, 3, 4>();
  if (e == hipSuccess) e = allow_glu_2pass<2, 3, 4>();
  if (e == hipSuccess) e = allow_glu_2pass<1, 2, 6>();
  if (e == hipSuccess) e = allow_glu_2pass<2, 2, 6>();
  if (e == hipSuccess) e = allow_glu_2pass<1, 3, 6>();
  if (e == hipSuccess) e = allow_glu_2pass<2, 3, 6>();
  return e == hipSuccess ? nullptr : "skinny gemm: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
}

const char* launch_skinny(const SkinnyArgs& a, bool split, hipStream_t s) {
  if (a.S <= 0) return nullptr;
  if (a.S > 32) return "skinny gemm: at most 32 sequences";
  if (a.K % 32 != 0 || a.ldx % 8 != 0) return "skinny gemm: K must be a multiple of 32, ldx of 8";
  if (a.mode == 2 && a.N % 32 != 0) return "skinny gemm: GLU needs N % 32 == 0";
  if (a.out16 && a.mode != 2) return "skinny gemm: bf16 output only in GLU mode";
  if (a.x16 && (split || a.rms_w)) return "skinny gemm: bf16 x excludes the precise mode and the fused RMSNorm";
  if (a.xw16f && (split || a.rms_w || a.x16 || !a.ss_parts || a.ss_nparts <= 0)) return "skinny gemm: pre-normalised input is exclusive and needs its partial sums";
  if (a.next_w && (a.mode != 1 || !a.next_xw16f || !a.next_ss)) return "skinny gemm: next-norm output needs mode 1 and both buffers";
  if (a.qsplit) {
    const int per = (a.K / 32 + SK_WAVES - 1) / SK_WAVES, unr = per <= 2 ? 2 : per <= 4 ? 4 : per <= 8 ? 8 : 12;
    if (split || a.mode != 1 || !a.x16 || !a.x16_frag || a.N % 64 != 0 || (a.K / 32) % SK_WAVES != 0 || per % unr != 0)
      return "skinny gemm: quarter workgroups need mode 1, bf16 fragment-order x, N % 64 == 0 and K % (256 * 2|4|8|12) == 0";
  }
  if (a.xw16f) launch_s<false, 3>(a, s);
  else if (a.x16) launch_s<false, 2>(a, s);
  else if (a.rms_w) { if (split) launch_s<true, 1>(a, s); else launch_s<false, 1>(a, s); }
  else { if (split) launch_s<true, 0>(a, s); else launch_s<false, 0>(a, s); }
  return nullptr;
}

}  // namespace q3a
