// MFMA flash attention over independent segments for gfx950 (default bf16 mode; the fp32 VALU kernel of
// k_attn.hip remains the precise-mode path and the on-device reference).
//
// Same contract as k_attn.hip (AttnArgs / AttnSeg): encoder window attention (src/layers.rs:152-172 with the
// block-diagonal mask of src/audio_encoder.rs:172-260 turned into segments) and the causal GQA prefill attention
// of src/layers.rs:321-335.
//
// Workgroup = 4 waves = (4/GROUP) tiles of 32 queries x GROUP query heads sharing one K/V head (KSPLIT = 2: one query tile,
// the waves also split the staged keys in halves -- see the kernel).  Per 32-key tile:
//   * the block stages K[32][HD] (bf16, rows padded by 16 B) and V^T[HD][32] (bf16) in LDS; the next tile's
//     global loads are already in flight in registers while the current one is consumed;
//   * S^T = K . Q^T on v_mfma_f32_32x32x16_bf16 ("swapped" product): a lane then owns ONE query column
//     (lane&31) and 16 of the 32 keys, so the online-softmax statistics are register-local plus one exchange with
//     lane^32, and the rescale factor of the running output is lane-local as well;
//   * P^T is packed to bf16 in registers and fed straight back as the B operand of O^T += V^T . P^T -- the
//     contraction index is permuted so that each lane's own 8 P values per k-step are exactly its B fragment;
//     the matching V^T fragment is two 8-byte LDS reads.
// Scores are divided by sqrt(hd) after the product as the reference does.  Q and P enter the MFMA as bf16, K/V
// are the bf16 cache (or fp32 projections rounded while staging): same rounding class as every other bf16 GEMM
// operand of the default mode.
#include <stdlib.h>

#include "dev.h"
#include "kernels.h"

namespace q3a {
namespace {

// Workgroup -> (query tile, kv head, segment).  The query tiles of one (segment, kv head) read the same K / V rows; with the plain
// blockIdx mapping their linear ids are consecutive, and the dispatcher deals consecutive workgroups to DIFFERENT XCDs -- at 32 clips
// the prefill's 7 query tiles per head sit on 7 XCDs, each XCD's L2 fetches the head's K / V for itself (round 6, from the memory-side
// counters: the launch moved several times its operands through the fabric).  Bijective remap (performance only; any placement is
// correct): workgroup b, on XCD b % 8, takes the (b / 8)-th triple of that XCD's contiguous chunk of the x-fastest order, so the
// query tiles of a head are neighbours ON ONE XCD and its K / V rows are fetched once per XCD.  -DQ3A_FATTN_XCD_REMAP=0: plain mapping.
#ifndef Q3A_FATTN_XCD_REMAP
#define Q3A_FATTN_XCD_REMAP 1
#endif
__device__ __forceinline__ void fattn_block(int& bx, int& by, int& bz) {
  bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
#if Q3A_FATTN_XCD_REMAP
  const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * (int)gridDim.z;
  const int lin = bx + gx * (by + gy * bz);
  const int xcd = lin & 7, loc = lin >> 3, q = total >> 3, r = total & 7;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  bx = id % gx;
  const int t = id / gx;
  by = t % gy;
  bz = t / gy;
#endif
}

template <typename KVT> struct Stage8;  // 8 consecutive head dims of one K/V row -> 8 packed bf16
template <> struct Stage8<uint16_t> {
  static __device__ __forceinline__ uint4 load(const uint16_t* p) { return *reinterpret_cast<const uint4*>(p); }
};
template <> struct Stage8<float> {
  static __device__ __forceinline__ uint4 load(const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    uint4 r;
    r.x = pack_bf16x2(a.x, a.y); r.y = pack_bf16x2(a.z, a.w); r.z = pack_bf16x2(b.x, b.y); r.w = pack_bf16x2(b.z, b.w);
    return r;
  }
};

// KSPLIT = 2 (one-clip prefill: few, long rows): the block's waves are (key half) x (query head) over ONE 32-query tile --
// a staged tile holds 2 x 32 keys, wave w works on half w / GROUP of it with its own online-softmax state, and the halves
// are merged through LDS at the end.  Twice the workgroups, half the barrier-separated iterations per workgroup.
template <int HD, int GROUP, bool CAUSAL, typename KVT, int KSPLIT = 1>
__global__ __launch_bounds__(256) void fattn_kernel(AttnArgs a) {
  constexpr int QTILES = 4 / (GROUP * KSPLIT);  // 32-query tiles per block
  static_assert(QTILES >= 1, "4 waves = query tiles x heads of the group x key halves");
  constexpr int QT = 32 * QTILES;            // queries per block
  constexpr int KT = 32 * KSPLIT;            // keys per staged tile (32 per wave)
  constexpr int KS = HD / 16;                // MFMA k-steps of the QK^T product
  constexpr int DT = HD / 32;                // 32-row tiles of O^T
  constexpr int K_STRIDE = HD + 8;           // bf16 per K row in LDS (16 B pad)
  constexpr int V_STRIDE = KT + 8;           // bf16 per V^T row in LDS
  constexpr int CPR = HD / 8;                // 16-B chunks per K/V row
  constexpr int LOADS = KT * CPR / 256;      // chunks per thread per tile (HD=128: 2, HD=64: 1)
  static_assert(LOADS >= 1, "tile too small");
  constexpr int KV_LDS = KT * K_STRIDE + HD * V_STRIDE;                       // bf16 elements of the staged tile
  constexpr int MERGE_LDS = KSPLIT > 1 ? GROUP * (32 * HD + 64) * 2 : 0;      // fp32 O^T + (m, l) of the second key half, in bf16 units
  __shared__ __attribute__((aligned(16))) uint16_t lds_all[KV_LDS > MERGE_LDS ? KV_LDS : MERGE_LDS];
  uint16_t* const k_lds = lds_all;
  uint16_t* const vt_lds = lds_all + KT * K_STRIDE;

  int blk_x, blk_y, blk_z;
  fattn_block(blk_x, blk_y, blk_z);
  const AttnSeg seg = a.segs[blk_z];
  const int kvh = blk_y;
  const int qb0 = blk_x * QT;
  if (qb0 >= seg.len) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int g = wave % GROUP;
  const int qs = KSPLIT > 1 ? 0 : wave / GROUP, kh = KSPLIT > 1 ? wave / GROUP : 0;  // query tile / key half of this wave
  const int head = kvh * GROUP + g;
  const int q0 = qb0 + qs * 32;              // first query of this wave
  const int qi = q0 + l31;                   // this lane's query (both halves of the wave hold the same 32 queries)
  const bool wave_has_q = q0 < seg.len;
  const KVT* kbase = reinterpret_cast<const KVT*>(a.k) + seg.kv_off + (int64_t)kvh * a.kv_hs;
  const KVT* vbase = reinterpret_cast<const KVT*>(a.v) + seg.kv_off + (int64_t)kvh * a.kv_hs;

  // ---- Q^T fragments (B operand): lane holds Q[qi][ks*16 + half*8 .. +8] as bf16 ----
  bf16x8_t qfrag[KS];
  if (a.q16) {  // Q already rounded to bf16 by the producing GEMM's epilogue (the very values the conversion below yields)
    const uint16_t* qrow = a.q16 + (size_t)(seg.q_row0 + (qi < seg.len ? qi : seg.len - 1)) * a.q_rs + head * HD;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qfrag[ks] = *reinterpret_cast<const bf16x8_t*>(qrow + ks * 16 + half * 8);
  } else {
    const float* qrow = a.q + (size_t)(seg.q_row0 + (qi < seg.len ? qi : seg.len - 1)) * a.q_rs + head * HD;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 x0 = *reinterpret_cast<const float4*>(qrow + ks * 16 + half * 8);
      const float4 x1 = *reinterpret_cast<const float4*>(qrow + ks * 16 + half * 8 + 4);
      uint4 p;
      p.x = pack_bf16x2(x0.x, x0.y); p.y = pack_bf16x2(x0.z, x0.w); p.z = pack_bf16x2(x1.x, x1.y); p.w = pack_bf16x2(x1.z, x1.w);
      qfrag[ks] = *reinterpret_cast<const bf16x8_t*>(&p);
    }
  }
  f32x16_t oacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;

  const int block_qmax = min(qb0 + QT, seg.len) - 1;
  const int n_keys = CAUSAL ? block_qmax + 1 : seg.len;
  const int n_tiles = (n_keys + KT - 1) / KT;
  const int wave_qmax = min(q0 + 32, seg.len) - 1;

  // staging assignment: chunk c of the tile = (row c / CPR, 8 dims (c % CPR)*8)
  uint4 kreg[LOADS], vreg[LOADS];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int c = tid + i * 256, row = c / CPR, d0 = (c % CPR) * 8;
      const int key = t * KT + row;
      if (key < seg.len) {
        kreg[i] = Stage8<KVT>::load(kbase + (int64_t)key * a.kv_rs + d0);
        vreg[i] = Stage8<KVT>::load(vbase + (int64_t)key * a.kv_rs + d0);
      } else {
        kreg[i] = make_uint4(0u, 0u, 0u, 0u);
        vreg[i] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int c = tid + i * 256, row = c / CPR, d0 = (c % CPR) * 8;
      *reinterpret_cast<uint4*>(&k_lds[row * K_STRIDE + d0]) = kreg[i];
      const uint32_t w[4] = {vreg[i].x, vreg[i].y, vreg[i].z, vreg[i].w};
      // transpose: V^T[d][key].  The 16 lanes that share a key write dims d0 = 0, 8, .., 120: rows 640 B apart, i.e.
      // two banks for all of them (measured: LDS bank-conflict rate 0.81, 27 % of the wave cycles waiting on LDS).
      // The 4-key groups of row d are therefore stored at group index (g ^ ((d >> 3) & 7)) -- the reads below apply
      // the same XOR -- which spreads those lanes over 8 bank pairs.
      const int ksw = row ^ (((d0 >> 3) & 7) << 2);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        vt_lds[(d0 + 2 * e) * V_STRIDE + ksw] = (uint16_t)(w[e] & 0xffffu);
        vt_lds[(d0 + 2 * e + 1) * V_STRIDE + ksw] = (uint16_t)(w[e] >> 16);
      }
    }
  };

  const float inv_scale = 1.0f / a.scale_div;  // one reciprocal instead of a division per score
  load_tile(0);
  for (int t = 0; t < n_tiles; ++t) {
    __syncthreads();  // every wave is done with the previous tile
    store_tile();
    __syncthreads();
    if (t + 1 < n_tiles) load_tile(t + 1);
    const int key0 = t * KT + kh * 32;  // first key of this wave's 32 of the staged tile
    if (!wave_has_q || (CAUSAL && key0 > wave_qmax)) continue;

    // ---- S^T[key][query] = sum_d K[key][d] Q[query][d] ----
    f32x16_t sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(&k_lds[(kh * 32 + l31) * K_STRIDE + ks * 16 + half * 8]);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qfrag[ks], sacc, 0, 0, 0);
    }
    // lane owns query qi and keys key0 + (r&3) + 8*(r>>2) + 4*half
    // (round 4, as in fattn_dma_kernel below: no masking on tiles that lie wholly in front of the wave's first query, and the
    // running output is rescaled only when some query of the wave saw a new maximum)
    const bool full = key0 + 31 <= (CAUSAL ? min(q0, seg.len - 1) : seg.len - 1);  // wave-uniform
    float tmax = -INFINITY;
    if (full) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sacc[r] *= inv_scale;
        tmax = fmaxf(tmax, sacc[r]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const bool valid = key < seg.len && (!CAUSAL || key <= qi);
        sacc[r] = valid ? sacc[r] * inv_scale : -INFINITY;
        tmax = fmaxf(tmax, sacc[r]);
      }
    }
    tmax = xor32_max(tmax);
    const float m_new = fmaxf(mrun, tmax);
    const bool rescale = __builtin_amdgcn_ballot_w64(m_new > mrun) != 0;
    float alpha = 1.f, psum = 0.f;
    if (m_new == -INFINITY) {  // nothing visible yet for this query (only for padding queries)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
    } else {
      alpha = __expf(mrun - m_new);  // hardware exponential: this kernel is the default (bf16) mode only
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sacc[r] = __expf(sacc[r] - m_new);  // exp(-inf) = 0 for masked keys
        psum += sacc[r];
      }
    }
    psum = xor32_sum(psum);
    lrun = lrun * alpha + psum;
    mrun = m_new;
    // ---- P^T fragments: k-step kk uses this lane's registers kk*8 .. kk*8+7 ----
    bf16x8_t pfrag[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 p;
      p.x = pack_bf16x2(sacc[kk * 8 + 0], sacc[kk * 8 + 1]); p.y = pack_bf16x2(sacc[kk * 8 + 2], sacc[kk * 8 + 3]);
      p.z = pack_bf16x2(sacc[kk * 8 + 4], sacc[kk * 8 + 5]); p.w = pack_bf16x2(sacc[kk * 8 + 6], sacc[kk * 8 + 7]);
      pfrag[kk] = *reinterpret_cast<const bf16x8_t*>(&p);
    }
    // ---- O^T[d][query] = alpha * O^T + sum_key V[key][d] P[key][query] ----
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      if (rescale) {
        asm volatile("" ::: "memory");  // keeps this a branch (hipcc if-converts the bare form into 64 multiplies + 64 selects)
        oacc[dt] *= alpha;  // vector op: packed multiplies
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        // contraction slot (half, e) <-> key 16*kk + 4*half + (e & 3) + 8*(e >> 2): same map as the P registers
        const uint16_t* vrow = &vt_lds[(dt * 32 + l31) * V_STRIDE];
        const int gsw = (dt * 4 + (l31 >> 3)) & 7;  // ((d >> 3) & 7) of this lane's row: the store-side XOR
        const uint2 lo = *reinterpret_cast<const uint2*>(vrow + kh * 32 + (((4 * kk + half) ^ gsw) << 2));
        const uint2 hi = *reinterpret_cast<const uint2*>(vrow + kh * 32 + (((4 * kk + half + 2) ^ gsw) << 2));
        uint4 v;
        v.x = lo.x; v.y = lo.y; v.z = hi.x; v.w = hi.y;
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&v), pfrag[kk], oacc[dt], 0, 0, 0);
      }
    }
  }
  if constexpr (KSPLIT > 1) {
    // ---- merge the key halves: half 1 parks (m, l, O^T) in LDS (the staged tiles are dead), half 0 folds it in ----
    __syncthreads();
    float* const mo = reinterpret_cast<float*>(lds_all) + g * (32 * HD + 64);  // [HD/32][16][64 lanes] O^T, then m[32], l[32]
    if (kh == 1) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mo[(dt * 16 + r) * 64 + lane] = oacc[dt][r];
      if (half == 0) { mo[32 * HD + l31] = mrun; mo[32 * HD + 32 + l31] = lrun; }
    }
    __syncthreads();
    if (kh == 1) return;
    const float m2 = mo[32 * HD + l31], l2 = mo[32 * HD + 32 + l31];
    const float mn = fmaxf(mrun, m2);
    if (mn != -INFINITY) {  // (padding queries past the segment see nothing in either half)
      const float f1 = __expf(mrun - mn), f2 = __expf(m2 - mn);  // exp(-inf) = 0 for a half that saw no key
      lrun = lrun * f1 + l2 * f2;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = oacc[dt][r] * f1 + mo[(dt * 16 + r) * 64 + lane] * f2;
    }
  }
  // ---- write O[query][d] = O^T / l ----
  if (wave_has_q && qi < seg.len) {
    const float inv = 1.0f / lrun;
    float* orow = a.o + (size_t)(seg.q_row0 + qi) * a.o_rs + head * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {  // registers 4*r4 .. 4*r4+3 are 4 consecutive head dims
        const int d = dt * 32 + 8 * r4 + 4 * half;
        const float4 v = make_float4(oacc[dt][4 * r4] * inv, oacc[dt][4 * r4 + 1] * inv, oacc[dt][4 * r4 + 2] * inv, oacc[dt][4 * r4 + 3] * inv);
        if (a.o16) {  // bf16 output feeds the out/o projection GEMM's LDS-DMA
          uint2 pk;
          pk.x = pack_bf16x2(v.x, v.y);
          pk.y = pack_bf16x2(v.z, v.w);
          *reinterpret_cast<uint2*>(a.o16 + (size_t)(seg.q_row0 + qi) * a.o_rs + head * HD + d) = pk;
        } else {
          *reinterpret_cast<float4*>(orow + d) = v;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 4: the same attention with the K/V tiles staged by LDS-DMA into a ring of NS stages, V read with the hardware transpose.
//
// The kernel above spends ~3 us per 32-key iteration for 16 MFMAs per wave (0.3 us): every iteration exposes a full memory round
// trip (load -> barrier -> register->LDS store with sixteen 2-byte transposing stores per thread -> barrier), and 244 registers
// leave no room for a second tile in flight.  Here
//   * K and V rows go HBM/L2 -> LDS with global_load_lds (1 KiB per wave instruction = 4 rows of 256 B / 8 rows of 128 B), no
//     staging registers, into a ring of NS = 3 (4) tiles of 32 keys (48 / 64 KiB at HD = 128: three / two workgroups per CU): NS - 1 tiles are in
//     flight while one is consumed, ONE raw s_barrier per tile, counted vmcnt (wait for my pieces of tile t -> barrier -> tile t
//     is whole and tile t-1's stage is free -> refill it);
//   * the LDS image is lane-linear, so the bank swizzles are applied to the SOURCE chunk index: K rows 16-B chunk c of row r at
//     position c ^ (r & 15) (HD = 64: c ^ ((r >> 1) & 7)) -- conflict-free ds_read_b128 fragments; V rows 32-B block b of row r at
//     b ^ (2 (r & 3)) (HD = 64: b ^ (r & 2)) -- the four rows and two column blocks a half-wave's ds_read_b64_tr_b16 touches fall
//     on eight distinct 8-bank slots;
//   * V stays row-major [key][HD]: ds_read_b64_tr_b16 hands lane (d = lane & 31) the four keys 16 kk + 4 half (+8) + 0..3 of
//     column d -- exactly the contraction slots the P^T registers of the swapped product hold (tools/tr_probe.hip prints the map).
// Everything else (swapped S^T = K Q^T, register-local online softmax, P^T fed back as the B operand) is the kernel above.
// bf16 q (a.q16) and bf16 K/V only: the default mode.
typedef const void __attribute__((address_space(1)))* fa_gptr_t;
typedef void __attribute__((address_space(3)))* fa_lptr_t;
typedef short fa_v4s __attribute__((ext_vector_type(4)));

template <int HD, int GROUP, bool CAUSAL, int NS>
__global__ __launch_bounds__(256, NS == 3 ? 3 : 2) void fattn_dma_kernel(AttnArgs a) {
  constexpr int KT = 32;
  constexpr int QTILES = 4 / GROUP, QT = 32 * QTILES;
  constexpr int KS = HD / 16, DT = HD / 32;
  constexpr int ROW_B = HD * 2;              // bytes per K / V row
  constexpr int RPP = 1024 / ROW_B;          // rows per DMA piece (one wave instruction)
  constexpr int CPR = HD / 8;                // 16-B chunks per row
  constexpr int PIECES = KT / RPP;           // pieces per operand and tile (HD 128: 8, HD 64: 4)
  constexpr int PPW = PIECES / 4;            // pieces per wave, operand and tile
  constexpr int OPB = KT * ROW_B;            // bytes of one operand of a stage
  constexpr int STAGE = 2 * OPB;
  static_assert(PPW >= 1, "tile too small for four waves");
  __shared__ __attribute__((aligned(1024))) uint8_t ring[NS * STAGE];  // the ONLY LDS object: [stage][K | V][row][HD]

  int blk_x, blk_y, blk_z;
  fattn_block(blk_x, blk_y, blk_z);
  const AttnSeg seg = a.segs[blk_z];
  const int kvh = blk_y;
  const int qb0 = blk_x * QT;
  if (qb0 >= seg.len) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int g = wave % GROUP, qs = wave / GROUP;
  const int head = kvh * GROUP + g;
  const int q0 = qb0 + qs * 32, qi = q0 + l31;
  const bool wave_has_q = q0 < seg.len;
  const uint16_t* kbase = reinterpret_cast<const uint16_t*>(a.k) + seg.kv_off + (int64_t)kvh * a.kv_hs;
  const uint16_t* vbase = reinterpret_cast<const uint16_t*>(a.v) + seg.kv_off + (int64_t)kvh * a.kv_hs;

  const int block_qmax = min(qb0 + QT, seg.len) - 1;
  const int n_keys = CAUSAL ? block_qmax + 1 : seg.len;
  const int n_tiles = (n_keys + KT - 1) / KT;
  const int wave_qmax = min(q0 + 32, seg.len) - 1;

  // ---- DMA assignment: piece p of an operand = rows p*RPP .. +RPP-1; lane -> (row inside the piece, LDS chunk position) ----
  const int prow = lane / CPR, pc = lane % CPR;
  int srcK[PPW], srcV[PPW], rowt[PPW];  // source chunk (16-B units) of this lane for K / V, tile row of the lane, per piece
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int p = wave * PPW + i, r = p * RPP + prow;
    rowt[i] = r;
    srcK[i] = HD == 128 ? (pc ^ (r & 15)) : (pc ^ ((r >> 1) & 7));
    const int b = pc >> 1, bs = HD == 128 ? (b ^ (2 * (r & 3))) : (b ^ (r & 2));
    srcV[i] = (bs << 1) | (pc & 1);
  }
  const int last_key = seg.len - 1;
  auto issue = [&](int t) {  // tile t -> stage t % NS; rows past the segment are clamped to its last row (masked in the scores)
    uint8_t* st = ring + (t % NS) * STAGE;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int key = min(t * KT + rowt[i], last_key);
      const int p = wave * PPW + i;
      __builtin_amdgcn_global_load_lds((fa_gptr_t)(kbase + (int64_t)key * a.kv_rs + srcK[i] * 8), (fa_lptr_t)(st + p * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((fa_gptr_t)(vbase + (int64_t)key * a.kv_rs + srcV[i] * 8), (fa_lptr_t)(st + OPB + p * 1024), 16, 0, 0);
    }
  };
  constexpr int PER_TILE = 2 * PPW;  // DMA instructions per wave and tile
#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < n_tiles) issue(t);

  // ---- Q^T fragments (B operand): lane holds Q[qi][ks*16 + half*8 .. +8].  Requested BEHIND the first tiles and pinned in
  // front of the loop: hipcc cannot count the loop's DMA instructions, so a first use inside the loop gets an s_waitcnt vmcnt(0)
  // in every iteration; here the one wait it inserts also covers the prologue tiles (one round trip, once per workgroup)
  bf16x8_t qfrag[KS];
  {
    const uint16_t* qrow = a.q16 + (size_t)(seg.q_row0 + (qi < seg.len ? qi : seg.len - 1)) * a.q_rs + head * HD;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qfrag[ks] = *reinterpret_cast<const bf16x8_t*>(qrow + ks * 16 + half * 8);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qfrag[ks]));
  }

  f32x16_t oacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;  // mrun: running maximum of s * scale2 (base-2 domain)
  const float scale2 = 1.44269504088896340736f / a.scale_div;  // log2(e) / sqrt(hd): layers.rs:327-328 divides after the product

  // fragment read offsets inside a stage (bytes)
  int koff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int c = ks * 2 + half;
    koff[ks] = l31 * ROW_B + ((HD == 128 ? (c ^ (l31 & 15)) : (c ^ ((l31 >> 1) & 7))) << 4);
  }
  const int vr = (lane & 15) >> 2, vu = lane & 3, db = (lane >> 4) & 1;  // tr-read: row inside the 4-key block, 8-B unit, column block
  unsigned voff[DT];  // byte offset inside a stage's V region of this lane's transpose-read source for O^T tile dt (k-step 0, first 4 keys)
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    const int b = dt * 2 + db, bs = HD == 128 ? (b ^ (2 * vr)) : (b ^ (vr & 2));  // (row & 3) == vr for every row this lane supplies
    voff[dt] = (4 * half + vr) * ROW_B + bs * 32 + vu * 8;
  }

  for (int t = 0; t < n_tiles; ++t) {
    // my pieces of tile t have landed when at most the later tiles' instructions are outstanding
    const int later = min(NS - 2, n_tiles - 1 - t);  // tiles issued after t that may still be in flight (wave-uniform)
    if (NS >= 4 && later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_TILE) : "memory");
    else if (later >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");  // tile t is whole; every wave is done with tile t - 1
    if (t + NS - 1 < n_tiles) issue(t + NS - 1);
    const uint8_t* st = ring + (t % NS) * STAGE;
    const int key0 = t * KT;
    if (!wave_has_q || (CAUSAL && key0 > wave_qmax)) continue;

    // ---- S^T[key][query] = sum_d K[key][d] Q[query][d] ----
    f32x16_t sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(st + koff[ks]);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qfrag[ks], sacc, 0, 0, 0);
    }
    // Scores in the base-2 domain: s2 = s * (log2 e / sqrt(hd)), so that a probability is ONE v_exp_f32 of ONE fma (the kernel
    // above spends mul + sub + mul + exp per score); the running max is kept in that domain too.  Tiles that lie wholly in front
    // of the wave's first query and inside the segment need no masking (wave-uniform test).
    const bool full = key0 + 31 <= (CAUSAL ? min(q0, seg.len - 1) : seg.len - 1);
    float tmax = -INFINITY;
    if (full) {
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[r]);
      tmax *= scale2;  // scale2 > 0: the maximum commutes with the scaling
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const bool valid = key < seg.len && (!CAUSAL || key <= qi);
        sacc[r] = valid ? sacc[r] : -INFINITY;
        tmax = fmaxf(tmax, sacc[r]);
      }
      tmax *= scale2;  // (-inf stays -inf)
    }
    tmax = xor32_max(tmax);
    const float m_new = fmaxf(mrun, tmax);
    float psum = 0.f;
    // the running output is rescaled only when some query of the wave saw a new maximum (alpha = 1 everywhere otherwise: the
    // 64 multiplies per lane would be the identity)
    const bool rescale = __builtin_amdgcn_ballot_w64(m_new > mrun) != 0;
    // (a padding query past the segment sees only masked keys: m_new = -inf.  Subtracting a FINITE floor instead keeps its
    // probabilities at exp2(-inf) = 0 without a branch around the 16 exponentials)
    const float m_use = fmaxf(m_new, -1e30f);
    float alpha = 1.f;
    if (rescale) alpha = __builtin_amdgcn_exp2f(mrun - m_use);  // exp2(-inf) = 0 on the first live tile
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sacc[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], scale2, -m_use));  // exp2(-inf) = 0 for masked keys
      psum += sacc[r];
    }
    psum = xor32_sum(psum);
    lrun = lrun * alpha + psum;
    mrun = m_new;
    bf16x8_t pfrag[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 p;
      p.x = pack_bf16x2(sacc[kk * 8 + 0], sacc[kk * 8 + 1]); p.y = pack_bf16x2(sacc[kk * 8 + 2], sacc[kk * 8 + 3]);
      p.z = pack_bf16x2(sacc[kk * 8 + 4], sacc[kk * 8 + 5]); p.w = pack_bf16x2(sacc[kk * 8 + 6], sacc[kk * 8 + 7]);
      pfrag[kk] = *reinterpret_cast<const bf16x8_t*>(&p);
    }
    // ---- O^T[d][query] = alpha * O^T + sum_key V[key][d] P[key][query]: V^T fragments by transpose reads ----
    // The transpose reads are inline asm: as a builtin (an LDS load hipcc can see) each of them got an s_waitcnt vmcnt(0) in
    // front -- the compiler cannot prove that the stage it reads is not the one a DMA in flight writes.  So the waits are ours:
    // the four reads of O^T tile dt + 1 are issued before the MFMAs of tile dt (LDS returns in order: lgkmcnt(4) = tile dt is in).
    const unsigned vstage = (unsigned)(size_t)(st + OPB);
    typedef unsigned fa_u2 __attribute__((ext_vector_type(2)));
    fa_u2 vt[DT][4];  // [dt][kk * 2 + (keys +8)]
    auto tr_reads = [&](int dt) {
      const unsigned ad = vstage + voff[dt];  // row 4 half + vr, unit vu, swizzled block of O^T tile dt; kk / +8 keys: immediates
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vt[dt][0]) : "v"(ad) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vt[dt][1]) : "v"(ad), "n"(8 * ROW_B) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vt[dt][2]) : "v"(ad), "n"(16 * ROW_B) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vt[dt][3]) : "v"(ad), "n"(24 * ROW_B) : "memory");
    };
    tr_reads(0);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      if (rescale) {
        asm volatile("" ::: "memory");  // keeps this a BRANCH: hipcc if-converts the bare form into 64 multiplies + 64 v_cndmask per tile
        oacc[dt] *= alpha;
      }
      // the wait statement takes tile dt's registers as in/out operands: the MFMAs below depend on IT, not on the reads (hipcc
      // assumes an asm's outputs are ready when the asm has been issued and would hoist the MFMAs above a bare wait)
      if (dt + 1 < DT) {
        tr_reads(dt + 1);
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(vt[dt][0]), "+v"(vt[dt][1]), "+v"(vt[dt][2]), "+v"(vt[dt][3])::"memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vt[dt][0]), "+v"(vt[dt][1]), "+v"(vt[dt][2]), "+v"(vt[dt][3])::"memory");
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        // slots e = 0..3: keys 16 kk + 4 half + 0..3 (first read), e = 4..7: the same + 8 (second read)
        const uint4 v = make_uint4(vt[dt][kk * 2].x, vt[dt][kk * 2].y, vt[dt][kk * 2 + 1].x, vt[dt][kk * 2 + 1].y);
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, v), pfrag[kk], oacc[dt], 0, 0, 0);
      }
    }
  }
  // ---- write O[query][d] = O^T / l ----
  if (wave_has_q && qi < seg.len) {
    const float inv = 1.0f / lrun;
    float* orow = a.o + (size_t)(seg.q_row0 + qi) * a.o_rs + head * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d = dt * 32 + 8 * r4 + 4 * half;
        const float4 v = make_float4(oacc[dt][4 * r4] * inv, oacc[dt][4 * r4 + 1] * inv, oacc[dt][4 * r4 + 2] * inv, oacc[dt][4 * r4 + 3] * inv);
        if (a.o16) {
          uint2 pk;
          pk.x = pack_bf16x2(v.x, v.y);
          pk.y = pack_bf16x2(v.z, v.w);
          *reinterpret_cast<uint2*>(a.o16 + (size_t)(seg.q_row0 + qi) * a.o_rs + head * HD + d) = pk;
        } else {
          *reinterpret_cast<float4*>(orow + d) = v;
        }
      }
  }
}

template <int HD, int GROUP, bool CAUSAL>
void launch_dma(const AttnArgs& a, hipStream_t s) {
  constexpr int QT = 32 * (4 / GROUP);
  dim3 grid((a.max_len + QT - 1) / QT, a.n_kv_heads, a.n_segs);
  // ring depth: 3 stages = 48 KiB at HD = 128, three workgroups per CU (A/B knob Q3A_FATTN_NS=4: 64 KiB, two per CU)
  static const int ns = [] { const char* e = getenv("Q3A_FATTN_NS"); return e ? atoi(e) : 3; }();
  // (round 5's hand-pipelined form of this kernel -- second score register set, softmax of tile t between the QK MFMAs of tile t + 1 --
  // was bit-identical and slower, 64.2 vs 60.5 us per prefill layer; removed in round 6, docs/HISTORY.md, last present at commit caf7a05)
  if (ns == 4) hipLaunchKernelGGL((fattn_dma_kernel<HD, GROUP, CAUSAL, 4>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((fattn_dma_kernel<HD, GROUP, CAUSAL, 3>), grid, dim3(256), 0, s, a);
}
// A/B knob: 0 = the register-staged kernel everywhere (round 3)
static bool fattn_dma_on() {
  static const bool on = [] { const char* e = getenv("Q3A_FATTN_DMA"); return !e || atoi(e) != 0; }();
  return on;
}

// What the LDS-DMA kernel assumes beyond the register-staged one: 16-byte global_load_lds sources (k / v base pointers, head and
// row strides in whole 8-element groups) and bf16x8 q loads.  A caller that does not meet it gets the register-staged kernel,
// not wrong data.  (AttnSeg::kv_off lives in device memory: multiples of 8 elements are the caller's contract -- the engine's
// offsets are multiples of the row size, engine.cpp set_batch / setup_prompts.)
static bool fattn_dma_ok(const AttnArgs& a) {
  const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return fattn_dma_on() && a.q16 && al16(a.q16) && al16(a.k) && al16(a.v) && a.q_rs % 8 == 0 && a.kv_rs % 8 == 0 && a.kv_hs % 8 == 0;
}

template <int HD, int GROUP, bool CAUSAL, typename KVT, int KSPLIT = 1>
void launch_f(const AttnArgs& a, hipStream_t s) {
  constexpr int QT = 32 * (4 / (GROUP * KSPLIT));
  dim3 grid((a.max_len + QT - 1) / QT, a.n_kv_heads, a.n_segs);
  hipLaunchKernelGGL((fattn_kernel<HD, GROUP, CAUSAL, KVT, KSPLIT>), grid, dim3(256), 0, s, a);
}

}  // namespace

const char* launch_fattn_enc(const AttnArgs& a, hipStream_t s) {
  if (a.n_segs <= 0) return nullptr;
  if (a.q_rs % 4 != 0 || a.kv_rs % 4 != 0 || a.o_rs % 4 != 0) return "fattn: row strides must be multiples of 4";
  if (a.q16) {  // bf16 q/k/v projections (k and v point into the same bf16 buffer)
    if (a.q_rs % 8 != 0 || a.kv_rs % 8 != 0) return "fattn: bf16 row strides must be multiples of 8";
    if (fattn_dma_ok(a)) launch_dma<64, 1, false>(a, s);
    else launch_f<64, 1, false, uint16_t>(a, s);
  } else {
    launch_f<64, 1, false, float>(a, s);
  }
  return nullptr;
}

const char* launch_fattn_prefill(const AttnArgs& a, int group, hipStream_t s) {
  if (a.n_segs <= 0) return nullptr;
  if (a.q_rs % 4 != 0 || a.o_rs % 4 != 0) return "fattn: row strides must be multiples of 4";
  if (a.q16 && a.q_rs % 8 != 0) return "fattn: bf16 q row stride must be a multiple of 8";
  // key halves per workgroup when the plain shape leaves most CUs without a workgroup (one or a few clips); knob for A/B runs
  static const int ks_wgs = [] { const char* e = getenv("Q3A_FATTN_KSPLIT_MAX_WGS"); return e ? atoi(e) : 128; }();  // measured: 1 clip (56 workgroups) 22.6 -> 17.7 us per layer, 8 clips (448) 6.8 -> 7.4 ms prefill
  const long wgs64 = (long)((a.max_len + 63) / 64) * a.n_kv_heads * a.n_segs;
  const bool dma = fattn_dma_ok(a);
  if (group == 1) { if (dma) launch_dma<128, 1, true>(a, s); else launch_f<128, 1, true, uint16_t>(a, s); }
  else if (group == 2 && wgs64 < ks_wgs) launch_f<128, 2, true, uint16_t, 2>(a, s);
  else if (group == 2) { if (dma) launch_dma<128, 2, true>(a, s); else launch_f<128, 2, true, uint16_t>(a, s); }
  else if (group == 4) { if (dma) launch_dma<128, 4, true>(a, s); else launch_f<128, 4, true, uint16_t>(a, s); }
  else return "fattn: GQA group must be 1, 2 or 4";
  return nullptr;
}

}  // namespace q3a
