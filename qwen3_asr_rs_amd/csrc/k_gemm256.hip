// 256 x 256 x 64 bf16 MFMA GEMM for gfx950, 8 waves, LDS-DMA staging with counted vmcnt across raw barriers.
//
//   Y[M][N] = act( X[M][K] . W[N][K]^T + bias ) (+ residual)      X, W bf16 (K contiguous), fp32 accumulate
//
// Same contract / epilogues as k_gemm16.hip (GemmEpilogue); this is the kernel for the batch-sized problems of the
// encoder and the prefill (M = 12 480 ... 768 000 rows at 32 clips), where the 128 x 128, 4-wave, one-barrier-per-K-tile
// kernel of k_gemm16.hip parks its waves 2/3 of the time at that barrier (profiles/r1_sq_breakdown.txt).
//
// Structure (cdna_hip_programming.md "256^2 8-phase template", re-derived here because its source is not in the image):
//   * workgroup = 8 waves = 2 (M) x 4 (N); a wave owns a 128 x 64 output tile = 8 x 4 fragments of
//     v_mfma_f32_16x16x32_bf16 (128 accumulator registers);
//   * LDS = 2 K-tile buffers x (A 256 x 64 + B 256 x 64) bf16 = 128 KiB, ONE __shared__ array, rows of 128 B (one K tile
//     row = one cache line).  One global_load_lds wave instruction fills 8 rows (1 KiB, 8 full cache lines: half the
//     line touches of a 16-row x 64-B fragment-shaped load).  The 16-B chunk c of row r sits at chunk position
//     c ^ ((r >> 1) & 7); the LDS-DMA image is lane-linear, so the permutation is applied to the SOURCE address (inside
//     one cache line) and again to the fragment read address: the 16 lanes of a ds_read_b128 group hit 16 distinct
//     16-B bank slots;
//   * a K tile is consumed in 4 phases of 16 MFMAs per wave (one 64 x 32 quadrant of the wave tile x K = 64):
//         P1  read A rows m0 (8 reads) + B rows n0 (4)     MFMA m0 x n0
//         P2  read B rows n1 (4)                            MFMA m0 x n1
//         P3  read A rows m1 (8)                            MFMA m1 x n1
//         P4  (B n0 still in registers)                     MFMA m1 x n0
//     every phase = { fragment reads + ONE 16 KiB staging unit (2 LDS-DMA instructions per wave) ; s_barrier ;
//     lgkmcnt(0) ; s_setprio 1 ; 16 MFMA ; s_setprio 0 ; s_barrier }.  The two wave groups (wr = 0 / 1, one wave of each
//     on every SIMD) run ONE barrier apart, so one group's MFMA segment overlaps the other's read/stage segment;
//   * staging units are the row sets that are freed together: A-m0 / B-n0 (last read in P1), B-n1 (P2), A-m1 (P3).  A unit
//     is re-staged >= 2 phases after its last read (WAR, the groups being one barrier apart) and waited for with a
//     COUNTED vmcnt one phase before it is read (RAW: wait -> barrier -> read):
//         P1(t) stages B-n1(t+1)   P2(t) stages A-m1(t+1)   P3(t) stages A-m0(t+2)   P4(t) stages B-n0(t+2)
//     so four units (8 DMA instructions per wave, 64 KiB per CU) are always in flight and nothing in the K loop ever
//     waits with vmcnt(0).  The barriers are raw s_barrier (a __syncthreads() would drain the DMA queue);
//   * round 6: a launch is min(tiles, CUs) PERSISTENT workgroups that walk the tiles, the K-tile stream running on across
//     output tiles (comment at the kernel), in a tile order chosen for the XCDs' L2s (launch256).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "dev.h"
#include "kernels.h"

namespace q3a {
namespace {

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

constexpr int G_BM = 256, G_BN = 256, G_BK = 64;
constexpr int G_BUF = 65536;    // bytes per K-tile buffer: A 32 KiB then B 32 KiB
constexpr int G_BOFF = 32768;   // B region inside a buffer

// ---- A-operand views --------------------------------------------------------------------------------------------
// Row state is per lane and per LDS-DMA instruction: the lane stages 16 B (8 bf16) of ONE row of the K tile,
// source chunk `chunk` (0..7) of the row's 64 columns.  src(row, ka, kb): ka / kb = column iterators of the first / second
// 32-column half of the tile (wave-uniform); the lane's half is chunk >> 2.
struct DenseA256 {
  const uint16_t* x;
  int lda;
  struct Row { const uint16_t* p; };
  __device__ __forceinline__ Row init(int m, int M, int chunk) const {
    if (m >= M) m = M - 1;  // tail rows are clamped: their products are never stored
    return Row{x + (size_t)m * lda + chunk * 8};
  }
  typedef int KIter;  // first column of a 32-wide half
  __device__ __forceinline__ KIter kbegin() const { return 0; }
  __device__ __forceinline__ KIter kbegin_fresh() const { return 0; }
  __device__ __forceinline__ void knext(KIter& k) const { k += 32; }
  __device__ __forceinline__ const uint16_t* src(const Row& r, const KIter& ka, const KIter&, int) const { return r.p + ka; }
};

// im2col view of an NHWC bf16 feature map [img][H][W][C] for a 3x3 stride-2 pad-1 convolution; K = 9 * C ordered
// (kh, kw, c).  C % 32 == 0, so a 32-wide half row of a K tile never straddles two filter taps (the two halves may).
struct ConvA256 {
  const uint16_t* x;
  const uint16_t* zero;  // >= 128 B of zeros for padded taps and for the half tile past K (K % 64 == 32)
  int H, W, C, OH, OW;
  // Per staged row (fixed for the whole tile): byte address of (img, ih0, iw0, channel chunk) -- may point in front of an image
  // row -- and a 9-bit mask of the filter taps that fall inside the image, so that the per-K-tile work of a lane is one shift +
  // one 64-bit add + one select (the round-3 form recomputed ih / iw and five comparisons per staging instruction: ~20 VALU
  // instructions x 4 per K tile and wave next to 64 MFMAs; profiles/r4_conv_loader.txt).
  struct Row { const uint16_t* p; unsigned taps; };  // (the zero-page pointer and the half index follow from the lane's chunk, which src() is given: 3 registers per row, not 6 -- the walk keeps the rows alive across the epilogue)
  __device__ __forceinline__ Row init(int m, int M, int chunk) const {
    if (m >= M) m = M - 1;
    // the divisors pass through an empty asm: hipcc otherwise hoists the reciprocal set-up of both divisions out of the walk's loops and
    // keeps it in vector registers across the K loop (rows are initialised once per output tile)
    int ohow = OH * OW, oww = OW;
    asm volatile("" : "+s"(ohow), "+s"(oww));
    const int img = m / ohow;
    const int r = m - img * ohow;
    const int oh = r / oww, ow = r - oh * oww;
    const int ih0 = oh * 2 - 1, iw0 = ow * 2 - 1;
    unsigned taps = 0;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
        if (ih0 + kh >= 0 && ih0 + kh < H && iw0 + kw >= 0 && iw0 + kw < W) taps |= 1u << (kh * 3 + kw);
    const int sub = (chunk & 3) * 8;
    return Row{x + ((((long)img * H + ih0) * W + iw0) * C + sub), taps};
  }
  struct KIter { int tap, off; };  // filter tap (kh * 3 + kw; 9 = past K) and element offset (kh * W + kw) * C + c0 of a 32-wide half (scalar registers)
  __device__ __forceinline__ KIter kbegin() const { return KIter{0, 0}; }
  // the same through an empty asm, for the K tile 0 of a NEXT output tile inside the walk: with literal zeros hipcc precomputes the
  // per-lane tap / offset selects of src() for those constant positions in front of all loops and keeps (spills) them
  __device__ __forceinline__ KIter kbegin_fresh() const {
    int t = 0, o = 0;
    asm volatile("" : "+s"(t), "+s"(o));
    return KIter{t, o};
  }
  __device__ __forceinline__ void knext(KIter& k) const {
    k.off += 32;
    const int c_end = ((k.tap / 3) * W + (k.tap % 3)) * C + C;  // first offset past this tap's channels
    if (k.off >= c_end) {
      ++k.tap;
      k.off = ((k.tap / 3) * W + (k.tap % 3)) * C;
    }
  }
  __device__ __forceinline__ const uint16_t* src(const Row& r, const KIter& ka, const KIter& kb, int chunk) const {
    const bool hb = chunk >= 4;
    const int tap = hb ? kb.tap : ka.tap, off = hb ? kb.off : ka.off;
    return ((r.taps >> tap) & 1u) ? r.p + off : zero + (chunk & 3) * 8;  // tap == 9 (the zero half tile past K = 9 * C): bit 9 is never set
  }
};

#define Q3A_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define Q3A_BARRIER() asm volatile("s_barrier" ::: "memory")
#define Q3A_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int N> __device__ __forceinline__ void wait_vm() {
  static_assert(N == 0 || N == 2 || N == 4 || N == 8, "vmcnt value");
  if constexpr (N == 0) Q3A_WAIT_VM(0);
  else if constexpr (N == 2) Q3A_WAIT_VM(2);
  else if constexpr (N == 4) Q3A_WAIT_VM(4);
  else Q3A_WAIT_VM(8);
}

// ROPE: the qkv projection of the prefill with the per-head RMSNorm, RoPE and KV-cache append of
// qknorm_rope_kv_kernel (k_decode.hip) as its epilogue -- see the epilogue below.
//
// PERSISTENT TILE LOOP (round 6).  The grid is min(tiles, CUs) workgroups; workgroup b (XCD b % 8 under round-robin dispatch) walks
// the tiles loc, loc + w, loc + 2 w, ... of ITS XCD's contiguous chunk of the tile order (tile_origin below; w = workgroups on that
// XCD), so the tiles an XCD holds at any time are consecutive in that order: neighbours in its L2.  The K-tile stream does not stop at an output tile's end: the
// staging slots of the last two K tiles of tile i, which used to stay empty (the N1 / N2 = false tails), carry K tile 0 of tile
// i + 1 -- the per-lane row pointers are re-initialised in place right behind their last use for tile i -- so tile i + 1's 64 KiB
// land UNDER tile i's epilogue, and the epilogue's stores drain under tile i + 1's first K tiles instead of in front of a workgroup
// exit.  For that the epilogue may only use the K-tile buffer the next tile is NOT arriving in: its staging is 8 KiB per wave
// (four 32-row passes) instead of 16 KiB (two 64-row passes).  Per-wave vmcnt bookkeeping across the seam:
//   * after pass 0's accumulators are written to LDS and BEFORE the epilogue's first global request, vmcnt(0): the only requests in
//     flight are the wave's four LDS-DMA instructions of the next K tile 0, issued two and three phases earlier (stores and LDS-DMA
//     loads share vmcnt but retire independently, so a counted wait with both kinds in flight proves nothing about the loads);
//   * the barrier in front of the re-entry publishes that to the workgroup and frees the staging half for K tile 1's early units;
//   * the first K tile behind an epilogue skips the P1 / P2 waits (its units have landed; waiting would wait for the store acks);
//     its P4 wait counts the eight youngest requests as always -- the stores are older, so it can only wait longer, never shorter.
// Same arithmetic in the same order as one workgroup per tile (knob gemm256_persist = 0: grid = tiles): bit-identical results.
template <bool GLU, class ALoader, bool ROPE = false>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(ALoader A, const uint16_t* __restrict__ Wt,
                                                         const uint16_t* __restrict__ zero, int M, int N, int K, GemmEpilogue ep,
                                                         RopeKvArgs rk, int group_m) {
  __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * G_BUF];  // the ONLY LDS object of this kernel

  const int tiles_m = (M + G_BM - 1) / G_BM, tiles_n = (N + G_BN - 1) / G_BN;
  const int nwg = tiles_m * tiles_n;
  // this XCD's chunk of the tile order (the bijective remap of the one-tile-per-workgroup form) and this workgroup's walk through it
  // (a grid of fewer than 8 workgroups -- only tests launch one -- cuts the tile order into as many chunks as it has workgroups)
  const int G = gridDim.x;
  int xcd, xq, xr, wpx, tj;  // chunk of this workgroup, chunk sizes (xq or xq + 1 tiles), workgroups of the launch on this chunk, index inside the chunk
  if (G >= 8) {
    xcd = blockIdx.x & 7; xq = nwg >> 3; xr = nwg & 7;
    wpx = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
    tj = blockIdx.x >> 3;
  } else {
    xcd = blockIdx.x; xq = nwg / G; xr = nwg % G;
    wpx = 1;
    tj = 0;
  }
  const int x_first = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq, x_cnt = xq + (xcd < xr ? 1 : 0);
  if (tj >= x_cnt) return;                              // (G <= tiles: never)
  int bid = x_first + tj;
  Q3A_STAMP_AT(ep.stamp, bid, 0);  // entry
  // Tile order: groups of group_m tile rows, M fastest inside a group (group_m = 1: N fastest over the whole matrix).  The 32 tiles an
  // XCD holds at a time are 32 consecutive ids = group_m rows x 32 / group_m columns: with 8 rows x 4 columns they share 8 A panels and 4
  // W panels instead of 2 + 24 (gate / up: 24 column tiles) -- the XCD's L2 fetches 12 K-tile rows per step instead of 26.
  auto tile_origin = [&](int id, int& mo, int& no) {
    const int per_group = group_m * tiles_n, grp = id / per_group, in_g = id - grp * per_group;
    const int first_m = grp * group_m, gsize = min(tiles_m - first_m, group_m);
    const int tn = in_g / gsize, tm = first_m + (in_g - tn * gsize);
    mo = tm * G_BM; no = tn * G_BN;
  };
  int m0, n0;
  tile_origin(bid, m0, n0);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // ---- staging assignment: per unit a wave fills ONE 16-row block with two LDS-DMA instructions of 8 rows x 128 B ----
  // instruction j: tile row rb*16 + j*8 + (lane >> 3), LDS chunk position lane & 7 = source chunk ^ ((row >> 1) & 7)
  const int srow = lane >> 3;
  const int sch0 = (lane & 7) ^ ((lane >> 4) & 7), sch1 = sch0 ^ 4;  // source chunks of instructions 0 / 1
  const int rbA0 = wave + (wave >= 4 ? 4 : 0), rbA1 = rbA0 + 4;    // A-m0: row blocks {0-3, 8-11}; A-m1: {4-7, 12-15}
  const int rbB0 = (wave >> 1) * 4 + (wave & 1), rbB1 = rbB0 + 2;  // B-n0: {0,1,4,5,..}; B-n1: {2,3,6,7,..}
  typedef typename ALoader::Row ARow;
  ARow rowA0a, rowA0b, rowA1a, rowA1b;                        // re-initialised in place when the stream crosses into the next tile
  const uint16_t *rowB0a, *rowB0b, *rowB1a, *rowB1b;
  auto brow = [&](int n, int ch) { return Wt + (size_t)(n < N ? n : N - 1) * K + ch * 8; };
  // (the lane-derived inputs are recomputed from a fresh lane id at every call: nothing of them lives across the K loop)
  auto init_a0 = [&](int mm) {
    const int l = lane_id_fresh(), sr = l >> 3, c0 = (l & 7) ^ ((l >> 4) & 7);
    rowA0a = A.init(mm + rbA0 * 16 + sr, M, c0); rowA0b = A.init(mm + rbA0 * 16 + 8 + sr, M, c0 ^ 4);
  };
  auto init_a1 = [&](int mm) {
    const int l = lane_id_fresh(), sr = l >> 3, c0 = (l & 7) ^ ((l >> 4) & 7);
    rowA1a = A.init(mm + rbA1 * 16 + sr, M, c0); rowA1b = A.init(mm + rbA1 * 16 + 8 + sr, M, c0 ^ 4);
  };
  auto init_b0 = [&](int nn) {
    const int l = lane_id_fresh(), sr = l >> 3, c0 = (l & 7) ^ ((l >> 4) & 7);
    rowB0a = brow(nn + rbB0 * 16 + sr, c0); rowB0b = brow(nn + rbB0 * 16 + 8 + sr, c0 ^ 4);
  };
  auto init_b1 = [&](int nn) {
    const int l = lane_id_fresh(), sr = l >> 3, c0 = (l & 7) ^ ((l >> 4) & 7);
    rowB1a = brow(nn + rbB1 * 16 + sr, c0); rowB1b = brow(nn + rbB1 * 16 + 8 + sr, c0 ^ 4);
  };
  init_a0(m0); init_a1(m0); init_b0(n0); init_b1(n0);
  const int dA0 = rbA0 * 2048, dA1 = rbA1 * 2048, dB0 = G_BOFF + rbB0 * 2048, dB1 = G_BOFF + rbB1 * 2048;  // wave-uniform

  typedef typename ALoader::KIter KIter;
  auto stage_a = [&](const ARow& ra, const ARow& rb, const KIter& ka, const KIter& kb, int dst) {  // ka / kb: the two k halves
    __builtin_amdgcn_global_load_lds((gptr_t)A.src(ra, ka, kb, sch0), (lptr_t)(lds + dst), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)A.src(rb, ka, kb, sch1), (lptr_t)(lds + dst + 1024), 16, 0, 0);
  };
  // K % 64 == 32 (the convolutions: K = 9 * 480): the second half of the last tile is read from the zero page on both sides
  auto stage_b = [&](const uint16_t* ra, const uint16_t* rb, int k0, int dst) {
    const bool tail = k0 + 32 >= K;  // wave-uniform
    const uint16_t* pa = tail && sch0 >= 4 ? zero + (sch0 & 3) * 8 : ra + k0;
    const uint16_t* pb = tail && sch1 >= 4 ? zero + (sch1 & 3) * 8 : rb + k0;
    __builtin_amdgcn_global_load_lds((gptr_t)pa, (lptr_t)(lds + dst), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)pb, (lptr_t)(lds + dst + 1024), 16, 0, 0);
  };

  // ---- fragment read addresses: lane reads row r = lane & 15, logical chunk ks * 4 + (lane >> 4) of a 16-row block ----
  const int fr = lane & 15, ft = (lane >> 4) ^ ((fr >> 1) & 7);
  const int roff0 = fr * 128 + ft * 16, roff1 = fr * 128 + (ft ^ 4) * 16;  // k-step 0 / 1
  const int ra_base = wr * 16384;            // A block i of this wave: + i * 2048
  const int rb_base = G_BOFF + wc * 8192;    // B block j:              + j * 2048

  f32x4_t acc[8][4];
  const int KT = (K + G_BK - 1) / G_BK;  // >= 2 (launcher)

  // ---- prologue of the FIRST tile: K tile 0 complete + the two early units of K tile 1 ----
  KIter kit = A.kbegin();  // A-operand column iterator; after the prologue: column (t + 1) * 64 at the start of K tile t
  {
    const KIter k00 = kit;
    A.knext(kit);
    const KIter k01 = kit;
    A.knext(kit);  // column 64
    KIter k11 = kit;
    A.knext(k11);
    stage_a(rowA0a, rowA0b, k00, k01, dA0);
    stage_b(rowB0a, rowB0b, 0, dB0);
    stage_b(rowB1a, rowB1b, 0, dB1);
    stage_a(rowA1a, rowA1b, k00, k01, dA1);
    stage_a(rowA0a, rowA0b, kit, k11, G_BUF + dA0);
    stage_b(rowB0a, rowB0b, G_BK, G_BUF + dB0);
  }
  Q3A_WAIT_VM(8);  // A-m0(0), B-n0(0) have landed (this wave's part)
  Q3A_BARRIER();
  if (wr == 1) Q3A_BARRIER();  // the second wave group runs one barrier behind the first

  bf16x8_t af[8], b0f[4], b1f[4];
  int par = 0;          // buffer of this tile's K tile 0 (the stream's running K-tile count, mod 2)
  bool seam = false;    // this tile was entered behind an epilogue: its K tile 0 is known to have landed
  int m0n = 0, n0n = 0; // the next tile of the walk

  // one K tile.  N1: the stream has a K tile t+1, N2: a K tile t+2 (compile time -> exact vmcnt values).  NX = 1: K tile t+2 is K tile
  // 0 of the NEXT output tile, NX = 2: K tile t+1 is (the row pointers switch over right in front of their first use for it)
  auto tile = [&](int t, auto n1_tag, auto n2_tag, auto nx_tag, bool landed) {
    constexpr bool N1 = decltype(n1_tag)::value, N2 = decltype(n2_tag)::value;
    constexpr int NX = decltype(nx_tag)::value;
    const int cb = ((t + par) & 1) * G_BUF, nb = G_BUF - cb;  // this tile's buffer / the other one (wave-uniform)
    const int k1 = NX == 2 ? 0 : (t + 1) * G_BK, k2 = NX == 1 ? 0 : (t + 2) * G_BK;
    KIter ka1, kb1, ka2, kb2;  // columns k1, k1 + 32, k2, k2 + 32 of the A operand
    if constexpr (NX == 2) {
      ka1 = A.kbegin_fresh();
      kb1 = ka1; A.knext(kb1);
      kit = kb1; A.knext(kit);  // column 64 of the next output tile: where its K tile 0 finds the iterator
      ka2 = kit; kb2 = kit;     // (unused: N2 is false)
    } else {
      ka1 = kit;
      A.knext(kit);
      kb1 = kit;
      A.knext(kit);             // = column k2: where the next K tile starts
      if constexpr (NX == 1) { ka2 = A.kbegin_fresh(); kb2 = ka2; A.knext(kb2); }
      else { ka2 = kit; kb2 = kit; A.knext(kb2); }
    }
    const bf16x8_t* ap0 = reinterpret_cast<const bf16x8_t*>(lds + cb + ra_base + roff0);
    const bf16x8_t* ap1 = reinterpret_cast<const bf16x8_t*>(lds + cb + ra_base + roff1);
    const bf16x8_t* bp0 = reinterpret_cast<const bf16x8_t*>(lds + cb + rb_base + roff0);
    const bf16x8_t* bp1 = reinterpret_cast<const bf16x8_t*>(lds + cb + rb_base + roff1);
    // ---------------- P1 ----------------
#pragma unroll
    for (int j = 0; j < 2; ++j) { b0f[j * 2] = bp0[j * 128]; b0f[j * 2 + 1] = bp1[j * 128]; }       // B n0: blocks 0,1
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { af[i * 2] = ap0[i * 128]; af[i * 2 + 1] = ap1[i * 128]; }          // A m0: blocks 0..3
    if constexpr (N1) {
      if constexpr (NX == 2) init_b1(n0n);
      stage_b(rowB1a, rowB1b, k1, nb + dB1);
    }
    if (!landed) wait_vm<N1 ? 8 : 2>();                       // B-n1(t) has landed
    __builtin_amdgcn_sched_barrier(0);
    Q3A_BARRIER();
    Q3A_WAIT_LGKM0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i * 2 + ks], b0f[j * 2 + ks], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    Q3A_BARRIER();
    // ---------------- P2 ----------------
#pragma unroll
    for (int j = 0; j < 2; ++j) { b1f[j * 2] = bp0[(2 + j) * 128]; b1f[j * 2 + 1] = bp1[(2 + j) * 128]; }  // B n1: blocks 2,3
    if constexpr (N1) {
      if constexpr (NX == 2) init_a1(m0n);
      stage_a(rowA1a, rowA1b, ka1, kb1, nb + dA1);
    }
    if (!landed) wait_vm<N1 ? 8 : 0>();                       // A-m1(t) has landed
    __builtin_amdgcn_sched_barrier(0);
    Q3A_BARRIER();
    Q3A_WAIT_LGKM0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i * 2 + ks], b1f[j * 2 + ks], acc[i][2 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    Q3A_BARRIER();
    // ---------------- P3 ----------------
#pragma unroll
    for (int i = 0; i < 4; ++i) { af[i * 2] = ap0[(4 + i) * 128]; af[i * 2 + 1] = ap1[(4 + i) * 128]; }  // A m1: blocks 4..7
    if constexpr (N2) {  // A-m0 / B-n0 of this buffer were last read in P1
      if constexpr (NX == 1) init_a0(m0n);
      stage_a(rowA0a, rowA0b, ka2, kb2, cb + dA0);
    }
    __builtin_amdgcn_sched_barrier(0);
    Q3A_BARRIER();
    Q3A_WAIT_LGKM0();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i * 2 + ks], b1f[j * 2 + ks], acc[4 + i][2 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    Q3A_BARRIER();
    // ---------------- P4 ----------------
    if constexpr (N2) {
      if constexpr (NX == 1) init_b0(n0n);
      stage_b(rowB0a, rowB0b, k2, cb + dB0);
    }
    if constexpr (N1) wait_vm<N2 ? 8 : 4>();                  // A-m0(t+1), B-n0(t+1) have landed
    __builtin_amdgcn_sched_barrier(0);
    Q3A_BARRIER();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i * 2 + ks], b0f[j * 2 + ks], acc[4 + i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    Q3A_BARRIER();
  };


  // ---- epilogue: the wave's 128 x 64 tile goes through its private 8 KiB of the staging half (the K-tile buffer the next tile is
  // NOT arriving in) in four 32-row passes, so that global memory sees whole rows: 256 B (fp32) / 128 B (bf16) contiguous per row
  // instead of the 64-B column slices of the MFMA C/D layout (col = lane & 15, row = (lane >> 4) * 4 + reg).  The next tile's K
  // tile 0 is in flight when it starts (see the header) ----
  auto epilogue = [&](const int ybuf) {
    const int lane = lane_id_fresh();  // (shadows the kernel's: the epilogue's lane constants are not kept alive across the K loop)
    const int col_in = lane & 15, row_in = (lane >> 4) * 4;
    const bool wide16 = (N % 8 == 0) && (ep.ldo % 8 == 0);  // bf16 rows that can be written 16 B per lane (kernel-uniform)
    float* stg = reinterpret_cast<float*>(lds + ybuf + wave * 8192);  // [32][64] fp32
    auto stage_pass = [&](int h) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) stg[(i * 16 + row_in + r) * 64 + j * 16 + col_in] = acc[h * 2 + i][j][r];
    };
    if constexpr (!GLU && !ROPE) {
      // fp32 residual output, nothing else in the epilogue (x += X.W^T + b: o / down / fc2 / out projections).  The general store
      // loop below asks for a residual row inside the iteration that needs it -- one exposed round trip per 4 rows, 11.7 us per
      // 64-row pass in the stamped timeline (profiles/r3_phase_probe_after_*.txt: 131 KiB read + 131 KiB written per CU at
      // 22 GB/s) next to 16.6 us of K loop at K = 896.  Here the residual rows are requested back to back two passes ahead: passes
      // 0 and 1 as soon as pass 0's accumulators are staged (their registers take the rows), passes 2 and 3 behind pass 1's, so only
      // pass 0 waits for a round trip, row by row (counted vmcnt).  Addresses = wave-uniform row base + ONE per-lane byte offset.
      // M % 4 == 0: the 4 rows of an iteration are inside or outside the matrix together.  Same arithmetic ((acc + bias) +
      // residual): bit-identical to the general loop (round 4 A/B, knob removed in round 6).
      if (ep.resid != nullptr && ep.out16 == nullptr && ep.rowmap == nullptr && ep.addend == nullptr && ep.act == 0 && (M & 3) == 0) {  // kernel-uniform
        const int n = n0 + wc * 64 + (lane & 15) * 4;
        const unsigned voff = ((unsigned)(lane >> 4) * (unsigned)ep.ldo + (unsigned)(n < N ? n : 0)) * 4u;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ep.bias && n < N) bv = *reinterpret_cast<const float4*>(ep.bias + n);
        float4 rpre[4][8];
        auto request = [&](int h) {
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int rb = m0 + wr * 128 + h * 32 + it * 4;
            const char* base = reinterpret_cast<const char*>(ep.resid) + (size_t)(rb < M ? rb : 0) * ep.ldo * 4;  // outside: a valid row, never stored
            rpre[h][it] = *reinterpret_cast<const float4*>(base + voff);
          }
        };
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          stage_pass(h);
          if (h == 0) {
            // the bias row (requested above, before the staging) must be in registers BEFORE the requests: hipcc sinks its load
            // behind them otherwise, and the first row's wait for it becomes a wait for all of them
            asm volatile("" : "+v"(bv.x), "+v"(bv.y), "+v"(bv.z), "+v"(bv.w));
            Q3A_WAIT_VM(0);  // the next tile's K tile 0 (this wave's part) has landed: nothing else is in flight yet
            request(0);
            request(1);  // 16 loads back to back; vmcnt is counted per row below
          }
          if (h == 1) { request(2); request(3); }
          if (h == 0) Q3A_STAMP_AT(ep.stamp, bid, 3);
          if (h == 3) Q3A_STAMP_AT(ep.stamp, bid, 5);
#pragma unroll
          for (int g = 0; g < 2; ++g) {  // 4 iterations of LDS reads at a time (the fence keeps the others from being hoisted into registers)
#pragma unroll
            for (int it = g * 4; it < g * 4 + 4; ++it) {
              float4 v = *reinterpret_cast<const float4*>(&stg[(it * 4 + (lane >> 4)) * 64 + (lane & 15) * 4]);
              v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
              const float4 b = rpre[h][it];
              v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
              asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));  // unconditional: or hipcc sinks row, adds AND the row's load into the store's branch
              const int rb = m0 + wr * 128 + h * 32 + it * 4;
              char* base = reinterpret_cast<char*>(ep.out) + (size_t)rb * ep.ldo * 4;
              if (rb < M && n < N) *reinterpret_cast<float4*>(base + voff) = v;
            }
            asm volatile("" ::: "memory");
          }
          if (h == 0) Q3A_STAMP_AT(ep.stamp, bid, 4);
          if (h == 3) Q3A_STAMP_AT(ep.stamp, bid, 6);
        }
        return;
      }
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      stage_pass(h);
      const int mrow0 = m0 + wr * 128 + h * 32;
      if (h == 0) Q3A_STAMP_AT(ep.stamp, bid, 3);  // pass 0 staged in LDS (wave 0)
      if (h == 3) Q3A_STAMP_AT(ep.stamp, bid, 5);
      if (h == 0) Q3A_WAIT_VM(0);                  // the next tile's K tile 0 (this wave's part) has landed; nothing else of this epilogue is in flight yet
      if constexpr (ROPE) {
        // A 128-wide head = the 64-column tiles of the wave pair (wc, wc ^ 1): column c of the even wave and column c of the
        // odd wave are the rotate_half partners (dims c, c + 64).  Both halves are staged; after a workgroup barrier a lane
        // reads its own 4 columns and the partner's 4 at the same offset, the 16 lanes of a row reduce the 128 squares with
        // DPP, and q leaves as bf16 [rows][n_q * 128], k / v go to their cache rows (src/layers.rs:303-319,361-375) --
        // the fp32 qkv matrix (16 KiB per token at 0.6B) is never written or re-read.
        // (raw barriers: only the LDS writes have to be visible; a __syncthreads() would also wait for the previous pass's stores)
        Q3A_WAIT_LGKM0();
        Q3A_BARRIER();
        // 8 columns per lane: 8 lanes per row, 8 rows per iteration, ONE 16-B store per lane and iteration; the row's position
        // (and sequence) are requested for all iterations up front so that the cos / sin rows -- whose address depends on
        // them -- are not a second dependent round trip inside every iteration
        const float* pst = reinterpret_cast<const float*>(lds + ybuf + (wave ^ 1) * 8192);
        const int c8 = (lane & 7) * 8;
        const int ncol = n0 + wc * 64;               // first W row of this wave's columns
        const int hv = ncol >> 7;                    // head vector index: q heads, then k heads, then v heads
        const bool second = (wc & 1) != 0;           // this wave holds dims 64..127 of the head
        const int d_own = (second ? 64 : 0) + c8, d_par = (second ? 0 : 64) + c8;
        const bool is_q = hv < rk.n_q, is_k = !is_q && hv < rk.n_q + rk.n_kv;
        const float* nw = is_q ? rk.q_norm : rk.k_norm;
        float w_own[8], w_par[8], b_own[8], b_par[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { w_own[e] = 1.f; w_par[e] = 1.f; b_own[e] = 0.f; b_par[e] = 0.f; }
        if (is_q || is_k) {
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const float4 a = *reinterpret_cast<const float4*>(nw + d_own + h2 * 4), c = *reinterpret_cast<const float4*>(nw + d_par + h2 * 4);
            w_own[h2 * 4] = a.x; w_own[h2 * 4 + 1] = a.y; w_own[h2 * 4 + 2] = a.z; w_own[h2 * 4 + 3] = a.w;
            w_par[h2 * 4] = c.x; w_par[h2 * 4 + 1] = c.y; w_par[h2 * 4 + 2] = c.z; w_par[h2 * 4 + 3] = c.w;
          }
        }
        if (ep.bias && ncol < N) {
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const float4 a = *reinterpret_cast<const float4*>(ep.bias + (hv << 7) + d_own + h2 * 4);
            const float4 c = *reinterpret_cast<const float4*>(ep.bias + (hv << 7) + d_par + h2 * 4);
            b_own[h2 * 4] = a.x; b_own[h2 * 4 + 1] = a.y; b_own[h2 * 4 + 2] = a.z; b_own[h2 * 4 + 3] = a.w;
            b_par[h2 * 4] = c.x; b_par[h2 * 4 + 1] = c.y; b_par[h2 * 4 + 2] = c.z; b_par[h2 * 4 + 3] = c.w;
          }
        }
        int posv[4], seqv[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int m = mrow0 + it * 8 + (lane >> 3), mc = m < M ? m : M - 1;
          posv[it] = rk.row_pos[mc];
          seqv[it] = is_q ? 0 : rk.row_seq[mc];
        }
        // cos / sin rows one iteration ahead (two register sets): inside the iteration that uses them the four loads were an exposed
        // round trip each time (ISA: global_load x4 ... s_waitcnt vmcnt(0) in every iteration)
        const bool roped = is_q || is_k;  // wave-uniform
        float4 rc0[2], rc1[2], rs0[2], rs1[2];
        auto rope_rows = [&](int it) {
          const float* cp = rk.cos_t + (size_t)posv[it] * 64 + c8;
          const float* sp = rk.sin_t + (size_t)posv[it] * 64 + c8;
          rc0[it & 1] = *reinterpret_cast<const float4*>(cp); rc1[it & 1] = *reinterpret_cast<const float4*>(cp + 4);
          rs0[it & 1] = *reinterpret_cast<const float4*>(sp); rs1[it & 1] = *reinterpret_cast<const float4*>(sp + 4);
        };
        // one straight-line loop per kind of wave (q / k: rotated; v: copied): with the requests under a condition hipcc's wait
        // counting falls back to vmcnt(0) in front of every use, i.e. waits for the rows just requested as well
        auto body = [&](auto roped_c, int it) {
          constexpr bool RP = decltype(roped_c)::value;
          const int row = it * 8 + (lane >> 3), m = mrow0 + row;
          const int pos = posv[it];
          float xo[8], xp[8];
          {
            const float4 o0 = *reinterpret_cast<const float4*>(&stg[row * 64 + c8]), o1 = *reinterpret_cast<const float4*>(&stg[row * 64 + c8 + 4]);
            xo[0] = o0.x; xo[1] = o0.y; xo[2] = o0.z; xo[3] = o0.w; xo[4] = o1.x; xo[5] = o1.y; xo[6] = o1.z; xo[7] = o1.w;
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) xo[e] += b_own[e];
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = xo[e];
          if constexpr (RP) {
            {
              const float4 p0 = *reinterpret_cast<const float4*>(&pst[row * 64 + c8]), p1 = *reinterpret_cast<const float4*>(&pst[row * 64 + c8 + 4]);
              xp[0] = p0.x; xp[1] = p0.y; xp[2] = p0.z; xp[3] = p0.w; xp[4] = p1.x; xp[5] = p1.y; xp[6] = p1.z; xp[7] = p1.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) xp[e] += b_par[e];
            float cs[8], sn[8];
            {
              const float4 c0 = rc0[it & 1], c1 = rc1[it & 1], s0 = rs0[it & 1], s1 = rs1[it & 1];
              cs[0] = c0.x; cs[1] = c0.y; cs[2] = c0.z; cs[3] = c0.w; cs[4] = c1.x; cs[5] = c1.y; cs[6] = c1.z; cs[7] = c1.w;
              sn[0] = s0.x; sn[1] = s0.y; sn[2] = s0.z; sn[3] = s0.w; sn[4] = s1.x; sn[5] = s1.y; sn[6] = s1.z; sn[7] = s1.w;
            }
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += xo[e] * xo[e] + xp[e] * xp[e];
            ss = row8_sum(ss);  // the 8 lanes of this row hold all 128 dims between them
            const float rstd = 1.0f / sqrtf(ss / 128.0f + rk.eps);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float no = (xo[e] * rstd) * w_own[e];
              float np = (xp[e] * rstd) * w_par[e];
              if (!second) np = -np;  // rotate_half = cat(-x2, x1)
              v[e] = no * cs[e] + np * sn[e];
            }
          }
          if (m >= M || ncol >= N) return;
          uint4 pk;
          pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]); pk.z = pack_bf16x2(v[4], v[5]); pk.w = pack_bf16x2(v[6], v[7]);
          if (is_q) {
            *reinterpret_cast<uint4*>(rk.q16 + (size_t)m * rk.n_q * 128 + (size_t)hv * 128 + d_own) = pk;
          } else {
            const int kvh = is_k ? hv - rk.n_q : hv - rk.n_q - rk.n_kv;
            uint16_t* c = reinterpret_cast<uint16_t*>(is_k ? rk.kcache : rk.vcache) +
                          (((size_t)seqv[it] * rk.n_kv + kvh) * rk.max_ctx + pos) * 128 + d_own;
            *reinterpret_cast<uint4*>(c) = pk;
          }
        };
        if (roped) {
          rope_rows(0);
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            if (it + 1 < 4) rope_rows(it + 1);
            asm volatile("" ::: "memory");  // the requests stay here: hipcc would sink them to their use in the next iteration
            body(std::true_type{}, it);
          }
        } else {
#pragma unroll
          for (int it = 0; it < 4; ++it) body(std::false_type{}, it);
        }
        Q3A_WAIT_LGKM0();
        Q3A_BARRIER();  // the partner is done with this wave's staged half before the next pass overwrites it
      } else if (!GLU && ep.out16 && wide16) {
        // bf16 output, 8 columns per lane: ONE 16-B store per lane and iteration (8 rows x 128 B per wave instruction).  The
        // stamped timeline (profiles/r3_phase_probe_start.txt) had this phase at 4.7 us per 64-row pass with 8-B stores --
        // 13 GB/s per CU, store-issue bound -- next to 17 us of K loop at K = 896.
        const int c8 = (lane & 7) * 8, n = n0 + wc * 64 + c8;
        float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
        if (ep.bias && n < N) { b0 = *reinterpret_cast<const float4*>(ep.bias + n); b1 = *reinterpret_cast<const float4*>(ep.bias + n + 4); }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = it * 8 + (lane >> 3), m = mrow0 + row;
          float4 v0 = *reinterpret_cast<const float4*>(&stg[row * 64 + c8]);
          float4 v1 = *reinterpret_cast<const float4*>(&stg[row * 64 + c8 + 4]);
          if (m >= M || n >= N) continue;
          const int orow = ep.rowmap ? ep.rowmap[m] : m;
          if (orow < 0) continue;
          v0.x += b0.x; v0.y += b0.y; v0.z += b0.z; v0.w += b0.w;
          v1.x += b1.x; v1.y += b1.y; v1.z += b1.z; v1.w += b1.w;
          if (ep.addend) {
            const float* ad = ep.addend + (size_t)(m % ep.addend_period) * ep.ldo + n;
            const float4 a0 = *reinterpret_cast<const float4*>(ad), a1 = *reinterpret_cast<const float4*>(ad + 4);
            v0.x += a0.x; v0.y += a0.y; v0.z += a0.z; v0.w += a0.w;
            v1.x += a1.x; v1.y += a1.y; v1.z += a1.z; v1.w += a1.w;
          }
          if (ep.act == 1) {  // packed: every multiply / fma of the polynomial serves two values
            const f32x2_t g0 = gelu_fast2(f32x2_t{v0.x, v0.y}), g1 = gelu_fast2(f32x2_t{v0.z, v0.w});
            const f32x2_t g2 = gelu_fast2(f32x2_t{v1.x, v1.y}), g3 = gelu_fast2(f32x2_t{v1.z, v1.w});
            v0 = make_float4(g0.x, g0.y, g1.x, g1.y);
            v1 = make_float4(g2.x, g2.y, g3.x, g3.y);
          }
          if (ep.resid) {
            const float* rs = ep.resid + (size_t)orow * ep.ldo + n;
            const float4 r0 = *reinterpret_cast<const float4*>(rs), r1 = *reinterpret_cast<const float4*>(rs + 4);
            v0.x += r0.x; v0.y += r0.y; v0.z += r0.z; v0.w += r0.w;
            v1.x += r1.x; v1.y += r1.y; v1.z += r1.z; v1.w += r1.w;
          }
          uint4 pk;
          pk.x = pack_bf16x2(v0.x, v0.y); pk.y = pack_bf16x2(v0.z, v0.w);
          pk.z = pack_bf16x2(v1.x, v1.y); pk.w = pack_bf16x2(v1.z, v1.w);
          *reinterpret_cast<uint4*>(ep.out16 + (size_t)orow * ep.ldo + n) = pk;
        }
      } else if (!GLU) {
        const int c4 = (lane & 15) * 4, n = n0 + wc * 64 + c4;
#pragma unroll 4
        for (int it = 0; it < 8; ++it) {
          const int row = it * 4 + (lane >> 4), m = mrow0 + row;
          float4 v = *reinterpret_cast<const float4*>(&stg[row * 64 + c4]);
          if (m >= M || n >= N) continue;
          const int orow = ep.rowmap ? ep.rowmap[m] : m;
          if (orow < 0) continue;
          if (ep.bias) {
            const float4 b = *reinterpret_cast<const float4*>(ep.bias + n);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
          }
          if (ep.addend) {
            const float4 b = *reinterpret_cast<const float4*>(ep.addend + (size_t)(m % ep.addend_period) * ep.ldo + n);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
          }
          if (ep.act == 1) { v.x = gelu_fast(v.x); v.y = gelu_fast(v.y); v.z = gelu_fast(v.z); v.w = gelu_fast(v.w); }
          if (ep.resid) {
            const float4 b = *reinterpret_cast<const float4*>(ep.resid + (size_t)orow * ep.ldo + n);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
          }
          if (ep.out16) {
            uint2 pk;
            pk.x = pack_bf16x2(v.x, v.y);
            pk.y = pack_bf16x2(v.z, v.w);
            *reinterpret_cast<uint2*>(ep.out16 + (size_t)orow * ep.ldo + n) = pk;
          } else {
            *reinterpret_cast<float4*>(ep.out + (size_t)orow * ep.ldo + n) = v;
          }
        }
      } else if (ep.out16 && wide16) {
        // GLU, bf16 output, 8 output columns per lane (one 16-B store): W rows are [16 gate | 16 up] blocks, so the staged
        // columns [0,16) / [16,32) are gate / up of output columns 0..15 and [32,64) likewise
        const int oc8 = (lane & 3) * 8, gc = (oc8 >> 4) * 32 + (oc8 & 15);
        const int nb = n0 + wc * 64 + gc;            // W row of the first gate value
        const int on = ((n0 + wc * 64) >> 1) + oc8;  // output column
        float4 bg0 = make_float4(0.f, 0.f, 0.f, 0.f), bg1 = bg0, bu0 = bg0, bu1 = bg0;
        if (ep.bias && nb + 16 < N) {
          bg0 = *reinterpret_cast<const float4*>(ep.bias + nb); bg1 = *reinterpret_cast<const float4*>(ep.bias + nb + 4);
          bu0 = *reinterpret_cast<const float4*>(ep.bias + nb + 16); bu1 = *reinterpret_cast<const float4*>(ep.bias + nb + 20);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 16 + (lane >> 2), m = mrow0 + row;
          float4 g0 = *reinterpret_cast<const float4*>(&stg[row * 64 + gc]), g1 = *reinterpret_cast<const float4*>(&stg[row * 64 + gc + 4]);
          float4 u0 = *reinterpret_cast<const float4*>(&stg[row * 64 + gc + 16]), u1 = *reinterpret_cast<const float4*>(&stg[row * 64 + gc + 20]);
          if (m >= M || nb + 16 >= N) continue;
          const int orow = ep.rowmap ? ep.rowmap[m] : m;
          if (orow < 0) continue;
          g0.x += bg0.x; g0.y += bg0.y; g0.z += bg0.z; g0.w += bg0.w; g1.x += bg1.x; g1.y += bg1.y; g1.z += bg1.z; g1.w += bg1.w;
          u0.x += bu0.x; u0.y += bu0.y; u0.z += bu0.z; u0.w += bu0.w; u1.x += bu1.x; u1.y += bu1.y; u1.z += bu1.z; u1.w += bu1.w;
          uint4 pk;
          pk.x = pack_bf16x2(silu_fast(g0.x) * u0.x, silu_fast(g0.y) * u0.y); pk.y = pack_bf16x2(silu_fast(g0.z) * u0.z, silu_fast(g0.w) * u0.w);
          pk.z = pack_bf16x2(silu_fast(g1.x) * u1.x, silu_fast(g1.y) * u1.y); pk.w = pack_bf16x2(silu_fast(g1.z) * u1.z, silu_fast(g1.w) * u1.w);
          *reinterpret_cast<uint4*>(ep.out16 + (size_t)orow * ep.ldo + on) = pk;
        }
      } else {
        // W rows are [16 gate | 16 up] blocks: staged columns [0,16) gate / [16,32) up of output columns 0..15, [32,64) likewise
        const int oc4 = (lane & 7) * 4, gc = (oc4 >> 4) * 32 + (oc4 & 15);
        const int nb = n0 + wc * 64 + gc;          // W row of the gate value
        const int on = ((n0 + wc * 64) >> 1) + oc4;  // output column
#pragma unroll 4
        for (int it = 0; it < 4; ++it) {
          const int row = it * 8 + (lane >> 3), m = mrow0 + row;
          float4 g = *reinterpret_cast<const float4*>(&stg[row * 64 + gc]);
          float4 u = *reinterpret_cast<const float4*>(&stg[row * 64 + gc + 16]);
          if (m >= M || nb + 16 >= N) continue;
          const int orow = ep.rowmap ? ep.rowmap[m] : m;
          if (orow < 0) continue;
          if (ep.bias) {
            const float4 bg = *reinterpret_cast<const float4*>(ep.bias + nb), bu = *reinterpret_cast<const float4*>(ep.bias + nb + 16);
            g.x += bg.x; g.y += bg.y; g.z += bg.z; g.w += bg.w;
            u.x += bu.x; u.y += bu.y; u.z += bu.z; u.w += bu.w;
          }
          const float4 v = make_float4(silu_fast(g.x) * u.x, silu_fast(g.y) * u.y, silu_fast(g.z) * u.z, silu_fast(g.w) * u.w);
          if (ep.out16) {
            uint2 pk;
            pk.x = pack_bf16x2(v.x, v.y);
            pk.y = pack_bf16x2(v.z, v.w);
            *reinterpret_cast<uint2*>(ep.out16 + (size_t)orow * ep.ldo + on) = pk;
          } else {
            *reinterpret_cast<float4*>(ep.out + (size_t)orow * ep.ldo + on) = v;
          }
        }
      }
      if (h == 0) Q3A_STAMP_AT(ep.stamp, bid, 4);  // pass 0 stored (wave 0)
      if (h == 3) Q3A_STAMP_AT(ep.stamp, bid, 6);  // last pass stored
    }
  };

  // ---- the walk ----
  // ONE K-loop shape for every tile (three instances of the K tile instead of five, no branch between two tails: with both shapes in
  // the loop hipcc's allocator copied the accumulators around and spilled 250 registers): the last tile of a walk prefetches too -- its
  // own K tile 0 again, 64 KiB that nobody reads -- and drains it like any other.
  for (;;) {
    const int tjn = tj + wpx;
    const bool has_next = tjn < x_cnt;  // workgroup-uniform
    tile_origin(has_next ? x_first + tjn : bid, m0n, n0n);
    Q3A_STAMP_AT(ep.stamp, bid, 1);  // first staging units landed: the K loop starts
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    {
      int t = 0;
      for (; t + 2 < KT; ++t) tile(t, std::true_type{}, std::true_type{}, std::integral_constant<int, 0>{}, seam && t == 0);
      tile(t, std::true_type{}, std::true_type{}, std::integral_constant<int, 1>{}, seam && t == 0);
      tile(t + 1, std::true_type{}, std::false_type{}, std::integral_constant<int, 2>{}, false);
    }
    if (wr == 0) Q3A_BARRIER();  // balance the extra barrier of the second group: every wave is past its last LDS read
    Q3A_STAMP_AT(ep.stamp, bid, 2);  // K loop done
    const int ybuf = ((KT - 1 + par) & 1) * G_BUF;  // the last K tile's buffer: free now; the next tile's K tile 0 arrives in the other one
    epilogue(ybuf);
    if (!has_next) break;
    // ---- seam: the next tile's K tile 0 is complete in the other buffer (every wave waited for its own part in front of its first
    // epilogue request); behind the barrier the staging half is free and takes the early units of K tile 1 ----
    tj = tjn; bid = x_first + tj; m0 = m0n; n0 = n0n;
    par = (par + KT) & 1;
    seam = true;
    Q3A_STAMP_AT(ep.stamp, bid, 0);
    Q3A_WAIT_LGKM0();
    Q3A_BARRIER();
    {
      KIter kb = kit;  // kit = column 64
      A.knext(kb);
      stage_a(rowA0a, rowA0b, kit, kb, ybuf + dA0);
      stage_b(rowB0a, rowB0b, G_BK, ybuf + dB0);
    }
    if (wr == 1) Q3A_BARRIER();  // the second wave group runs one barrier behind the first
  }
}

// compute units of the device the calling thread launches on (one workgroup of this kernel per CU: 128 KiB of LDS)
static int device_cus() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int v = cache[dev].load(std::memory_order_relaxed);
  if (v == 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cache[dev].store(v, std::memory_order_relaxed);
  }
  return v;
}

template <bool GLU, class ALoader, bool ROPE = false>
void launch256(const ALoader& A, const uint16_t* W, const uint16_t* zero, int M, int N, int K, const GemmEpilogue& ep, hipStream_t s,
               const RopeKvArgs& rk = RopeKvArgs{}) {
  const int tiles = ((M + G_BM - 1) / G_BM) * ((N + G_BN - 1) / G_BN);
  // gemm256_persist: 0 = one workgroup per tile, 1 = walk with one workgroup per CU, n > 1 = walk with exactly n workgroups (tests:
  // grids that are not a multiple of 8, fewer workgroups than XCDs, one workgroup walking everything)
  const int pk = knobs().gemm256_persist.load(std::memory_order_relaxed);
  const int cus = pk > 1 ? pk : device_cus();
  const int grid = pk != 0 && tiles > cus ? cus : tiles;
  // tile order (kernel: tile_origin): groups of 8 tile rows where the matrix is at least 8 tiles wide (qkv / fc1 / gate-up: the 32 tiles
  // of an XCD then share 8 A + 4 W panels instead of 2 + all), N fastest where it is narrow (the convolutions: 2 column tiles, the
  // 196 / 204-tile residual shapes: 4 -- measured: grouping them raises their memory-side traffic by 25-50 %).  Knob: 0 = this rule.
  const int gk = knobs().gemm256_group_m.load(std::memory_order_relaxed);
  const int gm = gk > 0 ? gk : ((N + G_BN - 1) / G_BN >= 8 ? 8 : 1);
  hipLaunchKernelGGL((gemm256_kernel<GLU, ALoader, ROPE>), dim3(grid), dim3(512), 0, s, A, W, zero, M, N, K, ep, rk, gm);
}

}  // namespace

// A/B knob: minimum number of 256 x 256 tiles for the dispatch to this kernel (0 = whenever the shape allows, huge = never);
// environment Q3A_GEMM256_MIN_TILES at start-up, q3a_debug_set("gemm256_min_tiles", v) afterwards (kernels.h Knobs)
bool gemm256_eligible(int M, int N, int K) {
  if (K % 32 != 0 || K < 2 * G_BK || N % 4 != 0) return false;
  // a few rows against a wide matrix (the decode-step lm_head at 5..64 sequences) is a weight stream, not MFMA work:
  // the 32-row tiles of the small kernel read it at 5.3 TB/s, a 256-row tile at 4.4 (58.8 vs 70 us for 32 x 151936)
  if (M < 128) return false;
  const long tiles = (long)((M + G_BM - 1) / G_BM) * ((N + G_BN - 1) / G_BN);
  return tiles >= knobs().gemm256_min_tiles.load(std::memory_order_relaxed);
}

// Rows [0, M1) that fill whole rounds of 256 tiles when the last round would hold at most `rem_max` tiles (0: no split).
static int split_rows(int M, int N) {
  static const int rem_max = [] { const char* e = getenv("Q3A_GEMM256_SPLIT_REM"); return e ? atoi(e) : 96; }();  // A/B knob (0 = off)
  const int tiles_m = (M + G_BM - 1) / G_BM, tiles_n = (N + G_BN - 1) / G_BN;
  const long tiles = (long)tiles_m * tiles_n;
  const int rem = (int)(tiles % 256);
  if (rem_max <= 0 || tiles <= 256 || rem == 0 || rem > rem_max) return 0;
  const int rows_m = (int)((tiles - rem) / tiles_n);  // whole tile rows inside the full rounds
  const int M1 = rows_m * G_BM;
  return (rows_m >= 1 && M1 < M) ? M1 : 0;
}

int gemm256_split_rows(int M, int N) { return split_rows(M, N); }

const char* launch_gemm256(const uint16_t* X, int lda, const uint16_t* W, int M, int N, int K, const GemmEpilogue& ep,
                           bool glu, hipStream_t s) {
  if (M <= 0) return nullptr;
  if (K % G_BK != 0 || K < 2 * G_BK || lda % 8 != 0) return "gemm256: K must be a multiple of 64 (>= 128) and lda of 8";
  if (glu && N % 32 != 0) return "gemm256: GLU needs N % 32 == 0";
  if (N % 4 != 0 || ep.ldo % 4 != 0) return "gemm256: N and ldo must be multiples of 4 (vector epilogue)";
  // Wave quantisation: one workgroup per CU, so T tiles take ceil(T / 256) rounds.  When the last round would be mostly
  // empty (encoder qkv at 32 clips: 539 tiles = 2 rounds + 27 tiles), the leading tile rows that fill whole rounds run here
  // and the remaining rows go to the small-tile kernel of k_gemm16.hip (its own launch, many cheap tiles).
  if (!ep.addend) {
    const int M1 = split_rows(M, N);
    if (M1 > 0) {
      DenseA256 A1{X, lda};
      if (glu) launch256<true>(A1, W, nullptr, M1, N, K, ep, s); else launch256<false>(A1, W, nullptr, M1, N, K, ep, s);
      GemmEpilogue e2 = ep;
      if (ep.rowmap) e2.rowmap = ep.rowmap + M1;  // the map is indexed by GEMM row; outputs stay absolute
      else {
        const size_t off = (size_t)M1 * ep.ldo;
        if (e2.out) e2.out += off;
        if (e2.out16) e2.out16 += off;
        if (e2.resid) e2.resid += off;
      }
      return launch_gemm16_small(X + (size_t)M1 * lda, lda, W, M - M1, N, K, e2, glu, s);
    }
  }
  DenseA256 A{X, lda};
  if (glu) launch256<true>(A, W, nullptr, M, N, K, ep, s); else launch256<false>(A, W, nullptr, M, N, K, ep, s);
  return nullptr;
}

const char* launch_gemm256_qkrope(const uint16_t* X, int lda, const uint16_t* W, int M, int K, const float* bias,
                                  const RopeKvArgs& rk, hipStream_t s) {
  if (M <= 0) return nullptr;
  const int N = (rk.n_q + 2 * rk.n_kv) * 128;
  if (K % G_BK != 0 || K < 2 * G_BK || lda % 8 != 0) return "gemm256 qkrope: K must be a multiple of 64 (>= 128) and lda of 8";
  if (!rk.q16 || !rk.kcache || !rk.vcache || !rk.row_pos || !rk.row_seq) return "gemm256 qkrope: bf16 q / KV cache / row tables required";
  GemmEpilogue ep;
  ep.bias = bias;
  DenseA256 A{X, lda};
  const int M1 = rk.qkv ? split_rows(M, N) : 0;
  if (M1 > 0) {  // wave quantisation as in launch_gemm256: the trailing rows take the small-tile GEMM + the separate kernel
    launch256<false, DenseA256, true>(A, W, nullptr, M1, N, K, ep, s, rk);
    GemmEpilogue e2;
    e2.out = rk.qkv; e2.ldo = N; e2.bias = bias;  // rk.qkv: fp32 scratch [>= M - M1][N]
    if (const char* err = launch_gemm16_small(X + (size_t)M1 * lda, lda, W, M - M1, N, K, e2, false, s)) return err;
    RopeKvArgs r2 = rk;
    r2.row_seq += M1; r2.row_pos += M1; r2.q16 += (size_t)M1 * rk.n_q * 128;
    return launch_qknorm_rope_kv(r2, M - M1, false, s);
  }
  launch256<false, DenseA256, true>(A, W, nullptr, M, N, K, ep, s, rk);
  return nullptr;
}

const char* launch_conv3x3s2_gemm256(const uint16_t* X, const uint16_t* zero_page, int imgs, int H, int Wd, int C,
                                     const uint16_t* Wt, int Cout, const GemmEpilogue& ep, hipStream_t s) {
  if (C % 32 != 0 || 9 * C < 2 * G_BK) return "conv gemm256: C must be a multiple of 32";
  if (Cout % 4 != 0 || ep.ldo % 4 != 0) return "conv gemm256: Cout and ldo must be multiples of 4 (vector epilogue)";
  ConvA256 A{X, zero_page, H, Wd, C, (H - 1) / 2 + 1, (Wd - 1) / 2 + 1};
  const int M = imgs * A.OH * A.OW;
  if (M <= 0) return nullptr;
  launch256<false>(A, Wt, zero_page, M, Cout, 9 * C, ep, s);
  return nullptr;
}

}  // namespace q3a
