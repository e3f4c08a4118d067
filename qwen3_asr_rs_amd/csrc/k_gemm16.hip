// bf16-in MFMA GEMM for gfx950 (default mode):  Y[M][N] = act( X[M][K] . W[N][K]^T + bias ) (+ residual).
//
// Same contract and epilogues as k_gemm.hip, but the activations arrive already rounded to bf16 (the producing
// kernels store bf16 in the default mode -- the very values k_gemm.hip would round to while staging, so results
// are bit-identical), which lets both operands take the direct HBM -> LDS path:
//   * global_load_lds (16 B per lane, 1 KiB per wave instruction): no VGPR round trip, no conversion VALU work;
//   * the LDS image of a tile is lane-linear, so the bank-conflict swizzle is applied to the SOURCE address:
//     LDS row r keeps its 16-B chunk c at position c ^ s(r) -- a permutation inside one 128-B (BK = 64) or
//     64-B (BK = 32) row, i.e. inside one coalesced global segment -- and the fragment reads apply the same XOR.
//     s(r) is chosen so that the 16 rows a ds_read_b128 lane group touches fall on 16 distinct 16-B slots of the
//     256-B bank row: BK = 64 -> two rows per bank row, s(r) = (r >> 1) & 7;  BK = 32 -> four, s(r) = (r >> 2) & 3;
//   * BM x BN tile (128x128 for big problems, 64x64 / 32x64 when the problem yields too few workgroups), 4 waves
//     (2 x 2), each wave (BM/32) x (BN/32) fragments of v_mfma_f32_16x16x32_bf16;
//   * K loop, two LDS buffers: one barrier per K tile (its workgroup release carries the vmcnt(0) that lands the
//     LDS-DMA issued one iteration earlier), then the next tile's loads are issued and fly under this tile's MFMAs;
//   * (always since round 6; rounds 3-5 kept the two-buffer loop above selectable by the knob gemm16_ring) NS = 3 / 4 (64x64 / 32x64 tiles of dense operands, launches of
//     up to 768 / 512 workgroups): a ring of LDS stages with two / three K tiles in flight and a COUNTED vmcnt in front of a raw s_barrier.  With two stages a K step costs
//     one full L2 round trip for eight MFMAs per wave: on the one-clip shapes (M ~ 400: 300-700 workgroups, 14-16 K steps)
//     the K loop took 9-10.5 us per workgroup, 4.9-5.9 us through the ring (profiles/r3_phase_probe_one_clip_gemms.txt);
//   * the W fragment is the FIRST MFMA operand: the accumulator holds the transposed tile, a lane owns four consecutive
//     output columns, and the epilogue requests all its operands (bias, addend, residual) before the first use;
//   * the im2col view of the 3x3/s2/p1 convolutions reads padded taps from a zero page.
// Measured (MI355X, random data): 460-590 TFLOP/s on the batch-32 encoder/prefill shapes, 637 at 8192^3,
// against 340-380 / 456 for the fp32-activation kernel of k_gemm.hip.
#include <stdlib.h>

#include <type_traits>

#include "dev.h"
#include "kernels.h"

namespace q3a {
namespace {

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

struct DenseA16 {
  const uint16_t* x;
  int lda;
  __device__ __forceinline__ void init_row(int m, int M, int& s0, int& s1, int& s2) const {
    s0 = m < M ? m : M - 1;  // tail rows are clamped: their products are never stored
    s1 = s2 = 0;
  }
  __device__ __forceinline__ const uint16_t* row_ptr(int s0, int, int, int k0) const { return x + (size_t)s0 * lda + k0; }
};

// im2col view of an NHWC bf16 feature map [img][H][W][C] for a 3x3 stride-2 pad-1 convolution
struct ConvA16 {
  const uint16_t* x;
  const uint16_t* zero;  // >= 128 B of zeros for padded taps
  int H, W, C, OH, OW;
  __device__ __forceinline__ void init_row(int m, int M, int& img, int& oh, int& ow) const {
    if (m >= M) m = M - 1;
    img = m / (OH * OW);
    const int r = m - img * OH * OW;
    oh = r / OW;
    ow = r - oh * OW;
  }
  __device__ __forceinline__ const uint16_t* row_ptr(int img, int oh, int ow, int k0) const {
    const int tap = k0 / C, c0 = k0 - tap * C;
    const int kh = tap / 3, kw = tap - kh * 3;
    const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
    if (ih < 0 || ih >= H || iw < 0 || iw >= W) return zero;
    return x + (((size_t)img * H + ih) * W + iw) * C + c0;
  }
};

template <int BK> __device__ __forceinline__ int swz(int row) { return BK == 64 ? (row >> 1) & 7 : (row >> 2) & 3; }

template <int N> __device__ __forceinline__ void g16_wait_vm() {  // N = (A_LOADS + W_LOADS) x tiles in flight
  static_assert(N >= 0 && N < 64, "vmcnt value");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BM, int BN, int BK, bool GLU, class ALoader, int NS = 2>
__global__ __launch_bounds__(256) void gemm16_kernel(ALoader A, const uint16_t* __restrict__ Wt, int M, int N, int K,
                                                     GemmEpilogue ep) {
  constexpr int CPR = BK / 8;                   // 16-B chunks per tile row
  constexpr int RPI = 64 / CPR;                 // tile rows covered by one wave instruction (8 / 16)
  constexpr int A_LOADS = BM / (4 * RPI) > 0 ? BM / (4 * RPI) : 1;  // glds instructions per thread per K tile
  constexpr int W_LOADS = BN / (4 * RPI) > 0 ? BN / (4 * RPI) : 1;
  constexpr bool A_PART = BM < 4 * RPI, W_PART = BN < 4 * RPI;      // tile smaller than one 4-wave sweep
  constexpr int MI = BM / 32, NI = BN / 32;
  static_assert(MI >= 1 && NI >= 1 && (!GLU || NI >= 2), "tile shape");
  static_assert(NS == 2 || ((NS == 3 || NS == 4) && !A_PART && !W_PART), "stage count");
  constexpr int STAGE = (BM + BN) * BK;  // elements per stage: A tile, then W tile
  __shared__ __attribute__((aligned(16))) uint16_t lds[NS * STAGE];  // the ONLY LDS object: NS stages, tile kt in stage kt % NS
  Q3A_STAMP_AT(ep.stamp, blockIdx.x, 0);  // entry

  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {  // bijective XCD remap: consecutive tile ids (sharing the W panel) stay on one XCD / L2
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = bid % tiles_m, tn = bid / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // ---- staging assignment: instruction i of wave w fills tile rows (i*4 + w)*RPI .. +RPI ----
  const int l_row = lane / CPR, l_chunk = lane % CPR;
  const bool a_on = !A_PART || wave * RPI < BM, w_on = !W_PART || wave * RPI < BN;  // wave-uniform
  int a_s0[A_LOADS], a_s1[A_LOADS], a_s2[A_LOADS], a_src_chunk[A_LOADS];
  const uint16_t* w_src[W_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int row = (i * 4 + wave) * RPI + l_row;
    a_src_chunk[i] = l_chunk ^ swz<BK>(row);  // LDS position l_chunk of this row holds global chunk a_src_chunk
    A.init_row(m0 + row, M, a_s0[i], a_s1[i], a_s2[i]);
  }
#pragma unroll
  for (int i = 0; i < W_LOADS; ++i) {
    const int row = (i * 4 + wave) * RPI + l_row;
    const int n = n0 + row;
    w_src[i] = Wt + (size_t)(n < N ? n : N - 1) * K + (l_chunk ^ swz<BK>(row)) * 8;
  }

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- operands of the epilogue (bias, positional addend, residual): every one of them is requested at a clamped
  // (always valid) address before anything uses it -- with small tiles and no row map already HERE, ahead of the first K
  // tile, so that they arrive under the K loop (they are older than every LDS-DMA below: the counted waits of the ring,
  // "at most n NEWER loads outstanding", still mean what they say); otherwise at the top of the epilogue.
  const int m_in = lane & 15, n_in = (lane >> 4) * 4;  // accumulator layout, see the MFMA call below
  const bool vec = N % 4 == 0 && ep.ldo % 4 == 0;      // uniform; 4 | n and 4 | ldo: every vector access is aligned
  int mrow[MI], orow[MI], ncl[NI];
  float4 bb[NI], ad[MI][NI], rs[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    mrow[i] = m0 + wr * (BM / 2) + i * 16 + m_in;
    orow[i] = mrow[i] < M ? mrow[i] : M - 1;
  }
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + wc * (BN / 2) + j * 16 + n_in;
    ncl[j] = n < N ? n : 0;
  }
  auto load_operands = [&]() {
    if (ep.bias) {
#pragma unroll
      for (int j = 0; j < NI; ++j) bb[j] = *reinterpret_cast<const float4*>(ep.bias + ncl[j]);
    }
    if (ep.addend) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int mc = mrow[i] < M ? mrow[i] : M - 1;
#pragma unroll
        for (int j = 0; j < NI; ++j) ad[i][j] = *reinterpret_cast<const float4*>(ep.addend + (size_t)(mc % ep.addend_period) * ep.ldo + ncl[j]);
      }
    }
    if (ep.resid) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int oc = orow[i] < 0 ? 0 : orow[i];
#pragma unroll
        for (int j = 0; j < NI; ++j) rs[i][j] = *reinterpret_cast<const float4*>(ep.resid + (size_t)oc * ep.ldo + ncl[j]);
      }
    }
  };
  const bool pre = !GLU && MI * NI <= 4 && vec && ep.rowmap == nullptr;  // uniform
  if (pre) load_operands();

  const int frag_row = lane & 15, frag_kc = lane >> 4;
  const int KT = K / BK;
  auto issue_tile = [&](int kt, int buf) {
    const int k0 = kt * BK;
    if (a_on) {
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) {
        const int base = (i * 4 + wave) * RPI * BK;  // wave-uniform LDS element offset of this instruction's 1 KiB
        const uint16_t* ap = A.row_ptr(a_s0[i], a_s1[i], a_s2[i], k0) + a_src_chunk[i] * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)ap, (lptr_t)(lds + buf * STAGE + base), 16, 0, 0);
      }
    }
    if (w_on) {
#pragma unroll
      for (int i = 0; i < W_LOADS; ++i) {
        const int base = (i * 4 + wave) * RPI * BK;
        __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + k0), (lptr_t)(lds + buf * STAGE + BM * BK + base), 16, 0, 0);
      }
    }
  };
  if constexpr (NS == 2) {
    issue_tile(0, 0);
  } else {
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
      if (t < KT) issue_tile(t, t);
  }
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = NS == 2 ? (kt & 1) : (kt % NS);
    if constexpr (NS == 2) {
      // one barrier per K tile: its workgroup release carries the vmcnt(0) that lands tile kt, and it orders every
      // wave's fragment reads of tile kt-1 before that buffer is refilled below
      __syncthreads();
      if (kt + 1 < KT) issue_tile(kt + 1, buf ^ 1);
    } else {
      // tiles kt .. kt+NS-2 are in flight (fewer at the tail); the loads retire in order, so "at most as many
      // outstanding as the later tiles issued" means tile kt has landed -- this wave's part; the raw barrier (a
      // __syncthreads() would drain the queue) covers the other waves' parts and orders everyone's fragment reads of
      // tile kt-1 before its stage is refilled with tile kt+NS-1
      constexpr int L = A_LOADS + W_LOADS;
      const int later = KT - 1 - kt;  // uniform
      if (NS == 4 && later >= 2) g16_wait_vm<2 * L>();
      else if (later >= 1) g16_wait_vm<L>();
      else g16_wait_vm<0>();
      asm volatile("s_barrier" ::: "memory");
      if (kt + NS - 1 < KT) issue_tile(kt + NS - 1, (kt + NS - 1) % NS);
    }
    if (kt == 0) Q3A_STAMP_AT(ep.stamp, blockIdx.x, 1);  // first K tile landed
    const uint16_t* as = lds + buf * STAGE;
    const uint16_t* ws = as + BM * BK;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8_t bfrag[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = wc * (BN / 2) + j * 16 + frag_row;
        bfrag[j] = *reinterpret_cast<const bf16x8_t*>(&ws[row * BK + (((ks * 4 + frag_kc) ^ swz<BK>(row)) * 8)]);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wr * (BM / 2) + i * 16 + frag_row;
        const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(&as[row * BK + (((ks * 4 + frag_kc) ^ swz<BK>(row)) * 8)]);
#pragma unroll
        // W fragment as the first operand: the accumulator holds the TRANSPOSED tile (rows = n, columns = m), i.e. a lane
        // ends up with four CONSECUTIVE output columns of one output row -> 8-B (bf16) / 16-B (fp32) stores below
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfrag[j], af, acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue (C/D layout of v_mfma_f32_16x16x32: col = lane&15 -> m, row = (lane>>4)*4 + reg -> n) ----
  Q3A_STAMP_AT(ep.stamp, blockIdx.x, 2);  // K loop done
  // Two things cost 2.3-4.7 us per workgroup on the one-clip shapes in the untransposed form of this epilogue
  // (profiles/r3_phase_probe_one_clip_gemms.txt): 16 two- or four-byte stores per lane and fragment row, and -- more -- the
  // operand loads: `if (ep.bias) v += bias[n]` per fragment makes a chain of dependent L2 round trips (rowmap -> bias ->
  // addend -> residual, each waited for where it is used).  Here every operand of every fragment is requested first, at a
  // clamped (always valid) address, and the arithmetic starts after ONE wait.
  if (ep.rowmap) {
#pragma unroll
    for (int i = 0; i < MI; ++i) orow[i] = ep.rowmap[orow[i]];  // orow held the clamped row
  }
  if (!GLU && vec) {
    if (!pre) load_operands();
    constexpr bool ARGMAX = BM == 32 && MI == 1;  // only the M <= 32 tile carries the code
    float bestv = -INFINITY;
    int besti = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (mrow[i] >= M || orow[i] < 0) continue;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = n0 + wc * (BN / 2) + j * 16 + n_in;
        if (n >= N) continue;
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        if (ep.bias) { v[0] += bb[j].x; v[1] += bb[j].y; v[2] += bb[j].z; v[3] += bb[j].w; }
        if (ep.addend) { v[0] += ad[i][j].x; v[1] += ad[i][j].y; v[2] += ad[i][j].z; v[3] += ad[i][j].w; }
        if (ep.act == 1) {  // default mode: the result is rounded to bf16 (dev.h)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_fast(v[r]);
        }
        if (ep.resid) { v[0] += rs[i][j].x; v[1] += rs[i][j].y; v[2] += rs[i][j].z; v[3] += rs[i][j].w; }
        if (ARGMAX && ep.part_val) {  // columns ascend with j and r: a strict > keeps the first index on ties
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (v[r] > bestv) { bestv = v[r]; besti = n + r; }
        }
        if (ep.out16) {
          uint2 pk;
          pk.x = pack_bf16x2(v[0], v[1]);
          pk.y = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<uint2*>(ep.out16 + (size_t)orow[i] * ep.ldo + n) = pk;
        } else if (ep.out) {
          *reinterpret_cast<float4*>(ep.out + (size_t)orow[i] * ep.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
    if constexpr (ARGMAX) {
      if (ep.part_val) {  // uniform
        // row m = lane & 15 of this wave's 16 rows: its four column groups sit in lanes l, l+16, l+32, l+48; then the two
        // column halves (waves wc = 0 / 1) meet in LDS (every wave has left the K loop: barrier first, the stages are free)
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
          const float ov = __shfl_xor(bestv, o, 64);
          const int oi = __shfl_xor(besti, o, 64);
          if (ov > bestv || (ov == bestv && oi < besti)) { bestv = ov; besti = oi; }
        }
        __syncthreads();
        float* pv = reinterpret_cast<float*>(lds);        // [wc][32 rows]
        int* pi = reinterpret_cast<int*>(lds) + 64;
        if (lane < 16) { pv[wc * 32 + wr * 16 + lane] = bestv; pi[wc * 32 + wr * 16 + lane] = besti; }
        __syncthreads();
        if (tid < 32 && m0 + tid < M) {
          float v = pv[tid];
          int ix = pi[tid];
          if (pv[32 + tid] > v || (pv[32 + tid] == v && pi[32 + tid] < ix)) { v = pv[32 + tid]; ix = pi[32 + tid]; }
          ep.part_val[(size_t)(m0 + tid) * ep.part_stride + tn] = v;
          ep.part_idx[(size_t)(m0 + tid) * ep.part_stride + tn] = ix;
        }
      }
    }
  } else if (!GLU) {  // N or ldo not a multiple of 4: element by element
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (mrow[i] >= M || orow[i] < 0) continue;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = n0 + wc * (BN / 2) + j * 16 + n_in;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r >= N) continue;
          float x = acc[i][j][r];
          if (ep.bias) x += ep.bias[n + r];
          if (ep.addend) x += ep.addend[(size_t)(mrow[i] % ep.addend_period) * ep.ldo + n + r];
          if (ep.act == 1) x = gelu_fast(x);
          if (ep.resid) x += ep.resid[(size_t)orow[i] * ep.ldo + n + r];
          if (ep.out16) ep.out16[(size_t)orow[i] * ep.ldo + n + r] = (uint16_t)f32_to_bf16_bits(x);
          else ep.out[(size_t)orow[i] * ep.ldo + n + r] = x;
        }
      }
    }
  } else {
    // W rows come in [16 gate | 16 up] blocks (launcher: N % 32 == 0): fragment j holds gate, j + 1 the matching up columns
    float4 bg[NI / 2 > 0 ? NI / 2 : 1], bu[NI / 2 > 0 ? NI / 2 : 1];
    if (ep.bias) {
#pragma unroll
      for (int j = 0; j + 1 < NI; j += 2) {
        const int nb = n0 + wc * (BN / 2) + j * 16;
        const int nc = nb + 32 <= N ? nb + n_in : 0;
        bg[j / 2] = *reinterpret_cast<const float4*>(ep.bias + nc);
        bu[j / 2] = *reinterpret_cast<const float4*>(ep.bias + nc + 16);
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (mrow[i] >= M || orow[i] < 0) continue;
#pragma unroll
      for (int j = 0; j + 1 < NI; j += 2) {
        const int nb = n0 + wc * (BN / 2) + j * 16;
        if (nb + 32 > N) continue;
        float g[4], u[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { g[r] = acc[i][j][r]; u[r] = acc[i][j + 1][r]; }
        if (ep.bias) {
          g[0] += bg[j / 2].x; g[1] += bg[j / 2].y; g[2] += bg[j / 2].z; g[3] += bg[j / 2].w;
          u[0] += bu[j / 2].x; u[1] += bu[j / 2].y; u[2] += bu[j / 2].z; u[3] += bu[j / 2].w;
        }
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_fast(g[r]) * u[r];
        const size_t o = (size_t)orow[i] * ep.ldo + (nb >> 1) + n_in;
        if (ep.out16) {
          if (ep.ldo % 4 == 0) {
            uint2 pk;
            pk.x = pack_bf16x2(v[0], v[1]);
            pk.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>(ep.out16 + o) = pk;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) ep.out16[o + r] = (uint16_t)f32_to_bf16_bits(v[r]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) ep.out[o + r] = v[r];
        }
      }
    }
  }
  Q3A_STAMP_AT(ep.stamp, blockIdx.x, 3);  // stores issued
}

// ---- small-M, K-heavy problems (one 30 s clip: M ~ 400, N ~ 1000, K 2048-7680) -------------------------------------
// A 32x64 tile grid yields fewer workgroups than CUs there and every workgroup walks all of K alone (48-120 barrier-
// separated steps).  Here the tile is 32 x BN and the workgroup's 4 waves split K instead: a K step stages
// (32 + BN) x BKT bf16 (BKT = 256: 32 KiB per stage, two stages), wave w multiplies the k range [w*BKT/4, (w+1)*BKT/4)
// of it, and the four partial 32 x BN accumulators are summed through LDS once at the end -- 4x fewer steps and BN = 32
// doubles the workgroup count.  The epilogue then has 4 consecutive columns per thread: 16-B (fp32) / 8-B (bf16) stores.
// LDS rows are BKT*2 = 512 / 256 B, so the 16 rows of a ds_read_b128 group share their bank offset; chunk c of row r is
// kept at position (c & ~15) | ((c ^ r) & 15).
__device__ __forceinline__ int kswz(int c, int row) { return (c & ~15) | ((c ^ row) & 15); }

template <int BN, int BKT, class ALoader, int NS = 2>
__global__ __launch_bounds__(256) void gemm16k_kernel(ALoader A, const uint16_t* __restrict__ Wt, int M, int N, int K,
                                                      GemmEpilogue ep) {
  constexpr int BM = 32;
  constexpr int CPR = BKT / 8;             // 16-B chunks per tile row (32 / 16)
  constexpr int RPI = 64 / CPR;            // tile rows per wave instruction (2 / 4)
  constexpr int A_LOADS = BM / (4 * RPI);  // glds instructions per wave per K step
  constexpr int W_LOADS = BN / (4 * RPI);
  constexpr int KS = BKT / 4 / 32;         // MFMA k-steps per wave per K step (2 / 1)
  constexpr int MI = BM / 16, NI = BN / 16;
  static_assert(A_LOADS >= 1 && W_LOADS >= 1 && KS >= 1, "tile shape");
  static_assert(NS == 2 || NS == 4, "stage count");
  constexpr int STAGE = (BM + BN) * BKT;  // elements per stage: A tile, then W tile
  static_assert(NS * STAGE * 2 >= 4 * BM * BN * 4, "reduction buffer must fit in the stages");
  __shared__ __attribute__((aligned(16))) uint16_t lds[NS * STAGE];  // the ONLY LDS object (k_gemm16 header: NS = 4)
  Q3A_STAMP_AT(ep.stamp, blockIdx.x, 0);  // entry

  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {  // bijective XCD remap: consecutive tile ids (sharing the W panel) stay on one XCD / L2
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = bid % tiles_m, tn = bid / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  const int l_row = lane / CPR, l_pos = lane % CPR;
  int a_s0[A_LOADS], a_s1[A_LOADS], a_s2[A_LOADS], a_chunk[A_LOADS];
  const uint16_t* w_src[W_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int row = (i * 4 + wave) * RPI + l_row;
    a_chunk[i] = kswz(l_pos, row);  // LDS position l_pos of this row holds global chunk kswz(l_pos, row) (an involution)
    A.init_row(m0 + row, M, a_s0[i], a_s1[i], a_s2[i]);
  }
#pragma unroll
  for (int i = 0; i < W_LOADS; ++i) {
    const int row = (i * 4 + wave) * RPI + l_row;
    const int n = n0 + row;
    w_src[i] = Wt + (size_t)(n < N ? n : N - 1) * K + kswz(l_pos, row) * 8;
  }
  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // The operands of the epilogue (bias, positional addend, residual) do not depend on the product: they are requested
  // here, ahead of the first K step, and have long arrived when the K loop ends.  (They are older than every LDS-DMA
  // below, so the counted waits of the ring -- "at most n NEWER loads outstanding" -- still mean what they say.)  With a
  // row map the residual address needs a loaded value first: that one launch per clip keeps the loads in the epilogue.
  constexpr int TPR = BN / 4;                     // threads per output row (4 columns each)
  constexpr int NIT = (BM * TPR + 255) / 256;    // epilogue items per thread
  const bool pre = ep.rowmap == nullptr;          // uniform
  float4 pb[NIT], pa[NIT], pr[NIT];
  if (pre) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + it * 256;
      const int row = (e / TPR) % BM, c4 = (e % TPR) * 4;
      const int mc = m0 + row < M ? m0 + row : M - 1, nc = n0 + c4 < N ? n0 + c4 : 0;
      if (ep.bias) pb[it] = *reinterpret_cast<const float4*>(ep.bias + nc);
      if (ep.addend) pa[it] = *reinterpret_cast<const float4*>(ep.addend + (size_t)(mc % ep.addend_period) * ep.ldo + nc);
      if (ep.resid) pr[it] = *reinterpret_cast<const float4*>(ep.resid + (size_t)mc * ep.ldo + nc);
    }
  }

  const int frag_row = lane & 15, frag_kc = lane >> 4;
  const int KT = K / BKT;
  auto issue_tile = [&](int kt, int buf) {
    const int k0 = kt * BKT;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      const uint16_t* ap = A.row_ptr(a_s0[i], a_s1[i], a_s2[i], k0) + a_chunk[i] * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)ap, (lptr_t)(lds + buf * STAGE + (i * 4 + wave) * RPI * BKT), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < W_LOADS; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + k0), (lptr_t)(lds + buf * STAGE + BM * BKT + (i * 4 + wave) * RPI * BKT), 16, 0,
                                       0);
  };
  if constexpr (NS == 2) {
    issue_tile(0, 0);
  } else {
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
      if (t < KT) issue_tile(t, t);
  }
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = NS == 2 ? (kt & 1) : (kt % NS);
    if constexpr (NS == 2) {
      __syncthreads();  // lands tile kt (vmcnt(0) rides on the barrier) and retires every wave's reads of tile kt-1
      if (kt + 1 < KT) issue_tile(kt + 1, buf ^ 1);
    } else {  // ring of NS stages, counted vmcnt + raw barrier: see gemm16_kernel
      constexpr int L = A_LOADS + W_LOADS;
      const int later = KT - 1 - kt;  // uniform
      if (later >= 2) g16_wait_vm<2 * L>();
      else if (later == 1) g16_wait_vm<L>();
      else g16_wait_vm<0>();
      asm volatile("s_barrier" ::: "memory");
      if (kt + NS - 1 < KT) issue_tile(kt + NS - 1, (kt + NS - 1) % NS);
    }
    if (kt == 0) Q3A_STAMP_AT(ep.stamp, blockIdx.x, 1);  // first K step landed
    const uint16_t* as = lds + buf * STAGE;
    const uint16_t* ws = as + BM * BKT;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = wave * (CPR / 4) + ks * 4 + frag_kc;  // this wave's quarter of the K step
      bf16x8_t bfrag[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = j * 16 + frag_row;
        bfrag[j] = *reinterpret_cast<const bf16x8_t*>(&ws[row * BKT + kswz(c, row) * 8]);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = i * 16 + frag_row;
        const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(&as[row * BKT + kswz(c, row) * 8]);
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfrag[j], acc[i][j], 0, 0, 0);
      }
    }
  }
  // ---- sum the four K quarters through LDS: red[wave][row][col] ----
  Q3A_STAMP_AT(ep.stamp, blockIdx.x, 2);  // K loop done
  __syncthreads();
  float* red = reinterpret_cast<float*>(lds);
  {
    const int col_in = lane & 15, row_in = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * BM + i * 16 + row_in + r) * BN + j * 16 + col_in] = acc[i][j][r];
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = tid + it * 256;
    if (e >= BM * TPR) continue;
    const int row = e / TPR, c4 = (e % TPR) * 4;
    const int m = m0 + row, n = n0 + c4;
    if (m >= M || n >= N) continue;
    const int orow = ep.rowmap ? ep.rowmap[m] : m;
    if (orow < 0) continue;
    float4 b4, a4, r4;
    if (pre) { b4 = pb[it]; a4 = pa[it]; r4 = pr[it]; }
    else {
      if (ep.bias) b4 = *reinterpret_cast<const float4*>(ep.bias + n);
      if (ep.addend) a4 = *reinterpret_cast<const float4*>(ep.addend + (size_t)(m % ep.addend_period) * ep.ldo + n);
      if (ep.resid) r4 = *reinterpret_cast<const float4*>(ep.resid + (size_t)orow * ep.ldo + n);
    }
    float4 v = *reinterpret_cast<const float4*>(&red[row * BN + c4]);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 p = *reinterpret_cast<const float4*>(&red[(w * BM + row) * BN + c4]);
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    if (ep.bias) { v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
    if (ep.addend) { v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w; }
    if (ep.act == 1) { v.x = gelu_fast(v.x); v.y = gelu_fast(v.y); v.z = gelu_fast(v.z); v.w = gelu_fast(v.w); }
    if (ep.resid) { v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
    if (ep.out16) {
      uint2 pk;
      pk.x = pack_bf16x2(v.x, v.y);
      pk.y = pack_bf16x2(v.z, v.w);
      *reinterpret_cast<uint2*>(ep.out16 + (size_t)orow * ep.ldo + n) = pk;
    } else {
      *reinterpret_cast<float4*>(ep.out + (size_t)orow * ep.ldo + n) = v;
    }
  }
  Q3A_STAMP_AT(ep.stamp, blockIdx.x, 3);  // stores issued
}

template <int BM, int BN, int BK, bool GLU, class ALoader>
void launch16(const ALoader& A, const uint16_t* W, int M, int N, int K, const GemmEpilogue& ep, hipStream_t s) {
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  // Dense operands, BK = 64, tiles up to 64x64, (the A/B knob gemm16_ring of rounds 3-5 is gone: the ring won) a ring of
  // LDS stages as deep as still lets every workgroup of the launch be resident at once (64x64: 16 KiB per stage, 160 KiB per
  // CU) -- four stages up to 2 workgroups per CU, three up to 3; beyond that (e.g. the lm_head of a batched decode step,
  // 2374 workgroups) residency is worth more than depth: two.
  constexpr bool ring = std::is_same<ALoader, DenseA16>::value && BK == 64 && BM <= 64 && BN <= 64;
  if constexpr (ring) {
    if (tiles <= 512) {
      hipLaunchKernelGGL((gemm16_kernel<BM, BN, BK, GLU, ALoader, 4>), dim3(tiles), dim3(256), 0, s, A, W, M, N, K, ep);
      return;
    }
    if (tiles <= 768) {
      hipLaunchKernelGGL((gemm16_kernel<BM, BN, BK, GLU, ALoader, 3>), dim3(tiles), dim3(256), 0, s, A, W, M, N, K, ep);
      return;
    }
  }
  hipLaunchKernelGGL((gemm16_kernel<BM, BN, BK, GLU, ALoader, 2>), dim3(tiles), dim3(256), 0, s, A, W, M, N, K, ep);
}

inline long tiles_of(int M, int N, int bm, int bn) { return (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn); }

// tile by workgroup count: enough 128x128 tiles to fill 256 CUs 1.5x over, else 64x64, else 32x64
template <int BK, bool GLU, class ALoader>
void launch_sized16(const ALoader& A, const uint16_t* W, int M, int N, int K, const GemmEpilogue& ep, hipStream_t s) {
  if (M <= 32) launch16<32, 64, BK, GLU>(A, W, M, N, K, ep, s);  // batched decode lm_head: pure weight streaming
  else if (tiles_of(M, N, 128, 128) >= 384) launch16<128, 128, BK, GLU>(A, W, M, N, K, ep, s);
  else if (tiles_of(M, N, 64, 64) >= 384) launch16<64, 64, BK, GLU>(A, W, M, N, K, ep, s);
  else launch16<32, 64, BK, GLU>(A, W, M, N, K, ep, s);
}

__global__ void to_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    uint2 p;
    p.x = pack_bf16x2(v.x, v.y);
    p.y = pack_bf16x2(v.z, v.w);
    reinterpret_cast<uint2*>(y)[i] = p;
  }
}

__global__ void from_bf16_kernel(const uint16_t* __restrict__ x, float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = bf16_bits_to_f32(x[i]);
}

}  // namespace

const char* launch_from_bf16(const uint16_t* x, float* y, size_t n, hipStream_t s) {
  if (n == 0) return nullptr;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(from_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, y, n);
  return nullptr;
}

const char* launch_gemm16(const uint16_t* X, int lda, const uint16_t* W, int M, int N, int K, const GemmEpilogue& ep,
                          bool glu, hipStream_t s) {
  if (M <= 0) return nullptr;
  if (K % 32 != 0 || lda % 8 != 0) return "gemm16: K must be a multiple of 32 and lda of 8";
  if (K % 64 == 0 && !ep.part_val && gemm256_eligible(M, N, K)) return launch_gemm256(X, lda, W, M, N, K, ep, glu, s);
  return launch_gemm16_small(X, lda, W, M, N, K, ep, glu, s);
}

const char* launch_gemm16_small(const uint16_t* X, int lda, const uint16_t* W, int M, int N, int K, const GemmEpilogue& ep,
                                bool glu, hipStream_t s) {
  if (M <= 0) return nullptr;
  if (K % 32 != 0 || lda % 8 != 0) return "gemm16: K must be a multiple of 32 and lda of 8";
  if (glu && N % 32 != 0) return "gemm16: GLU needs N % 32 == 0";
  if (ep.part_val && (M > 32 || glu || N % 4 != 0 || ep.ldo % 4 != 0 || K % 64 != 0 || ep.rowmap || ep.part_stride < (N + 63) / 64))
    return "gemm16: argmax partials need M <= 32, 4 | N, 4 | ldo, 64 | K, no row map and part_stride >= ceil(N / 64)";
  DenseA16 A{X, lda};
  // few tiles and a long K: 32x32 tiles whose 4 waves split K (A/B knob: Q3A_GEMM16_KSPLIT=0 disables)
  static const bool ksplit_on = [] { const char* e = getenv("Q3A_GEMM16_KSPLIT"); return !e || atoi(e) != 0; }();
  if (ksplit_on && !glu && !ep.part_val && tiles_of(M, N, 32, 64) < 384 && K >= 512 && K % 128 == 0 && N % 4 == 0 && ep.ldo % 4 == 0) {
    const int tiles = ((M + 31) / 32) * ((N + 31) / 32);
    // K steps of 256 in two stages (32 KiB in flight per workgroup); where 256 does not divide K (the encoder's d_model 896)
    // steps of 128 in a ring of four stages (48 KiB in flight) instead of two (16 KiB): 6.5 vs 7.9 us on enc out.  The ring
    // measured no gain over the 256-steps (profiles/r3_phase_probe_one_clip_gemms.txt).
    // (Measured and dropped: 32x64 tiles, at most one per CU, ring of four 128-steps -- the launch then moves 96 instead of
    // 2 x 64 operand rows per K step on its busiest CUs, but 48-74 CUs idle: 926 vs 930 us over the 92 launches of a clip.
    // These launches run at ~75 GB/s of LDS-DMA per CU whatever the tile, profiles/r3_phase_probe_one_clip_gemms.txt.)
    if (K % 256 == 0) hipLaunchKernelGGL((gemm16k_kernel<32, 256, DenseA16, 2>), dim3(tiles), dim3(256), 0, s, A, W, M, N, K, ep);
    else hipLaunchKernelGGL((gemm16k_kernel<32, 128, DenseA16, 4>), dim3(tiles), dim3(256), 0, s, A, W, M, N, K, ep);
    return nullptr;
  }
  static const bool force_bk32 = [] { const char* e = getenv("Q3A_GEMM16_BK32"); return e && atoi(e) != 0; }();  // A/B knob
  if (K % 64 == 0 && !force_bk32) {
    if (glu) launch_sized16<64, true>(A, W, M, N, K, ep, s); else launch_sized16<64, false>(A, W, M, N, K, ep, s);
  } else {
    if (glu) launch_sized16<32, true>(A, W, M, N, K, ep, s); else launch_sized16<32, false>(A, W, M, N, K, ep, s);
  }
  return nullptr;
}

const char* launch_to_bf16(const float* x, uint16_t* y, size_t n, hipStream_t s) {
  if (n == 0) return nullptr;
  if (n % 4 != 0) return "to_bf16: n must be a multiple of 4";
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, y, n / 4);
  return nullptr;
}

const char* launch_conv3x3s2_gemm16(const uint16_t* X, const uint16_t* zero_page, int imgs, int H, int Wd, int C,
                                    const uint16_t* Wt, int Cout, const GemmEpilogue& ep, hipStream_t s) {
  if (C % 32 != 0) return "conv gemm16: C must be a multiple of 32";
  ConvA16 A{X, zero_page, H, Wd, C, (H - 1) / 2 + 1, (Wd - 1) / 2 + 1};
  const int M = imgs * A.OH * A.OW;
  if (M <= 0) return nullptr;
  if (gemm256_eligible(M, Cout, 9 * C)) return launch_conv3x3s2_gemm256(X, zero_page, imgs, H, Wd, C, Wt, Cout, ep, s);
  // a K tile must not straddle two filter taps: BK has to divide C
  if (C % 64 == 0) launch_sized16<64, false>(A, Wt, M, Cout, 9 * C, ep, s);
  else launch_sized16<32, false>(A, Wt, M, Cout, 9 * C, ep, s);
  return nullptr;
}

}  // namespace q3a
