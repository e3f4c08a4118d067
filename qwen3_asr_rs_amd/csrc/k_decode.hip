// Decoder glue + the single-token (decode) kernels for gfx950.
//
//   embed_kernel            Tensor::embedding (src/tensor.rs:209-213, src/text_decoder.rs:90-92)
//   qknorm_rope_kv_kernel   per-head RMSNorm on q/k (src/layers.rs:303-304), RoPE x*cos+rotate_half(x)*sin
//                           (src/layers.rs:361-375) and the KV-cache append (src/layers.rs:311-319) -- into a
//                           pre-allocated contiguous cache instead of the reference's per-step `cat`
//   gemv_kernel             Linear::forward at one token per sequence (src/layers.rs:74-80) with the
//                           RMSNorm (layers.rs:48-54), bias, residual add and SiLU(gate)*up (layers.rs:396-400)
//                           fused: the weight matrix is streamed from HBM exactly once, 16 B per lane per
//                           load, x stays in LDS as fp32 (weights are exact bf16 -> fp32-accurate products)
//   decode_attn_kernel      q/k norm + RoPE + cache append + attention over the cache for the new token
//   argmax_finalize_kernel  argmax (first-index tie-break, src/tensor.rs:370-372), EOS flag, id store and the
//                           embedding of the chosen token for the next step -- no per-token D2H sync
//                           (the reference does int64_value per token, src/inference.rs:161)
#include "dev.h"
#include "kernels.h"

namespace q3a {

int gemv_rows_per_wave(const GemvArgs& a) {  // physical rows per wave
  const int logical = (a.mode == 2) ? a.N / 2 : a.N;
  if (logical >= 32768) return 4;
  if (logical >= 4096 || a.mode == 2) return 2;
  return 1;
}
int gemv_blocks(const GemvArgs& a) {
  const int pr = gemv_rows_per_wave(a);
  const int rows_per_wave = (a.mode == 2) ? pr / 2 : pr;
  const int logical = (a.mode == 2) ? a.N / 2 : a.N;
  return (logical + 4 * rows_per_wave - 1) / (4 * rows_per_wave);
}

namespace {

// --------------------------------------------------------------------------------------------------
__global__ void embed_kernel(const int* __restrict__ ids, const uint16_t* __restrict__ embed, int H, int skip_id,
                             float* __restrict__ out) {
  const int r = blockIdx.x;
  const int id = ids[r];
  if (id == skip_id) return;
  const uint2* src = reinterpret_cast<const uint2*>(embed + (size_t)id * H);
  float4* dst = reinterpret_cast<float4*>(out + (size_t)r * H);
  for (int i = threadIdx.x; i < H / 4; i += blockDim.x) {
    const uint2 v = src[i];
    dst[i] = make_float4(bf16lo(v.x), bf16hi(v.x), bf16lo(v.y), bf16hi(v.y));
  }
}

__global__ void set_tokens_kernel(const int* __restrict__ tok, const uint16_t* __restrict__ embed, int H,
                                  float* __restrict__ x_next, int* __restrict__ next_tok) {
  const int s = blockIdx.x;
  const int id = tok[s];
  if (threadIdx.x == 0) next_tok[s] = id;
  const uint2* src = reinterpret_cast<const uint2*>(embed + (size_t)id * H);
  float4* dst = reinterpret_cast<float4*>(x_next + (size_t)s * H);
  for (int i = threadIdx.x; i < H / 4; i += blockDim.x) {
    const uint2 v = src[i];
    dst[i] = make_float4(bf16lo(v.x), bf16hi(v.x), bf16lo(v.y), bf16hi(v.y));
  }
}

// --------------------------------------------------------------------------------------------------
// One wave per 128-wide head vector.  lane owns dims (lane, lane+64): the rotate_half partners.
__device__ __forceinline__ void head_norm_rope(float& x1, float& x2, const float* __restrict__ w, float eps,
                                               const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                               int pos, int lane) {
  const float ss = wave_sum(x1 * x1 + x2 * x2);
  const float rstd = 1.0f / sqrtf(ss / 128.0f + eps);  // layers.rs:50-52
  const float n1 = (x1 * rstd) * w[lane], n2 = (x2 * rstd) * w[lane + 64];
  const float c = cos_t[(size_t)pos * 64 + lane], sn = sin_t[(size_t)pos * 64 + lane];
  x1 = n1 * c + (-n2) * sn;  // layers.rs:366: x*cos + rotate_half(x)*sin, rotate_half = cat(-x2, x1)
  x2 = n2 * c + n1 * sn;
}

template <typename KVT>
__global__ __launch_bounds__(256) void qknorm_rope_kv_kernel(RopeKvArgs a, int rows) {
  const int lane = threadIdx.x & 63;
  const int nvec = a.n_q + 2 * a.n_kv;
  const long vi = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (vi >= (long)rows * nvec) return;
  const int row = (int)(vi / nvec), hv = (int)(vi % nvec);
  float* base = a.qkv + (size_t)row * nvec * 128 + (size_t)hv * 128;
  float x1 = base[lane], x2 = base[lane + 64];
  const int pos = a.row_pos[row], seq = a.row_seq[row];
  if (hv < a.n_q) {
    head_norm_rope(x1, x2, a.q_norm, a.eps, a.cos_t, a.sin_t, pos, lane);
    base[lane] = x1;
    base[lane + 64] = x2;
  } else {
    const bool is_k = hv < a.n_q + a.n_kv;
    const int kvh = is_k ? hv - a.n_q : hv - a.n_q - a.n_kv;
    if (is_k) head_norm_rope(x1, x2, a.k_norm, a.eps, a.cos_t, a.sin_t, pos, lane);
    KVT* c = reinterpret_cast<KVT*>(is_k ? a.kcache : a.vcache) + (((size_t)seq * a.n_kv + kvh) * a.max_ctx + pos) * 128;
    KvIo<KVT>::store(c + lane, x1);
    KvIo<KVT>::store(c + lane + 64, x2);
  }
}

// --------------------------------------------------------------------------------------------------
// GEMV: PR physical weight rows per wave, NB activation rows, PF k-iterations of weights prefetched.
// Order of work inside a block (everything is latency: a 4-12 MB matrix is ~16-48 KB per CU):
//   1. issue the first PF x 16-B weight loads of every row this wave owns (they do not depend on x)
//   2. stage x (times the RMSNorm weight) into LDS while those loads fly, accumulate sum(x^2)
//   3. FMA; the RMSNorm scale 1/sqrt(mean(x^2)+eps) is a scalar, so it multiplies the finished dot
//      product instead of every x (same value up to fp32 rounding of the reference's (x*rstd)*w order)
//   4. epilogue: bias / residual / SiLU(gate)*up / logits + per-block argmax partial
template <int NB, int PR, int PF>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [NB][K]
  __shared__ float red[NB][4];
  __shared__ float am_v[4][NB];
  __shared__ int am_i[4][NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K;
  const bool glu = a.mode == 2;
  const int g = blockIdx.x * 4 + wave;
  int prow[PR];
#pragma unroll
  for (int i = 0; i < PR; ++i) {
    if (glu) {
      const int j = g * (PR / 2) + (i >> 1);  // logical row
      prow[i] = (j / 16) * 32 + (j % 16) + ((i & 1) ? 16 : 0);
    } else {
      prow[i] = g * PR + i;
    }
    if (prow[i] >= a.N) prow[i] = -1;
  }
  // ---- 1. weight prefetch ----
  uint4 wq[PF][PR];
#pragma unroll
  for (int it = 0; it < PF; ++it) {
    const int k = lane * 8 + it * 512;
#pragma unroll
    for (int i = 0; i < PR; ++i)
      wq[it][i] = (prow[i] >= 0 && k < K) ? *reinterpret_cast<const uint4*>(a.W + (size_t)prow[i] * K + k)
                                          : make_uint4(0u, 0u, 0u, 0u);
  }
  // ---- 2. stage x * rms_w ----
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const float4* src = reinterpret_cast<const float4*>(a.x + (size_t)b * a.ldx);
    float4* dst = reinterpret_cast<float4*>(xs + (size_t)b * K);
    float ss = 0.f;
    for (int i = tid; i < K / 4; i += 256) {
      float4 v = src[i];
      ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      if (a.rms_w) {
        const float4 w = reinterpret_cast<const float4*>(a.rms_w)[i];
        v.x *= w.x; v.y *= w.y; v.z *= w.z; v.w *= w.w;
      }
      dst[i] = v;
    }
    if (a.rms_w) {
      ss = wave_sum(ss);
      if (lane == 0) red[b][wave] = ss;
    }
  }
  __syncthreads();

  // ---- 3. dot products ----
  float acc[PR][NB];
#pragma unroll
  for (int i = 0; i < PR; ++i)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[i][b] = 0.f;
  auto fma8 = [&](const uint4 (&w)[PR], int k) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float4 xa = *reinterpret_cast<const float4*>(xs + (size_t)b * K + k);
      const float4 xb = *reinterpret_cast<const float4*>(xs + (size_t)b * K + k + 4);
#pragma unroll
      for (int i = 0; i < PR; ++i) {
        float s = acc[i][b];
        s += bf16lo(w[i].x) * xa.x; s += bf16hi(w[i].x) * xa.y; s += bf16lo(w[i].y) * xa.z; s += bf16hi(w[i].y) * xa.w;
        s += bf16lo(w[i].z) * xb.x; s += bf16hi(w[i].z) * xb.y; s += bf16lo(w[i].w) * xb.z; s += bf16hi(w[i].w) * xb.w;
        acc[i][b] = s;
      }
    }
  };
#pragma unroll
  for (int it = 0; it < PF; ++it) {
    const int k = lane * 8 + it * 512;
    if (k < K) fma8(wq[it], k);
  }
  for (int k = lane * 8 + PF * 512; k < K; k += 512) {
    uint4 w[PR];
#pragma unroll
    for (int i = 0; i < PR; ++i)
      w[i] = prow[i] >= 0 ? *reinterpret_cast<const uint4*>(a.W + (size_t)prow[i] * K + k) : make_uint4(0u, 0u, 0u, 0u);
    fma8(w, k);
  }
#pragma unroll
  for (int i = 0; i < PR; ++i)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[i][b] = wave_sum(acc[i][b]);
  if (a.rms_w) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float rstd = 1.0f / sqrtf((red[b][0] + red[b][1] + red[b][2] + red[b][3]) / (float)K + a.eps);
#pragma unroll
      for (int i = 0; i < PR; ++i) acc[i][b] *= rstd;
    }
  }

  // ---- 4. epilogue ----
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
      a.part_val[(size_t)tid * a.part_stride + blockIdx.x] = v;
      a.part_idx[(size_t)tid * a.part_stride + blockIdx.x] = ix;
    }
    return;
  }
  if (lane == 0) {
    if (!glu) {
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
          a.out[(size_t)b * a.ldo + j] = silu_f(gv) * uv;
        }
      }
    }
  }
}

template <int NB, int PR, int PF>
void gemv_launch_t(const GemvArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((gemv_kernel<NB, PR, PF>), dim3(gemv_blocks(a)), dim3(256), (size_t)NB * a.K * sizeof(float), s, a);
}
template <int NB>
void gemv_launch_nb(const GemvArgs& a, hipStream_t s) {
  const int pr = gemv_rows_per_wave(a);
  if (pr == 4) gemv_launch_t<NB, 4, 2>(a, s);
  else if (pr == 2) gemv_launch_t<NB, 2, 4>(a, s);
  else gemv_launch_t<NB, 1, 6>(a, s);
}

// --------------------------------------------------------------------------------------------------
// DPP sum over the 16 lanes of a row (all 16 lanes end up with the total)
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

constexpr int DA_WAVES = 8;

// One workgroup per (sequence, kv head).  Cache rows are read with fully coalesced 16-B-per-lane loads:
// LPK lanes share one key (each owns DPL head dims), a wave instruction covers KPI consecutive keys.
// Scores need a DPP reduction over the LPK lanes; P.V needs no cross-lane traffic until the very end.
template <int GROUP, typename KVT>
__global__ __launch_bounds__(DA_WAVES * 64) void decode_attn_kernel(DecodeAttnArgs a) {
  constexpr int DPL = Frag16<KVT>::DPL;  // head dims per lane: 8 (bf16) / 4 (f32)
  constexpr int LPK = 128 / DPL;         // lanes per key: 16 / 32
  constexpr int KPI = 64 / LPK;          // keys per load instruction: 4 / 2
  constexpr int NI = 8;                  // load instructions per key block (K and V: 2*NI*4 raw VGPRs in flight)
  constexpr int KG = KPI * NI;           // keys per block: 32 / 16
  __shared__ float q_s[GROUP][128];
  __shared__ float k_s[128];
  __shared__ float v_s[128];
  __shared__ float cm[DA_WAVES][GROUP], cl[DA_WAVES][GROUP];
  __shared__ float co[DA_WAVES][GROUP][128];
  const int s = blockIdx.y, kvh = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane % LPK, kq = lane / LPK;
  const int pos = a.pos[s];
  const int qkv_dim = (a.n_q + 2 * a.n_kv) * 128;
  const float* row = a.qkv + (size_t)s * qkv_dim;
  KVT* kc = reinterpret_cast<KVT*>(a.kcache) + ((size_t)s * a.n_kv + kvh) * (size_t)a.max_ctx * 128;
  KVT* vc = reinterpret_cast<KVT*>(a.vcache) + ((size_t)s * a.n_kv + kvh) * (size_t)a.max_ctx * 128;
  const int nkeys = pos + 1;
  const int nblocks = (nkeys + KG - 1) / KG;

  uint4 kraw[NI], vraw[NI];
  auto load_block = [&](int kb) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int key = kb * KG + i * KPI + kq;
      if (key < pos) {
        kraw[i] = *reinterpret_cast<const uint4*>(kc + (size_t)key * 128 + sub * DPL);
        vraw[i] = *reinterpret_cast<const uint4*>(vc + (size_t)key * 128 + sub * DPL);
      } else {
        kraw[i] = make_uint4(0u, 0u, 0u, 0u);
        vraw[i] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  if (wave < nblocks) load_block(wave);  // cached keys do not depend on the new token: fetch them first

  // ---- phase A: normalise/rotate the new q (GROUP heads) and k, append k/v to the cache ----
  if (wave < GROUP) {
    const int h = kvh * GROUP + wave;
    float x1 = row[h * 128 + lane], x2 = row[h * 128 + lane + 64];
    head_norm_rope(x1, x2, a.q_norm, a.eps, a.cos_t, a.sin_t, pos, lane);
    q_s[wave][lane] = x1;
    q_s[wave][lane + 64] = x2;
  } else if (wave == GROUP) {
    const float* p = row + (a.n_q + kvh) * 128;
    float x1 = p[lane], x2 = p[lane + 64];
    head_norm_rope(x1, x2, a.k_norm, a.eps, a.cos_t, a.sin_t, pos, lane);
    KvIo<KVT>::store(kc + (size_t)pos * 128 + lane, x1);
    KvIo<KVT>::store(kc + (size_t)pos * 128 + lane + 64, x2);
    k_s[lane] = KvIo<KVT>::round(x1);
    k_s[lane + 64] = KvIo<KVT>::round(x2);
  } else if (wave == GROUP + 1) {
    const float* p = row + (a.n_q + a.n_kv + kvh) * 128;
    const float x1 = p[lane], x2 = p[lane + 64];
    KvIo<KVT>::store(vc + (size_t)pos * 128 + lane, x1);
    KvIo<KVT>::store(vc + (size_t)pos * 128 + lane + 64, x2);
    v_s[lane] = KvIo<KVT>::round(x1);
    v_s[lane + 64] = KvIo<KVT>::round(x2);
  }
  __syncthreads();

  // ---- phase B ----
  float qf[GROUP][DPL], acc[GROUP][DPL], mrun[GROUP], lrun[GROUP];
#pragma unroll
  for (int g = 0; g < GROUP; ++g) {
#pragma unroll
    for (int e = 0; e < DPL; ++e) { qf[g][e] = q_s[g][sub * DPL + e]; acc[g][e] = 0.f; }
    mrun[g] = -INFINITY;
    lrun[g] = 0.f;
  }
  for (int kb = wave; kb < nblocks; kb += DA_WAVES) {
    if (kb != wave) load_block(kb);
    const int key_base = kb * KG + kq;
    float sc[NI][GROUP];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int key = key_base + i * KPI;
      float kf[DPL];
      Frag16<KVT>::unpack(kraw[i], kf);
      if (key == pos) {
#pragma unroll
        for (int e = 0; e < DPL; ++e) kf[e] = k_s[sub * DPL + e];
      }
#pragma unroll
      for (int g = 0; g < GROUP; ++g) {
        float p = 0.f;
#pragma unroll
        for (int e = 0; e < DPL; ++e) p += qf[g][e] * kf[e];
        p = row16_sum(p);
        if (LPK == 32) p += __shfl_xor(p, 16, 64);
        sc[i][g] = (key <= pos) ? p / a.scale_div : -INFINITY;
      }
    }
#pragma unroll
    for (int g = 0; g < GROUP; ++g) {
      float mx = sc[0][g];
#pragma unroll
      for (int i = 1; i < NI; ++i) mx = fmaxf(mx, sc[i][g]);
      if (LPK == 16) mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(mrun[g], mx);  // finite: key kb*KG <= pos is valid
      const float alpha = expf(mrun[g] - m_new);
      mrun[g] = m_new;
      lrun[g] *= alpha;
#pragma unroll
      for (int e = 0; e < DPL; ++e) acc[g][e] *= alpha;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int key = key_base + i * KPI;
      float vf[DPL];
      Frag16<KVT>::unpack(vraw[i], vf);
      if (key == pos) {
#pragma unroll
        for (int e = 0; e < DPL; ++e) vf[e] = v_s[sub * DPL + e];
      }
#pragma unroll
      for (int g = 0; g < GROUP; ++g) {
        const float p = (key <= pos) ? expf(sc[i][g] - mrun[g]) : 0.f;
        lrun[g] += p;
#pragma unroll
        for (int e = 0; e < DPL; ++e) acc[g][e] += p * vf[e];
      }
    }
  }
  // fold the KPI key columns of the wave (lanes with equal `sub`) together
#pragma unroll
  for (int g = 0; g < GROUP; ++g) {
    if (LPK == 16) lrun[g] += __shfl_xor(lrun[g], 16, 64);
    lrun[g] += __shfl_xor(lrun[g], 32, 64);
#pragma unroll
    for (int e = 0; e < DPL; ++e) {
      if (LPK == 16) acc[g][e] += __shfl_xor(acc[g][e], 16, 64);
      acc[g][e] += __shfl_xor(acc[g][e], 32, 64);
    }
    if (lane == 0) { cm[wave][g] = mrun[g]; cl[wave][g] = lrun[g]; }
    if (kq == 0) {
#pragma unroll
      for (int e = 0; e < DPL; ++e) co[wave][g][sub * DPL + e] = acc[g][e];
    }
  }
  __syncthreads();
  if (wave < GROUP) {
    const int g = wave;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < DA_WAVES; ++w) M = fmaxf(M, cm[w][g]);
    float L = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
    for (int w = 0; w < DA_WAVES; ++w) {
      const float f = (cm[w][g] == -INFINITY) ? 0.f : expf(cm[w][g] - M);
      L += cl[w][g] * f;
      o0 += co[w][g][lane] * f;
      o1 += co[w][g][lane + 64] * f;
    }
    const int h = kvh * GROUP + g;
    float* o = a.out + (size_t)s * a.n_q * 128 + (size_t)h * 128;
    o[lane] = o0 / L;
    o[lane + 64] = o1 / L;
  }
}

// --------------------------------------------------------------------------------------------------
// argmax, stage 1 (GEMM path only; the GEMV lm_head writes its partials itself): block partial maxima
__global__ __launch_bounds__(256) void argmax_partial_kernel(const float* __restrict__ logits, int V, float* __restrict__ pval,
                                                             int* __restrict__ pidx, int stride) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  const int s = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (V + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * per, hi = min(V, lo + per);
  const float* lg = logits + (size_t)s * V;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int i = lo + tid; i < hi; i += 256) {
    const float v = lg[i];
    if (v > best) { best = v; idx = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
  }
  if (lane == 0) { bv[wave] = best; bi[wave] = idx; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    pval[(size_t)s * stride + blockIdx.x] = best;
    pidx[(size_t)s * stride + blockIdx.x] = idx;
  }
}

// argmax, stage 2 + bookkeeping of the greedy loop
__global__ __launch_bounds__(1024) void argmax_finalize_kernel(FinalizeArgs a) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  __shared__ int tok_s;
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* pv = a.part_val + (size_t)s * a.part_stride;
  const int* pi = a.part_idx + (size_t)s * a.part_stride;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int i = tid; i < a.n_part; i += 1024) {
    const float v = pv[i];
    const int ix = pi[i];
    if (v > best || (v == best && ix < idx)) { best = v; idx = ix; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
  }
  if (lane == 0) { bv[wave] = best; bi[wave] = idx; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    if (idx < 0 || idx >= a.V) idx = 0;  // all-NaN guard
    tok_s = idx;
    a.next_tok[s] = idx;
    const int sc = a.step_count[s];
    if (sc < a.out_stride) a.out_ids[(size_t)s * a.out_stride + sc] = idx;
    a.step_count[s] = sc + 1;
    a.pos[s] += a.advance;
    if (idx == a.eos0 || idx == a.eos1) a.done[s] = 1;
  }
  __syncthreads();
  const int tok = tok_s;
  const uint2* src = reinterpret_cast<const uint2*>(a.embed + (size_t)tok * a.H);
  float4* dst = reinterpret_cast<float4*>(a.x_next + (size_t)s * a.H);
  for (int i = tid; i < a.H / 4; i += 1024) {
    const uint2 v = src[i];
    dst[i] = make_float4(bf16lo(v.x), bf16hi(v.x), bf16lo(v.y), bf16hi(v.y));
  }
}

}  // namespace

const char* launch_embed(const int* ids, int rows, const uint16_t* embed, int H, int skip_id, float* out,
                         hipStream_t s) {
  if (rows <= 0) return nullptr;
  hipLaunchKernelGGL(embed_kernel, dim3(rows), dim3(256), 0, s, ids, embed, H, skip_id, out);
  return nullptr;
}
const char* launch_set_tokens(const int* tok, int S, const uint16_t* embed, int H, float* x_next, int* next_tok,
                              hipStream_t s) {
  hipLaunchKernelGGL(set_tokens_kernel, dim3(S), dim3(256), 0, s, tok, embed, H, x_next, next_tok);
  return nullptr;
}
const char* launch_qknorm_rope_kv(const RopeKvArgs& a, int rows, bool kv_f32, hipStream_t s) {
  if (rows <= 0) return nullptr;
  const long nvec = (long)rows * (a.n_q + 2 * a.n_kv);
  const int blocks = (int)((nvec + 3) / 4);
  if (kv_f32) hipLaunchKernelGGL(qknorm_rope_kv_kernel<float>, dim3(blocks), dim3(256), 0, s, a, rows);
  else hipLaunchKernelGGL(qknorm_rope_kv_kernel<uint16_t>, dim3(blocks), dim3(256), 0, s, a, rows);
  return nullptr;
}

const char* launch_gemv(const GemvArgs& a0, int NB, hipStream_t s) {
  if (a0.K % 8 != 0) return "gemv: K must be a multiple of 8";
  if (a0.mode == 2 && a0.N % 32 != 0) return "gemv: GLU needs N % 32 == 0";
  if (a0.mode == 3 && (!a0.part_val || !a0.part_idx || gemv_blocks(a0) > a0.part_stride)) return "gemv: argmax partial buffer missing/too small";
  if ((size_t)a0.K * 4 > 64 * 1024) return "gemv: K too large for the LDS staging buffer";
  const int nb_cap = (int)((64 * 1024) / ((size_t)a0.K * 4));  // rows of x that fit in 64 KiB of LDS
  int done = 0;
  while (done < NB) {
    GemvArgs a = a0;
    a.x = a0.x + (size_t)done * a0.ldx;
    if (a0.out) a.out = a0.out + (size_t)done * a0.ldo;
    if (a0.resid) a.resid = a0.resid + (size_t)done * a0.ldo;
    if (a0.part_val) { a.part_val = a0.part_val + (size_t)done * a0.part_stride; a.part_idx = a0.part_idx + (size_t)done * a0.part_stride; }
    int nb = NB - done;
    if (nb > nb_cap) nb = nb_cap;
    if (nb >= 4) { nb = 4; gemv_launch_nb<4>(a, s); }
    else if (nb >= 2) { nb = 2; gemv_launch_nb<2>(a, s); }
    else { nb = 1; gemv_launch_nb<1>(a, s); }
    done += nb;
  }
  return nullptr;
}

const char* launch_decode_attn(const DecodeAttnArgs& a, int S, bool kv_f32, hipStream_t s) {
  if (S <= 0) return nullptr;
  const int group = a.n_q / a.n_kv;
  dim3 grid(a.n_kv, S), block(DA_WAVES * 64);
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

const char* launch_argmax_partials(const float* logits, int V, int S, float* pval, int* pidx, int stride, int nblk,
                                   hipStream_t s) {
  if (S <= 0) return nullptr;
  if (nblk > stride) return "argmax: partial buffer too small";
  hipLaunchKernelGGL(argmax_partial_kernel, dim3(nblk, S), dim3(256), 0, s, logits, V, pval, pidx, stride);
  return nullptr;
}
const char* launch_argmax_finalize(const FinalizeArgs& a, int S, hipStream_t s) {
  if (S <= 0) return nullptr;
  hipLaunchKernelGGL(argmax_finalize_kernel, dim3(S), dim3(1024), 0, s, a);
  return nullptr;
}

}  // namespace q3a
