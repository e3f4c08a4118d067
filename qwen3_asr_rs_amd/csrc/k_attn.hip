// Segment attention for gfx950: softmax(Q K^T / sqrt(hd) [+ causal]) V inside independent segments.
//
// Replaces (a) AudioAttention::forward's dense masked attention (src/layers.rs:152-172) together
// with the T x T block-diagonal mask of build_window_mask (src/audio_encoder.rs:172-260): each window
// of <= 8 chunks (<= 104 tokens) is one segment, so the -inf blocks are never computed; and
// (b) TextAttention::forward's prefill attention with repeat_kv and the causal mask
// (src/layers.rs:321-335, src/text_decoder.rs:121-131): GQA shares each K/V head between GROUP query
// heads inside the workgroup instead of materialising the repeat.
//
// fp32 VALU kernel (attention is ~1-5 % of the path's FLOPs): one lane per key.  A wave keeps a
// 64-key block of K in registers (lane j = key j), broadcasts the query through v_readlane, does the
// softmax statistics with wavefront shuffles (online softmax across key blocks) and accumulates P.V
// with V staged in LDS (lane d owns output dims d, d+64).  Scores are divided by sqrt(hd) after the
// dot product exactly as the reference does (layers.rs:161-162,327-328).
#include "dev.h"
#include "kernels.h"

namespace q3a {
namespace {

template <typename KVT> struct RowLoad;
template <> struct RowLoad<float> {
  template <int HD> static __device__ __forceinline__ void load(const float* p, float (&r)[HD]) {
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
      const float4 v = reinterpret_cast<const float4*>(p)[i];
      r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
    }
  }
};
template <> struct RowLoad<uint16_t> {
  template <int HD> static __device__ __forceinline__ void load(const uint16_t* p, float (&r)[HD]) {
#pragma unroll
    for (int i = 0; i < HD / 8; ++i) {
      const uint4 v = reinterpret_cast<const uint4*>(p)[i];
      r[8 * i] = bf16lo(v.x); r[8 * i + 1] = bf16hi(v.x); r[8 * i + 2] = bf16lo(v.y); r[8 * i + 3] = bf16hi(v.y);
      r[8 * i + 4] = bf16lo(v.z); r[8 * i + 5] = bf16hi(v.z); r[8 * i + 6] = bf16lo(v.w); r[8 * i + 7] = bf16hi(v.w);
    }
  }
};

template <int HD, int GROUP, bool CAUSAL, typename KVT, int QPW>
__global__ __launch_bounds__(256) void attn_kernel(AttnArgs a) {
  constexpr int NV = HD / 64;         // output dims per lane
  constexpr int ITEMS = QPW * GROUP;  // (query, head) pairs per wave
  __shared__ float v_lds[64 * HD];
  const AttnSeg seg = a.segs[blockIdx.z];
  const int kvh = blockIdx.y;
  const int q_block0 = blockIdx.x * (4 * QPW);
  if (q_block0 >= seg.len) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q_wave0 = q_block0 + wave * QPW;
  const KVT* kbase = reinterpret_cast<const KVT*>(a.k) + seg.kv_off + (int64_t)kvh * a.kv_hs;
  const KVT* vbase = reinterpret_cast<const KVT*>(a.v) + seg.kv_off + (int64_t)kvh * a.kv_hs;

  float qreg[ITEMS][NV], oacc[ITEMS][NV], mrun[ITEMS], lrun[ITEMS];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int qi = q_wave0 + it / GROUP, h = kvh * GROUP + it % GROUP;
    mrun[it] = -INFINITY;
    lrun[it] = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      oacc[it][v] = 0.f;
      qreg[it][v] = (qi < seg.len) ? a.q[(size_t)(seg.q_row0 + qi) * a.q_rs + h * HD + v * 64 + lane] : 0.f;
    }
  }
  const int block_qmax = min(q_block0 + 4 * QPW, seg.len) - 1;
  const int wave_qmax = min(q_wave0 + QPW, seg.len) - 1;  // < q_wave0 when the wave has no query
  const int n_keys_block = CAUSAL ? block_qmax + 1 : seg.len;
  const int n_kgroups = (n_keys_block + 63) / 64;

  for (int kg = 0; kg < n_kgroups; ++kg) {
    const int key0 = kg * 64;
    __syncthreads();  // previous group's V fully consumed
    for (int i = tid; i < 64 * HD / 4; i += 256) {
      const int r = i / (HD / 4), c4 = i % (HD / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (key0 + r < seg.len) {
        const KVT* p = vbase + (int64_t)(key0 + r) * a.kv_rs + c4 * 4;
        v.x = KvIo<KVT>::load(p); v.y = KvIo<KVT>::load(p + 1); v.z = KvIo<KVT>::load(p + 2); v.w = KvIo<KVT>::load(p + 3);
      }
      reinterpret_cast<float4*>(v_lds)[i] = v;
    }
    __syncthreads();
    const bool wave_active = (wave_qmax >= q_wave0) && (!CAUSAL || key0 <= wave_qmax);
    if (!wave_active) continue;

    const int kj = key0 + lane;
    float kreg[HD];
    if (kj < seg.len) {
      RowLoad<KVT>::template load<HD>(kbase + (int64_t)kj * a.kv_rs, kreg);
    } else {
#pragma unroll
      for (int d = 0; d < HD; ++d) kreg[d] = 0.f;
    }
    float sc[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) sc[it] = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      const float kd = kreg[d];
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) sc[it] += lane_bcast(qreg[it][d / 64], d % 64) * kd;
    }
    float pr[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int qi = q_wave0 + it / GROUP;
      const bool valid = (kj < seg.len) && (qi < seg.len) && (!CAUSAL || kj <= qi);
      const float s = valid ? sc[it] / a.scale_div : -INFINITY;
      const float gmax = wave_max(s);
      const float m_new = fmaxf(mrun[it], gmax);
      float alpha = 1.f, p = 0.f;
      if (m_new != -INFINITY) {
        alpha = expf(mrun[it] - m_new);  // exp(-inf) = 0 on the first block
        p = valid ? expf(s - m_new) : 0.f;
      }
      lrun[it] = lrun[it] * alpha + wave_sum(p);
      mrun[it] = m_new;
#pragma unroll
      for (int v = 0; v < NV; ++v) oacc[it][v] *= alpha;
      pr[it] = p;
    }
    const int jmax = min(64, seg.len - key0);
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      if (j < jmax) {
        float vv[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) vv[v] = v_lds[j * HD + v * 64 + lane];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
          const float pj = lane_bcast(pr[it], j);
#pragma unroll
          for (int v = 0; v < NV; ++v) oacc[it][v] += pj * vv[v];
        }
      }
    }
  }
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int qi = q_wave0 + it / GROUP, h = kvh * GROUP + it % GROUP;
    if (qi < seg.len) {
      const float inv = 1.0f / lrun[it];
#pragma unroll
      for (int v = 0; v < NV; ++v) a.o[(size_t)(seg.q_row0 + qi) * a.o_rs + h * HD + v * 64 + lane] = oacc[it][v] * inv;
    }
  }
}

template <int HD, int GROUP, bool CAUSAL, typename KVT, int QPW>
void launch_t(const AttnArgs& a, hipStream_t s) {
  dim3 grid((a.max_len + 4 * QPW - 1) / (4 * QPW), a.n_kv_heads, a.n_segs);
  hipLaunchKernelGGL((attn_kernel<HD, GROUP, CAUSAL, KVT, QPW>), grid, dim3(256), 0, s, a);
}

}  // namespace

const char* launch_attn_enc(const AttnArgs& a, hipStream_t s) {
  if (a.n_segs <= 0) return nullptr;
  launch_t<64, 1, false, float, 8>(a, s);
  return nullptr;
}

const char* launch_attn_prefill(const AttnArgs& a, int group, bool kv_f32, hipStream_t s) {
  if (a.n_segs <= 0) return nullptr;
  if (group == 1) {
    if (kv_f32) launch_t<128, 1, true, float, 4>(a, s); else launch_t<128, 1, true, uint16_t, 4>(a, s);
  } else if (group == 2) {
    if (kv_f32) launch_t<128, 2, true, float, 4>(a, s); else launch_t<128, 2, true, uint16_t, 4>(a, s);
  } else if (group == 4) {
    if (kv_f32) launch_t<128, 4, true, float, 2>(a, s); else launch_t<128, 4, true, uint16_t, 2>(a, s);
  } else {
    return "attn: GQA group must be 1, 2 or 4";
  }
  return nullptr;
}

}  // namespace q3a
