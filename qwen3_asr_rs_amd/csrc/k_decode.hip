// Decoder glue + the single-token (decode) kernels for gfx950.
//
//   embed_kernel            Tensor::embedding (src/tensor.rs:209-213, src/text_decoder.rs:90-92)
//   qknorm_rope_kv_kernel   per-head RMSNorm on q/k (src/layers.rs:303-304), RoPE x*cos+rotate_half(x)*sin
//                           (src/layers.rs:361-375) and the KV-cache append (src/layers.rs:311-319) -- into a
//                           pre-allocated contiguous cache instead of the reference's per-step `cat`
//   (the GEMV family lives in k_gemv.hip, the single-token attention in k_dattn.hip)
//   argmax_finalize_kernel  argmax (first-index tie-break, src/tensor.rs:370-372), EOS flag, id store and the
//                           embedding of the chosen token for the next step -- no per-token D2H sync
//                           (the reference does int64_value per token, src/inference.rs:161)
#include "dev.h"
#include "kernels.h"

namespace q3a {

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

// x_next[s] = embedding row `id` (fp32).  With nn.next_w set (skinny decode path) the row is also handed to the next
// GEMM pre-normalised: bf16(x * w_norm) in fragment order plus its sum of squares (kernels.h NextNormOut), so that GEMM
// neither re-reads the fp32 row with 16-line fragment loads nor needs a norm launch.  `red`: shared, one float per wave.
__device__ __forceinline__ void embed_row(const uint16_t* __restrict__ embed, int id, int H, float* __restrict__ x_next, int s,
                                          const NextNormOut& nn, float* red) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const uint2* src = reinterpret_cast<const uint2*>(embed + (size_t)id * H);
  float4* dst = reinterpret_cast<float4*>(x_next + (size_t)s * H);
  const int gs = nn.group_size > 0 ? nn.group_size : 32;
  const int sl = s % gs;  // position inside its group of sequences (fragment-order buffers hold one group each)
  uint16_t* const xw = nn.next_xw16f + (size_t)(s / gs) * nn.group_stride_x;
  float* const nss = nn.next_ss + (size_t)(s / gs) * nn.group_stride_ss;
  float ss = 0.f;
  for (int i = tid; i < H / 4; i += nthr) {
    const uint2 v = src[i];
    const float4 f = make_float4(bf16lo(v.x), bf16hi(v.x), bf16lo(v.y), bf16hi(v.y));
    dst[i] = f;
    if (nn.next_w) {
      const float4 w = reinterpret_cast<const float4*>(nn.next_w)[i];
      ss += f.x * f.x + f.y * f.y + f.z * f.z + f.w * f.w;
      uint2 pk;  // k = 4i .. 4i+3 are consecutive in fragment order
      pk.x = pack_bf16x2(f.x * w.x, f.y * w.y);
      pk.y = pack_bf16x2(f.z * w.z, f.w * w.w);
      *reinterpret_cast<uint2*>(xw + skinny_frag_index(sl, 4 * i)) = pk;
    }
  }
  if (nn.next_w) {  // kernel-argument condition: uniform
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int w = 0; w < (nthr + 63) / 64; ++w) t += red[w];
      nss[sl] = t;
    }
    for (int p = 1 + tid; p < nn.nparts; p += nthr) nss[(size_t)p * 32 + sl] = 0.f;
  }
}

__global__ void set_tokens_kernel(const int* __restrict__ tok, const uint16_t* __restrict__ embed, int H,
                                  float* __restrict__ x_next, int* __restrict__ next_tok, NextNormOut nn) {
  __shared__ float red[16];
  const int s = blockIdx.x;
  const int id = tok[s];
  if (threadIdx.x == 0) next_tok[s] = id;
  embed_row(embed, id, H, x_next, s, nn, red);
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
    if (a.q16) {
      uint16_t* q = a.q16 + (size_t)row * a.n_q * 128 + (size_t)hv * 128;
      q[lane] = (uint16_t)f32_to_bf16_bits(x1);
      q[lane + 64] = (uint16_t)f32_to_bf16_bits(x2);
    } else {
      base[lane] = x1;
      base[lane + 64] = x2;
    }
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
  // Everything that does not depend on the chosen token is requested up front, in ONE batch: the sequence's counters, the partial
  // maxima (up to FIN_PRE per lane, clamped addresses -- round 6: the plain loop `for (i = tid; i < n_part; i += 1024)` compiled to
  // two loads and a vmcnt(0) per iteration, ten dependent L2 round trips for the 9496 partials of the one-sequence lm_head) and,
  // behind `pos`, the RoPE row of the next position.
  constexpr int FIN_PRE = 10;  // 10 240 partials (vocabulary 151 936 in 16-row blocks: 9496); more are folded by the loop below
  const int np = a.pos[s] + a.advance;
  const int sc = a.step_count[s];
  const int was_done = a.done[s];
  float pre_v[FIN_PRE];
  int pre_i[FIN_PRE];
#pragma unroll
  for (int j = 0; j < FIN_PRE; ++j) {
    const int i = tid + j * 1024, ic = i < a.n_part ? i : a.n_part - 1;
    pre_v[j] = pv[ic];
    pre_i[j] = pi[ic];
  }
  float rope_v = 0.f;
  if (a.rope_cur && tid < 128) rope_v = tid < 64 ? a.cos_t[(size_t)np * 64 + tid] : a.sin_t[(size_t)np * 64 + tid - 64];
  float best = -INFINITY;
  int idx = 0x7fffffff;
#pragma unroll
  for (int j = 0; j < FIN_PRE; ++j) {
    const float v = tid + j * 1024 < a.n_part ? pre_v[j] : -INFINITY;
    const int ix = pre_i[j];
    if (v > best || (v == best && ix < idx)) { best = v; idx = ix; }
  }
  for (int i = tid + FIN_PRE * 1024; i < a.n_part; i += 1024) {
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
    if (sc < a.out_stride) a.out_ids[(size_t)s * a.out_stride + sc] = idx;
    a.step_count[s] = sc + 1;
    a.pos[s] = np;
    if ((idx == a.eos0 || idx == a.eos1) && !was_done) {  // this sequence's first EOS (inference.rs:163-165)
      a.done[s] = 1;
      if (a.n_done) {
        const int n = atomicAdd(a.n_done, 1) + 1;
        if (n == a.n_seq && a.host_progress) __hip_atomic_store(a.host_progress + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (s == 0 && a.host_progress) __hip_atomic_store(a.host_progress, sc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  // RoPE row of the position the next decode step works at, at a fixed address: the attention kernels of that step
  // request it together with q/k/v instead of after a dependent load of pos
  if (a.rope_cur && tid < 128) a.rope_cur[(size_t)s * 128 + tid] = rope_v;
  embed_row(a.embed, tok_s, a.H, a.x_next, s, a.nn, bv);  // bv: its 16 floats are free again after the barrier above
}

}  // namespace

const char* launch_embed(const int* ids, int rows, const uint16_t* embed, int H, int skip_id, float* out,
                         hipStream_t s) {
  if (rows <= 0) return nullptr;
  hipLaunchKernelGGL(embed_kernel, dim3(rows), dim3(256), 0, s, ids, embed, H, skip_id, out);
  return nullptr;
}
const char* launch_set_tokens(const int* tok, int S, const uint16_t* embed, int H, float* x_next, int* next_tok,
                              hipStream_t s, const NextNormOut& nn) {
  if (nn.next_w && S > 32 && (nn.group_stride_x <= 0 || nn.group_stride_ss <= 0)) return "set_tokens: more than 32 sequences need group strides";
  hipLaunchKernelGGL(set_tokens_kernel, dim3(S), dim3(256), 0, s, tok, embed, H, x_next, next_tok, nn);
  return nullptr;
}
const char* launch_qknorm_rope_kv(const RopeKvArgs& a, int rows, bool kv_f32, hipStream_t s) {
  if (rows <= 0) return nullptr;
  const long nvec = (long)rows * (a.n_q + 2 * a.n_kv);
  const int blocks = (int)((nvec + 3) / 4);
  if (kv_f32) hipLaunchKernelGGL((qknorm_rope_kv_kernel<float>), dim3(blocks), dim3(256), 0, s, a, rows);
  else hipLaunchKernelGGL((qknorm_rope_kv_kernel<uint16_t>), dim3(blocks), dim3(256), 0, s, a, rows);
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
