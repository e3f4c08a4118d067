// Batched decode step: the qkv projection (skinny MFMA GEMM, k_skinny.hip) and the attention (k_dattn.hip) of a group of sequences
// as ONE launch, ordered by dispatch (round 6).
//
// Replaces, per decoder layer of the batched greedy step, the boundary between Linear::forward of q / k / v (src/layers.rs:295-297)
// and the rest of TextAttention::forward (layers.rs:303-335).  Why: the attention launch streams 60 MB of K / V rows and is
// HBM-bound while it streams (5.9 TB/s), but it cannot request a row before it has started, and it cannot start before the
// projection has finished: the projection's 3.8 us span, the 1.7 us boundary and the attention's own 3.9 us head (first tiles in
// flight, nothing to score them against) are 9 us per layer in which HBM is nearly idle.  Here
//   * workgroups [0, NQ) are the projection's 16-row tiles (skinny_body<..., PUB>: the arithmetic of skinny_kernel, bit-identical
//     rows), workgroups NQ .. NQ + 8 S are the attention's (kv head, sequence) pairs (decode_attn_batched_body<..., FUSED>);
//   * nothing waits for a workgroup with a HIGHER index: the projection never waits, an attention workgroup waits only for the 32
//     projection tiles of its kv head, which the dispatcher has started before it (workgroups are dispatched in index order; the
//     tiles of kv head h are the workgroups b with b % 8 = h and its consumers are NQ + h + 8 s: under the observed placement
//     (block b -> XCD b % 8) producers and consumers of a head share one XCD dispatcher, so the order argument holds per XCD
//     and no circular wait can form between several engines on one GPU) -- and the wait is bounded: a lost arrival fails the
//     call, it cannot hang the GPU;
//   * the attention requests its first key tiles at entry, so its cache-row stream runs under the projection;
//   * both kinds fit a CU together: <= 128 VGPRs (key tiles of 64, two in flight) and 60 KiB of LDS each.
// The hand-off is placement-independent (cdna_hip_programming.md Guideline 16): projection rows leave with agent-scope
// (write-through) stores, every storing wave drains them, ONE lane bumps the kv head's counter (one 64-B line per counter) with an
// agent-scope RMW; the consumer polls that word with relaxed agent-scope loads and reads the rows with agent-scope loads (served past
// its L1, which may hold the previous layer's rows of the same buffer).  Counters are zeroed by argmax_finalize at the end of every
// step.
#define Q3A_BODY_ONLY 1
#include "k_skinny.hip"
#include "k_dattn.hip"

namespace q3a {
namespace {

constexpr int FUSE_TILE = 64, FUSE_RING = 2;  // attention key tiles: 119 VGPRs, so that a projection and an attention workgroup share a CU

template <int GROUP, int SH, int UNR>
__global__ __launch_bounds__(SK_WAVES * 64) void qkv_dattn_batched_kernel(SkinnyArgs q, DecodeAttnArgs a, unsigned* ready, int cnt_stride, unsigned* err) {
  static_assert(SK_WAVES == DA_WAVES, "one block size");
  constexpr int TQ = GROUP * 8;          // 16-row tiles of a kv head's q rows; + 8 of k + 8 of v
  const int NQ = 8 * (TQ + 16);          // projection workgroups (8 kv heads)
  const int b = blockIdx.x;
  if (b < NQ) {
    const int h = b & 7, j = b >> 3;     // kv head (= XCD under the observed placement), tile of its TQ + 16
    const int tile = j < TQ ? TQ * h + j : (j < TQ + 8 ? a.n_q * 8 + 8 * h + (j - TQ) : (a.n_q + a.n_kv) * 8 + 8 * h + (j - TQ - 8));
    skinny_body<false, 1, SH, 3, UNR, true, false, false, false, true>(q, tile, ready + h * (cnt_stride & 1023));
    return;
  }
  const int c = b - NQ;
  decode_attn_batched_body<GROUP, uint16_t, FUSE_TILE, FUSE_RING, true>(a, c & 7, c >> 3, FuseWait{ready + (c & 7) * (cnt_stride & 1023), cnt_stride >= 1024 ? 0u : (unsigned)(TQ + 16), err});  // (cnt_stride >= 1024: timing experiment, no wait, wrong rows)
}

}  // namespace

// q: the qkv projection in its pre-normalised form (xw16f + ss_parts), mode 0, fp32 out = a.qkv; a: the batched attention of the
// same S sequences (bf16 cache).  ready: 8 counters cnt_stride words apart, zeroed before the step; err: sticky time-out count.
const char* launch_qkv_dattn_batched(const SkinnyArgs& q, const DecodeAttnArgs& a, int S, unsigned* ready, int cnt_stride, unsigned* err, hipStream_t s) {
  const int group = a.n_kv ? a.n_q / a.n_kv : 0;
  if (a.n_kv != 8 || (group != 1 && group != 2 && group != 4) || a.n_q != group * 8) return "qkv_dattn_batched: 8 kv heads with 1, 2 or 4 query heads each";
  if (S < 1 || S > 32 || q.S != S) return "qkv_dattn_batched: 1..32 sequences, the same for both halves";
  if (!q.xw16f || !q.ss_parts || q.mode != 0 || !q.out || q.out != a.qkv || q.N != (a.n_q + 2 * a.n_kv) * 128 || q.out16 || q.rms_w || q.x16)
    return "qkv_dattn_batched: the projection must be the pre-normalised mode-0 form writing the attention's qkv rows";
  if (!a.out && !a.out16) return "qkv_dattn_batched: no attention output buffer";
  if (a.out16 && a.out_frag && S > 32) return "qkv_dattn_batched: fragment order holds at most 32 sequences";
  if (a.max_ctx % 128 != 0) return "qkv_dattn_batched: max_ctx must be a multiple of 128";
  if (!ready || !err || cnt_stride < 1) return "qkv_dattn_batched: counters required";
  const int steps = q.K / 32;
  if (q.K % 256 != 0 || (steps / SK_WAVES) % 4 != 0) return "qkv_dattn_batched: K must be a multiple of 1024";
  const dim3 grid(q.N / 16 + 8 * S), block(SK_WAVES * 64);
  const size_t dyn = (size_t)SK_WAVES * 1 * (4 / 2) * 2 * 1024;  // UNR = 4 k-steps per pass: 4 KiB per wave
#define Q3A_QDB(G)                                                                                                                  \
  do {                                                                                                                              \
    if (S <= 16) hipLaunchKernelGGL((qkv_dattn_batched_kernel<G, 1, 4>), grid, block, dyn, s, q, a, ready, cnt_stride, err);        \
    else hipLaunchKernelGGL((qkv_dattn_batched_kernel<G, 2, 4>), grid, block, dyn, s, q, a, ready, cnt_stride, err);                \
  } while (0)
  if (group == 1) Q3A_QDB(1);
  else if (group == 2) Q3A_QDB(2);
  else Q3A_QDB(4);
#undef Q3A_QDB
  return nullptr;
}

}  // namespace q3a
