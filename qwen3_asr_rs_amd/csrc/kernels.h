// Launch wrappers of the gfx950 kernels.  Every launch goes to the caller's HIP stream; nothing here
// allocates or synchronises (graph-capture safe).  A non-null return value is a static error string.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#ifndef Q3A_STAMP_FIELD  // (dev.h defines it for device code; host-only translation units get the same layout)
#ifdef Q3A_STAMP
#define Q3A_STAMP_FIELD unsigned long long* stamp = nullptr;
#else
#define Q3A_STAMP_FIELD
#endif
#endif

namespace q3a {

// ---- GEMM (k_gemm.hip) -----------------------------------------------------------------------------
struct GemmEpilogue {
  float* out = nullptr;          // [rows][ldo] fp32
  uint16_t* out16 = nullptr;     // alternative bf16 output (k_gemm16.hip only): feeds the next GEMM's LDS-DMA directly
  int ldo = 0;
  const float* bias = nullptr;   // [N] or null
  int act = 0;                   // 0 none, 1 erf-GELU
  const float* resid = nullptr;  // [rows][ldo] added after the activation (may alias out)
  const int* rowmap = nullptr;   // GEMM row m -> output row (negative: drop the row); null = identity
  const float* addend = nullptr; // [addend_period][ldo] added before the activation (positional embedding)
  int addend_period = 1;
  // k_gemm16.hip, M <= 32 (lm_head of a batched decode step): per output row the maximum of each 64-column tile and its
  // column (first index on ties) -> part_val / part_idx[row * part_stride + tile]; `out` may then be null (no logits stored)
  float* part_val = nullptr;
  int* part_idx = nullptr;
  int part_stride = 0;
  Q3A_STAMP_FIELD
};
// Y = X[M][K](fp32, row stride lda) . W[N][K]^T(bf16).  glu: W rows are [16 gate|16 up] blocks, out has N/2 columns.
const char* launch_gemm(const float* X, int lda, const uint16_t* W, int M, int N, int K, const GemmEpilogue& ep,
                        bool glu, bool split, hipStream_t s);
// 3x3 / stride 2 / pad 1 convolution as implicit GEMM over an NHWC fp32 map [imgs][H][W][C]; W bf16 [Cout][3][3][C].
const char* launch_conv3x3s2_gemm(const float* X, int imgs, int H, int Wd, int C, const uint16_t* Wt, int Cout,
                                  const GemmEpilogue& ep, bool split, hipStream_t s);
void launch_gemm_ref(const float* X, const uint16_t* W, float* Y, int M, int N, int K, hipStream_t s);
// bf16-activation versions (k_gemm16.hip, default mode): both operands go HBM -> LDS with global_load_lds.
const char* launch_gemm16(const uint16_t* X, int lda, const uint16_t* W, int M, int N, int K, const GemmEpilogue& ep,
                          bool glu, hipStream_t s);
// zero_page: >= 128 B of zeros (padded filter taps read it)
const char* launch_conv3x3s2_gemm16(const uint16_t* X, const uint16_t* zero_page, int imgs, int H, int Wd, int C,
                                    const uint16_t* Wt, int Cout, const GemmEpilogue& ep, hipStream_t s);
// 256 x 256 x 64, 8-wave, counted-vmcnt version of the two above for batch-sized problems (k_gemm256.hip); launch_gemm16 /
// launch_conv3x3s2_gemm16 dispatch to it when gemm256_eligible (enough 256 x 256 tiles, K % 64 == 0, K >= 128)
bool gemm256_eligible(int M, int N, int K);
// launch_gemm16 without the dispatch to gemm256 (the 4-wave tiles of k_gemm16.hip): small problems and gemm256's split remainder
const char* launch_gemm16_small(const uint16_t* X, int lda, const uint16_t* W, int M, int N, int K, const GemmEpilogue& ep,
                                bool glu, hipStream_t s);
// Process-wide A/B knobs (engine.cpp).  Initialised from the environment exactly once (std::call_once) and atomics
// afterwards: q3a_group_transcribe drives one host thread per GPU through the code that reads them, q3a_debug_set writes them.
struct Knobs {
  std::atomic<int> gemm256_min_tiles{128};      // Q3A_GEMM256_MIN_TILES: 256 x 256 tiles from which gemm256 is used (k_gemm256.hip)
  std::atomic<int> gemm256_persist{1};          // Q3A_GEMM256_PERSIST: gemm256 launches min(tiles, CUs) workgroups that walk the tiles, the next tile's first K tile arriving under the epilogue (0 = one workgroup per tile, n > 1 = exactly n workgroups; bit-identical)
  std::atomic<int> gemm256_group_m{0};          // Q3A_GEMM256_GROUP_M: gemm256's tile order -- groups of this many tile rows, M fastest inside a group (1 = N fastest over the whole matrix; 0 = 8 where the matrix is at least 8 tiles wide, else 1); the same tiles, another assignment to workgroups: bit-identical
  std::atomic<int> dattn_batched_min_wgs{128};  // Q3A_DATTN_BATCHED_MIN_WGS: S * n_kv from which launch_decode_attn_batched is used
  std::atomic<int> decode_group_size{0};        // Q3A_DECODE_GROUP: sequences per group of the batched decode step (0 = 32)
  std::atomic<int> decode_parallel_groups{1};   // Q3A_DECODE_PARALLEL: groups as parallel stream / graph branches
  std::atomic<int> fuse_qkrope{1};              // Q3A_FUSE_QKROPE: QK-norm + RoPE + cache append as the qkv GEMM's epilogue
  std::atomic<int> skinny_q{1};                 // Q3A_SKINNY_Q: quarter workgroups for the o / down projections
  std::atomic<int> eos_run_ahead{1};            // Q3A_EOS_RUN_AHEAD: decode steps kept enqueued ahead of the device in natural-EOS mode.  1: exactly the steps needed are executed; paired with a fixed-N run of the same engine (profiles/r5_eos_run_ahead_ab.txt, two processes): 1 costs +0.03 / +1.24 ms per 100 tokens, 2 costs +2.07 / +1.86 ms (one wasted step + the same launch latency), 3 +1.3 / +1.2, 4 +3.3 / +3.1
  std::atomic<int> skinny_glu_hp3{1};           // Q3A_SKINNY_GLU_HP3: gate/up skinny GEMM as 3 half-pair tiles per workgroup when the pair form has more workgroups than CUs (k_skinny.hip HP; 0 = off, 2 = whenever the shape allows)
  // Round 6 removed nine knobs together with the code only they selected (docs/HISTORY.md "Pruned in round 6"; last present at commit
  // caf7a05): fuse_qkv_attn, dattn_pair_split, fattn_pipe, rope_variant, rope_twice (measured alternatives that lost / finished debug
  // aids), gemm16_ring, gemm256_resid_prefetch, live_key_splits, skinny_glu_2pass (the winning form is now the only form).
};
Knobs& knobs();
const char* launch_gemm256(const uint16_t* X, int lda, const uint16_t* W, int M, int N, int K, const GemmEpilogue& ep,
                           bool glu, hipStream_t s);
const char* launch_conv3x3s2_gemm256(const uint16_t* X, const uint16_t* zero_page, int imgs, int H, int Wd, int C,
                                     const uint16_t* Wt, int Cout, const GemmEpilogue& ep, hipStream_t s);
// fp32 -> bf16 (round to nearest even), n elements
const char* launch_to_bf16(const float* x, uint16_t* y, size_t n, hipStream_t s);
// bf16 -> fp32 (debug taps of bf16 intermediates)
const char* launch_from_bf16(const uint16_t* x, float* y, size_t n, hipStream_t s);

// ---- log-mel front end (k_mel.hip) -------------------------------------------------------------------
struct MelBatch {
  const float* pcm;            // concatenated utterances
  const int64_t* pcm_off;      // [B] sample offset of each utterance (device)
  const int64_t* n_samples;    // [B] (device)
  const int64_t* mel_off;      // [B] float offset of each (128, F_b) block (device)
  const int* n_frames;         // [B] (device)
  float* mel;                  // output, concatenated (128, F_b) blocks
  unsigned* gmax_key;          // [B] order-preserving key of the per-utterance max (zeroed by the launcher)
};
const char* launch_mel(const MelBatch& mb, int B, int max_frames, const float* dft /*[400][416]*/,
                       const float* filt_t /*[202][128]*/, hipStream_t s);

// ---- conv stem first layer (k_conv1.hip) ---------------------------------------------------------------
struct ChunkTable {
  const int* chunk_utt;    // [C] utterance of each chunk (device)
  const int* chunk_frame0; // [C] first mel frame of the chunk inside its utterance (device)
};
// mel (128, F_b) blocks -> NHWC fp32 [chunk][64][W/2][Cout] = gelu(conv3x3 s2 p1 + bias), chunk zero-padded to `chunk_frames`
const char* launch_conv1(const float* mel, const int64_t* mel_off, const int* n_frames, const ChunkTable& ct,
                         int n_chunks, int n_mels, int chunk_frames, const float* w /*[Cout][9]*/,
                         const float* b, int Cout, float* out, hipStream_t s, uint16_t* out16 = nullptr);

// ---- norms (k_norm.hip) ----------------------------------------------------------------------------------
const char* launch_layernorm(const float* x, const float* w, const float* b, float* y, int rows, int D, float eps,
                             hipStream_t s, uint16_t* y16 = nullptr);  // y16 != null: write bf16 there instead of y
const char* launch_rmsnorm(const float* x, const float* w, float* y, int rows, int D, float eps, hipStream_t s,
                           uint16_t* y16 = nullptr);

// ---- attention over independent segments (k_attn.hip) ------------------------------------------------------
// One segment = a set of `len` queries/keys that attend to each other (encoder windows, non-causal) or a
// whole decoder sequence (causal).  Element (row j, head h, dim d):
//   Q: q + (q_row0 + j) * q_rs + h * HD + d                          (fp32)
//   K: k + kv_off + kvh * kv_hs + j * kv_rs + d   (V likewise)       (KVT = float or bf16)
//   O: o + (q_row0 + j) * o_rs + h * HD + d                          (fp32)
struct AttnSeg {
  int q_row0;
  int len;
  int64_t kv_off;
};
struct AttnArgs {
  const float* q; int q_rs;
  const uint16_t* q16;  // MFMA flash attention only: when non-null, Q is read from here as bf16 (same strides) instead of q
  const void* k; const void* v; int64_t kv_hs; int kv_rs;
  float* o; int o_rs;
  uint16_t* o16;    // MFMA flash attention only: when non-null the output is written here as bf16 (same strides)
  const AttnSeg* segs; int n_segs; int max_len;
  int n_kv_heads;
  float scale_div;  // scores are divided by this (sqrt(head_dim)), as the reference does
};
const char* launch_attn_enc(const AttnArgs& a, hipStream_t s);                            // HD=64, 1 q-head per kv head, non-causal, fp32 K/V
const char* launch_attn_prefill(const AttnArgs& a, int group, bool kv_f32, hipStream_t s);  // HD=128, causal, GQA group 1/2/4
// MFMA flash-attention versions of the two above (k_fattn.hip): bf16 operands, default mode only
const char* launch_fattn_enc(const AttnArgs& a, hipStream_t s);                 // fp32 K/V rounded to bf16 while staging
const char* launch_fattn_prefill(const AttnArgs& a, int group, hipStream_t s);  // bf16 KV cache

// ---- decoder glue (k_decode.hip) ----------------------------------------------------------------------------
// hidden[r] = embed[ids[r]] for rows whose id is not `skip_id` (audio rows are written by the encoder tail)
const char* launch_embed(const int* ids, int rows, const uint16_t* embed, int H, int skip_id, float* out, hipStream_t s);
struct RopeKvArgs {
  float* qkv;               // [rows][qkv_dim] fp32; q part is normalised + rotated in place
  const int* row_seq;       // [rows] sequence of the row
  const int* row_pos;       // [rows] position inside the sequence
  const float* q_norm; const float* k_norm; float eps;
  const float* cos_t; const float* sin_t;  // [max_pos][64]
  void* kcache; void* vcache;              // this layer: [S][n_kv][max_ctx][128]
  int n_q, n_kv, max_ctx;
  uint16_t* q16;            // non-null (default mode, MFMA attention): q is written HERE as bf16 [rows][n_q*128] -- the value the
                            // attention kernel rounds it to on load anyway -- instead of in place as fp32 (half the bytes twice)
};
const char* launch_qknorm_rope_kv(const RopeKvArgs& a, int rows, bool kv_f32, hipStream_t s);
// rows [0, M1) of an M x N problem that launch_gemm256 / launch_gemm256_qkrope give to the 256 x 256 kernel when they split off
// the trailing rows (0: no split)
int gemm256_split_rows(int M, int N);
// qkv projection (bf16 X [M][K] . W[(n_q + 2 n_kv) * 128][K]^T + bias) with the kernel above as its epilogue: q -> a.q16, k / v
// -> bf16 KV cache; a.qkv is not touched.  256 x 256 tiles (k_gemm256.hip): call when gemm256_eligible(M, N, K).
const char* launch_gemm256_qkrope(const uint16_t* X, int lda, const uint16_t* W, int M, int K, const float* bias,
                                  const RopeKvArgs& a, hipStream_t s);
// dst[i] = src[row_idx[i]]
const char* launch_gather_rows(const float* src, const int* row_idx, int n, int D, float* dst, hipStream_t s);
// dst[dst_row[i]] = src[i]  (negative dst_row: skip)
const char* launch_scatter_rows(const float* src, const int* dst_row, int n, int D, float* dst, hipStream_t s);

// GEMV family: y[b][n] = sum_k xn[b][k] * W[n][k], b < NB <= 4, x fp32 staged in LDS, W bf16 streamed once.
struct GemvArgs {
  const float* x; int ldx;      // [NB][K]
  const float* rms_w;           // non-null: x is RMS-normalised with this weight first (eps below)
  float eps;
  const uint16_t* W; int N; int K;
  const float* bias;            // [N] or null
  int mode;                     // 0: store, 1: out = resid + y, 2: GLU (W rows in [16 gate|16 up] blocks; N = 2*inter),
                                // 3: logits (out nullable) + per-block argmax partials
  float* out; int ldo;
  const float* resid;
  float* part_val; int* part_idx; int part_stride;  // mode 3: [NB][part_stride], entry = blockIdx.x
  // optional: x = attention output merged on the fly from the flash-decoding partials of launch_decode_attn
  // (x/ldx ignored; K must equal attn_heads*128)
  const float* attn_pm; const float* attn_pl; const float* attn_po; int attn_nsplit; int attn_heads;
  int attn_fast_exp;            // merge weights with the hardware exponential (default mode) instead of expf
  int fast_math;                // default mode: hardware rsq / exp / rcp in the RMSNorm scale and SiLU of the epilogue (dev.h rstd_of)
  Q3A_STAMP_FIELD
};
const char* launch_gemv(const GemvArgs& a, int NB, hipStream_t s);
constexpr int GEMV_ATTN_MAX_TABLE = 1024;  // min(NB,4) * heads * nsplit must fit (else merge with launch_attn_combine first)
int gemv_blocks(const GemvArgs& a);        // grid size launch_gemv uses (= number of argmax partials in mode 3)
int gemv_rows_per_wave(const GemvArgs& a);

// Skinny MFMA GEMM (k_skinny.hip): 4 < S <= 32 sequences, weights streamed once.
struct SkinnyArgs {
  const float* x; int ldx; int S;   // [S][K] fp32 ...
  const uint16_t* x16;              // ... or, when non-null, the same as bf16 (default mode: written by the producer)
  int x16_frag;                     // x16 is in MFMA-fragment order (skinny_frag_index) instead of row-major [S][K]
  int out16_frag;                   // out16 is written in that order (for the next skinny GEMM)
  const float* rms_w; float eps;    // non-null (fp32 x only): RMSNorm fused -- x*w enters the MFMA, rstd scales the result
  const uint16_t* W; int N; int K;  // bf16 [N][K]
  const float* bias;                // [N] or null
  int mode;                         // 0: store, 1: out = resid + y, 2: GLU ([16 gate|16 up] row blocks, out has N/2 columns)
  float* out; int ldo;
  uint16_t* out16;                  // mode 2 only: when non-null the result is written here as bf16 instead of out
  const float* resid;
  // Pre-normalised input (default mode, replaces x + rms_w): xw16f = bf16(x * w_norm) in fragment order and
  // ss_parts[p][32] = partial sums of x^2 (p < ss_nparts) -- both written by the kernel that produced x (below), so
  // this GEMM's activation loads are lane-linear and the RMSNorm still costs no launch.  eps as above.
  const uint16_t* xw16f; const float* ss_parts; int ss_nparts;
  // mode 1 only, producer side of the above for the NEXT GEMM (null: off): after out = resid + y is stored,
  // next_xw16f[frag(s, n)] = bf16(out * next_w[n]) and next_ss[blockIdx.x][s] = sum over the block's 16 columns of out^2
  const float* next_w; uint16_t* next_xw16f; float* next_ss;
  // mode 1, bf16 fragment-order x, N % 64 == 0: "quarter" workgroups of 8 rows x 16 sequences (k_skinny.hip); next_ss then has
  // N / 8 partial rows instead of N / 16 -- the caller sizes its buffers and the consumer's ss_nparts accordingly
  int qsplit;
  int qs_halves;  // (set by the launcher: 16-sequence halves per row tile)
  int glu_hp3;    // gate/up as 256 workgroups x 3 half-pair tiles (k_skinny.hip HP): 1 = when that balances the CUs (set by the engine from knob skinny_glu_hp3), 0 = never, 2 = whenever the shape allows
  int n_cu;       // compute units of the device the launch goes to (0: 256) -- chooses between the gate/up forms; set by the engine
  int fast_math;  // default mode: hardware rsq / exp / rcp in the RMSNorm scale and SiLU of the epilogue (dev.h rstd_of)
  Q3A_STAMP_FIELD
};
// the same producer duty for kernels that write a whole row of the residual stream (token embedding)
struct NextNormOut {
  const float* next_w;      // [H] norm weight of the GEMM that consumes the row next (null: off)
  uint16_t* next_xw16f;     // [32 * H] fragment order
  float* next_ss;           // [nparts][32]: row 0 receives the row's sum of squares, rows 1..nparts-1 zero
  int nparts;
  // more than 32 sequences: sequence s belongs to group s / 32, whose buffers start group_stride_* elements further on
  long group_stride_x, group_stride_ss;
  int group_size;  // sequences per group (0 means 32)
};
const char* launch_skinny(const SkinnyArgs& a, bool split, hipStream_t s);
const char* skinny_init();  // once per device before the first launch_skinny (sets the large-LDS kernel attributes)
// Fragment order of a bf16 activation matrix for up to 32 sequences: the 16 B that lane (sequence s & 15, k-chunk
// (k >> 3) & 3) of a v_mfma_f32_16x16x32_bf16 B operand needs for k-step k >> 5 and sequence half s >> 4 sit at
// lane-linear offsets, so one wave load is 1 KiB contiguous (8 cache lines) instead of 16 rows x 64 B (16 lines).
// Buffers in this order always hold 32 sequences: 32 * K elements.
__host__ __device__ inline size_t skinny_frag_index(int s, int k) {
  return ((((size_t)(k >> 5) * 2 + (s >> 4)) * 4 + ((k >> 3) & 3)) * 16 + (s & 15)) * 8 + (k & 7);
}

struct DecodeAttnArgs {
  const float* qkv;            // [S][qkv_dim] fp32 (raw projections of the current token)
  const int* pos;              // [S] number of tokens already in the cache = position of the current token
  const float* q_norm; const float* k_norm; float eps;
  const float* rope_cur;       // [S][128]: cos (64) | sin (64) of position pos[s], maintained by launch_argmax_finalize
  void* kcache; void* vcache;  // this layer
  float* pm; float* pl;       // [S][n_q][nsplit] partial softmax max / sum per key split
  float* po;                   // [S][n_q][nsplit][128] unnormalised partial outputs
  int nsplit;                  // >= ceil(max_ctx / dattn_keys_per_split(kv_f32))
  int n_q, n_kv, max_ctx;
  float scale_div;
  // launch_decode_attn_batched only: when the batch alone fills the chip (S * n_kv workgroups), one workgroup walks ALL
  // keys of its (sequence, kv head) and writes the normalised context itself -- no partials, no merge launch:
  float* out;                  // [S][n_q*128] fp32 (precise mode) ...
  uint16_t* out16;             // ... or bf16 (default mode), row-major or, with out_frag, in skinny_frag_index order
  int out_frag;
  int trim_prologue;           // batched kernel: 1 = some sequence of the batch is shorter than the two prologue key tiles, trim them to the live rows too
  Q3A_STAMP_FIELD
};
#ifndef Q3A_DATTN_SPLIT_KEYS
#define Q3A_DATTN_SPLIT_KEYS 128  // keys per flash-decoding split of the one-sequence path (A/B builds: 64)
#endif
constexpr int DATTN_KEYS_PER_SPLIT_BF16 = Q3A_DATTN_SPLIT_KEYS, DATTN_KEYS_PER_SPLIT_F32 = 128;
inline int dattn_keys_per_split(bool kv_f32) { return kv_f32 ? DATTN_KEYS_PER_SPLIT_F32 : DATTN_KEYS_PER_SPLIT_BF16; }
const char* launch_decode_attn(const DecodeAttnArgs& a, int S, bool kv_f32, hipStream_t s);
// one workgroup per (sequence, kv head), online softmax over 128-key tiles, final output written directly (a.out / a.out16)
const char* launch_decode_attn_batched(const DecodeAttnArgs& a, int S, bool kv_f32, hipStream_t s);
// out[S][n_q*128] = merged partials (needed as its own launch only on the GEMM decode path)
const char* launch_attn_combine(const float* pm, const float* pl, const float* po, int nsplit, int S, int n_q, float* out,
                                hipStream_t s, uint16_t* out16 = nullptr, bool frag = false);
// out16 != null: bf16 there instead of out; frag: in skinny_frag_index order (S <= 32)

struct FinalizeArgs {
  const float* part_val;   // [S][part_stride] block-partial maxima ...
  const int* part_idx;     // ... and their vocabulary indices
  int part_stride, n_part;
  int V;
  int* next_tok;           // [S] token to feed next (written)
  int* out_ids;            // [S][out_stride] generated ids (written at step_count[s])
  int out_stride;
  int* step_count;         // [S]
  int* pos;                // [S] += advance
  int advance;
  uint8_t* done;           // [S] sticky: set when the argmax is EOS
  int* n_done;             // device counter of sequences whose done flag is set (nullable); n_seq = sequences in the batch
  int n_seq;
  int* host_progress;      // pinned host words the greedy loop's host side polls WITHOUT synchronising the stream (nullable):
                           // [0] = finalize launches completed for sequence 0 (prefill counts as 1), [1] = 1 once every
                           // sequence has produced its EOS (src/inference.rs:163-165 evaluated for the whole batch on the device)
  const uint16_t* embed; int H;
  float* x_next;           // [S][H] embedding of the chosen token
  int eos0, eos1;
  const float* cos_t; const float* sin_t;  // RoPE tables [max_pos][64]
  float* rope_cur;         // [S][128] (written): cos | sin row of the updated pos[s] (DecodeAttnArgs::rope_cur); nullable
  NextNormOut nn;          // pre-normalised copy of x_next for the first layer's qkv GEMM (skinny path)
};
// block partials of logits [S][V] (GEMM decode path; the GEMV lm_head produces its own)
const char* launch_argmax_partials(const float* logits, int V, int S, float* pval, int* pidx, int stride, int nblk,
                                   hipStream_t s);
const char* launch_argmax_finalize(const FinalizeArgs& a, int S, hipStream_t s);
// x_next[s] = embed[tok[s]]; next_tok[s] = tok[s]  (teacher forcing)
const char* launch_set_tokens(const int* tok, int S, const uint16_t* embed, int H, float* x_next, int* next_tok, hipStream_t s,
                              const NextNormOut& nn = NextNormOut{});

}  // namespace q3a
