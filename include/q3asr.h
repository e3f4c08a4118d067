/* q3asr.h -- C ABI of libq3asr_hip.so: the MI355X (gfx950) backend for the Qwen3-ASR hot path of
 * second-state/qwen3_asr_rs.  Plain pointers and sizes only; no C++ or torch types cross this line.
 *
 * Each entry point names the reference interface it replaces (file:line in the reference tree).
 * A Rust `feature = "hip"` arm binds these with `extern "C"` exactly as src/backend/mlx/ffi.rs binds
 * mlx-c (see INTEGRATION.md for the stub).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; q3a_last_error() returns the message
 *     (the reference's forward paths are infallible/panic, src/tensor.rs:228,232; its load paths
 *     return anyhow::Result, src/inference.rs:34-65 -- the shim maps non-zero to panic!/bail!).
 *   - an engine handle is thread-compatible (one host thread per handle / per GPU).
 *   - host buffers are caller-owned and never retained after return.
 *   - "utterance" = one independent clip; a batch of B utterances is the data-parallel unit
 *     (new functionality whose semantics equal running the reference B times).
 */
#ifndef Q3ASR_H
#define Q3ASR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct q3a_engine q3a_engine;

/* Engine options. Zero-initialise, then override. */
typedef struct q3a_opts {
  int32_t precise;        /* 0: activations enter the MFMA as bf16, bf16 KV cache (default)
                             1: activations split into bf16 hi+lo pairs (2 MFMAs per tile), fp32 KV cache:
                                near-fp32 logits, used to separate kernel bugs from bf16 rounding        */
  int32_t max_new_tokens; /* generation cap per utterance (reference: 4096, src/inference.rs:153); 0 -> 4096 */
  int32_t use_graph;      /* 1: replay the decode step from a captured hipGraph (default 1 when 0/unset -> see q3a_opts_default) */
  int32_t debug_taps;     /* 1: keep per-stage intermediate tensors readable through q3a_debug_read      */
  int32_t valu_attention; /* 1: use the fp32 VALU attention kernels in default mode too (A/B against the
                             MFMA flash-attention kernels; precise mode always uses them)                 */
  int32_t reserved[11];
} q3a_opts;

/* Fill `o` with the defaults (precise=0, max_new_tokens=4096, use_graph=1, debug_taps=0). */
void q3a_opts_default(q3a_opts* o);

/* Model dimensions as parsed from config.json (replaces AsrConfig::from_file, src/config.rs:115-121). */
typedef struct q3a_dims {
  int32_t enc_d_model, enc_layers, enc_heads, enc_ffn, num_mel_bins, n_window, n_window_infer,
      conv_channels, enc_output_dim, max_source_positions;
  int32_t vocab_size, hidden_size, intermediate_size, dec_layers, num_q_heads, num_kv_heads, head_dim,
      tie_word_embeddings, mrope_interleaved;
  int32_t mrope_section[4];
  float rms_norm_eps;
  double rope_theta;
} q3a_dims;

/* ---- load -------------------------------------------------------------------------------------- */

/* Device selection of the CLI (src/main.rs:51-65: tch::Cuda::is_available / init_mlx): number of HIP devices this process
 * can use, 0 when there is none (the library has no CPU path).  Never fails. */
int32_t q3a_device_count(void);

/* AsrInference::load (src/inference.rs:30-86): parse config.json, read model.safetensors or the
 * sharded index (src/weights.rs:10-58), build the bf16 device weight arena on GPU `device`. */
int32_t q3a_engine_create(const char* model_dir, int32_t device, const q3a_opts* opts, q3a_engine** out);

/* Weight arena for multi-GPU replicas: pack on one rank, broadcast the bytes over RCCL/xGMI, and
 * create every replica from its device copy (no file reads on the replicas besides config.json). */
int32_t q3a_arena_bytes(const char* model_dir, uint64_t* bytes);
int32_t q3a_arena_pack(const char* model_dir, void* host_dst, uint64_t bytes);
/* `device_arena` stays owned by the caller and must outlive the engine. */
int32_t q3a_engine_create_from_arena(const char* model_dir, int32_t device, void* device_arena, uint64_t bytes,
                                     const q3a_opts* opts, q3a_engine** out);

void q3a_engine_destroy(q3a_engine* e);
/* Last error of `e` (or of the calling thread when e == NULL, e.g. after a failed create). */
const char* q3a_last_error(const q3a_engine* e);
int32_t q3a_get_dims(const q3a_engine* e, q3a_dims* out);
/* 1 when the checkpoint held F16 / F32 matrices: the arena stores every matrix as bf16, so those were rounded to nearest-even
 * (the reference widens them to f32 instead, src/weights.rs:74-89,134-181).  0 for the published BF16 checkpoints, whose
 * bf16 -> f32 widening is exact: arena storage is lossless there.  Vectors (norm weights, biases, conv1) are kept in f32. */
int32_t q3a_weights_rounded(const q3a_engine* e);

/* ---- shape helpers (host integer arithmetic) --------------------------------------------------------- */
/* mel frames for n samples: ceil(n/160) (src/mel.rs:51,83-84). */
int64_t q3a_num_frames(int64_t n_samples);
/* AudioEncoder::get_output_length (src/audio_encoder.rs:269-279). */
int32_t q3a_num_audio_tokens(const q3a_engine* e, int64_t n_frames);
/* AsrInference::build_prompt (src/inference.rs:215-257). Writes 15 + num_audio_tokens + n_prefix ids;
 * `ids` may be NULL to query the length (returned through *len). */
int32_t q3a_build_prompt(int32_t num_audio_tokens, const int32_t* lang_prefix_ids, int32_t n_prefix,
                         int32_t* ids, int32_t* len);

/* ---- stage API (device state chains from one stage to the next) -------------------------------------- */

/* WhisperFeatureExtractor::extract (src/mel.rs:49-96) for B utterances.
 * pcm16k: host f32, utterances concatenated; n_samples[B].  mel_out (nullable): host f32, per
 * utterance a (num_mel_bins, F_b) row-major block, concatenated.  n_frames_out (nullable): [B]. */
int32_t q3a_mel(q3a_engine* e, const float* pcm16k, const int64_t* n_samples, int32_t B, float* mel_out,
                int32_t* n_frames_out);

/* AudioEncoder::forward (src/audio_encoder.rs:79-169) on the mel of the last q3a_mel call.
 * audio_embeds_out (nullable): host f32 [sum T_b][enc_output_dim].  T_out (nullable): [B]. */
int32_t q3a_encode(q3a_engine* e, float* audio_embeds_out, int32_t* T_out);

/* Steps 5-7 of AsrInference::transcribe (src/inference.rs:110-149): embed `ids`, overwrite the
 * <|audio_pad|> rows with the encoder output of the last q3a_encode, RoPE tables, causal prefill.
 * ids: concatenated prompts, lens[B].  last_logits_out (nullable): host f32 [B][vocab] (last row
 * only; the reference computes all rows and keeps the last, src/inference.rs:156).
 * next_ids (nullable): [B] greedy argmax (first-index tie-break). */
int32_t q3a_prefill(q3a_engine* e, const int32_t* ids, const int32_t* lens, int32_t B, float* last_logits_out,
                    int32_t* next_ids);

/* One iteration of the greedy loop (src/inference.rs:160-200) for all B sequences: feeds the
 * previous argmax (or the ids set by q3a_set_next_tokens), returns the new argmax.
 * done[b]=1 once sequence b has produced an EOS {151643,151645} (the reference breaks out of its
 * loop there, src/inference.rs:163-165).  logits_out nullable host f32 [B][vocab]. */
int32_t q3a_decode_step(q3a_engine* e, int32_t* next_ids, uint8_t* done, float* logits_out);

/* Teacher forcing for parity tests: override the token fed by the next q3a_decode_step. */
int32_t q3a_set_next_tokens(q3a_engine* e, const int32_t* ids, int32_t B);

/* ---- whole path ---------------------------------------------------------------------------------- */

/* Copy B utterances (host f32 16 kHz, concatenated) into the engine's HBM input buffer. */
int32_t q3a_upload_pcm(q3a_engine* e, const float* pcm16k, const int64_t* n_samples, int32_t B);

/* Steps 2-8 of AsrInference::transcribe (src/inference.rs:95-200) on the resident batch: mel ->
 * encoder -> prompt/injection -> prefill -> greedy decode.  lang_prefix_ids (nullable) =
 * tokenizer.encode("language {Lang}") when the language is forced (src/inference.rs:246-251).
 * fixed_new_tokens > 0: run exactly that many decode iterations ignoring EOS (throughput mode);
 * 0: stop every sequence at its first EOS, capped by max_new (<= opts.max_new_tokens). */
int32_t q3a_run_resident(q3a_engine* e, const int32_t* lang_prefix_ids, int32_t n_prefix, int32_t max_new,
                         int32_t fixed_new_tokens);

/* Generated ids of the last q3a_run_resident: out_ids host [B][stride], out_lens [B] (EOS excluded,
 * as generated_ids in src/inference.rs:167). */
int32_t q3a_fetch_ids(q3a_engine* e, int32_t* out_ids, int32_t stride, int32_t* out_lens);

/* upload + run + fetch: AsrInference::transcribe steps 2-8 (src/inference.rs:95-200) for B utterances, host PCM in,
 * generated ids on the host out -- the window SURVEY.md section 8d times.  The input copy is overlapped with the front
 * end: the (pageable) caller buffers are copied by a few host threads into a pinned mirror of the device layout and shipped
 * in pieces on a copy stream; the log-mel kernel of a piece's utterances runs as soon as that piece has landed. */
int32_t q3a_transcribe_batch(q3a_engine* e, const float* pcm16k, const int64_t* n_samples, int32_t B,
                             const int32_t* lang_prefix_ids, int32_t n_prefix, int32_t max_new,
                             int32_t fixed_new_tokens, int32_t* out_ids, int32_t stride, int32_t* out_lens);
/* The same with one pointer per utterance (pcm16k[b] -> n_samples[b] floats): what a caller that decoded B files into B
 * buffers has (src/audio.rs:7 returns one Vec<f32> per file) -- no concatenation on the host. */
int32_t q3a_transcribe_batch_ptrs(q3a_engine* e, const float* const* pcm16k, const int64_t* n_samples, int32_t B,
                                  const int32_t* lang_prefix_ids, int32_t n_prefix, int32_t max_new,
                                  int32_t fixed_new_tokens, int32_t* out_ids, int32_t stride, int32_t* out_lens);
/* Where the input side of the last q3a_transcribe_batch[_ptrs] call spent its time. */
typedef struct q3a_io_timings {
  float stage_ms; /* host clock: entry -> last piece copied to pinned memory and its H2D copy + log-mel enqueued */
  float h2d_ms;   /* hipEvents on the copy stream: first piece issued -> last piece landed */
  float wall_ms;  /* host clock of the whole call (upload + hot path + ids on the host) */
  int32_t pieces, threads; /* H2D pieces (runs of utterances) and host copy threads used */
  int32_t mode;   /* 1: staged + overlapped (default); 0: Q3A_UPLOAD_MODE=0, one pageable copy per utterance on the compute stream */
} q3a_io_timings;
int32_t q3a_io_timings_last(const q3a_engine* e, q3a_io_timings* out);

/* ---- multi-GPU: one process, one host thread per GPU ------------------------------------ */
/* The reference is single-device (src/main.rs:51-65 picks ONE device); a batch of independent utterances is new
 * functionality whose result equals running AsrInference::transcribe (src/inference.rs:89) once per utterance. */
typedef struct q3a_group q3a_group;
/* Load once, replicate to n_gpus GPUs: the checkpoint is read and packed on the host ONCE, uploaded to devices[0] and
 * shipped to the other GPUs with ONE ncclBroadcast of the whole weight arena over xGMI (RCCL, ncclCommInitAll; librccl is
 * dlopen'ed here).  devices == NULL means 0..n_gpus-1.  Every GPU then owns a q3a_engine created from its copy. */
int32_t q3a_group_create(const char* model_dir, int32_t n_gpus, const int32_t* devices, const q3a_opts* opts, q3a_group** out);
void q3a_group_destroy(q3a_group* g);
int32_t q3a_group_size(const q3a_group* g);
int32_t q3a_group_used_rccl(const q3a_group* g);  /* 1 when the arena went through ncclBroadcast */
/* Start-up stage times of q3a_group_create in seconds: out4 = {pack (read + lay out the checkpoint into pinned host memory),
 * upload (one asynchronous H2D copy to the first GPU), broadcast (RCCL init + ONE ncclBroadcast of the arena), engines}. */
int32_t q3a_group_startup_seconds(const q3a_group* g, double* out4);
const char* q3a_group_last_error(const q3a_group* g);
q3a_engine* q3a_group_engine(q3a_group* g, int32_t rank);  /* borrowed handle of rank's engine (stage API, timings) */
/* Static contiguous split of n_items utterances: rank gets [*begin, *end) (the first n_items % world_size ranks one more). */
void q3a_group_partition(int32_t n_items, int32_t world_size, int32_t rank, int32_t* begin, int32_t* end);
/* q3a_transcribe_batch over the group: utterances are partitioned with q3a_group_partition, every GPU runs its slice from
 * its own host thread, results land in the caller's arrays in utterance order.  No collective on the data path. */
int32_t q3a_group_transcribe(q3a_group* g, const float* pcm16k, const int64_t* n_samples, int32_t B,
                             const int32_t* lang_prefix_ids, int32_t n_prefix, int32_t max_new, int32_t fixed_new_tokens,
                             int32_t* out_ids, int32_t stride, int32_t* out_lens);
/* The same with one pointer per utterance (see q3a_transcribe_batch_ptrs). */
int32_t q3a_group_transcribe_ptrs(q3a_group* g, const float* const* pcm16k, const int64_t* n_samples, int32_t B,
                                  const int32_t* lang_prefix_ids, int32_t n_prefix, int32_t max_new, int32_t fixed_new_tokens,
                                  int32_t* out_ids, int32_t stride, int32_t* out_lens);

/* ---- measurement ---------------------------------------------------------------------------------- */

typedef struct q3a_timings {
  float mel_ms, encoder_ms, prefill_ms, decode_ms, total_ms; /* hipEvent times of the last q3a_run_resident */
  int32_t decode_steps, batch, total_audio_tokens, total_prompt_tokens;
} q3a_timings;
int32_t q3a_stage_timings(const q3a_engine* e, q3a_timings* out);

/* Per-kernel-class timing of ONE decode step, each launch bracketed by hipEvents on the engine's
 * stream (eager, not graph).  Requires a batch in decode state (after q3a_run_resident/q3a_prefill).
 * class ids: see Q3A_KC_* ; arrays have Q3A_KC_COUNT entries. */
enum { Q3A_KC_GEMV = 0 /* qkv and gate/up GEMVs (2 weight rows per wave) */, Q3A_KC_DECODE_ATTN = 1, Q3A_KC_ARGMAX = 2,
       Q3A_KC_GEMM = 3, Q3A_KC_NORM = 4, Q3A_KC_OTHER = 5, Q3A_KC_GEMV_O = 6, Q3A_KC_GEMV_DOWN = 7,
       Q3A_KC_GEMV_LM_HEAD = 8, Q3A_KC_COUNT = 9 };
typedef struct q3a_kernel_profile {
  float total_us[Q3A_KC_COUNT];
  int32_t launches[Q3A_KC_COUNT];
  double weight_bytes[Q3A_KC_COUNT]; /* algorithmic weight bytes streamed by the class in the step */
} q3a_kernel_profile;
int32_t q3a_profile_decode_step(q3a_engine* e, q3a_kernel_profile* out);

/* Back-to-back timing of the dominant decode kernel (the qkv and gate/up GEMVs, gemv1_kernel<2,..>): `reps`
 * sweeps over ALL decoder layers' qkv and gate/up matrices (0.59 GB at 0.6B, larger than the 256 MB
 * Infinity Cache, so every launch streams from HBM as in a real step), captured in a hipGraph and replayed between
 * HIP events on the engine's stream (average of 3 replays; an eager host loop would time the host).
 * avg_us = elapsed / launches; bytes_per_launch = algorithmic weight bytes per launch.
 * Needs decode state with <= 2 sequences (the GEMV path); does not change it. */
int32_t q3a_profile_weight_stream(q3a_engine* e, int32_t reps, float* avg_us, double* bytes_per_launch, int32_t* launches);

/* Measured denominators for the roofline fractions (SURVEY.md section 8d "Peaks to divide by: measure on the box"; csrc/k_peaks.hip):
 * a read-only HBM stream over 2 GiB (the access pattern of the decode-step weight streams), a 1 GiB device copy and a triad
 * (bytes counted on every stream they touch), and the library's own 256x256x64 bf16 GEMM on 8192^3, once on constant and once on random operands -- each the best of `reps`
 * launches between two HIP events.  Needs 2 GiB of free device memory; not on the product path. */
typedef struct q3a_peaks {
  double hbm_read_gbps, hbm_copy_gbps, hbm_triad_gbps; /* GB/s (1e9 bytes per second) */
  double hbm_read_bytes;                                /* bytes one read sweep covers */
  double mfma_bf16_tflops;                              /* 2 M N K / time, operands = one constant (the chip's clock stays high: the ceiling) */
  int32_t gemm_m, gemm_n, gemm_k, n_cu, reps, reserved;
  double mfma_bf16_tflops_random;                       /* the same GEMM on random operands (sign + mantissa bits toggling: what real data draws; the chip clocks down) */
} q3a_peaks;
int32_t q3a_measure_peaks(int32_t device, int32_t reps, q3a_peaks* out);

/* Debug taps (opts.debug_taps=1): copy a named intermediate to host. `bytes` = capacity of dst;
 * *actual receives the tap size. Names: mel conv1 conv2 conv3 enc_in enc_layer0 enc_last
 * audio_embeds dec_embed dec_layer0 dec_last_hidden logits. */
int32_t q3a_debug_read(q3a_engine* e, const char* name, void* dst, uint64_t bytes, uint64_t* actual);

/* ---- pipeline shell: host-only helpers around the hot path -------------- */

/* load_audio (src/audio.rs:7): RIFF/WAVE PCM s8/s16/s24/s32/f32 -> mono (channel mean, audio.rs:192-200) -> f32
 * scaled by 1/2^(bits-1) (audio.rs:177) -> `target_sr` with this backend's deterministic windowed-sinc polyphase
 * resampler (the reference's FFmpeg / rubato resamplers are not reproducible outside their code bases).
 * *samples_out is malloc'ed; release it with q3a_free. */
int32_t q3a_load_audio(const char* path, int32_t target_sr, float** samples_out, int64_t* n_out);
int32_t q3a_resample(const float* in, int64_t n, int32_t sr_in, int32_t sr_out, float** samples_out, int64_t* n_out);
/* The reference's WAV-fallback resampler (src/audio.rs:220-245: rubato SincFixedIn, sinc_len 256, f_cutoff 0.95, Linear,
 * oversampling 256, BlackmanHarris2, one `process` call over the whole clip), restated from the crate's published
 * algorithm (rubato 0.16.2 is not vendored in the reference and cannot be built here): same filter family and
 * parameters, output n at input time (n+1)/ratio, the last ~129 input samples produce no output.  q3a_load_audio uses it
 * when the environment has Q3A_RESAMPLER=rubato. */
int32_t q3a_resample_rubato(const float* in, int64_t n, int32_t sr_in, int32_t sr_out, float** samples_out, int64_t* n_out);
void q3a_free(void* p);

/* AsrTokenizer (src/tokenizer.rs:4-50) over HuggingFace's tokenizer.json (byte-level BPE). */
typedef struct q3a_tokenizer q3a_tokenizer;
int32_t q3a_tokenizer_create(const char* tokenizer_json_path, q3a_tokenizer** out);
void q3a_tokenizer_destroy(q3a_tokenizer* t);
/* decode(ids, skip_special_tokens) -> UTF-8 (tokenizer.rs:42-49). *len = bytes needed (excluding NUL). */
int32_t q3a_tokenizer_decode(const q3a_tokenizer* t, const int32_t* ids, int32_t n, int32_t skip_special, char* out,
                             int32_t cap, int32_t* len);
/* encode(text, add_special_tokens = false) (tokenizer.rs:33-39): added tokens cut out first, then the normaliser
 * tokenizer.json names (NFC for Qwen), the Qwen2 pre-tokenisation pattern over Unicode code points and byte-level BPE.  UTF-8 in. */
int32_t q3a_tokenizer_encode(const q3a_tokenizer* t, const char* text, int32_t* ids, int32_t cap, int32_t* n);
/* The normaliser step on its own: Unicode NFC (UAX #15; decomposition / composition data of Unicode 13.0 -- later additions
 * pass through unchanged).  *len = bytes needed (excluding NUL). */
int32_t q3a_normalize_nfc(const char* utf8, char* out, int32_t cap, int32_t* len);

/* parse_asr_output (src/inference.rs:276-305); capitalize_first (inference.rs:307-313). */
int32_t q3a_parse_asr_output(const char* raw, int32_t language_forced, char* language, int32_t language_cap, char* text,
                             int32_t text_cap);
int32_t q3a_capitalize_first(const char* s, char* out, int32_t cap);

/* A/B knobs for kernel experiments (process-wide atomics, read from the environment once; not part of the reference
 * interface).  The knobs that shape the decode step are latched per batch at the next prefill and are part of the captured
 * graph's signature, so changing one on a live engine re-captures instead of replaying a stale graph.  Keys:
 *   "gemm256_min_tiles"  minimum number of 256x256 output tiles for which the bf16 GEMM dispatches to the 8-wave
 *                        counted-vmcnt kernel (k_gemm256.hip): 0 = whenever the shape allows, a huge value = never.
 *   "gemm256_persist"    1 (default): a gemm256 launch is min(tiles, CUs) workgroups that walk the tiles of their XCD's chunk, the next
 *                        tile's first K tile arriving under the epilogue; 0: one workgroup per tile; n > 1: a walk of exactly n workgroups (tests).
 *                        Same arithmetic: bit-identical.
 *   "gemm256_group_m"    gemm256's tile order: groups of this many tile rows, M fastest inside a group (1 = N fastest over the whole
 *                        matrix; 0, the default = 8 where the matrix is at least 8 tiles wide, else 1).  The same tiles assigned to other workgroups: bit-identical.
 *   "dattn_batched_min_wgs"  sequences x kv heads of a decode group from which the batched decode step uses the
 *                        one-workgroup-per-(sequence, kv head) attention kernel (k_dattn.hip) instead of key splits + merge.
 *   "decode_group_size"  sequences per group of the batched decode step (1..32; 0 = 32), taken at the next prefill.
 *   "decode_parallel_groups"  1: groups run as parallel stream / hipGraph branches (default), 0: one after the other.
 *   "fuse_qkrope"        1 (default): batch-sized prefills run QK-norm + RoPE + the KV-cache append as the epilogue of the qkv
 *                        GEMM; 0: as the separate kernel (taken at the next prefill).
 *   "skinny_q"           1 (default): o / down projections of the batched decode step as 8-row x 16-sequence workgroups;
 *                        0: 16 rows x 32 sequences (taken at the next engine / batch set-up: it sizes a buffer).
 *   "eos_run_ahead"      decode steps the natural-EOS greedy loop keeps enqueued ahead of the device (default 1): the stop
 *                        condition is evaluated on the device and read from pinned host memory without synchronising.
 *   "skinny_glu_hp3"     1 (default): when the gate/up projection of the batched decode step has more 32-row pair tiles than the GPU has
 *                        CUs and its output columns divide into 3 half-pair tiles (8 gate + 8 up rows in one MFMA fragment) per
 *                        workgroup with at most one workgroup per CU (hidden 2048 / inter 6144: 256 workgroups), it runs in that
 *                        form: every CU streams the same number of weight bytes and the activations once; 0: never; 2: whenever
 *                        the shape allows (tests).  Same arithmetic.  Taken at the next prefill.
 * Round 6 removed the keys whose A/B is settled, together with the code only they selected (docs/HISTORY.md "Pruned in round 6"):
 * fuse_qkv_attn, dattn_pair_split, fattn_pipe, rope_variant, rope_twice, gemm16_ring, gemm256_resid_prefetch, live_key_splits,
 * skinny_glu_2pass.  An unknown key returns non-zero. */
int32_t q3a_debug_set(const char* key, int32_t value);

/* Kernel self-tests against naive device references (no model needed): returns max abs error. */
int32_t q3a_selftest_gemm(int32_t device, int32_t M, int32_t N, int32_t K, int32_t split, float* max_abs_err,
                          float* ref_abs_max);
/* Same for the bf16-activation GEMM (global_load_lds path); with reps > 0 also the average launch time of it and
 * of the fp32-activation kernel on the same shape (HIP events around `reps` back-to-back launches). */
int32_t q3a_selftest_gemm16(int32_t device, int32_t M, int32_t N, int32_t K, int32_t reps, float* max_abs_err,
                            float* ref_abs_max, float* avg_us_bf16, float* avg_us_f32);

#ifdef __cplusplus
}
#endif
#endif /* Q3ASR_H */
