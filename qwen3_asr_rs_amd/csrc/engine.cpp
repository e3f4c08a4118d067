// Engine: owns the weight arena, the workspace and the HIP stream of one GPU and sequences the
// kernels of the hot path.  Exposes the C ABI of include/q3asr.h.
//
// Pipeline per batch of B independent utterances (reference: AsrInference::transcribe,
// src/inference.rs:89-200, run B times):
//   pcm -> k_mel -> k_conv1 -> conv2/conv3 (implicit GEMM) -> conv_out GEMM (+pos-emb, valid-token gather)
//       -> encoder layers {LN, QKV GEMM, window attention, out GEMM+res, LN, fc1 GEMM+GELU, fc2 GEMM+res}
//       -> ln_post, proj1+GELU, proj2 -> audio embeds -> scattered into the prompt embeddings
//       -> decoder prefill layers {RMSNorm, QKV GEMM, qk-norm+RoPE+KV append, causal GQA attention,
//          o GEMM+res, RMSNorm, gate/up GEMM+SiLU*up, down GEMM+res} -> last-row lm_head -> argmax
//       -> decode steps (GEMV path for 1-2 sequences, skinny MFMA GEMM for 3-32, tiled GEMM above), replayed
//          from a hipGraph.
// Residual streams / projections fp32 in HBM, GEMM-input activations bf16 in the default mode (fp32 in precise mode);
// weights bf16; KV cache bf16 (fp32 in precise mode).
#include <hip/hip_runtime.h>

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/q3asr.h"
#include "kernels.h"
#include "model.h"

using namespace q3a;

namespace {

thread_local std::string g_last_error;

#define HIPCHK(expr)                                                                                    \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess)                                                                               \
      fail(std::string("HIP error: ") + hipGetErrorString(_e) + " at " + __FILE__ + ":" + std::to_string(__LINE__) + " (" #expr ")"); \
  } while (0)
#define KCHK(expr)                                  \
  do {                                              \
    const char* _m = (expr);                        \
    if (_m) fail(std::string("kernel launch: ") + _m); \
  } while (0)

constexpr int kAudioPad = 151676, kEos0 = 151643, kEos1 = 151645;  // src/tokenizer.rs:52-59
// Decode-step projections: fp32-x GEMV kernels up to this many sequences, the skinny MFMA GEMM above.  Measured
// (0.6B, 100 tokens): 2 sequences 96.6 ms on the GEMV path; 4 sequences 141 ms on it against 111 ms for FIVE
// sequences on the skinny path.
constexpr int kGemvMaxSeq = 2;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool grew = false;
  void ensure(size_t bytes) {
    if (bytes <= cap) return;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = (bytes + 255) & ~size_t(255);
    HIPCHK(hipMalloc(&p, want));
    cap = want;
    grew = true;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

template <typename T>
void upload(DevBuf& b, const std::vector<T>& v, hipStream_t s) {
  b.ensure(std::max<size_t>(v.size() * sizeof(T), 16));
  if (!v.empty()) HIPCHK(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
}

struct ProfEvent {
  int kclass;
  double bytes;
  hipEvent_t a, b;
};

}  // namespace

namespace q3a {
void set_thread_error(const std::string& msg) { g_last_error = msg; }  // used by host_abi.cpp
}  // namespace q3a
// A/B knobs (kernels.h Knobs): read from the environment exactly once, atomics afterwards (q3a_debug_set writes them while
// q3a_group_transcribe may have one host thread per GPU reading them)
namespace q3a {
Knobs& knobs() {
  static Knobs k;
  static std::once_flag once;
  std::call_once(once, [] {
    auto env = [](const char* name, std::atomic<int>& dst) { if (const char* e = getenv(name)) dst.store(atoi(e)); };
    env("Q3A_GEMM256_MIN_TILES", k.gemm256_min_tiles);
    env("Q3A_GEMM256_PERSIST", k.gemm256_persist);
    env("Q3A_GEMM256_GROUP_M", k.gemm256_group_m);
    env("Q3A_DATTN_BATCHED_MIN_WGS", k.dattn_batched_min_wgs);
    env("Q3A_DECODE_GROUP", k.decode_group_size);
    env("Q3A_DECODE_PARALLEL", k.decode_parallel_groups);
    env("Q3A_FUSE_QKROPE", k.fuse_qkrope);
    env("Q3A_SKINNY_Q", k.skinny_q);
    env("Q3A_EOS_RUN_AHEAD", k.eos_run_ahead);
    env("Q3A_SKINNY_GLU_HP3", k.skinny_glu_hp3);
  });
  return k;
}
}  // namespace q3a

struct q3a_engine {
  Dims d;
  ArenaLayout L;
  q3a_opts opts{};
  int device = 0;
  hipStream_t stream = nullptr;
  uint8_t* arena = nullptr;
  bool own_arena = false;
  uint32_t arena_flags = 0;
  std::string err;

  // constant tables
  DevBuf dft, filt_t, pos_emb, rope_cos, rope_sin;
  int rope_positions = 0;

  // ---- batch description (host) ----
  int B = 0;
  std::vector<int64_t> n_samples, pcm_off, mel_off;
  std::vector<int> n_frames, n_chunks, T, tok_off;   // per utterance; tok_off = first encoder token
  int total_chunks = 0, total_T = 0, max_frames = 0;
  bool have_mel = false, have_enc = false, have_prefill = false;
  std::vector<int> P, seq_off;  // prompt lengths / first row
  int total_P = 0, max_ctx = 0, max_new = 0;
  std::vector<AttnSeg> enc_segs_h;
  int enc_max_seg = 0;

  // ---- device workspace ----
  DevBuf pcm, d_pcm_off, d_n_samples, d_mel_off, d_n_frames, mel, gmax, d_chunk_utt, d_chunk_frame0;
  DevBuf conv1, conv2, conv3, conv3_rowmap, convout_rowmap;
  DevBuf enc_x, enc_ln, enc_qkv, enc_ctx, enc_ffn, enc_segs, audio_embeds;
  DevBuf ids, audio_rowmap, row_seq, row_pos, dec_segs, last_rows;
  DevBuf dec_x, dec_ln, dec_qkv, dec_ctx, dec_act, kcache, vcache;
  DevBuf x_dec, d_pos, next_tok, out_ids, step_count, done, s_ln, s_qkv, s_ctx, s_act, logits, forced_tok, part_val, part_idx;
  DevBuf attn_pm, attn_pl, attn_po;
  DevBuf nn_x, nn_ss;  // pre-normalised residual row for the next skinny GEMM: [32 * hidden] bf16 fragment order, [hidden/16][32] f32
  DevBuf rope_cur;  // [B][128] cos|sin row of each sequence's current position (kept by argmax_finalize for decode attention)
  DevBuf enc_ctx16, dec_ctx16;  // opts.valu_attention in the default mode: bf16 copy of the fp32 attention context
  DevBuf dec_q16;
  int n_cu = 256;
  DevBuf n_done;    // device counter of sequences that have produced their EOS (argmax_finalize)
  int* host_prog = nullptr;      // pinned host words written by argmax_finalize (FinalizeArgs::host_progress), polled without a sync
  int* host_prog_dev = nullptr;  // the same words as the device sees them
  DevBuf zero_page;  // 256 B of zeros: padded filter taps of the bf16 implicit-GEMM convolutions read it
  int part_stride = 0, attn_nsplit = 0;
  size_t kv_layer_elems = 0;

  // ---- batched decode: sequence groups and their streams ----
  int gsize = 32;  // sequences per group of the batched decode step (<= 32: one skinny-GEMM weight sweep), fixed per batch
  // knobs that shape the decode step, latched per batch in setup_prompts: producers outside the captured graph (prefill
  // finalize, set_tokens) and the captured step must agree on them, and the graph signature names them
  int k_parallel_groups = 1, k_skinny_q = 1, k_dattn_batched_min_wgs = 128, k_skinny_glu_hp3 = 1;
  std::vector<hipStream_t> chain_streams;
  std::vector<hipEvent_t> join_ev;
  hipEvent_t fork_ev = nullptr;

  // ---- graph ----
  // instantiated decode-step graphs by signature (make_graph_sig): a batch shape met before -- or the same batch at a key-split
  // count met before -- replays without a new capture; oldest entry dropped beyond kGraphCacheMax
  static constexpr size_t kGraphCacheMax = 16;
  std::vector<std::pair<std::string, hipGraphExec_t>> graph_cache;
  // Key splits the one-sequence decode attention launches (k_dattn.hip decode_attn_kernel: one workgroup per kv head and
  // 128-key split).  Sized by what the caches HOLD, not by their capacity: pos_hi_ = longest prompt + decode steps enqueued
  // (host-side count, an upper bound of every sequence's position), so a generous max_new_tokens costs no empty splits.
  int pos_hi_ = 0, live_nsplit_ = 0;
  int min_P_ = 0;  // shortest prompt of the batch (decides DecodeAttnArgs::trim_prologue; part of the graph signature)

  // ---- input upload of q3a_transcribe_batch[_ptrs] (the window is host PCM -> ids on the host) ----
  // The caller's buffers are pageable: they are copied (several host threads for a batch-sized input) into a pinned
  // mirror of the device layout and shipped piece by piece on a copy stream of its own; the compute stream waits for a
  // piece's event and runs the log-mel kernel of exactly those utterances, so the front end runs under the rest of the
  // upload.  Q3A_UPLOAD_MODE=0 keeps the old form (one pageable hipMemcpyAsync per utterance on the compute stream).
  hipStream_t up_stream = nullptr;
  void* pin_p = nullptr;
  size_t pin_cap = 0;
  std::vector<hipEvent_t> up_ev;
  hipEvent_t up_t0 = nullptr, up_t1 = nullptr;
  q3a_io_timings io{};
  bool geom_valid = false;  // the device tables of set_batch describe n_samples (same lengths again: nothing to rebuild)

  // ---- measurement ----
  q3a_timings timings{};
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  std::vector<ProfEvent>* prof = nullptr;
  std::map<std::string, DevBuf> taps;
  std::map<std::string, size_t> tap_bytes;

  bool precise() const { return opts.precise != 0; }
  bool kv_f32() const { return opts.precise != 0; }
  size_t kv_elem() const { return kv_f32() ? 4 : 2; }
  // GEMM-input activations (conv maps, norm outputs, attention context, FFN hidden) are stored as bf16 in the default
  // mode -- exactly the values the MFMA consumes -- and as fp32 in the precise mode (hi+lo split inside the GEMM)
  size_t act_elem() const { return precise() ? 4 : 2; }
  template <typename T> const T* w(uint64_t off) const { return reinterpret_cast<const T*>(arena + off); }
  const float* wf(uint64_t off) const { return w<float>(off); }
  const uint16_t* wh(uint64_t off) const { return w<uint16_t>(off); }

  // opts.debug_taps: 1 = every tap; 2 = only the small ones a soak run compares (last hidden rows, head input)
  static bool light_tap(const char* name) { return strcmp(name, "dec_last_hidden") == 0 || strcmp(name, "head_in") == 0; }
  void tap(const char* name, const void* ptr, size_t bytes) {
    if (!opts.debug_taps || bytes == 0 || (opts.debug_taps == 2 && !light_tap(name))) return;
    DevBuf& t = taps[name];
    t.ensure(bytes);
    tap_bytes[name] = bytes;
    HIPCHK(hipMemcpyAsync(t.p, ptr, bytes, hipMemcpyDeviceToDevice, stream));
  }
  // tap of an activation buffer that is bf16 in the default mode and fp32 in the precise mode: always read back as fp32
  void tap_act(const char* name, const void* ptr, size_t elems) {
    if (!opts.debug_taps || elems == 0 || opts.debug_taps == 2) return;
    if (precise()) return tap(name, ptr, elems * 4);
    DevBuf& t = taps[name];
    t.ensure(elems * 4);
    tap_bytes[name] = elems * 4;
    KCHK(launch_from_bf16((const uint16_t*)ptr, t.as<float>(), elems, stream));
  }

  // GEMM over an activation buffer: LDS-DMA bf16 kernel (k_gemm16.hip) in the default mode, fp32 hi+lo split kernel
  // (k_gemm.hip) in the precise mode
  void act_gemm(const DevBuf& x, int lda, const uint16_t* W, int M, int N, int K, const GemmEpilogue& ep, bool glu) {
    if (precise()) KCHK(launch_gemm(x.as<float>(), lda, W, M, N, K, ep, glu, true, stream));
    else KCHK(launch_gemm16(x.as<uint16_t>(), lda, W, M, N, K, ep, glu, stream));
  }
  void act_conv(const DevBuf& x, int imgs, int H, int Wd, int C, const uint16_t* W, int Cout, const GemmEpilogue& ep) {
    if (precise()) KCHK(launch_conv3x3s2_gemm(x.as<float>(), imgs, H, Wd, C, W, Cout, ep, true, stream));
    else KCHK(launch_conv3x3s2_gemm16(x.as<uint16_t>(), zero_page.as<uint16_t>(), imgs, H, Wd, C, W, Cout, ep, stream));
  }
  // point a GEMM epilogue / producer at an activation buffer in the mode's storage type
  void act_out(GemmEpilogue& ep, const DevBuf& buf) const {
    if (precise()) ep.out = buf.as<float>(); else ep.out16 = buf.as<uint16_t>();
  }
  uint16_t* act16(const DevBuf& buf) const { return precise() ? nullptr : buf.as<uint16_t>(); }

  // run fn() (which enqueues exactly one kernel class) optionally bracketed by events
  template <class F> void timed(int kclass, double bytes, F&& fn) {
    if (!prof) {
      fn();
      return;
    }
    ProfEvent pe{kclass, bytes, nullptr, nullptr};
    HIPCHK(hipEventCreate(&pe.a));
    HIPCHK(hipEventCreate(&pe.b));
    HIPCHK(hipEventRecord(pe.a, stream));
    fn();
    HIPCHK(hipEventRecord(pe.b, stream));
    prof->push_back(pe);
  }

  // =====================================================================================
  void init_common(const std::string& model_dir, int dev, const q3a_opts* o) {
    if (o) opts = *o; else q3a_opts_default(&opts);
    if (opts.max_new_tokens <= 0) opts.max_new_tokens = 4096;
    d = parse_config_file(model_dir + "/config.json");
    validate_dims(d);
    L = plan_arena(d);
    device = dev;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev == 0)
      fail("no HIP device available: libq3asr_hip has no CPU fallback (hipGetDeviceCount: " + std::string(hipGetErrorString(e)) + ")");
    if (dev < 0 || dev >= n_dev) fail("device index out of range");
    HIPCHK(hipSetDevice(dev));
    KCHK(skinny_init());
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    for (auto& x : ev) HIPCHK(hipEventCreate(&x));
    HIPCHK(hipHostMalloc((void**)&host_prog, 64, hipHostMallocMapped));
    memset(host_prog, 0, 64);
    HIPCHK(hipHostGetDevicePointer((void**)&host_prog_dev, host_prog, 0));
    n_done.ensure(64);
  }

  void init_tables() {
    auto dftm = make_dft_matrix(400, 416);
    auto filt = make_mel_filterbank_T(d.n_mels, 400, 16000, 202);
    auto pe = make_sinusoid_rows(d.tokens_per_chunk(), d.enc_d);
    upload(dft, dftm, stream);
    upload(filt_t, filt, stream);
    upload(pos_emb, pe, stream);
    zero_page.ensure(256);
    HIPCHK(hipMemsetAsync(zero_page.p, 0, 256, stream));
    ensure_rope(8192);
    HIPCHK(hipStreamSynchronize(stream));
  }

  void ensure_rope(int n_pos) {
    if (n_pos <= rope_positions) return;
    int n = std::max(n_pos, 8192);
    std::vector<float> c, s;
    make_rope_tables(n, d.head_dim, d.rope_theta, c, s);
    HIPCHK(hipStreamSynchronize(stream));
    upload(rope_cos, c, stream);
    upload(rope_sin, s, stream);
    HIPCHK(hipStreamSynchronize(stream));
    rope_positions = n;
  }

  void check_arena_header() {
    ArenaHeader h;
    HIPCHK(hipMemcpy(&h, arena, sizeof(h), hipMemcpyDeviceToHost));
    if (h.magic != kArenaMagic || h.version != kArenaVersion) fail("weight arena: bad magic/version");
    if (h.total_bytes != L.total) fail("weight arena: size does not match config.json");
    arena_flags = h.flags;
  }

  // =====================================================================================
  // batch geometry (host integers; audio_encoder.rs:79-121, 263-279)
  void set_batch(const int64_t* ns, int b) {
    // a failure below must not leave the previous batch's device tables paired with new host geometry
    have_mel = have_enc = have_prefill = false;
    if (b <= 0) { B = 0; fail("batch must be >= 1"); }
    // the same lengths as the batch before (fixed-length serving windows; every repetition of a benchmark): all tables
    // below are functions of the lengths alone and are already on the device
    if (geom_valid && b == B && (int)n_samples.size() == b && std::equal(ns, ns + b, n_samples.begin())) return;
    geom_valid = false;
    B = 0;
    for (int u = 0; u < b; ++u)
      if (ns[u] < 161) fail("utterance too short: reflection padding needs more than 160 samples (src/mel.rs:63-65)");
    n_samples.assign(ns, ns + b);
    pcm_off.resize(b); mel_off.resize(b); n_frames.resize(b); n_chunks.resize(b); T.resize(b); tok_off.resize(b);
    std::vector<int> chunk_utt, chunk_frame0, convout_map, conv3_map;
    enc_segs_h.clear();
    int64_t po = 0, mo = 0;
    int tok = 0;
    total_chunks = 0; max_frames = 0; enc_max_seg = 0;
    const int cf = d.chunk_frames(), tpc = d.tokens_per_chunk(), cpw = d.chunks_per_window();
    for (int u = 0; u < b; ++u) {
      pcm_off[u] = po;
      po += (ns[u] + 3) & ~int64_t(3);
      int F = (int)((ns[u] + 159) / 160);  // mel.rs:51,83-84
      n_frames[u] = F;
      max_frames = std::max(max_frames, F);
      mel_off[u] = mo;
      mo += (int64_t)F * d.n_mels;
      int nc = (F + cf - 1) / cf;
      n_chunks[u] = nc;
      tok_off[u] = tok;
      std::vector<int> valid(nc);
      for (int c = 0; c < nc; ++c) {
        int frames = std::min(cf, F - c * cf);
        valid[c] = Dims::feat_len(frames);
        chunk_utt.push_back(u);
        chunk_frame0.push_back(c * cf);
        for (int t = 0; t < tpc; ++t) convout_map.push_back(t < valid[c] ? tok + t : -1);
        tok += valid[c];
      }
      T[u] = tok - tok_off[u];
      // attention windows (audio_encoder.rs:172-260): None (= one segment) when nc <= cpw
      int t0 = tok_off[u];
      if (nc <= cpw) {
        enc_segs_h.push_back(AttnSeg{t0, T[u], (int64_t)t0 * 3 * d.enc_d});
        enc_max_seg = std::max(enc_max_seg, T[u]);
      } else {
        for (int c0 = 0; c0 < nc; c0 += cpw) {
          int len = 0;
          for (int c = c0; c < std::min(nc, c0 + cpw); ++c) len += valid[c];
          enc_segs_h.push_back(AttnSeg{t0, len, (int64_t)t0 * 3 * d.enc_d});
          enc_max_seg = std::max(enc_max_seg, len);
          t0 += len;
        }
      }
      total_chunks += nc;
    }
    total_T = tok;
    const int f3 = d.freq3();
    conv3_map.resize((size_t)total_chunks * f3 * tpc);
    for (int c = 0; c < total_chunks; ++c)
      for (int f = 0; f < f3; ++f)
        for (int t = 0; t < tpc; ++t) conv3_map[((size_t)c * f3 + f) * tpc + t] = (c * tpc + t) * f3 + f;
    pcm.ensure((size_t)po * 4);
    upload(d_pcm_off, pcm_off, stream);
    upload(d_n_samples, n_samples, stream);
    upload(d_mel_off, mel_off, stream);
    upload(d_n_frames, n_frames, stream);
    upload(d_chunk_utt, chunk_utt, stream);
    upload(d_chunk_frame0, chunk_frame0, stream);
    upload(convout_rowmap, convout_map, stream);
    upload(conv3_rowmap, conv3_map, stream);
    upload(enc_segs, enc_segs_h, stream);
    mel.ensure((size_t)mo * 4);
    gmax.ensure((size_t)b * 4);
    HIPCHK(hipStreamSynchronize(stream));  // host vectors above go out of scope
    B = b;  // committed only after every table is on the device
    geom_valid = true;
  }

  void upload_pcm(const float* host, const int64_t* ns, int b) {
    set_batch(ns, b);
    int64_t src = 0;
    for (int u = 0; u < b; ++u) {
      HIPCHK(hipMemcpyAsync(pcm.as<float>() + pcm_off[u], host + src, (size_t)ns[u] * 4, hipMemcpyHostToDevice, stream));
      src += ns[u];
    }
    HIPCHK(hipStreamSynchronize(stream));
  }

  // Host PCM of B utterances (one pointer each, pageable) -> HBM -> log-mel, overlapped (see the members above).  On return the
  // log-mel of the whole batch is ENQUEUED on `stream` (ev[0] / ev[1] recorded around it); nothing has been waited for.
  void upload_ptrs_and_mel(const float* const* ptrs, const int64_t* ns, int b) {
    const auto w0 = std::chrono::steady_clock::now();
    set_batch(ns, b);
    static const int mode = [] { const char* e = getenv("Q3A_UPLOAD_MODE"); return e ? atoi(e) : 1; }();
    static const int thr_env = [] { const char* e = getenv("Q3A_UPLOAD_THREADS"); return e ? atoi(e) : -1; }();
    static const int pieces_env = [] { const char* e = getenv("Q3A_UPLOAD_PIECES"); return e ? atoi(e) : 8; }();
    io = q3a_io_timings{};
    io.mode = mode;
    HIPCHK(hipEventRecord(ev[0], stream));
    const size_t total = (size_t)(pcm_off[b - 1] + ((ns[b - 1] + 3) & ~int64_t(3))) * 4;  // bytes of the device layout
    bool staged = mode != 0;
    if (staged && total > pin_cap) {  // (re)allocate the pinned mirror; a host that refuses to pin it gets the direct copies below
      if (up_stream) HIPCHK(hipStreamSynchronize(up_stream));  // (a call that threw may have left copies in flight that read the old buffer)
      if (pin_p) (void)hipHostFree(pin_p);
      pin_p = nullptr; pin_cap = 0;
      const size_t want = (total + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
      if (hipHostMalloc(&pin_p, want, hipHostMallocDefault) == hipSuccess && pin_p) pin_cap = want;
      else { (void)hipGetLastError(); pin_p = nullptr; staged = false; io.mode = 0; }
    }
    if (!staged) {  // round 4's form (A/B knob Q3A_UPLOAD_MODE=0, or no pinned memory to be had)
      for (int u = 0; u < b; ++u)
        HIPCHK(hipMemcpyAsync(pcm.as<float>() + pcm_off[u], ptrs[u], (size_t)ns[u] * 4, hipMemcpyHostToDevice, stream));
      HIPCHK(hipStreamSynchronize(stream));
      io.pieces = b;
      run_mel();
      HIPCHK(hipEventRecord(ev[1], stream));
      io.stage_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - w0).count();
      return;
    }
    if (!up_stream) {
      HIPCHK(hipStreamCreateWithFlags(&up_stream, hipStreamNonBlocking));
      HIPCHK(hipEventCreate(&up_t0));
      HIPCHK(hipEventCreate(&up_t1));
    }
    HIPCHK(hipStreamSynchronize(up_stream));  // (a call that threw may have left copies in flight that read the staging buffer)
    // pieces = runs of consecutive utterances of roughly equal bytes
    const int want_pieces = std::max(1, std::min(b, pieces_env));
    std::vector<int> first;  // first utterance of every piece (+ sentinel b)
    {
      const size_t per = (total + want_pieces - 1) / want_pieces;
      size_t acc = 0;
      first.push_back(0);
      for (int u = 0; u < b; ++u) {
        acc += (size_t)ns[u] * 4;
        if (acc >= per && u + 1 < b && (int)first.size() < want_pieces) { first.push_back(u + 1); acc = 0; }
      }
      first.push_back(b);
    }
    const int np = (int)first.size() - 1;
    while ((int)up_ev.size() < np) {
      hipEvent_t e2;
      HIPCHK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
      up_ev.push_back(e2);
    }
    const int nthr = thr_env >= 0 ? std::min(thr_env, np) : (total >= (size_t(8) << 20) ? std::min(8, np) : 0);
    io.pieces = np;
    io.threads = nthr;
    std::vector<std::atomic<int>> ready(np);
    for (auto& r : ready) r.store(0, std::memory_order_relaxed);
    std::atomic<int> next{0};
    uint8_t* const pin = (uint8_t*)pin_p;
    auto copy_piece = [&](int p) {
      for (int u = first[p]; u < first[p + 1]; ++u) memcpy(pin + (size_t)pcm_off[u] * 4, ptrs[u], (size_t)ns[u] * 4);
      ready[p].store(1, std::memory_order_release);
    };
    struct Joiner {  // an exception on the issuing thread must not unwind past running workers (they reference this frame)
      std::vector<std::thread> th;
      ~Joiner() { for (auto& t : th) if (t.joinable()) t.join(); }
    } workers;
    for (int t = 0; t < nthr; ++t)
      workers.th.emplace_back([&] { for (int p; (p = next.fetch_add(1, std::memory_order_relaxed)) < np;) copy_piece(p); });
    MelBatch mb{pcm.as<float>(), d_pcm_off.as<int64_t>(), d_n_samples.as<int64_t>(), d_mel_off.as<int64_t>(),
                d_n_frames.as<int>(), mel.as<float>(), gmax.as<unsigned>()};
    HIPCHK(hipEventRecord(up_t0, up_stream));
    for (int p = 0; p < np; ++p) {
      if (nthr == 0) copy_piece(p);
      for (unsigned spins = 0; !ready[p].load(std::memory_order_acquire);)
        if (++spins > 256) sched_yield();
      const int u0 = first[p], u1 = first[p + 1];
      const size_t o0 = (size_t)pcm_off[u0] * 4, o1 = (size_t)(pcm_off[u1 - 1] + ns[u1 - 1]) * 4;
      HIPCHK(hipMemcpyAsync((uint8_t*)pcm.p + o0, pin + o0, o1 - o0, hipMemcpyHostToDevice, up_stream));
      HIPCHK(hipEventRecord(up_ev[p], up_stream));
      HIPCHK(hipStreamWaitEvent(stream, up_ev[p], 0));
      MelBatch piece = mb;  // the tables are indexed by utterance: a piece is the same launch on offset tables
      piece.pcm_off += u0; piece.n_samples += u0; piece.mel_off += u0; piece.n_frames += u0; piece.gmax_key += u0;
      int mf = 0;
      for (int u = u0; u < u1; ++u) mf = std::max(mf, n_frames[u]);
      KCHK(launch_mel(piece, u1 - u0, mf, dft.as<float>(), filt_t.as<float>(), stream));
    }
    HIPCHK(hipEventRecord(up_t1, up_stream));
    io.stage_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - w0).count();
    mel_done();
    HIPCHK(hipEventRecord(ev[1], stream));
  }

  // =====================================================================================
  void run_mel() {
    MelBatch mb{pcm.as<float>(), d_pcm_off.as<int64_t>(), d_n_samples.as<int64_t>(), d_mel_off.as<int64_t>(),
                d_n_frames.as<int>(), mel.as<float>(), gmax.as<unsigned>()};
    KCHK(launch_mel(mb, B, max_frames, dft.as<float>(), filt_t.as<float>(), stream));
    mel_done();
  }
  void mel_done() {
    have_mel = true;
    size_t bytes = 0;
    for (int u = 0; u < B; ++u) bytes += (size_t)n_frames[u] * d.n_mels * 4;
    tap("mel", mel.p, bytes);
  }

  // =====================================================================================
  void run_encoder() {
    if (!have_mel) fail("q3a_encode: no mel (call q3a_mel first)");
    const int C = d.conv_ch, D = d.enc_d, Fn = d.enc_ffn;
    const int H1 = Dims::conv_len(d.n_mels), W1 = Dims::conv_len(d.chunk_frames());
    const int H2 = Dims::conv_len(H1), W2 = Dims::conv_len(W1);
    const int H3 = Dims::conv_len(H2), W3 = Dims::conv_len(W2);
    const bool sp = precise();
    const size_t nch = (size_t)total_chunks;
    const size_t ae = act_elem();
    conv1.ensure(nch * H1 * W1 * C * ae);
    conv2.ensure(nch * H2 * W2 * C * ae);
    conv3.ensure(nch * H3 * W3 * C * ae);
    const size_t Tt = (size_t)total_T;
    // the fp32 VALU attention (precise mode / opts.valu_attention) writes an fp32 context that is rounded afterwards
    const bool valu_attn = sp || opts.valu_attention;
    // the MFMA attention consumes q/k/v as bf16: the qkv GEMM then stores exactly those values (half the bytes)
    const bool qkv16 = !valu_attn;
    enc_x.ensure(Tt * D * 4); enc_ln.ensure(Tt * D * ae); enc_qkv.ensure(Tt * 3 * D * (qkv16 ? 2 : 4));
    enc_ctx.ensure(Tt * D * (valu_attn ? 4 : ae));
    if (valu_attn && !sp) enc_ctx16.ensure(Tt * D * 2);
    enc_ffn.ensure(Tt * Fn * ae); audio_embeds.ensure(Tt * d.enc_out * 4);

    ChunkTable ct{d_chunk_utt.as<int>(), d_chunk_frame0.as<int>()};
    KCHK(launch_conv1(mel.as<float>(), d_mel_off.as<int64_t>(), d_n_frames.as<int>(), ct, total_chunks, d.n_mels,
                      d.chunk_frames(), wf(L.conv1_w), wf(L.conv1_b), C, conv1.as<float>(), stream, act16(conv1)));
    tap_act("conv1", conv1.p, nch * H1 * W1 * C);
    {
      GemmEpilogue ep;
      act_out(ep, conv2); ep.ldo = C; ep.bias = wf(L.conv2_b); ep.act = 1;
      act_conv(conv1, total_chunks, H1, W1, C, wh(L.conv2_w), C, ep);
      tap_act("conv2", conv2.p, nch * H2 * W2 * C);
    }
    {
      GemmEpilogue ep;  // rows land as [chunk][t][f][c]  (audio_encoder.rs:132-133 permute fused away)
      act_out(ep, conv3); ep.ldo = C; ep.bias = wf(L.conv3_b); ep.act = 1; ep.rowmap = conv3_rowmap.as<int>();
      act_conv(conv2, total_chunks, H2, W2, C, wh(L.conv3_w), C, ep);
      tap_act("conv3", conv3.p, nch * H3 * W3 * C);
    }
    {
      GemmEpilogue ep;  // conv_out + positional embedding (restarting per chunk) + valid-token gather
      ep.out = enc_x.as<float>(); ep.ldo = D;
      ep.bias = (arena_flags & kFlagConvOutBias) ? wf(L.conv_out_b) : nullptr;
      ep.rowmap = convout_rowmap.as<int>();
      ep.addend = pos_emb.as<float>(); ep.addend_period = W3;
      act_gemm(conv3, H3 * C, wh(L.conv_out_w), total_chunks * W3, D, H3 * C, ep, false);
      tap("enc_in", enc_x.p, Tt * D * 4);
    }
    AttnArgs at{};
    at.q = enc_qkv.as<float>(); at.q_rs = 3 * D;
    at.k = enc_qkv.as<float>() + D; at.v = enc_qkv.as<float>() + 2 * D; at.kv_hs = 64; at.kv_rs = 3 * D;
    if (qkv16) {
      at.q16 = enc_qkv.as<uint16_t>();
      at.k = enc_qkv.as<uint16_t>() + D; at.v = enc_qkv.as<uint16_t>() + 2 * D;
    }
    at.o = enc_ctx.as<float>(); at.o_rs = D;
    at.o16 = valu_attn ? nullptr : enc_ctx.as<uint16_t>();
    at.segs = enc_segs.as<AttnSeg>(); at.n_segs = (int)enc_segs_h.size(); at.max_len = enc_max_seg;
    at.n_kv_heads = d.enc_heads; at.scale_div = 8.0f;  // sqrt(64), layers.rs:161
    const DevBuf& ctx_in = (valu_attn && !sp) ? enc_ctx16 : enc_ctx;  // what the out projection reads
    for (int li = 0; li < d.enc_layers; ++li) {
      const EncLayerOff& e = L.enc[li];
      KCHK(launch_layernorm(enc_x.as<float>(), wf(e.ln1_w), wf(e.ln1_b), enc_ln.as<float>(), total_T, D, 1e-5f, stream, act16(enc_ln)));
      {
        GemmEpilogue ep; ep.ldo = 3 * D; ep.bias = wf(e.qkv_b);
        if (qkv16) ep.out16 = enc_qkv.as<uint16_t>(); else ep.out = enc_qkv.as<float>();
        act_gemm(enc_ln, D, wh(e.qkv_w), total_T, 3 * D, D, ep, false);
      }
      if (valu_attn) {
        KCHK(launch_attn_enc(at, stream));
        if (!sp) KCHK(launch_to_bf16(enc_ctx.as<float>(), enc_ctx16.as<uint16_t>(), Tt * D, stream));
      } else {
        KCHK(launch_fattn_enc(at, stream));
      }
      {
        GemmEpilogue ep; ep.out = enc_x.as<float>(); ep.ldo = D; ep.bias = wf(e.out_b); ep.resid = enc_x.as<float>();
        act_gemm(ctx_in, D, wh(e.out_w), total_T, D, D, ep, false);
      }
      KCHK(launch_layernorm(enc_x.as<float>(), wf(e.ln2_w), wf(e.ln2_b), enc_ln.as<float>(), total_T, D, 1e-5f, stream, act16(enc_ln)));
      {
        GemmEpilogue ep; act_out(ep, enc_ffn); ep.ldo = Fn; ep.bias = wf(e.fc1_b); ep.act = 1;
        act_gemm(enc_ln, D, wh(e.fc1_w), total_T, Fn, D, ep, false);
      }
      {
        GemmEpilogue ep; ep.out = enc_x.as<float>(); ep.ldo = D; ep.bias = wf(e.fc2_b); ep.resid = enc_x.as<float>();
        act_gemm(enc_ffn, Fn, wh(e.fc2_w), total_T, D, Fn, ep, false);
      }
      if (li == 0) tap("enc_layer0", enc_x.p, Tt * D * 4);
      static const bool enc_layer_taps = [] { const char* e = getenv("Q3A_DEBUG_LAYER_TAPS"); return e && atoi(e) != 0; }();
      if (enc_layer_taps && opts.debug_taps == 1) {  // (tools/bisect_layers.py)
        char name[32];
        snprintf(name, sizeof(name), "E%02d_x", li);
        tap(name, enc_x.p, Tt * D * 4);
      }
    }
    tap("enc_last", enc_x.p, Tt * D * 4);
    KCHK(launch_layernorm(enc_x.as<float>(), wf(L.ln_post_w), wf(L.ln_post_b), enc_ln.as<float>(), total_T, D, 1e-5f, stream, act16(enc_ln)));
    {
      GemmEpilogue ep; act_out(ep, enc_ffn); ep.ldo = D; ep.bias = wf(L.proj1_b); ep.act = 1;  // enc_ffn doubles as the proj1 output
      act_gemm(enc_ln, D, wh(L.proj1_w), total_T, D, D, ep, false);
    }
    {
      GemmEpilogue ep; ep.out = audio_embeds.as<float>(); ep.ldo = d.enc_out; ep.bias = wf(L.proj2_b);
      act_gemm(enc_ffn, D, wh(L.proj2_w), total_T, d.enc_out, D, ep, false);
    }
    tap("audio_embeds", audio_embeds.p, Tt * d.enc_out * 4);
    HIPCHK(hipGetLastError());
    have_enc = true;
  }

  // =====================================================================================
  // prompts: ids concatenated, lens[B]  (inference.rs:104-137)
  // before_encoder: the whole-path entry points build the prompts BEFORE they enqueue anything of the batch -- the tables only
  // depend on T[] (host integers of set_batch), and the stream synchronisation below then meets an idle stream instead of
  // draining the encoder between its last kernel and the prefill's first
  void setup_prompts(const int32_t* ids_h, const int32_t* lens, int b, int max_new_req, bool before_encoder = false) {
    if (!have_enc && !before_encoder) fail("q3a_prefill: no encoder output (call q3a_encode first)");
    if (b != B) fail("q3a_prefill: batch size differs from the encoded batch");
    P.assign(lens, lens + b);
    seq_off.resize(b);
    std::vector<int> rowseq, rowpos, amap((size_t)total_T, -1), last(b);
    int off = 0, maxP = 0;
    for (int s = 0; s < b; ++s) {
      seq_off[s] = off;
      int na = 0;
      for (int i = 0; i < P[s]; ++i) {
        rowseq.push_back(s);
        rowpos.push_back(i);
        int id = ids_h[off + i];
        if (id < 0 || id >= d.vocab) fail("q3a_prefill: token id out of range");
        if (id == kAudioPad) {
          if (na >= T[s]) fail("q3a_prefill: more <|audio_pad|> tokens than encoder tokens");
          amap[(size_t)tok_off[s] + na] = off + i;
          ++na;
        }
      }
      if (na != T[s]) fail("q3a_prefill: number of <|audio_pad|> tokens does not match the encoder output length");
      if (P[s] <= 0) fail("q3a_prefill: empty prompt");
      last[s] = off + P[s] - 1;
      off += P[s];
      maxP = std::max(maxP, P[s]);
    }
    total_P = off;
    max_new = std::min(std::max(max_new_req, 1), opts.max_new_tokens);
    max_ctx = ((maxP + max_new + 1 + 127) / 128) * 128;  // whole 128-key tiles (the batched decode attention reads cache rows tile by tile)
    pos_hi_ = maxP;
    min_P_ = *std::min_element(P.begin(), P.end());
    {  // A/B knob: extra keys per (sequence, kv head) cache row block, so that the streams of the batched decode attention
       // do not all start a power of two apart (512 keys x 256 B = 128 KiB)
      static const int pad = [] { const char* e = getenv("Q3A_CTX_PAD"); return e ? atoi(e) : 0; }();
      if (pad > 0) max_ctx += (pad + 127) / 128 * 128;
    }
    ensure_rope(max_ctx + 1);  // argmax_finalize reads the row of the position AFTER the last one the cache can hold
    std::vector<int> ids_v(ids_h, ids_h + total_P);
    std::vector<AttnSeg> segs(b);
    for (int s = 0; s < b; ++s)
      segs[s] = AttnSeg{seq_off[s], P[s], (int64_t)s * d.n_kv * max_ctx * 128};
    upload(ids, ids_v, stream);
    upload(row_seq, rowseq, stream);
    upload(row_pos, rowpos, stream);
    upload(audio_rowmap, amap, stream);
    upload(last_rows, last, stream);
    upload(dec_segs, segs, stream);
    upload(d_pos, P, stream);
    const size_t Pt = (size_t)total_P, H = d.hidden;
    const bool valu_attn = kv_f32() || opts.valu_attention;
    dec_x.ensure(Pt * H * 4); dec_ln.ensure(Pt * H * act_elem()); dec_qkv.ensure(Pt * d.qkv_dim() * 4);
    dec_ctx.ensure(Pt * d.q_dim() * (valu_attn ? 4 : act_elem())); dec_act.ensure(Pt * d.inter * act_elem());
    if (!valu_attn) dec_q16.ensure(Pt * d.q_dim() * 2);  // q after QK-norm + RoPE as bf16 for the MFMA prefill attention
    if (valu_attn && !precise()) dec_ctx16.ensure(Pt * d.q_dim() * 2);
    kv_layer_elems = (size_t)b * d.n_kv * max_ctx * 128;
    kcache.ensure(kv_layer_elems * d.dec_layers * kv_elem());
    vcache.ensure(kv_layer_elems * d.dec_layers * kv_elem());
    rope_cur.ensure((size_t)b * 128 * 4);
    {
      Knobs& kn = knobs();
      const int gs = kn.decode_group_size.load();
      gsize = (gs >= 1 && gs <= 32) ? gs : 32;
      k_parallel_groups = kn.decode_parallel_groups.load(); k_skinny_q = kn.skinny_q.load();
      k_dattn_batched_min_wgs = kn.dattn_batched_min_wgs.load();
      k_skinny_glu_hp3 = kn.skinny_glu_hp3.load();
    }
    const size_t ng = (size_t)n_groups(b);  // groups of <= gsize sequences of the batched decode step
    nn_x.ensure(ng * 32 * H * 2); nn_ss.ensure((size_t)ng * (H / 8) * 32 * 4);  // room for the finer (8-column) partial rows whichever shape the knob selects later
    x_dec.ensure((size_t)b * H * 4); next_tok.ensure((size_t)b * 4); forced_tok.ensure((size_t)b * 4);
    out_ids.ensure((size_t)b * max_new * 4); step_count.ensure((size_t)b * 4); done.ensure((size_t)b);
    // s_ctx / s_act also hold the bf16 fragment-order copies of the skinny GEMM path: always 32 sequences there
    s_ln.ensure((size_t)b * H * 4); s_qkv.ensure((size_t)b * d.qkv_dim() * 4); s_ctx.ensure(ng * 32 * d.q_dim() * 4);
    s_act.ensure(ng * 32 * d.inter * 4); logits.ensure((size_t)b * d.vocab * 4);
    part_stride = std::max(128, (d.vocab + 3) / 4);  // >= blocks of the lm_head GEMV at 1 row per wave
    part_val.ensure((size_t)b * part_stride * 4); part_idx.ensure((size_t)b * part_stride * 4);
    attn_nsplit = (max_ctx + dattn_keys_per_split(kv_f32()) - 1) / dattn_keys_per_split(kv_f32());
    attn_pm.ensure((size_t)b * d.n_q * attn_nsplit * 4); attn_pl.ensure((size_t)b * d.n_q * attn_nsplit * 4);
    attn_po.ensure((size_t)b * d.n_q * attn_nsplit * 128 * 4);
    HIPCHK(hipMemsetAsync(step_count.p, 0, (size_t)b * 4, stream));
    HIPCHK(hipMemsetAsync(done.p, 0, (size_t)b, stream));
    HIPCHK(hipMemsetAsync(n_done.p, 0, 64, stream));
    HIPCHK(hipMemsetAsync(out_ids.p, 0, (size_t)b * max_new * 4, stream));
    // drain the stream -- and the group streams: a run that threw mid-loop may have left run-ahead steps enqueued whose
    // argmax_finalize would still write the progress words -- BEFORE the words are reset
    HIPCHK(hipStreamSynchronize(stream));
    for (auto cs : chain_streams) HIPCHK(hipStreamSynchronize(cs));
    __atomic_store_n(&host_prog[0], 0, __ATOMIC_RELAXED);
    __atomic_store_n(&host_prog[1], 0, __ATOMIC_RELAXED);
    drain_retired_graphs();
    {  // a step-visible buffer was reallocated: cached graphs may replay launches that point at the freed allocation
      bool any = false;
      for (auto* sb : step_bufs()) { any = any || sb->grew; sb->grew = false; }
      if (any) drop_graphs();
    }
  }

  // Every DevBuf a launch of the captured decode step reads or writes (decode_layer, run_head, enqueue_decode_step and the
  // accessors they call).  ONE list serves both guards against a stale graph: the reallocation sweep of setup_prompts and the
  // address hash inside make_graph_sig.  tests/test_host.py::test_step_buffers_cover_the_captured_step scans those functions'
  // source for DevBuf members and fails when one is missing here.
  std::vector<DevBuf*> step_bufs() {
    return {&kcache, &vcache, &x_dec, &d_pos, &next_tok, &out_ids, &step_count, &done, &s_ln, &s_qkv, &s_ctx, &s_act, &logits,
            &part_val, &part_idx, &attn_pm, &attn_pl, &attn_po, &nn_x, &nn_ss, &rope_cur, &rope_cos, &rope_sin, &n_done, &forced_tok};
  }
  std::vector<const DevBuf*> step_bufs() const {
    auto v = const_cast<q3a_engine*>(this)->step_bufs();
    return std::vector<const DevBuf*>(v.begin(), v.end());
  }

  void* kc_layer(int l) { return (uint8_t*)kcache.p + (size_t)l * kv_layer_elems * kv_elem(); }
  void* vc_layer(int l) { return (uint8_t*)vcache.p + (size_t)l * kv_layer_elems * kv_elem(); }

  // final norm + lm_head on x_dec -> logits (+ block argmax partials), then argmax/finalize
  void run_head(int advance) {
    const int S = B, H = d.hidden, V = d.vocab;
    tap("head_in", x_dec.p, (size_t)S * H * 4);  // (debug taps) last-layer residual rows the final norm + lm_head read
    const double wbytes = 2.0 * V * H;
    int n_part = 0;
    if (S <= kGemvMaxSeq) {
      GemvArgs g{};
      g.fast_math = precise() ? 0 : 1;
      g.x = x_dec.as<float>(); g.ldx = H; g.rms_w = wf(L.final_norm); g.eps = d.rms_eps;
      g.W = wh(L.lm_head); g.N = V; g.K = H; g.mode = 3; g.out = logits.as<float>(); g.ldo = V;
      g.part_val = part_val.as<float>(); g.part_idx = part_idx.as<int>(); g.part_stride = part_stride;
      n_part = gemv_blocks(g);
      timed(Q3A_KC_GEMV_LM_HEAD, wbytes, [&] { KCHK(launch_gemv(g, S, stream)); });
    } else if (!precise()) {
      // default mode: bf16 normed rows, then the LDS-DMA GEMM at M = S (32-row tiles) -- the vocabulary matrix is
      // streamed once through double-buffered LDS; the skinny kernel re-reads x per 16 vocabulary rows (3x slower here)
      timed(Q3A_KC_NORM, 0, [&] { KCHK(launch_rmsnorm(x_dec.as<float>(), wf(L.final_norm), s_ln.as<float>(), S, H, d.rms_eps, stream, s_ln.as<uint16_t>())); });
      GemmEpilogue ep; ep.out = logits.as<float>(); ep.ldo = V;
      if (S <= 32 && V % 4 == 0 && H % 64 == 0 && part_stride >= (V + 63) / 64) {
        // the argmax partials are the GEMM's epilogue (one per 64-column tile and row), and inside q3a_transcribe_batch /
        // q3a_run_resident nobody reads the logits: they are not stored (19 MB per step at 32 sequences)
        ep.part_val = part_val.as<float>(); ep.part_idx = part_idx.as<int>(); ep.part_stride = part_stride;
        if (!head_logits_) ep.out = nullptr;
        n_part = (V + 63) / 64;
        timed(Q3A_KC_GEMM, wbytes, [&] { KCHK(launch_gemm16(s_ln.as<uint16_t>(), H, wh(L.lm_head), S, V, H, ep, false, stream)); });
      } else {
        timed(Q3A_KC_GEMM, wbytes, [&] { KCHK(launch_gemm16(s_ln.as<uint16_t>(), H, wh(L.lm_head), S, V, H, ep, false, stream)); });
        n_part = 128;
        timed(Q3A_KC_ARGMAX, 0, [&] { KCHK(launch_argmax_partials(logits.as<float>(), V, S, part_val.as<float>(), part_idx.as<int>(), part_stride, n_part, stream)); });
      }
    } else {
      timed(Q3A_KC_NORM, 0, [&] { KCHK(launch_rmsnorm(x_dec.as<float>(), wf(L.final_norm), s_ln.as<float>(), S, H, d.rms_eps, stream)); });
      timed(Q3A_KC_GEMM, wbytes, [&] { batched_proj(s_ln.as<float>(), H, wh(L.lm_head), V, H, nullptr, 0, logits.as<float>(), V, nullptr); });
      n_part = 128;
      timed(Q3A_KC_ARGMAX, 0, [&] { KCHK(launch_argmax_partials(logits.as<float>(), V, S, part_val.as<float>(), part_idx.as<int>(), part_stride, n_part, stream)); });
    }
    FinalizeArgs f{};
    f.part_val = part_val.as<float>(); f.part_idx = part_idx.as<int>(); f.part_stride = part_stride; f.n_part = n_part;
    f.V = V; f.next_tok = next_tok.as<int>(); f.out_ids = out_ids.as<int>();
    f.out_stride = max_new; f.step_count = step_count.as<int>(); f.pos = d_pos.as<int>(); f.advance = advance;
    f.done = done.as<uint8_t>(); f.n_done = n_done.as<int>(); f.n_seq = S; f.host_progress = host_prog_dev; f.embed = wh(L.embed); f.H = H; f.x_next = x_dec.as<float>(); f.eos0 = kEos0; f.eos1 = kEos1;
    f.cos_t = rope_cos.as<float>(); f.sin_t = rope_sin.as<float>(); f.rope_cur = rope_cur.as<float>();
    f.nn = first_layer_norm_out();
    timed(Q3A_KC_ARGMAX, 0, [&] { KCHK(launch_argmax_finalize(f, S, stream)); });
  }

  void run_prefill() {
    const int H = d.hidden, I = d.inter, QD = d.q_dim(), QKV = d.qkv_dim();
    const bool sp = precise();
    KCHK(launch_embed(ids.as<int>(), total_P, wh(L.embed), H, kAudioPad, dec_x.as<float>(), stream));
    // audio injection (inference.rs:114-124): one row scatter instead of T full-tensor slice_scatter copies
    scatter_audio_rows();
    tap("dec_embed", dec_x.p, (size_t)total_P * H * 4);
    AttnArgs at{};
    at.q = dec_qkv.as<float>(); at.q_rs = QKV; at.kv_hs = (int64_t)max_ctx * 128; at.kv_rs = 128;
    at.o = dec_ctx.as<float>(); at.o_rs = QD; at.segs = dec_segs.as<AttnSeg>(); at.n_segs = B;
    at.max_len = *std::max_element(P.begin(), P.end()); at.n_kv_heads = d.n_kv;
    at.scale_div = sqrtf((float)d.head_dim);  // layers.rs:327-328
    const bool valu_attn = kv_f32() || opts.valu_attention;
    at.o16 = valu_attn ? nullptr : dec_ctx.as<uint16_t>();
    if (!valu_attn) { at.q16 = dec_q16.as<uint16_t>(); at.q_rs = QD; }  // the rope kernel leaves q as bf16 [rows][QD]
    const DevBuf& ctx_in = (valu_attn && !sp) ? dec_ctx16 : dec_ctx;  // what the o projection reads
    const bool qkv_bias = arena_flags & kFlagDecQkvBias, o_bias = arena_flags & kFlagDecOBias, mlp_bias = arena_flags & kFlagDecMlpBias;
    const bool fuse_rope = knobs().fuse_qkrope.load() != 0 && !valu_attn && !precise() && d.head_dim == 128 && gemm256_eligible(total_P, QKV, H) && H % 64 == 0;
    for (int li = 0; li < d.dec_layers; ++li) {
      const DecLayerOff& l = L.dec[li];
      KCHK(launch_rmsnorm(dec_x.as<float>(), wf(l.in_ln), dec_ln.as<float>(), total_P, H, d.rms_eps, stream, act16(dec_ln)));
      RopeKvArgs rk{};
      rk.qkv = dec_qkv.as<float>(); rk.row_seq = row_seq.as<int>(); rk.row_pos = row_pos.as<int>();
      rk.q_norm = wf(l.q_norm); rk.k_norm = wf(l.k_norm); rk.eps = d.rms_eps;
      rk.cos_t = rope_cos.as<float>(); rk.sin_t = rope_sin.as<float>();
      rk.kcache = kc_layer(li); rk.vcache = vc_layer(li); rk.n_q = d.n_q; rk.n_kv = d.n_kv; rk.max_ctx = max_ctx;
      rk.q16 = valu_attn ? nullptr : dec_q16.as<uint16_t>();
      if (fuse_rope) {
        // batch-sized prefill: QK-norm, RoPE and the cache append are the epilogue of the qkv GEMM (k_gemm256.hip)
        KCHK(launch_gemm256_qkrope(dec_ln.as<uint16_t>(), H, wh(l.qkv_w), total_P, H, qkv_bias ? wf(l.qkv_b) : nullptr, rk, stream));
      } else {
        GemmEpilogue ep; ep.out = dec_qkv.as<float>(); ep.ldo = QKV; ep.bias = qkv_bias ? wf(l.qkv_b) : nullptr;
        act_gemm(dec_ln, H, wh(l.qkv_w), total_P, QKV, H, ep, false);
        KCHK(launch_qknorm_rope_kv(rk, total_P, kv_f32(), stream));
      }
      at.k = kc_layer(li); at.v = vc_layer(li);
      // (debug_taps + Q3A_DEBUG_LAYER_TAPS=1: raw copies of every prefill layer's intermediate buffers, for bisecting a
      // run-to-run difference to one launch -- tools/bisect_layers.py)
      static const bool layer_taps = [] { const char* e = getenv("Q3A_DEBUG_LAYER_TAPS"); return e && atoi(e) != 0; }();
      auto ltap = [&](const char* what, const void* ptr, size_t bytes) {
        if (!layer_taps || !opts.debug_taps) return;
        char name[32];
        snprintf(name, sizeof(name), "L%02d_%s", li, what);
        tap(name, ptr, bytes);
      };
      const size_t act_b = sp ? 4 : 2;  // bytes per activation element (bf16 in the default mode)
      if (fuse_rope) {
        ltap("qkvs", dec_qkv.p, std::min((size_t)1024, (size_t)total_P) * QKV * 4);  // fp32 scratch of the trailing rows (small GEMM -> separate rope kernel)
      }
      ltap("k", kc_layer(li), (size_t)B * d.n_kv * max_ctx * 128 * kv_elem());
      ltap("v", vc_layer(li), (size_t)B * d.n_kv * max_ctx * 128 * kv_elem());
      if (valu_attn) {
        KCHK(launch_attn_prefill(at, d.n_q / d.n_kv, kv_f32(), stream));
        if (!sp) KCHK(launch_to_bf16(dec_ctx.as<float>(), dec_ctx16.as<uint16_t>(), (size_t)total_P * QD, stream));
      } else {
        KCHK(launch_fattn_prefill(at, d.n_q / d.n_kv, stream));
      }
      ltap("attn", ctx_in.p, (size_t)total_P * QD * act_b);
      {
        GemmEpilogue ep; ep.out = dec_x.as<float>(); ep.ldo = H; ep.resid = dec_x.as<float>(); ep.bias = o_bias ? wf(l.o_b) : nullptr;
        act_gemm(ctx_in, QD, wh(l.o_w), total_P, H, QD, ep, false);
      }
      ltap("o", dec_x.p, (size_t)total_P * H * 4);
      KCHK(launch_rmsnorm(dec_x.as<float>(), wf(l.post_ln), dec_ln.as<float>(), total_P, H, d.rms_eps, stream, act16(dec_ln)));
      ltap("ln2", dec_ln.p, (size_t)total_P * H * act_b);
      {
        GemmEpilogue ep; act_out(ep, dec_act); ep.ldo = I; ep.bias = mlp_bias ? wf(l.gu_b) : nullptr;
        act_gemm(dec_ln, H, wh(l.gu_w), total_P, 2 * I, H, ep, true);
      }
      ltap("act", dec_act.p, (size_t)total_P * I * act_b);
      {
        GemmEpilogue ep; ep.out = dec_x.as<float>(); ep.ldo = H; ep.resid = dec_x.as<float>(); ep.bias = mlp_bias ? wf(l.down_b) : nullptr;
        act_gemm(dec_act, I, wh(l.down_w), total_P, H, I, ep, false);
      }
      ltap("x", dec_x.p, (size_t)total_P * H * 4);
      if (li == 0) tap("dec_layer0", dec_x.p, (size_t)total_P * H * 4);
    }
    // only the last row of every sequence feeds the lm_head (the reference computes all rows and keeps
    // the last, text_decoder.rs:111-112 + inference.rs:156)
    KCHK(launch_gather_rows(dec_x.as<float>(), last_rows.as<int>(), B, H, x_dec.as<float>(), stream));
    tap("dec_last_hidden", x_dec.p, (size_t)B * H * 4);
    run_head(0);
    tap("logits", logits.p, (size_t)B * d.vocab * 4);
    HIPCHK(hipGetLastError());
    have_prefill = true;
  }

  void scatter_audio_rows();

  // Skinny decode path in the default mode: every kernel that writes a row of the residual stream (token embedding,
  // o / down projection) also leaves it pre-normalised for the GEMM that reads it next -- bf16(x * w_norm) in MFMA
  // fragment order in nn_x plus partial sums of x^2 in nn_ss (kernels.h NextNormOut / SkinnyArgs::xw16f).
  bool prenorm_path() const { return B > kGemvMaxSeq && !precise(); }
  // o / down projections of the batched decode step as 8-row x 16-sequence workgroups (k_skinny.hip QS; Q3A_SKINNY_Q=0: off)
  bool skinny_q() const {
    auto whole = [](int K) { const int st = K / 32, per = (st + 7) / 8, unr = per <= 2 ? 2 : per <= 4 ? 4 : per <= 8 ? 8 : 12; return K % 256 == 0 && per % unr == 0; };
    return k_skinny_q != 0 && !precise() && d.hidden % 64 == 0 && whole(d.q_dim()) && whole(d.inter);
  }
  int nn_parts() const { return d.hidden / (skinny_q() ? 8 : 16); }  // one partial per 16- (8-) column block of a hidden-wide GEMM output
  // Batched decode (more than kGemvMaxSeq sequences) runs in groups of <= 32 sequences: the skinny MFMA GEMM holds 32
  // sequences per weight sweep, and the second group's sweep of a 4-13 MB matrix is served by the L2 / Infinity Cache.
  int n_groups(int b) const { return b <= kGemvMaxSeq ? 1 : (b + gsize - 1) / gsize; }
  size_t nn_x_stride() const { return (size_t)32 * d.hidden; }         // elements per group
  size_t nn_ss_stride() const { return (size_t)nn_parts() * 32; }
  uint16_t* nn_x_g(int g) const { return nn_x.as<uint16_t>() + (size_t)g * nn_x_stride(); }
  float* nn_ss_g(int g) const { return nn_ss.as<float>() + (size_t)g * nn_ss_stride(); }
  float* s_ctx_g(int g) const { return s_ctx.as<float>() + (size_t)g * 32 * d.q_dim(); }   // fp32 [S][QD] or bf16 fragment order
  float* s_act_g(int g) const { return s_act.as<float>() + (size_t)g * 32 * d.inter; }
  NextNormOut first_layer_norm_out() const {
    NextNormOut nn{};
    if (prenorm_path()) {
      nn.next_w = wf(L.dec[0].in_ln); nn.next_xw16f = nn_x.as<uint16_t>(); nn.next_ss = nn_ss.as<float>(); nn.nparts = nn_parts();
      nn.group_stride_x = (long)nn_x_stride(); nn.group_stride_ss = (long)nn_ss_stride(); nn.group_size = gsize;
    }
    return nn;
  }

  // decode-step projection for more than 4 sequences: skinny MFMA GEMM up to 32, generic tiles above
  void batched_proj(const float* x, int ldx, const uint16_t* W, int N, int K, const float* bias, int mode, float* out,
                    int ldo, const float* resid) {
    const int S = B;  // (precise-mode lm_head only)
    if (S <= 32) {
      SkinnyArgs sk{};
      sk.fast_math = precise() ? 0 : 1;
      sk.x = x; sk.ldx = ldx; sk.S = S; sk.W = W; sk.N = N; sk.K = K; sk.bias = bias; sk.mode = mode; sk.out = out; sk.ldo = ldo; sk.resid = resid;
      KCHK(launch_skinny(sk, precise(), stream));
    } else {
      GemmEpilogue ep; ep.out = out; ep.ldo = ldo; ep.bias = bias; ep.resid = (mode == 1) ? resid : nullptr;
      KCHK(launch_gemm(x, ldx, W, S, N, K, ep, mode == 2, precise(), stream));
    }
  }

  // one decoder layer of the greedy-loop iteration (inference.rs:172-197) for sequences [s0, s0 + S) -- the whole batch on
  // the GEMV path, one group of <= 32 on the skinny MFMA path
  void decode_layer(int li, int grp, int s0, int S, hipStream_t ks) {
    const int H = d.hidden, I = d.inter, QD = d.q_dim(), QKV = d.qkv_dim();
    const bool qkv_bias = arena_flags & kFlagDecQkvBias, o_bias = arena_flags & kFlagDecOBias, mlp_bias = arena_flags & kFlagDecMlpBias;
    const bool gemv = B <= kGemvMaxSeq;
    const DecLayerOff& l = L.dec[li];
    float* const x = x_dec.as<float>() + (size_t)s0 * H;
    float* const qkv = s_qkv.as<float>() + (size_t)s0 * QKV;
    const size_t kv_seq = (size_t)d.n_kv * max_ctx * 128 * kv_elem();  // bytes per sequence and layer
    DecodeAttnArgs da{};
    da.qkv = qkv; da.pos = d_pos.as<int>() + s0; da.eps = d.rms_eps;
    da.rope_cur = rope_cur.as<float>() + (size_t)s0 * 128;
    da.pm = attn_pm.as<float>() + (size_t)s0 * d.n_q * attn_nsplit; da.pl = attn_pl.as<float>() + (size_t)s0 * d.n_q * attn_nsplit;
    da.po = attn_po.as<float>() + (size_t)s0 * d.n_q * attn_nsplit * 128; da.nsplit = live_nsplit_;  // (buffers sized for attn_nsplit)
    da.n_q = d.n_q; da.n_kv = d.n_kv; da.max_ctx = max_ctx; da.scale_div = sqrtf((float)d.head_dim);
    da.q_norm = wf(l.q_norm); da.k_norm = wf(l.k_norm);
    da.kcache = (uint8_t*)kc_layer(li) + (size_t)s0 * kv_seq; da.vcache = (uint8_t*)vc_layer(li) + (size_t)s0 * kv_seq;
    if (gemv) {
      GemvArgs g{};
      g.fast_math = precise() ? 0 : 1;
      g.x = x; g.ldx = H; g.rms_w = wf(l.in_ln); g.eps = d.rms_eps; g.W = wh(l.qkv_w); g.N = QKV; g.K = H;
      g.bias = qkv_bias ? wf(l.qkv_b) : nullptr; g.mode = 0; g.out = qkv; g.ldo = QKV;
      timed(Q3A_KC_GEMV, 2.0 * QKV * H, [&] { KCHK(launch_gemv(g, S, ks)); });
      timed(Q3A_KC_DECODE_ATTN, 0, [&] { KCHK(launch_decode_attn(da, S, kv_f32(), ks)); });
      GemvArgs o{};
      o.fast_math = precise() ? 0 : 1;
      if (std::min(S, 4) * d.n_q * live_nsplit_ <= GEMV_ATTN_MAX_TABLE) {  // merge the key splits inside the o_proj GEMV
        o.attn_pm = da.pm; o.attn_pl = da.pl; o.attn_po = da.po;
        o.attn_nsplit = live_nsplit_; o.attn_heads = d.n_q; o.attn_fast_exp = precise() ? 0 : 1;
      } else {  // very long contexts: separate merge launch
        timed(Q3A_KC_DECODE_ATTN, 0, [&] { KCHK(launch_attn_combine(da.pm, da.pl, da.po, live_nsplit_, S, d.n_q, s_ctx_g(grp), ks)); });
        o.x = s_ctx_g(grp);
      }
      o.ldx = QD; o.W = wh(l.o_w); o.N = H; o.K = QD; o.bias = o_bias ? wf(l.o_b) : nullptr;
      o.mode = 1; o.out = x; o.ldo = H; o.resid = x;
      timed(Q3A_KC_GEMV_O, 2.0 * H * QD, [&] { KCHK(launch_gemv(o, S, ks)); });
      GemvArgs u{};
      u.fast_math = precise() ? 0 : 1;
      u.x = x; u.ldx = H; u.rms_w = wf(l.post_ln); u.eps = d.rms_eps; u.W = wh(l.gu_w); u.N = 2 * I; u.K = H;
      u.bias = mlp_bias ? wf(l.gu_b) : nullptr; u.mode = 2; u.out = s_act_g(grp); u.ldo = I;
      timed(Q3A_KC_GEMV, 4.0 * I * H, [&] { KCHK(launch_gemv(u, S, ks)); });
      GemvArgs dn{};
      dn.fast_math = precise() ? 0 : 1;
      dn.x = s_act_g(grp); dn.ldx = I; dn.W = wh(l.down_w); dn.N = H; dn.K = I; dn.bias = mlp_bias ? wf(l.down_b) : nullptr;
      dn.mode = 1; dn.out = x; dn.ldo = H; dn.resid = x;
      timed(Q3A_KC_GEMV_DOWN, 2.0 * H * I, [&] { KCHK(launch_gemv(dn, S, ks)); });
      return;
    }
    // skinny MFMA GEMMs: the norms are fused (no norm launches), and in the default mode the two K-heavy projections read
    // bf16 activations written by their producers (attention, SwiGLU epilogue) in fragment order
    const bool b16 = !precise(), pre = prenorm_path();
    SkinnyArgs q{};
    q.fast_math = precise() ? 0 : 1;
    q.x = x; q.ldx = H; q.S = S; q.eps = d.rms_eps; q.W = wh(l.qkv_w); q.N = QKV; q.K = H;
    if (pre) { q.xw16f = nn_x_g(grp); q.ss_parts = nn_ss_g(grp); q.ss_nparts = nn_parts(); }
    else q.rms_w = wf(l.in_ln);
    q.bias = qkv_bias ? wf(l.qkv_b) : nullptr; q.mode = 0; q.out = qkv; q.ldo = QKV;
    timed(Q3A_KC_GEMM, 2.0 * QKV * H, [&] { KCHK(launch_skinny(q, precise(), ks)); });
    if (S * d.n_kv >= k_dattn_batched_min_wgs) {
      // the group alone fills the chip: one workgroup per (sequence, kv head) walks all keys and writes the context itself
      if (b16) { da.out16 = reinterpret_cast<uint16_t*>(s_ctx_g(grp)); da.out_frag = 1; } else da.out = s_ctx_g(grp);
      da.trim_prologue = min_P_ < 2 * 128;  // some sequence is shorter than the kernel's two prologue key tiles
      timed(Q3A_KC_DECODE_ATTN, 0, [&] { KCHK(launch_decode_attn_batched(da, S, kv_f32(), ks)); });
    } else {
      timed(Q3A_KC_DECODE_ATTN, 0, [&] { KCHK(launch_decode_attn(da, S, kv_f32(), ks)); });
      timed(Q3A_KC_DECODE_ATTN, 0, [&] { KCHK(launch_attn_combine(da.pm, da.pl, da.po, live_nsplit_, S, d.n_q, s_ctx_g(grp), ks, b16 ? reinterpret_cast<uint16_t*>(s_ctx_g(grp)) : nullptr, b16)); });
    }
    SkinnyArgs o{};
    o.fast_math = precise() ? 0 : 1;
    o.x = s_ctx_g(grp); o.x16 = b16 ? reinterpret_cast<uint16_t*>(s_ctx_g(grp)) : nullptr; o.x16_frag = b16; o.ldx = QD; o.S = S; o.W = wh(l.o_w); o.N = H; o.K = QD;
    o.bias = o_bias ? wf(l.o_b) : nullptr; o.mode = 1; o.out = x; o.ldo = H; o.resid = x;
    if (pre) { o.next_w = wf(l.post_ln); o.next_xw16f = nn_x_g(grp); o.next_ss = nn_ss_g(grp); }
    o.qsplit = b16 && skinny_q();
    timed(Q3A_KC_GEMM, 2.0 * H * QD, [&] { KCHK(launch_skinny(o, precise(), ks)); });
    SkinnyArgs u{};
    u.fast_math = precise() ? 0 : 1;
    u.x = x; u.ldx = H; u.S = S; u.eps = d.rms_eps; u.W = wh(l.gu_w); u.N = 2 * I; u.K = H;
    if (pre) { u.xw16f = nn_x_g(grp); u.ss_parts = nn_ss_g(grp); u.ss_nparts = nn_parts(); }
    else u.rms_w = wf(l.post_ln);
    u.bias = mlp_bias ? wf(l.gu_b) : nullptr; u.mode = 2; u.out = s_act_g(grp); u.out16 = b16 ? reinterpret_cast<uint16_t*>(s_act_g(grp)) : nullptr; u.out16_frag = b16; u.ldo = I;
    u.n_cu = n_cu;
    u.glu_hp3 = k_skinny_glu_hp3;
    timed(Q3A_KC_GEMM, 4.0 * I * H, [&] { KCHK(launch_skinny(u, precise(), ks)); });
    SkinnyArgs dn{};
    dn.fast_math = precise() ? 0 : 1;
    dn.x = s_act_g(grp); dn.x16 = b16 ? reinterpret_cast<uint16_t*>(s_act_g(grp)) : nullptr; dn.x16_frag = b16; dn.ldx = I; dn.S = S; dn.W = wh(l.down_w); dn.N = H; dn.K = I;
    dn.bias = mlp_bias ? wf(l.down_b) : nullptr; dn.mode = 1; dn.out = x; dn.ldo = H; dn.resid = x;
    dn.qsplit = b16 && skinny_q();
    if (pre && li + 1 < d.dec_layers) {  // (the last layer feeds the final norm + lm_head, which read x_dec)
      dn.next_w = wf(L.dec[li + 1].in_ln); dn.next_xw16f = nn_x_g(grp); dn.next_ss = nn_ss_g(grp);
    }
    timed(Q3A_KC_GEMM, 2.0 * H * I, [&] { KCHK(launch_skinny(dn, precise(), ks)); });
  }

  // one greedy-loop iteration for all sequences (inference.rs:160-200).  Batched path: the groups of `gsize` sequences are
  // independent chains of latency-bound kernels (5 per layer), so every group runs on its own stream -- forked from and
  // joined back into the engine's stream with events, which a hipGraph capture records as parallel branches: one group's
  // KV-streaming attention overlaps another group's weight-streaming GEMMs.  (Profiling runs keep everything on one stream.)
  void enqueue_decode_step() {
    const int ng = n_groups(B);
    const bool chains = ng > 1 && !prof && k_parallel_groups != 0;
    if (chains) {
      ensure_chain_streams(ng - 1);
      HIPCHK(hipEventRecord(fork_ev, stream));
    }
    for (int g = 0; g < ng; ++g) {
      const int s0 = ng == 1 ? 0 : g * gsize;
      const int S = ng == 1 ? B : std::min(gsize, B - s0);
      hipStream_t ks = (chains && g > 0) ? chain_streams[g - 1] : stream;
      if (chains && g > 0) HIPCHK(hipStreamWaitEvent(ks, fork_ev, 0));
      for (int li = 0; li < d.dec_layers; ++li) decode_layer(li, g, s0, S, ks);
      if (chains && g > 0) {
        HIPCHK(hipEventRecord(join_ev[g - 1], ks));
        HIPCHK(hipStreamWaitEvent(stream, join_ev[g - 1], 0));
      }
    }
    run_head(1);
  }
  void ensure_chain_streams(int n) {
    while ((int)chain_streams.size() < n) {
      hipStream_t s2;
      hipEvent_t e2;
      HIPCHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
      chain_streams.push_back(s2);
      join_ev.push_back(e2);
    }
    if (!fork_ev) HIPCHK(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
  }

  std::string make_graph_sig() const {
    // everything the captured step's launches depend on: geometry, latched knobs and the address of EVERY step-visible buffer
    // (FNV-1a over step_bufs(): the same list the reallocation sweep of setup_prompts walks)
    uint64_t h = 1469598103934665603ull;
    for (const DevBuf* sb : step_bufs()) {
      uint64_t v = (uint64_t)(uintptr_t)sb->p;
      for (int i = 0; i < 8; ++i) { h ^= (v >> (8 * i)) & 0xff; h *= 1099511628211ull; }
    }
    char buf[256];
    // (one field per latched knob, as latched: packing several into one integer let distinct settings collide -- ADVICE r5)
    snprintf(buf, sizeof(buf), "%d.%d.%d.%d.%d.%d.%d.%d/%d/%d/%d/%p/%016llx", B, gsize, k_parallel_groups, k_skinny_q,
             k_dattn_batched_min_wgs, k_skinny_glu_hp3, (int)(min_P_ < 256), (int)head_logits_, live_nsplit_, max_ctx, max_new, (const void*)arena, (unsigned long long)h);
    return buf;
  }

  // does any group of the batch run the key-split attention (GEMV path, or a group too small for the batched kernel)?
  bool uses_key_splits() const {
    if (B <= kGemvMaxSeq) return true;
    const int ng = n_groups(B);
    const int smallest = ng == 1 ? B : std::min(gsize, B - (ng - 1) * gsize);
    return smallest * d.n_kv < k_dattn_batched_min_wgs;
  }
  hipGraphExec_t graph_for_step() {
    const std::string sig = make_graph_sig();
    for (size_t i = 0; i < graph_cache.size(); ++i)
      if (graph_cache[i].first == sig) {  // LRU: a hit moves the entry to the back, eviction takes the front
        if (i + 1 != graph_cache.size()) std::rotate(graph_cache.begin() + i, graph_cache.begin() + i + 1, graph_cache.end());
        return graph_cache.back().second;
      }
    hipGraph_t g = nullptr;
    hipGraphExec_t exec = nullptr;
    HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    try {
      enqueue_decode_step();
    } catch (...) {
      hipGraph_t tmp = nullptr;
      (void)hipStreamEndCapture(stream, &tmp);
      if (tmp) (void)hipGraphDestroy(tmp);
      throw;
    }
    HIPCHK(hipStreamEndCapture(stream, &g));
    const hipError_t ie = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    HIPCHK(ie);
    if (graph_cache.size() >= kGraphCacheMax) {
      // the natural-EOS loop does not synchronise between steps, so the evicted exec may still be executing: it is only
      // destroyed after the next stream synchronisation (drain_retired_graphs)
      retired_graphs.push_back(graph_cache.front().second);
      graph_cache.erase(graph_cache.begin());
    }
    graph_cache.emplace_back(sig, exec);
    return exec;
  }
  std::vector<hipGraphExec_t> retired_graphs;
  // call only when the stream is known to be idle
  void drain_retired_graphs() {
    for (auto ge : retired_graphs) (void)hipGraphExecDestroy(ge);
    retired_graphs.clear();
  }
  // every cached graph holds raw device addresses: none may outlive a reallocation of ANY step-visible buffer (the signature
  // names only some of them).  Call with the stream idle.
  void drop_graphs() {
    drain_retired_graphs();
    for (auto& ge : graph_cache) (void)hipGraphExecDestroy(ge.second);
    graph_cache.clear();
  }

  void decode_steps(int n) {
    if (!have_prefill) fail("decode: no prefill state");
    if (n <= 0) return;
    const bool eager = !opts.use_graph || prof;
    const int kps = dattn_keys_per_split(kv_f32());
    // (Round 6 measured several steps per graph for the fixed-length mode -- a kernel trace shows 8.4 us between a step's finalize and
    // the next step's first GEMV against ~1.4 us inside a graph -- and un-traced it does not pay: one clip 592 -> 590 / 592 / 599 /
    // 607 / 610 us per step at 2 / 4 / 8 / 16 / 32 steps per graph, 32 clips 1131 -> 1126: the seam is the tracer's.  Code at commit
    // 03b3a8b, profiles/r6_ab_decode_steps_per_graph.txt.)
    for (int i = 0; i < n; ++i) {
      // this step feeds the token at position <= pos_hi_ and attends keys 0 .. pos_hi_
      live_nsplit_ = uses_key_splits() ? std::min(attn_nsplit, pos_hi_ / kps + 1) : attn_nsplit;
      if (eager) enqueue_decode_step();
      else HIPCHK(hipGraphLaunch(graph_for_step(), stream));
      ++pos_hi_;
    }
    if (eager) HIPCHK(hipGetLastError());
  }

  // steps 2-8 on the resident batch
  // prompts of the resident batch (inference.rs:104-137, 215-257) and every table / buffer of the prefill + decode: host integers
  // only (T[] of set_batch), so this runs before the batch's first kernel is enqueued
  void prepare_prompts(const int32_t* lang_ids, int n_prefix, int max_new_req, int fixed_new) {
    if (B <= 0) fail("q3a_run_resident: no batch uploaded");
    if (fixed_new > 0) max_new_req = fixed_new;
    if (max_new_req <= 0) max_new_req = opts.max_new_tokens;
    std::vector<int32_t> ids_v, lens(B);
    for (int s = 0; s < B; ++s) {
      int32_t len = 0;
      q3a_build_prompt(T[s], lang_ids, n_prefix, nullptr, &len);
      size_t o = ids_v.size();
      ids_v.resize(o + len);
      q3a_build_prompt(T[s], lang_ids, n_prefix, ids_v.data() + o, &len);
      lens[s] = len;
    }
    setup_prompts(ids_v.data(), lens.data(), B, max_new_req, true);
  }

  // mel_enqueued: upload_ptrs_and_mel() has put the (upload-overlapped) log-mel on the stream and recorded ev[0] / ev[1].
  // The caller has run prepare_prompts() for this batch.
  void run_resident(int fixed_new, bool mel_enqueued = false) {
    if (B <= 0) fail("q3a_run_resident: no batch uploaded");
    if (!mel_enqueued) {
      HIPCHK(hipEventRecord(ev[0], stream));
      run_mel();
      HIPCHK(hipEventRecord(ev[1], stream));
    }
    run_encoder();
    HIPCHK(hipEventRecord(ev[2], stream));
    run_prefill();
    HIPCHK(hipEventRecord(ev[3], stream));
    // the prefill already produced token 0; every decode step feeds one token and yields the next
    int steps = 0;
    static const bool host_timing = [] { const char* e = getenv("Q3A_DEBUG_HOST_TIMING"); return e && atoi(e) != 0; }();
    const auto h0 = std::chrono::steady_clock::now();
    if (fixed_new > 0) {
      steps = fixed_new - 1;
      decode_steps(steps);
      if (host_timing) fprintf(stderr, "[q3a host timing] %d graph launches enqueued in %.1f us\n", steps,
                               std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count());
    } else {
      // Natural EOS (inference.rs:160-167).  The stop condition of the whole batch is evaluated on the device
      // (argmax_finalize: n_done == B) and published to pinned host memory together with a progress counter; the host keeps
      // `ahead` graph replays enqueued in front of the device and reads those two words -- no stream synchronisation and no
      // D2H copy inside the loop, so the device never idles between steps.  Steps enqueued past a sequence's EOS only
      // append ids that fetch_ids cuts off (each utterance ends at ITS first EOS); at most `ahead` whole-batch steps are
      // wasted after the last sequence finishes.
      const int ahead = std::max(1, knobs().eos_run_ahead.load());
      const int limit = max_new - 1;
      unsigned spins = 0;
      while (steps < limit) {
        if (__atomic_load_n(&host_prog[1], __ATOMIC_RELAXED)) break;
        const int fin = __atomic_load_n(&host_prog[0], __ATOMIC_RELAXED);  // finalize launches completed (prefill = 1)
        if (steps - std::max(fin - 1, 0) < ahead) {
          decode_steps(1);
          ++steps;
          spins = 0;
        } else if (++spins > 64) {
          sched_yield();
          if ((spins & 0xffff) == 0) {  // a device fault must end the wait: the progress words would never move again
            const hipError_t q = hipStreamQuery(stream);
            if (q != hipSuccess && q != hipErrorNotReady) HIPCHK(q);
            if (q == hipSuccess && steps - std::max(__atomic_load_n(&host_prog[0], __ATOMIC_RELAXED) - 1, 0) >= ahead)
              fail("greedy loop: the stream drained but the device-side progress counter did not advance");
          }
        }
      }
    }
    HIPCHK(hipEventRecord(ev[4], stream));
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipGetLastError());
    float ms;
    HIPCHK(hipEventElapsedTime(&ms, ev[0], ev[1])); timings.mel_ms = ms;
    HIPCHK(hipEventElapsedTime(&ms, ev[1], ev[2])); timings.encoder_ms = ms;
    HIPCHK(hipEventElapsedTime(&ms, ev[2], ev[3])); timings.prefill_ms = ms;
    HIPCHK(hipEventElapsedTime(&ms, ev[3], ev[4])); timings.decode_ms = ms;
    HIPCHK(hipEventElapsedTime(&ms, ev[0], ev[4])); timings.total_ms = ms;
    timings.decode_steps = steps; timings.batch = B; timings.total_audio_tokens = total_T; timings.total_prompt_tokens = total_P;
  }

  void fetch_ids(int32_t* out, int stride, int32_t* out_lens) {
    if (!have_prefill) fail("q3a_fetch_ids: nothing generated");
    std::vector<int> all((size_t)B * max_new), sc(B);
    HIPCHK(hipMemcpy(all.data(), out_ids.p, all.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(sc.data(), step_count.p, (size_t)B * 4, hipMemcpyDeviceToHost));
    for (int s = 0; s < B; ++s) {
      int n = std::min(sc[s], max_new), len = 0;
      for (; len < n; ++len) {
        int t = all[(size_t)s * max_new + len];
        if (!fixed_mode_ && (t == kEos0 || t == kEos1)) break;  // generated_ids excludes EOS (inference.rs:163-167)
      }
      out_lens[s] = len;
      for (int i = 0; i < std::min(len, stride); ++i) out[(size_t)s * stride + i] = all[(size_t)s * max_new + i];
    }
  }
  bool fixed_mode_ = false;
  bool head_logits_ = true;  // store the logits of a batched decode step (step API, debug taps); off inside run_resident

  ~q3a_engine() {
    drop_graphs();
    for (auto cs : chain_streams) (void)hipStreamDestroy(cs);
    for (auto ce : join_ev) (void)hipEventDestroy(ce);
    if (fork_ev) (void)hipEventDestroy(fork_ev);
    DevBuf* bufs[] = {&dft, &filt_t, &pos_emb, &rope_cos, &rope_sin, &pcm, &d_pcm_off, &d_n_samples, &d_mel_off, &d_n_frames,
                      &mel, &gmax, &d_chunk_utt, &d_chunk_frame0, &conv1, &conv2, &conv3, &conv3_rowmap, &convout_rowmap,
                      &enc_x, &enc_ln, &enc_qkv, &enc_ctx, &enc_ffn, &enc_segs, &audio_embeds, &ids, &audio_rowmap, &row_seq,
                      &row_pos, &dec_segs, &last_rows, &dec_x, &dec_ln, &dec_qkv, &dec_ctx, &dec_act, &kcache, &vcache, &x_dec,
                      &d_pos, &next_tok, &out_ids, &step_count, &done, &s_ln, &s_qkv, &s_ctx, &s_act, &logits, &forced_tok, &part_val, &part_idx, &attn_pm, &attn_pl, &attn_po,
                      &enc_ctx16, &dec_ctx16, &dec_q16, &zero_page, &rope_cur, &nn_x, &nn_ss, &n_done};
    for (auto* b : bufs) b->release();
    for (auto& kv : taps) kv.second.release();
    if (own_arena && arena) (void)hipFree(arena);
    if (host_prog) (void)hipHostFree(host_prog);
    if (up_stream) { (void)hipStreamSynchronize(up_stream); (void)hipStreamDestroy(up_stream); }
    if (pin_p) (void)hipHostFree(pin_p);
    for (auto ue : up_ev) (void)hipEventDestroy(ue);
    if (up_t0) (void)hipEventDestroy(up_t0);
    if (up_t1) (void)hipEventDestroy(up_t1);
    for (auto& x : ev)
      if (x) (void)hipEventDestroy(x);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

// audio rows: dec_x[audio_rowmap[t]] = audio_embeds[t]
void q3a_engine::scatter_audio_rows() {
  KCHK(launch_scatter_rows(audio_embeds.as<float>(), audio_rowmap.as<int>(), total_T, d.hidden, dec_x.as<float>(), stream));
}

// =========================================================================================
// C ABI
// =========================================================================================
#define Q3A_TRY(e) try {
#define Q3A_CATCH(e)                                 \
  }                                                  \
  catch (const std::exception& ex) {                 \
    if (e) (e)->err = ex.what();                     \
    g_last_error = ex.what();                        \
    return 1;                                        \
  }                                                  \
  catch (...) {                                      \
    if (e) (e)->err = "unknown error";               \
    g_last_error = "unknown error";                  \
    return 1;                                        \
  }                                                  \
  return 0;

extern "C" {

void q3a_opts_default(q3a_opts* o) {
  memset(o, 0, sizeof(*o));
  o->precise = 0;
  o->max_new_tokens = 4096;
  o->use_graph = 1;
  o->debug_taps = 0;
}

static int32_t create_impl(const char* model_dir, int32_t device, void* dev_arena, uint64_t bytes, const q3a_opts* opts,
                           q3a_engine** out) {
  q3a_engine* e = nullptr;
  try {
    if (!model_dir || !out) fail("null argument");
    e = new q3a_engine();
    e->init_common(model_dir, device, opts);
    if (dev_arena) {
      if (bytes != e->L.total) fail("q3a_engine_create_from_arena: arena size does not match config.json");
      e->arena = (uint8_t*)dev_arena;
      e->own_arena = false;
    } else {
      Checkpoint ck(model_dir);
      std::vector<uint8_t> host(e->L.total);
      pack_arena(e->d, e->L, ck, host.data());
      HIPCHK(hipMalloc((void**)&e->arena, e->L.total));
      e->own_arena = true;
      HIPCHK(hipMemcpy(e->arena, host.data(), e->L.total, hipMemcpyHostToDevice));
    }
    e->check_arena_header();
    e->init_tables();
    *out = e;
    return 0;
  } catch (const std::exception& ex) {
    g_last_error = ex.what();
    delete e;
    if (out) *out = nullptr;
    return 1;
  }
}

int32_t q3a_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int32_t q3a_engine_create(const char* model_dir, int32_t device, const q3a_opts* opts, q3a_engine** out) {
  return create_impl(model_dir, device, nullptr, 0, opts, out);
}
int32_t q3a_engine_create_from_arena(const char* model_dir, int32_t device, void* device_arena, uint64_t bytes,
                                     const q3a_opts* opts, q3a_engine** out) {
  if (!device_arena) { g_last_error = "null arena"; return 1; }
  return create_impl(model_dir, device, device_arena, bytes, opts, out);
}

int32_t q3a_arena_bytes(const char* model_dir, uint64_t* bytes) {
  q3a_engine* e = nullptr;
  Q3A_TRY(e)
  Dims d = parse_config_file(std::string(model_dir) + "/config.json");
  validate_dims(d);
  *bytes = plan_arena(d).total;
  Q3A_CATCH(e)
}
int32_t q3a_arena_pack(const char* model_dir, void* host_dst, uint64_t bytes) {
  q3a_engine* e = nullptr;
  Q3A_TRY(e)
  Dims d = parse_config_file(std::string(model_dir) + "/config.json");
  validate_dims(d);
  ArenaLayout L = plan_arena(d);
  if (bytes != L.total) fail("q3a_arena_pack: buffer size does not match q3a_arena_bytes");
  Checkpoint ck(model_dir);
  pack_arena(d, L, ck, (uint8_t*)host_dst);
  Q3A_CATCH(e)
}

void q3a_engine_destroy(q3a_engine* e) { delete e; }
const char* q3a_last_error(const q3a_engine* e) { return e ? e->err.c_str() : g_last_error.c_str(); }

int32_t q3a_get_dims(const q3a_engine* e, q3a_dims* o) {
  if (!e || !o) return 1;
  const Dims& d = e->d;
  memset(o, 0, sizeof(*o));
  o->enc_d_model = d.enc_d; o->enc_layers = d.enc_layers; o->enc_heads = d.enc_heads; o->enc_ffn = d.enc_ffn;
  o->num_mel_bins = d.n_mels; o->n_window = d.n_window; o->n_window_infer = d.n_window_infer; o->conv_channels = d.conv_ch;
  o->enc_output_dim = d.enc_out; o->max_source_positions = d.max_source_positions;
  o->vocab_size = d.vocab; o->hidden_size = d.hidden; o->intermediate_size = d.inter; o->dec_layers = d.dec_layers;
  o->num_q_heads = d.n_q; o->num_kv_heads = d.n_kv; o->head_dim = d.head_dim; o->tie_word_embeddings = d.tie_embeddings;
  o->mrope_interleaved = d.mrope_interleaved;
  for (size_t i = 0; i < 4 && i < d.mrope_section.size(); ++i) o->mrope_section[i] = d.mrope_section[i];
  o->rms_norm_eps = d.rms_eps; o->rope_theta = d.rope_theta;
  return 0;
}

int32_t q3a_weights_rounded(const q3a_engine* e) { return e && (e->arena_flags & kFlagWeightsRounded) ? 1 : 0; }

int64_t q3a_num_frames(int64_t n_samples) { return (n_samples + 159) / 160; }
int32_t q3a_num_audio_tokens(const q3a_engine* e, int64_t n_frames) { return e ? e->d.audio_tokens(n_frames) : -1; }

int32_t q3a_build_prompt(int32_t num_audio_tokens, const int32_t* lang_prefix_ids, int32_t n_prefix, int32_t* ids,
                         int32_t* len) {
  // inference.rs:215-257
  static const int32_t head[9] = {151644, 8948, 198, 151645, 198, 151644, 872, 198, 151669};
  static const int32_t tail[6] = {151670, 151645, 198, 151644, 77091, 198};
  if (!lang_prefix_ids) n_prefix = 0;
  int32_t n = 9 + num_audio_tokens + 6 + n_prefix;
  if (len) *len = n;
  if (!ids) return 0;
  int32_t* p = ids;
  for (int i = 0; i < 9; ++i) *p++ = head[i];
  for (int i = 0; i < num_audio_tokens; ++i) *p++ = kAudioPad;
  for (int i = 0; i < 6; ++i) *p++ = tail[i];
  for (int i = 0; i < n_prefix; ++i) *p++ = lang_prefix_ids[i];
  return 0;
}

int32_t q3a_upload_pcm(q3a_engine* e, const float* pcm16k, const int64_t* n_samples, int32_t B) {
  if (!e) return 1;
  Q3A_TRY(e)
  HIPCHK(hipSetDevice(e->device));
  e->upload_pcm(pcm16k, n_samples, B);
  Q3A_CATCH(e)
}

int32_t q3a_mel(q3a_engine* e, const float* pcm16k, const int64_t* n_samples, int32_t B, float* mel_out,
                int32_t* n_frames_out) {
  if (!e) return 1;
  Q3A_TRY(e)
  HIPCHK(hipSetDevice(e->device));
  e->upload_pcm(pcm16k, n_samples, B);
  e->run_mel();
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipGetLastError());
  if (n_frames_out)
    for (int u = 0; u < B; ++u) n_frames_out[u] = e->n_frames[u];
  if (mel_out) {
    size_t total = 0;
    for (int u = 0; u < B; ++u) total += (size_t)e->n_frames[u] * e->d.n_mels;
    HIPCHK(hipMemcpy(mel_out, e->mel.p, total * 4, hipMemcpyDeviceToHost));
  }
  Q3A_CATCH(e)
}

int32_t q3a_encode(q3a_engine* e, float* audio_embeds_out, int32_t* T_out) {
  if (!e) return 1;
  Q3A_TRY(e)
  HIPCHK(hipSetDevice(e->device));
  e->run_encoder();
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipGetLastError());
  if (T_out)
    for (int u = 0; u < e->B; ++u) T_out[u] = e->T[u];
  if (audio_embeds_out)
    HIPCHK(hipMemcpy(audio_embeds_out, e->audio_embeds.p, (size_t)e->total_T * e->d.enc_out * 4, hipMemcpyDeviceToHost));
  Q3A_CATCH(e)
}

int32_t q3a_prefill(q3a_engine* e, const int32_t* ids, const int32_t* lens, int32_t B, float* last_logits_out,
                    int32_t* next_ids) {
  if (!e) return 1;
  Q3A_TRY(e)
  HIPCHK(hipSetDevice(e->device));
  e->fixed_mode_ = false;
  e->head_logits_ = true;
  e->setup_prompts(ids, lens, B, e->opts.max_new_tokens);
  e->run_prefill();
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipGetLastError());
  if (last_logits_out) HIPCHK(hipMemcpy(last_logits_out, e->logits.p, (size_t)B * e->d.vocab * 4, hipMemcpyDeviceToHost));
  if (next_ids) HIPCHK(hipMemcpy(next_ids, e->next_tok.p, (size_t)B * 4, hipMemcpyDeviceToHost));
  Q3A_CATCH(e)
}

int32_t q3a_decode_step(q3a_engine* e, int32_t* next_ids, uint8_t* done, float* logits_out) {
  if (!e) return 1;
  Q3A_TRY(e)
  HIPCHK(hipSetDevice(e->device));
  {
    // capacity guard: position of the token being fed must stay inside the cache
    std::vector<int> pos(e->B);
    HIPCHK(hipMemcpy(pos.data(), e->d_pos.p, (size_t)e->B * 4, hipMemcpyDeviceToHost));
    for (int s = 0; s < e->B; ++s)
      if (pos[s] + 1 > e->max_ctx) fail("q3a_decode_step: KV cache capacity exhausted (max_new_tokens reached)");
  }
  e->head_logits_ = true;
  e->decode_steps(1);
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipGetLastError());
  if (next_ids) HIPCHK(hipMemcpy(next_ids, e->next_tok.p, (size_t)e->B * 4, hipMemcpyDeviceToHost));
  if (done) HIPCHK(hipMemcpy(done, e->done.p, (size_t)e->B, hipMemcpyDeviceToHost));
  if (logits_out) HIPCHK(hipMemcpy(logits_out, e->logits.p, (size_t)e->B * e->d.vocab * 4, hipMemcpyDeviceToHost));
  Q3A_CATCH(e)
}

int32_t q3a_set_next_tokens(q3a_engine* e, const int32_t* ids, int32_t B) {
  if (!e) return 1;
  Q3A_TRY(e)
  HIPCHK(hipSetDevice(e->device));
  if (!e->have_prefill || B != e->B) fail("q3a_set_next_tokens: no decode state / batch mismatch");
  for (int s = 0; s < B; ++s)
    if (ids[s] < 0 || ids[s] >= e->d.vocab) fail("q3a_set_next_tokens: token id out of range");
  HIPCHK(hipMemcpy(e->forced_tok.p, ids, (size_t)B * 4, hipMemcpyHostToDevice));
  KCHK(launch_set_tokens(e->forced_tok.as<int>(), B, e->wh(e->L.embed), e->d.hidden, e->x_dec.as<float>(),
                         e->next_tok.as<int>(), e->stream, e->first_layer_norm_out()));
  HIPCHK(hipStreamSynchronize(e->stream));
  Q3A_CATCH(e)
}

int32_t q3a_run_resident(q3a_engine* e, const int32_t* lang_prefix_ids, int32_t n_prefix, int32_t max_new,
                         int32_t fixed_new_tokens) {
  if (!e) return 1;
  Q3A_TRY(e)
  HIPCHK(hipSetDevice(e->device));
  e->fixed_mode_ = fixed_new_tokens > 0;
  e->head_logits_ = e->opts.debug_taps != 0;
  e->have_mel = e->have_enc = e->have_prefill = false;
  e->prepare_prompts(lang_prefix_ids, n_prefix, max_new, fixed_new_tokens);
  e->run_resident(fixed_new_tokens);
  Q3A_CATCH(e)
}

int32_t q3a_fetch_ids(q3a_engine* e, int32_t* out_ids, int32_t stride, int32_t* out_lens) {
  if (!e) return 1;
  Q3A_TRY(e)
  HIPCHK(hipSetDevice(e->device));
  e->fetch_ids(out_ids, stride, out_lens);
  Q3A_CATCH(e)
}

int32_t q3a_transcribe_batch_ptrs(q3a_engine* e, const float* const* pcm16k, const int64_t* n_samples, int32_t B,
                                  const int32_t* lang_prefix_ids, int32_t n_prefix, int32_t max_new, int32_t fixed_new_tokens,
                                  int32_t* out_ids, int32_t stride, int32_t* out_lens) {
  if (!e) return 1;
  Q3A_TRY(e)
  if (!pcm16k || !n_samples || B < 1) fail("q3a_transcribe_batch: bad argument");
  for (int u = 0; u < B; ++u)
    if (!pcm16k[u]) fail("q3a_transcribe_batch: null utterance pointer");
  const auto w0 = std::chrono::steady_clock::now();
  HIPCHK(hipSetDevice(e->device));
  e->fixed_mode_ = fixed_new_tokens > 0;
  e->head_logits_ = e->opts.debug_taps != 0;
  e->set_batch(n_samples, B);  // geometry first: the prompts only need the lengths, and their set-up synchronises an idle stream
  e->prepare_prompts(lang_prefix_ids, n_prefix, max_new, fixed_new_tokens);
  e->upload_ptrs_and_mel(pcm16k, n_samples, B);
  e->run_resident(fixed_new_tokens, true);
  e->fetch_ids(out_ids, stride, out_lens);
  if (e->io.mode != 0) {
    // (a timing probe must not be able to fail the call: the marker sits on the copy stream, which fetch_ids did not synchronise)
    float ms = 0.f;
    if (hipEventSynchronize(e->up_t1) != hipSuccess || hipEventElapsedTime(&ms, e->up_t0, e->up_t1) != hipSuccess) { (void)hipGetLastError(); ms = 0.f; }
    e->io.h2d_ms = ms;
  }
  e->io.wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - w0).count();
  Q3A_CATCH(e)
}

int32_t q3a_transcribe_batch(q3a_engine* e, const float* pcm16k, const int64_t* n_samples, int32_t B,
                             const int32_t* lang_prefix_ids, int32_t n_prefix, int32_t max_new, int32_t fixed_new_tokens,
                             int32_t* out_ids, int32_t stride, int32_t* out_lens) {
  if (!e) return 1;
  if (!pcm16k || !n_samples || B < 1) { e->err = "q3a_transcribe_batch: bad argument"; return 1; }
  std::vector<const float*> ptrs((size_t)B);
  int64_t off = 0;
  for (int u = 0; u < B; ++u) { ptrs[u] = pcm16k + off; off += n_samples[u]; }
  return q3a_transcribe_batch_ptrs(e, ptrs.data(), n_samples, B, lang_prefix_ids, n_prefix, max_new, fixed_new_tokens, out_ids, stride, out_lens);
}

int32_t q3a_io_timings_last(const q3a_engine* e, q3a_io_timings* out) {
  if (!e || !out) return 1;
  *out = e->io;
  return 0;
}

int32_t q3a_stage_timings(const q3a_engine* e, q3a_timings* out) {
  if (!e || !out) return 1;
  *out = e->timings;
  return 0;
}

int32_t q3a_profile_decode_step(q3a_engine* e, q3a_kernel_profile* out) {
  if (!e || !out) return 1;
  Q3A_TRY(e)
  HIPCHK(hipSetDevice(e->device));
  if (!e->have_prefill) fail("q3a_profile_decode_step: no decode state");
  std::vector<ProfEvent> evs;
  e->prof = &evs;
  try {
    e->decode_steps(1);  // eager while `prof` is set: every launch between a pair of events
    HIPCHK(hipStreamSynchronize(e->stream));
  } catch (...) {
    e->prof = nullptr;
    throw;
  }
  e->prof = nullptr;
  memset(out, 0, sizeof(*out));
  for (auto& pe : evs) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, pe.a, pe.b));
    out->total_us[pe.kclass] += ms * 1000.f;
    out->launches[pe.kclass] += 1;
    out->weight_bytes[pe.kclass] += pe.bytes;
    (void)hipEventDestroy(pe.a);
    (void)hipEventDestroy(pe.b);
  }
  Q3A_CATCH(e)
}

int32_t q3a_profile_weight_stream(q3a_engine* e, int32_t reps, float* avg_us, double* bytes_per_launch, int32_t* launches) {
  if (!e) return 1;
  Q3A_TRY(e)
  HIPCHK(hipSetDevice(e->device));
  if (!e->have_prefill) fail("q3a_profile_weight_stream: no decode state");
  if (e->B > kGemvMaxSeq) fail("q3a_profile_weight_stream: the GEMV path serves at most 2 sequences");
  if (reps < 1) reps = 1;
  const Dims& d = e->d;
  const int S = e->B, H = d.hidden, I = d.inter, QKV = d.qkv_dim();
  auto sweep = [&]() {
    for (int li = 0; li < d.dec_layers; ++li) {
      const DecLayerOff& l = e->L.dec[li];
      GemvArgs g{};
      g.fast_math = e->precise() ? 0 : 1;
      g.x = e->x_dec.as<float>(); g.ldx = H; g.rms_w = e->wf(l.in_ln); g.eps = d.rms_eps; g.W = e->wh(l.qkv_w); g.N = QKV; g.K = H;
      g.mode = 0; g.out = e->s_qkv.as<float>(); g.ldo = QKV;
      KCHK(launch_gemv(g, S, e->stream));
      GemvArgs u{};
      u.fast_math = e->precise() ? 0 : 1;
      u.x = e->x_dec.as<float>(); u.ldx = H; u.rms_w = e->wf(l.post_ln); u.eps = d.rms_eps; u.W = e->wh(l.gu_w); u.N = 2 * I; u.K = H;
      u.mode = 2; u.out = e->s_act.as<float>(); u.ldo = I;
      KCHK(launch_gemv(u, S, e->stream));
    }
  };
  // replayed from a hipGraph like the decode step itself: an eager host loop issues ~6 us per launch, slower than
  // these kernels run, and would time the host
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
  try {
    for (int r = 0; r < reps; ++r) sweep();
  } catch (...) {
    hipGraph_t tmp = nullptr;
    (void)hipStreamEndCapture(e->stream, &tmp);
    if (tmp) (void)hipGraphDestroy(tmp);
    throw;
  }
  HIPCHK(hipStreamEndCapture(e->stream, &g));
  HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  (void)hipGraphDestroy(g);
  HIPCHK(hipGraphLaunch(ge, e->stream));  // warm-up (code objects, clocks)
  HIPCHK(hipStreamSynchronize(e->stream));
  hipEvent_t a, b;
  HIPCHK(hipEventCreate(&a));
  HIPCHK(hipEventCreate(&b));
  float ms = 0.f;
  constexpr int kReplays = 3;
  for (int rep = 0; rep < kReplays; ++rep) {  // average over the replays
    HIPCHK(hipEventRecord(a, e->stream));
    HIPCHK(hipGraphLaunch(ge, e->stream));
    HIPCHK(hipEventRecord(b, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, a, b));
    ms += t / kReplays;
  }
  HIPCHK(hipGetLastError());
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  (void)hipGraphExecDestroy(ge);
  const int n = reps * d.dec_layers * 2;
  if (avg_us) *avg_us = ms * 1000.f / (float)n;
  if (bytes_per_launch) *bytes_per_launch = (2.0 * QKV * H + 4.0 * I * H) / 2.0;
  if (launches) *launches = n;
  Q3A_CATCH(e)
}

int32_t q3a_debug_read(q3a_engine* e, const char* name, void* dst, uint64_t bytes, uint64_t* actual) {
  if (!e) return 1;
  Q3A_TRY(e)
  HIPCHK(hipSetDevice(e->device));
  auto it = e->taps.find(name);
  if (it == e->taps.end()) fail(std::string("q3a_debug_read: no tap named '") + name + "' (opts.debug_taps set?)");
  size_t n = e->tap_bytes[name];
  if (actual) *actual = n;
  if (dst) {
    if (bytes < n) fail("q3a_debug_read: destination too small");
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(dst, it->second.p, n, hipMemcpyDeviceToHost));
  }
  Q3A_CATCH(e)
}

int32_t q3a_debug_set(const char* key, int32_t value) {
  if (!key) return 1;
  Knobs& kn = knobs();
  if (strcmp(key, "gemm256_min_tiles") == 0) { kn.gemm256_min_tiles = value; return 0; }
  if (strcmp(key, "gemm256_persist") == 0) { kn.gemm256_persist = value; return 0; }
  if (strcmp(key, "gemm256_group_m") == 0) { kn.gemm256_group_m = value; return 0; }
  if (strcmp(key, "dattn_batched_min_wgs") == 0) { kn.dattn_batched_min_wgs = value; return 0; }
  if (strcmp(key, "decode_group_size") == 0) { kn.decode_group_size = value; return 0; }
  if (strcmp(key, "decode_parallel_groups") == 0) { kn.decode_parallel_groups = value; return 0; }
  if (strcmp(key, "fuse_qkrope") == 0) { kn.fuse_qkrope = value; return 0; }
  if (strcmp(key, "skinny_q") == 0) { kn.skinny_q = value; return 0; }
  if (strcmp(key, "eos_run_ahead") == 0) { kn.eos_run_ahead = value; return 0; }
  if (strcmp(key, "skinny_glu_hp3") == 0) { kn.skinny_glu_hp3 = value; return 0; }
  g_last_error = std::string("q3a_debug_set: unknown key '") + key + "'";
  return 1;
}

int32_t q3a_selftest_gemm(int32_t device, int32_t M, int32_t N, int32_t K, int32_t split, float* max_abs_err,
                          float* ref_abs_max) {
  q3a_engine* e = nullptr;
  Q3A_TRY(e)
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) fail("no HIP device available");
  HIPCHK(hipSetDevice(device));
  std::vector<float> X((size_t)M * K);
  std::vector<uint16_t> W((size_t)N * K);
  uint32_t st = 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : X) v = rnd();
  for (auto& v : W) { float f = rnd(); uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }  // truncation is fine for a test
  DevBuf dX, dW, dY, dR;
  dX.ensure(X.size() * 4); dW.ensure(W.size() * 2); dY.ensure((size_t)M * N * 4); dR.ensure((size_t)M * N * 4);
  HIPCHK(hipMemcpy(dX.p, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dW.p, W.data(), W.size() * 2, hipMemcpyHostToDevice));
  GemmEpilogue ep; ep.out = dY.as<float>(); ep.ldo = N;
  KCHK(launch_gemm(dX.as<float>(), K, dW.as<uint16_t>(), M, N, K, ep, false, split != 0, nullptr));
  launch_gemm_ref(dX.as<float>(), dW.as<uint16_t>(), dR.as<float>(), M, N, K, nullptr);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipGetLastError());
  std::vector<float> Y((size_t)M * N), R((size_t)M * N);
  HIPCHK(hipMemcpy(Y.data(), dY.p, Y.size() * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(R.data(), dR.p, R.size() * 4, hipMemcpyDeviceToHost));
  float me = 0.f, rm = 0.f;
  for (size_t i = 0; i < Y.size(); ++i) { me = std::max(me, std::fabs(Y[i] - R[i])); rm = std::max(rm, std::fabs(R[i])); }
  if (max_abs_err) *max_abs_err = me;
  if (ref_abs_max) *ref_abs_max = rm;
  dX.release(); dW.release(); dY.release(); dR.release();
  Q3A_CATCH(e)
}

int32_t q3a_selftest_gemm16(int32_t device, int32_t M, int32_t N, int32_t K, int32_t reps, float* max_abs_err,
                            float* ref_abs_max, float* avg_us_bf16, float* avg_us_f32) {
  q3a_engine* e = nullptr;
  Q3A_TRY(e)
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) fail("no HIP device available");
  HIPCHK(hipSetDevice(device));
  std::vector<float> X((size_t)M * K);
  std::vector<uint16_t> W((size_t)N * K);
  uint32_t st = 777u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
  auto trunc16 = [](float f) { uint32_t u; memcpy(&u, &f, 4); u &= 0xffff0000u; memcpy(&f, &u, 4); return f; };
  for (auto& v : X) v = trunc16(rnd());  // bf16-representable activations: the bf16 copy is exact
  for (auto& v : W) { float f = rnd(); uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
  DevBuf dX, dX16, dW, dY, dR;
  dX.ensure(X.size() * 4); dX16.ensure(X.size() * 2); dW.ensure(W.size() * 2); dY.ensure((size_t)M * N * 4); dR.ensure((size_t)M * N * 4);
  HIPCHK(hipMemcpy(dX.p, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dW.p, W.data(), W.size() * 2, hipMemcpyHostToDevice));
  KCHK(launch_to_bf16(dX.as<float>(), dX16.as<uint16_t>(), X.size(), nullptr));
  GemmEpilogue ep; ep.out = dY.as<float>(); ep.ldo = N;
  KCHK(launch_gemm16(dX16.as<uint16_t>(), K, dW.as<uint16_t>(), M, N, K, ep, false, nullptr));
  launch_gemm_ref(dX.as<float>(), dW.as<uint16_t>(), dR.as<float>(), M, N, K, nullptr);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipGetLastError());
  std::vector<float> Y((size_t)M * N), R((size_t)M * N);
  HIPCHK(hipMemcpy(Y.data(), dY.p, Y.size() * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(R.data(), dR.p, R.size() * 4, hipMemcpyDeviceToHost));
  float me = 0.f, rm = 0.f;
  for (size_t i = 0; i < Y.size(); ++i) { me = std::max(me, std::fabs(Y[i] - R[i])); rm = std::max(rm, std::fabs(R[i])); }
  // ---- the epilogue variants, against the reference product R ----
  {
    const int P = 7;  // addend period
    std::vector<float> bias(N), addend((size_t)P * N), resid((size_t)M * N);
    std::vector<int> rowmap(M);
    for (auto& v : bias) v = rnd();
    for (auto& v : addend) v = rnd();
    for (auto& v : resid) v = rnd();
    for (int m = 0; m < M; ++m) rowmap[m] = (m % 11 == 5) ? -1 : M - 1 - m;  // reversed rows, some dropped
    DevBuf dB, dA, dS, dM, dY2, dY16;
    dB.ensure(bias.size() * 4); dA.ensure(addend.size() * 4); dS.ensure(resid.size() * 4); dM.ensure(rowmap.size() * 4);
    dY2.ensure((size_t)M * N * 4); dY16.ensure((size_t)M * N * 2);
    HIPCHK(hipMemcpy(dB.p, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dA.p, addend.data(), addend.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dS.p, resid.data(), resid.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dM.p, rowmap.data(), rowmap.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dY2.p, 0, (size_t)M * N * 4));
    // (a) fp32 out: bias + periodic addend + row map + residual
    GemmEpilogue e2; e2.out = dY2.as<float>(); e2.ldo = N; e2.bias = dB.as<float>(); e2.addend = dA.as<float>(); e2.addend_period = P;
    e2.rowmap = dM.as<int>(); e2.resid = dS.as<float>();
    KCHK(launch_gemm16(dX16.as<uint16_t>(), K, dW.as<uint16_t>(), M, N, K, e2, false, nullptr));
    HIPCHK(hipDeviceSynchronize());
    std::vector<float> Y2((size_t)M * N);
    HIPCHK(hipMemcpy(Y2.data(), dY2.p, Y2.size() * 4, hipMemcpyDeviceToHost));
    std::vector<char> hit(M, 0);
    for (int m = 0; m < M; ++m) {
      const int o = rowmap[m];
      if (o < 0) continue;
      hit[o] = 1;
      for (int n = 0; n < N; ++n) {
        const float want = R[(size_t)m * N + n] + bias[n] + addend[(size_t)(m % P) * N + n] + resid[(size_t)o * N + n];
        me = std::max(me, std::fabs(Y2[(size_t)o * N + n] - want));
      }
    }
    for (int o = 0; o < M; ++o)
      if (!hit[o])
        for (int n = 0; n < N; ++n)
          if (Y2[(size_t)o * N + n] != 0.f) fail("selftest_gemm16: a row dropped by the row map was written");
    // (b) bf16 out: bias only; one bf16 ulp
    auto bf = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
    GemmEpilogue e3; e3.out16 = dY16.as<uint16_t>(); e3.ldo = N; e3.bias = dB.as<float>();
    KCHK(launch_gemm16(dX16.as<uint16_t>(), K, dW.as<uint16_t>(), M, N, K, e3, false, nullptr));
    HIPCHK(hipDeviceSynchronize());
    std::vector<uint16_t> Y16((size_t)M * N);
    HIPCHK(hipMemcpy(Y16.data(), dY16.p, Y16.size() * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < Y16.size(); ++i) {
      const float want = R[i] + bias[i % N];
      if (std::fabs(bf(Y16[i]) - want) > std::fabs(want) / 128.f + 1e-4f * std::max(rm, 1.f)) fail("selftest_gemm16: bf16 output with bias is off by more than an ulp");
    }
    // (c) SwiGLU pairs ([16 gate | 16 up] row blocks of W), bf16 out
    if (N % 32 == 0) {
      GemmEpilogue e4; e4.out16 = dY16.as<uint16_t>(); e4.ldo = N / 2;
      KCHK(launch_gemm16(dX16.as<uint16_t>(), K, dW.as<uint16_t>(), M, N, K, e4, true, nullptr));
      HIPCHK(hipDeviceSynchronize());
      HIPCHK(hipMemcpy(Y16.data(), dY16.p, (size_t)M * (N / 2) * 2, hipMemcpyDeviceToHost));
      for (int m = 0; m < M; ++m)
        for (int c = 0; c < N / 2; ++c) {
          const float g = R[(size_t)m * N + (c / 16) * 32 + c % 16], u = R[(size_t)m * N + (c / 16) * 32 + 16 + c % 16];
          const float want = g / (1.f + std::exp(-g)) * u;
          if (std::fabs(bf(Y16[(size_t)m * (N / 2) + c]) - want) > std::fabs(want) / 64.f + 1e-3f * std::max(rm * rm, 1.f))
            fail("selftest_gemm16: SwiGLU epilogue is off");
        }
    }
    dB.release(); dA.release(); dS.release(); dM.release(); dY2.release(); dY16.release();
  }
  if (max_abs_err) *max_abs_err = me;
  if (ref_abs_max) *ref_abs_max = rm;
  if (reps > 0) {
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    float ms = 0.f;
    HIPCHK(hipEventRecord(a, nullptr));
    for (int r = 0; r < reps; ++r) KCHK(launch_gemm16(dX16.as<uint16_t>(), K, dW.as<uint16_t>(), M, N, K, ep, false, nullptr));
    HIPCHK(hipEventRecord(b, nullptr));
    HIPCHK(hipEventSynchronize(b));
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    if (avg_us_bf16) *avg_us_bf16 = ms * 1000.f / reps;
    HIPCHK(hipEventRecord(a, nullptr));
    for (int r = 0; r < reps; ++r) KCHK(launch_gemm(dX.as<float>(), K, dW.as<uint16_t>(), M, N, K, ep, false, false, nullptr));
    HIPCHK(hipEventRecord(b, nullptr));
    HIPCHK(hipEventSynchronize(b));
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    if (avg_us_f32) *avg_us_f32 = ms * 1000.f / reps;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
  }
  dX.release(); dX16.release(); dW.release(); dY.release(); dR.release();
  Q3A_CATCH(e)
}

}  // extern "C"
