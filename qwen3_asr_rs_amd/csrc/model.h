// Host-side model description: config.json -> Dims, safetensors -> tensor views, and the layout /
// packing of the single device weight arena.  Pure C++ (no HIP): runs on CPU-only boxes too.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace q3a {

[[noreturn]] inline void fail(const std::string& msg) { throw std::runtime_error(msg); }

// Reference: src/config.rs:4-137 (serde defaults reproduced field by field).
struct Dims {
  // audio encoder
  int enc_d = 896, enc_layers = 18, enc_heads = 14, enc_ffn = 3584, n_mels = 128, max_source_positions = 1500,
      n_window = 50, n_window_infer = 800, conv_ch = 480, enc_out = 1024;
  // text decoder
  int vocab = 151936, hidden = 1024, inter = 3072, dec_layers = 28, n_q = 16, n_kv = 8, head_dim = 128;
  float rms_eps = 1e-6f;
  double rope_theta = 1e6;
  bool tie_embeddings = true;
  std::vector<int> mrope_section{24, 20, 20};
  bool mrope_interleaved = false;

  // derived
  int chunk_frames() const { return n_window * 2; }                       // audio_encoder.rs:83
  static int conv_len(int l) { return (l - 1) / 2 + 1; }                  // audio_encoder.rs:264
  static int feat_len(int l) { return conv_len(conv_len(conv_len(l))); }  // audio_encoder.rs:263-266
  int tokens_per_chunk() const { return feat_len(chunk_frames()); }       // 13
  int freq3() const { return feat_len(n_mels); }                          // 16
  int conv_out_in() const { return conv_ch * freq3(); }                   // 7680
  int audio_tokens(int64_t frames) const {                                // audio_encoder.rs:269-279
    int64_t full = frames / chunk_frames(), tail = frames % chunk_frames();
    int64_t t = full * tokens_per_chunk();
    if (tail > 0) t += feat_len((int)tail);
    return (int)t;
  }
  int chunks_per_window() const { return n_window_infer / chunk_frames(); }  // audio_encoder.rs:178-179
  int q_dim() const { return n_q * head_dim; }
  int kv_dim() const { return n_kv * head_dim; }
  int qkv_dim() const { return q_dim() + 2 * kv_dim(); }
};

Dims parse_config_file(const std::string& path);
void validate_dims(const Dims& d);  // throws with a message naming the unsupported dimension

// ---- safetensors (reference: src/weights.rs:10-142) ------------------------------------------------
enum class StDtype { BF16, F16, F32, I64 };
struct TensorView {
  StDtype dtype;
  std::vector<int64_t> shape;
  const uint8_t* data = nullptr;
  uint64_t nbytes = 0;
  int64_t numel() const {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return n;
  }
};

class Checkpoint {
 public:
  // model.safetensors, else model.safetensors.index.json -> sorted unique shards (weights.rs:10-58)
  explicit Checkpoint(const std::string& model_dir);
  ~Checkpoint();
  const TensorView& get(const std::string& key) const;      // throws "Weight not found: key" (weights.rs:196)
  const TensorView* get_opt(const std::string& key) const;  // weights.rs:200-212
  size_t size() const { return tensors_.size(); }

 private:
  struct Mapping { void* addr; size_t len; };
  std::vector<Mapping> maps_;
  std::map<std::string, TensorView> tensors_;
  void load_file(const std::string& path);
};

// ---- device weight arena ----------------------------------------------------------------------------
// One contiguous allocation, 256-B aligned sub-tensors.  Matrices are bf16 (the published checkpoints
// are bf16, so this is lossless w.r.t. the reference, which widens them to f32: weights.rs:74-89);
// vectors (biases, norm weights) and conv1 are f32.  Layout choices that differ from the checkpoint:
//   * conv2/conv3:  [Cout][Cin][3][3] -> [Cout][kh][kw][Cin]   (implicit-GEMM K axis, NHWC activations)
//   * conv_out:     column c*F+f -> f*C+c                       (matches the [t][f][c] conv3 output rows,
//                                                                 so permute+contiguous of audio_encoder.rs:133 vanishes)
//   * q/k/v:        concatenated rows [q|k|v] -> one GEMM / GEMV
//   * gate/up:      interleaved in 16-row blocks [g0..15|u0..15|g16..31|...] -> SiLU*up fused in the epilogue
constexpr uint64_t kNone = ~0ull;
constexpr uint32_t kArenaMagic = 0x41335141u;  // "AQ3A"
constexpr uint32_t kArenaVersion = 2;
constexpr uint64_t kArenaHeaderBytes = 256;
enum ArenaFlags : uint32_t {
  kFlagConvOutBias = 1u << 0,
  kFlagDecQkvBias = 1u << 1,
  kFlagDecOBias = 1u << 2,
  kFlagDecMlpBias = 1u << 3,
  // at least one matrix tensor of the checkpoint was F16 / F32 and has been rounded (nearest-even) to the arena's bf16:
  // the reference widens such checkpoints to f32 instead (weights.rs:74-89), so results may differ beyond bf16-checkpoint parity
  kFlagWeightsRounded = 1u << 4,
};
struct ArenaHeader {
  uint32_t magic, version;
  uint64_t total_bytes;
  uint32_t flags;
  uint32_t pad[59];
};
static_assert(sizeof(ArenaHeader) == kArenaHeaderBytes, "arena header size");

struct EncLayerOff { uint64_t ln1_w, ln1_b, qkv_w, qkv_b, out_w, out_b, ln2_w, ln2_b, fc1_w, fc1_b, fc2_w, fc2_b; };
struct DecLayerOff { uint64_t in_ln, qkv_w, qkv_b, q_norm, k_norm, o_w, o_b, post_ln, gu_w, gu_b, down_w, down_b; };
struct ArenaLayout {
  uint64_t conv1_w, conv1_b, conv2_w, conv2_b, conv3_w, conv3_b, conv_out_w, conv_out_b;
  std::vector<EncLayerOff> enc;
  uint64_t ln_post_w, ln_post_b, proj1_w, proj1_b, proj2_w, proj2_b;
  uint64_t embed, lm_head, final_norm;
  std::vector<DecLayerOff> dec;
  uint64_t total = 0;
  // algorithmic weight bytes one decode token streams (bf16 matrices + vectors actually read)
  double decode_weight_bytes = 0;
};

ArenaLayout plan_arena(const Dims& d);
// Fills dst[0..layout.total) (header included). Throws on missing/ill-shaped tensors.
void pack_arena(const Dims& d, const ArenaLayout& L, const Checkpoint& ck, uint8_t* dst);

// ---- host tables (f64 math, f32 store: mirrors the reference's host-side constant construction) -------
std::vector<float> make_mel_filterbank_T(int n_mels, int n_fft, int sample_rate, int k_pad);  // [k_pad][n_mels], mel.rs:115-187
std::vector<float> make_dft_matrix(int n_fft, int n_cols_pad);      // [n_fft][n_cols_pad], col 2k=Re, 2k+1=Im, periodic Hann folded in
std::vector<float> make_sinusoid_rows(int rows, int dim);           // audio_encoder.rs:283-301
std::vector<int> make_mrope_dim_map(const std::vector<int>& sections, int half, bool interleaved);  // layers.rs:524-562
// cos/sin [n_pos][half] for positions 0..n_pos-1 with all three MRoPE rows equal (inference.rs:259-266)
void make_rope_tables(int n_pos, int head_dim, double theta, std::vector<float>& cos_t, std::vector<float>& sin_t);

}  // namespace q3a
