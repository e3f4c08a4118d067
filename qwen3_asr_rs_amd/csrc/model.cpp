#include "model.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <set>
#include <sstream>

#include "json.h"

namespace q3a {

static std::string read_text(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) fail("cannot open " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

static bool file_exists(const std::string& p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0;
}

// ---------------------------------------------------------------------------------------------------
// config.json (src/config.rs)
// ---------------------------------------------------------------------------------------------------
Dims parse_config_file(const std::string& path) {
  Json root = parse_json(read_text(path));
  const Json& th = root.at("thinker_config");
  Dims d;
  if (const Json* a = th.find("audio_config")) {
    d.enc_d = (int)a->num_or("d_model", d.enc_d);
    d.enc_layers = (int)a->num_or("encoder_layers", d.enc_layers);
    d.enc_heads = (int)a->num_or("encoder_attention_heads", d.enc_heads);
    d.enc_ffn = (int)a->num_or("encoder_ffn_dim", d.enc_ffn);
    d.n_mels = (int)a->num_or("num_mel_bins", d.n_mels);
    d.max_source_positions = (int)a->num_or("max_source_positions", d.max_source_positions);
    d.n_window = (int)a->num_or("n_window", d.n_window);
    d.n_window_infer = (int)a->num_or("n_window_infer", d.n_window_infer);
    d.conv_ch = (int)a->num_or("downsample_hidden_size", d.conv_ch);
    d.enc_out = (int)a->num_or("output_dim", d.enc_out);
  }
  if (const Json* t = th.find("text_config")) {
    d.vocab = (int)t->num_or("vocab_size", d.vocab);
    d.hidden = (int)t->num_or("hidden_size", d.hidden);
    d.inter = (int)t->num_or("intermediate_size", d.inter);
    d.dec_layers = (int)t->num_or("num_hidden_layers", d.dec_layers);
    d.n_q = (int)t->num_or("num_attention_heads", d.n_q);
    d.n_kv = (int)t->num_or("num_key_value_heads", d.n_kv);
    d.head_dim = (int)t->num_or("head_dim", d.head_dim);
    d.rms_eps = (float)t->num_or("rms_norm_eps", 1e-6);
    d.rope_theta = t->num_or("rope_theta", 1e6);
    d.tie_embeddings = t->bool_or("tie_word_embeddings", true);
    const Json* rs = t->find("rope_scaling");
    if (rs && rs->kind == Json::Obj) {  // config.rs:101-136
      if (const Json* ms = rs->find("mrope_section")) {
        if (ms->kind == Json::Arr) {
          d.mrope_section.clear();
          for (auto& v : ms->arr) d.mrope_section.push_back((int)v.num);
        }
      }
      d.mrope_interleaved = rs->bool_or("mrope_interleaved", false) || rs->bool_or("interleaved", false);
    }
  }
  return d;
}

void validate_dims(const Dims& d) {
  auto req = [](bool ok, const std::string& what) {
    if (!ok) fail("unsupported model dimension: " + what);
  };
  req(d.enc_d % d.enc_heads == 0 && d.enc_d / d.enc_heads == 64, "encoder head_dim must be 64");
  req(d.head_dim == 128, "decoder head_dim must be 128");
  req(d.n_q % d.n_kv == 0 && (d.n_q / d.n_kv == 1 || d.n_q / d.n_kv == 2 || d.n_q / d.n_kv == 4), "GQA ratio must be 1, 2 or 4");
  req(d.conv_ch % 32 == 0, "downsample_hidden_size must be a multiple of 32");
  req(d.enc_d % 32 == 0 && d.enc_ffn % 32 == 0 && d.hidden % 32 == 0 && d.inter % 32 == 0, "GEMM K dims must be multiples of 32");
  req(d.inter % 16 == 0, "intermediate_size must be a multiple of 16");
  req(d.enc_d % 4 == 0 && d.hidden % 4 == 0, "hidden sizes must be multiples of 4");
  req(d.hidden % 512 == 0 || d.hidden % 256 == 0, "decoder hidden_size must be a multiple of 256");
  req(d.enc_out == d.hidden, "audio output_dim must equal decoder hidden_size (embedding injection)");
  req(d.chunks_per_window() >= 1, "n_window_infer must cover at least one chunk");
  req(d.chunks_per_window() * d.tokens_per_chunk() <= 128, "attention window must be <= 128 tokens");
  req(d.n_mels == 128, "num_mel_bins must be 128");
  req(d.hidden <= 2048 && d.enc_d <= 2048, "hidden sizes above 2048 need a wider norm kernel");
  req(d.inter <= 16384 && d.q_dim() <= 16384, "decoder width too large for the GEMV staging buffer");
}

// ---------------------------------------------------------------------------------------------------
// safetensors (src/weights.rs)
// ---------------------------------------------------------------------------------------------------
Checkpoint::Checkpoint(const std::string& model_dir) {
  std::string single = model_dir + "/model.safetensors";
  std::string index = model_dir + "/model.safetensors.index.json";
  if (file_exists(single)) {
    load_file(single);
  } else if (file_exists(index)) {
    Json idx = parse_json(read_text(index));
    const Json* wm = idx.find("weight_map");
    if (!wm || wm->kind != Json::Obj) fail("Missing weight_map in index");
    std::set<std::string> shards;  // sorted + unique (weights.rs:41-45)
    for (auto& kv : wm->obj)
      if (kv.second.kind == Json::Str) shards.insert(kv.second.str);
    for (auto& s : shards) load_file(model_dir + "/" + s);
  } else {
    fail("No model weights found in " + model_dir + " (expected model.safetensors or model.safetensors.index.json)");
  }
}

Checkpoint::~Checkpoint() {
  for (auto& m : maps_) munmap(m.addr, m.len);
}

void Checkpoint::load_file(const std::string& path) {
  int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) fail("Failed to read safetensors: " + path);
  struct stat st;
  if (fstat(fd, &st) != 0) { close(fd); fail("Failed to stat safetensors: " + path); }
  size_t len = (size_t)st.st_size;
  if (len < 8) { close(fd); fail("Failed to deserialize safetensors (shorter than its 8-byte length prefix): " + path); }
  void* addr = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (addr == MAP_FAILED) fail("mmap failed: " + path);
  maps_.push_back({addr, len});
  const uint8_t* base = (const uint8_t*)addr;
  uint64_t hlen;
  memcpy(&hlen, base, 8);
  if (hlen > len - 8) fail("Failed to deserialize safetensors (header length): " + path);  // no 8 + hlen wrap-around
  Json hdr = JsonParser((const char*)base + 8, (size_t)hlen).parse();
  const uint8_t* data = base + 8 + hlen;
  uint64_t data_len = len - 8 - hlen;
  for (auto& kv : hdr.obj) {
    if (kv.first == "__metadata__") continue;
    const Json& m = kv.second;
    TensorView tv;
    const std::string& dt = m.at("dtype").str;
    if (dt == "BF16") tv.dtype = StDtype::BF16;
    else if (dt == "F16") tv.dtype = StDtype::F16;
    else if (dt == "F32") tv.dtype = StDtype::F32;
    else if (dt == "I64") tv.dtype = StDtype::I64;
    else fail("Unsupported dtype in safetensors: " + dt);  // weights.rs:114
    for (auto& s : m.at("shape").arr) tv.shape.push_back((int64_t)s.num);
    const Json& off = m.at("data_offsets");
    uint64_t b = (uint64_t)off.arr.at(0).num, e = (uint64_t)off.arr.at(1).num;
    if (e < b || e > data_len) fail("safetensors: bad data_offsets for " + kv.first);
    tv.data = data + b;
    tv.nbytes = e - b;
    size_t esz = (tv.dtype == StDtype::F32) ? 4 : (tv.dtype == StDtype::I64 ? 8 : 2);
    if ((uint64_t)tv.numel() * esz != tv.nbytes) fail("safetensors: size mismatch for " + kv.first);
    tensors_[kv.first] = tv;
  }
}

const TensorView& Checkpoint::get(const std::string& key) const {
  auto it = tensors_.find(key);
  if (it == tensors_.end()) fail("Weight not found: " + key);
  return it->second;
}
const TensorView* Checkpoint::get_opt(const std::string& key) const {
  auto it = tensors_.find(key);
  return it == tensors_.end() ? nullptr : &it->second;
}

// ---------------------------------------------------------------------------------------------------
// element conversion
// ---------------------------------------------------------------------------------------------------
static inline float bf16_to_f32(uint16_t v) {  // weights.rs:134-142
  uint32_t u = (uint32_t)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline uint16_t f32_to_bf16(float f) {  // round-to-nearest-even
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float f16_to_f32(uint16_t h) {  // weights.rs:156-181
  uint32_t sign = (h >> 15) & 1u, exponent = (h >> 10) & 0x1Fu, mantissa = h & 0x3FFu, bits;
  if (exponent == 0) {
    if (mantissa == 0) {
      bits = sign << 31;
    } else {
      int e = 0;
      uint32_t m = mantissa;
      while ((m & 0x400u) == 0) { m <<= 1; --e; }
      m &= 0x3FFu;
      bits = (sign << 31) | ((uint32_t)(127 - 15 + 1 + e) << 23) | (m << 13);
    }
  } else if (exponent == 31) {
    bits = (sign << 31) | (0xFFu << 23) | (mantissa << 13);
  } else {
    bits = (sign << 31) | ((exponent + (127 - 15)) << 23) | (mantissa << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

static inline float elem_f32(const TensorView& t, int64_t i) {
  switch (t.dtype) {
    case StDtype::BF16: return bf16_to_f32(((const uint16_t*)t.data)[i]);
    case StDtype::F16: return f16_to_f32(((const uint16_t*)t.data)[i]);
    case StDtype::F32: { float f; memcpy(&f, t.data + 4 * i, 4); return f; }
    default: fail("integer tensor where a float tensor is expected");
  }
}
static thread_local bool g_rounded_weights = false;  // set when a non-bf16 matrix element passes through elem_bf16 (pack_arena reads it)
static inline uint16_t elem_bf16(const TensorView& t, int64_t i) {
  if (t.dtype == StDtype::BF16) return ((const uint16_t*)t.data)[i];
  g_rounded_weights = true;
  return f32_to_bf16(elem_f32(t, i));
}

static void expect_shape(const TensorView& t, std::initializer_list<int64_t> shape, const std::string& key) {
  std::vector<int64_t> s(shape);
  if (t.shape != s) {
    std::string msg = "shape mismatch for " + key + ": got [";
    for (auto v : t.shape) msg += std::to_string(v) + ",";
    msg += "] expected [";
    for (auto v : s) msg += std::to_string(v) + ",";
    fail(msg + "]");
  }
}

// ---------------------------------------------------------------------------------------------------
// arena plan
// ---------------------------------------------------------------------------------------------------
namespace {
struct Planner {
  uint64_t cur = kArenaHeaderBytes;
  uint64_t take(uint64_t bytes) {
    uint64_t o = cur;
    cur += (bytes + 255) & ~255ull;
    return o;
  }
  uint64_t f32(int64_t n) { return take((uint64_t)n * 4); }
  uint64_t bf16(int64_t n) { return take((uint64_t)n * 2); }
};
}  // namespace

ArenaLayout plan_arena(const Dims& d) {
  Planner p;
  ArenaLayout L;
  const int64_t C = d.conv_ch, D = d.enc_d, Fn = d.enc_ffn;
  L.conv1_w = p.f32(C * 9);
  L.conv1_b = p.f32(C);
  L.conv2_w = p.bf16(C * 9 * C);
  L.conv2_b = p.f32(C);
  L.conv3_w = p.bf16(C * 9 * C);
  L.conv3_b = p.f32(C);
  L.conv_out_w = p.bf16(D * d.conv_out_in());
  L.conv_out_b = p.f32(D);
  L.enc.resize(d.enc_layers);
  for (auto& e : L.enc) {
    e.ln1_w = p.f32(D); e.ln1_b = p.f32(D);
    e.qkv_w = p.bf16(3 * D * D); e.qkv_b = p.f32(3 * D);
    e.out_w = p.bf16(D * D); e.out_b = p.f32(D);
    e.ln2_w = p.f32(D); e.ln2_b = p.f32(D);
    e.fc1_w = p.bf16(Fn * D); e.fc1_b = p.f32(Fn);
    e.fc2_w = p.bf16(D * Fn); e.fc2_b = p.f32(D);
  }
  L.ln_post_w = p.f32(D); L.ln_post_b = p.f32(D);
  L.proj1_w = p.bf16(D * D); L.proj1_b = p.f32(D);
  L.proj2_w = p.bf16((int64_t)d.enc_out * D); L.proj2_b = p.f32(d.enc_out);
  const int64_t H = d.hidden, I = d.inter, V = d.vocab;
  L.embed = p.bf16(V * H);
  L.lm_head = d.tie_embeddings ? L.embed : p.bf16(V * H);
  L.dec.resize(d.dec_layers);
  for (auto& l : L.dec) {
    l.in_ln = p.f32(H);
    l.qkv_w = p.bf16((int64_t)d.qkv_dim() * H); l.qkv_b = p.f32(d.qkv_dim());
    l.q_norm = p.f32(d.head_dim); l.k_norm = p.f32(d.head_dim);
    l.o_w = p.bf16(H * d.q_dim()); l.o_b = p.f32(H);
    l.post_ln = p.f32(H);
    l.gu_w = p.bf16(2 * I * H); l.gu_b = p.f32(2 * I);
    l.down_w = p.bf16(H * I); l.down_b = p.f32(H);
  }
  L.final_norm = p.f32(H);
  L.total = p.cur;
  // per decode token: every decoder matrix once + lm_head once (bf16)
  double per_layer = 2.0 * ((double)d.qkv_dim() * H + (double)H * d.q_dim() + 2.0 * I * H + (double)H * I);
  L.decode_weight_bytes = per_layer * d.dec_layers + 2.0 * V * H;
  return L;
}

// ---------------------------------------------------------------------------------------------------
// arena packing
// ---------------------------------------------------------------------------------------------------
static void put_f32(uint8_t* dst, uint64_t off, const TensorView& t, int64_t n, const std::string& key) {
  if (t.numel() != n) fail("size mismatch for " + key);
  float* o = (float*)(dst + off);
  for (int64_t i = 0; i < n; ++i) o[i] = elem_f32(t, i);
}
static void put_f32_opt(uint8_t* dst, uint64_t off, const TensorView* t, int64_t n, const std::string& key) {
  if (t) put_f32(dst, off, *t, n, key);
  else memset(dst + off, 0, (size_t)n * 4);
}
static void put_bf16_rows(uint8_t* dst, uint64_t off, int64_t row0, const TensorView& t, int64_t rows, int64_t cols,
                          const std::string& key) {
  expect_shape(t, {rows, cols}, key);
  uint16_t* o = (uint16_t*)(dst + off) + row0 * cols;
  if (t.dtype == StDtype::BF16) {
    memcpy(o, t.data, (size_t)(rows * cols * 2));
  } else {
    for (int64_t i = 0; i < rows * cols; ++i) o[i] = elem_bf16(t, i);
  }
}

void pack_arena(const Dims& d, const ArenaLayout& L, const Checkpoint& ck, uint8_t* dst) {
  g_rounded_weights = false;
  memset(dst, 0, kArenaHeaderBytes);
  ArenaHeader hdr;
  memset(&hdr, 0, sizeof(hdr));
  hdr.magic = kArenaMagic;
  hdr.version = kArenaVersion;
  hdr.total_bytes = L.total;
  const std::string at = "thinker.audio_tower";  // inference.rs:47
  const int64_t C = d.conv_ch, D = d.enc_d, Fn = d.enc_ffn;

  // conv1: [C][1][3][3] -> f32 [C][9]   (audio_encoder.rs:37)
  {
    const TensorView& w = ck.get(at + ".conv2d1.weight");
    expect_shape(w, {C, 1, 3, 3}, at + ".conv2d1.weight");
    put_f32(dst, L.conv1_w, w, C * 9, "conv2d1.weight");
    put_f32_opt(dst, L.conv1_b, ck.get_opt(at + ".conv2d1.bias"), C, "conv2d1.bias");
  }
  // conv2/conv3: [Co][Ci][3][3] -> bf16 [Co][kh][kw][Ci]   (audio_encoder.rs:38-39)
  auto pack_conv = [&](const std::string& name, uint64_t w_off, uint64_t b_off) {
    const TensorView& w = ck.get(at + "." + name + ".weight");
    expect_shape(w, {C, C, 3, 3}, at + "." + name + ".weight");
    uint16_t* o = (uint16_t*)(dst + w_off);
    for (int64_t co = 0; co < C; ++co)
      for (int64_t ci = 0; ci < C; ++ci)
        for (int64_t t = 0; t < 9; ++t) o[(co * 9 + t) * C + ci] = elem_bf16(w, (co * C + ci) * 9 + t);
    put_f32_opt(dst, b_off, ck.get_opt(at + "." + name + ".bias"), C, name + ".bias");
  };
  pack_conv("conv2d2", L.conv2_w, L.conv2_b);
  pack_conv("conv2d3", L.conv3_w, L.conv3_b);
  // conv_out: column c*F+f -> f*C+c   (audio_encoder.rs:40,132-134)
  {
    const int64_t F3 = d.freq3(), K = d.conv_out_in();
    const TensorView& w = ck.get(at + ".conv_out.weight");
    expect_shape(w, {D, K}, at + ".conv_out.weight");
    uint16_t* o = (uint16_t*)(dst + L.conv_out_w);
    for (int64_t r = 0; r < D; ++r)
      for (int64_t c = 0; c < C; ++c)
        for (int64_t f = 0; f < F3; ++f) o[r * K + f * C + c] = elem_bf16(w, r * K + c * F3 + f);
    const TensorView* b = ck.get_opt(at + ".conv_out.bias");
    if (b) hdr.flags |= kFlagConvOutBias;
    put_f32_opt(dst, L.conv_out_b, b, D, "conv_out.bias");
  }
  for (int i = 0; i < d.enc_layers; ++i) {  // layers.rs:143-146,187-188,217-226
    const EncLayerOff& e = L.enc[i];
    std::string p = at + ".layers." + std::to_string(i);
    put_f32(dst, e.ln1_w, ck.get(p + ".self_attn_layer_norm.weight"), D, p + ".self_attn_layer_norm.weight");
    put_f32(dst, e.ln1_b, ck.get(p + ".self_attn_layer_norm.bias"), D, p + ".self_attn_layer_norm.bias");
    const char* names[3] = {"q_proj", "k_proj", "v_proj"};
    for (int j = 0; j < 3; ++j) {
      std::string k = p + ".self_attn." + names[j];
      put_bf16_rows(dst, e.qkv_w, (int64_t)j * D, ck.get(k + ".weight"), D, D, k + ".weight");
      put_f32_opt(dst, e.qkv_b + (uint64_t)j * D * 4, ck.get_opt(k + ".bias"), D, k + ".bias");
    }
    put_bf16_rows(dst, e.out_w, 0, ck.get(p + ".self_attn.out_proj.weight"), D, D, p + ".self_attn.out_proj.weight");
    put_f32_opt(dst, e.out_b, ck.get_opt(p + ".self_attn.out_proj.bias"), D, "out_proj.bias");
    put_f32(dst, e.ln2_w, ck.get(p + ".final_layer_norm.weight"), D, p + ".final_layer_norm.weight");
    put_f32(dst, e.ln2_b, ck.get(p + ".final_layer_norm.bias"), D, p + ".final_layer_norm.bias");
    put_bf16_rows(dst, e.fc1_w, 0, ck.get(p + ".fc1.weight"), Fn, D, p + ".fc1.weight");
    put_f32_opt(dst, e.fc1_b, ck.get_opt(p + ".fc1.bias"), Fn, "fc1.bias");
    put_bf16_rows(dst, e.fc2_w, 0, ck.get(p + ".fc2.weight"), D, Fn, p + ".fc2.weight");
    put_f32_opt(dst, e.fc2_b, ck.get_opt(p + ".fc2.bias"), D, "fc2.bias");
  }
  put_f32(dst, L.ln_post_w, ck.get(at + ".ln_post.weight"), D, at + ".ln_post.weight");  // audio_encoder.rs:53-55
  put_f32(dst, L.ln_post_b, ck.get(at + ".ln_post.bias"), D, at + ".ln_post.bias");
  put_bf16_rows(dst, L.proj1_w, 0, ck.get(at + ".proj1.weight"), D, D, at + ".proj1.weight");
  put_f32_opt(dst, L.proj1_b, ck.get_opt(at + ".proj1.bias"), D, "proj1.bias");
  put_bf16_rows(dst, L.proj2_w, 0, ck.get(at + ".proj2.weight"), d.enc_out, D, at + ".proj2.weight");
  put_f32_opt(dst, L.proj2_b, ck.get_opt(at + ".proj2.bias"), d.enc_out, "proj2.bias");

  const std::string tm = "thinker.model";  // inference.rs:57
  const int64_t H = d.hidden, I = d.inter, V = d.vocab;
  put_bf16_rows(dst, L.embed, 0, ck.get(tm + ".embed_tokens.weight"), V, H, tm + ".embed_tokens.weight");
  if (!d.tie_embeddings)  // text_decoder.rs:71-79
    put_bf16_rows(dst, L.lm_head, 0, ck.get("thinker.lm_head.weight"), V, H, "thinker.lm_head.weight");
  for (int i = 0; i < d.dec_layers; ++i) {  // layers.rs:271-276,390-392,424-438
    const DecLayerOff& l = L.dec[i];
    std::string p = tm + ".layers." + std::to_string(i);
    put_f32(dst, l.in_ln, ck.get(p + ".input_layernorm.weight"), H, p + ".input_layernorm.weight");
    put_f32(dst, l.post_ln, ck.get(p + ".post_attention_layernorm.weight"), H, p + ".post_attention_layernorm.weight");
    const char* names[3] = {"q_proj", "k_proj", "v_proj"};
    const int64_t rows[3] = {d.q_dim(), d.kv_dim(), d.kv_dim()};
    int64_t r0 = 0;
    for (int j = 0; j < 3; ++j) {
      std::string k = p + ".self_attn." + names[j];
      put_bf16_rows(dst, l.qkv_w, r0, ck.get(k + ".weight"), rows[j], H, k + ".weight");
      const TensorView* b = ck.get_opt(k + ".bias");
      if (b) hdr.flags |= kFlagDecQkvBias;
      put_f32_opt(dst, l.qkv_b + (uint64_t)r0 * 4, b, rows[j], k + ".bias");
      r0 += rows[j];
    }
    put_f32(dst, l.q_norm, ck.get(p + ".self_attn.q_norm.weight"), d.head_dim, p + ".self_attn.q_norm.weight");
    put_f32(dst, l.k_norm, ck.get(p + ".self_attn.k_norm.weight"), d.head_dim, p + ".self_attn.k_norm.weight");
    put_bf16_rows(dst, l.o_w, 0, ck.get(p + ".self_attn.o_proj.weight"), H, d.q_dim(), p + ".self_attn.o_proj.weight");
    {
      const TensorView* b = ck.get_opt(p + ".self_attn.o_proj.bias");
      if (b) hdr.flags |= kFlagDecOBias;
      put_f32_opt(dst, l.o_b, b, H, "o_proj.bias");
    }
    // gate/up interleave in 16-row blocks
    {
      const TensorView& g = ck.get(p + ".mlp.gate_proj.weight");
      const TensorView& u = ck.get(p + ".mlp.up_proj.weight");
      expect_shape(g, {I, H}, p + ".mlp.gate_proj.weight");
      expect_shape(u, {I, H}, p + ".mlp.up_proj.weight");
      uint16_t* o = (uint16_t*)(dst + l.gu_w);
      const TensorView* gb = ck.get_opt(p + ".mlp.gate_proj.bias");
      const TensorView* ub = ck.get_opt(p + ".mlp.up_proj.bias");
      if (gb || ub) hdr.flags |= kFlagDecMlpBias;
      float* ob = (float*)(dst + l.gu_b);
      for (int64_t j = 0; j < I; ++j) {
        int64_t rg = (j / 16) * 32 + (j % 16), ru = rg + 16;
        if (g.dtype == StDtype::BF16) memcpy(o + rg * H, (const uint16_t*)g.data + j * H, (size_t)H * 2);
        else for (int64_t c = 0; c < H; ++c) o[rg * H + c] = elem_bf16(g, j * H + c);
        if (u.dtype == StDtype::BF16) memcpy(o + ru * H, (const uint16_t*)u.data + j * H, (size_t)H * 2);
        else for (int64_t c = 0; c < H; ++c) o[ru * H + c] = elem_bf16(u, j * H + c);
        ob[rg] = gb ? elem_f32(*gb, j) : 0.f;
        ob[ru] = ub ? elem_f32(*ub, j) : 0.f;
      }
    }
    put_bf16_rows(dst, l.down_w, 0, ck.get(p + ".mlp.down_proj.weight"), H, I, p + ".mlp.down_proj.weight");
    {
      const TensorView* b = ck.get_opt(p + ".mlp.down_proj.bias");
      if (b) hdr.flags |= kFlagDecMlpBias;
      put_f32_opt(dst, l.down_b, b, H, "down_proj.bias");
    }
  }
  put_f32(dst, L.final_norm, ck.get(tm + ".norm.weight"), H, tm + ".norm.weight");  // text_decoder.rs:69
  if (g_rounded_weights) hdr.flags |= kFlagWeightsRounded;
  memcpy(dst, &hdr, sizeof(hdr));
}

// ---------------------------------------------------------------------------------------------------
// host tables
// ---------------------------------------------------------------------------------------------------
std::vector<float> make_mel_filterbank_T(int n_mels, int n_fft, int sample_rate, int k_pad) {
  // mel.rs:115-187 -- f64 construction, f32 store before the Slaney normalisation multiply (in f32).
  const int n_freqs = n_fft / 2 + 1;
  const double sr = (double)sample_rate, fmin = 0.0, fmax = sr / 2.0;
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = (min_log_hz - 0.0) / f_sp;
  const double logstep = std::log(6.4) / 27.0;
  auto hz_to_mel = [&](double f) { return f < min_log_hz ? f / f_sp : min_log_mel + std::log(f / min_log_hz) / logstep; };
  auto mel_to_hz = [&](double m) { return m < min_log_mel ? f_sp * m : min_log_hz * std::exp(logstep * (m - min_log_mel)); };
  const double mel_min = hz_to_mel(fmin), mel_max = hz_to_mel(fmax);
  std::vector<double> ff(n_mels + 2), af(n_freqs), fd(n_mels + 1);
  for (int i = 0; i < n_mels + 2; ++i) ff[i] = mel_to_hz(mel_min + (mel_max - mel_min) * (double)i / (double)(n_mels + 1));
  for (int j = 0; j < n_freqs; ++j) af[j] = (double)j * sr / (double)n_fft;
  for (int i = 0; i < n_mels + 1; ++i) fd[i] = ff[i + 1] - ff[i];
  std::vector<float> out((size_t)k_pad * n_mels, 0.f);
  for (int j = 0; j < n_freqs; ++j)
    for (int i = 0; i < n_mels; ++i) {
      double down = (af[j] - ff[i]) / fd[i], up = (ff[i + 2] - af[j]) / fd[i + 1];
      float v = (float)std::max(std::min(down, up), 0.0);
      float enorm = (float)(2.0 / (ff[i + 2] - ff[i]));
      out[(size_t)j * n_mels + i] = v * enorm;
    }
  return out;
}

std::vector<float> make_dft_matrix(int n_fft, int n_cols_pad) {
  // periodic Hann (tensor.rs:215-219 -> at::hann_window default periodic=true) folded into the DFT basis
  const int n_freqs = n_fft / 2 + 1;
  std::vector<float> out((size_t)n_fft * n_cols_pad, 0.f);
  const double two_pi = 6.283185307179586476925286766559;
  for (int n = 0; n < n_fft; ++n) {
    float w = (float)(0.5 - 0.5 * std::cos(two_pi * (double)n / (double)n_fft));  // f32 window, as torch computes it
    for (int k = 0; k < n_freqs; ++k) {
      int kn = (int)(((long long)k * n) % n_fft);  // exact argument reduction
      double ang = two_pi * (double)kn / (double)n_fft;
      out[(size_t)n * n_cols_pad + 2 * k] = (float)((double)w * std::cos(ang));
      out[(size_t)n * n_cols_pad + 2 * k + 1] = (float)(-(double)w * std::sin(ang));
    }
  }
  return out;
}

std::vector<float> make_sinusoid_rows(int rows, int dim) {  // audio_encoder.rs:283-301
  const int half = dim / 2;
  const double inc = std::log(10000.0) / (double)(half - 1);
  std::vector<float> out((size_t)rows * dim);
  for (int pos = 0; pos < rows; ++pos)
    for (int i = 0; i < half; ++i) {
      double inv = std::exp(-(double)i * inc), ang = (double)pos * inv;
      out[(size_t)pos * dim + i] = (float)std::sin(ang);
      out[(size_t)pos * dim + half + i] = (float)std::cos(ang);
    }
  return out;
}

std::vector<int> make_mrope_dim_map(const std::vector<int>& sections, int half, bool interleaved) {
  std::vector<int> map;
  if (!interleaved) {  // layers.rs:524-538
    for (size_t dim = 0; dim < sections.size(); ++dim)
      for (int k = 0; k < sections[dim]; ++k) {
        if ((int)map.size() >= half) break;
        map.push_back((int)dim);
      }
    while ((int)map.size() < half) map.push_back((int)sections.size() - 1);
  } else {  // layers.rs:540-562
    std::vector<int> counts(sections.size(), 0);
    while ((int)map.size() < half) {
      size_t prev = map.size();
      for (size_t dim = 0; dim < sections.size(); ++dim) {
        if ((int)map.size() >= half) break;
        if (counts[dim] < sections[dim]) { map.push_back((int)dim); counts[dim]++; }
      }
      if (map.size() == prev) break;
    }
  }
  return map;
}

void make_rope_tables(int n_pos, int head_dim, double theta, std::vector<float>& cos_t, std::vector<float>& sin_t) {
  // layers.rs:471-522 with the three position rows equal (inference.rs:259-266,172-176): the dim map
  // selects among identical rows, so angle = pos * theta^(-2j/head_dim) for every section layout.
  const int half = head_dim / 2;
  cos_t.resize((size_t)n_pos * half);
  sin_t.resize((size_t)n_pos * half);
  std::vector<double> inv(half);
  for (int j = 0; j < half; ++j) inv[j] = 1.0 / std::pow(theta, 2.0 * (double)j / (double)head_dim);
  for (int t = 0; t < n_pos; ++t)
    for (int j = 0; j < half; ++j) {
      double ang = (double)t * inv[j];
      cos_t[(size_t)t * half + j] = (float)std::cos(ang);
      sin_t[(size_t)t * half + j] = (float)std::sin(ang);
    }
}

}  // namespace q3a
