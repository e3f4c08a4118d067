// Host-side pipeline shell around the engine: WAV ingest, byte-level BPE
// tokenizer (decode + the narrow encode the forced-language prompt needs), output parsing.  Pure C++, no HIP.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace q3a {

// ---- audio (host_audio.cpp; reference: src/audio.rs) ----
void read_wav_mono(const std::string& path, std::vector<float>& samples, int& sample_rate);
void resample_rational(const std::vector<float>& in, int sr_in, int sr_out, std::vector<float>& out);
// rubato 0.16.2 SincFixedIn with the reference fallback's parameters (src/audio.rs:220-245), restated
void resample_rubato_sincfixedin(const std::vector<float>& in, int sr_in, int sr_out, std::vector<float>& out);
std::vector<float> load_audio(const std::string& path, int target_sr);  // audio.rs:7

// ---- tokenizer (host_text.cpp; reference: src/tokenizer.rs over HF tokenizers' tokenizer.json) ----
class BpeTokenizer {
 public:
  explicit BpeTokenizer(const std::string& tokenizer_json_path);      // tokenizer.rs:11-30
  std::string decode(const std::vector<int64_t>& ids, bool skip_special = true) const;  // tokenizer.rs:42-49
  // encode(text, add_special_tokens = false), tokenizer.rs:33-39: added tokens are cut out first (longest match), the rest goes
  // through the normaliser (NFC when tokenizer.json asks for it, as Qwen's does), the Qwen2 pre-tokenisation pattern over
  // Unicode code points (\p{L}, \p{N}, \s from generated tables) and byte-level BPE.
  std::vector<int64_t> encode(const std::string& text) const;
  bool normalizes_nfc() const { return nfc_; }
  size_t vocab_size() const { return id_to_token_.size(); }

 private:
  std::vector<std::string> id_to_token_;                 // byte-level token strings (UTF-8 of the mapped code points)
  std::unordered_map<std::string, int64_t> token_to_id_;
  std::unordered_map<std::string, int> merge_rank_;      // "left right" -> rank
  std::vector<bool> is_special_, is_added_;
  std::vector<int> byte_of_cp_;                          // mapped code point -> byte (-1: none)
  std::vector<std::string> cp_of_byte_;                  // byte -> UTF-8 of its mapped code point
  std::vector<std::vector<int64_t>> added_by_first_;     // added tokens by first byte, longest first
  bool nfc_ = false;                                     // tokenizer.json "normalizer" is (or contains) NFC
};
// Unicode NFC of a UTF-8 string (generated tables: unicode_tables.h)
std::string normalize_nfc(const std::string& utf8);

// ---- output parsing (host_text.cpp; reference: src/inference.rs:276-313) ----
void parse_asr_output(const std::string& raw, bool language_forced, std::string& language, std::string& text);
std::string capitalize_first(const std::string& s);

}  // namespace q3a
