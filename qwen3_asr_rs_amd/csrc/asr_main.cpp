// `asr <model_path> <audio_file> [language]` -- the reference's CLI (src/main.rs:7-81) on top of libq3asr_hip.so.
// Same argv contract, same usage text on stderr + exit status 1, same stdout ("Language: ...\nText: ...").
// Logging: RUST_LOG=info|debug (default info) prints the reference's progress lines to stderr
// (src/inference.rs:31-103,203-205).  Input: WAV files (the FFmpeg path of src/audio.rs is out of scope).
#include <sys/stat.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/q3asr.h"

static int g_level = 1;  // 0 warn, 1 info, 2 debug
static void logf(int level, const char* fmt, const std::string& a = "") {
  if (level > g_level) return;
  fprintf(stderr, "%s ", level == 2 ? "DEBUG" : " INFO");
  fprintf(stderr, fmt, a.c_str());
  fputc('\n', stderr);
}
static bool exists(const char* p) { struct stat st; return stat(p, &st) == 0; }
static int die(const std::string& msg) { fprintf(stderr, "Error: %s\n", msg.c_str()); return 1; }

int main(int argc, char** argv) {
  if (const char* rl = getenv("RUST_LOG")) {
    if (strstr(rl, "debug") || strstr(rl, "trace")) g_level = 2;
    else if (strstr(rl, "warn") || strstr(rl, "error") || strstr(rl, "off")) g_level = 0;
  }
  if (argc < 3) {  // main.rs:18-35
    fprintf(stderr, "Qwen3 ASR - Automatic Speech Recognition\n\n");
    fprintf(stderr, "Usage: asr <model_path> <audio_file> [language]\n\n");
    fprintf(stderr, "Arguments:\n");
    fprintf(stderr, "  model_path   Path to the Qwen3-ASR model directory\n");
    fprintf(stderr, "  audio_file   Path to the input audio file (WAV; PCM 8/16/24/32-bit or float32)\n");
    fprintf(stderr, "  language     Optional: force language (e.g., chinese, english, japanese)\n\n");
    fprintf(stderr, "The audio file will be automatically converted to mono 16kHz f32 for the model.\n\n");
    fprintf(stderr, "Environment variables:\n");
    fprintf(stderr, "  RUST_LOG     Set logging level (e.g., info, debug, trace)\n");
    return 1;
  }
  const char* model_path = argv[1];
  const char* audio_file = argv[2];
  const char* language = argc > 3 ? argv[3] : nullptr;
  if (!exists(model_path)) return die(std::string("Model directory not found: ") + model_path);  // main.rs:43-45
  if (!exists(audio_file)) return die(std::string("Audio file not found: ") + audio_file);       // main.rs:46-48

  const int n_dev = q3a_device_count();  // main.rs:51-65
  if (n_dev <= 0) return die("No HIP device available (libq3asr_hip has no CPU path)");
  logf(1, "Using HIP device 0 of %s (MI355X / gfx950)", std::to_string(n_dev));
  logf(1, "Loading model from \"%s\"", model_path);
  q3a_opts opts;
  q3a_opts_default(&opts);
  q3a_engine* eng = nullptr;
  if (q3a_engine_create(model_path, 0, &opts, &eng) != 0) return die(std::string("Failed to load model: ") + q3a_last_error(nullptr));
  if (q3a_weights_rounded(eng))
    logf(0, "warning: the checkpoint stores F16/F32 matrices; the HIP backend keeps matrices as bf16 (rounded to nearest-even)");
  logf(1, "Loading tokenizer...");
  q3a_tokenizer* tok = nullptr;
  const std::string tj = std::string(model_path) + "/tokenizer.json";
  if (q3a_tokenizer_create(tj.c_str(), &tok) != 0) {  // tokenizer.rs:19-29
    std::string m = q3a_last_error(nullptr);
    q3a_engine_destroy(eng);
    return die("Failed to load tokenizer: " + m);
  }
  logf(1, "Model loaded successfully");

  logf(1, "Transcribing: %s", audio_file);
  logf(1, "Loading audio from %s", audio_file);
  float* pcm = nullptr;
  int64_t n = 0;
  if (q3a_load_audio(audio_file, 16000, &pcm, &n) != 0) return die(std::string("Transcription failed: ") + q3a_last_error(nullptr));

  std::vector<int32_t> prefix;
  if (language) {  // inference.rs:246-251
    char cap[256];
    q3a_capitalize_first(language, cap, sizeof(cap));
    const std::string text = std::string("language ") + cap;
    int32_t np = 0;
    prefix.resize(64);
    if (q3a_tokenizer_encode(tok, text.c_str(), prefix.data(), (int32_t)prefix.size(), &np) != 0)
      return die(std::string("Transcription failed: ") + q3a_last_error(nullptr));
    prefix.resize((size_t)np);
  }
  const int32_t max_new = 4096;  // inference.rs:153
  std::vector<int32_t> ids((size_t)max_new);
  int32_t len = 0;
  if (q3a_transcribe_batch(eng, pcm, &n, 1, prefix.empty() ? nullptr : prefix.data(), (int32_t)prefix.size(), max_new, 0,
                           ids.data(), max_new, &len) != 0)
    return die(std::string("Transcription failed: ") + q3a_last_error(eng));
  q3a_free(pcm);
  q3a_timings tm;
  q3a_stage_timings(eng, &tm);
  if (g_level >= 1) {
    fprintf(stderr, " INFO Mel spectrogram: %lld frames\n", (long long)q3a_num_frames(n));
    fprintf(stderr, " INFO Audio encoder: %d tokens\n", tm.total_audio_tokens);
    fprintf(stderr, " INFO Generated %d tokens\n", len);
    fprintf(stderr, " INFO timings: mel %.2f ms, encoder %.2f ms, prefill %.2f ms, decode %.2f ms\n", tm.mel_ms, tm.encoder_ms,
            tm.prefill_ms, tm.decode_ms);
  }
  int32_t need = 0;
  q3a_tokenizer_decode(tok, ids.data(), len, 1, nullptr, 0, &need);
  std::string raw((size_t)need + 1, '\0');
  q3a_tokenizer_decode(tok, ids.data(), len, 1, &raw[0], need + 1, &need);
  raw.resize((size_t)need);
  if (g_level >= 2) fprintf(stderr, "DEBUG Raw output: \"%s\"\n", raw.c_str());
  std::vector<char> lang(256), text(raw.size() + 16);
  q3a_parse_asr_output(raw.c_str(), language != nullptr, lang.data(), (int32_t)lang.size(), text.data(), (int32_t)text.size());
  printf("Language: %s\n", lang.data());  // main.rs:77-78
  printf("Text: %s\n", text.data());
  q3a_tokenizer_destroy(tok);
  q3a_engine_destroy(eng);
  return 0;
}
