// C ABI of the host-only pipeline-shell helpers (include/q3asr.h, "pipeline shell" section).
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/q3asr.h"
#include "host.h"
#include "model.h"

namespace q3a {
void set_thread_error(const std::string& msg);  // engine.cpp
}
using namespace q3a;

struct q3a_tokenizer {
  BpeTokenizer tok;
  explicit q3a_tokenizer(const std::string& p) : tok(p) {}
};

#define HOST_TRY try {
#define HOST_CATCH                                   \
  }                                                  \
  catch (const std::exception& ex) {                 \
    set_thread_error(ex.what());                     \
    return 1;                                        \
  }                                                  \
  catch (...) {                                      \
    set_thread_error("unknown error");               \
    return 1;                                        \
  }                                                  \
  return 0;

static int32_t give(const std::vector<float>& v, float** out, int64_t* n) {
  float* p = (float*)malloc(std::max<size_t>(v.size(), 1) * sizeof(float));
  if (!p) fail("out of memory");
  if (!v.empty()) memcpy(p, v.data(), v.size() * sizeof(float));
  *out = p;
  *n = (int64_t)v.size();
  return 0;
}
static int32_t put_str(const std::string& s, char* out, int32_t cap) {
  if (!out || cap <= 0) return 0;
  size_t n = std::min<size_t>(s.size(), (size_t)cap - 1);
  memcpy(out, s.data(), n);
  out[n] = 0;
  return (int32_t)n;
}

extern "C" {

int32_t q3a_load_audio(const char* path, int32_t target_sr, float** samples_out, int64_t* n_out) {
  HOST_TRY
  if (!path || !samples_out || !n_out) fail("null argument");
  give(load_audio(path, target_sr), samples_out, n_out);
  HOST_CATCH
}
int32_t q3a_resample(const float* in, int64_t n, int32_t sr_in, int32_t sr_out, float** samples_out, int64_t* n_out) {
  HOST_TRY
  if (!in || n < 0 || sr_in <= 0 || sr_out <= 0) fail("bad argument");
  std::vector<float> v(in, in + n), o;
  resample_rational(v, sr_in, sr_out, o);
  give(o, samples_out, n_out);
  HOST_CATCH
}
int32_t q3a_resample_rubato(const float* in, int64_t n, int32_t sr_in, int32_t sr_out, float** samples_out, int64_t* n_out) {
  HOST_TRY
  if (!in || n < 0 || sr_in <= 0 || sr_out <= 0) fail("bad argument");
  std::vector<float> v(in, in + n), o;
  resample_rubato_sincfixedin(v, sr_in, sr_out, o);
  give(o, samples_out, n_out);
  HOST_CATCH
}
void q3a_free(void* p) { free(p); }

int32_t q3a_tokenizer_create(const char* path, q3a_tokenizer** out) {
  HOST_TRY
  if (!path || !out) fail("null argument");
  *out = new q3a_tokenizer(path);
  HOST_CATCH
}
void q3a_tokenizer_destroy(q3a_tokenizer* t) { delete t; }

int32_t q3a_tokenizer_decode(const q3a_tokenizer* t, const int32_t* ids, int32_t n, int32_t skip_special, char* out,
                             int32_t cap, int32_t* len) {
  HOST_TRY
  if (!t) fail("null tokenizer");
  std::vector<int64_t> v(ids, ids + n);
  std::string s = t->tok.decode(v, skip_special != 0);
  if (len) *len = (int32_t)s.size();
  put_str(s, out, cap);
  HOST_CATCH
}
int32_t q3a_tokenizer_encode(const q3a_tokenizer* t, const char* text, int32_t* ids, int32_t cap, int32_t* n) {
  HOST_TRY
  if (!t || !text) fail("null argument");
  std::vector<int64_t> v = t->tok.encode(text);
  if (n) *n = (int32_t)v.size();
  for (size_t i = 0; i < v.size() && (int32_t)i < cap; ++i) ids[i] = (int32_t)v[i];
  HOST_CATCH
}

int32_t q3a_parse_asr_output(const char* raw, int32_t language_forced, char* language, int32_t language_cap, char* text,
                             int32_t text_cap) {
  HOST_TRY
  std::string l, t;
  parse_asr_output(raw ? raw : "", language_forced != 0, l, t);
  put_str(l, language, language_cap);
  put_str(t, text, text_cap);
  HOST_CATCH
}
int32_t q3a_normalize_nfc(const char* utf8, char* out, int32_t cap, int32_t* len) {
  HOST_TRY
  const std::string r = normalize_nfc(utf8 ? utf8 : "");
  if (len) *len = (int32_t)r.size();
  put_str(r, out, cap);
  HOST_CATCH
}
int32_t q3a_capitalize_first(const char* s, char* out, int32_t cap) {
  HOST_TRY
  put_str(capitalize_first(s ? s : ""), out, cap);
  HOST_CATCH
}

}  // extern "C"
