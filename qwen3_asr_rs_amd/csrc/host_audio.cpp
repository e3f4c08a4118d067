// WAV ingest for the CLI / pipeline shell.  Mirrors the reference's hound fallback
// (src/audio.rs:162-245): RIFF PCM s8/s16/s24/s32 or IEEE f32 -> mono by channel mean (audio.rs:192-200) with
// integer scale 1/2^(bits-1) (audio.rs:177) -> target rate.  The reference resamples with FFmpeg swresample or
// rubato's sinc resampler, neither of which is reproducible outside its own code base; this file DEFINES the
// resampler of this backend instead (deterministic, double precision): a Kaiser-windowed-sinc polyphase filter
// for the rational ratio L/M, 16 zero crossings per side at the lower of the two rates, cutoff 0.95 x Nyquist
// (the reference fallback's f_cutoff, audio.rs:225), beta 8.6.  Parity is pinned at the 16 kHz vector.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "host.h"
#include "model.h"

namespace q3a {

namespace {
uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

double bessel_i0(double x) {
  double sum = 1.0, term = 1.0;
  for (int k = 1; k < 64; ++k) {
    term *= (x / (2.0 * k)) * (x / (2.0 * k));
    sum += term;
    if (term < 1e-18 * sum) break;
  }
  return sum;
}
}  // namespace

void read_wav_mono(const std::string& path, std::vector<float>& samples, int& sample_rate) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) fail("Audio file not found: " + path);
  std::vector<uint8_t> buf;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  buf.resize(sz > 0 ? (size_t)sz : 0);
  if (sz > 0 && fread(buf.data(), 1, buf.size(), f) != buf.size()) { fclose(f); fail("short read: " + path); }
  fclose(f);
  if (buf.size() < 12 || memcmp(buf.data(), "RIFF", 4) != 0 || memcmp(buf.data() + 8, "WAVE", 4) != 0)
    fail("not a RIFF/WAVE file: " + path);
  int fmt = 0, channels = 0, bits = 0;
  sample_rate = 0;
  const uint8_t* data = nullptr;
  size_t data_len = 0;
  for (size_t p = 12; p + 8 <= buf.size();) {
    uint32_t len = rd32(&buf[p + 4]);
    const uint8_t* body = &buf[p + 8];
    size_t avail = buf.size() - (p + 8);
    if (memcmp(&buf[p], "fmt ", 4) == 0 && len >= 16 && avail >= 16) {
      fmt = rd16(body); channels = rd16(body + 2); sample_rate = (int)rd32(body + 4); bits = rd16(body + 14);
      if (fmt == 0xFFFE && len >= 26 && avail >= 26) fmt = rd16(body + 24);  // WAVE_FORMAT_EXTENSIBLE sub-format
    } else if (memcmp(&buf[p], "data", 4) == 0) {
      data = body;
      data_len = std::min<size_t>(len, avail);
      break;
    }
    p += 8 + (size_t)len + (len & 1);
  }
  if (!data || channels <= 0 || sample_rate <= 0) fail("WAV without fmt/data chunk: " + path);
  const int bps = bits / 8;
  if (!((fmt == 1 && (bits == 8 || bits == 16 || bits == 24 || bits == 32)) || (fmt == 3 && bits == 32)))
    fail("unsupported WAV encoding (format " + std::to_string(fmt) + ", " + std::to_string(bits) + " bits)");
  const size_t frames = data_len / ((size_t)bps * channels);
  samples.resize(frames);
  const double scale = 1.0 / std::ldexp(1.0, bits - 1);  // audio.rs:177
  for (size_t i = 0; i < frames; ++i) {
    double acc = 0.0;
    for (int c = 0; c < channels; ++c) {
      const uint8_t* s = data + (i * channels + c) * bps;
      double v;
      if (fmt == 3) { float x; memcpy(&x, s, 4); v = x; }
      else if (bits == 8) v = ((int)s[0] - 128) * scale;
      else if (bits == 16) v = (int16_t)rd16(s) * scale;
      else if (bits == 24) { int32_t x = (int32_t)((uint32_t)s[0] << 8 | (uint32_t)s[1] << 16 | (uint32_t)s[2] << 24) >> 8; v = x * scale; }
      else v = (int32_t)rd32(s) * scale;
      acc += v;
    }
    samples[i] = (float)(acc / channels);  // audio.rs:192-200
  }
}

void resample_rational(const std::vector<float>& in, int sr_in, int sr_out, std::vector<float>& out) {
  if (sr_in == sr_out) { out = in; return; }
  const int g = std::gcd(sr_in, sr_out);
  const int64_t L = sr_out / g, M = sr_in / g;  // out[n] sits at input position n*M/L
  const double fc = 0.95 * 0.5 / (double)std::max(L, M);  // cutoff in cycles per sample of the L-times upsampled grid
  const int zc = 16;
  const int64_t half = (int64_t)zc * std::max(L, M);     // filter half-length on the upsampled grid
  const double beta = 8.6, i0b = bessel_i0(beta);
  std::vector<double> h((size_t)(2 * half + 1));
  const double pi = 3.14159265358979323846;
  for (int64_t t = -half; t <= half; ++t) {
    double x = 2.0 * fc * (double)t;
    double sinc = (t == 0) ? 1.0 : std::sin(pi * x) / (pi * x);
    double r = (double)t / (double)half;
    double w = bessel_i0(beta * std::sqrt(std::max(0.0, 1.0 - r * r))) / i0b;
    h[(size_t)(t + half)] = 2.0 * fc * (double)L * sinc * w;  // gain L compensates the zero-stuffing
  }
  const int64_t n_in = (int64_t)in.size();
  const int64_t n_out = (n_in * L + M - 1) / M;
  out.resize((size_t)n_out);
  for (int64_t n = 0; n < n_out; ++n) {
    const int64_t pos = n * M;  // position on the upsampled grid; input k sits at k*L
    int64_t k_lo = (pos - half + L - 1) / L, k_hi = (pos + half) / L;
    if (pos - half < 0) k_lo = 0;
    if (k_hi >= n_in) k_hi = n_in - 1;
    double acc = 0.0;
    for (int64_t k = k_lo; k <= k_hi; ++k) acc += (double)in[(size_t)k] * h[(size_t)(pos - k * L + half)];
    out[(size_t)n] = (float)acc;
  }
}

std::vector<float> load_audio(const std::string& path, int target_sr) {  // audio.rs:7
  std::vector<float> raw, out;
  int sr = 0;
  read_wav_mono(path, raw, sr);
  resample_rational(raw, sr, target_sr, out);
  return out;
}

}  // namespace q3a
