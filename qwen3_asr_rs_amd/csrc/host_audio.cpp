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
#include <cstdlib>
#include <cstring>
#include <algorithm>
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

// The reference's WAV fallback resampler (src/audio.rs:220-245): rubato 0.16.2 `SincFixedIn::<f32>::new(ratio, 2.0,
// {sinc_len 256, f_cutoff 0.95, Linear, oversampling_factor 256, BlackmanHarris2}, chunk = whole clip, 1 channel)` and ONE
// `process` call.  rubato is a third-party crate absent from /root/reference (Cargo.lock: rubato 0.16.2); this restates its
// published algorithm (sinc.rs `make_sincs`, windows.rs `blackman_harris` squared, asynchro_sinc.rs `process_into_buffer`
// with `InterpolationType::Linear`) in double precision -- it cannot be run against the crate here (no Rust toolchain), so it
// is "same filter family and parameters", not a bit-level pin:
//   * table: y[x] = w[x] * sinc((x - T/2) * fc / F), T = 256 * F, F = 256, fc = 0.95 * min(1, ratio), w = squared 4-term
//     Blackman-Harris over T points; sincs[F-1-n][p] = y[F*p + n] / (sum(y) / F);
//   * the clip is preceded by 2*256 zeros; idx starts at -128 and advances by 1/ratio BEFORE each output; an output is the
//     linear interpolation (fraction of idx*F) of the two nearest sub-filters applied to 256 input samples starting at
//     floor(idx); the loop ends at idx >= n_in - (sinc_len + 1) - ceil(1/ratio) (rubato's end_idx: the increment happens
//     after the test, so without the ceil term the last window of a >= 2x decimation would start past the buffer).  Output n
//     therefore sits at input time (n+1)/ratio and the last ~129 input samples produce nothing (rubato would deliver them
//     with the next chunk, which the reference never feeds).
void resample_rubato_sincfixedin(const std::vector<float>& in, int sr_in, int sr_out, std::vector<float>& out) {
  if (sr_in == sr_out) { out = in; return; }
  const double ratio = (double)sr_out / (double)sr_in;
  const int sinc_len = 256, F = 256;
  const double fc = (ratio >= 1.0) ? 0.95 : 0.95 * (double)(float)ratio;  // rubato scales f_cutoff (an f32) when downsampling
  const int T = sinc_len * F;
  const double pi = 3.14159265358979323846;
  std::vector<double> y((size_t)T);
  double sum = 0.0;
  for (int x = 0; x < T; ++x) {
    const double ph = 2.0 * pi * (double)x / (double)T;
    double w = 0.35875 - 0.48829 * std::cos(ph) + 0.14128 * std::cos(2.0 * ph) - 0.01168 * std::cos(3.0 * ph);
    w *= w;  // BlackmanHarris2
    const double v = ((double)x - (double)(T / 2)) * fc / (double)F;
    const double sc = (v == 0.0) ? 1.0 : std::sin(pi * v) / (pi * v);
    y[(size_t)x] = w * sc;
    sum += y[(size_t)x];
  }
  sum /= (double)F;
  std::vector<float> sincs((size_t)T);  // [sub-filter][tap], stored as f32 like SincFixedIn::<f32>
  for (int p = 0; p < sinc_len; ++p)
    for (int n = 0; n < F; ++n) sincs[(size_t)(F - n - 1) * sinc_len + p] = (float)(y[(size_t)F * p + n] / sum);
  const int64_t n_in = (int64_t)in.size();
  std::vector<float> buf((size_t)(n_in + 2 * sinc_len), 0.f);
  std::copy(in.begin(), in.end(), buf.begin() + 2 * sinc_len);
  const double t_ratio = 1.0 / ratio, end_idx = (double)(n_in - (sinc_len + 1)) - std::ceil(t_ratio);
  double idx = -(double)sinc_len / 2.0;
  out.clear();
  out.reserve((size_t)((double)n_in * ratio) + 8);
  auto dot = [&](int64_t index, int64_t sub) {
    const float* w = &sincs[(size_t)sub * sinc_len];
    const float* x = &buf[(size_t)(index + 2 * sinc_len)];
    double acc = 0.0;
    for (int k = 0; k < sinc_len; ++k) acc += (double)x[k] * (double)w[k];
    return acc;
  };
  while (idx < end_idx) {
    idx += t_ratio;
    const double fl = std::floor(idx);
    int64_t i0 = (int64_t)fl, s0 = (int64_t)std::floor((idx - fl) * (double)F);
    int64_t i1 = i0, s1 = s0 + 1;
    if (s1 >= F) { s1 -= F; ++i1; }
    const double frac = idx * (double)F - std::floor(idx * (double)F);
    const double p0 = dot(i0, s0), p1 = dot(i1, s1);
    out.push_back((float)((1.0 - frac) * p0 + frac * p1));
  }
}

// $Q3A_RESAMPLER = "rubato": use the reference fallback's resampler (above) instead of this backend's polyphase filter
std::vector<float> load_audio(const std::string& path, int target_sr) {  // audio.rs:7
  std::vector<float> raw, out;
  int sr = 0;
  read_wav_mono(path, raw, sr);
  const char* e = getenv("Q3A_RESAMPLER");
  if (e && std::string(e) == "rubato") resample_rubato_sincfixedin(raw, sr, target_sr, out);
  else resample_rational(raw, sr, target_sr, out);
  return out;
}

}  // namespace q3a
