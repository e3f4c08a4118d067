// Byte-level BPE tokenizer over HuggingFace's tokenizer.json and the output parsing of the pipeline shell.
// Replaces src/tokenizer.rs (which wraps the `tokenizers` crate, a dependency outside the reference tree) for
// the two uses the path has: decode(generated ids, skip_special_tokens = true) (tokenizer.rs:42-49,
// inference.rs:204) and encode("language {Lang}") for a forced language (tokenizer.rs:33-39,
// inference.rs:249-250); and src/inference.rs:276-313 (parse_asr_output, capitalize_first).
#include <algorithm>
#include <climits>
#include <fstream>
#include <sstream>

#include "host.h"
#include "unicode_tables.h"
#include "json.h"
#include "model.h"

namespace q3a {

namespace {

std::string utf8_of(uint32_t cp) {
  std::string s;
  if (cp < 0x80) s += (char)cp;
  else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
  else if (cp < 0x10000) { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
  else { s += (char)(0xF0 | (cp >> 18)); s += (char)(0x80 | ((cp >> 12) & 0x3F)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
  return s;
}
// decode one UTF-8 code point starting at s[i]; advances i (invalid bytes come back as themselves)
uint32_t next_cp(const std::string& s, size_t& i) {
  unsigned char c = (unsigned char)s[i];
  int n = c < 0x80 ? 0 : (c >> 5) == 6 ? 1 : (c >> 4) == 14 ? 2 : (c >> 3) == 30 ? 3 : -1;
  if (n <= 0 || i + (size_t)n >= s.size()) { ++i; return c; }  // ASCII, bad lead byte or truncated sequence
  uint32_t cp = c & (0xFF >> (n + 2));
  for (int k = 1; k <= n; ++k) cp = (cp << 6) | ((unsigned char)s[i + k] & 0x3F);
  i += (size_t)n + 1;
  return cp;
}

std::string read_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) fail("Failed to load tokenizer: cannot open " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

// Unicode classes of the Qwen2 pre-tokenisation pattern (tables generated from unicodedata, unicode_tables.h)
bool is_letter(uint32_t c) { return c < 0x80 ? ((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z')) : cp_in(kUnicodeLetter, kUnicodeLetter_n, c); }
bool is_digit(uint32_t c) { return c < 0x80 ? (c >= '0' && c <= '9') : cp_in(kUnicodeNumber, kUnicodeNumber_n, c); }
bool is_space(uint32_t c) { return cp_in(kUnicodeSpace, kUnicodeSpace_n, c); }
// simple case folding of the seven letters the contraction alternative can see (U+017F LONG S folds to s, U+212A KELVIN to k)
uint32_t fold(uint32_t c) { return (c >= 'A' && c <= 'Z') ? c + 32 : (c == 0x17F ? 's' : c); }

// Qwen2 pre-tokenisation pattern over code points (leftmost alternative first, greedy quantifiers, as the regex engine does):
// (?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+
// t = code points of the text; returns the index one past the piece that starts at i
size_t match_piece(const std::vector<uint32_t>& t, size_t i) {
  const size_t n = t.size();
  if (t[i] == '\'' && i + 1 < n) {
    const uint32_t a = fold(t[i + 1]);
    // alternation order 's 't 're 've 'm 'll 'd: the two-letter forms can only match when the one-letter ones did not
    if (a == 's' || a == 't') return i + 2;
    if (i + 2 < n) {
      const uint32_t b = fold(t[i + 2]);
      if ((a == 'r' && b == 'e') || (a == 'v' && b == 'e')) return i + 3;
    }
    if (a == 'm') return i + 2;
    if (i + 2 < n && a == 'l' && fold(t[i + 2]) == 'l') return i + 3;
    if (a == 'd') return i + 2;
  }
  {  // [^\r\n L N]? L+
    size_t j = i;
    const uint32_t c = t[j];
    if (!(c == '\r' || c == '\n' || is_letter(c) || is_digit(c)) && j + 1 < n && is_letter(t[j + 1])) ++j;
    if (is_letter(t[j])) {
      while (j < n && is_letter(t[j])) ++j;
      return j;
    }
  }
  if (is_digit(t[i])) return i + 1;
  {  // " ?[^\s L N]+[\r\n]*"
    size_t j = i;
    auto other = [&](size_t k) { const uint32_t c = t[k]; return !is_space(c) && !is_letter(c) && !is_digit(c); };
    if (t[j] == ' ' && j + 1 < n && other(j + 1)) ++j;
    if (j < n && other(j)) {
      while (j < n && other(j)) ++j;
      while (j < n && (t[j] == '\r' || t[j] == '\n')) ++j;
      return j;
    }
  }
  if (is_space(t[i])) {
    size_t j = i;
    while (j < n && is_space(t[j])) ++j;  // maximal whitespace run [i, j)
    size_t last_nl = std::string::npos;
    for (size_t k = i; k < j; ++k)
      if (t[k] == '\r' || t[k] == '\n') last_nl = k;
    if (last_nl != std::string::npos) return last_nl + 1;      // \s*[\r\n]+
    if (j == n) return j;                                       // \s+(?!\S) at end of text
    if (j - i > 1) return j - 1;                                // \s+(?!\S): leave one space for the next piece
    return j;                                                   // \s+
  }
  return i + 1;
}

// ---- NFC (Unicode Standard Annex #15) over code points; tables in unicode_tables.h ----
constexpr uint32_t SBase = 0xAC00, LBase = 0x1100, VBase = 0x1161, TBase = 0x11A7, LCount = 19, VCount = 21, TCount = 28,
                   NCount = VCount * TCount, SCount = LCount * NCount;
uint8_t ccc_of(uint32_t cp) {
  int lo = 0, hi = kNfcCcc_n - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    if (cp < kNfcCcc[mid].lo) hi = mid - 1;
    else if (cp > kNfcCcc[mid].hi) lo = mid + 1;
    else return kNfcCcc[mid].ccc;
  }
  return 0;
}
void decompose_into(uint32_t cp, std::vector<uint32_t>& out) {  // full canonical decomposition (recursive)
  if (cp >= SBase && cp < SBase + SCount) {  // Hangul syllable -> L V (T)
    const uint32_t si = cp - SBase;
    out.push_back(LBase + si / NCount);
    out.push_back(VBase + (si % NCount) / TCount);
    if (si % TCount) out.push_back(TBase + si % TCount);
    return;
  }
  int lo = 0, hi = kNfcDecomp_n - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    if (cp < kNfcDecomp[mid].cp) hi = mid - 1;
    else if (cp > kNfcDecomp[mid].cp) lo = mid + 1;
    else {
      decompose_into(kNfcDecomp[mid].a, out);
      if (kNfcDecomp[mid].b) decompose_into(kNfcDecomp[mid].b, out);
      return;
    }
  }
  out.push_back(cp);
}
uint32_t compose_pair(uint32_t a, uint32_t b) {  // 0: no primary composite
  if (a >= LBase && a < LBase + LCount && b >= VBase && b < VBase + VCount) return SBase + ((a - LBase) * VCount + (b - VBase)) * TCount;
  if (a >= SBase && a < SBase + SCount && (a - SBase) % TCount == 0 && b > TBase && b < TBase + TCount) return a + (b - TBase);
  int lo = 0, hi = kNfcComp_n - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const CpComp& e = kNfcComp[mid];
    if (a < e.a || (a == e.a && b < e.b)) hi = mid - 1;
    else if (a > e.a || (a == e.a && b > e.b)) lo = mid + 1;
    else return e.cp;
  }
  return 0;
}

}  // namespace

// NFC of a UTF-8 string (the normaliser of Qwen's tokenizer.json).  Text below U+0300 holds no combining mark and no
// decomposable-but-uncomposed sequence, so it is returned as is (the reference's only use, "language English", is ASCII).
std::string normalize_nfc(const std::string& in) {
  std::vector<uint32_t> cps;
  bool low = true;
  for (size_t i = 0; i < in.size();) { const uint32_t c = next_cp(in, i); cps.push_back(c); low = low && c < 0x300; }
  if (low) return in;
  std::vector<uint32_t> d;
  for (uint32_t c : cps) decompose_into(c, d);
  // canonical ordering: stable sort of every run of non-starters by combining class
  for (size_t i = 0; i < d.size();) {
    if (ccc_of(d[i]) == 0) { ++i; continue; }
    size_t j = i;
    while (j < d.size() && ccc_of(d[j]) != 0) ++j;
    std::stable_sort(d.begin() + (long)i, d.begin() + (long)j, [](uint32_t x, uint32_t y) { return ccc_of(x) < ccc_of(y); });
    i = j;
  }
  // canonical composition
  std::vector<uint32_t> o;
  long starter = -1;      // index in o of the last starter
  int last_ccc = -1;      // combining class of the last character kept after that starter (-1: none yet)
  for (uint32_t c : d) {
    const int k = ccc_of(c);
    if (starter >= 0 && (last_ccc < k || last_ccc == -1)) {  // not blocked (adjacent, or every class in between is lower)
      const uint32_t comp = (last_ccc == -1 || k != 0) ? compose_pair(o[(size_t)starter], c) : 0;
      if (comp) { o[(size_t)starter] = comp; continue; }
    }
    if (k == 0) { starter = (long)o.size(); last_ccc = -1; } else last_ccc = k;
    o.push_back(c);
  }
  std::string out;
  for (uint32_t c : o) out += utf8_of(c);
  return out;
}

BpeTokenizer::BpeTokenizer(const std::string& path) {
  Json root = parse_json(read_file(path));
  const Json& model = root.at("model");
  const Json& vocab = model.at("vocab");
  if (vocab.kind != Json::Obj) fail("tokenizer.json: model.vocab missing");
  size_t max_id = 0;
  for (auto& kv : vocab.obj) max_id = std::max(max_id, (size_t)kv.second.num);
  if (const Json* added = root.find("added_tokens"))
    for (auto& t : added->arr) max_id = std::max(max_id, (size_t)t.at("id").num);
  id_to_token_.assign(max_id + 1, std::string());
  is_special_.assign(max_id + 1, false);
  is_added_.assign(max_id + 1, false);
  for (auto& kv : vocab.obj) {
    id_to_token_[(size_t)kv.second.num] = kv.first;
    token_to_id_[kv.first] = (int64_t)kv.second.num;
  }
  if (const Json* added = root.find("added_tokens"))
    for (auto& t : added->arr) {
      size_t id = (size_t)t.at("id").num;
      id_to_token_[id] = t.at("content").str;
      token_to_id_[t.at("content").str] = (int64_t)id;
      is_added_[id] = true;
      is_special_[id] = t.bool_or("special", false);
    }
  if (const Json* merges = model.find("merges")) {
    int rank = 0;
    for (auto& m : merges->arr) {
      if (m.kind == Json::Str) merge_rank_[m.str] = rank++;                                   // "left right"
      else if (m.kind == Json::Arr && m.arr.size() == 2) merge_rank_[m.arr[0].str + " " + m.arr[1].str] = rank++;
    }
  }
  // added tokens bucketed by their first byte, longest first: the cut in encode() looks at one short bucket per position
  added_by_first_.assign(256, {});
  for (size_t id = 0; id < id_to_token_.size(); ++id)
    if (is_added_[id] && !id_to_token_[id].empty()) added_by_first_[(unsigned char)id_to_token_[id][0]].push_back((int64_t)id);
  for (auto& b : added_by_first_)
    std::stable_sort(b.begin(), b.end(), [&](int64_t x, int64_t y) { return id_to_token_[(size_t)x].size() > id_to_token_[(size_t)y].size(); });
  // "normalizer": {"type": "NFC"} (Qwen) or a Sequence that contains one
  if (const Json* nz = root.find("normalizer")) {
    auto is_nfc = [](const Json& j) { const Json* t = j.kind == Json::Obj ? j.find("type") : nullptr; return t && t->kind == Json::Str && t->str == "NFC"; };
    nfc_ = is_nfc(*nz);
    if (nz->kind == Json::Obj)
      if (const Json* seq = nz->find("normalizers"))
        for (auto& e : seq->arr) nfc_ = nfc_ || is_nfc(e);
  }
  // GPT-2 byte <-> unicode table: printable bytes map to themselves, the rest to U+0100 + n
  byte_of_cp_.assign(512, -1);
  cp_of_byte_.resize(256);
  int extra = 0;
  for (int b = 0; b < 256; ++b) {
    bool keep = (b >= 33 && b <= 126) || (b >= 161 && b <= 172) || (b >= 174 && b <= 255);
    uint32_t cp = keep ? (uint32_t)b : (uint32_t)(256 + extra++);
    byte_of_cp_[cp] = b;
    cp_of_byte_[b] = utf8_of(cp);
  }
}

std::string BpeTokenizer::decode(const std::vector<int64_t>& ids, bool skip_special) const {
  std::string bytes;
  for (int64_t id : ids) {
    if (id < 0 || (size_t)id >= id_to_token_.size()) continue;
    if (skip_special && is_special_[(size_t)id]) continue;
    const std::string& tok = id_to_token_[(size_t)id];
    // ByteLevel decoder: every char -> byte; a token containing an unmapped char is passed through as UTF-8
    std::string conv;
    bool ok = true;
    for (size_t i = 0; i < tok.size();) {
      uint32_t cp = next_cp(tok, i);
      if (cp < byte_of_cp_.size() && byte_of_cp_[cp] >= 0) conv += (char)byte_of_cp_[cp];
      else { ok = false; break; }
    }
    bytes += ok ? conv : tok;
  }
  // String::from_utf8_lossy: replace malformed sequences by U+FFFD
  std::string out;
  for (size_t i = 0; i < bytes.size();) {
    unsigned char c = (unsigned char)bytes[i];
    int n = c < 0x80 ? 0 : (c >= 0xC2 && c <= 0xDF) ? 1 : (c >= 0xE0 && c <= 0xEF) ? 2 : (c >= 0xF0 && c <= 0xF4) ? 3 : -1;
    bool good = n >= 0 && i + (size_t)n < bytes.size();
    if (good)
      for (int k = 1; k <= n; ++k)
        if (((unsigned char)bytes[i + k] & 0xC0) != 0x80) { good = false; break; }
    if (good) { out.append(bytes, i, (size_t)n + 1); i += (size_t)n + 1; }
    else { out += "\xEF\xBF\xBD"; ++i; }
  }
  return out;
}

std::vector<int64_t> BpeTokenizer::encode(const std::string& text) const {
  // tokenizers' AddedVocabulary first cuts the added tokens (special or not) out of the text, longest match at the
  // leftmost position; the stretches in between go through the normaliser (NFC when tokenizer.json asks for it), the
  // pre-tokeniser and byte-level BPE.
  std::vector<int64_t> ids;
  auto encode_plain = [&](const std::string& raw_seg) {
    const std::string seg = nfc_ ? normalize_nfc(raw_seg) : raw_seg;
    std::vector<uint32_t> cps;
    std::vector<size_t> off;  // byte offset of every code point (+ end)
    for (size_t i = 0; i < seg.size();) { off.push_back(i); cps.push_back(next_cp(seg, i)); }
    off.push_back(seg.size());
    for (size_t i = 0; i < cps.size();) {
      const size_t j = match_piece(cps, i);
      std::vector<std::string> sym;
      for (size_t k = off[i]; k < off[j]; ++k) sym.push_back(cp_of_byte_[(unsigned char)seg[k]]);
      while (sym.size() > 1) {  // merge the lowest-ranked adjacent pair until none is left
        int best = INT_MAX;
        size_t at = 0;
        for (size_t k = 0; k + 1 < sym.size(); ++k) {
          auto it = merge_rank_.find(sym[k] + " " + sym[k + 1]);
          if (it != merge_rank_.end() && it->second < best) { best = it->second; at = k; }
        }
        if (best == INT_MAX) break;
        sym[at] += sym[at + 1];
        sym.erase(sym.begin() + (long)at + 1);
      }
      for (auto& sy : sym) {
        auto it = token_to_id_.find(sy);
        if (it == token_to_id_.end()) fail("Tokenization failed: symbol not in vocabulary");
        ids.push_back(it->second);
      }
      i = j;
    }
  };
  size_t seg0 = 0, i = 0;
  while (i < text.size()) {
    size_t best_len = 0;
    int64_t best_id = -1;
    for (int64_t id : added_by_first_[(unsigned char)text[i]]) {  // longest first: the first hit is the longest match
      const std::string& tk = id_to_token_[(size_t)id];
      if (text.compare(i, tk.size(), tk) == 0) { best_len = tk.size(); best_id = id; break; }
    }
    if (best_id >= 0) {
      if (i > seg0) encode_plain(text.substr(seg0, i - seg0));
      ids.push_back(best_id);
      i += best_len;
      seg0 = i;
    } else {
      ++i;
      while (i < text.size() && ((unsigned char)text[i] & 0xC0) == 0x80) ++i;  // next code point boundary
    }
  }
  if (seg0 < text.size()) encode_plain(text.substr(seg0));
  return ids;
}

// ---------------------------------------------------------------------------------------------------
namespace {
bool cp_is_space(uint32_t cp) {
  return cp == ' ' || (cp >= 9 && cp <= 13) || cp == 0x85 || cp == 0xA0 || cp == 0x1680 || (cp >= 0x2000 && cp <= 0x200A) ||
         cp == 0x2028 || cp == 0x2029 || cp == 0x202F || cp == 0x205F || cp == 0x3000;
}
// char::is_alphabetic approximation: ASCII letters; outside ASCII everything except whitespace, general/CJK
// punctuation and symbol blocks (exact Unicode tables are not needed for "language <Name> text" outputs)
bool cp_is_alpha(uint32_t cp) {
  if (cp < 0x80) return (cp >= 'A' && cp <= 'Z') || (cp >= 'a' && cp <= 'z');
  if (cp_is_space(cp)) return false;
  if ((cp >= 0xA1 && cp <= 0xBF && cp != 0xAA && cp != 0xB5 && cp != 0xBA) || cp == 0xD7 || cp == 0xF7) return false;
  if ((cp >= 0x2000 && cp <= 0x2BFF) || (cp >= 0x3000 && cp <= 0x303F) || (cp >= 0xFF00 && cp <= 0xFF20) ||
      (cp >= 0xFF3B && cp <= 0xFF40) || (cp >= 0xFF5B && cp <= 0xFF65) || (cp >= 0xFE30 && cp <= 0xFE6F))
    return false;
  return true;
}
std::string trim(const std::string& s) {  // str::trim (Unicode white space)
  size_t b = 0, e = s.size();
  while (b < e) { size_t i = b; uint32_t cp = next_cp(s, i); if (!cp_is_space(cp)) break; b = i; }
  while (e > b) {
    size_t k = e - 1;
    while (k > b && ((unsigned char)s[k] & 0xC0) == 0x80) --k;
    size_t i = k;
    uint32_t cp = next_cp(s, i);
    if (!cp_is_space(cp)) break;
    e = k;
  }
  return s.substr(b, e - b);
}
}  // namespace

void parse_asr_output(const std::string& raw_in, bool language_forced, std::string& language, std::string& text) {
  if (language_forced) { language = "forced"; text = trim(raw_in); return; }  // inference.rs:277-279
  const std::string raw = trim(raw_in);
  const std::string pre = "language ";
  if (raw.compare(0, pre.size(), pre) == 0) {
    const std::string rest = raw.substr(pre.size());
    const std::string tag = "<asr_text>";
    size_t p = rest.find(tag);
    if (p != std::string::npos) {  // inference.rs:284-288
      language = trim(rest.substr(0, p));
      text = trim(rest.substr(p + tag.size()));
      return;
    }
    size_t lang_end = 0;  // inference.rs:289-301
    for (size_t i = 0; i < rest.size();) {
      size_t start = i;
      uint32_t cp = next_cp(rest, i);
      if (cp_is_space(cp) || !cp_is_alpha(cp)) { lang_end = start; break; }
      lang_end = i;
    }
    if (lang_end > 0) {
      language = rest.substr(0, lang_end);
      text = trim(rest.substr(lang_end));
      return;
    }
  }
  language = "unknown";
  text = raw;
}

std::string capitalize_first(const std::string& s) {  // inference.rs:307-313 (ASCII upper-casing of the first char)
  if (s.empty()) return s;
  std::string r = s;
  if ((unsigned char)r[0] < 0x80) r[0] = (char)toupper((unsigned char)r[0]);
  return r;
}

}  // namespace q3a
