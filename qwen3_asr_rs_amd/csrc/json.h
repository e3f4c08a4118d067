// Minimal JSON reader (objects, arrays, strings, numbers, bools, null) -- enough for config.json,
// the safetensors header and model.safetensors.index.json.  Header-only, no dependencies.
#pragma once
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace q3a {

struct Json {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;  // insertion order kept

  const Json* find(const std::string& key) const {
    if (kind != Obj) return nullptr;
    for (auto& kv : obj)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
  const Json& at(const std::string& key) const {
    const Json* j = find(key);
    if (!j) throw std::runtime_error("json: missing key '" + key + "'");
    return *j;
  }
  double num_or(const std::string& key, double dflt) const {
    const Json* j = find(key);
    return (j && j->kind == Num) ? j->num : dflt;
  }
  bool bool_or(const std::string& key, bool dflt) const {
    const Json* j = find(key);
    return (j && j->kind == Bool) ? j->b : dflt;
  }
};

class JsonParser {
 public:
  JsonParser(const char* s, size_t n) : p_(s), end_(s + n) {}
  Json parse() {
    Json v = value();
    ws();
    if (p_ != end_) fail("trailing characters");
    return v;
  }

 private:
  const char* p_;
  const char* end_;
  [[noreturn]] void fail(const char* msg) { throw std::runtime_error(std::string("json parse error: ") + msg); }
  void ws() {
    while (p_ < end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_;
  }
  Json value() {
    ws();
    if (p_ >= end_) fail("unexpected end");
    char c = *p_;
    if (c == '{') return object();
    if (c == '[') return array();
    if (c == '"') {
      Json j;
      j.kind = Json::Str;
      j.str = string();
      return j;
    }
    if (c == 't' || c == 'f' || c == 'n') return literal();
    return number();
  }
  Json literal() {
    Json j;
    auto match = [&](const char* w) {
      size_t n = strlen(w);
      if ((size_t)(end_ - p_) >= n && memcmp(p_, w, n) == 0) {
        p_ += n;
        return true;
      }
      return false;
    };
    if (match("true")) {
      j.kind = Json::Bool;
      j.b = true;
    } else if (match("false")) {
      j.kind = Json::Bool;
      j.b = false;
    } else if (match("null")) {
      j.kind = Json::Null;
    } else
      fail("bad literal");
    return j;
  }
  Json number() {
    const char* s = p_;
    while (p_ < end_ && (strchr("+-0123456789.eE", *p_) != nullptr)) ++p_;
    if (s == p_) fail("bad number");
    Json j;
    j.kind = Json::Num;
    j.num = strtod(std::string(s, p_).c_str(), nullptr);
    return j;
  }
  std::string string() {
    ++p_;  // opening quote
    std::string out;
    while (p_ < end_ && *p_ != '"') {
      if (*p_ == '\\') {
        ++p_;
        if (p_ >= end_) fail("bad escape");
        switch (*p_) {
          case 'n': out += '\n'; break;
          case 't': out += '\t'; break;
          case 'r': out += '\r'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'u': {
            if (end_ - p_ < 5) fail("bad \\u");
            unsigned cp = (unsigned)strtoul(std::string(p_ + 1, p_ + 5).c_str(), nullptr, 16);
            p_ += 4;
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: out += *p_;
        }
        ++p_;
      } else {
        out += *p_++;
      }
    }
    if (p_ >= end_) fail("unterminated string");
    ++p_;
    return out;
  }
  Json array() {
    Json j;
    j.kind = Json::Arr;
    ++p_;
    ws();
    if (p_ < end_ && *p_ == ']') { ++p_; return j; }
    for (;;) {
      j.arr.push_back(value());
      ws();
      if (p_ >= end_) fail("unterminated array");
      if (*p_ == ',') { ++p_; continue; }
      if (*p_ == ']') { ++p_; break; }
      fail("expected , or ]");
    }
    return j;
  }
  Json object() {
    Json j;
    j.kind = Json::Obj;
    ++p_;
    ws();
    if (p_ < end_ && *p_ == '}') { ++p_; return j; }
    for (;;) {
      ws();
      if (p_ >= end_ || *p_ != '"') fail("expected key");
      std::string k = string();
      ws();
      if (p_ >= end_ || *p_ != ':') fail("expected :");
      ++p_;
      j.obj.emplace_back(std::move(k), value());
      ws();
      if (p_ >= end_) fail("unterminated object");
      if (*p_ == ',') { ++p_; continue; }
      if (*p_ == '}') { ++p_; break; }
      fail("expected , or }");
    }
    return j;
  }
};

inline Json parse_json(const std::string& text) { return JsonParser(text.data(), text.size()).parse(); }

}  // namespace q3a
