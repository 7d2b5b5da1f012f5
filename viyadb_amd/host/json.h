// json.h — a small JSON reader + the util::Config accessor surface the path uses.
// The reference wraps nlohmann::json in util::Config (src/util/config.h:29-77); that library is
// not available here, so this is a from-scratch reader exposing the same accessor names
// (str / num / boolean / exists / sub / sublist / strlist) with the same "missing key throws
// std::invalid_argument" behaviour the callers rely on.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace viya {
namespace util {

class Json {
public:
  enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ };
  Kind kind = NUL;
  bool b = false;
  double d = 0;
  bool integral = false;
  bool negative = false;
  uint64_t u = 0;  // magnitude when integral
  std::string s;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;

  const Json* find(const std::string& key) const {
    for (auto& kv : obj)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }

  static Json parse(const std::string& text) {
    size_t p = 0;
    Json j = parse_value(text, p);
    skip_ws(text, p);
    if (p != text.size()) throw std::invalid_argument("JSON: trailing characters");
    return j;
  }

private:
  static void skip_ws(const std::string& t, size_t& p) {
    while (p < t.size() && (t[p] == ' ' || t[p] == '\n' || t[p] == '\t' || t[p] == '\r')) ++p;
  }
  static void append_utf8(std::string& out, unsigned cp) {
    if (cp < 0x80) out += (char)cp;
    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
    else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
  }
  static std::string parse_string(const std::string& t, size_t& p) {
    std::string out;
    ++p;  // opening quote
    while (p < t.size() && t[p] != '"') {
      char c = t[p++];
      if (c != '\\') { out += c; continue; }
      if (p >= t.size()) break;
      char e = t[p++];
      switch (e) {
        case 'n': out += '\n'; break;
        case 't': out += '\t'; break;
        case 'r': out += '\r'; break;
        case 'b': out += '\b'; break;
        case 'f': out += '\f'; break;
        case 'u': {
          if (p + 4 > t.size()) throw std::invalid_argument("JSON: bad \\u escape");
          append_utf8(out, (unsigned)std::strtoul(t.substr(p, 4).c_str(), nullptr, 16));
          p += 4;
        } break;
        default: out += e; break;
      }
    }
    if (p >= t.size()) throw std::invalid_argument("JSON: unterminated string");
    ++p;
    return out;
  }
  static Json parse_value(const std::string& t, size_t& p) {
    skip_ws(t, p);
    if (p >= t.size()) throw std::invalid_argument("JSON: unexpected end");
    Json j;
    char c = t[p];
    if (c == '{') {
      j.kind = OBJ;
      ++p;
      skip_ws(t, p);
      if (p < t.size() && t[p] == '}') { ++p; return j; }
      for (;;) {
        skip_ws(t, p);
        if (p >= t.size() || t[p] != '"') throw std::invalid_argument("JSON: expected key");
        std::string k = parse_string(t, p);
        skip_ws(t, p);
        if (p >= t.size() || t[p] != ':') throw std::invalid_argument("JSON: expected ':'");
        ++p;
        j.obj.emplace_back(k, parse_value(t, p));
        skip_ws(t, p);
        if (p < t.size() && t[p] == ',') { ++p; continue; }
        if (p < t.size() && t[p] == '}') { ++p; return j; }
        throw std::invalid_argument("JSON: expected ',' or '}'");
      }
    }
    if (c == '[') {
      j.kind = ARR;
      ++p;
      skip_ws(t, p);
      if (p < t.size() && t[p] == ']') { ++p; return j; }
      for (;;) {
        j.arr.push_back(parse_value(t, p));
        skip_ws(t, p);
        if (p < t.size() && t[p] == ',') { ++p; continue; }
        if (p < t.size() && t[p] == ']') { ++p; return j; }
        throw std::invalid_argument("JSON: expected ',' or ']'");
      }
    }
    if (c == '"') { j.kind = STR; j.s = parse_string(t, p); return j; }
    if (t.compare(p, 4, "true") == 0) { j.kind = BOOL; j.b = true; p += 4; return j; }
    if (t.compare(p, 5, "false") == 0) { j.kind = BOOL; j.b = false; p += 5; return j; }
    if (t.compare(p, 4, "null") == 0) { p += 4; return j; }
    size_t q = p;
    if (q < t.size() && (t[q] == '-' || t[q] == '+')) ++q;
    bool frac = false;
    while (q < t.size() && (isdigit((unsigned char)t[q]) || t[q] == '.' || t[q] == 'e' || t[q] == 'E' || t[q] == '-' || t[q] == '+')) {
      if (t[q] == '.' || t[q] == 'e' || t[q] == 'E') frac = true;
      ++q;
    }
    if (q == p) throw std::invalid_argument("JSON: unexpected character");
    std::string num = t.substr(p, q - p);
    j.kind = NUM;
    j.d = std::strtod(num.c_str(), nullptr);
    if (!frac) {
      j.integral = true;
      j.negative = num[0] == '-';
      j.u = std::strtoull(num.c_str() + (j.negative || num[0] == '+' ? 1 : 0), nullptr, 10);
    }
    p = q;
    return j;
  }
};

// Accessors named after util::Config (src/util/config.h:29-77).
class Config {
public:
  Config() : j_(std::make_shared<Json>()) { j_->kind = Json::OBJ; }
  explicit Config(const std::string& text) : j_(std::make_shared<Json>(Json::parse(text))) {}
  explicit Config(const Json& j) : j_(std::make_shared<Json>(j)) {}

  bool exists(const std::string& key) const { return j_->kind == Json::OBJ && j_->find(key) != nullptr; }

  std::string str(const std::string& key) const {
    const Json& v = need(key);
    if (v.kind != Json::STR) throw std::invalid_argument("Config key '" + key + "' is not a string");
    return v.s;
  }
  std::string str(const std::string& key, const std::string& dflt) const { return exists(key) ? str(key) : dflt; }

  // numbers may be given as JSON numbers or (as the reference's tests sometimes do) strings
  long num(const std::string& key) const {
    const Json& v = need(key);
    if (v.kind == Json::NUM) return v.integral ? (v.negative ? -(long)v.u : (long)v.u) : (long)v.d;
    if (v.kind == Json::STR) return std::stol(v.s);
    throw std::invalid_argument("Config key '" + key + "' is not a number");
  }
  long num(const std::string& key, long dflt) const { return exists(key) ? num(key) : dflt; }
  uint64_t unum(const std::string& key, uint64_t dflt) const {
    if (!exists(key)) return dflt;
    const Json& v = need(key);
    if (v.kind == Json::NUM) return v.integral ? v.u : (uint64_t)v.d;
    if (v.kind == Json::STR) return std::stoull(v.s);
    throw std::invalid_argument("Config key '" + key + "' is not a number");
  }
  bool boolean(const std::string& key, bool dflt) const {
    if (!exists(key)) return dflt;
    const Json& v = need(key);
    if (v.kind != Json::BOOL) throw std::invalid_argument("Config key '" + key + "' is not a boolean");
    return v.b;
  }
  Config sub(const std::string& key, bool return_empty = false) const {
    if (!exists(key)) {
      if (return_empty) return Config();
      throw std::invalid_argument("Missing configuration key: " + key);
    }
    const Json& v = need(key);
    if (v.kind != Json::OBJ) throw std::invalid_argument("Config key '" + key + "' is not an object");
    return Config(v);
  }
  std::vector<Config> sublist(const std::string& key) const {
    const Json& v = need(key);
    if (v.kind != Json::ARR) throw std::invalid_argument("Config key '" + key + "' is not a list");
    std::vector<Config> out;
    for (auto& e : v.arr) {
      if (e.kind != Json::OBJ) throw std::invalid_argument("Config key '" + key + "' is not a list of objects");
      out.emplace_back(e);
    }
    return out;
  }
  std::vector<std::string> strlist(const std::string& key) const {
    const Json& v = need(key);
    if (v.kind != Json::ARR) throw std::invalid_argument("Config key '" + key + "' is not a list");
    std::vector<std::string> out;
    for (auto& e : v.arr) {
      if (e.kind != Json::STR) throw std::invalid_argument("Config key '" + key + "' is not a list of strings");
      out.push_back(e.s);
    }
    return out;
  }
  const Json& json() const { return *j_; }

private:
  const Json& need(const std::string& key) const {
    const Json* v = j_->kind == Json::OBJ ? j_->find(key) : nullptr;
    if (!v) throw std::invalid_argument("Missing configuration key: " + key);
    return *v;
  }
  std::shared_ptr<Json> j_;
};

}  // namespace util
}  // namespace viya
