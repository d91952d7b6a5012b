// json.h — minimal JSON value + recursive-descent parser (PFW1 header, asr.json).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.h"

namespace pf {

struct Json {
  enum Type { Null, Bool, Num, Str, Arr, Obj } type = Null;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;

  const Json* get(const std::string& k) const {
    for (auto& kv : obj)
      if (kv.first == k) return &kv.second;
    return nullptr;
  }
  double num_or(const std::string& k, double d) const {
    const Json* j = get(k);
    if (!j) return d;
    if (j->type == Num) return j->num;
    if (j->type == Bool) return j->b ? 1 : 0;
    return d;
  }
  // an integer field: the number must be finite, integral and within +-2^53 (a "1e400" or "0.5" where a size belongs is
  // a format error, not a cast of inf to int)
  int64_t int_or(const std::string& k, int64_t d, int64_t lo = -((int64_t)1 << 53), int64_t hi = (int64_t)1 << 53) const {
    const Json* j = get(k);
    if (!j) return d;
    if (j->type == Bool) return j->b ? 1 : 0;
    if (j->type != Num) return d;
    if (!(j->num >= (double)lo && j->num <= (double)hi) || j->num != (double)(int64_t)j->num)
      throw Error(PF_ERR_FORMAT, "json: '" + k + "' must be an integer in range");
    return (int64_t)j->num;
  }
  std::string str_or(const std::string& k, const std::string& d) const {
    const Json* j = get(k);
    return (j && j->type == Str) ? j->str : d;
  }
  bool bool_or(const std::string& k, bool d) const {
    const Json* j = get(k);
    if (!j) return d;
    if (j->type == Bool) return j->b;
    if (j->type == Num) return j->num != 0;
    if (j->type == Str) return j->str == "true" || j->str == "True";
    return d;
  }
};

class JsonParser {
 public:
  JsonParser(const char* s, size_t n) : p_(s), e_(s + n) {}
  Json parse() {
    Json v = value();
    ws();
    return v;
  }

 private:
  const char* p_;
  const char* e_;
  [[noreturn]] void fail(const char* m) { throw Error(PF_ERR_FORMAT, std::string("json: ") + m); }
  void ws() {
    while (p_ < e_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\r' || *p_ == '\t')) ++p_;
  }
  int depth_ = 0;
  struct Depth {                                  // containers nest at most 64 deep (a "[[[[..." header must
    int& d;                                       // fail with PF_ERR_FORMAT, not overflow the stack)
    explicit Depth(int& x) : d(x) { ++d; }
    ~Depth() { --d; }
  };
  Json value() {
    Depth guard(depth_);
    if (depth_ > 64) fail("nesting too deep");
    ws();
    if (p_ >= e_) fail("unexpected end");
    Json v;
    char c = *p_;
    if (c == '{') {
      ++p_;
      v.type = Json::Obj;
      ws();
      if (p_ < e_ && *p_ == '}') { ++p_; return v; }
      for (;;) {
        ws();
        Json k = string();
        ws();
        if (p_ >= e_ || *p_ != ':') fail("expected ':'");
        ++p_;
        v.obj.emplace_back(k.str, value());
        ws();
        if (p_ < e_ && *p_ == ',') { ++p_; continue; }
        if (p_ < e_ && *p_ == '}') { ++p_; break; }
        fail("expected ',' or '}'");
      }
      return v;
    }
    if (c == '[') {
      ++p_;
      v.type = Json::Arr;
      ws();
      if (p_ < e_ && *p_ == ']') { ++p_; return v; }
      for (;;) {
        v.arr.push_back(value());
        ws();
        if (p_ < e_ && *p_ == ',') { ++p_; continue; }
        if (p_ < e_ && *p_ == ']') { ++p_; break; }
        fail("expected ',' or ']'");
      }
      return v;
    }
    if (c == '"') return string();
    if (c == 't' && e_ - p_ >= 4 && std::string(p_, 4) == "true") { p_ += 4; v.type = Json::Bool; v.b = true; return v; }
    if (c == 'f' && e_ - p_ >= 5 && std::string(p_, 5) == "false") { p_ += 5; v.type = Json::Bool; v.b = false; return v; }
    if (c == 'n' && e_ - p_ >= 4 && std::string(p_, 4) == "null") { p_ += 4; return v; }
    char* end = nullptr;
    std::string tmp(p_, std::min<size_t>(64, e_ - p_));
    double d = std::strtod(tmp.c_str(), &end);
    if (end == tmp.c_str()) fail("bad token");
    p_ += end - tmp.c_str();
    v.type = Json::Num;
    v.num = d;
    return v;
  }
  static void put_utf8(std::string& o, unsigned cp) {
    if (cp < 0x80) o += (char)cp;
    else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
    else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
  }
  Json string() {
    if (p_ >= e_ || *p_ != '"') fail("expected string");
    ++p_;
    Json v;
    v.type = Json::Str;
    while (p_ < e_ && *p_ != '"') {
      if (*p_ == '\\') {
        ++p_;
        if (p_ >= e_) fail("bad escape");
        char c = *p_++;
        switch (c) {
          case 'n': v.str += '\n'; break;
          case 't': v.str += '\t'; break;
          case 'r': v.str += '\r'; break;
          case 'b': v.str += '\b'; break;
          case 'f': v.str += '\f'; break;
          case 'u': {
            if (e_ - p_ < 4) fail("bad \\u");
            unsigned cp = (unsigned)std::strtoul(std::string(p_, 4).c_str(), nullptr, 16);
            p_ += 4;
            if (cp >= 0xD800 && cp < 0xDC00 && e_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
              unsigned lo = (unsigned)std::strtoul(std::string(p_ + 2, 4).c_str(), nullptr, 16);
              p_ += 6;
              cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
            }
            put_utf8(v.str, cp);
            break;
          }
          default: v.str += c;
        }
      } else {
        v.str += *p_++;
      }
    }
    if (p_ >= e_) fail("unterminated string");
    ++p_;
    return v;
  }
};

}  // namespace pf
