// mini_json.hpp - a small dependency-free JSON DOM (RapidJSON is not available in this image).
//
// Format-agnostic utility shared by the product host code and by the test oracle: it knows
// nothing about GenomicsDB.  Objects keep their members in document order (the reference relies
// on RapidJSON's document-order iteration for vid "fields"/"contigs": vid_mapper.cc:1224-1330).
#pragma once
#include <cerrno>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace mini_json {

class Value {
 public:
  enum Type { Null, Bool, Int, Double, String, Array, Object };
  Type type = Null;
  bool b = false;
  int64_t i = 0;
  double d = 0;
  std::string s;
  std::vector<Value> arr;
  std::vector<std::pair<std::string, Value>> obj;

  bool IsNull() const { return type == Null; }
  bool IsBool() const { return type == Bool; }
  bool IsInt64() const { return type == Int; }
  bool IsNumber() const { return type == Int || type == Double; }
  bool IsString() const { return type == String; }
  bool IsArray() const { return type == Array; }
  bool IsObject() const { return type == Object; }
  bool GetBool() const { need(Bool); return b; }
  int64_t GetInt64() const { need(Int); return i; }
  double GetDouble() const { if (type == Int) return (double)i; need(Double); return d; }
  const std::string& GetString() const { need(String); return s; }
  size_t Size() const { return type == Array ? arr.size() : obj.size(); }
  size_t MemberCount() const { need(Object); return obj.size(); }
  const Value& operator[](size_t idx) const { need(Array); return arr.at(idx); }
  const Value& operator[](int idx) const { return (*this)[(size_t)idx]; }
  const Value& operator[](unsigned idx) const { return (*this)[(size_t)idx]; }
  bool HasMember(const std::string& k) const {
    if (type != Object) return false;
    for (auto& kv : obj) if (kv.first == k) return true;
    return false;
  }
  const Value& operator[](const std::string& k) const {
    need(Object);
    for (auto& kv : obj) if (kv.first == k) return kv.second;
    throw std::runtime_error("mini_json: missing key '" + k + "'");
  }
  const Value& operator[](const char* k) const { return (*this)[std::string(k)]; }

 private:
  void need(Type t) const {
    if (type != t) throw std::runtime_error("mini_json: wrong value type");
  }
};

class Parser {
 public:
  explicit Parser(const std::string& text) : p_(text.c_str()), end_(text.c_str() + text.size()) {}
  Value parse() {
    Value v = value();
    ws();
    if (p_ != end_) fail("trailing characters");
    return v;
  }

 private:
  const char* p_;
  const char* end_;
  [[noreturn]] void fail(const char* m) { throw std::runtime_error(std::string("mini_json: ") + m); }
  void ws() { while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) ++p_; }
  Value value() {
    ws();
    if (p_ >= end_) fail("unexpected end");
    switch (*p_) {
      case '{': return object();
      case '[': return array();
      case '"': { Value v; v.type = Value::String; v.s = string(); return v; }
      case 't': lit("true"); { Value v; v.type = Value::Bool; v.b = true; return v; }
      case 'f': lit("false"); { Value v; v.type = Value::Bool; v.b = false; return v; }
      case 'n': lit("null"); return Value();
      default: return number();
    }
  }
  void lit(const char* w) {
    size_t n = strlen(w);
    if ((size_t)(end_ - p_) < n || strncmp(p_, w, n) != 0) fail("bad literal");
    p_ += n;
  }
  Value number() {
    const char* s = p_;
    bool is_int = true;
    if (p_ < end_ && (*p_ == '-' || *p_ == '+')) ++p_;
    while (p_ < end_ && ((*p_ >= '0' && *p_ <= '9') || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || *p_ == '-' || *p_ == '+')) {
      if (*p_ == '.' || *p_ == 'e' || *p_ == 'E') is_int = false;
      ++p_;
    }
    if (p_ == s) fail("bad number");
    std::string t(s, p_);
    Value v;
    if (is_int) {
      v.type = Value::Int;
      errno = 0;
      v.i = strtoll(t.c_str(), nullptr, 10);
      if (errno == ERANGE) { v.type = Value::Double; v.d = strtod(t.c_str(), nullptr); }
    } else {
      v.type = Value::Double;
      v.d = strtod(t.c_str(), nullptr);
    }
    return v;
  }
  static void utf8(std::string& o, unsigned cp) {
    if (cp < 0x80) o += (char)cp;
    else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
    else { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
  }
  std::string string() {
    ++p_;  // opening quote
    std::string o;
    while (p_ < end_ && *p_ != '"') {
      if (*p_ == '\\') {
        ++p_;
        if (p_ >= end_) fail("bad escape");
        switch (*p_) {
          case 'n': o += '\n'; break;
          case 't': o += '\t'; break;
          case 'r': o += '\r'; break;
          case 'b': o += '\b'; break;
          case 'f': o += '\f'; break;
          case 'u': {
            if (end_ - p_ < 5) fail("bad \\u");
            unsigned cp = (unsigned)strtoul(std::string(p_ + 1, p_ + 5).c_str(), nullptr, 16);
            utf8(o, cp);
            p_ += 4;
            break;
          }
          default: o += *p_;
        }
        ++p_;
      } else {
        o += *p_++;
      }
    }
    if (p_ >= end_) fail("unterminated string");
    ++p_;
    return o;
  }
  Value array() {
    ++p_;
    Value v;
    v.type = Value::Array;
    ws();
    if (p_ < end_ && *p_ == ']') { ++p_; return v; }
    for (;;) {
      v.arr.push_back(value());
      ws();
      if (p_ < end_ && *p_ == ',') { ++p_; continue; }
      if (p_ < end_ && *p_ == ']') { ++p_; return v; }
      fail("expected , or ]");
    }
  }
  Value object() {
    ++p_;
    Value v;
    v.type = Value::Object;
    ws();
    if (p_ < end_ && *p_ == '}') { ++p_; return v; }
    for (;;) {
      ws();
      if (p_ >= end_ || *p_ != '"') fail("expected member name");
      std::string k = string();
      ws();
      if (p_ >= end_ || *p_ != ':') fail("expected :");
      ++p_;
      v.obj.emplace_back(std::move(k), value());
      ws();
      if (p_ < end_ && *p_ == ',') { ++p_; continue; }
      if (p_ < end_ && *p_ == '}') { ++p_; return v; }
      fail("expected , or }");
    }
  }
};

inline Value parse(const std::string& text) { return Parser(text).parse(); }

inline std::string read_text_file(const std::string& path) {
  std::ifstream ifs(path.c_str(), std::ios::binary);
  if (!ifs.is_open()) throw std::runtime_error("cannot open file " + path);
  std::stringstream ss;
  ss << ifs.rdbuf();
  return ss.str();
}

inline Value parse_file(const std::string& path) { return parse(read_text_file(path)); }

}  // namespace mini_json
