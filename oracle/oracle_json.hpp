// oracle_json.hpp - the test oracle's OWN JSON reader and (b)gzip reader.  TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg use oracle/.
//
// Rounds 1-3 shared csrc/common/mini_json.hpp and gz_text.hpp with the product: a misreading there would have been common to checker and
// checked.  This file is written independently of them (a token scanner feeding an explicit stack of open containers instead of recursive
// value functions; inflate() driven member by member instead of gzread) and is itself compared with Python's json / gzip modules
// (tests/test_common_utils.py), as is the product's reader (through tests/hostsim).  Same surface as far as the oracle uses one: the
// RapidJSON-like accessors of the reference's own code (json_config.cc reads its files through RapidJSON).
#pragma once
#include <zlib.h>

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace oracle_json {

struct Value {
  enum Type { Null, Bool, Int, Double, String, Array, Object } type = Null;
  bool b = false;
  int64_t i = 0;
  double d = 0;
  std::string s;
  std::vector<Value> arr;
  std::vector<std::pair<std::string, Value>> obj;   // document order, duplicates kept (lookups take the first)
  bool IsNull() const { return type == Null; }
  bool IsBool() const { return type == Bool; }
  bool IsInt64() const { return type == Int; }
  bool IsNumber() const { return type == Int || type == Double; }
  bool IsString() const { return type == String; }
  bool IsArray() const { return type == Array; }
  bool IsObject() const { return type == Object; }
  bool GetBool() const { need(Bool); return b; }
  int64_t GetInt64() const { if (type == Double) return (int64_t)d; need(Int); return i; }
  double GetDouble() const { if (type == Int) return (double)i; need(Double); return d; }
  const std::string& GetString() const { need(String); return s; }
  size_t Size() const { return type == Object ? obj.size() : (need(Array), arr.size()); }
  size_t MemberCount() const { need(Object); return obj.size(); }
  bool HasMember(const char* k) const { return find(k) != nullptr; }
  const Value& operator[](const char* k) const { const Value* v = find(k); if (!v) throw std::runtime_error(std::string("JSON: no member \"") + k + "\""); return *v; }
  const Value& operator[](const std::string& k) const { return (*this)[k.c_str()]; }
  const Value& operator[](size_t n) const { need(Array); if (n >= arr.size()) throw std::runtime_error("JSON: array index out of range"); return arr[n]; }
  const Value& operator[](int n) const { return (*this)[(size_t)n]; }
  const Value& operator[](unsigned n) const { return (*this)[(size_t)n]; }
 private:
  void need(Type t) const { if (type != t) throw std::runtime_error("JSON: value has another type than the caller expects"); }
  const Value* find(const char* k) const {
    if (type != Object) return nullptr;
    for (const auto& kv : obj) if (kv.first == k) return &kv.second;
    return nullptr;
  }
};

namespace detail {
struct Scanner {
  const char* p; const char* e;
  [[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("JSON: ") + what); }
  void space() { while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
  static void utf8(std::string& o, uint32_t c) {
    if (c < 0x80) o += (char)c;
    else if (c < 0x800) { o += (char)(0xC0 | (c >> 6)); o += (char)(0x80 | (c & 0x3F)); }
    else if (c < 0x10000) { o += (char)(0xE0 | (c >> 12)); o += (char)(0x80 | ((c >> 6) & 0x3F)); o += (char)(0x80 | (c & 0x3F)); }
    else { o += (char)(0xF0 | (c >> 18)); o += (char)(0x80 | ((c >> 12) & 0x3F)); o += (char)(0x80 | ((c >> 6) & 0x3F)); o += (char)(0x80 | (c & 0x3F)); }
  }
  uint32_t hex4() {
    if (e - p < 4) fail("short \\u escape");
    uint32_t v = 0;
    for (int k = 0; k < 4; ++k, ++p) {
      const char c = *p;
      v = v * 16 + (c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : (fail("bad \\u escape"), 0));
    }
    return v;
  }
  std::string string_body() {          // p is behind the opening quote
    std::string o;
    for (;;) {
      if (p >= e) fail("unterminated string");
      const unsigned char c = (unsigned char)*p++;
      if (c == '"') return o;
      if (c < 0x20) fail("control character in string");
      if (c != '\\') { o += (char)c; continue; }
      if (p >= e) fail("unterminated escape");
      const char x = *p++;
      switch (x) {
        case '"': o += '"'; break; case '\\': o += '\\'; break; case '/': o += '/'; break;
        case 'b': o += '\b'; break; case 'f': o += '\f'; break; case 'n': o += '\n'; break; case 'r': o += '\r'; break; case 't': o += '\t'; break;
        case 'u': {
          uint32_t c1 = hex4();
          if (c1 >= 0xD800 && c1 < 0xDC00 && e - p >= 6 && p[0] == '\\' && p[1] == 'u') {
            const char* save = p; p += 2;
            const uint32_t c2 = hex4();
            if (c2 >= 0xDC00 && c2 < 0xE000) c1 = 0x10000 + ((c1 - 0xD800) << 10) + (c2 - 0xDC00); else p = save;
          }
          utf8(o, c1);
          break;
        }
        default: fail("unknown escape");
      }
    }
  }
  Value number() {
    const char* b = p;
    if (p < e && *p == '-') ++p;
    if (p >= e || *p < '0' || *p > '9') fail("bad number");
    if (*p == '0') ++p; else while (p < e && *p >= '0' && *p <= '9') ++p;
    bool real = false;
    if (p < e && *p == '.') { real = true; ++p; if (p >= e || *p < '0' || *p > '9') fail("bad fraction"); while (p < e && *p >= '0' && *p <= '9') ++p; }
    if (p < e && (*p == 'e' || *p == 'E')) { real = true; ++p; if (p < e && (*p == '+' || *p == '-')) ++p; if (p >= e || *p < '0' || *p > '9') fail("bad exponent"); while (p < e && *p >= '0' && *p <= '9') ++p; }
    const std::string t(b, p);
    Value v;
    if (!real) {
      errno = 0;
      char* end = nullptr;
      const long long x = strtoll(t.c_str(), &end, 10);
      if (errno == 0 && end && *end == 0) { v.type = Value::Int; v.i = x; return v; }
    }
    v.type = Value::Double; v.d = strtod(t.c_str(), nullptr);
    return v;
  }
  bool word(const char* w) { const size_t n = strlen(w); if ((size_t)(e - p) >= n && memcmp(p, w, n) == 0) { p += n; return true; } return false; }
};
}  // namespace detail

// One loop over the tokens; `open` holds the containers being filled (no recursion: depth costs heap, not stack).
inline Value parse(const std::string& text) {
  detail::Scanner sc{text.data(), text.data() + text.size()};
  struct Open { Value v; std::string key; bool have_key = false, want_value = true, first = true; };
  std::vector<Open> open;
  Value root;
  bool have_root = false;
  auto deliver = [&](Value&& v) {
    if (open.empty()) { if (have_root) sc.fail("more than one value"); root = std::move(v); have_root = true; return; }
    Open& o = open.back();
    if (o.v.type == Value::Array) o.v.arr.push_back(std::move(v));
    else { o.v.obj.emplace_back(std::move(o.key), std::move(v)); o.have_key = false; }
    o.want_value = false;
  };
  for (;;) {
    sc.space();
    if (sc.p >= sc.e) break;
    if (have_root && open.empty()) sc.fail("text behind the document");
    const char c = *sc.p;
    if (!open.empty()) {
      Open& o = open.back();
      const bool is_obj = o.v.type == Value::Object;
      if (c == (is_obj ? '}' : ']')) {
        if (o.want_value && !o.first) sc.fail("trailing comma");
        if (is_obj && o.have_key) sc.fail("member without a value");
        ++sc.p;
        Value done = std::move(o.v);
        open.pop_back();
        deliver(std::move(done));
        continue;
      }
      if (!o.want_value) { if (c != ',') sc.fail("',' expected"); ++sc.p; o.want_value = true; o.first = false; continue; }
      if (is_obj && !o.have_key) {
        if (c != '"') sc.fail("member name expected");
        ++sc.p;
        o.key = sc.string_body();
        sc.space();
        if (sc.p >= sc.e || *sc.p != ':') sc.fail("':' expected");
        ++sc.p;
        o.have_key = true;
        continue;
      }
    }
    if (c == '{' || c == '[') { ++sc.p; Open o; o.v.type = c == '{' ? Value::Object : Value::Array; open.push_back(std::move(o)); continue; }
    Value v;
    if (c == '"') { ++sc.p; v.type = Value::String; v.s = sc.string_body(); }
    else if (c == '-' || (c >= '0' && c <= '9')) v = sc.number();
    else if (sc.word("true")) { v.type = Value::Bool; v.b = true; }
    else if (sc.word("false")) { v.type = Value::Bool; v.b = false; }
    else if (sc.word("null")) v.type = Value::Null;
    else sc.fail("unexpected character");
    deliver(std::move(v));
  }
  if (!open.empty()) sc.fail("unterminated array or object");
  if (!have_root) sc.fail("empty document");
  return root;
}

inline std::string read_text_file(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open " + path);
  std::string out;
  char buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.append(buf, n);
  fclose(f);
  return out;
}
inline Value parse_file(const std::string& path) { return parse(read_text_file(path)); }

// the bytes of a plain, gzip or BGZF (= multi-member gzip) file: members are inflated one after the other with inflate(); a file that does not
// begin with the gzip magic is returned as it is
inline std::string gz_read_all(const std::string& path) {
  const std::string raw = read_text_file(path);
  if (raw.size() < 2 || (unsigned char)raw[0] != 0x1f || (unsigned char)raw[1] != 0x8b) return raw;
  std::string out;
  size_t at = 0;
  while (at < raw.size()) {
    if (raw.size() - at < 2 || (unsigned char)raw[at] != 0x1f || (unsigned char)raw[at + 1] != 0x8b) break;   // (trailing bytes that are no member: ignored, like gzip's readers)
    z_stream z;
    memset(&z, 0, sizeof(z));
    if (inflateInit2(&z, 15 + 16) != Z_OK) throw std::runtime_error("inflateInit2 failed");
    z.next_in = (Bytef*)(raw.data() + at);
    z.avail_in = (uInt)std::min<size_t>(raw.size() - at, 1u << 30);
    char buf[1 << 16];
    int rc;
    do {
      z.next_out = (Bytef*)buf; z.avail_out = sizeof(buf);
      rc = inflate(&z, Z_NO_FLUSH);
      if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&z); throw std::runtime_error("inflate failed for " + path); }
      out.append(buf, sizeof(buf) - z.avail_out);
    } while (rc != Z_STREAM_END);
    at = (size_t)((const char*)z.next_in - raw.data());
    inflateEnd(&z);
  }
  return out;
}

}  // namespace oracle_json
