// gz_text.hpp - read a (b)gzip-compressed or plain text file fully into memory through zlib.
// Format-agnostic utility shared by product host code and the test oracle.
#pragma once
#include <zlib.h>
#include <stdexcept>
#include <string>

namespace gz_text {

inline std::string read_all(const std::string& path) {
  gzFile f = gzopen(path.c_str(), "rb");  // transparently handles plain files and multi-member (bgzf) gzip
  if (!f) throw std::runtime_error("cannot open " + path);
  std::string out;
  char buf[1 << 16];
  int n;
  while ((n = gzread(f, buf, sizeof(buf))) > 0) out.append(buf, (size_t)n);
  gzclose(f);
  if (n < 0) throw std::runtime_error("gzread failed for " + path);
  return out;
}

}  // namespace gz_text
