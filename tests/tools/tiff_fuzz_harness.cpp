// ASan/UBSan harness for the host-side TIFF decoder: decodes every fixture, then thousands of corrupted copies.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <fstream>
#include <random>
#include "aliby_hip.h"
static std::vector<unsigned char> slurp(const std::string& p) { std::ifstream f(p, std::ios::binary); return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), {}); }
int main(int argc, char** argv) {
  // usage: harness <scratch file> <trials per fixture> <fixture.tif>...
  const std::string scratch = argv[1];
  const int trials = atoi(argv[2]);
  std::mt19937 rng(12345);
  int decoded = 0, rejected = 0;
  for (int i = 3; i < argc; ++i) {
    std::vector<unsigned char> orig = slurp(argv[i]);
    for (int trial = 0; trial < trials; ++trial) {
      std::vector<unsigned char> b = orig;
      if (trial > 0) {
        int kind = rng() % 3;
        if (kind == 0) b.resize(rng() % b.size());
        else { int n = 1 + rng() % 8; for (int k = 0; k < n; ++k) b[rng() % (kind == 1 ? std::min<size_t>(b.size(), 512) : b.size())] = (unsigned char)rng(); }
        if (b.size() < 8) b.resize(8);
      }
      const char* tmp = scratch.c_str();
      { std::ofstream o(tmp, std::ios::binary); o.write((const char*)b.data(), b.size()); }
      int64_t info[12]; char desc[256];
      if (aliby_tiff_probe(tmp, info, desc, sizeof desc) != 0) { ++rejected; continue; }
      long long w = info[1], h = info[2], bits = info[3], pages = info[0];
      if (w <= 0 || h <= 0 || w > 4096 || h > 4096 || (bits != 8 && bits != 16 && bits != 32 && bits != 64) || pages <= 0) { ++rejected; continue; }
      std::vector<unsigned char> dst((size_t)w * h * (bits / 8));
      const char* paths[1] = {tmp}; int32_t pg[1] = {0};
      int rc = aliby_ingest_tiff_planes(nullptr, paths, pg, 1, (int)w, (int)h, (int)(bits / 8), dst.data(), dst.size(), 0, 2, nullptr);
      if (rc == 0) ++decoded; else ++rejected;
    }
  }
  printf("decoded %d, rejected %d, no memory error\n", decoded, rejected);
  return 0;
}
