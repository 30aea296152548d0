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
// ---- directed mutations: patch one field of the first IFD (classic or BigTIFF, either byte order) -----------------
struct Ifd {
  std::vector<unsigned char>& b; bool be, big; size_t first, n;
  explicit Ifd(std::vector<unsigned char>& bytes) : b(bytes), be(false), big(false), first(0), n(0) {
    if (b.size() < 16) return;
    be = b[0] == 'M';
    big = rd(2, 2) == 43;
    size_t off = big ? (size_t)rd(8, 8) : (size_t)rd(4, 4);
    if (off + (big ? 8 : 2) > b.size()) return;
    n = (size_t)rd(off, big ? 8 : 2);
    first = off + (big ? 8 : 2);
    if (n > 4096 || first + n * (big ? 20 : 12) > b.size()) n = 0;
  }
  unsigned long long rd(size_t o, int w) const { unsigned long long v = 0; for (int i = 0; i < w; ++i) v |= (unsigned long long)b[o + (be ? w - 1 - i : i)] << (8 * i); return v; }
  void wr(size_t o, int w, unsigned long long v) { for (int i = 0; i < w; ++i) b[o + (be ? w - 1 - i : i)] = (unsigned char)(v >> (8 * i)); }
  size_t entry(int tag) const { for (size_t e = 0; e < n; ++e) { size_t eo = first + e * (big ? 20 : 12); if ((int)rd(eo, 2) == tag) return eo; } return 0; }
  // set tag's type / count / inline value (type 0 or count ~0ull = keep)
  bool patch(int tag, int type, unsigned long long count, unsigned long long value, bool set_value = true) {
    size_t eo = entry(tag); if (!eo) return false;
    if (type) wr(eo + 2, 2, (unsigned long long)type);
    if (count != ~0ull) wr(eo + 4, big ? 8 : 4, count);
    if (set_value) { int ty = type ? type : (int)rd(eo + 2, 2); int w = ty == 3 ? 2 : (ty == 16 ? 8 : 4); size_t vo = eo + (big ? 12 : 8); wr(vo, big ? 8 : 4, 0); wr(vo, w, value); }
    return true;
  }
};
static int directed_cases() { return 16; }
static bool directed(std::vector<unsigned char>& b, int k) {
  Ifd d(b);
  if (!d.n) return false;
  switch (k) {
    case 0: return d.patch(277, 0, ~0ull, 0);                       // SamplesPerPixel = 0
    case 1: return d.patch(277, 0, ~0ull, 0xFFFF);                  // SamplesPerPixel = 65535 (-1 as a short)
    case 2: return d.patch(277, 4, ~0ull, 0xFFFFFFFFull);           // SamplesPerPixel = -1 as a LONG
    case 3: return d.patch(322, 0, ~0ull, 0) | d.patch(323, 0, ~0ull, 0);                        // tile size 0
    case 4: return d.patch(322, 4, ~0ull, 0x40000000ull) | d.patch(323, 4, ~0ull, 0x40000000ull);  // huge tiles
    case 5: return d.patch(322, 4, ~0ull, 0xFFFFFFFFull);           // tile width 2^32-1
    case 6: return d.big && d.patch(256, 16, ~0ull, (1ull << 32) + d.rd(d.entry(256) + 12, 4));   // LONG8 width = 2^32 + W
    case 7: return d.big && d.patch(257, 16, ~0ull, (1ull << 32) + d.rd(d.entry(257) + 12, 4));   // LONG8 height
    case 8: return d.big && (d.patch(273, 16, 1ull << 61, 0, false) | d.patch(324, 16, 1ull << 61, 0, false));  // count * 8 wraps
    case 9: return d.big && (d.patch(279, 16, (1ull << 61) + 1, 0, false) | d.patch(325, 16, (1ull << 61) + 1, 0, false));
    case 10: return d.patch(273, 0, 0x7FFFFFFFull, 0, false) | d.patch(324, 0, 0x7FFFFFFFull, 0, false);        // count far past the file
    case 11: return d.patch(258, 0, ~0ull, 0);                      // BitsPerSample = 0
    case 12: return d.patch(258, 0, ~0ull, 0xFFF8);                 // BitsPerSample = 65528
    case 13: return d.patch(278, 4, ~0ull, 0xFFFFFFFFull);          // RowsPerStrip huge
    case 14: return d.patch(284, 0, ~0ull, 7);                      // PlanarConfiguration nonsense
    case 15: return d.patch(256, 0, ~0ull, 0) | d.patch(257, 0, ~0ull, 0);  // zero-sized image
  }
  return false;
}

int main(int argc, char** argv) {
  // usage: harness <scratch file> <trials per fixture> <fixture.tif>...
  const std::string scratch = argv[1];
  const int trials = atoi(argv[2]);
  std::mt19937 rng(12345);
  int decoded = 0, rejected = 0;
  for (int i = 3; i < argc; ++i) {
    std::vector<unsigned char> orig = slurp(argv[i]);
    for (int trial = -directed_cases(); trial < trials; ++trial) {
      std::vector<unsigned char> b = orig;
      if (trial < 0) {
        if (!directed(b, -trial - 1)) continue;
      } else if (trial > 0) {
        int kind = rng() % 3;
        if (kind == 0) b.resize(rng() % b.size());
        else { int n = 1 + rng() % 8; for (int k = 0; k < n; ++k) b[rng() % (kind == 1 ? std::min<size_t>(b.size(), 512) : b.size())] = (unsigned char)rng(); }
        if (b.size() < 8) b.resize(8);
      }
      const char* tmp = scratch.c_str();
      { std::ofstream o(tmp, std::ios::binary); o.write((const char*)b.data(), b.size()); }
      int64_t info[12]; char desc[256];
      if (trial < 0) {  // directed: also decode with the geometry the pristine file declares (what a caller holding the first file's shape passes)
        int64_t oi[12];
        if (aliby_tiff_probe(argv[i], oi, desc, sizeof desc) == 0 && oi[1] <= 4096 && oi[2] <= 4096) {
          std::vector<unsigned char> dst0((size_t)oi[1] * oi[2] * (oi[3] / 8));
          const char* paths0[1] = {tmp}; int32_t pg0[1] = {0};
          int rc0 = aliby_ingest_tiff_planes(nullptr, paths0, pg0, 1, (int)oi[1], (int)oi[2], (int)(oi[3] / 8), dst0.data(), dst0.size(), 0, 2, nullptr);
          if (rc0 == 0) ++decoded; else ++rejected;
        }
      }
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
