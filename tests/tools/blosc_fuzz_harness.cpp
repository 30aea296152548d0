// ASan/UBSan harness for the host-side Blosc frame decoder (aliby_ingest_inflate, codec 2): decodes every fixture frame, then
// corrupted copies — truncations, byte flips in the header / offsets / anywhere, and directed header patches — into a buffer of
// exactly the size the ORIGINAL frame declares, so that any write past it is a sanitizer error.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>
#include "aliby_hip.h"
static std::vector<unsigned char> slurp(const std::string& p) { std::ifstream f(p, std::ios::binary); return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), {}); }
static void wr32(std::vector<unsigned char>& b, size_t o, unsigned v) { if (o + 4 <= b.size()) for (int i = 0; i < 4; ++i) b[o + i] = (unsigned char)(v >> (8 * i)); }
static unsigned rd32(const std::vector<unsigned char>& b, size_t o) { return o + 4 <= b.size() ? b[o] | (b[o + 1] << 8) | (b[o + 2] << 16) | ((unsigned)b[o + 3] << 24) : 0; }
static const int DIRECTED = 14;
static void directed(std::vector<unsigned char>& b, int k) {
  switch (k) {
    case 0: wr32(b, 4, 0xFFFFFFFFu); break;            // nbytes huge
    case 1: wr32(b, 4, rd32(b, 4) + 1); break;          // nbytes one more than the room
    case 2: wr32(b, 8, 0); break;                       // blocksize 0
    case 3: wr32(b, 8, 1); break;                       // blocksize 1: a block offset per byte
    case 4: wr32(b, 8, 0xFFFFFFFFu); break;             // blocksize huge
    case 5: wr32(b, 12, 0xFFFFFFFFu); break;            // cbytes past the buffer
    case 6: wr32(b, 12, 16); break;                     // cbytes = header only
    case 7: wr32(b, 16, 0xFFFFFFF0u); break;            // first block offset far away
    case 8: wr32(b, 16, 3); break;                      // first block offset inside the header
    case 9: if (b.size() > 3) b[3] = 0; break;          // typesize 0
    case 10: if (b.size() > 3) b[3] = 255; break;       // typesize 255
    case 11: if (b.size() > 2) b[2] ^= 0x10; break;     // flip the split flag
    case 12: if (b.size() > 2) b[2] ^= 0x07; break;     // flip shuffle / memcpy / bitshuffle
    case 13: { unsigned o = rd32(b, 16); wr32(b, o, 0x7FFFFFFFu); } break;  // first stream length huge
  }
}
int main(int argc, char** argv) {
  const int trials = atoi(argv[1]);
  std::mt19937 rng(4321);
  int decoded = 0, rejected = 0;
  for (int i = 2; i < argc; ++i) {
    const std::vector<unsigned char> orig = slurp(argv[i]);
    if (orig.size() < 16) continue;
    const size_t room = rd32(orig, 4);
    for (int trial = -DIRECTED - 1; trial < trials; ++trial) {
      std::vector<unsigned char> b = orig;
      if (trial == -DIRECTED - 1) { /* the frame as it is */ }
      else if (trial < 0) directed(b, -trial - 1);
      else {
        const int kind = rng() % 4;
        if (kind == 0) b.resize(rng() % b.size());
        else { const int n = 1 + rng() % 6; for (int k = 0; k < n; ++k) b[rng() % (kind == 1 ? std::min<size_t>(b.size(), 16) : kind == 2 ? std::min<size_t>(b.size(), 64) : b.size())] = (unsigned char)rng(); }
      }
      std::vector<unsigned char> out(room ? room : 1);  // exactly the declared size: heap redzones on both sides
      size_t got = 0;
      // (the frame lives in an exactly-sized heap block too, so reads past its end are caught as well)
      std::vector<unsigned char> in(b.begin(), b.end());
      const int rc = aliby_ingest_inflate(2, in.empty() ? (const void*)out.data() : (const void*)in.data(), in.size(), out.data(), room, &got);
      if (rc == 0) ++decoded; else ++rejected;
      if (trial == -DIRECTED - 1 && rc != 0) { printf("fixture %s did not decode: %s\n", argv[i], aliby_last_error()); return 2; }
      if (rc == 0 && got > room) { printf("decoder reported %zu bytes for a %zu-byte buffer\n", got, room); return 3; }
    }
  }
  printf("no memory error: %d decoded, %d rejected\n", decoded, rejected);
  return 0;
}
