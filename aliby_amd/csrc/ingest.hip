// ingest.hip — image ingest for the tile stager: TIFF planes decoded by a host thread pool straight into pinned
// staging memory and pushed to HBM plane by plane while the remaining planes are still decoding.
//
// Replaces the per-file `imageio.imread` calls of ImageList.get_data_lazy (src/aliby/io/image.py:395-408), the
// `dask.array.image.imread` of ImageDir / ImageMultiTiff (image.py:190, 285) and the chunk decompression zarr does
// for ImageZarr (image.py:246-259).  Baseline TIFF 6.0 + BigTIFF: strips or tiles, little/big endian, 8/16/32/64-bit
// unsigned / signed / float samples, compression none / LZW / Deflate / PackBits / Zstandard, horizontal predictor.
// Host code only; the one device operation is the asynchronous plane upload.
#include "common.h"
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct MappedFile {
  const uint8_t* p = nullptr;
  size_t n = 0;
  int fd = -1;
  bool open(const char* path, std::string& err) {
    fd = ::open(path, O_RDONLY);
    if (fd < 0) { err = std::string("cannot open ") + path; return false; }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 8) { err = std::string("not a TIFF (too short): ") + path; return false; }
    n = (size_t)st.st_size;
    void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { err = std::string("mmap failed: ") + path; return false; }
    p = (const uint8_t*)m;
    return true;
  }
  ~MappedFile() {
    if (p) munmap((void*)p, n);
    if (fd >= 0) ::close(fd);
  }
};

struct Page {
  uint64_t width = 0, height = 0;
  int bits = 1, spp = 1, compression = 1, predictor = 1, sample_format = 1, planar = 1, subfile = 0;
  uint64_t rows_per_strip = 0, tile_w = 0, tile_h = 0;
  std::vector<uint64_t> offsets, counts;
  bool tiled = false;
  std::string description;
};

struct Tiff {
  MappedFile f;
  bool be = false, big = false;
  uint64_t first_ifd = 0;

  uint16_t u16(size_t o) const { const uint8_t* q = f.p + o; return be ? (uint16_t)(q[0] << 8 | q[1]) : (uint16_t)(q[1] << 8 | q[0]); }
  uint32_t u32(size_t o) const {
    const uint8_t* q = f.p + o;
    return be ? ((uint32_t)q[0] << 24 | (uint32_t)q[1] << 16 | (uint32_t)q[2] << 8 | q[3])
              : ((uint32_t)q[3] << 24 | (uint32_t)q[2] << 16 | (uint32_t)q[1] << 8 | q[0]);
  }
  uint64_t u64(size_t o) const { return be ? ((uint64_t)u32(o) << 32 | u32(o + 4)) : ((uint64_t)u32(o + 4) << 32 | u32(o)); }
  bool in(size_t o, size_t len) const { return o <= f.n && len <= f.n - o; }

  bool open(const char* path, std::string& err) {
    if (!f.open(path, err)) return false;
    if (f.p[0] == 'I' && f.p[1] == 'I') be = false;
    else if (f.p[0] == 'M' && f.p[1] == 'M') be = true;
    else { err = std::string("not a TIFF (byte-order mark): ") + path; return false; }
    uint16_t magic = u16(2);
    if (magic == 42) { big = false; first_ifd = u32(4); }
    else if (magic == 43) {
      if (f.n < 16 || u16(4) != 8) { err = "BigTIFF with an offset size other than 8"; return false; }
      big = true; first_ifd = u64(8);
    } else { err = std::string("not a TIFF (magic): ") + path; return false; }
    return true;
  }

  static int type_size(int t) {
    switch (t) {
      case 1: case 2: case 6: case 7: return 1;
      case 3: case 8: return 2;
      case 4: case 9: case 11: case 13: return 4;
      case 5: case 10: case 12: case 16: case 17: case 18: return 8;
      default: return 0;
    }
  }

  // one IFD: fills `pg` when non-null; returns the offset of the next IFD through `next`
  bool read_ifd(uint64_t off, Page* pg, uint64_t& next, std::string& err) const {
    const size_t cnt_sz = big ? 8 : 2, ent_sz = big ? 20 : 12, off_sz = big ? 8 : 4;
    if (!in(off, cnt_sz)) { err = "IFD offset outside the file"; return false; }
    uint64_t n = big ? u64(off) : u16(off);
    if (n > 65535 || !in(off + cnt_sz, n * ent_sz + off_sz)) { err = "IFD runs past the end of the file"; return false; }
    next = big ? u64(off + cnt_sz + n * ent_sz) : u32(off + cnt_sz + n * ent_sz);
    if (!pg) return true;
    for (uint64_t e = 0; e < n; ++e) {
      size_t eo = off + cnt_sz + e * ent_sz;
      int tag = u16(eo), type = u16(eo + 2);
      uint64_t count = big ? u64(eo + 4) : u32(eo + 4);
      int ts = type_size(type);
      if (ts == 0) continue;
      size_t vo = eo + (big ? 12 : 8);
      if (count > (uint64_t)off_sz / ts) {  // out-of-line values: the multiply below cannot wrap once count <= n / ts
        if (count > f.n / ts) { err = "tag data outside the file"; return false; }
        vo = big ? u64(vo) : u32(vo);
      }
      if (!in(vo, count * ts)) { err = "tag data outside the file"; return false; }
      if (count == 0) continue;
      auto val = [&](uint64_t i) -> uint64_t {
        size_t o = vo + i * ts;
        switch (ts) { case 1: return f.p[o]; case 2: return u16(o); case 4: return u32(o); default: return u64(o); }
      };
      switch (tag) {
        case 254: pg->subfile = (int)val(0); break;
        case 256: pg->width = val(0); break;
        case 257: pg->height = val(0); break;
        case 258:
          pg->bits = (int)val(0);
          for (uint64_t i = 1; i < count; ++i)
            if ((int)val(i) != pg->bits) { err = "samples with differing bit depths"; return false; }
          break;
        case 259: pg->compression = (int)val(0); break;
        case 270: pg->description.assign((const char*)f.p + vo, strnlen((const char*)f.p + vo, count)); break;
        case 273: case 324: pg->offsets.resize(count); for (uint64_t i = 0; i < count; ++i) pg->offsets[i] = val(i); if (tag == 324) pg->tiled = true; break;
        case 277: pg->spp = (int)val(0); break;
        case 278: pg->rows_per_strip = val(0); break;
        case 279: case 325: pg->counts.resize(count); for (uint64_t i = 0; i < count; ++i) pg->counts[i] = val(i); break;
        case 284: pg->planar = (int)val(0); break;
        case 317: pg->predictor = (int)val(0); break;
        case 322: pg->tile_w = val(0); break;
        case 323: pg->tile_h = val(0); break;
        case 339: pg->sample_format = (int)val(0); break;
        default: break;
      }
    }
    if (pg->rows_per_strip == 0 || pg->rows_per_strip > pg->height) pg->rows_per_strip = pg->height;
    return true;
  }

  // Field ranges every later size computation relies on (each product below stays far inside 64 bits).
  static bool validate(const Page& pg, std::string& err) {
    const uint64_t DIM_MAX = 0x7fffffffull, BLOCK_MAX = 1ull << 33;
    if (pg.width < 1 || pg.height < 1 || pg.width > DIM_MAX || pg.height > DIM_MAX) { err = "image width / height out of range"; return false; }
    if (pg.spp < 1 || pg.spp > 64) { err = "SamplesPerPixel out of range: " + std::to_string(pg.spp); return false; }
    if (pg.bits % 8 != 0 || pg.bits < 8 || pg.bits > 64) { err = "only 8/16/32/64-bit samples are supported, got " + std::to_string(pg.bits); return false; }
    if (pg.planar != 1 && pg.planar != 2) { err = "PlanarConfiguration out of range"; return false; }
    const uint64_t px = (uint64_t)pg.spp * (uint64_t)(pg.bits / 8);  // <= 512
    if (pg.tiled) {
      if (pg.tile_w < 1 || pg.tile_h < 1 || pg.tile_w > DIM_MAX || pg.tile_h > DIM_MAX) { err = "tile size out of range"; return false; }
      if (pg.tile_w * pg.tile_h > BLOCK_MAX / px) { err = "tile larger than 8 GiB"; return false; }
    } else if (pg.rows_per_strip * pg.width > BLOCK_MAX / px) { err = "strip larger than 8 GiB"; return false; }
    return true;
  }

  // full-resolution pages only (NewSubfileType bit 0 marks a reduced copy)
  bool page(int index, Page& pg, std::string& err) const {
    uint64_t off = first_ifd;
    int seen = 0;
    for (int guard = 0; off != 0 && guard < (1 << 24); ++guard) {
      Page cur;
      uint64_t next;
      if (!read_ifd(off, &cur, next, err)) return false;
      if (!(cur.subfile & 1)) {
        if (seen == index) {
          if (!validate(cur, err)) return false;
          pg = std::move(cur);
          return true;
        }
        ++seen;
      }
      off = next;
    }
    err = "page index beyond the last page";
    return false;
  }
};

// ---- decompressors --------------------------------------------------------------------------------------------

bool unpack_bits(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t& out) {
  size_t i = 0;
  out = 0;
  while (i < n && out < cap) {
    int8_t c = (int8_t)s[i++];
    if (c >= 0) {
      size_t len = (size_t)c + 1;
      if (i + len > n) len = n - i;
      if (out + len > cap) len = cap - out;
      memcpy(d + out, s + i, len);
      i += (size_t)c + 1;
      out += len;
    } else if (c != -128) {
      if (i >= n) break;
      size_t len = (size_t)(1 - c);
      if (out + len > cap) len = cap - out;
      memset(d + out, s[i++], len);
      out += len;
    }
  }
  return true;
}

// TIFF-flavoured LZW: MSB-first codes from 9 bits, clear = 256, end = 257, width grows one code early
bool unlzw(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t& out) {
  struct Entry { uint16_t prefix; uint8_t last; uint8_t first; uint32_t len; };
  static thread_local std::vector<Entry> table(4096);
  for (int i = 0; i < 256; ++i) table[i] = {0xFFFF, (uint8_t)i, (uint8_t)i, 1};
  out = 0;
  uint64_t acc = 0;
  int nbits = 0, width = 9, next = 258, prev = -1;
  size_t i = 0;
  for (;;) {
    while (nbits < width && i < n) { acc = (acc << 8) | s[i++]; nbits += 8; }
    if (nbits < width) break;
    int code = (int)((acc >> (nbits - width)) & ((1u << width) - 1));
    nbits -= width;
    if (code == 257) break;
    if (code == 256) { width = 9; next = 258; prev = -1; continue; }
    if (prev < 0) {
      if (code > 255) return false;
      if (out < cap) d[out] = (uint8_t)code;
      ++out;
      prev = code;
      continue;
    }
    int emit;
    if (code < next) {
      emit = code;
      if (next < 4096) table[next] = {(uint16_t)prev, table[code].first, table[prev].first, table[prev].len + 1};
    } else if (code == next && next < 4096) {
      table[next] = {(uint16_t)prev, table[prev].first, table[prev].first, table[prev].len + 1};
      emit = next;
    } else {
      return false;
    }
    uint32_t len = table[emit].len;
    size_t end = out + len;
    int c = emit;
    for (size_t pos = end; pos-- > out;) {
      if (pos < cap) d[pos] = table[c].last;
      c = table[c].prefix;
    }
    out = end;
    if (next < 4096) ++next;
    if (next == 511 || next == 1023 || next == 2047) ++width;
    prev = code;
    if (out >= cap) break;
  }
  if (out > cap) out = cap;
  return true;
}

typedef size_t (*zstd_decompress_fn)(void*, size_t, const void*, size_t);
typedef unsigned (*zstd_iserror_fn)(size_t);
bool unzstd(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t& out, std::string& err) {
  static zstd_decompress_fn dec = nullptr;
  static zstd_iserror_fn iserr = nullptr;
  static std::atomic<int> state{0};  // 0 = untried, 1 = ready, 2 = absent
  if (state.load() == 0) {
    void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (h) {
      dec = (zstd_decompress_fn)dlsym(h, "ZSTD_decompress");
      iserr = (zstd_iserror_fn)dlsym(h, "ZSTD_isError");
    }
    state.store(dec && iserr ? 1 : 2);
  }
  if (state.load() != 1) { err = "Zstandard data but libzstd.so.1 is not on this machine"; return false; }
  size_t r = dec(d, cap, s, n);
  if (iserr(r)) { err = "Zstandard stream is corrupt"; return false; }
  out = r;
  return true;
}

bool inflate_any(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t& out, std::string& err) {
  z_stream z;
  memset(&z, 0, sizeof z);
  if (inflateInit2(&z, 15 + 32) != Z_OK) { err = "zlib initialisation failed"; return false; }  // zlib or gzip wrapper
  z.next_in = (Bytef*)s;
  z.avail_in = (uInt)n;
  z.next_out = d;
  z.avail_out = (uInt)cap;
  int r = inflate(&z, Z_FINISH);
  out = z.total_out;
  inflateEnd(&z);
  if (r != Z_STREAM_END && !(r == Z_BUF_ERROR && out == cap) && r != Z_OK) { err = "Deflate stream is corrupt"; return false; }
  return true;
}

// ------------------------------------------------------------------------------------------------ Blosc (version 1 frames)
// zarr v2's default compressor is Blosc(cname="lz4", clevel=5, shuffle=SHUFFLE) (numcodecs), so a "Zarr-backed" store
// (reference: ImageZarr, src/aliby/io/image.py:236-264) is normally a directory of Blosc frames.  c-blosc is not in the image;
// the frame format is small (c-blosc README_HEADER.rst, blosc.c blosc_d):
//   header, 16 bytes: version, versionlz, flags, typesize, nbytes u32, blocksize u32, cbytes u32 (little endian)
//     flags: 0x1 byte shuffle, 0x2 memcpyed (raw bytes follow the header), 0x4 bit shuffle, 0x10 blocks are not split,
//            bits 5-7 the codec: 0 blosclz, 1 lz4 / lz4hc, 2 snappy, 3 zlib, 4 zstd
//   then one int32 offset per block (from the frame start); a block is `nsplits` streams, each int32 length + data, a stream
//   whose length equals its uncompressed size is stored raw; nsplits = typesize for full blocks of >= 128 elements of <= 16
//   bytes unless 0x10 is set, else 1; the decoded block is un-shuffled per block (byte planes; or bit planes when the block holds a
//   multiple of 8 elements, else it was stored unshuffled; bytes past the last whole element are copied as they are).
// LZ4 blocks and blosclz are decoded here by hand; zlib and zstd streams go through the decoders above.
bool unlz4_block(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t& out) {
  size_t ip = 0, op = 0;
  while (ip < n) {
    const unsigned token = s[ip++];
    size_t lit = token >> 4;
    if (lit == 15) {
      unsigned b;
      do { if (ip >= n) return false; b = s[ip++]; lit += b; } while (b == 255);
    }
    if (lit > n - ip || lit > cap - op) return false;
    memcpy(d + op, s + ip, lit);
    ip += lit;
    op += lit;
    if (ip >= n) break;  // the last sequence is literals only
    if (n - ip < 2) return false;
    const size_t off = (size_t)s[ip] | ((size_t)s[ip + 1] << 8);
    ip += 2;
    if (off == 0 || off > op) return false;
    size_t len = (token & 15u);
    if (len == 15) {
      unsigned b;
      do { if (ip >= n) return false; b = s[ip++]; len += b; } while (b == 255);
    }
    len += 4;
    if (len > cap - op) return false;
    for (size_t k = 0; k < len; ++k) d[op + k] = d[op + k - off];  // (overlapping copies repeat the pattern)
    op += len;
  }
  out = op;
  return true;
}

bool unblosclz(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t& out) {
  if (n == 0) return false;
  size_t ip = 0, op = 0;
  unsigned ctrl = s[ip++] & 31u;
  for (;;) {
    if (ctrl >= 32) {
      size_t len = (ctrl >> 5) - 1, ofs = (size_t)(ctrl & 31u) << 8;
      if (len == 7 - 1) {
        unsigned code;
        do { if (ip >= n) return false; code = s[ip++]; len += code; } while (code == 255);
      }
      if (ip >= n) return false;
      const unsigned code = s[ip++];
      size_t dist = ofs + code + 1;  // ref = op - ofs - code, the copy starts at ref - 1
      if (code == 255 && ofs == (31u << 8)) {  // far match: 16-bit distance beyond the 13-bit window
        if (n - ip < 2) return false;
        const size_t far = ((size_t)s[ip] << 8) | s[ip + 1];
        ip += 2;
        dist = far + 8191 + 1;
      }
      len += 3;
      if (dist > op || len > cap - op) return false;
      for (size_t k = 0; k < len; ++k) d[op + k] = d[op + k - dist];
      op += len;
    } else {
      const size_t run = ctrl + 1;
      if (run > n - ip || run > cap - op) return false;
      memcpy(d + op, s + ip, run);
      ip += run;
      op += run;
    }
    if (ip >= n) break;
    ctrl = s[ip++];
  }
  out = op;
  return true;
}

void blosc_unshuffle(const uint8_t* src, uint8_t* dst, size_t n, size_t ts) {  // byte planes -> elements
  const size_t ne = n / ts;
  for (size_t j = 0; j < ts; ++j) {
    const uint8_t* plane = src + j * ne;
    for (size_t i = 0; i < ne; ++i) dst[i * ts + j] = plane[i];
  }
  memcpy(dst + ne * ts, src + ne * ts, n - ne * ts);
}

void blosc_bitunshuffle(const uint8_t* src, uint8_t* dst, size_t n, size_t ts) {  // bit planes -> elements
  // (c-blosc shuffle.c, bitshuffle(): a block whose element count is not a multiple of 8 is stored as it is)
  if ((n / ts) % 8 != 0) { memcpy(dst, src, n); return; }
  const size_t ne = n / ts, rows = ne / 8;  // bit plane (byte j, bit b) is `rows` bytes: bit k of byte r = element 8 r + k
  memset(dst, 0, ne * ts);
  for (size_t j = 0; j < ts; ++j)
    for (unsigned b = 0; b < 8; ++b) {
      const uint8_t* plane = src + (j * 8 + b) * rows;
      for (size_t r = 0; r < rows; ++r) {
        const unsigned byte = plane[r];
        for (unsigned k = 0; k < 8; ++k) dst[(8 * r + k) * ts + j] |= (uint8_t)(((byte >> k) & 1u) << b);
      }
    }
  memcpy(dst + ne * ts, src + ne * ts, n - ne * ts);
}

inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

bool unblosc(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t& out, std::string& err) {
  if (n < 16) { err = "Blosc frame shorter than its header"; return false; }
  const unsigned version = s[0], flags = s[2], ts = s[3] ? s[3] : 1;
  const size_t nbytes = rd32(s + 4), blocksize = rd32(s + 8), cbytes = rd32(s + 12);
  if (version != 2) { err = "Blosc frame version " + std::to_string(version) + " is not supported (2 is)"; return false; }
  if (cbytes > n || cbytes < 16) { err = "Blosc frame is truncated"; return false; }
  if (nbytes > cap) { err = "Blosc frame holds " + std::to_string(nbytes) + " bytes, the chunk has room for " + std::to_string(cap); return false; }
  out = nbytes;
  if (nbytes == 0) return true;
  if (flags & 0x2) {  // memcpyed
    if (cbytes < 16 + nbytes) { err = "Blosc frame is truncated"; return false; }
    memcpy(d, s + 16, nbytes);
    return true;
  }
  if (blocksize == 0 || blocksize > (1u << 31)) { err = "Blosc frame has an invalid block size"; return false; }
  const unsigned codec = flags >> 5;
  const bool shuffle = flags & 0x1, bitshuffle = flags & 0x4, dont_split = flags & 0x10;
  const size_t nblocks = (nbytes + blocksize - 1) / blocksize;
  if (nblocks > (cbytes - 16) / 4) { err = "Blosc frame is truncated (block offsets)"; return false; }
  std::vector<uint8_t> tmp((shuffle || bitshuffle) ? (blocksize < nbytes ? blocksize : nbytes) : 0);
  for (size_t b = 0; b < nblocks; ++b) {
    const size_t start = rd32(s + 16 + 4 * b);
    const size_t bsize = b + 1 < nblocks ? blocksize : nbytes - b * blocksize;
    const bool leftover = bsize != blocksize;
    const size_t nsplits = (!dont_split && !leftover && ts <= 16 && blocksize / ts >= 128) ? ts : 1;
    const size_t neblock = bsize / nsplits;
    uint8_t* target = (shuffle || bitshuffle) ? tmp.data() : d + b * blocksize;
    size_t ip = start;
    for (size_t k = 0; k < nsplits; ++k) {
      if (ip > cbytes || cbytes - ip < 4) { err = "Blosc block runs past the frame"; return false; }
      const size_t clen = rd32(s + ip);
      ip += 4;
      if (clen > cbytes - ip) { err = "Blosc stream runs past the frame"; return false; }
      uint8_t* dst = target + k * neblock;
      if (clen == neblock) {
        memcpy(dst, s + ip, neblock);
      } else {
        size_t got = 0;
        bool ok;
        switch (codec) {
          case 0: ok = unblosclz(s + ip, clen, dst, neblock, got); break;
          case 1: ok = unlz4_block(s + ip, clen, dst, neblock, got); break;
          case 3: ok = inflate_any(s + ip, clen, dst, neblock, got, err); break;
          case 4: ok = unzstd(s + ip, clen, dst, neblock, got, err); break;
          default: err = "Blosc codec " + std::to_string(codec) + " is not supported (blosclz, lz4, zlib, zstd are)"; return false;
        }
        if (!ok || got != neblock) { if (err.empty()) err = "Blosc stream is corrupt"; return false; }
      }
      ip += clen;
    }
    if (bitshuffle) blosc_bitunshuffle(tmp.data(), d + b * blocksize, bsize, ts);
    else if (shuffle) blosc_unshuffle(tmp.data(), d + b * blocksize, bsize, ts);
  }
  return true;
}

bool decompress(int scheme, const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t& out, std::string& err) {
  switch (scheme) {
    case 1:
      out = n < cap ? n : cap;
      memcpy(d, s, out);
      return true;
    case 5:
      if (!unlzw(s, n, d, cap, out)) { err = "LZW stream is corrupt"; return false; }
      return true;
    case 8: case 32946: return inflate_any(s, n, d, cap, out, err);
    case 32773: return unpack_bits(s, n, d, cap, out);
    case 50000: return unzstd(s, n, d, cap, out, err);
    default:
      err = "TIFF compression scheme " + std::to_string(scheme) + " is not supported (none, LZW, Deflate, PackBits, Zstandard are)";
      return false;
  }
}

template <typename T>
void undo_predictor_t(T* row, size_t width, int spp) {
  for (size_t x = (size_t)spp; x < width * spp; ++x) row[x] = (T)(row[x] + row[x - spp]);
}

void swap_bytes(uint8_t* p, size_t count, int bytes) {
  if (bytes == 2) for (size_t i = 0; i < count; ++i) { uint8_t t = p[2 * i]; p[2 * i] = p[2 * i + 1]; p[2 * i + 1] = t; }
  else if (bytes == 4) for (size_t i = 0; i < count; ++i) { uint8_t* q = p + 4 * i; uint8_t a = q[0], b = q[1]; q[0] = q[3]; q[1] = q[2]; q[2] = b; q[3] = a; }
  else if (bytes == 8) for (size_t i = 0; i < count; ++i) { uint8_t* q = p + 8 * i; for (int k = 0; k < 4; ++k) { uint8_t t = q[k]; q[k] = q[7 - k]; q[7 - k] = t; } }
}

// One decoded block (strip or tile) of `rows` x `cols` pixels -> destination rows; keeps sample 0 of chunky pixels.
void finish_block(const Tiff& t, const Page& pg, uint8_t* blk, size_t rows, size_t cols, uint8_t* dst, size_t dst_pitch,
                  size_t copy_cols, size_t copy_rows) {
  const int bps = pg.bits / 8, spp = pg.planar == 1 ? pg.spp : 1;
  const size_t row_bytes = cols * spp * bps;
  for (size_t r = 0; r < copy_rows; ++r) {
    uint8_t* row = blk + r * row_bytes;
    if (t.be && bps > 1) swap_bytes(row, cols * spp, bps);
    if (pg.predictor == 2) {
      if (bps == 1) undo_predictor_t((uint8_t*)row, cols, spp);
      else if (bps == 2) undo_predictor_t((uint16_t*)row, cols, spp);
      else if (bps == 4) undo_predictor_t((uint32_t*)row, cols, spp);
      else undo_predictor_t((uint64_t*)row, cols, spp);
    }
    uint8_t* out = dst + r * dst_pitch;
    if (spp == 1) memcpy(out, row, copy_cols * bps);
    else for (size_t x = 0; x < copy_cols; ++x) memcpy(out + x * bps, row + x * spp * bps, bps);
  }
  (void)rows;
}

// Number of independently decodable blocks (strips, or rows of tiles) of a page.
size_t page_blocks(const Page& pg) {
  if (pg.tiled) return pg.tile_h ? (pg.height + pg.tile_h - 1) / pg.tile_h : 0;
  return pg.rows_per_strip ? (pg.height + pg.rows_per_strip - 1) / pg.rows_per_strip : 0;
}

// Decodes blocks [b0, b1) of the page into the full-plane destination.
bool decode_page(const Tiff& t, const Page& pg, uint8_t* dst, std::string& err, size_t b0, size_t b1) {
  if (!Tiff::validate(pg, err)) return false;
  if (pg.bits % 8 != 0 || pg.bits < 8 || pg.bits > 64) { err = "only 8/16/32/64-bit samples are supported, got " + std::to_string(pg.bits); return false; }
  if (pg.predictor != 1 && pg.predictor != 2) { err = "floating-point predictor (3) is not supported"; return false; }
  if (pg.offsets.empty() || pg.offsets.size() != pg.counts.size()) { err = "strip/tile offsets and byte counts disagree"; return false; }
  const int bps = pg.bits / 8, spp = pg.planar == 1 ? pg.spp : 1;
  const size_t W = pg.width, H = pg.height, dst_pitch = W * bps;
  std::vector<uint8_t> blk;
  if (pg.tiled) {
    if (!pg.tile_w || !pg.tile_h) { err = "tiled page without tile size"; return false; }
    const size_t tx = (W + pg.tile_w - 1) / pg.tile_w, ty = (H + pg.tile_h - 1) / pg.tile_h;
    if (pg.offsets.size() < tx * ty) { err = "fewer tiles than the image needs"; return false; }
    const size_t cap = pg.tile_w * pg.tile_h * spp * bps;
    blk.resize(cap);
    for (size_t j = b0; j < std::min(b1, ty); ++j)
      for (size_t i = 0; i < tx; ++i) {
        size_t k = j * tx + i, got = 0;
        if (!t.in(pg.offsets[k], pg.counts[k])) { err = "tile data outside the file"; return false; }
        if (!decompress(pg.compression, t.f.p + pg.offsets[k], pg.counts[k], blk.data(), cap, got, err)) return false;
        if (got < cap) memset(blk.data() + got, 0, cap - got);
        size_t cols = std::min<size_t>(pg.tile_w, W - i * pg.tile_w), rows = std::min<size_t>(pg.tile_h, H - j * pg.tile_h);
        finish_block(t, pg, blk.data(), pg.tile_h, pg.tile_w, dst + (j * pg.tile_h) * dst_pitch + i * pg.tile_w * bps, dst_pitch, cols, rows);
      }
    return true;
  }
  const size_t rps = pg.rows_per_strip, strips = (H + rps - 1) / rps;
  if (pg.offsets.size() < strips) { err = "fewer strips than the image needs"; return false; }
  const bool direct = pg.compression == 1 && spp == 1 && pg.predictor == 1 && !(t.be && bps > 1);
  if (!direct) blk.resize(rps * W * spp * bps);
  for (size_t s = b0; s < std::min(b1, strips); ++s) {
    const size_t rows = std::min(rps, H - s * rps), want = rows * W * spp * bps;
    if (!t.in(pg.offsets[s], pg.counts[s])) { err = "strip data outside the file"; return false; }
    uint8_t* out = dst + s * rps * dst_pitch;
    if (direct) {
      size_t n = std::min<size_t>(want, pg.counts[s]);
      memcpy(out, t.f.p + pg.offsets[s], n);
      if (n < want) memset(out + n, 0, want - n);
      continue;
    }
    size_t got = 0;
    if (!decompress(pg.compression, t.f.p + pg.offsets[s], pg.counts[s], blk.data(), want, got, err)) return false;
    if (got < want) memset(blk.data() + got, 0, want - got);
    finish_block(t, pg, blk.data(), rows, W, out, dst_pitch, W, rows);
  }
  return true;
}

// numpy-style type code of a page: 'u', 'i' or 'f'
char kind_of(const Page& pg) { return pg.sample_format == 2 ? 'i' : (pg.sample_format == 3 ? 'f' : 'u'); }

struct PinnedStage {
  void* p = nullptr;
  size_t bytes = 0;
};

}  // namespace

// info[0..11] = pages, width, height, bits, sample_format (1 uint, 2 int, 3 float), samples_per_pixel, compression,
// predictor, tiled, bigtiff, big_endian, uniform (every full-resolution page has page 0's geometry and sample type)
int aliby_tiff_probe(const char* path, int64_t* info, char* description, int description_len) {
  ARG_CHECK(path && info, "path and info must be given");
  try {
  std::string err;
  Tiff t;
  if (!t.open(path, err)) { aliby_set_error("%s", err.c_str()); return ALIBY_ERR_INVALID; }
  Page first;
  if (!t.page(0, first, err)) { aliby_set_error("%s: %s", path, err.c_str()); return ALIBY_ERR_INVALID; }
  int pages = 0;
  bool uniform = true;
  uint64_t off = t.first_ifd;
  for (int guard = 0; off != 0 && guard < (1 << 24); ++guard) {
    Page cur;
    uint64_t next;
    if (!t.read_ifd(off, &cur, next, err)) { aliby_set_error("%s: %s", path, err.c_str()); return ALIBY_ERR_INVALID; }
    if (!(cur.subfile & 1)) {
      ++pages;
      if (cur.width != first.width || cur.height != first.height || cur.bits != first.bits || cur.spp != first.spp ||
          cur.sample_format != first.sample_format)
        uniform = false;
    }
    off = next;
  }
  int64_t v[12] = {pages, (int64_t)first.width, (int64_t)first.height, first.bits, first.sample_format, first.spp,
                   first.compression, first.predictor, first.tiled, t.big, t.be, uniform};
  memcpy(info, v, sizeof v);
  if (description && description_len > 0) {
    snprintf(description, (size_t)description_len, "%s", first.description.c_str());
  }
  return ALIBY_OK;
  } catch (const std::bad_alloc&) { aliby_set_error("%s: out of memory while reading the directory", path); return ALIBY_ERR_TOO_LARGE;
  } catch (const std::exception& e) { aliby_set_error("%s: %s", path, e.what()); return ALIBY_ERR_INVALID; }
}

// Decode n planes (page `pages[i]` of `paths[i]`, sample 0) of width x height x bytes_per_sample into
// dst + i * plane_stride.  dst_is_device = 0: dst is host memory (pinned or not) and the call returns when every plane
// is decoded.  dst_is_device = 1: planes are decoded into the library's pinned staging block and each one is queued
// for upload on `stream` the moment its decoder thread finishes, so PCIe transfers overlap the remaining decodes; the
// call returns after the last upload has completed.
static int ingest_tiff_planes_impl(aliby_ctx* ctx, const char* const* paths, const int32_t* pages, int n, int width, int height,
                                   int bytes_per_sample, void* dst, size_t plane_stride, int dst_is_device, int n_threads,
                                   void* stream) {
  ARG_CHECK(n >= 0 && width > 0 && height > 0, "plane geometry");
  ARG_CHECK(bytes_per_sample == 1 || bytes_per_sample == 2 || bytes_per_sample == 4 || bytes_per_sample == 8, "bytes_per_sample");
  if (n == 0) return ALIBY_OK;
  ARG_CHECK(paths && dst, "paths and dst must be given");
  const size_t plane_bytes = (size_t)width * height * bytes_per_sample;
  ARG_CHECK(plane_stride >= plane_bytes, "plane_stride smaller than one plane");
  static PinnedStage stage;  // shared staging block: device ingests of one process take turns
  static std::mutex stage_lock;
  std::unique_lock<std::mutex> turn(stage_lock, std::defer_lock);
  if (dst_is_device) turn.lock();
  uint8_t* host = (uint8_t*)dst;
  size_t host_stride = plane_stride;
  int device = 0;
  if (dst_is_device) {
    ARG_CHECK(ctx, "a context is needed to upload");
    device = ctx->device;
    HIP_TRY(hipSetDevice(device));  // the caller may be a helper thread that never selected the context's GPU
    if (stage.bytes < plane_bytes * n) {
      if (stage.p) HIP_TRY(hipHostFree(stage.p));
      stage = PinnedStage();
      HIP_TRY(hipHostMalloc(&stage.p, plane_bytes * n, hipHostMallocDefault));
      stage.bytes = plane_bytes * n;
    }
    host = (uint8_t*)stage.p;
    host_stride = plane_bytes;
  }
  if (n_threads <= 0) n_threads = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
  std::string first_error;
  hipError_t first_hip = hipSuccess;
  // Planes are opened in batches (bounded number of mappings) and every plane is cut into parts of about 512 KB of
  // pixels — whole strips or rows of tiles — so that one FOV of five large Deflate planes still feeds every thread.
  // The thread that finishes the last part of a plane queues its upload.
  struct PlaneState {
    Tiff t;
    Page pg;
    size_t blocks = 0, parts = 0;
    std::atomic<int> remaining{0};
  };
  const int BATCH = 128;
  for (int base = 0; base < n && first_error.empty() && first_hip == hipSuccess; base += BATCH) {
    const int m = std::min(BATCH, n - base);
    std::vector<std::unique_ptr<PlaneState>> planes(m);
    struct Part { int plane; size_t b0, b1; };
    std::vector<Part> work;
    for (int k = 0; k < m && first_error.empty(); ++k) {
      const int i = base + k;
      planes[k].reset(new PlaneState());
      PlaneState& ps = *planes[k];
      std::string err;
      bool ok = ps.t.open(paths[i], err) && ps.t.page(pages ? pages[i] : 0, ps.pg, err);
      if (ok && (ps.pg.width != (uint64_t)width || ps.pg.height != (uint64_t)height || ps.pg.bits != 8 * bytes_per_sample)) {
        err = "geometry differs from the first file: " + std::to_string(ps.pg.width) + "x" + std::to_string(ps.pg.height) + "x" +
              std::to_string(ps.pg.bits) + " bits";
        ok = false;
      }
      if (ok && ps.pg.planar != 1 && ps.pg.spp > 1) { err = "planar multi-sample pages are not supported"; ok = false; }
      if (!ok) { first_error = std::string(paths[i]) + ": " + err; break; }
      ps.blocks = page_blocks(ps.pg);
      size_t want = std::max<size_t>(1, plane_bytes / (512u << 10));
      ps.parts = std::max<size_t>(1, std::min(ps.blocks, want));
      ps.remaining.store((int)ps.parts);
      for (size_t q = 0; q < ps.parts; ++q) work.push_back({k, ps.blocks * q / ps.parts, ps.blocks * (q + 1) / ps.parts});
    }
    if (!first_error.empty()) break;
    const int threads = std::max(1, std::min<int>(n_threads, (int)work.size()));
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    std::vector<std::string> errors(threads);
    std::vector<hipError_t> hip_errors(threads, hipSuccess);
    auto worker = [&](int tid) {
      try {  // nothing may unwind out of a std::thread (std::terminate) or across the C ABI
      if (dst_is_device) hip_errors[tid] = hipSetDevice(device);
      for (;;) {
        size_t w = next.fetch_add(1);
        if (w >= work.size() || failed.load()) return;
        const Part& part = work[w];
        PlaneState& ps = *planes[part.plane];
        const int i = base + part.plane;
        std::string err;
        if (!decode_page(ps.t, ps.pg, host + (size_t)i * host_stride, err, part.b0, part.b1)) {
          errors[tid] = std::string(paths[i]) + ": " + err;
          failed.store(1);
          return;
        }
        if (ps.remaining.fetch_sub(1) == 1 && dst_is_device && hip_errors[tid] == hipSuccess)
          hip_errors[tid] = hipMemcpyAsync((uint8_t*)dst + (size_t)i * plane_stride, host + (size_t)i * host_stride,
                                           plane_bytes, hipMemcpyHostToDevice, as_stream(stream));
      }
      } catch (const std::exception& e) {
        errors[tid] = std::string("decoder failed: ") + e.what();
        failed.store(1);
      }
    };
    if (threads == 1) worker(0);
    else {
      std::vector<std::thread> pool;
      for (int tid = 0; tid < threads; ++tid) pool.emplace_back(worker, tid);
      for (auto& th : pool) th.join();
    }
    for (int tid = 0; tid < threads; ++tid) {
      if (first_error.empty() && !errors[tid].empty()) first_error = errors[tid];
      if (first_hip == hipSuccess && hip_errors[tid] != hipSuccess) first_hip = hip_errors[tid];
    }
  }
  if (dst_is_device) {
    int rc = aliby_wait_stream(as_stream(stream));  // the staging block is reused by the next call
    if (rc != ALIBY_OK) return rc;
  }
  if (!first_error.empty()) { aliby_set_error("%s", first_error.c_str()); return ALIBY_ERR_INVALID; }
  if (first_hip != hipSuccess) { aliby_set_error("plane upload failed: %s", hipGetErrorString(first_hip)); return ALIBY_ERR_HIP; }
  return ALIBY_OK;
}

int aliby_ingest_tiff_planes(aliby_ctx* ctx, const char* const* paths, const int32_t* pages, int n, int width, int height,
                             int bytes_per_sample, void* dst, size_t plane_stride, int dst_is_device, int n_threads,
                             void* stream) {
  try {
    return ingest_tiff_planes_impl(ctx, paths, pages, n, width, height, bytes_per_sample, dst, plane_stride, dst_is_device,
                                   n_threads, stream);
  } catch (const std::bad_alloc&) {
    aliby_set_error("TIFF ingest: out of memory (implausible sizes in the file's directory?)");
    return ALIBY_ERR_TOO_LARGE;
  } catch (const std::exception& e) {
    aliby_set_error("TIFF ingest: %s", e.what());
    return ALIBY_ERR_INVALID;
  }
}

// Chunk decompression for zarr stores: codec 0 = zlib / gzip, 1 = Zstandard, 2 = a Blosc-1 frame.  *out_bytes = bytes produced.
int aliby_ingest_inflate(int codec, const void* src, size_t src_bytes, void* dst, size_t dst_bytes, size_t* out_bytes) {
  ARG_CHECK(src && dst && out_bytes, "buffers must be given");
  try {
  std::string err;
  size_t got = 0;
  bool ok;
  if (codec == 0) ok = inflate_any((const uint8_t*)src, src_bytes, (uint8_t*)dst, dst_bytes, got, err);
  else if (codec == 1) ok = unzstd((const uint8_t*)src, src_bytes, (uint8_t*)dst, dst_bytes, got, err);
  else if (codec == 2) ok = unblosc((const uint8_t*)src, src_bytes, (uint8_t*)dst, dst_bytes, got, err);
  else { aliby_set_error("unknown codec %d", codec); return ALIBY_ERR_UNSUPPORTED; }
  if (!ok) { aliby_set_error("%s", err.c_str()); return ALIBY_ERR_INVALID; }
  *out_bytes = got;
  return ALIBY_OK;
  } catch (const std::exception& e) { aliby_set_error("inflate: %s", e.what()); return ALIBY_ERR_INVALID; }
}
