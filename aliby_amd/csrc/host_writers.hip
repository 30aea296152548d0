// host_writers.hip — the step API's two on-disk outputs, encoded natively (host code only: no kernels in this file).
//
//   profiles/<name>.parquet   pyarrow.parquet.write_table(profiles, compression="zstd")        src/aliby/pipe_core.py:412-413
//   steps/<name>/<step>/<tp:04d>.npz   numpy.savez_compressed(out_file, labels)                   src/aliby/io/write.py:25-51
//
// Once the feature kernels need ~2 ms per FOV these two calls are what a position costs: pyarrow spends ~15 CPU-ms on a
// thousand-column table (a fresh zstd context, an encoder object, statistics and a Python-visible column writer per column
// chunk of 2 KB), numpy + zlib ~5.5 ms on a label image.  Both formats are simple enough to write directly:
//
//   * parquet: one row group, one PLAIN data page (v1) per column chunk, RLE definition levels (fields stay OPTIONAL as pyarrow
//     writes them, every value present), zstd (level chosen by the caller) through ONE reused compression context per thread (libzstd.so.1, dlopen),
//     Thrift compact protocol for the page headers and the footer.  Readers see the same schema (names, order, DOUBLE / INT64 /
//     UINT_16 / STRING) and the same values as from a pyarrow-written file; no statistics are written.
//   * npz: a zip archive of .npy members, deflated with libdeflate (libdeflate.so.0, dlopen) when it is there and zlib
//     otherwise; numpy.load reads it like any savez_compressed file.
//
// Both entry points take plain pointers; the Python side (aliby_amd/io/write.py) calls them from its writer threads with the
// interpreter lock released, so files of different positions are encoded side by side without writer processes.
#include "common.h"
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <unistd.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>
#include <atomic>
#include <mutex>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------------ Thrift compact protocol
struct Thrift {
  std::vector<uint8_t>& b;
  std::vector<int> last;  // last field id per open struct
  explicit Thrift(std::vector<uint8_t>& buf) : b(buf) { last.push_back(0); }
  void varint(uint64_t v) {
    while (v >= 0x80) { b.push_back((uint8_t)(v | 0x80)); v >>= 7; }
    b.push_back((uint8_t)v);
  }
  static uint64_t zz(int64_t v) { return ((uint64_t)v << 1) ^ (uint64_t)(v >> 63); }
  void field(int id, int type) {
    const int delta = id - last.back();
    if (delta > 0 && delta <= 15) b.push_back((uint8_t)((delta << 4) | type));
    else { b.push_back((uint8_t)type); varint(zz(id)); }
    last.back() = id;
  }
  void i32(int id, int32_t v) { field(id, 5); varint(zz(v)); }
  void i64(int id, int64_t v) { field(id, 6); varint(zz(v)); }
  void i8(int id, int8_t v) { field(id, 3); b.push_back((uint8_t)v); }
  void boolean(int id, bool v) { field(id, v ? 1 : 2); }
  void str(int id, const char* s, size_t n) { field(id, 8); varint(n); b.insert(b.end(), s, s + n); }
  void str(int id, const std::string& s) { str(id, s.data(), s.size()); }
  void begin(int id) { field(id, 12); last.push_back(0); }
  void begin_elem() { last.push_back(0); }  // struct as a list element: no field header
  void end() { b.push_back(0); last.pop_back(); }
  void list(int id, int elem_type, size_t n) {
    field(id, 9);
    if (n < 15) b.push_back((uint8_t)((n << 4) | elem_type));
    else { b.push_back((uint8_t)(0xF0 | elem_type)); varint(n); }
  }
  void elem_i32(int32_t v) { varint(zz(v)); }
  void elem_str(const char* s, size_t n) { varint(n); b.insert(b.end(), s, s + n); }
};

// ------------------------------------------------------------------------------------------------ zstd / libdeflate (dlopen)
struct Zstd {
  void* (*createCCtx)() = nullptr;
  size_t (*freeCCtx)(void*) = nullptr;
  size_t (*compressCCtx)(void*, void*, size_t, const void*, size_t, int) = nullptr;
  size_t (*compressBound)(size_t) = nullptr;
  unsigned (*isError)(size_t) = nullptr;
  bool ok = false;
  Zstd() {
    void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    createCCtx = (void* (*)())dlsym(h, "ZSTD_createCCtx");
    freeCCtx = (size_t(*)(void*))dlsym(h, "ZSTD_freeCCtx");
    compressCCtx = (size_t(*)(void*, void*, size_t, const void*, size_t, int))dlsym(h, "ZSTD_compressCCtx");
    compressBound = (size_t(*)(size_t))dlsym(h, "ZSTD_compressBound");
    isError = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
    ok = createCCtx && freeCCtx && compressCCtx && compressBound && isError;
  }
};
const Zstd& zstd() { static Zstd z; return z; }

struct CCtx {  // one compression context per writer thread, alive as long as the thread
  void* p = nullptr;
  ~CCtx() { if (p) zstd().freeCCtx(p); }
};
void* thread_cctx() {
  thread_local CCtx c;
  if (!c.p && zstd().ok) c.p = zstd().createCCtx();
  return c.p;
}

struct Deflate {
  void* (*alloc)(int) = nullptr;
  void (*release)(void*) = nullptr;
  size_t (*compress)(void*, const void*, size_t, void*, size_t) = nullptr;
  size_t (*bound)(void*, size_t) = nullptr;
  uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;
  bool ok = false;
  Deflate() {
    void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    alloc = (void* (*)(int))dlsym(h, "libdeflate_alloc_compressor");
    release = (void (*)(void*))dlsym(h, "libdeflate_free_compressor");
    compress = (size_t(*)(void*, const void*, size_t, void*, size_t))dlsym(h, "libdeflate_deflate_compress");
    bound = (size_t(*)(void*, size_t))dlsym(h, "libdeflate_deflate_compress_bound");
    crc = (uint32_t(*)(uint32_t, const void*, size_t))dlsym(h, "libdeflate_crc32");
    ok = alloc && release && compress && bound && crc;
  }
};
const Deflate& deflater() { static Deflate d; return d; }

// A file is assembled in memory (a buffer the thread keeps from file to file) and handed to the kernel in ONE write: through
// stdio's 4 KiB buffer a 2 MB profiles file is ~500 write calls, and on a journalled file system a dozen writer threads doing
// that side by side ran at 2.5x the thread-time per file of the same code on tmpfs.
struct File {
  int f = -1;  // (tested by the callers: < 0 = not open)
  std::vector<uint8_t>& buf;
  static std::vector<uint8_t>& thread_buffer() { thread_local std::vector<uint8_t> b; return b; }
  explicit File(const char* path) : buf(thread_buffer()) {
    f = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    buf.clear();
  }
  ~File() { if (f >= 0) close(f); }
  bool put(const void* p, size_t n) {
    if (n) buf.insert(buf.end(), static_cast<const uint8_t*>(p), static_cast<const uint8_t*>(p) + n);
    return buf.size() < (16u << 20) || flush();  // (large step outputs go out in 16 MB pieces)
  }
  bool flush() {  // everything so far -> the file
    const uint8_t* p = buf.data();
    size_t left = buf.size();
    while (left) {
      const ssize_t got = write(f, p, left);
      if (got < 0) { if (errno == EINTR) continue; return false; }
      p += got;
      left -= (size_t)got;
    }
    buf.clear();
    return true;
  }
};

// ------------------------------------------------------------------------------------------------ deflate for label images
// A label image is runs: long stretches of background, rows that repeat the row above except at object borders.  A general
// LZ77 matcher (zlib, libdeflate) spends its time hashing 2 MB to rediscover exactly two distances — one item back (the run)
// and one row back — so this encoder tries only those two: greedy longest match of the two candidates (8 bytes per compare),
// literals otherwise, one fixed-Huffman block (RFC 1951 section 3.2.6).  ~7x the speed of zlib level 6 on label images at
// ~1.6x its size; data without that structure compresses poorly here, and the caller then falls back to the general encoder.
struct BitSink {
  std::vector<uint8_t>& out;
  uint64_t acc = 0;
  int nbits = 0;
  explicit BitSink(std::vector<uint8_t>& o) : out(o) {}
  inline void put(uint32_t v, int n) {  // n <= 32 bits, LSB first
    acc |= (uint64_t)v << nbits;
    nbits += n;
    while (nbits >= 8) { out.push_back((uint8_t)acc); acc >>= 8; nbits -= 8; }
  }
  void flush() { if (nbits > 0) { out.push_back((uint8_t)acc); acc = 0; nbits = 0; } }
};

struct FixedHuffman {
  uint16_t lit_code[288];
  uint8_t lit_bits[288];
  uint8_t dist_code[30];
  uint16_t len_sym[259];   // match length -> length symbol
  uint8_t len_extra_bits[259];
  uint16_t len_extra[259];
  static uint32_t rev(uint32_t v, int n) { uint32_t r = 0; for (int i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i); return r; }
  FixedHuffman() {
    for (int s = 0; s < 288; ++s) {
      uint32_t code; int n;
      if (s < 144) { code = 0x30 + s; n = 8; }
      else if (s < 256) { code = 0x190 + (s - 144); n = 9; }
      else if (s < 280) { code = s - 256; n = 7; }
      else { code = 0xC0 + (s - 280); n = 8; }
      lit_code[s] = (uint16_t)rev(code, n);
      lit_bits[s] = (uint8_t)n;
    }
    for (int d = 0; d < 30; ++d) dist_code[d] = (uint8_t)rev((uint32_t)d, 5);
    static const int base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const int ebits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    for (int len = 3; len <= 258; ++len) {
      int k = 28;
      while (base[k] > len) --k;
      if (len == 258) k = 28;
      len_sym[len] = (uint16_t)(257 + k);
      len_extra_bits[len] = (uint8_t)ebits[k];
      len_extra[len] = (uint16_t)(len - base[k]);
    }
  }
};

struct DistCode { int sym, ebits; uint32_t extra; };
inline DistCode dist_code_of(uint32_t d) {  // 1 <= d <= 32768
  static const int base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
  static const int ebits[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
  int k = 29;
  while (base[k] > (int)d) --k;
  return {k, ebits[k], d - (uint32_t)base[k]};
}

inline size_t match_len(const uint8_t* a, const uint8_t* b, size_t max) {  // common prefix of a[] and b[], <= max
  size_t n = 0;
  while (n + 8 <= max) {
    uint64_t x, y;
    memcpy(&x, a + n, 8);
    memcpy(&y, b + n, 8);
    if (x != y) return n + (size_t)(__builtin_ctzll(x ^ y) >> 3);
    n += 8;
  }
  while (n < max && a[n] == b[n]) ++n;
  return n;
}

// raw deflate stream of src[0..n) into out; d1 / d2: the two candidate distances (d2 = 0: only d1)
void deflate_two_distances(const uint8_t* src, size_t n, uint32_t d1, uint32_t d2, std::vector<uint8_t>& out) {
  static const FixedHuffman H;
  const DistCode c1 = dist_code_of(d1), c2 = d2 ? dist_code_of(d2) : DistCode{0, 0, 0};
  out.clear();
  out.reserve(n / 16 + 64);
  BitSink bs(out);
  bs.put(1, 1);  // BFINAL
  bs.put(1, 2);  // BTYPE = 01, fixed Huffman codes
  size_t i = 0;
  while (i < n) {
    const size_t room = n - i < 258 ? n - i : 258;
    size_t l1 = i >= d1 ? match_len(src + i, src + i - d1, room) : 0;
    size_t l2 = (d2 && i >= d2) ? match_len(src + i, src + i - d2, room) : 0;
    // the far candidate costs ~9 bits more: it must win by more than a literal's worth
    const bool far = l2 > l1 + 1;
    const size_t len = far ? l2 : l1;
    if (len >= 3) {
      const DistCode& dc = far ? c2 : c1;
      bs.put(H.lit_code[H.len_sym[len]], H.lit_bits[H.len_sym[len]]);
      if (H.len_extra_bits[len]) bs.put(H.len_extra[len], H.len_extra_bits[len]);
      bs.put(H.dist_code[dc.sym], 5);
      if (dc.ebits) bs.put(dc.extra, dc.ebits);
      i += len;
    } else {
      bs.put(H.lit_code[src[i]], H.lit_bits[src[i]]);
      ++i;
    }
  }
  bs.put(H.lit_code[256], H.lit_bits[256]);  // end of block
  bs.flush();
}

void le32(std::vector<uint8_t>& b, uint32_t v) { for (int i = 0; i < 4; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
void le16(std::vector<uint8_t>& b, uint16_t v) { b.push_back((uint8_t)v); b.push_back((uint8_t)(v >> 8)); }
void le64(std::vector<uint8_t>& b, uint64_t v) { for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(v >> (8 * i))); }

// parquet enums
enum { PQ_INT32 = 1, PQ_INT64 = 2, PQ_DOUBLE = 5, PQ_BYTE_ARRAY = 6 };
enum { PQ_PLAIN = 0, PQ_RLE = 3 };
enum { PQ_CODEC_NONE = 0, PQ_CODEC_ZSTD = 6 };

}  // namespace

extern "C" {

int aliby_parquet_write(const char* path, const aliby_pq_column* cols, int n_cols, const int64_t* seg_rows, int n_segs,
                        const void* const* values, const void* const* aux, int zstd_level) {
  ARG_CHECK(path && cols && n_cols > 0 && n_segs >= 0 && (n_segs == 0 || (seg_rows && values)), "parquet_write: null argument");
  try {
    int64_t n_rows = 0;
    for (int s = 0; s < n_segs; ++s) { ARG_CHECK(seg_rows[s] >= 0, "parquet_write: negative segment length"); n_rows += seg_rows[s]; }
    ARG_CHECK(n_rows < (1ll << 31), "parquet_write: more than 2^31 rows in one row group");
    const bool use_zstd = zstd_level != ALIBY_PQ_UNCOMPRESSED;
    if (use_zstd && !zstd().ok) { aliby_set_error("parquet_write: zstd compression asked for but libzstd.so.1 is not on this machine"); return ALIBY_ERR_UNSUPPORTED; }
    void* cctx = use_zstd ? thread_cctx() : nullptr;
    File out(path);
    if (out.f < 0) { aliby_set_error("parquet_write: cannot open %s for writing", path); return ALIBY_ERR_INVALID; }
    if (!out.put("PAR1", 4)) { aliby_set_error("parquet_write: write to %s failed", path); return ALIBY_ERR_INVALID; }
    int64_t offset = 4;

    // definition levels of a page whose values are all present: one RLE run of n_rows ones (bit width 1), length-prefixed
    std::vector<uint8_t> deflev;
    {
      std::vector<uint8_t> run;
      uint64_t h = (uint64_t)n_rows << 1;
      while (h >= 0x80) { run.push_back((uint8_t)(h | 0x80)); h >>= 7; }
      run.push_back((uint8_t)h);
      run.push_back(1);
      le32(deflev, (uint32_t)run.size());
      deflev.insert(deflev.end(), run.begin(), run.end());
    }
    struct Chunk { int64_t page_offset, comp, uncomp; };
    std::vector<Chunk> chunks((size_t)n_cols);
    std::vector<uint8_t> page, comp, header;
    int64_t total_uncomp = 0, total_comp = 0;
    for (int c = 0; c < n_cols; ++c) {
      const int type = cols[c].type;
      page.assign(deflev.begin(), deflev.end());
      for (int s = 0; s < n_segs; ++s) {
        const void* v = values[(size_t)c * n_segs + s];
        const int64_t n = seg_rows[s];
        if (n == 0) continue;
        ARG_CHECK(v != nullptr, "parquet_write: null value buffer");
        if (type == ALIBY_PQ_F64 || type == ALIBY_PQ_I64) {
          const uint8_t* p = static_cast<const uint8_t*>(v);
          page.insert(page.end(), p, p + 8 * n);
        } else if (type == ALIBY_PQ_U16) {  // physical INT32
          const uint16_t* p = static_cast<const uint16_t*>(v);
          const size_t at = page.size();
          page.resize(at + 4 * (size_t)n);
          for (int64_t i = 0; i < n; ++i) { const uint32_t w = p[i]; memcpy(&page[at + 4 * (size_t)i], &w, 4); }
        } else if (type == ALIBY_PQ_STR) {  // Arrow utf8: int32 offsets[n + 1] + character data
          const int32_t* off = static_cast<const int32_t*>(v);
          const char* data = aux ? static_cast<const char*>(aux[(size_t)c * n_segs + s]) : nullptr;
          ARG_CHECK(data != nullptr || off[n] == off[0], "parquet_write: string column without character data");
          for (int64_t i = 0; i < n; ++i) {
            const uint32_t len = (uint32_t)(off[i + 1] - off[i]);
            le32(page, len);
            if (len) page.insert(page.end(), data + off[i], data + off[i] + len);
          }
        } else {
          aliby_set_error("parquet_write: column %d has unknown type %d", c, type);
          return ALIBY_ERR_INVALID;
        }
      }
      const uint8_t* body = page.data();
      size_t body_n = page.size();
      if (use_zstd) {
        comp.resize(zstd().compressBound(page.size()));
        const size_t got = zstd().compressCCtx(cctx, comp.data(), comp.size(), page.data(), page.size(), zstd_level);
        if (zstd().isError(got)) { aliby_set_error("parquet_write: zstd failed on column %d", c); return ALIBY_ERR_INVALID; }
        body = comp.data();
        body_n = got;
      }
      header.clear();
      Thrift t(header);
      t.i32(1, 0);                       // PageHeader.type = DATA_PAGE
      t.i32(2, (int32_t)page.size());    // uncompressed_page_size
      t.i32(3, (int32_t)body_n);         // compressed_page_size
      t.begin(5);                        // data_page_header
      t.i32(1, (int32_t)n_rows);         //   num_values
      t.i32(2, PQ_PLAIN);                //   encoding
      t.i32(3, PQ_RLE);                  //   definition_level_encoding
      t.i32(4, PQ_RLE);                  //   repetition_level_encoding
      t.end();
      t.end();
      if (!out.put(header.data(), header.size()) || !out.put(body, body_n)) { aliby_set_error("parquet_write: write to %s failed", path); return ALIBY_ERR_INVALID; }
      chunks[c] = {offset, (int64_t)(header.size() + body_n), (int64_t)(header.size() + page.size())};
      offset += chunks[c].comp;
      total_uncomp += chunks[c].uncomp;
      total_comp += chunks[c].comp;
    }

    // ---- footer: FileMetaData
    std::vector<uint8_t> meta;
    Thrift m(meta);
    m.i32(1, 2);  // version
    m.list(2, 12, (size_t)n_cols + 1);  // schema
    m.begin_elem();
    m.str(4, "schema", 6);
    m.i32(5, n_cols);
    m.end();
    for (int c = 0; c < n_cols; ++c) {
      const int type = cols[c].type;
      m.begin_elem();
      m.i32(1, type == ALIBY_PQ_F64 ? PQ_DOUBLE : type == ALIBY_PQ_I64 ? PQ_INT64 : type == ALIBY_PQ_U16 ? PQ_INT32 : PQ_BYTE_ARRAY);
      m.i32(3, 1);  // repetition_type = OPTIONAL
      m.str(4, cols[c].name, strlen(cols[c].name));
      if (type == ALIBY_PQ_STR) {
        m.i32(6, 0);   // converted_type UTF8
        m.begin(10);   // logicalType
        m.begin(1);    //   STRING
        m.end();
        m.end();
      } else if (type == ALIBY_PQ_U16) {
        m.i32(6, 12);  // converted_type UINT_16
        m.begin(10);
        m.begin(10);   //   INTEGER
        m.i8(1, 16);
        m.boolean(2, false);
        m.end();
        m.end();
      }
      m.end();
    }
    m.i64(3, n_rows);
    m.list(4, 12, 1);  // row_groups
    m.begin_elem();
    m.list(1, 12, (size_t)n_cols);  // columns
    for (int c = 0; c < n_cols; ++c) {
      const int type = cols[c].type;
      m.begin_elem();                    // ColumnChunk
      m.i64(2, 0);                       //   file_offset (as pyarrow >= 15 writes it)
      m.begin(3);                        //   meta_data
      m.i32(1, type == ALIBY_PQ_F64 ? PQ_DOUBLE : type == ALIBY_PQ_I64 ? PQ_INT64 : type == ALIBY_PQ_U16 ? PQ_INT32 : PQ_BYTE_ARRAY);
      m.list(2, 5, 2);                   //     encodings
      m.elem_i32(PQ_PLAIN);
      m.elem_i32(PQ_RLE);
      m.list(3, 8, 1);                   //     path_in_schema
      m.elem_str(cols[c].name, strlen(cols[c].name));
      m.i32(4, use_zstd ? PQ_CODEC_ZSTD : PQ_CODEC_NONE);
      m.i64(5, n_rows);                  //     num_values
      m.i64(6, chunks[c].uncomp);        //     total_uncompressed_size
      m.i64(7, chunks[c].comp);          //     total_compressed_size
      m.i64(9, chunks[c].page_offset);   //     data_page_offset
      m.end();
      m.end();
    }
    m.i64(2, total_uncomp);  // total_byte_size
    m.i64(3, n_rows);        // num_rows
    m.i64(5, 4);             // file_offset of the first column chunk
    m.i64(6, total_comp);    // total_compressed_size
    m.field(7, 4);           // ordinal (i16)
    m.varint(0);
    m.end();
    m.str(6, "aliby_amd libaliby_hip parquet writer");
    m.end();
    std::vector<uint8_t> tail;
    le32(tail, (uint32_t)meta.size());
    tail.insert(tail.end(), {'P', 'A', 'R', '1'});
    if (!out.put(meta.data(), meta.size()) || !out.put(tail.data(), tail.size()) || !out.flush()) {
      aliby_set_error("parquet_write: write to %s failed", path);
      return ALIBY_ERR_INVALID;
    }
    return ALIBY_OK;
  } catch (const std::exception& e) {
    aliby_set_error("parquet_write: %s", e.what());
    return ALIBY_ERR_INVALID;
  }
}

int aliby_npz_write(const char* path, const aliby_npy_member* members, int n_members, int level) {
  ARG_CHECK(path && members && n_members > 0, "npz_write: null argument");
  ARG_CHECK(n_members <= 65535, "npz_write: more than 65535 members need zip64 (not written here)");
  ARG_CHECK(level >= 0 && level <= 9, "npz_write: deflate level 0..9");
  try {
    File out(path);
    if (out.f < 0) { aliby_set_error("npz_write: cannot open %s for writing", path); return ALIBY_ERR_INVALID; }
    struct Entry { std::string name; uint32_t crc; uint64_t comp, uncomp, offset; };
    std::vector<Entry> entries;
    std::vector<uint8_t> raw, comp, head;
    uint64_t offset = 0;
    const Deflate& d = deflater();
    for (int k = 0; k < n_members; ++k) {
      const aliby_npy_member& mb = members[k];
      ARG_CHECK(mb.name && mb.descr && mb.ndim >= 0 && mb.ndim <= 32 && (mb.ndim == 0 || mb.shape) && mb.itemsize > 0, "npz_write: bad member");
      // ---- the .npy image: magic, version 1.0, header dict padded to a multiple of 64 bytes, C-order data
      std::string dict = "{'descr': '" + std::string(mb.descr) + "', 'fortran_order': False, 'shape': (";
      uint64_t count = 1;
      for (int i = 0; i < mb.ndim; ++i) {
        ARG_CHECK(mb.shape[i] >= 0, "npz_write: negative dimension");
        dict += std::to_string(mb.shape[i]) + (mb.ndim == 1 || i + 1 < mb.ndim ? "," : "");
        if (i + 1 < mb.ndim) dict += " ";
        count *= (uint64_t)mb.shape[i];
      }
      dict += "), }";
      size_t total = 10 + dict.size() + 1;
      const size_t padded = (total + 63) / 64 * 64;
      dict.append(padded - total, ' ');
      dict += "\n";
      ARG_CHECK(dict.size() < 65536, "npz_write: header too long for .npy version 1.0");
      const uint64_t nbytes = count * (uint64_t)mb.itemsize;
      ARG_CHECK(nbytes == 0 || mb.data, "npz_write: null data");
      raw.clear();
      raw.insert(raw.end(), {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0});
      le16(raw, (uint16_t)dict.size());
      raw.insert(raw.end(), dict.begin(), dict.end());
      const uint8_t* dp = static_cast<const uint8_t*>(mb.data);
      raw.insert(raw.end(), dp, dp + nbytes);
      ARG_CHECK(raw.size() < 0xFFFFFFFFull, "npz_write: members of 4 GiB and more need zip64 (not written here)");
      // ---- deflate + crc.  Integer images first go through the two-distance encoder (item back, row back); its output is
      // kept when it shows the data has that structure, otherwise the general encoder runs
      uint32_t crc;
      size_t got = 0;
      const bool integer = mb.descr[1] == 'u' || mb.descr[1] == 'i' || mb.descr[1] == 'b';
      if (level > 0 && integer && mb.ndim >= 2 && nbytes >= 4096) {
        const uint64_t row = (uint64_t)mb.shape[mb.ndim - 1] * (uint64_t)mb.itemsize;
        deflate_two_distances(raw.data(), raw.size(), (uint32_t)mb.itemsize, row >= 3 && row <= 32768 ? (uint32_t)row : 0u, comp);
        if (comp.size() * 8 <= raw.size()) got = comp.size();
      }
      if (got) {
        crc = d.ok ? d.crc(0, raw.data(), raw.size()) : (uint32_t)crc32(0L, raw.data(), (uInt)raw.size());
      } else if (d.ok) {
        void* c = d.alloc(level < 1 ? 1 : level);
        if (!c) { aliby_set_error("npz_write: libdeflate_alloc_compressor failed"); return ALIBY_ERR_INVALID; }
        comp.resize(d.bound(c, raw.size()));
        got = d.compress(c, raw.data(), raw.size(), comp.data(), comp.size());
        d.release(c);
        crc = d.crc(0, raw.data(), raw.size());
        if (got == 0) { aliby_set_error("npz_write: deflate failed"); return ALIBY_ERR_INVALID; }
      } else {
        z_stream z;
        memset(&z, 0, sizeof(z));
        if (deflateInit2(&z, level, Z_DEFLATED, -15, 9, Z_DEFAULT_STRATEGY) != Z_OK) { aliby_set_error("npz_write: zlib initialisation failed"); return ALIBY_ERR_INVALID; }
        comp.resize(deflateBound(&z, (uLong)raw.size()));
        z.next_in = raw.data(); z.avail_in = (uInt)raw.size();
        z.next_out = comp.data(); z.avail_out = (uInt)comp.size();
        const int r = deflate(&z, Z_FINISH);
        got = z.total_out;
        deflateEnd(&z);
        if (r != Z_STREAM_END) { aliby_set_error("npz_write: deflate failed (%d)", r); return ALIBY_ERR_INVALID; }
        crc = (uint32_t)crc32(0L, raw.data(), (uInt)raw.size());
      }
      ARG_CHECK(offset + got < 0xFFFFFFFFull, "npz_write: archives of 4 GiB and more need zip64 (not written here)");
      Entry e{std::string(mb.name) + ".npy", crc, (uint64_t)got, (uint64_t)raw.size(), offset};
      head.clear();
      le32(head, 0x04034b50u);
      le16(head, 20); le16(head, 0); le16(head, 8);  // version needed, flags, method = deflate
      le16(head, 0); le16(head, 0x21);                // DOS time 00:00:00, date 1980-01-01 (as numpy's ZipInfo default)
      le32(head, crc); le32(head, (uint32_t)e.comp); le32(head, (uint32_t)e.uncomp);
      le16(head, (uint16_t)e.name.size()); le16(head, 0);
      head.insert(head.end(), e.name.begin(), e.name.end());
      if (!out.put(head.data(), head.size()) || !out.put(comp.data(), got)) { aliby_set_error("npz_write: write to %s failed", path); return ALIBY_ERR_INVALID; }
      offset += head.size() + got;
      entries.push_back(e);
    }
    std::vector<uint8_t> cd;
    for (const Entry& e : entries) {
      le32(cd, 0x02014b50u);
      le16(cd, 0x0314); le16(cd, 20); le16(cd, 0); le16(cd, 8);  // made by (unix, 2.0), needed, flags, method
      le16(cd, 0); le16(cd, 0x21);
      le32(cd, e.crc); le32(cd, (uint32_t)e.comp); le32(cd, (uint32_t)e.uncomp);
      le16(cd, (uint16_t)e.name.size()); le16(cd, 0); le16(cd, 0); le16(cd, 0); le16(cd, 0);
      le32(cd, 0x01800000u);  // external attributes: -rw------- like zipfile.writestr
      le32(cd, (uint32_t)e.offset);
      cd.insert(cd.end(), e.name.begin(), e.name.end());
    }
    const size_t cd_size = cd.size();
    le32(cd, 0x06054b50u);
    le16(cd, 0); le16(cd, 0); le16(cd, (uint16_t)entries.size()); le16(cd, (uint16_t)entries.size());
    le32(cd, (uint32_t)cd_size); le32(cd, (uint32_t)offset); le16(cd, 0);
    if (!out.put(cd.data(), cd.size()) || !out.flush()) { aliby_set_error("npz_write: write to %s failed", path); return ALIBY_ERR_INVALID; }
    return ALIBY_OK;
  } catch (const std::exception& e) {
    aliby_set_error("npz_write: %s", e.what());
    return ALIBY_ERR_INVALID;
  }
}

int aliby_host_copy(void* dst, const void* src, size_t bytes, int threads) {
  ARG_CHECK((dst && src) || bytes == 0, "host_copy: null argument");
  threads = std::max(1, std::min(threads, 16));
  const size_t piece = ((bytes + threads - 1) / threads + 4095) & ~(size_t)4095;  // whole pages: a page is first touched by one thread
  if (threads == 1 || bytes < (8u << 20)) { memcpy(dst, src, bytes); return ALIBY_OK; }
  try {
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) {
      const size_t lo = std::min(bytes, piece * t), hi = std::min(bytes, piece * (t + 1));
      if (hi > lo) pool.emplace_back([=] { memcpy(static_cast<uint8_t*>(dst) + lo, static_cast<const uint8_t*>(src) + lo, hi - lo); });
    }
    memcpy(dst, src, std::min(bytes, piece));
    for (auto& th : pool) th.join();
  } catch (const std::exception& e) {
    aliby_set_error("host_copy: %s", e.what());
    return ALIBY_ERR_INVALID;
  }
  return ALIBY_OK;
}

int aliby_host_codecs(int* have_zstd, int* have_libdeflate) {
  if (have_zstd) *have_zstd = zstd().ok ? 1 : 0;
  if (have_libdeflate) *have_libdeflate = deflater().ok ? 1 : 0;
  return ALIBY_OK;
}

}  // extern "C"
