// The predict -> chop wire format written and read natively (include/deepchopper_io.h: dc_pt_*).
//
// Reference behaviour restated here (never its code):
//   CustomWriter.write_on_batch_end            deepchopper/models/callbacks.py:12-25   torch.save(prediction_dict, "{rank}_{batch}.pt")
//   the dict a predict step returns            deepchopper/models/basic_module.py      {prediction f32[B,L,2], target i64[B,L], seq i64[B,L],
//                                                                                       qual f32[B,L], id i64[B,256]}
//   the id row [len, truncated, ord(c)...]      deepchopper/models/llm/tokenizer.py:170-175 (+ the collator's int8 cast, :91)
//   the consumer                               src/smooth/predict.rs:264-317 (candle_core::pickle::read_all: zip entry "*/data.pkl"
//                                              + "*/data/<key>" storages)
// torch.save's container is a ZIP of STORED entries: `<stem>/data.pkl` (pickle protocol 2; tensors are persistent-id
// references to storages by key), `<stem>/data/<key>` (raw little-endian storage bytes, 64-byte aligned in the file),
// `byteorder`, `version` and a few markers.  torch.save spends its time in a byte-wise CRC-32 and in Python; a fused launch
// is ~85 files of ~1 MB.  Here a pool of threads builds every file of a launch in memory -- rows cut out of the launch's dense
// host arrays, token ids / labels widened to int64, zlib's CRC-32 -- and writes it with one write().
#include "../../include/deepchopper_io.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace dcio {
int fail(int code, const char* fmt, ...);  // fastq_io.cpp: sets the thread-local message of dc_io_last_error()
}

namespace {

struct Bytes {
  std::vector<uint8_t> v;
  void u8(unsigned x) { v.push_back((uint8_t)x); }
  void u16(unsigned x) {
    u8(x & 255);
    u8((x >> 8) & 255);
  }
  void u32(uint64_t x) {
    u16((unsigned)(x & 0xffff));
    u16((unsigned)((x >> 16) & 0xffff));
  }
  void raw(const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    v.insert(v.end(), b, b + n);
  }
  void str(const std::string& s) { raw(s.data(), s.size()); }
};

// ---- pickle protocol 2, exactly the opcode stream torch.save emits for this dict ----------------------------------
struct Pickler {
  Bytes b;
  int memo = 0;
  void put() {  // BINPUT / LONG_BINPUT
    if (memo < 256) {
      b.u8('q');
      b.u8(memo);
    } else {
      b.u8('r');
      b.u32(memo);
    }
    ++memo;
  }
  void get(int i) {  // BINGET (every memo index referenced here is < 256)
    b.u8('h');
    b.u8(i);
  }
  void unicode(const char* s) {
    b.u8('X');
    b.u32(strlen(s));
    b.raw(s, strlen(s));
    put();
  }
  void global(const char* module, const char* name) {
    b.u8('c');
    b.raw(module, strlen(module));
    b.u8('\n');
    b.raw(name, strlen(name));
    b.u8('\n');
    put();
  }
  void integer(int64_t x) {  // pickle.save_long, protocol 2
    if (x >= 0 && x <= 0xff) {
      b.u8('K');
      b.u8((unsigned)x);
    } else if (x >= 0 && x <= 0xffff) {
      b.u8('M');
      b.u16((unsigned)x);
    } else if (x >= -0x80000000ll && x <= 0x7fffffffll) {
      b.u8('J');
      b.u32((uint32_t)(int32_t)x);
    } else {  // LONG1: little-endian two's complement, minimal length
      uint8_t le[9];
      int n = 0;
      int64_t y = x;
      for (;;) {
        le[n++] = (uint8_t)(y & 255);
        const bool sign = (le[n - 1] & 0x80) != 0;
        y >>= 8;
        if ((y == 0 && !sign) || (y == -1 && sign)) break;
      }
      b.u8(0x8a);
      b.u8(n);
      b.raw(le, n);
    }
  }
  void tuple_of(const int64_t* x, int n) {  // n = 2 or 3
    for (int i = 0; i < n; ++i) integer(x[i]);
    b.u8(n == 2 ? 0x86 : 0x87);
    put();
  }
};

// storage class memo indices as torch's pickler assigns them in this dict: 4 = FloatStorage (first use: prediction),
// 15 = LongStorage (first use: target)
std::vector<uint8_t> make_pickle(int64_t B, int64_t L) {
  Pickler p;
  p.b.u8(0x80);
  p.b.u8(2);
  p.b.u8('}');
  p.put();  // 0
  p.b.u8('(');
  struct T {
    const char* key;
    bool is_float;
    int dims;
    int64_t shape[3];
  } ts[5] = {{"prediction", true, 3, {B, L, 2}}, {"target", false, 2, {B, L, 0}}, {"seq", false, 2, {B, L, 0}},
             {"qual", true, 2, {B, L, 0}},       {"id", false, 2, {B, 256, 0}}};
  int m_rebuild = -1, m_storage = -1, m_float = -1, m_long = -1, m_cpu = -1, m_odict = -1;
  for (int i = 0; i < 5; ++i) {
    const T& t = ts[i];
    int64_t numel = 1, stride[3];
    for (int d = t.dims - 1; d >= 0; --d) {
      stride[d] = numel;
      numel *= t.shape[d];
    }
    p.unicode(t.key);
    if (m_rebuild < 0) {
      m_rebuild = p.memo;
      p.global("torch._utils", "_rebuild_tensor_v2");
    } else {
      p.get(m_rebuild);
    }
    p.b.u8('(');
    p.b.u8('(');
    if (m_storage < 0) {
      m_storage = p.memo;
      p.unicode("storage");
    } else {
      p.get(m_storage);
    }
    int& m_cls = t.is_float ? m_float : m_long;
    if (m_cls < 0) {
      m_cls = p.memo;
      p.global("torch", t.is_float ? "FloatStorage" : "LongStorage");
    } else {
      p.get(m_cls);
    }
    const char key[2] = {(char)('0' + i), 0};
    p.unicode(key);
    if (m_cpu < 0) {
      m_cpu = p.memo;
      p.unicode("cpu");
    } else {
      p.get(m_cpu);
    }
    p.integer(numel);
    p.b.u8('t');
    p.put();
    p.b.u8('Q');   // BINPERSID
    p.integer(0);  // storage offset
    p.tuple_of(t.shape, t.dims);
    p.tuple_of(stride, t.dims);
    p.b.u8(0x89);  // requires_grad = False
    if (m_odict < 0) {
      m_odict = p.memo;
      p.global("collections", "OrderedDict");
    } else {
      p.get(m_odict);
    }
    p.b.u8(')');
    p.b.u8('R');
    p.put();
    p.b.u8('t');
    p.put();
    p.b.u8('R');
    p.put();
  }
  p.b.u8('u');
  p.b.u8('.');
  return std::move(p.b.v);
}

// ---- CRC-32 (the ZIP one: reflected 0x04C11DB7) by carry-less multiplication ---------------------------------------
// Folding as in Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ" (Intel, 2009): four 128-bit
// lanes are multiplied by x^(512 +- 32) mod P and xored into the next 64 bytes, then folded to one lane (x^(128 +- 32)),
// to 64 bits and reduced (Barrett).  Constants for the reflected polynomial.  zlib 1.2.11's table CRC runs at ~1 GB/s and was
// the largest share of a file's cost; this one at > 10 GB/s.  Checked against zlib's on first use (every length remainder x
// every pointer misalignment) and on every file's small entries (`Zip::small`: data.pkl, byteorder, version); a CPU without
// PCLMULQDQ or any mismatch keeps zlib's.
#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) inline __m128i crc_fold(__m128i x, __m128i k, __m128i next) {
  return _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x, k, 0x00), _mm_clmulepi64_si128(x, k, 0x11)), next);
}
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_clmul(uint32_t crc0, const uint8_t* p, size_t n) {
  // n >= 64 and a multiple of 16
  const __m128i k512 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);  // x^(512-32), x^(512+32) mod P
  const __m128i k128 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);  // x^(128-32), x^(128+32)
  const __m128i k64 = _mm_set_epi64x(0, 0x0163cd6124ll);                // x^64
  const __m128i kbar = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);  // mu, P
  const __m128i lo32 = _mm_setr_epi32(-1, 0, -1, 0);
  __m128i a = _mm_xor_si128(_mm_loadu_si128((const __m128i*)p), _mm_cvtsi32_si128((int)~crc0));
  __m128i b = _mm_loadu_si128((const __m128i*)(p + 16)), c = _mm_loadu_si128((const __m128i*)(p + 32)),
          d = _mm_loadu_si128((const __m128i*)(p + 48));
  p += 64;
  n -= 64;
  for (; n >= 64; p += 64, n -= 64) {
    a = crc_fold(a, k512, _mm_loadu_si128((const __m128i*)p));
    b = crc_fold(b, k512, _mm_loadu_si128((const __m128i*)(p + 16)));
    c = crc_fold(c, k512, _mm_loadu_si128((const __m128i*)(p + 32)));
    d = crc_fold(d, k512, _mm_loadu_si128((const __m128i*)(p + 48)));
  }
  a = crc_fold(a, k128, b);
  a = crc_fold(a, k128, c);
  a = crc_fold(a, k128, d);
  for (; n >= 16; p += 16, n -= 16) a = crc_fold(a, k128, _mm_loadu_si128((const __m128i*)p));
  // 128 -> 64 bits
  __m128i t = _mm_clmulepi64_si128(a, k128, 0x10);
  a = _mm_xor_si128(_mm_srli_si128(a, 8), t);
  t = _mm_srli_si128(a, 4);
  a = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(a, lo32), k64, 0x00), t);
  // Barrett reduction
  t = _mm_and_si128(_mm_clmulepi64_si128(_mm_and_si128(a, lo32), kbar, 0x10), lo32);
  a = _mm_xor_si128(a, _mm_clmulepi64_si128(t, kbar, 0x00));
  return ~(uint32_t)_mm_extract_epi32(a, 1);
}
#endif

// Cleared when the carry-less CRC ever disagrees with zlib's (the self-test below, or the cross-check `small()` makes on every
// small ZIP entry, data.pkl among them): from then on every CRC of the process is zlib's.
std::atomic<bool> g_clmul_ok{true};

uint32_t crc32_fast(const uint8_t* p, size_t n) {
#if defined(__x86_64__)
  static const bool ok = [] {
    if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
    // every length remainder mod 64 (the fold loop's tail, the 16-byte steps, zlib's byte tail) at every pointer
    // misalignment mod 16, and two long runs
    static uint8_t t[8192 + 64];
    uint32_t x = 12345;
    for (auto& v : t) v = (uint8_t)((x = x * 1664525u + 1013904223u) >> 24);
    auto both = [&](const uint8_t* q, size_t len) {
      const size_t body = len & ~(size_t)15;
      uint32_t c = crc32_clmul(0, q, body);
      if (body != len) c = (uint32_t)crc32_z(c, q + body, len - body);
      return c == (uint32_t)crc32_z(0L, q, len);
    };
    for (size_t off = 0; off < 16; ++off)
      for (size_t len = 64; len < 64 + 130; ++len)
        if (!both(t + off, len)) return false;
    return both(t + 3, 4096 + 48) && both(t, 8192 + 61);
  }();
  if (ok && n >= 64 && g_clmul_ok.load(std::memory_order_relaxed)) {
    const size_t body = n & ~(size_t)15;
    const uint32_t c = crc32_clmul(0, p, body);
    return body == n ? c : (uint32_t)crc32_z(c, p + body, n - body);
  }
#endif
  return (uint32_t)crc32_z(0L, p, n);
}

// ---- ZIP container (STORED entries, data 64-byte aligned as torch's writer pads it) -----------------------------------
// A file is written with one writev(): the metadata stream (local headers, the small entries, the central directory) is
// built in memory and cut where a storage goes; the storages themselves are the iovecs in between -- the pinned host copy
// of what the GPU packed (dc_pt_submit_packed: nothing is copied or converted on the host, the CRC is the only pass over
// the data), or a worker's conversion buffer (dc_pt_submit: dense host arrays).
struct Entry {
  std::string name;
  uint32_t crc, size, offset;
};

// A worker's conversion buffer: 64-byte aligned, grown without zero-filling (every byte is written exactly once;
// std::vector::resize would memset the megabyte the rows are copied into next)
struct RawBuf {
  uint8_t* p = nullptr;
  size_t cap = 0;
  ~RawBuf() { free(p); }
  bool reserve(size_t want) {
    if (want <= cap) return true;
    const size_t c = (want + (want >> 2) + 4095) & ~(size_t)4095;
    uint8_t* q = (uint8_t*)aligned_alloc(64, c);
    if (!q) return false;
    free(p);
    p = q;
    cap = c;
    return true;
  }
};

struct Zip {
  Bytes meta;                  // every byte of the file that is not a storage, in file order
  std::vector<size_t> cuts;    // meta offsets at which storage k is spliced in
  std::vector<Entry> entries;
  size_t pos = 0;              // file offset reached so far
  void header(const std::string& name, size_t size, uint32_t crc) {
    size_t extra = 4;
    extra += (64 - (pos + 30 + name.size() + extra) % 64) % 64;
    entries.push_back({name, crc, (uint32_t)size, (uint32_t)pos});
    meta.u32(0x04034b50u);
    meta.u16(20);      // version needed
    meta.u16(0x0800);  // UTF-8 names
    meta.u16(0);       // stored
    meta.u16(0);
    meta.u16(0x21);    // 1980-01-01
    meta.u32(crc);
    meta.u32(size);
    meta.u32(size);
    meta.u16((unsigned)name.size());
    meta.u16((unsigned)extra);
    meta.str(name);
    meta.u8('F');
    meta.u8('B');
    meta.u16((unsigned)(extra - 4));
    for (size_t i = 4; i < extra; ++i) meta.u8(0);
    pos += 30 + name.size() + extra;
  }
  void small(const std::string& name, const void* p, size_t n) {
    // (a few hundred bytes: both CRCs, every file -- a length-specific fault of the fast one would otherwise write entries
    //  torch.load's and the reference's readers reject)
    const uint32_t slow = (uint32_t)crc32_z(0L, (const uint8_t*)p, n);
    if (crc32_fast((const uint8_t*)p, n) != slow) g_clmul_ok.store(false, std::memory_order_relaxed);
    header(name, n, slow);
    meta.raw(p, n);
    pos += n;
  }
  void storage(const std::string& name, size_t size, uint32_t crc) {  // its bytes follow here in the file
    header(name, size, crc);
    cuts.push_back(meta.v.size());
    pos += size;
  }
  void finish() {
    const size_t cd = pos;
    const size_t m0 = meta.v.size();
    for (const Entry& e : entries) {
      meta.u32(0x02014b50u);
      meta.u16(20);
      meta.u16(20);
      meta.u16(0x0800);
      meta.u16(0);
      meta.u16(0);
      meta.u16(0x21);
      meta.u32(e.crc);
      meta.u32(e.size);
      meta.u32(e.size);
      meta.u16((unsigned)e.name.size());
      meta.u16(0);
      meta.u16(0);
      meta.u16(0);
      meta.u16(0);
      meta.u32(0);
      meta.u32(e.offset);
      meta.str(e.name);
    }
    const size_t cd_size = meta.v.size() - m0;
    meta.u32(0x06054b50u);
    meta.u16(0);
    meta.u16(0);
    meta.u16((unsigned)entries.size());
    meta.u16((unsigned)entries.size());
    meta.u32(cd_size);
    meta.u32(cd);
    meta.u16(0);
    pos += meta.v.size() - m0;
  }
};

struct Group;

struct FileJob {
  Group* g;
  int index;  // file of the group
};

struct Group {
  int64_t ticket = 0;
  std::string dir;
  int rank = 0;
  int64_t first_index = 0;
  std::vector<int32_t> row0, n_rows, col0;  // packed: col0 holds the file's columns L
  // dense host arrays of the launch (dc_pt_submit) ...
  const float* logits = nullptr;
  const int8_t* labels = nullptr;
  const uint8_t* ids = nullptr;
  const float* quals = nullptr;
  int64_t stride = 0;
  // ... or the payloads the GPU packed (dc_pt_submit_packed): file i's storages 0..3 back to back at packed + pay_off[i]
  const uint8_t* packed = nullptr;
  std::vector<int64_t> pay_off;
  std::vector<int64_t> id_rows;  // [rows of the launch][256]
  int remaining = 0;
  int err = 0;
  std::string err_msg;
};

int writev_all(int fd, std::vector<iovec>& iov) {
  size_t k = 0;
  while (k < iov.size()) {
    const int cnt = (int)std::min<size_t>(iov.size() - k, 64);
    ssize_t w = ::writev(fd, iov.data() + k, cnt);
    if (w < 0) {
      if (errno == EINTR) continue;
      return -1;
    }
    while (w > 0 && k < iov.size()) {
      if ((size_t)w >= iov[k].iov_len) {
        w -= (ssize_t)iov[k].iov_len;
        ++k;
      } else {
        iov[k].iov_base = (uint8_t*)iov[k].iov_base + w;
        iov[k].iov_len -= (size_t)w;
        w = 0;
      }
    }
    while (k < iov.size() && iov[k].iov_len == 0) ++k;
  }
  return 0;
}

// One `{rank}_{index}.pt`.  Returns 0 or fills `msg`.
int write_one(const Group& g, int i, RawBuf& buf, std::string& msg) {
  const int64_t B = g.n_rows[i], r0 = g.row0[i];
  const int64_t L = g.packed ? g.col0[i] : g.stride - g.col0[i];
  const std::string stem = std::to_string(g.rank) + "_" + std::to_string(g.first_index + i);
  const std::string path = g.dir + "/" + stem + ".pt";
  const size_t len[5] = {(size_t)(B * L * 8), (size_t)(B * L * 8), (size_t)(B * L * 8), (size_t)(B * L * 4), (size_t)(B * 2048)};
  if ((uint64_t)B * L * 28 + (uint64_t)B * 2048 + 65536 >= 0xffff0000ull) {
    msg = path + ": a .pt file of 4 GiB or more needs ZIP64, which this writer does not emit";
    return 1;
  }
  const uint8_t* sec[5];
  if (g.packed) {
    const uint8_t* base = g.packed + g.pay_off[i];
    sec[0] = base;
    sec[1] = sec[0] + len[0];
    sec[2] = sec[1] + len[1];
    sec[3] = sec[2] + len[2];
  } else {
    const int64_t c0 = g.col0[i];
    if (!buf.reserve(len[0] + len[1] + len[2] + len[3])) {
      msg = path + ": out of memory";
      return 1;
    }
    uint8_t* d0 = buf.p;
    int64_t* d1 = reinterpret_cast<int64_t*>(d0 + len[0]);
    int64_t* d2 = d1 + B * L;
    uint8_t* d3 = reinterpret_cast<uint8_t*>(d2 + B * L);
    for (int64_t r = 0; r < B; ++r) {
      const int64_t src = (r0 + r) * g.stride + c0;
      memcpy(d0 + (size_t)(r * L * 8), g.logits + src * 2, (size_t)(L * 8));  // prediction f32 [B, L, 2]
      const int8_t* sl = g.labels + src;                                        // target i64 [B, L] (-100 = ignore)
      const uint8_t* si = g.ids + src;                                          // seq i64 [B, L]
      for (int64_t c = 0; c < L; ++c) d1[r * L + c] = sl[c];
      for (int64_t c = 0; c < L; ++c) d2[r * L + c] = si[c];
      memcpy(d3 + (size_t)(r * L * 4), g.quals + src, (size_t)(L * 4));       // qual f32 [B, L]
    }
    sec[0] = d0;
    sec[1] = reinterpret_cast<const uint8_t*>(d1);
    sec[2] = reinterpret_cast<const uint8_t*>(d2);
    sec[3] = d3;
  }
  sec[4] = reinterpret_cast<const uint8_t*>(g.id_rows.data() + (size_t)r0 * 256);  // id i64 [B, 256]
  uint32_t crc[5];
  for (int k = 0; k < 5; ++k) crc[k] = crc32_fast(sec[k], len[k]);

  Zip z;
  z.meta.v.reserve(4096);
  const std::vector<uint8_t> pk = make_pickle(B, L);
  z.small(stem + "/data.pkl", pk.data(), pk.size());
  z.small(stem + "/.format_version", "1", 1);
  z.small(stem + "/.storage_alignment", "64", 2);
  z.small(stem + "/byteorder", "little", 6);
  for (int k = 0; k < 5; ++k) z.storage(stem + "/data/" + std::to_string(k), len[k], crc[k]);
  z.small(stem + "/version", "3\n", 2);
  {  // 40 decimal digits, as torch writes; its readers do not look at it
    char sid[48];
    uint64_t h = 1469598103934665603ull;
    for (const Entry& e : z.entries) h = (h ^ e.crc) * 1099511628211ull;
    snprintf(sid, sizeof sid, "%020llu%020llu", (unsigned long long)h, (unsigned long long)(h * 0x9E3779B97F4A7C15ull));
    z.small(stem + "/.data/serialization_id", sid, 40);
  }
  z.finish();
  std::vector<iovec> iov;
  size_t m = 0;
  for (int k = 0; k < 5; ++k) {
    iov.push_back({z.meta.v.data() + m, z.cuts[k] - m});
    iov.push_back({const_cast<uint8_t*>(sec[k]), len[k]});
    m = z.cuts[k];
  }
  iov.push_back({z.meta.v.data() + m, z.meta.v.size() - m});
  const int fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
  if (fd < 0) {
    msg = path + ": " + strerror(errno);
    return 1;
  }
  const int rc = writev_all(fd, iov);
  const int e = errno;
  if (::close(fd) != 0 && rc == 0) {
    msg = path + ": " + strerror(errno);
    return 1;
  }
  if (rc) {
    msg = path + ": " + strerror(e);
    return 1;
  }
  return 0;
}

}  // namespace

struct dc_pt_writer {
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::deque<FileJob> jobs;
  std::deque<std::unique_ptr<Group>> groups;  // oldest first
  std::vector<std::thread> threads;
  int64_t next_ticket = 1;
  bool stop = false;

  void worker() {
    RawBuf buf;
    for (;;) {
      FileJob j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_job.wait(lk, [&] { return stop || !jobs.empty(); });
        if (jobs.empty()) return;
        j = jobs.front();
        jobs.pop_front();
      }
      std::string msg;
      const int rc = write_one(*j.g, j.index, buf, msg);
      {
        std::lock_guard<std::mutex> lk(mu);
        if (rc && !j.g->err) {
          j.g->err = rc;
          j.g->err_msg = msg;
        }
        if (--j.g->remaining == 0) cv_done.notify_all();
      }
    }
  }
};

extern "C" {

int dc_pt_open(int threads, dc_pt_writer** out) {
  if (!out) return dcio::fail(DC_FQ_ERR_ARG, "dc_pt_open: null argument");
  auto* w = new dc_pt_writer();
  const int n = threads < 1 ? 1 : threads > 64 ? 64 : threads;
  for (int i = 0; i < n; ++i) w->threads.emplace_back([w] { w->worker(); });
  *out = w;
  return 0;
}

// id rows: [len(id), truncated, ord(c)...] clipped / zero padded to 256 (tokenizer.py:170-175); the reference's collator casts
// them to int8 (tokenizer.py:91) and fails on anything above 127
static int build_id_rows(Group& g, int64_t n_total_rows, const char* names, const int64_t* name_offsets, const uint8_t* truncated) {
  g.id_rows.assign((size_t)n_total_rows * 256, 0);
  for (int64_t r = 0; r < n_total_rows; ++r) {
    const int64_t len = name_offsets[r + 1] - name_offsets[r];
    const uint8_t* s = (const uint8_t*)names + name_offsets[r];
    int64_t* row = g.id_rows.data() + (size_t)r * 256;
    bool bad = len > 127;
    row[0] = len;
    row[1] = truncated[r] ? 1 : 0;
    for (int64_t c = 0; c < len && c + 2 < 256; ++c) {
      row[c + 2] = s[c];
      bad |= s[c] > 127;
    }
    if (bad) return dcio::fail(DC_FQ_ERR_FORMAT, "value cannot be converted to type int8 without overflow");
  }
  return 0;
}

static int enqueue(dc_pt_writer* w, std::unique_ptr<Group> g, int n_files, int64_t* ticket_out) {
  std::lock_guard<std::mutex> lk(w->mu);
  g->ticket = w->next_ticket++;
  g->remaining = n_files;
  if (ticket_out) *ticket_out = g->ticket;
  Group* gp = g.get();
  w->groups.push_back(std::move(g));
  for (int i = 0; i < n_files; ++i) w->jobs.push_back({gp, i});
  w->cv_job.notify_all();
  return 0;
}

int dc_pt_submit(dc_pt_writer* w, const char* dir, int32_t rank, int64_t first_index, int32_t n_files, const int32_t* row0,
                 const int32_t* n_rows, const int32_t* col0, const float* logits, const int8_t* labels, const uint8_t* ids,
                 const float* quals, int64_t n_total_rows, int64_t row_stride, const char* names, const int64_t* name_offsets,
                 const uint8_t* truncated, int64_t* ticket_out) {
  if (!w || !dir || n_files < 0 || (n_files && (!row0 || !n_rows || !col0 || !logits || !labels || !ids || !quals || !names ||
                                                !name_offsets || !truncated)))
    return dcio::fail(DC_FQ_ERR_ARG, "dc_pt_submit: null argument");
  auto g = std::make_unique<Group>();
  g->dir = dir;
  g->rank = rank;
  g->first_index = first_index;
  g->row0.assign(row0, row0 + n_files);
  g->n_rows.assign(n_rows, n_rows + n_files);
  g->col0.assign(col0, col0 + n_files);
  for (int i = 0; i < n_files; ++i)
    if (row0[i] < 0 || n_rows[i] < 1 || (int64_t)row0[i] + n_rows[i] > n_total_rows || col0[i] < 0 || col0[i] >= row_stride)
      return dcio::fail(DC_FQ_ERR_ARG, "dc_pt_submit: file %d cuts rows [%d, +%d) from column %d out of a [%lld, %lld] launch", i,
                        row0[i], n_rows[i], col0[i], (long long)n_total_rows, (long long)row_stride);
  g->logits = logits;
  g->labels = labels;
  g->ids = ids;
  g->quals = quals;
  g->stride = row_stride;
  if (int rc = build_id_rows(*g, n_total_rows, names, name_offsets, truncated)) return rc;
  return enqueue(w, std::move(g), n_files, ticket_out);
}

int dc_pt_submit_packed(dc_pt_writer* w, const char* dir, int32_t rank, int64_t first_index, int32_t n_files, const int32_t* row0,
                        const int32_t* n_rows, const int32_t* n_cols, const int64_t* payload_offset, const void* payload,
                        int64_t payload_bytes, int64_t n_total_rows, const char* names, const int64_t* name_offsets,
                        const uint8_t* truncated, int64_t* ticket_out) {
  if (!w || !dir || n_files < 0 || (n_files && (!row0 || !n_rows || !n_cols || !payload_offset || !payload || !names ||
                                                !name_offsets || !truncated)))
    return dcio::fail(DC_FQ_ERR_ARG, "dc_pt_submit_packed: null argument");
  auto g = std::make_unique<Group>();
  g->dir = dir;
  g->rank = rank;
  g->first_index = first_index;
  g->row0.assign(row0, row0 + n_files);
  g->n_rows.assign(n_rows, n_rows + n_files);
  g->col0.assign(n_cols, n_cols + n_files);
  g->pay_off.assign(payload_offset, payload_offset + n_files);
  for (int i = 0; i < n_files; ++i)
    if (row0[i] < 0 || n_rows[i] < 1 || (int64_t)row0[i] + n_rows[i] > n_total_rows || n_cols[i] < 1 || payload_offset[i] < 0 ||
        payload_offset[i] + (int64_t)n_rows[i] * n_cols[i] * 28 > payload_bytes)
      return dcio::fail(DC_FQ_ERR_ARG, "dc_pt_submit_packed: file %d ([%d, %d] at byte %lld) does not fit a payload of %lld bytes", i,
                        n_rows[i], n_cols[i], (long long)payload_offset[i], (long long)payload_bytes);
  g->packed = (const uint8_t*)payload;
  if (int rc = build_id_rows(*g, n_total_rows, names, name_offsets, truncated)) return rc;
  return enqueue(w, std::move(g), n_files, ticket_out);
}

int dc_pt_wait(dc_pt_writer* w, int64_t ticket) {
  if (!w) return dcio::fail(DC_FQ_ERR_ARG, "dc_pt_wait: null writer");
  std::unique_lock<std::mutex> lk(w->mu);
  int rc = 0;
  std::string msg;
  while (!w->groups.empty() && w->groups.front()->ticket <= ticket) {
    Group* g = w->groups.front().get();
    w->cv_done.wait(lk, [&] { return g->remaining == 0; });
    if (g->err && !rc) {
      rc = DC_FQ_ERR_IO;
      msg = g->err_msg;
    }
    w->groups.pop_front();
  }
  lk.unlock();
  return rc ? dcio::fail(rc, "%s", msg.c_str()) : 0;
}

// ------------------------------------------------------------------------------------------------------------------
// The consumer side: load_predicts_from_batch_pt(s) (src/smooth/predict.rs:211-317) for the chop driver -- every read's
// id, truncation flag and argmax labels over its unmasked positions, straight out of the files (mmap, the ZIP central
// directory, the storages by key), on a pool of threads.  Shapes follow from the storage sizes: id i64[B][256] gives B,
// target i64[B][L] gives L.  A file this reader does not recognise is reported as failed and left to the caller's general
// path (torch.load).
// ------------------------------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

struct PtRows {
  std::vector<int8_t> labels;
  std::vector<int64_t> label_off{0};
  std::string names;
  std::vector<int64_t> name_off{0};
  std::vector<uint8_t> truncated;
  bool failed = false;
};

inline uint32_t rd32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint32_t rd16(const uint8_t* p) { return p[0] | (p[1] << 8); }

// ---- data.pkl: the subset of pickle protocol 2 torch.save emits for a dict of tensors ------------------------------
// (what candle_core::pickle::read_all interprets for the reference, src/smooth/predict.rs:268.)  A small stack machine:
// strings, integers, tuples, globals, the persistent-id storage tuple and the `_rebuild_tensor_v2` REDUCE.  Anything else --
// an opcode, a global or a shape this reader does not know -- makes the file "not recognised" and the caller's
// torch.load route takes it; nothing is guessed from byte sizes.
struct PtTensor {
  int dtype = -1;  // index into kStorage
  std::string key;  // storage entry "<stem>/data/<key>"
  int64_t storage_numel = 0, offset = 0;
  std::vector<int64_t> shape, stride;
};
struct StorageKind {
  const char* name;
  int width;
};
const StorageKind kStorage[] = {{"FloatStorage", 4}, {"LongStorage", 8}, {"CharStorage", 1}, {"IntStorage", 4},
                                {"ShortStorage", 2}, {"ByteStorage", 1}, {"DoubleStorage", 8}, {"BoolStorage", 1}};
enum { ST_FLOAT = 0, ST_LONG, ST_CHAR, ST_INT, ST_SHORT, ST_BYTE, ST_DOUBLE, ST_BOOL };

struct PVal {
  enum Kind { NONE, MARK, INT, STR, GLOBAL, TUPLE, STORAGE, TENSOR, DICT, BOOL, OPAQUE } kind = NONE;
  int depth = 0;            // nesting of TUPLEs below this value (bounded: destruction is recursive)
  int64_t i = 0;
  std::string s;            // STR; GLOBAL "module name"
  std::vector<PVal> items;  // TUPLE; DICT (key, value, key, value ...)
  PtTensor t;               // STORAGE (dtype, key, numel) / TENSOR
};

bool parse_pickle(const uint8_t* p, size_t n, std::vector<std::pair<std::string, PtTensor>>& out) {
  // torch.save's stream for this dict is ~500 bytes and nests three tuples deep; a "pickle" of megabytes or of thousands of
  // nested tuples is not one of ours (and must not be allowed to recurse through the value destructors)
  constexpr size_t kMaxPickle = 1u << 20;
  constexpr int kMaxDepth = 16;
  if (n > kMaxPickle) return false;
  auto nest = [](std::vector<PVal>& items) {
    int d = 0;
    for (const PVal& x : items) d = std::max(d, x.depth);
    return d + 1;
  };
  std::vector<PVal> st;
  std::vector<PVal> memo;
  size_t i = 0;
  auto need = [&](size_t k) { return i + k <= n; };
  auto pop_to_mark = [&](std::vector<PVal>& items) {
    size_t m = st.size();
    while (m > 0 && st[m - 1].kind != PVal::MARK) --m;
    if (m == 0) return false;
    items.assign(std::make_move_iterator(st.begin() + (long)m), std::make_move_iterator(st.end()));
    st.resize(m - 1);
    return true;
  };
  auto set_memo = [&](size_t k) {
    if (st.empty() || k > 65536) return false;
    if (memo.size() <= k) memo.resize(k + 1);
    memo[k] = st.back();
    return true;
  };
  auto tuple_of = [&](size_t k) {
    if (st.size() < k) return false;
    PVal t;
    t.kind = PVal::TUPLE;
    t.items.assign(std::make_move_iterator(st.end() - (long)k), std::make_move_iterator(st.end()));
    st.resize(st.size() - k);
    t.depth = nest(t.items);
    if (t.depth > kMaxDepth) return false;
    st.push_back(std::move(t));
    return true;
  };
  auto ints = [](const PVal& v, std::vector<int64_t>& o) {
    if (v.kind != PVal::TUPLE) return false;
    for (const PVal& x : v.items) {
      if (x.kind != PVal::INT) return false;
      o.push_back(x.i);
    }
    return true;
  };
  while (i < n) {
    const uint8_t op = p[i++];
    PVal v;
    switch (op) {
      case 0x80:  // PROTO
        if (!need(1) || p[i] < 2 || p[i] > 5) return false;
        ++i;
        break;
      case '}':
        v.kind = PVal::DICT;
        st.push_back(std::move(v));
        break;
      case ')':
        v.kind = PVal::TUPLE;
        st.push_back(std::move(v));
        break;
      case '(':
        v.kind = PVal::MARK;
        st.push_back(std::move(v));
        break;
      case 'N':
        st.push_back(std::move(v));
        break;
      case 0x88:
      case 0x89:
        v.kind = PVal::BOOL;
        v.i = op == 0x88;
        st.push_back(std::move(v));
        break;
      case 'K':
        if (!need(1)) return false;
        v.kind = PVal::INT;
        v.i = p[i];
        i += 1;
        st.push_back(std::move(v));
        break;
      case 'M':
        if (!need(2)) return false;
        v.kind = PVal::INT;
        v.i = rd16(p + i);
        i += 2;
        st.push_back(std::move(v));
        break;
      case 'J':
        if (!need(4)) return false;
        v.kind = PVal::INT;
        v.i = (int32_t)rd32(p + i);
        i += 4;
        st.push_back(std::move(v));
        break;
      case 0x8a: {  // LONG1
        if (!need(1)) return false;
        const unsigned k = p[i++];
        if (k > 8 || !need(k)) return false;
        uint64_t u = 0;
        for (unsigned b = 0; b < k; ++b) u |= (uint64_t)p[i + b] << (8 * b);
        if (k && k < 8 && (p[i + k - 1] & 0x80)) u |= ~(uint64_t)0 << (8 * k);
        i += k;
        v.kind = PVal::INT;
        v.i = (int64_t)u;
        st.push_back(std::move(v));
        break;
      }
      case 'X': {  // BINUNICODE
        if (!need(4)) return false;
        const size_t k = rd32(p + i);
        i += 4;
        if (!need(k)) return false;
        v.kind = PVal::STR;
        v.s.assign((const char*)p + i, k);
        i += k;
        st.push_back(std::move(v));
        break;
      }
      case 0x8c: {  // SHORT_BINUNICODE (protocol 4 writers)
        if (!need(1)) return false;
        const size_t k = p[i++];
        if (!need(k)) return false;
        v.kind = PVal::STR;
        v.s.assign((const char*)p + i, k);
        i += k;
        st.push_back(std::move(v));
        break;
      }
      case 'c': {  // GLOBAL "module\nname\n"
        const uint8_t* a = (const uint8_t*)memchr(p + i, '\n', n - i);
        if (!a) return false;
        const uint8_t* b = (const uint8_t*)memchr(a + 1, '\n', n - (size_t)(a + 1 - p));
        if (!b) return false;
        v.kind = PVal::GLOBAL;
        v.s.assign((const char*)p + i, (size_t)(a - (p + i)));
        v.s += ' ';
        v.s.append((const char*)a + 1, (size_t)(b - a - 1));
        i = (size_t)(b + 1 - p);
        st.push_back(std::move(v));
        break;
      }
      case 'q':
        if (!need(1) || !set_memo(p[i])) return false;
        i += 1;
        break;
      case 'r':
        if (!need(4) || !set_memo(rd32(p + i))) return false;
        i += 4;
        break;
      case 0x94:  // MEMOIZE
        if (!set_memo(memo.size())) return false;
        break;
      case 'h':
      case 'j': {
        const size_t w = op == 'h' ? 1 : 4;
        if (!need(w)) return false;
        const size_t k = op == 'h' ? p[i] : rd32(p + i);
        i += w;
        if (k >= memo.size()) return false;
        st.push_back(memo[k]);
        break;
      }
      case 't':
        v.kind = PVal::TUPLE;
        if (!pop_to_mark(v.items)) return false;
        v.depth = nest(v.items);
        if (v.depth > kMaxDepth) return false;
        st.push_back(std::move(v));
        break;
      case 0x85:
        if (!tuple_of(1)) return false;
        break;
      case 0x86:
        if (!tuple_of(2)) return false;
        break;
      case 0x87:
        if (!tuple_of(3)) return false;
        break;
      case 'Q': {  // BINPERSID: ("storage", <torch XStorage>, key, location, numel)
        if (st.empty()) return false;
        PVal pid = std::move(st.back());
        st.pop_back();
        if (pid.kind != PVal::TUPLE || pid.items.size() != 5 || pid.items[0].kind != PVal::STR || pid.items[0].s != "storage" ||
            pid.items[1].kind != PVal::GLOBAL || pid.items[2].kind != PVal::STR || pid.items[4].kind != PVal::INT)
          return false;
        v.kind = PVal::STORAGE;
        for (int k = 0; k < (int)(sizeof(kStorage) / sizeof(kStorage[0])); ++k)
          if (pid.items[1].s == std::string("torch ") + kStorage[k].name) v.t.dtype = k;
        if (v.t.dtype < 0) return false;
        v.t.key = pid.items[2].s;
        v.t.storage_numel = pid.items[4].i;
        st.push_back(std::move(v));
        break;
      }
      case 'R': {  // REDUCE: callable(*args)
        if (st.size() < 2) return false;
        PVal args = std::move(st.back());
        st.pop_back();
        PVal fn = std::move(st.back());
        st.pop_back();
        if (fn.kind != PVal::GLOBAL || args.kind != PVal::TUPLE) return false;
        if (fn.s == "collections OrderedDict" && args.items.empty()) {
          v.kind = PVal::OPAQUE;  // backward hooks
        } else if (fn.s == "torch._utils _rebuild_tensor_v2" && args.items.size() >= 5 && args.items[0].kind == PVal::STORAGE &&
                   args.items[1].kind == PVal::INT) {
          v.kind = PVal::TENSOR;
          v.t = args.items[0].t;
          v.t.offset = args.items[1].i;
          if (!ints(args.items[2], v.t.shape) || !ints(args.items[3], v.t.stride) || v.t.shape.size() != v.t.stride.size()) return false;
        } else {
          return false;
        }
        st.push_back(std::move(v));
        break;
      }
      case 's': {  // SETITEM
        if (st.size() < 3 || st[st.size() - 3].kind != PVal::DICT) return false;
        PVal val = std::move(st.back());
        st.pop_back();
        PVal key = std::move(st.back());
        st.pop_back();
        st.back().items.push_back(std::move(key));
        st.back().items.push_back(std::move(val));
        break;
      }
      case 'u': {  // SETITEMS
        std::vector<PVal> kv;
        if (!pop_to_mark(kv) || kv.size() % 2 || st.empty() || st.back().kind != PVal::DICT) return false;
        for (PVal& x : kv) st.back().items.push_back(std::move(x));
        break;
      }
      case '.': {
        if (st.size() != 1 || st[0].kind != PVal::DICT) return false;
        const std::vector<PVal>& kv = st[0].items;
        for (size_t k = 0; k + 1 < kv.size(); k += 2) {
          if (kv[k].kind != PVal::STR || kv[k + 1].kind != PVal::TENSOR) return false;
          out.emplace_back(kv[k].s, kv[k + 1].t);
        }
        return true;
      }
      default:
        return false;
    }
  }
  return false;  // no STOP
}

// One tensor of the file, resolved against its storage entry: element (r, c[, k]) by shape / stride / offset.
struct PtView {
  const uint8_t* base = nullptr;
  int dtype = -1;
  int64_t offset = 0;
  int64_t shape[3] = {1, 1, 1}, stride[3] = {0, 0, 0};
  int dims = 0;
  bool contiguous = false;
  int64_t at_i64(int64_t r, int64_t c) const {
    const int64_t e = offset + r * stride[0] + c * stride[1];
    switch (dtype) {
      case ST_LONG: return ((const int64_t*)base)[e];
      case ST_CHAR: return ((const int8_t*)base)[e];
      case ST_INT: return ((const int32_t*)base)[e];
      case ST_SHORT: return ((const int16_t*)base)[e];
      default: return base[e];  // ByteStorage, BoolStorage
    }
  }
};

bool load_one(const char* path, int64_t ignore_label, PtRows& out) {
  const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return false;
  struct stat stt;
  if (fstat(fd, &stt) != 0 || stt.st_size < 22) {
    ::close(fd);
    return false;
  }
  const size_t size = (size_t)stt.st_size;
  const uint8_t* f = (const uint8_t*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
  ::close(fd);
  if (f == MAP_FAILED) return false;
  bool ok = false;
  do {
    // end of central directory: the last 64 KiB + 22 bytes
    size_t e = size - 22;
    const size_t lo = size > 65557 ? size - 65557 : 0;
    for (;; --e) {
      if (rd32(f + e) == 0x06054b50u) break;
      if (e == lo) {
        e = (size_t)-1;
        break;
      }
    }
    if (e == (size_t)-1) break;
    const unsigned n_entries = rd16(f + e + 10);
    const size_t cd_size = rd32(f + e + 12), cd_off = rd32(f + e + 16);
    if (cd_off + cd_size > size) break;
    // every STORED entry "<stem>/data.pkl" and "<stem>/data/<key>": where its bytes lie, and the directory's CRC of them
    struct Ent {
      std::string key;
      const uint8_t* p;
      size_t n;
      uint32_t crc;
    };
    std::vector<Ent> ents;
    const uint8_t* pkl = nullptr;
    size_t pkl_len = 0;
    uint32_t pkl_crc = 0;
    size_t p = cd_off;
    bool bad = false;
    for (unsigned i = 0; i < n_entries; ++i) {
      if (p + 46 > size || rd32(f + p) != 0x02014b50u) {
        bad = true;
        break;
      }
      const unsigned method = rd16(f + p + 10), nlen = rd16(f + p + 28), xlen = rd16(f + p + 30), clen = rd16(f + p + 32);
      const size_t csize = rd32(f + p + 20), usize = rd32(f + p + 24), lho = rd32(f + p + 42);
      const uint32_t crc = rd32(f + p + 16);
      if (p + 46 + nlen > size) {
        bad = true;
        break;
      }
      const std::string name((const char*)f + p + 46, nlen);
      const size_t dpos = name.rfind("/data/");
      const bool is_pkl = nlen >= 9 && name.compare(nlen - 9, 9, "/data.pkl") == 0;
      const bool is_storage = dpos != std::string::npos && name.find('/', dpos + 6) == std::string::npos && dpos + 6 < nlen;
      if (is_pkl || is_storage) {
        if (method != 0 || csize != usize || csize == 0xffffffffu || lho + 30 > size || rd32(f + lho) != 0x04034b50u) {
          bad = true;  // deflated or ZIP64: not what torch.save or this library's writer emits
          break;
        }
        const size_t data = lho + 30 + rd16(f + lho + 26) + rd16(f + lho + 28);
        if (data + usize > size) {
          bad = true;
          break;
        }
        if (is_pkl) {
          pkl = f + data;
          pkl_len = usize;
          pkl_crc = crc;
        } else {
          ents.push_back({name.substr(dpos + 6), f + data, usize, crc});
        }
      }
      p += 46 + nlen + xlen + clen;
    }
    if (bad || !pkl || crc32_fast(pkl, pkl_len) != pkl_crc) break;
    std::vector<std::pair<std::string, PtTensor>> tensors;
    if (!parse_pickle(pkl, pkl_len, tensors)) break;
    // the three tensors the chop driver needs (src/smooth/predict.rs:266-300): prediction [B, L, 2] floating point,
    // target [B, L] and id [B, 256] of any integer storage (CustomWriter widens them to int64; files of older writers hold
    // the collator's int8, deepchopper/models/llm/tokenizer.py:89-91)
    PtView view[3];
    const char* want[3] = {"prediction", "target", "id"};
    for (int w = 0; w < 3 && !bad; ++w) {
      const PtTensor* t = nullptr;
      for (const auto& kv : tensors)
        if (kv.first == want[w]) t = &kv.second;
      const Ent* en = nullptr;
      if (t)
        for (const Ent& x : ents)
          if (x.key == t->key) en = &x;
      const int dims = w == 0 ? 3 : 2;
      if (!t || !en || (int)t->shape.size() != dims || t->offset < 0 || t->storage_numel < 0 ||
          (w == 0 ? t->dtype != ST_FLOAT : (t->dtype == ST_FLOAT || t->dtype == ST_DOUBLE))) {
        bad = true;
        break;
      }
      const int width = kStorage[t->dtype].width;
      if ((size_t)t->storage_numel * (size_t)width != en->n || ((uintptr_t)en->p & (uintptr_t)(width - 1))) {
        bad = true;
        break;
      }
      PtView& v = view[w];
      v.base = en->p;
      v.dtype = t->dtype;
      v.offset = t->offset;
      v.dims = dims;
      int64_t last = t->offset, dense = 1;
      v.contiguous = true;
      for (int d = dims - 1; d >= 0; --d) {
        v.shape[d] = t->shape[(size_t)d];
        v.stride[d] = t->stride[(size_t)d];
        if (v.shape[d] < 0 || v.stride[d] < 0) bad = true;
        if (v.shape[d] > 1 && v.stride[d] != dense) v.contiguous = false;
        dense *= v.shape[d];
        if (v.shape[d] > 0) last += (v.shape[d] - 1) * v.stride[d];
      }
      if (bad || (dense > 0 && last >= t->storage_numel)) {  // an element outside its storage
        bad = true;
        break;
      }
      if (crc32_fast(en->p, en->n) != en->crc) bad = true;  // (torch.load's miniz checks it too; candle's zip crate as well)
    }
    if (bad) break;
    const int64_t B = view[0].shape[0], L = view[0].shape[1];
    if (view[0].shape[2] != 2 || view[1].shape[0] != B || view[1].shape[1] != L || view[2].shape[0] != B || view[2].shape[1] < 2) break;
    const int64_t id_cols = view[2].shape[1];
    const float* logit = (const float*)view[0].base;
    for (int64_t r = 0; r < B; ++r) {
      const int64_t n = view[2].at_i64(r, 0);
      if (n < 0 || n + 2 > id_cols) {
        bad = true;
        break;
      }
      for (int64_t c = 0; c < n; ++c) out.names.push_back((char)view[2].at_i64(r, 2 + c));
      out.name_off.push_back((int64_t)out.names.size());
      out.truncated.push_back(view[2].at_i64(r, 1) != 0);
      if (view[0].contiguous && view[1].contiguous && view[1].dtype == ST_LONG) {  // what CustomWriter and dc_pt_submit write
        const float* lg = logit + view[0].offset + r * L * 2;
        const int64_t* tg = (const int64_t*)view[1].base + view[1].offset + r * L;
        for (int64_t c = 0; c < L; ++c)
          if (tg[c] != ignore_label) out.labels.push_back(lg[2 * c + 1] > lg[2 * c] ? 1 : 0);  // argmax, first maximum wins
      } else {
        for (int64_t c = 0; c < L; ++c) {
          if (view[1].at_i64(r, c) == ignore_label) continue;
          const int64_t e0 = view[0].offset + r * view[0].stride[0] + c * view[0].stride[1];
          out.labels.push_back(logit[e0 + view[0].stride[2]] > logit[e0] ? 1 : 0);
        }
      }
      out.label_off.push_back((int64_t)out.labels.size());
    }
    ok = !bad;
  } while (false);
  munmap((void*)f, size);
  return ok;
}

}  // namespace

struct dc_pt_set {
  std::vector<PtRows> files;
};

extern "C" {

int dc_pt_load(const char* const* paths, int64_t n_paths, int32_t threads, int64_t ignore_label, dc_pt_set** out, int64_t* n_reads,
               int64_t* label_bytes, int64_t* name_bytes, int64_t* n_failed) {
  if (!paths || n_paths < 0 || !out) return dcio::fail(DC_FQ_ERR_ARG, "dc_pt_load: null argument");
  auto* set = new dc_pt_set();
  set->files.resize((size_t)n_paths);
  std::atomic<int64_t> next{0};
  const int nt = threads < 1 ? 1 : threads > 64 ? 64 : threads;
  std::vector<std::thread> pool;
  for (int t = 0; t < nt; ++t)
    pool.emplace_back([&] {
      for (;;) {
        const int64_t i = next.fetch_add(1);
        if (i >= n_paths) return;
        PtRows& r = set->files[(size_t)i];
        if (!load_one(paths[i], ignore_label, r)) {
          r = PtRows();
          r.failed = true;
        }
      }
    });
  for (auto& th : pool) th.join();
  int64_t nr = 0, lb = 0, nb = 0, nf = 0;
  for (const PtRows& r : set->files) {
    nr += (int64_t)r.truncated.size();
    lb += (int64_t)r.labels.size();
    nb += (int64_t)r.names.size();
    nf += r.failed;
  }
  if (n_reads) *n_reads = nr;
  if (label_bytes) *label_bytes = lb;
  if (name_bytes) *name_bytes = nb;
  if (n_failed) *n_failed = nf;
  *out = set;
  return 0;
}

int dc_pt_export(const dc_pt_set* set, int8_t* labels, int64_t* label_offsets, char* names, int64_t* name_offsets,
                 uint8_t* truncated, int32_t* file_of_read, uint8_t* file_failed) {
  if (!set || !label_offsets || !name_offsets) return dcio::fail(DC_FQ_ERR_ARG, "dc_pt_export: null argument");
  int64_t r = 0, lb = 0, nb = 0;
  label_offsets[0] = name_offsets[0] = 0;
  for (size_t fi = 0; fi < set->files.size(); ++fi) {
    const PtRows& f = set->files[fi];
    if (file_failed) file_failed[fi] = f.failed;
    if (!f.labels.empty() && labels) memcpy(labels + lb, f.labels.data(), f.labels.size());
    if (!f.names.empty() && names) memcpy(names + nb, f.names.data(), f.names.size());
    for (size_t k = 0; k < f.truncated.size(); ++k, ++r) {
      label_offsets[r + 1] = lb + f.label_off[k + 1];
      name_offsets[r + 1] = nb + f.name_off[k + 1];
      if (truncated) truncated[r] = f.truncated[k];
      if (file_of_read) file_of_read[r] = (int32_t)fi;
    }
    lb += (int64_t)f.labels.size();
    nb += (int64_t)f.names.size();
  }
  return 0;
}

void dc_pt_free(dc_pt_set* set) { delete set; }

int dc_pt_close(dc_pt_writer* w) {
  if (!w) return 0;
  const int rc = dc_pt_wait(w, INT64_MAX);
  {
    std::lock_guard<std::mutex> lk(w->mu);
    w->stop = true;
  }
  w->cv_job.notify_all();
  for (auto& t : w->threads) t.join();
  delete w;
  return rc;
}

}  // extern "C"
