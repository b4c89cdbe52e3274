// Host-side FASTQ / gzip / BGZF streaming I/O (include/deepchopper_io.h).  No GPU code: the product of
// this file is the packed (seq, qual, offsets) triple that dc_tokenize_batch consumes, and the BGZF
// stream the chop driver emits.
//
// Reference behaviour restated here (never its code):
//   check_compressed_type                     src/output/writefq.rs:84-147
//   create_reader_for_compressed_file         src/output/writefq.rs:148-161
//   StreamingFastqReader                      src/output/writefq.rs:172-199  (4-line records, noodles)
//   parse_fastq_file validation               deepchopper/data/only_fq.py:21-85
//   bgzf::io::MultithreadedWriter usage       src/output/writefq.rs:252-276, src/bin/predict.rs:266-270
#include "../../include/deepchopper_io.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "fast_deflate.h"
#include "fast_inflate.h"

#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

}  // namespace

namespace dcio {
// the same thread-local message for the library's other translation units (pt_io.cpp)
int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
}  // namespace dcio

namespace {

constexpr size_t kBlock = 4u << 20;  // decompressed / plain bytes handed from the source thread per block

// ------------------------------------------------------------------------------------------------
// Source thread: file -> (inflate) -> bounded queue of byte blocks
// ------------------------------------------------------------------------------------------------
// A decompressed block on its way to the parser.  `marks`: gzip members that end inside the block, with the
// CRC-32 / length their trailers promise -- the in-house inflater leaves the check to the consumer thread, so the
// two threads share the per-byte work (zlib's own inflate checks internally and leaves `marks` empty).
struct Mark {
  uint32_t end, crc, isize;
};
struct Block {
  std::vector<uint8_t> data;
  std::vector<Mark> marks;
};

// ------------------------------------------------------------------------------------------------
// Parallel inflate of ONE large gzip member (a plain `.fastq.gz` has a single DEFLATE stream, which cannot be
// split without decoding it).  The compressed range is cut into chunks; a worker thread
//   1. finds the first dynamic-Huffman block header at or after its chunk's byte offset by scanning bit positions
//      (FastInflate::probe_text_block: strict header check + a trial decode that must yield text),
//   2. decodes from there up to the start the next chunk found, as 16-bit symbols behind a 32 768-entry prefix of
//      placeholders (256 + j = "byte j of the 32 KiB before this chunk"): back-references into the unseen window
//      simply copy placeholders.
// The coordinator takes chunks in stream order.  A chunk counts only if it starts exactly where its predecessor
// ended (the predecessor was decoded from a confirmed position, so its block boundaries are the true ones -- by
// induction from chunk 0 the result is the sequential decode, bit for bit; a start that is not confirmed is
// discarded and that stretch is decoded again sequentially).  Placeholders are resolved through a 33 024-entry
// byte table (identity + the previous chunk's last 32 KiB).  CRC-32 / ISIZE are checked as in the serial path.
// ------------------------------------------------------------------------------------------------
class ParInflate {
 public:
  struct Chunk {
    uint64_t start_bit = 0, end_bit = 0;
    bool ended = false;   // the member's final block was decoded: end_bit is the (byte aligned) position of the trailer
    bool failed = false;  // decode error (a false start, or corrupt data: the sequential re-decode decides which)
    std::unique_ptr<uint16_t[]> sym;  // 32 768 placeholders + symbols (not value-initialised: tens of MB per chunk)
    size_t n_out = 0;
  };
  static constexpr size_t SOFT_MAX = 48u << 20;  // symbols per chunk before a worker stops at the next block boundary
  static constexpr uint64_t NONE = ~0ull;
  static constexpr int TRIAL = 2048;

  ParInflate(const uint8_t* origin, const uint8_t* end, int threads, size_t chunk_bytes, std::function<bool()> cancelled,
             std::atomic<int64_t>* cpu_ns = nullptr)
      : origin_(origin), end_(end), chunk_bytes_(chunk_bytes), cancelled_(std::move(cancelled)), cpu_ns_(cpu_ns) {
    total_bits_ = (uint64_t)(end - origin) * 8;
    n_ = (int)(((size_t)(end - origin) + chunk_bytes - 1) / chunk_bytes);
    starts_.assign((size_t)n_ + 1, NONE);
    start_state_.assign((size_t)n_ + 1, 0);
    chunks_.resize((size_t)n_);
    ready_.assign((size_t)n_, 0);
    window_ = 2 * threads + 1;
    for (int t = 0; t < threads; ++t)
      workers_.emplace_back([this] {
        work();
        if (cpu_ns_) cpu_ns_->fetch_add(thread_cpu_ns());  // (a worker lives for one member: its whole CPU time is inflate)
      });
  }
  static int64_t thread_cpu_ns() {
    timespec t;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t);
    return (int64_t)t.tv_sec * 1000000000ll + t.tv_nsec;
  }
  ~ParInflate() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      cv_.notify_all();
    }
    for (auto& w : workers_) w.join();
  }
  int n_chunks() const { return n_; }
  // chunk i, decoded (blocks); nullptr when cancelled
  Chunk* wait(int i) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return ready_[(size_t)i] || stop_; });
    return ready_[(size_t)i] ? &chunks_[(size_t)i] : nullptr;
  }
  // the coordinator is done with chunks < i: free them, let the workers run further ahead
  void release_upto(int i) {
    std::lock_guard<std::mutex> lk(mu_);
    for (int k = released_; k < i && k < n_; ++k) chunks_[(size_t)k].sym.reset();
    if (i > released_) released_ = i;
    cv_.notify_all();
  }
  // first detected block start at or after chunk j's byte offset, searched over at most two chunk lengths (blocks are
  // tens of KB apart: nothing within 8 MiB means stored / fixed-Huffman / non-text data, which is decoded serially);
  // memoised.  total_bits = none found.
  uint64_t start_of(int j) {
    if (j <= 0) return 0;
    if (j >= n_) return total_bits_;
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return start_state_[(size_t)j] != 1 || stop_; });
      if (start_state_[(size_t)j] == 2 || stop_) return start_state_[(size_t)j] == 2 ? starts_[(size_t)j] : total_bits_;
      start_state_[(size_t)j] = 1;
    }
    const uint64_t from = (uint64_t)j * chunk_bytes_ * 8;
    const uint64_t s = scan(from, from + 2 * (uint64_t)chunk_bytes_ * 8);
    std::lock_guard<std::mutex> lk(mu_);
    starts_[(size_t)j] = s;
    start_state_[(size_t)j] = 2;
    // the first start at or after offset j is also the first one for every later chunk whose offset it lies behind
    for (int k = j + 1; k < n_ && s < total_bits_ && (uint64_t)k * chunk_bytes_ * 8 <= s; ++k) {
      if (start_state_[(size_t)k] == 0) {
        starts_[(size_t)k] = s;
        start_state_[(size_t)k] = 2;
      }
    }
    cv_.notify_all();
    return s;
  }

 private:
  const uint8_t* origin_;
  const uint8_t* end_;
  size_t chunk_bytes_;
  std::function<bool()> cancelled_;
  std::atomic<int64_t>* cpu_ns_ = nullptr;
  uint64_t total_bits_ = 0;
  int n_ = 0, next_ = 0, released_ = 0, window_ = 0;
  std::atomic<bool> stop_{false};  // set by the destructor; scan() and decode() poll it so that speculative work past the
                                   // end of a member (the range covers the rest of the file) is dropped promptly
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<uint64_t> starts_;
  std::vector<char> start_state_;  // 0 unknown, 1 being computed, 2 known
  std::vector<Chunk> chunks_;
  std::vector<char> ready_;
  std::vector<std::thread> workers_;

  static uint64_t load64(const uint8_t* p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
  }
  uint64_t scan(uint64_t from_bit, uint64_t limit_bit) {
    std::unique_ptr<dcio::FastInflate> probe(new dcio::FastInflate);
    std::vector<uint16_t> scratch(32768 + TRIAL + dcio::FastInflate::SLACK + 16);
    // a header needs 17 bits + up to 19 x 3: stop while 16 readable bytes remain (the tail belongs to the previous chunk)
    uint64_t last = total_bits_ > 160 ? total_bits_ - 160 : 0;
    if (limit_bit < last) last = limit_bit;
    unsigned tick = 0;
    for (uint64_t p = from_bit; p < last; ++p) {
      if ((p & 0xFFFF) == 0 && stop_.load(std::memory_order_relaxed)) break;
      const uint8_t* b8 = origin_ + (p >> 3);
      const uint64_t b = load64(b8) >> (p & 7);  // >= 57 valid bits
      if ((b & 7) != 4) continue;                // BFINAL = 0, BTYPE = 2
      if (((b >> 3) & 31) > 29 || ((b >> 8) & 31) > 29) continue;
      const int hclen = (int)((b >> 13) & 15) + 4;
      // code-length code: complete prefix code over lengths 1..7 <=> sum 2^(7 - len) == 128
      int kraft = 0;
      const int n1 = hclen < 13 ? hclen : 13;  // 17 + 13 * 3 = 56 bits from the first word
      uint64_t w = b >> 17;
      for (int k = 0; k < n1; ++k, w >>= 3)
        if (w & 7) kraft += 128 >> (w & 7);
      if (hclen > 13) {
        const uint64_t p2 = p + 56;
        uint64_t w2 = load64(origin_ + (p2 >> 3)) >> (p2 & 7);
        for (int k = 13; k < hclen; ++k, w2 >>= 3)
          if (w2 & 7) kraft += 128 >> (w2 & 7);
      }
      if (kraft != 128) continue;
      if (probe->probe_text_block(origin_, p, end_, scratch.data(), TRIAL)) return p;
      if ((++tick & 63) == 0 && (stop_.load(std::memory_order_relaxed) || cancelled_())) break;
    }
    return total_bits_;
  }

  void work() {
    std::unique_ptr<dcio::FastInflate> inf(new dcio::FastInflate);
    for (;;) {
      int i;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || next_ >= n_ || next_ < released_ + window_; });
        if (stop_ || next_ >= n_) return;
        i = next_++;
      }
      Chunk c;
      c.start_bit = start_of(i);
      const uint64_t stop_bit = start_of(i + 1);
      if (c.start_bit >= stop_bit && i > 0) {
        c.end_bit = c.start_bit;  // no block starts inside this chunk: the previous chunk runs through it
      } else {
        decode(*inf, c, i + 1 < n_ ? stop_bit : dcio::FastInflate::NO_STOP);
      }
      std::lock_guard<std::mutex> lk(mu_);
      chunks_[(size_t)i] = std::move(c);
      ready_[(size_t)i] = 1;
      cv_.notify_all();
    }
  }

  void decode(dcio::FastInflate& inf, Chunk& c, uint64_t stop_bit) {
    if (!inf.seek_bit(origin_, c.start_bit, end_)) { c.failed = true; return; }
    constexpr size_t PAD = dcio::FastInflate::SLACK + 16;
    size_t cap = 32768 + chunk_bytes_ * 3;
    c.sym.reset(new uint16_t[cap + PAD]);
    for (int j = 0; j < 32768; ++j) c.sym[(size_t)j] = (uint16_t)(256 + j);
    size_t n = 32768;
    unsigned tick = 0;
    for (;;) {
      uint16_t* out = c.sym.get() + n;
      const dcio::FastInflate::Status st = inf.run_t<uint16_t>(out, c.sym.get() + cap, (uint64_t)n, stop_bit);
      n = (size_t)(out - c.sym.get());
      if (st == dcio::FastInflate::NEED_OUTPUT) {
        if (stop_.load(std::memory_order_relaxed) || ((++tick & 3) == 0 && cancelled_())) { c.failed = true; return; }
        // far more output than expected (or no later start was found): end the chunk at the next block boundary,
        // the coordinator carries on from there
        if (n - 32768 > SOFT_MAX && stop_bit > inf.bit_pos()) stop_bit = inf.bit_pos();
        const size_t ncap = cap + cap / 2;
        std::unique_ptr<uint16_t[]> bigger(new uint16_t[ncap + PAD]);
        memcpy(bigger.get(), c.sym.get(), n * sizeof(uint16_t));
        c.sym = std::move(bigger);
        cap = ncap;
        continue;
      }
      if (st == dcio::FastInflate::BLOCK_STOP || st == dcio::FastInflate::MEMBER_END) {
        c.ended = st == dcio::FastInflate::MEMBER_END;
        c.end_bit = c.ended ? (uint64_t)(inf.input_pos() - origin_) * 8 : inf.bit_pos();
        c.n_out = n - 32768;
        return;
      }
      c.failed = true;
      return;
    }
  }
};

struct Source {
  FILE* fp = nullptr;
  bool compressed = false;
  int depth = 4;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv_put, cv_get;
  std::deque<Block> q;
  bool done = false, cancel = false;
  std::string err;
  std::atomic<int64_t> consumed{0};
  std::atomic<bool> fast_crc{false};  // blocks come from the in-house inflater: the consumer keeps the running CRC-32
  std::atomic<int64_t> helper_cpu_ns{0};  // CPU time of the reading / inflating threads that have finished (dc_fq_helper_cpu_seconds)
  // BGZF byte-range shards: compressed offset the stream starts at / the block boundary that ends the
  // range; range_end_u = (decompressed bytes emitted up to that boundary) + 1, published once reached
  int64_t start_off = 0, end_off = INT64_MAX;
  std::atomic<int64_t> range_end_u{INT64_MAX};

  void push(std::vector<uint8_t>&& b) {
    Block blk;
    blk.data = std::move(b);
    push(std::move(blk));
  }
  void push(Block&& b) {
    std::unique_lock<std::mutex> lk(mu);
    cv_put.wait(lk, [&] { return (int)q.size() < depth || cancel; });
    if (cancel) return;
    q.emplace_back(std::move(b));
    cv_get.notify_one();
  }
  void finish(const std::string& e) {
    std::lock_guard<std::mutex> lk(mu);
    err = e;
    done = true;
    cv_get.notify_all();
  }
  bool cancelled() {
    std::lock_guard<std::mutex> lk(mu);
    return cancel;
  }
  // returns false at end of stream (err set on failure)
  bool pop(Block& out, std::string& e) {
    std::unique_lock<std::mutex> lk(mu);
    cv_get.wait(lk, [&] { return !q.empty() || done; });
    if (!q.empty()) {
      out = std::move(q.front());
      q.pop_front();
      cv_put.notify_one();
      return true;
    }
    e = err;
    return false;
  }

  void run_plain() {
    for (;;) {
      if (cancelled()) break;
      std::vector<uint8_t> b(kBlock);
      size_t n = fread(b.data(), 1, b.size(), fp);
      if (n == 0) {
        if (ferror(fp)) return finish("read error");
        break;
      }
      consumed += (int64_t)n;
      b.resize(n);
      push(std::move(b));
    }
    finish("");
  }

  void run_gzip() {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 32) != Z_OK) return finish("inflateInit2 failed");
    std::vector<uint8_t> in(1u << 20);
    std::vector<uint8_t> out(kBlock);
    size_t have = 0;
    int64_t pushed = 0, fed = 0;  // decompressed bytes already queued / compressed bytes read from the file
    bool member_open = false;  // inside a gzip member
    bool eof = false;
    std::string e;
    for (;;) {
      if (cancelled()) break;
      if (zs.avail_in == 0 && !eof) {
        size_t n = fread(in.data(), 1, in.size(), fp);
        if (n == 0) {
          if (ferror(fp)) { e = "read error"; break; }
          eof = true;
        }
        consumed += (int64_t)n;
        fed += (int64_t)n;
        zs.next_in = in.data();
        zs.avail_in = (uInt)n;
      }
      if (zs.avail_in == 0 && eof) {
        if (member_open) e = "unexpected end of gzip stream";
        break;
      }
      if (!member_open) {
        // trailing zero padding after the last member is tolerated (as gzip(1) does)
        bool all_zero = true;
        for (uInt i = 0; i < zs.avail_in; ++i) if (zs.next_in[i]) { all_zero = false; break; }
        if (all_zero) { zs.avail_in = 0; continue; }
      }
      zs.next_out = out.data() + have;
      zs.avail_out = (uInt)(out.size() - have);
      member_open = true;
      int rc = inflate(&zs, Z_NO_FLUSH);
      have = out.size() - zs.avail_out;
      if (rc == Z_STREAM_END) {
        member_open = false;
        inflateReset(&zs);
        if (range_end_u.load() == INT64_MAX && start_off + fed - (int64_t)zs.avail_in >= end_off)
          range_end_u.store(pushed + (int64_t)have + 1);
      } else if (rc != Z_OK && rc != Z_BUF_ERROR) {
        e = std::string("inflate failed: ") + (zs.msg ? zs.msg : "data error");
        break;
      }
      if (have == out.size()) {
        pushed += (int64_t)have;
        push(std::move(out));
        out.assign(kBlock, 0);
        have = 0;
      }
    }
    inflateEnd(&zs);
    if (e.empty() && have) {
      out.resize(have);
      push(std::move(out));
    }
    finish(e);
  }

  // The memory-mapped input counts towards the process's resident set as it is touched: pages behind the decoder are
  // handed back every 16 MiB so that a multi-GB file does not show up as multi-GB of RSS.
  const uint8_t* map_base_ = nullptr;
  size_t map_dropped_ = 0;
  void drop_input_behind(const uint8_t* pos) {
    if (!map_base_ || pos <= map_base_) return;
    const size_t upto = ((size_t)(pos - map_base_)) & ~(size_t)4095;
    if (upto >= map_dropped_ + (16u << 20)) {
      madvise((void*)(map_base_ + map_dropped_), upto - map_dropped_, MADV_DONTNEED);
      map_dropped_ = upto;
    }
  }

  static void append_window(std::vector<uint8_t>& win, const uint8_t* d, size_t n) {
    constexpr size_t W = 32768;
    if (n >= W) {
      win.assign(d + n - W, d + n);
    } else {
      if (win.size() + n > W) win.erase(win.begin(), win.begin() + (ptrdiff_t)(win.size() + n - W));
      win.insert(win.end(), d, d + n);
    }
  }

  // Coordinator of ParInflate for the member whose DEFLATE stream starts at q.  Pushes the member's bytes in order
  // (the last block carries the CRC mark).  Returns false with `e` set on a data error; trailer = nullptr when
  // cancelled; otherwise trailer points at the member's CRC-32 / ISIZE.
  bool inflate_member_parallel(const uint8_t* q, const uint8_t* fend, int threads, size_t chunk_bytes, uint64_t& member_out,
                               const uint8_t*& trailer, int64_t& pushed, std::string& e) {
    ParInflate par(q, fend, threads, chunk_bytes, [this] { return cancelled(); }, &helper_cpu_ns);
    std::vector<uint8_t> win;  // last <= 32 KiB of this member's output
    std::vector<uint8_t> tab(256 + 32768);
    for (int k = 0; k < 256; ++k) tab[(size_t)k] = (uint8_t)k;
    uint64_t expected = 0;  // bit position (from q) where the next confirmed block starts
    bool ended = false;
    std::unique_ptr<dcio::FastInflate> seq;
    auto finish_member = [&](Block& blk) -> bool {  // trailer check + CRC mark on the member's last block
      const uint8_t* t = q + (expected >> 3);
      if (fend - t < 8) { e = "unexpected end of gzip stream"; return false; }
      const uint32_t want_crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
      const uint32_t want_len = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
      if (want_len != (uint32_t)member_out) { e = "inflate failed: incorrect length check"; return false; }
      blk.marks.push_back(Mark{(uint32_t)blk.data.size(), want_crc, want_len});
      trailer = t;
      return true;
    };
    auto deliver = [&](Block& blk) -> bool {  // bookkeeping + trailer check on the member's last block, then queue it
      append_window(win, blk.data.data(), blk.data.size());
      member_out += blk.data.size();
      pushed += (int64_t)blk.data.size();
      if (ended && !finish_member(blk)) return false;
      push(std::move(blk));
      drop_input_behind(q + (expected >> 3));  // workers only run ahead of the confirmed position
      return true;
    };
    const int n = par.n_chunks();
    int i = 0;
    uint64_t n_par = 0, n_ser = 0, b_par = 0, b_ser = 0;  // $DC_IO_DEBUG: how the member was decoded
    while (!ended) {
      if (cancelled()) return true;  // trailer stays null
      // chunks whose detected start lies behind the confirmed position, or that contain no block start, are skipped
      while (i < n && (par.start_of(i) < expected || par.start_of(i) >= par.start_of(i + 1))) ++i;
      par.release_upto(i);
      if (i < n && par.start_of(i) == expected) {
        ParInflate::Chunk* c = par.wait(i);
        if (!c) return true;
        if (!c->failed) {
          // confirmed start: resolve the placeholders through (identity | previous 32 KiB)
          const size_t w = win.size();
          memset(tab.data() + 256, 0, 32768);
          if (w) memcpy(tab.data() + 256 + (32768 - w), win.data(), w);
          const unsigned min_ok = 256 + (unsigned)(32768 - w);
          Block blk;
          blk.data.resize(c->n_out);
          const uint16_t* sy = c->sym.get() + 32768;
          uint8_t* o = blk.data.data();
          unsigned bad = 0;
          for (size_t k = 0; k < c->n_out; ++k) {
            const unsigned v = sy[k];
            o[k] = tab[v];
            bad |= (unsigned)(v >= 256) & (unsigned)(v < min_ok);
          }
          if (bad) { e = "inflate failed: invalid distance too far back"; return false; }
          expected = c->end_bit;
          ended = c->ended;
          par.release_upto(++i);
          ++n_par;
          b_par += blk.data.size();
          if (!deliver(blk)) return false;
          continue;
        }
        ++i;  // decode error behind a confirmed start: the serial decode below reports it (or gets past a decoder limit)
      }
      // serial stretch from the confirmed position to the next detected start beyond it (an unconfirmed start, a
      // chunk that stopped early, a stream in which no start was found): streamed in blocks like the serial path
      int j = i;
      while (j < n && (par.start_of(j) <= expected || par.start_of(j) >= par.start_of(j + 1))) ++j;
      const uint64_t stop = j < n ? par.start_of(j) : dcio::FastInflate::NO_STOP;
      ++n_ser;
      if (!seq) seq.reset(new dcio::FastInflate);
      if (!seq->seek_bit(q, expected, fend)) { e = "unexpected end of gzip stream"; return false; }
      const size_t cap = kBlock;
      std::vector<uint8_t> rb(32768 + cap + dcio::FastInflate::SLACK + 16);
      size_t w = win.size();
      if (w) memcpy(rb.data() + 32768 - w, win.data(), w);
      uint8_t* o = rb.data() + 32768;
      for (;;) {
        const size_t before = (size_t)(o - (rb.data() + 32768));
        const dcio::FastInflate::Status st = seq->run_t<uint8_t>(o, rb.data() + 32768 + cap, (uint64_t)(w + before), stop);
        const size_t have = (size_t)(o - (rb.data() + 32768));
        if (st == dcio::FastInflate::NEED_OUTPUT) {
          if (cancelled()) return true;
          Block blk;
          blk.data.assign(rb.data() + 32768, rb.data() + 32768 + have);
          b_ser += have;
          if (!deliver(blk)) return false;
          memmove(rb.data(), rb.data() + have, 32768);  // have >= cap >= 32 KiB: the last 32 KiB become the history
          w = 32768;
          o = rb.data() + 32768;
          continue;
        }
        if (st == dcio::FastInflate::BLOCK_STOP || st == dcio::FastInflate::MEMBER_END) {
          ended = st == dcio::FastInflate::MEMBER_END;
          expected = ended ? (uint64_t)(seq->input_pos() - q) * 8 : seq->bit_pos();
          Block blk;
          blk.data.assign(rb.data() + 32768, rb.data() + 32768 + have);
          b_ser += have;
          if (!deliver(blk)) return false;
          break;
        }
        e = st == dcio::FastInflate::ERR_TRUNCATED ? "unexpected end of gzip stream" : std::string("inflate failed: ") + seq->error();
        return false;
      }
      i = j;
    }
    if (getenv("DC_IO_DEBUG"))
      fprintf(stderr, "[dc_io] parallel inflate: %d chunks, %llu decoded by workers (%llu bytes), %llu serial stretches (%llu bytes)\n", n,
              (unsigned long long)n_par, (unsigned long long)b_par, (unsigned long long)n_ser, (unsigned long long)b_ser);
    return true;
  }

  // gzip / BGZF through the in-house decoder (fast_inflate.h) on a memory-mapped file.  Same contract as run_gzip:
  // multi-member streams, trailing zero padding, CRC-32 and length of every member checked, BGZF range ends.
  bool run_fast() {
    const int fd = fileno(fp);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size <= 0) return false;
    const size_t fsize = (size_t)sb.st_size;
    void* map = mmap(nullptr, fsize, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) return false;
    madvise(map, fsize, MADV_SEQUENTIAL);
    fast_crc.store(true);
    const uint8_t* const base = (const uint8_t*)map;
    map_base_ = base;
    map_dropped_ = 0;
    const uint8_t* p = base + (start_off > 0 && (size_t)start_off < fsize ? (size_t)start_off : 0);
    if (start_off >= (int64_t)fsize) p = base + fsize;
    const uint8_t* const fend = base + fsize;
    constexpr size_t WIN = 32768;
    std::vector<uint8_t> buf(WIN + kBlock + dcio::FastInflate::SLACK + 64);
    uint8_t* const out0 = buf.data() + WIN;
    uint8_t* out = out0;
    uint8_t* const out_limit = out0 + kBlock;
    std::unique_ptr<dcio::FastInflate> inf(new dcio::FastInflate);
    int64_t pushed = 0;
    std::string e;
    std::vector<Mark> marks;
    // parallel inflate of large single members: $DC_IO_INFLATE_THREADS (default 4, fewer on small hosts; 1 = off),
    // $DC_IO_INFLATE_CHUNK compressed bytes per chunk (default 4 MiB; a member shorter than four chunks stays serial)
    int par_threads = 4;
    {
      const unsigned hw = std::thread::hardware_concurrency();
      if (hw && hw < 16) par_threads = hw >= 8 ? 2 : 1;
      if (const char* v = getenv("DC_IO_INFLATE_THREADS")) par_threads = atoi(v);
      if (par_threads > 32) par_threads = 32;
    }
    size_t par_chunk = 4u << 20;
    if (const char* v = getenv("DC_IO_INFLATE_CHUNK")) {
      const long c = atol(v);
      if (c >= 4096) par_chunk = (size_t)c;
    }
    bool par_allowed = true;
    auto flush = [&](bool all) {  // hand over [out0, out); keep the last 32 KiB as history
      const size_t have = (size_t)(out - out0);
      if (have == 0 && marks.empty()) return;
      if (!all && have < kBlock) return;
      Block blk;
      blk.data.assign(out0, out);
      blk.marks.swap(marks);
      pushed += (int64_t)have;
      push(std::move(blk));
      const size_t keep = have < WIN ? have : WIN;
      memmove(out0 - keep, out - keep, keep);
      out = out0;
    };
    while (p < fend && !cancelled()) {
      // between members: zero padding after the last one is tolerated (as gzip(1) does)
      if (*p == 0) {
        const uint8_t* q = p;
        while (q < fend && *q == 0) ++q;
        if (q == fend) {
          consumed.store((int64_t)(fend - base) - (start_off > 0 ? start_off : 0));
          break;
        }
        e = "inflate failed: invalid data after the gzip member";
        break;
      }
      // ---- gzip member header (RFC 1952) ----
      if (fend - p < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8) { e = "inflate failed: incorrect header check"; break; }
      const unsigned flg = p[3];
      const uint8_t* q = p + 10;
      bool ok = true;
      if (flg & 4) {  // FEXTRA
        if (fend - q < 2) ok = false;
        else {
          const size_t xlen = q[0] | (q[1] << 8);
          q += 2;
          if ((size_t)(fend - q) < xlen) ok = false;
          else q += xlen;
        }
      }
      for (int f = 8; ok && f <= 16; f <<= 1) {  // FNAME, FCOMMENT: zero-terminated
        if (flg & f) {
          while (q < fend && *q) ++q;
          if (q >= fend) ok = false;
          else ++q;
        }
      }
      if (ok && (flg & 2)) {  // FHCRC
        if (fend - q < 2) ok = false;
        else q += 2;
      }
      if (!ok) { e = "unexpected end of gzip stream"; break; }
      uint64_t member_out = 0;
      bool member_done = false;
      // ---- one large member: decode it on several threads (ParInflate above) ----
      if (par_threads > 1 && par_allowed && !(flg & 4) && start_off <= 0 && end_off == INT64_MAX &&
          (size_t)(fend - q) >= 4 * par_chunk) {
        flush(true);
        const uint8_t* trailer = nullptr;
        if (!inflate_member_parallel(q, fend, par_threads, par_chunk, member_out, trailer, pushed, e)) break;
        if (trailer == nullptr) break;  // cancelled
        if ((size_t)(trailer - q) < par_chunk) par_allowed = false;  // many small members: not worth speculating again
        p = trailer + 8;
        consumed.store((int64_t)(p - base));
        continue;
      }
      inf->set_input(q, fend);
      inf->begin_member();
      uint8_t* seg = out;  // start of the not yet check-summed output of this member
      while (!cancelled()) {
        const dcio::FastInflate::Status st = inf->run(out, out_limit, member_out + (uint64_t)(out - seg));
        if (st == dcio::FastInflate::NEED_OUTPUT || st == dcio::FastInflate::MEMBER_END) {
          member_out += (uint64_t)(out - seg);
          if (st == dcio::FastInflate::MEMBER_END) { member_done = true; break; }
          flush(false);
          drop_input_behind(inf->input_pos() - 16);
          seg = out;
          continue;
        }
        e = st == dcio::FastInflate::ERR_TRUNCATED ? "unexpected end of gzip stream" : std::string("inflate failed: ") + inf->error();
        break;
      }
      if (!member_done) break;
      const uint8_t* t = inf->input_pos();
      if (fend - t < 8) { e = "unexpected end of gzip stream"; break; }
      const uint32_t want_crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
      const uint32_t want_len = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
      if (want_len != (uint32_t)member_out) { e = "inflate failed: incorrect length check"; break; }
      marks.push_back(Mark{(uint32_t)(out - out0), want_crc, want_len});  // CRC-32: checked by the consumer
      p = t + 8;
      consumed.store((int64_t)(p - base) - (start_off > 0 ? start_off : 0));
      if (range_end_u.load() == INT64_MAX && (int64_t)(p - base) >= end_off)
        range_end_u.store(pushed + (int64_t)(out - out0) + 1);
      if (out - out0 >= (ptrdiff_t)kBlock) flush(false);
    }
    if (e.empty()) flush(true);
    munmap(map, fsize);
    finish(e);
    return true;
  }

  void start() {
    th = std::thread([this] {
      struct Account {  // this thread's CPU time, added when it ends (whichever way it returns)
        std::atomic<int64_t>& to;
        ~Account() { to.fetch_add(ParInflate::thread_cpu_ns()); }
      } account{helper_cpu_ns};
      if (!compressed) return run_plain();
      const char* sel = getenv("DC_IO_INFLATE");
      if (sel && strcmp(sel, "zlib") == 0) return run_gzip();
      if (!run_fast()) run_gzip();  // not a regular file (pipe, ...): stream it through zlib
    });
  }
  ~Source() {
    {
      std::lock_guard<std::mutex> lk(mu);
      cancel = true;
      cv_put.notify_all();
    }
    if (th.joinable()) th.join();
    if (fp) fclose(fp);
  }
};

int sniff(const char* path, int* type) {
  FILE* fp = fopen(path, "rb");
  if (!fp) return fail(DC_FQ_ERR_IO, "cannot open '%s': %s", path, strerror(errno));
  unsigned char b[18] = {0};
  size_t n = fread(b, 1, sizeof b, fp);
  fclose(fp);
  int t = DC_CT_UNKNOWN;
  if (n < 2) t = DC_CT_UNCOMPRESS;
  else if (n >= 18 && b[0] == 0x1f && b[1] == 0x8b && b[2] == 0x08 && b[3] == 0x04) {
    unsigned xlen = b[10] | (b[11] << 8);
    t = (xlen >= 6 && b[12] == 'B' && b[13] == 'C' && b[14] == 2 && b[15] == 0) ? DC_CT_BGZIP : DC_CT_GZIP;
  } else if (b[0] == 0x1f && b[1] == 0x8b) t = DC_CT_GZIP;
  else if (n >= 4 && b[0] == 'P' && b[1] == 'K' && ((b[2] == 3 && b[3] == 4) || (b[2] == 5 && b[3] == 6) || (b[2] == 7 && b[3] == 8))) t = DC_CT_ZIP;
  else if (n >= 3 && b[0] == 'B' && b[1] == 'Z' && b[2] == 'h') t = DC_CT_BZIP2;
  else if (n >= 6 && b[0] == 0xfd && b[1] == '7' && b[2] == 'z' && b[3] == 'X' && b[4] == 'Z' && b[5] == 0) t = DC_CT_XZ;
  else if (n >= 4 && b[0] == 0x28 && b[1] == 0xb5 && b[2] == 0x2f && b[3] == 0xfd) t = DC_CT_ZSTD;
  else t = DC_CT_UNCOMPRESS;  // no known signature: treated as plain text
  *type = t;
  return 0;
}

}  // namespace

struct dc_fq_reader {
  Source src;
  std::string path;
  std::vector<uint8_t> buf;  // pending bytes; buf[pos..] not yet parsed
  size_t pos = 0;
  int64_t base = 0;          // absolute (plain-file) offset of buf[0]
  bool eof = false;
  bool ranged = false, synced = true;
  int64_t range_begin = 0, range_end = INT64_MAX;
  int64_t n_records = 0;
  bool finished = false;
  bool strict4 = false;  // dc_fq_set_strict: records are exactly four lines (the reference's chop reads with noodles)

  int64_t range_limit() const { return src.compressed ? src.range_end_u.load() : range_end; }

  // Drops the parsed prefix.  Only called between records, so indices held while a record is being
  // parsed stay valid (more() appends, never moves).
  void compact() {
    if (pos >= kBlock) {
      buf.erase(buf.begin(), buf.begin() + (ptrdiff_t)pos);
      base += (int64_t)pos;
      pos = 0;
    }
  }

  // appends the next block; false at end of stream (rc set on a source error)
  uint32_t crc_run = 0;  // CRC-32 of the current gzip member so far (members checked on this thread)
  bool more(int& rc) {
    if (eof) return false;
    Block b;
    std::string e;
    if (!src.pop(b, e)) {
      eof = true;
      if (!e.empty()) rc = fail(DC_FQ_ERR_IO, "%s: %s", path.c_str(), e.c_str());
      return false;
    }
    if (!b.marks.empty()) {
      size_t pos = 0;
      for (const Mark& m : b.marks) {
        crc_run = (uint32_t)crc32(crc_run, b.data.data() + pos, (uInt)(m.end - pos));
        pos = m.end;
        if (crc_run != m.crc) {
          eof = true;
          rc = fail(DC_FQ_ERR_IO, "%s: inflate failed: incorrect data check", path.c_str());
          return false;
        }
        crc_run = 0;
      }
      crc_run = (uint32_t)crc32(crc_run, b.data.data() + pos, (uInt)(b.data.size() - pos));
    } else if (src.fast_crc) {
      crc_run = (uint32_t)crc32(crc_run, b.data.data(), (uInt)b.data.size());
    }
    buf.insert(buf.end(), b.data.begin(), b.data.end());
    return true;
  }

  // End of the line starting at `from`: *eol = index of its '\n', or buf.size() for an unterminated
  // last line.  false = no line starts at `from` (end of input) or a source error (rc != 0).
  bool line_end(size_t from, size_t* eol, int& rc) {
    size_t scan = from;
    for (;;) {
      if (scan < buf.size()) {
        const void* p = memchr(buf.data() + scan, '\n', buf.size() - scan);
        if (p) { *eol = (size_t)((const uint8_t*)p - buf.data()); return true; }
        scan = buf.size();
      }
      if (!more(rc)) {
        if (rc || from >= buf.size()) return false;
        *eol = buf.size();
        return true;
      }
    }
  }

  size_t next_line(size_t eol) const { return eol < buf.size() ? eol + 1 : eol; }

  // Byte-range shards: buf[pos] is file byte range_begin-1.  Move to the first line that starts at or
  // after range_begin and begins a record: an '@' line whose next-but-one line starts with '+'
  // (a quality line may start with '@', but then the next-but-one line is a sequence, never '+').
  int sync_to_record(int& rc) {
    size_t e;
    if (!line_end(pos, &e, rc)) { finished = true; return rc; }
    pos = next_line(e);
    for (;;) {
      compact();
      size_t e1, e2, e3;
      if (!line_end(pos, &e1, rc)) { finished = true; return rc; }
      if (buf[pos] == '@' && e1 < buf.size() && line_end(e1 + 1, &e2, rc) && e2 < buf.size() &&
          line_end(e2 + 1, &e3, rc) && buf[e2 + 1] == '+')
        break;
      if (rc) return rc;
      pos = next_line(e1);
    }
    synced = true;
    return 0;
  }
};

extern "C" {

const char* dc_io_last_error(void) { return g_err.c_str(); }

int dc_io_check_compressed_type(const char* path, int* type_out) {
  if (!path || !type_out) return fail(DC_FQ_ERR_ARG, "null argument");
  return sniff(path, type_out);
}

// First BGZF block boundary at or after `want` (hopping block headers from the start of the file:
// each header carries BSIZE = block length - 1 at bytes 16-17).  File size if none.
static int bgzf_snap(FILE* fp, int64_t want, int64_t fsize, int64_t* snapped) {
  int64_t off = 0;
  unsigned char h[18];
  while (off < want && off < fsize) {
    if (fseeko(fp, (off_t)off, SEEK_SET) != 0 || fread(h, 1, 18, fp) != 18) return fail(DC_FQ_ERR_IO, "truncated BGZF block header at %lld", (long long)off);
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || h[3] != 4 || h[12] != 'B' || h[13] != 'C')
      return fail(DC_FQ_ERR_IO, "not a BGZF block at offset %lld", (long long)off);
    off += (int64_t)(h[16] | (h[17] << 8)) + 1;
  }
  *snapped = off < fsize ? off : fsize;
  return 0;
}

static int open_impl(const char* path, int64_t b0, int64_t b1, bool ranged, int readahead, dc_fq_reader** out) {
  if (!path || !out) return fail(DC_FQ_ERR_ARG, "null argument");
  int t = 0;
  if (int rc = sniff(path, &t)) return rc;
  if (t != DC_CT_UNCOMPRESS && t != DC_CT_GZIP && t != DC_CT_BGZIP)
    return fail(DC_FQ_ERR_IO, "unsupported compression type");  // writefq.rs:159
  if (ranged && t == DC_CT_GZIP) return fail(DC_FQ_ERR_ARG, "byte-range shards need an uncompressed or BGZF FASTQ (plain gzip cannot be entered mid-stream)");
  if (ranged && (b0 < 0 || b1 < b0)) return fail(DC_FQ_ERR_ARG, "invalid byte range [%lld, %lld)", (long long)b0, (long long)b1);
  std::unique_ptr<dc_fq_reader> r(new dc_fq_reader);
  r->path = path;
  r->src.fp = fopen(path, "rb");
  if (!r->src.fp) return fail(DC_FQ_ERR_IO, "cannot open '%s': %s", path, strerror(errno));
  r->src.compressed = t != DC_CT_UNCOMPRESS;
  r->src.depth = readahead > 0 ? readahead : 4;
  if (ranged && t == DC_CT_UNCOMPRESS) {
    // ownership rule: a record belongs to the range that holds the byte BEFORE its '@' (record 0 to the
    // range that starts at 0), so reading starts one byte early and ranges tile without gaps or overlap
    r->ranged = true;
    r->range_begin = b0;
    r->range_end = b1;
    int64_t start = b0 > 0 ? b0 - 1 : 0;
    if (fseeko(r->src.fp, (off_t)start, SEEK_SET) != 0) return fail(DC_FQ_ERR_IO, "seek failed in '%s'", path);
    r->base = start;
    r->synced = b0 == 0;
  } else if (ranged) {
    // BGZF: the range is snapped to block boundaries (every rank computes the same snapping); positions
    // are then decompressed offsets relative to the first block of the range, same ownership rule
    if (fseeko(r->src.fp, 0, SEEK_END) != 0) return fail(DC_FQ_ERR_IO, "seek failed in '%s'", path);
    int64_t fsize = (int64_t)ftello(r->src.fp), s0 = 0, s1 = 0;
    if (int rc = bgzf_snap(r->src.fp, b0, fsize, &s0)) return rc;
    if (int rc = bgzf_snap(r->src.fp, b1, fsize, &s1)) return rc;
    if (fseeko(r->src.fp, (off_t)s0, SEEK_SET) != 0) return fail(DC_FQ_ERR_IO, "seek failed in '%s'", path);
    r->ranged = true;
    r->src.start_off = s0;
    r->src.end_off = s1;
    if (s1 <= s0) r->src.range_end_u.store(0);  // empty range
    r->base = 0;  // stream byte 0 plays the part of "range_begin - 1"
    r->synced = s0 == 0;
  }
  r->src.start();
  *out = r.release();
  return 0;
}

int dc_fq_open(const char* path, int readahead, dc_fq_reader** out) { return open_impl(path, 0, 0, false, readahead, out); }

int dc_fq_open_range(const char* path, int64_t byte_begin, int64_t byte_end, int readahead, dc_fq_reader** out) {
  return open_impl(path, byte_begin, byte_end, true, readahead, out);
}

int64_t dc_fq_records_read(const dc_fq_reader* r) { return r ? r->n_records : 0; }
int dc_fq_set_strict(dc_fq_reader* r, int strict_four_line) {
  if (!r) return fail(DC_FQ_ERR_ARG, "dc_fq_set_strict: null reader");
  r->strict4 = strict_four_line != 0;
  return 0;
}
int64_t dc_fq_bytes_consumed(const dc_fq_reader* r) { return r ? r->src.consumed.load() : 0; }
double dc_fq_helper_cpu_seconds(const dc_fq_reader* r) { return r ? 1e-9 * (double)r->src.helper_cpu_ns.load() : 0.0; }
void dc_fq_close(dc_fq_reader* r) { delete r; }

// name_len == nullptr: `names` receives the name part of every definition line (up to the first blank or
// tab).  Otherwise `names` receives the whole definition line after '@' and name_len[r] the length of its
// name part.
static int read_chunk_impl(dc_fq_reader* r, int64_t max_reads, uint8_t* seq, uint8_t* qual, int64_t cap_bases,
                           int64_t* offsets, char* names, int64_t cap_names, int64_t* name_offsets, int32_t* name_len,
                           int64_t* n_reads) {
  if (!r || !seq || !qual || !offsets || !names || !name_offsets || !n_reads || max_reads < 0 || cap_bases < 0 || cap_names < 0)
    return fail(DC_FQ_ERR_ARG, "invalid argument");
  int rc = 0;
  int64_t n = 0, nb = 0, nn = 0;
  offsets[0] = 0;
  name_offsets[0] = 0;
  *n_reads = 0;
  if (r->ranged && !r->synced && !r->finished) {
    if (r->sync_to_record(rc)) return rc;
  }
  while (n < max_reads && !r->finished) {
    r->compact();
    // skip blank lines between records
    size_t e[4];
    for (;;) {
      if (!r->line_end(r->pos, &e[0], rc)) {
        if (rc) return rc;
        r->finished = true;
        break;
      }
      size_t len = e[0] - r->pos;
      if (len == 0 || (len == 1 && r->buf[r->pos] == '\r')) { r->pos = r->next_line(e[0]); continue; }
      break;
    }
    if (r->finished) break;
    if (r->ranged && r->base + (int64_t)r->pos >= r->range_limit()) { r->finished = true; break; }
    bool ok = true;
    for (int k = 1; k < 4 && ok; ++k) ok = e[k - 1] < r->buf.size() && r->line_end(e[k - 1] + 1, &e[k], rc);
    if (rc) return rc;
    int64_t recno = r->n_records;
    // Sequence / quality wrapped over several lines (what kseq -- pyfastx.Fastx, deepchopper/data/only_fq.py:34 -- accepts):
    // sequence lines up to the first line that starts with '+', then quality lines until as many quality characters as
    // bases have been read.  Taken only when the third line of the record is not the separator; the 4-line form below
    // is untouched.  Pieces = [begin, end) byte ranges of buf, '\r' stripped.
    std::vector<std::pair<size_t, size_t>> sq_pieces, q_pieces;
    // (also when the third line IS the separator but the quality line is shorter than the sequence line: a one-line sequence
    //  with a wrapped quality.  If that does not add up exactly, the record is judged as the 4-line record it looks like.)
    const bool sep3 = ok && r->buf[e[1] + 1] == '+';
    auto bare = [&](size_t a, size_t b) { while (b > a && r->buf[b - 1] == '\r') --b; return b - a; };
    if (!r->strict4 && ok && r->buf[r->pos] == '@' && (!sep3 || bare(e[2] + 1, e[3]) < bare(e[0] + 1, e[1]))) {
      const size_t e3_four_line = e[3];
      if (r->ranged && sep3) goto four_line;
      if (r->ranged)
        return fail(DC_FQ_ERR_FORMAT, "FASTQ parsing error in %s: record %lld wraps its sequence over several lines; a multi-line FASTQ "
                    "cannot be split by byte ranges (run it on one GPU, or gzip it: a gzip stream is dealt record by record)",
                    r->path.c_str(), (long long)recno);
      auto stripped = [&](size_t a, size_t b) { while (b > a && r->buf[b - 1] == '\r') --b; return b; };
      size_t ls = 0, lq = 0, at = e[0] + 1, le = 0;
      bool sep = false;
      for (;;) {  // sequence lines
        if (!r->line_end(at, &le, rc)) break;
        if (r->buf[at] == '+') { sep = true; break; }
        const size_t b1 = stripped(at, le);
        if (b1 > at) sq_pieces.emplace_back(at, b1), ls += b1 - at;
        if (le >= r->buf.size()) { at = le; break; }
        at = le + 1;
      }
      if (rc) return rc;
      if (!sep)
        return fail(DC_FQ_ERR_FORMAT, "FASTQ parsing error in %s: record %lld has no '+' separator", r->path.c_str(), (long long)recno);
      at = le < r->buf.size() ? le + 1 : le;  // behind the separator line
      size_t last = le;
      while (lq < ls) {  // quality lines (a quality line may start with '@' or '+': only the count ends the record)
        if (!r->line_end(at, &le, rc)) break;
        const size_t b1 = stripped(at, le);
        if (b1 > at) q_pieces.emplace_back(at, b1), lq += b1 - at;
        last = le;
        if (le >= r->buf.size()) break;
        at = le + 1;
      }
      if (rc) return rc;
      e[3] = last;
      const uint8_t* dd = r->buf.data();
      size_t s0w = r->pos, hw = e[0];
      while (hw > s0w && dd[hw - 1] == '\r') --hw;
      size_t nm0w = s0w + 1, nm1w = nm0w;
      while (nm1w < hw && dd[nm1w] != ' ' && dd[nm1w] != '\t') ++nm1w;
      const int64_t lnamew = (int64_t)(nm1w - nm0w);
      int64_t lnw = name_len ? (int64_t)(hw - nm0w) : lnamew;
      if (sep3 && (lnamew == 0 || ls == 0 || ls != lq)) {  // not a wrapped quality after all
        e[3] = e3_four_line;
        goto four_line;
      }
      if (lnamew == 0 || ls == 0 || lq == 0)
        return fail(DC_FQ_ERR_INCOMPLETE, "Incomplete FASTQ record at position %lld in %s", (long long)recno, r->path.c_str());
      if (ls != lq)
        return fail(DC_FQ_ERR_LENGTH, "Sequence/quality length mismatch in record '%.*s': seq=%lld, qual=%lld", (int)lnamew,
                    (const char*)dd + nm0w, (long long)ls, (long long)lq);
      if (nb + (int64_t)ls > cap_bases || nn + lnw > cap_names) {
        if (n == 0) return fail(DC_FQ_ERR_ARG, "record '%.*s' (%lld bases) does not fit the chunk buffers", (int)lnamew, (const char*)dd + nm0w, (long long)ls);
        break;
      }
      int64_t w = nb;
      for (auto& pc : sq_pieces) { memcpy(seq + w, dd + pc.first, pc.second - pc.first); w += (int64_t)(pc.second - pc.first); }
      w = nb;
      for (auto& pc : q_pieces) { memcpy(qual + w, dd + pc.first, pc.second - pc.first); w += (int64_t)(pc.second - pc.first); }
      memcpy(names + nn, dd + nm0w, (size_t)lnw);
      if (name_len) name_len[n] = (int32_t)lnamew;
      nb += (int64_t)ls;
      nn += lnw;
      ++n;
      offsets[n] = nb;
      name_offsets[n] = nn;
      ++r->n_records;
      r->pos = r->next_line(e[3]);
      continue;
    }
  four_line:
    if (!ok) return fail(DC_FQ_ERR_FORMAT, "FASTQ parsing error in %s: truncated record %lld", r->path.c_str(), (long long)recno);
    const uint8_t* d = r->buf.data();
    size_t s0 = r->pos, s1 = e[0] + 1, s2 = e[1] + 1, s3 = e[2] + 1;
    auto strip = [&](size_t a, size_t b) { while (b > a && (d[b - 1] == '\r')) --b; return b; };
    size_t h_end = strip(s0, e[0]), q_end = strip(s3, e[3]), sq_end = strip(s1, e[1]);
    if (d[s0] != '@')
      return fail(DC_FQ_ERR_FORMAT, "FASTQ parsing error in %s: record %lld does not start with '@'", r->path.c_str(), (long long)recno);
    if (s2 >= r->buf.size() || d[s2] != '+')
      return fail(DC_FQ_ERR_FORMAT, "FASTQ parsing error in %s: record %lld has no '+' separator", r->path.c_str(), (long long)recno);
    size_t nm0 = s0 + 1, nm1 = nm0;
    while (nm1 < h_end && d[nm1] != ' ' && d[nm1] != '\t') ++nm1;
    int64_t ls = (int64_t)(sq_end - s1), lq = (int64_t)(q_end - s3), ln = (int64_t)(nm1 - nm0);
    const int64_t lname = ln;
    if (name_len) ln = (int64_t)(h_end - nm0);  // whole definition line
    if (lname == 0 || ls == 0 || lq == 0)
      return fail(DC_FQ_ERR_INCOMPLETE, "Incomplete FASTQ record at position %lld in %s", (long long)recno, r->path.c_str());
    if (ls != lq)
      return fail(DC_FQ_ERR_LENGTH, "Sequence/quality length mismatch in record '%.*s': seq=%lld, qual=%lld", (int)lname,
                  (const char*)d + nm0, (long long)ls, (long long)lq);
    if (nb + ls > cap_bases || nn + ln > cap_names) {
      if (n == 0) return fail(DC_FQ_ERR_ARG, "record '%.*s' (%lld bases) does not fit the chunk buffers", (int)lname, (const char*)d + nm0, (long long)ls);
      break;  // deliver what we have; this record starts the next chunk
    }
    memcpy(seq + nb, d + s1, (size_t)ls);
    memcpy(qual + nb, d + s3, (size_t)ls);
    memcpy(names + nn, d + nm0, (size_t)ln);
    if (name_len) name_len[n] = (int32_t)lname;
    nb += ls;
    nn += ln;
    ++n;
    offsets[n] = nb;
    name_offsets[n] = nn;
    ++r->n_records;
    r->pos = r->next_line(e[3]);
  }
  *n_reads = n;
  return 0;
}

int dc_fq_read_chunk(dc_fq_reader* r, int64_t max_reads, uint8_t* seq, uint8_t* qual, int64_t cap_bases,
                     int64_t* offsets, char* names, int64_t cap_names, int64_t* name_offsets, int64_t* n_reads) {
  return read_chunk_impl(r, max_reads, seq, qual, cap_bases, offsets, names, cap_names, name_offsets, nullptr, n_reads);
}

int dc_fq_read_chunk_full(dc_fq_reader* r, int64_t max_reads, uint8_t* seq, uint8_t* qual, int64_t cap_bases,
                          int64_t* offsets, char* headers, int64_t cap_headers, int64_t* header_offsets,
                          int32_t* name_len, int64_t* n_reads) {
  if (!name_len) return fail(DC_FQ_ERR_ARG, "invalid argument");
  return read_chunk_impl(r, max_reads, seq, qual, cap_bases, offsets, headers, cap_headers, header_offsets, name_len, n_reads);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// BGZF writer: blocks of <= 65280 input bytes, each an independent gzip member with the BC extra
// field; N workers compress, the caller's thread writes finished blocks in submission order.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr size_t kBgzfIn = 0xff00;
const unsigned char kBgzfEof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 'B', 'C', 0x02, 0,
                                    0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

struct Job {
  std::vector<uint8_t> in, out;
  bool done = false, failed = false;
};

// gzip member around `clen` raw-deflate bytes at j.out + 18
void bgzf_frame(Job& j, size_t clen) {
  const size_t total = 18 + clen + 8;
  uint8_t* h = j.out.data();
  const unsigned char head[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 'B', 'C', 0x02, 0};
  memcpy(h, head, 16);
  h[16] = (uint8_t)((total - 1) & 0xff);
  h[17] = (uint8_t)((total - 1) >> 8);
  const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), j.in.data(), (uInt)j.in.size());
  const uint32_t isize = (uint32_t)j.in.size();
  uint8_t* t = h + 18 + clen;
  for (int k = 0; k < 4; ++k) {
    t[k] = (uint8_t)(crc >> (8 * k));
    t[4 + k] = (uint8_t)(isize >> (8 * k));
  }
  j.out.resize(total);
}

// The default level (6, noodles' and flate2's default) goes through this library's own encoder (fast_deflate.h: zlib 6's
// ratio on nanopore FASTQ at 2.5x its speed; deterministic per block, so the output stays byte-identical for any number of
// workers); other levels, or DC_BGZF_DEFLATE=zlib, through zlib.
bool bgzf_compress_fast(dcdeflate::Workspace& ws, Job& j) {
  j.out.resize(18 + j.in.size() + 64 + 8);
  const size_t clen = dcdeflate::compress_block(j.in.data(), j.in.size(), j.out.data() + 18, j.out.size() - 18 - 8, ws);
  if (clen == 0 || 18 + clen + 8 > 65536) return false;
  bgzf_frame(j, clen);
  return true;
}

bool bgzf_compress(z_stream& zs, int level, Job& j) {
  auto attempt = [&](int lvl) -> bool {
    if (deflateParams(&zs, lvl, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    deflateReset(&zs);
    j.out.resize(18 + deflateBound(&zs, (uLong)j.in.size()) + 8);
    zs.next_in = j.in.data();
    zs.avail_in = (uInt)j.in.size();
    zs.next_out = j.out.data() + 18;
    zs.avail_out = (uInt)(j.out.size() - 18 - 8);
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) return false;
    const size_t clen = j.out.size() - 18 - 8 - zs.avail_out;
    if (18 + clen + 8 > 65536) return false;
    bgzf_frame(j, clen);
    return true;
  };
  return attempt(level) || attempt(0);  // incompressible data: stored block still fits (65280 + 5 + 26)
}
}  // namespace

struct dc_bgzf_writer {
  FILE* fp = nullptr;
  int level = 6;
  std::vector<std::thread> workers;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::deque<Job*> todo;                       // waiting for a worker
  std::deque<std::unique_ptr<Job>> inflight;   // submission order
  size_t max_inflight = 8;
  bool stop = false;
  std::unique_ptr<Job> cur;
  bool io_failed = false;
  // marks (dc_bgzf_mark): a mark ends the current block; its file offset is known once every block submitted before
  // it has been written -- recorded by the writing thread, never waited for
  int64_t written = 0;                 // bytes written to fp
  uint64_t n_submitted = 0, n_written = 0;
  double deflate_cpu_s = 0.0;          // CPU seconds the workers spent inside deflate (host-budget accounting)
  std::vector<uint64_t> mark_block;    // blocks submitted before the mark
  std::vector<int64_t> mark_offset;    // file offset of the mark (filled in order)

  void settle_marks() {
    while (mark_offset.size() < mark_block.size() && mark_block[mark_offset.size()] <= n_written) mark_offset.push_back(written);
  }

  void worker() {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    bool ok = deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK;
    const char* which = getenv("DC_BGZF_DEFLATE");
    const bool own = level == 6 && !(which && strcmp(which, "zlib") == 0);
    std::unique_ptr<dcdeflate::Workspace> ws(own ? new dcdeflate::Workspace() : nullptr);
    for (;;) {
      Job* j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_work.wait(lk, [&] { return stop || !todo.empty(); });
        if (todo.empty()) break;
        j = todo.front();
        todo.pop_front();
      }
      timespec c0, c1;
      clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c0);
      bool good = ok && ((own && bgzf_compress_fast(*ws, *j)) || bgzf_compress(zs, level, *j));
      clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c1);
      {
        std::lock_guard<std::mutex> lk(mu);
        j->failed = !good;
        j->done = true;
        deflate_cpu_s += (double)(c1.tv_sec - c0.tv_sec) + 1e-9 * (double)(c1.tv_nsec - c0.tv_nsec);
      }
      cv_done.notify_all();
    }
    if (ok) deflateEnd(&zs);
  }

  // write finished blocks at the head; if `all`, wait until everything in flight is written
  int drain(bool all, size_t keep) {
    for (;;) {
      std::unique_ptr<Job> j;
      {
        std::unique_lock<std::mutex> lk(mu);
        if (inflight.empty()) return 0;
        if (!inflight.front()->done) {
          if (!all && inflight.size() <= keep) return 0;
          cv_done.wait(lk, [&] { return inflight.front()->done; });
        }
        j = std::move(inflight.front());
        inflight.pop_front();
      }
      if (j->failed) return fail(DC_FQ_ERR_IO, "BGZF block compression failed");
      if (fwrite(j->out.data(), 1, j->out.size(), fp) != j->out.size()) {
        io_failed = true;
        return fail(DC_FQ_ERR_IO, "write failed: %s", strerror(errno));
      }
      written += (int64_t)j->out.size();
      ++n_written;
      settle_marks();
    }
  }

  int submit() {
    if (!cur || cur->in.empty()) return 0;
    Job* raw = cur.get();
    {
      std::lock_guard<std::mutex> lk(mu);
      inflight.emplace_back(std::move(cur));
      todo.push_back(raw);
    }
    ++n_submitted;
    cv_work.notify_one();
    cur.reset(new Job);
    cur->in.reserve(kBgzfIn);
    return drain(false, max_inflight);
  }

  int put(const uint8_t* p, size_t n) {
    while (n) {
      size_t room = kBgzfIn - cur->in.size();
      size_t k = n < room ? n : room;
      cur->in.insert(cur->in.end(), p, p + k);
      p += k;
      n -= k;
      if (cur->in.size() == kBgzfIn)
        if (int rc = submit()) return rc;
    }
    return 0;
  }
};

extern "C" {

int dc_bgzf_open(const char* path, int threads, int level, dc_bgzf_writer** out) {
  if (!path || !out || threads < 1 || level < 0 || level > 9) return fail(DC_FQ_ERR_ARG, "invalid argument");
  std::unique_ptr<dc_bgzf_writer> w(new dc_bgzf_writer);
  w->fp = fopen(path, "wb");
  if (!w->fp) return fail(DC_FQ_ERR_IO, "cannot create '%s': %s", path, strerror(errno));
  w->level = level == 0 ? 6 : level;
  w->max_inflight = (size_t)threads * 4;
  w->cur.reset(new Job);
  w->cur->in.reserve(kBgzfIn);
  dc_bgzf_writer* raw = w.get();
  for (int i = 0; i < threads; ++i) w->workers.emplace_back([raw] { raw->worker(); });
  *out = w.release();
  return 0;
}

int dc_bgzf_write(dc_bgzf_writer* w, const void* data, int64_t n) {
  if (!w || (!data && n) || n < 0) return fail(DC_FQ_ERR_ARG, "invalid argument");
  return w->put((const uint8_t*)data, (size_t)n);
}

int dc_bgzf_write_fastq(dc_bgzf_writer* w, int64_t n, const char* names, const int64_t* name_offsets,
                        const uint8_t* seq, const uint8_t* qual, const int64_t* offsets) {
  if (!w || n < 0 || (n && (!names || !name_offsets || !seq || !qual || !offsets))) return fail(DC_FQ_ERR_ARG, "invalid argument");
  for (int64_t i = 0; i < n; ++i) {
    int64_t a = offsets[i], b = offsets[i + 1], na = name_offsets[i], nb = name_offsets[i + 1];
    if (b < a || nb < na) return fail(DC_FQ_ERR_ARG, "offsets not monotonic at record %lld", (long long)i);
    int rc = 0;
    if ((rc = w->put((const uint8_t*)"@", 1)) || (rc = w->put((const uint8_t*)names + na, (size_t)(nb - na))) ||
        (rc = w->put((const uint8_t*)"\n", 1)) || (rc = w->put(seq + a, (size_t)(b - a))) ||
        (rc = w->put((const uint8_t*)"\n+\n", 3)) || (rc = w->put(qual + a, (size_t)(b - a))) ||
        (rc = w->put((const uint8_t*)"\n", 1)))
      return rc;
  }
  return 0;
}

int dc_bgzf_mark(dc_bgzf_writer* w) {
  if (!w) return fail(DC_FQ_ERR_ARG, "null writer");
  int rc = w->submit();  // the bytes written so far end their block here (no-op when the block is empty)
  w->mark_block.push_back(w->n_submitted);
  w->settle_marks();
  return rc;
}

int dc_bgzf_close2(dc_bgzf_writer* w, int64_t* mark_offsets, int64_t cap, int64_t* n_marks, int64_t* data_bytes,
                   double* deflate_cpu_seconds) {
  if (!w) return fail(DC_FQ_ERR_ARG, "null writer");
  int rc = w->submit();
  int rc2 = w->drain(true, 0);
  if (!rc) rc = rc2;
  {
    std::lock_guard<std::mutex> lk(w->mu);
    w->stop = true;
  }
  w->cv_work.notify_all();
  for (auto& t : w->workers) t.join();
  w->settle_marks();
  if (!rc && w->mark_offset.size() != w->mark_block.size()) rc = fail(DC_FQ_ERR_IO, "BGZF writer lost track of a mark");
  if (n_marks) *n_marks = (int64_t)w->mark_offset.size();
  if (data_bytes) *data_bytes = w->written;  // everything in front of the EOF block
  if (deflate_cpu_seconds) *deflate_cpu_seconds = w->deflate_cpu_s;
  if (!rc && mark_offsets) {
    if ((int64_t)w->mark_offset.size() > cap) rc = fail(DC_FQ_ERR_ARG, "mark buffer too small: %lld marks", (long long)w->mark_offset.size());
    else if (!w->mark_offset.empty()) memcpy(mark_offsets, w->mark_offset.data(), w->mark_offset.size() * sizeof(int64_t));
  }
  if (!rc && fwrite(kBgzfEof, 1, sizeof kBgzfEof, w->fp) != sizeof kBgzfEof) rc = fail(DC_FQ_ERR_IO, "write failed");
  if (fclose(w->fp) != 0 && !rc) rc = fail(DC_FQ_ERR_IO, "close failed: %s", strerror(errno));
  delete w;
  return rc;
}

int dc_bgzf_close(dc_bgzf_writer* w) { return dc_bgzf_close2(w, nullptr, 0, nullptr, nullptr, nullptr); }

int64_t dc_bgzf_mark_count(const dc_bgzf_writer* w) { return w ? (int64_t)w->mark_block.size() : -1; }

int dc_deflate_block(const uint8_t* in, int64_t n, uint8_t* out, int64_t cap, int64_t* out_len) {
  if ((!in && n) || !out || !out_len || n < 0 || n >= 65535 || cap < 0) return fail(DC_FQ_ERR_ARG, "dc_deflate_block: bad argument");
  thread_local std::unique_ptr<dcdeflate::Workspace> ws;
  if (!ws) ws.reset(new dcdeflate::Workspace());
  const size_t k = dcdeflate::compress_block(in, (size_t)n, out, (size_t)cap, *ws);
  if (k == 0) return fail(DC_FQ_ERR_ARG, "dc_deflate_block: %lld bytes of output space for %lld bytes of input", (long long)cap, (long long)n);
  *out_len = (int64_t)k;
  return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Chop stage of the fused predict -> chop flow: gating + record splitting + FASTQ text, straight from the
// reader's packed chunk and the smoothing kernel's per-read intervals into the BGZF writer.
//   process_chunk                              src/bin/predict.rs:130-192
//   split_noodle_records_by_remove_intervals   src/output/split.rs:171-226 (+ helper :60-136)
//   split_noodle_records_by_intervals (--ocq)  src/output/split.rs:138-169
//   Predict.seq = id_list2seq(tokens)          src/smooth/predict.rs:264-317, src/smooth/utils.rs:6-25
// ------------------------------------------------------------------------------------------------
namespace {

// Letter the reference's `Predict.seq` holds for a raw FASTQ base: pyfastx upper-cases, needletail
// normalises (acgtn -> upper, U -> T, everything else -> N or '-'), the character tokenizer maps
// anything outside ACGTN to [UNK], and id_list2seq turns every id outside 7..11 into 'N'.
struct LetterLut {
  uint8_t t[256];
  LetterLut() {
    memset(t, 'N', sizeof t);
    t['A'] = t['a'] = 'A';
    t['C'] = t['c'] = 'C';
    t['G'] = t['g'] = 'G';
    t['T'] = t['t'] = t['U'] = t['u'] = 'T';
  }
};
const LetterLut kLetters;

}  // namespace

extern "C" int dc_chop_write_chunk(dc_bgzf_writer* w, const dc_chop_params* p, int64_t n, const char* headers,
                                   const int64_t* header_offsets, const int32_t* name_len, const uint8_t* seq,
                                   const uint8_t* qual, const int64_t* offsets, const int32_t* pred_len,
                                   const int32_t* n_intervals, const int32_t* intervals, int32_t max_intervals,
                                   const int32_t* n_keep, const int32_t* keep, int32_t max_keep,
                                   int32_t* records_per_read, int64_t* n_records_out) {
  if (!w || !p || n < 0 || (n && (!headers || !header_offsets || !seq || !qual || !offsets || !pred_len || !n_intervals)))
    return fail(DC_FQ_ERR_ARG, "invalid argument");
  if (p->chop_type < 0 || p->chop_type > 2) return fail(DC_FQ_ERR_ARG, "Invalid chop type");
  if (p->output_chopped_seqs ? !intervals : (!keep || !n_keep)) return fail(DC_FQ_ERR_ARG, "missing interval arrays");
  std::vector<uint8_t> rec, letters;
  int64_t total = 0;
  char tag[64];
  auto emit = [&](const char* id, size_t id_len, const char* suffix, size_t suffix_len, const uint8_t* s,
                  const uint8_t* q, size_t len) -> int {
    rec.clear();
    rec.push_back('@');
    rec.insert(rec.end(), id, id + id_len);
    rec.insert(rec.end(), suffix, suffix + suffix_len);
    rec.push_back('\n');
    rec.insert(rec.end(), s, s + len);
    rec.push_back('\n');
    rec.push_back('+');
    rec.push_back('\n');
    rec.insert(rec.end(), q, q + len);
    rec.push_back('\n');
    ++total;
    return w->put(rec.data(), rec.size());
  };
  for (int64_t i = 0; i < n; ++i) {
    const int64_t a = offsets[i], b = offsets[i + 1], ha = header_offsets[i], hb = header_offsets[i + 1];
    if (b < a || hb < ha) return fail(DC_FQ_ERR_ARG, "offsets not monotonic at record %lld", (long long)i);
    const int64_t L = b - a, before = total;
    const int64_t idl = name_len ? name_len[i] : hb - ha;
    if (idl < 0 || idl > hb - ha) return fail(DC_FQ_ERR_ARG, "bad name length at record %lld", (long long)i);
    const char* id = headers + ha;
    const int32_t pl = pred_len[i], niv = n_intervals[i];
    int rc = 0;
    // predict.rs:146-164: short reads, no / too many intervals, truncated predictions -> the record as read
    if (pl < p->min_read_len || niv == 0 || niv > p->max_process_intervals || (int64_t)pl != L) {
      if ((rc = emit(id, (size_t)(hb - ha), "", 0, seq + a, qual + a, (size_t)L))) return rc;
      if (records_per_read) records_per_read[i] = 1;
      continue;
    }
    if (niv < 0 || niv > max_intervals) return fail(DC_FQ_ERR_ARG, "interval count out of range at record %lld", (long long)i);
    letters.resize((size_t)L);
    for (int64_t k = 0; k < L; ++k) letters[(size_t)k] = kLetters.t[seq[a + k]];
    if (p->output_chopped_seqs) {  // --ocq: the chopped (adapter) segments themselves
      const int32_t* iv = intervals + (size_t)i * max_intervals * 2;
      for (int32_t k = 0; k < niv; ++k) {
        const int64_t s = iv[2 * k], e = iv[2 * k + 1];
        if (s < 0 || e < s || e > L) return fail(DC_FQ_ERR_ARG, "The interval is invalid: %lld..%lld", (long long)s, (long long)e);
        int tl = snprintf(tag, sizeof tag, "|%lld:%lld", (long long)s, (long long)e);
        if ((rc = emit(id, (size_t)idl, tag, (size_t)tl, letters.data() + s, qual + a + s, (size_t)(e - s)))) return rc;
      }
    } else {
      const int32_t nk = n_keep[i];
      if (nk < 0 || nk > max_keep) return fail(DC_FQ_ERR_ARG, "segment count out of range at record %lld", (long long)i);
      const int32_t* kp = keep + (size_t)i * max_keep * 2;
      const bool terminal = nk == 1;  // record count BEFORE the length filter (split.rs:182-188)
      int64_t first_len = -1;
      for (int32_t k = 0; k < nk && first_len < 0; ++k) {
        const int64_t s = kp[2 * k], e = kp[2 * k + 1];
        if (s < 0 || e < s || e > L) return fail(DC_FQ_ERR_ARG, "The interval is invalid: %lld..%lld", (long long)s, (long long)e);
        if (e - s >= p->min_read_length_after_chop) first_len = e - s;
      }
      if ((p->chop_type == 0 && !terminal) || (p->chop_type == 1 && terminal) || first_len == L) {
        if ((rc = emit(id, (size_t)idl, "", 0, letters.data(), qual + a, (size_t)L))) return rc;
      } else {
        for (int32_t k = 0; k < nk; ++k) {
          const int64_t s = kp[2 * k], e = kp[2 * k + 1];
          if (s < 0 || e < s || e > L) return fail(DC_FQ_ERR_ARG, "The interval is invalid: %lld..%lld", (long long)s, (long long)e);
          if (e - s < p->min_read_length_after_chop) continue;
          int tl = snprintf(tag, sizeof tag, "|%lld:%lld%s", (long long)s, (long long)e,
                            p->id_annotation ? (terminal ? "|T" : "|I") : "");
          if ((rc = emit(id, (size_t)idl, tag, (size_t)tl, letters.data() + s, qual + a + s, (size_t)(e - s)))) return rc;
        }
      }
    }
    if (records_per_read) records_per_read[i] = (int32_t)(total - before);
  }
  if (n_records_out) *n_records_out = total;
  return 0;
}
