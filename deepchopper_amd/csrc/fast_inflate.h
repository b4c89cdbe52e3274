// A DEFLATE (RFC 1951) / gzip (RFC 1952) decoder for the FASTQ reader: one plain .gz stream cannot be inflated in
// parallel, so the node's single inflater has to be fast.  zlib's inflate sustains ~150-330 MB/s on FASTQ text
// (level-6 FASTQ: quality characters as literals, bases mostly as 6-byte matches); this decoder keeps 56+ bits in a
// 64-bit buffer (one branch-free refill per symbol group), resolves a literal / length code with one lookup in an
// 11-bit table (sub-tables behind it for the rare long codes), takes runs of literals from a second table that
// delivers up to four per lookup, and copies matches eight bytes at a time.  Measured on 1 kb synthetic FASTQ:
// 1.9x zlib; what is left is the two dependent lookups per match, ~1 mispredicted branch per literal/match
// transition and the per-block table builds.
//
// Header-only, host code, no dependencies besides <cstdint>/<cstring>.  Written from the RFCs; the table layout is
// this file's own.  Used by fastq_io.cpp on a memory-mapped file (the whole input is addressable, the output is
// produced block by block); DC_IO_INFLATE=zlib selects the zlib path instead.
#pragma once
#include <cstdint>
#include <cstring>

namespace dcio {

class FastInflate {
 public:
  enum Status { NEED_OUTPUT = 0, MEMBER_END = 1, BLOCK_STOP = 2, ERR_DATA = -1, ERR_TRUNCATED = -2 };
  static constexpr uint64_t NO_STOP = ~0ull;

  void set_input(const uint8_t* p, const uint8_t* end) {
    origin_ = p;
    in_ = p;
    in_end_ = end;
    bitbuf_ = 0;
    bitcnt_ = 0;
    state_ = ST_BLOCK_HEADER;
    last_block_ = false;
  }
  // Positions the decoder at bit `bitpos` of the stream that starts at `origin` (a block header is expected there).
  bool seek_bit(const uint8_t* origin, uint64_t bitpos, const uint8_t* end) {
    set_input(origin, end);
    in_ = origin + (bitpos >> 3);
    if (in_ > end) return false;
    const int r = (int)(bitpos & 7);
    if (r) {
      if (!need(r)) return false;
      bits(r);
    }
    return true;
  }
  // Bits consumed since set_input / relative to seek_bit's origin.
  uint64_t bit_pos() const { return (uint64_t)(in_ - origin_) * 8 - (uint64_t)bitcnt_; }
  // Bytes of input consumed so far (whole bytes; valid after MEMBER_END, when the stream is byte aligned).
  const uint8_t* input_pos() const { return in_ - (bitcnt_ >> 3); }
  const char* error() const { return err_; }

  // Starts the DEFLATE stream of a new member (after its gzip header).
  void begin_member() {
    state_ = ST_BLOCK_HEADER;
    last_block_ = false;
    align_drop();
  }

  // Decodes into [out, out_limit) -- the caller keeps 32 KiB of history in front of `out` valid and at least
  // SLACK elements writable behind out_limit.  `hist` = elements of history valid in front of `out` (window check).
  // Returns NEED_OUTPUT (out reached out_limit), MEMBER_END (final block done, byte aligned), BLOCK_STOP (a block
  // ended at or after bit `stop_bit`, the next header is unread) or an error.
  // T = uint8_t: bytes.  T = uint16_t: the same stream as 16-bit symbols, so that a decoder started in the middle
  // of a stream can carry references into the 32 KiB it has not seen as placeholder values (>= 256) that are
  // resolved later (fastq_io.cpp, parallel inflate).
  static constexpr int SLACK = 320;
  Status run(uint8_t*& out, uint8_t* out_limit, uint64_t hist) { return run_t<uint8_t>(out, out_limit, hist, NO_STOP); }
  template <class T>
  Status run_t(T*& out, T* out_limit, uint64_t hist, uint64_t stop_bit) {
    T* const out_begin = out;
    for (;;) {
      switch (state_) {
        case ST_BLOCK_HEADER: {
          if (last_block_) {
            align_drop();
            return MEMBER_END;
          }
          if (stop_bit != NO_STOP && bit_pos() >= stop_bit) return BLOCK_STOP;
          if (!need(3)) return fail_trunc();
          last_block_ = bits(1);
          const unsigned type = bits(2);
          if (type == 0) {
            align_drop();
            if (!need(32)) return fail_trunc();
            const unsigned len = bits(16), nlen = bits(16);
            if ((len ^ 0xFFFFu) != nlen) return fail("stored block length check failed");
            stored_left_ = len;
            state_ = ST_STORED;
          } else if (type == 1) {
            build_fixed();
            state_ = ST_HUFF;
          } else if (type == 2) {
            const Status s = read_dynamic(false);
            if (s != MEMBER_END) return s;
            state_ = ST_HUFF;
          } else {
            return fail("invalid block type");
          }
          break;
        }
        case ST_STORED: {
          // bit buffer holds whole bytes here: hand them back to the byte stream
          in_ -= bitcnt_ >> 3;
          bitbuf_ = 0;
          bitcnt_ = 0;
          while (stored_left_) {
            if (out >= out_limit) return NEED_OUTPUT;
            size_t n = stored_left_;
            if (n > (size_t)(out_limit - out)) n = (size_t)(out_limit - out);
            if (n > (size_t)(in_end_ - in_)) return fail_trunc();
            for (size_t k = 0; k < n; ++k) out[k] = in_[k];
            out += n;
            in_ += n;
            stored_left_ -= (unsigned)n;
          }
          state_ = ST_BLOCK_HEADER;
          break;
        }
        case ST_HUFF: {
          const Status s = huff<T>(out, out_limit, hist, out_begin);
          if (s != MEMBER_END) return s;  // MEMBER_END here means "end of block"
          state_ = ST_BLOCK_HEADER;
          break;
        }
      }
    }
  }

  // Is bit `bitpos` of the stream at `origin` the start of a dynamic-Huffman block of a text stream?  Strict header
  // check (complete code-length, literal/length and distance codes, as every deflate encoder emits them), then a
  // trial decode of up to `trial` symbols whose literals must be printable ASCII / line ends.  `scratch` holds
  // 32768 + trial + SLACK elements.  False positives are possible in principle; callers confirm a start by
  // decoding up to it from a known-good position.
  bool probe_text_block(const uint8_t* origin, uint64_t bitpos, const uint8_t* end, uint16_t* scratch, int trial) {
    if (!seek_bit(origin, bitpos, end)) return false;
    if (!need(3)) return false;
    if (bits(1) != 0) return false;  // the final block is reached by decoding, never searched for
    if (bits(2) != 2) return false;
    if (read_dynamic(true) != MEMBER_END) return false;
    for (int j = 0; j < 32768; ++j) scratch[j] = (uint16_t)(256 + j);
    uint16_t* out = scratch + 32768;
    uint16_t* const lim = out + trial;
    const Status st = huff<uint16_t>(out, lim, 32768, scratch + 32768);
    if (st != NEED_OUTPUT && st != MEMBER_END) return false;
    if (out - (scratch + 32768) < 64 && st != MEMBER_END) return false;
    for (const uint16_t* q = scratch + 32768; q < out; ++q) {
      const unsigned c = *q;
      if (c < 256 && !(c == '\n' || c == '\r' || c == '\t' || (c >= 32 && c < 127))) return false;
    }
    return true;
  }

 private:
  enum State { ST_BLOCK_HEADER, ST_STORED, ST_HUFF };
  static constexpr int LL_BITS = 11, D_BITS = 8;  // 11: measured best (table builds are ~20 % of a level-6 stream's time)
  // entry: [4:0] bits to consume, [7:5] kind, [12:8] extra bits / sub-table bits, [31:16] value
  enum Kind { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4, K_DIST = 5 };
  static uint32_t entry(unsigned nbits, Kind k, unsigned extra, unsigned value) {
    return nbits | ((unsigned)k << 5) | (extra << 8) | (value << 16);
  }

  const uint8_t* origin_ = nullptr;
  const uint8_t* in_ = nullptr;
  const uint8_t* in_end_ = nullptr;
  uint64_t bitbuf_ = 0;
  int bitcnt_ = 0;
  State state_ = ST_BLOCK_HEADER;
  bool last_block_ = false;
  unsigned stored_left_ = 0;
  const char* err_ = "";
  uint32_t ll_[(1 << LL_BITS) + 288 * 8];
  uint64_t lm_[1 << LL_BITS];    // literal runs (built lazily, once a block has proven long enough to pay for it)
  bool pairs_ready_ = false;
  unsigned block_syms_ = 0;
  uint32_t dd_[(1 << D_BITS) + 32 * 128];
  bool fixed_built_ = false;
  uint32_t fixed_ll_[(1 << LL_BITS)];
  uint32_t fixed_dd_[(1 << D_BITS)];

  Status fail(const char* m) {
    err_ = m;
    return ERR_DATA;
  }
  Status fail_trunc() {
    err_ = "unexpected end of deflate stream";
    return ERR_TRUNCATED;
  }
  void align_drop() {  // drop the bits up to the next byte boundary
    const int r = bitcnt_ & 7;
    bitbuf_ >>= r;
    bitcnt_ -= r;
  }
  // slow, exact refill (headers, stream tail)
  bool need(int n) {
    while (bitcnt_ < n) {
      if (in_ >= in_end_) return false;
      bitbuf_ |= (uint64_t)*in_++ << bitcnt_;
      bitcnt_ += 8;
    }
    return true;
  }
  unsigned bits(int n) {
    const unsigned v = (unsigned)(bitbuf_ & ((1ull << n) - 1));
    bitbuf_ >>= n;
    bitcnt_ -= n;
    return v;
  }

  static unsigned reverse(unsigned code, int len) {
    unsigned r = 0;
    for (int i = 0; i < len; ++i) r |= ((code >> i) & 1u) << (len - 1 - i);
    return r;
  }

  // Canonical Huffman -> lookup table with `pbits` primary bits.  sym_entry(sym, len) gives the leaf entry with the
  // "bits to consume" field left 0.  Returns false on an over-subscribed code; an incomplete code leaves K_BAD
  // entries (decoding one is a data error), as zlib tolerates for the single-code distance tree.
  template <class F>
  static bool build(const uint8_t* lens, int n, int pbits, uint32_t* tab, int tab_cap, F sym_entry, bool* complete = nullptr) {
    int count[16] = {0};
    for (int i = 0; i < n; ++i) ++count[lens[i]];
    count[0] = 0;
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
      left <<= 1;
      left -= count[l];
      if (left < 0) return false;
    }
    if (complete) *complete = left == 0;
    unsigned next[16];
    unsigned code = 0;
    for (int l = 1; l <= 15; ++l) {
      code = (code + count[l - 1]) << 1;
      next[l] = code;
    }
    const int psize = 1 << pbits;
    // a complete code (the usual case) writes every primary entry below; only an incomplete one needs the fill
    if (left != 0)
      for (int i = 0; i < psize; ++i) tab[i] = entry(0, K_BAD, 0, 0);
    int n_long = 0;
    for (int l = pbits + 1; l <= 15; ++l) n_long += count[l];
    if (n_long) {
      // longest code behind every primary prefix
      uint8_t sub_bits[1 << 12];  // pbits <= 12
      memset(sub_bits, 0, (size_t)psize);
      unsigned nx[16];
      memcpy(nx, next, sizeof nx);
      for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (l > pbits) {
          const unsigned r = reverse(nx[l], l);
          const unsigned pre = r & (psize - 1);
          if (l - pbits > sub_bits[pre]) sub_bits[pre] = (uint8_t)(l - pbits);
        }
        if (l) ++nx[l];
      }
      int used = psize;
      for (int pre = 0; pre < psize; ++pre) {
        if (sub_bits[pre]) {
          const int sz = 1 << sub_bits[pre];
          if (used + sz > tab_cap) return false;
          tab[pre] = entry((unsigned)pbits, K_SUB, sub_bits[pre], (unsigned)used);
          for (int i = 0; i < sz; ++i) tab[used + i] = entry(0, K_BAD, 0, 0);
          used += sz;
        }
      }
    }
    for (int s = 0; s < n; ++s) {
      const int l = lens[s];
      if (!l) continue;
      const unsigned r = reverse(next[l]++, l);
      const uint32_t e = sym_entry(s, l);
      if (l <= pbits) {
        for (unsigned i = r; i < (unsigned)psize; i += 1u << l) tab[i] = e | (unsigned)l;
      } else {
        const unsigned pre = r & (psize - 1);
        const unsigned base = tab[pre] >> 16, sb = (tab[pre] >> 8) & 31u;
        const unsigned hi = r >> pbits;
        for (unsigned i = hi; i < (1u << sb); i += 1u << (l - pbits)) tab[base + i] = e | (unsigned)(l - pbits);
      }
    }
    return true;
  }

  static uint32_t ll_entry(int sym, int) {
    static const uint16_t lbase[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                       31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    if (sym < 256) return entry(0, K_LIT, 0, (unsigned)sym);
    if (sym == 256) return entry(0, K_EOB, 0, 0);
    if (sym > 285) return entry(0, K_BAD, 0, 0);
    return entry(0, K_LEN, lext[sym - 257], lbase[sym - 257]);
  }
  static uint32_t d_entry(int sym, int) {
    static const uint16_t dbase[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
                                       193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    if (sym > 29) return entry(0, K_BAD, 0, 0);
    return entry(0, K_DIST, dext[sym], dbase[sym]);
  }

  // FASTQ text is almost all literals with short codes (2-3 bits for bases, 5-7 for qualities), and the chain
  // "look up -> shift by the code length -> look up" is what bounds a table decoder (~7 cycles per lookup).
  // The run table delivers every literal whose code fits the primary index together with the first one (up to four:
  // a whole 3-bit-per-base quadruple, or a quality pair) as one 64-bit entry: [5:0] bits consumed, [10:8] literals
  // (0 = not a literal, use the symbol table), [47:16] the bytes.  Building it costs one pass over the primary
  // table, so a block earns it only after PAIR_AFTER symbols (fast compressors emit a block every ~16 k symbols).
  static constexpr unsigned PAIR_AFTER = 1000;
  void pair_literals() {
    for (unsigned i = 0; i < (1u << LL_BITS); ++i) {
      const uint32_t e1 = ll_[i];
      uint64_t r = 0;
      if (((e1 >> 5) & 7u) == K_LIT) {
        unsigned nb = e1 & 31u, n = 1, idx = i >> nb;
        uint64_t val = e1 >> 16;
        while (n < 4) {
          const uint32_t e2 = ll_[idx];
          const unsigned l2 = e2 & 31u;
          if (((e2 >> 5) & 7u) != K_LIT || nb + l2 > (unsigned)LL_BITS) break;
          val |= (uint64_t)(e2 >> 16) << (8 * n);
          nb += l2;
          idx >>= l2;
          ++n;
        }
        r = nb | (n << 8) | (val << 16);
      }
      lm_[i] = r;
    }
    pairs_ready_ = true;
  }

  void build_fixed() {
    if (!fixed_built_) {
      uint8_t l[288];
      for (int i = 0; i < 144; ++i) l[i] = 8;
      for (int i = 144; i < 256; ++i) l[i] = 9;
      for (int i = 256; i < 280; ++i) l[i] = 7;
      for (int i = 280; i < 288; ++i) l[i] = 8;
      build(l, 288, LL_BITS, fixed_ll_, 1 << LL_BITS, ll_entry);
      uint8_t d[32];
      for (int i = 0; i < 32; ++i) d[i] = 5;
      build(d, 32, D_BITS, fixed_dd_, 1 << D_BITS, d_entry);
      fixed_built_ = true;
    }
    memcpy(ll_, fixed_ll_, sizeof fixed_ll_);
    memcpy(dd_, fixed_dd_, sizeof fixed_dd_);
    pairs_ready_ = false;
    block_syms_ = 0;
  }

  // strict: block-start probing -- every code must be complete (a distance code may also be a single symbol)
  Status read_dynamic(bool strict) {
    if (!need(14)) return fail_trunc();
    const int hlit = (int)bits(5) + 257, hdist = (int)bits(5) + 1, hclen = (int)bits(4) + 4;
    if (hlit > 286 || hdist > 30) return fail("too many length or distance symbols");
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; ++i) {
      if (!need(3)) return fail_trunc();
      cl[order[i]] = (uint8_t)bits(3);
    }
    uint32_t ct[1 << 7];
    bool full = false;
    if (!build(cl, 19, 7, ct, 1 << 7, [](int s, int) { return entry(0, K_LIT, 0, (unsigned)s); }, &full))
      return fail("invalid code lengths set");
    if (strict && !full) return fail("incomplete code lengths set");
    uint8_t lens[286 + 30 + 16];
    int i = 0;
    while (i < hlit + hdist) {
      if (!need(7 + 7)) {
        if (in_ >= in_end_ && bitcnt_ == 0) return fail_trunc();
      }
      const uint32_t e = ct[bitbuf_ & 127u];
      if (((e >> 5) & 7u) != K_LIT) return fail("invalid code length code");
      if ((int)(e & 31u) > bitcnt_) return fail_trunc();
      bits((int)(e & 31u));
      const unsigned sym = e >> 16;
      if (sym < 16) {
        lens[i++] = (uint8_t)sym;
      } else {
        unsigned rep, val = 0;
        if (sym == 16) {
          if (i == 0) return fail("invalid bit length repeat");
          if (!need(2)) return fail_trunc();
          val = lens[i - 1];
          rep = 3 + bits(2);
        } else if (sym == 17) {
          if (!need(3)) return fail_trunc();
          rep = 3 + bits(3);
        } else {
          if (!need(7)) return fail_trunc();
          rep = 11 + bits(7);
        }
        if (i + (int)rep > hlit + hdist) return fail("invalid bit length repeat");
        while (rep--) lens[i++] = (uint8_t)val;
      }
    }
    if (lens[256] == 0) return fail("invalid code -- missing end-of-block");
    if (strict) {  // cheap rejections before any table is built
      int n_d = 0;
      for (int k = 0; k < hdist; ++k) n_d += lens[hlit + k] != 0;
      long kraft = 0;
      for (int k = 0; k < hlit; ++k)
        if (lens[k]) kraft += 1l << (15 - lens[k]);
      if (kraft != (1l << 15)) return fail("incomplete literal/lengths set");
      if (n_d > 1) {
        kraft = 0;
        for (int k = 0; k < hdist; ++k)
          if (lens[hlit + k]) kraft += 1l << (15 - lens[hlit + k]);
        if (kraft != (1l << 15)) return fail("incomplete distances set");
      }
    }
    if (!build(lens, hlit, LL_BITS, ll_, (int)(sizeof ll_ / sizeof ll_[0]), ll_entry)) return fail("invalid literal/lengths set");
    pairs_ready_ = false;
    block_syms_ = 0;
    if (!build(lens + hlit, hdist, D_BITS, dd_, (int)(sizeof dd_ / sizeof dd_[0]), d_entry)) return fail("invalid distances set");
    return MEMBER_END;
  }

  static uint64_t load64(const uint8_t* p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;  // little endian host
  }

  // Symbols of one Huffman block.  Returns MEMBER_END at the end-of-block symbol, NEED_OUTPUT when out_limit is
  // reached, or an error.
  template <class T>
  Status huff(T*& out_ref, T* out_limit, uint64_t hist0, const T* out_begin) {
    T* out = out_ref;
    const uint8_t* in = in_;
    uint64_t bb = bitbuf_;
    int bc = bitcnt_;
    const uint8_t* const in_fast_end = in_end_ - 8;
    Status ret = NEED_OUTPUT;
    const uint64_t ll_mask = (1u << LL_BITS) - 1, d_mask = (1u << D_BITS) - 1;
#define DC_SAVE()   \
  do {              \
    out_ref = out;  \
    in_ = in;       \
    bitbuf_ = bb;   \
    bitcnt_ = bc;   \
  } while (0)
    while (out < out_limit) {
      if (in <= in_fast_end) {  // branch-free refill to >= 56 bits
        bb |= load64(in) << bc;
        in += (63 - bc) >> 3;
        bc |= 56;
      } else {
        while (bc <= 56 && in < in_end_) {
          bb |= (uint64_t)*in++ << bc;
          bc += 8;
        }
      }
      if (!pairs_ready_ && block_syms_ > PAIR_AFTER) pair_literals();
      // literal runs: after a refill at least 56 bits are valid whenever 8 input bytes were left, so four entries
      // (<= LL_BITS bits each) are consumed without re-checking the count.  Four bytes are stored per entry and the
      // pointer advances by the run length (the caller's SLACK covers the overshoot): no branch on the run length.
      if (in <= in_fast_end) {
        int n = 0;
        if (pairs_ready_) {
          while (n < 4) {
            const uint64_t m = lm_[bb & ll_mask];
            const unsigned cnt = (unsigned)(m >> 8) & 7u;
            if (!cnt) break;
            const uint32_t v = (uint32_t)(m >> 16);
            if (sizeof(T) == 1) {
              memcpy(out, &v, 4);
            } else {  // four bytes widened to four 16-bit symbols
              const uint64_t w = (uint64_t)(v & 0xFFu) | ((uint64_t)(v & 0xFF00u) << 8) | ((uint64_t)(v & 0xFF0000u) << 16) |
                                 ((uint64_t)(v & 0xFF000000u) << 24);
              memcpy(out, &w, 8);
            }
            out += cnt;
            bb >>= m & 63u;
            bc -= (int)(m & 63u);
            ++n;
          }
        } else {
          while (n < 4) {
            const uint32_t e1 = ll_[bb & ll_mask];
            if (((e1 >> 5) & 7u) != K_LIT) break;
            *out++ = (T)(uint8_t)(e1 >> 16);
            bb >>= e1 & 31u;
            bc -= (int)(e1 & 31u);
            ++n;
          }
          block_syms_ += (unsigned)n;
        }
        if (n == 4) continue;
        if (n > 0) {  // a length / distance pair needs up to 48 bits: top up (idempotent)
          bb |= load64(in) << bc;
          in += (63 - bc) >> 3;
          bc |= 56;
        }
      }
      uint32_t e = ll_[bb & ll_mask];
      if (((e >> 5) & 7u) == K_SUB) {
        const unsigned sb = (e >> 8) & 31u;
        e = ll_[(e >> 16) + ((bb >> LL_BITS) & ((1u << sb) - 1))];
        if (((e >> 5) & 7u) != K_BAD) {
          bb >>= LL_BITS;
          bc -= LL_BITS;
        }
      }
      unsigned kind = (e >> 5) & 7u;
      if (kind == K_LIT) {
        if ((int)(e & 31u) > bc) { ret = fail_trunc(); break; }
        bb >>= e & 31u;
        bc -= (int)(e & 31u);
        *out++ = (T)(uint8_t)(e >> 16);
        continue;
      }
      if (kind == K_LEN) {
        unsigned need_bits = (e & 31u) + ((e >> 8) & 31u);
        if ((int)need_bits > bc) { ret = fail_trunc(); break; }
        bb >>= e & 31u;
        unsigned len = (e >> 16) + (unsigned)(bb & ((1u << ((e >> 8) & 31u)) - 1));
        bb >>= (e >> 8) & 31u;
        bc -= (int)need_bits;
        uint32_t d = dd_[bb & d_mask];
        if (((d >> 5) & 7u) == K_SUB) {
          const unsigned sb = (d >> 8) & 31u;
          d = dd_[(d >> 16) + ((bb >> D_BITS) & ((1u << sb) - 1))];
          if (((d >> 5) & 7u) != K_BAD) {
            bb >>= D_BITS;
            bc -= D_BITS;
          }
        }
        if (((d >> 5) & 7u) != K_DIST) { ret = fail("invalid distance code"); break; }
        need_bits = (d & 31u) + ((d >> 8) & 31u);
        if ((int)need_bits > bc) { ret = fail_trunc(); break; }
        bb >>= d & 31u;
        const unsigned dist = (d >> 16) + (unsigned)(bb & ((1u << ((d >> 8) & 31u)) - 1));
        bb >>= (d >> 8) & 31u;
        bc -= (int)need_bits;
        if ((uint64_t)dist > hist0 + (uint64_t)(out - out_begin)) { ret = fail("invalid distance too far back"); break; }
        const T* src = out - dist;
        T* const end = out + len;
        constexpr unsigned STEP = 8 / sizeof(T);  // elements per 8-byte copy
        if (dist >= STEP) {
          do {
            memcpy(out, src, 8);
            out += STEP;
            src += STEP;
          } while (out < end);
        } else if (dist == 1) {
          const T c = *src;
          for (unsigned k = 0; k < len; ++k) out[k] = c;
        } else {
          do {
            *out++ = *src++;
          } while (out < end);
        }
        out = end;
        continue;
      }
      if (kind == K_EOB) {
        if ((int)(e & 31u) > bc) { ret = fail_trunc(); break; }
        bb >>= e & 31u;
        bc -= (int)(e & 31u);
        ret = MEMBER_END;
        break;
      }
      ret = bc < 15 && in >= in_end_ ? fail_trunc() : fail("invalid literal/length code");
      break;
    }
    DC_SAVE();
#undef DC_SAVE
    return ret;
  }
};

}  // namespace dcio
