// A DEFLATE (RFC 1951) / gzip (RFC 1952) decoder for the FASTQ reader: one plain .gz stream cannot be inflated in
// parallel, so the node's single inflater has to be fast.  zlib's inflate sustains ~150-330 MB/s on FASTQ text
// (almost all symbols are literals: ~13 cycles each); this decoder keeps 56+ bits in a 64-bit buffer (one
// branch-free refill per symbol group), resolves a literal / length code with one lookup in an 11-bit table
// (sub-tables behind it for the rare long codes) and copies matches eight bytes at a time.
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
  enum Status { NEED_OUTPUT = 0, MEMBER_END = 1, ERR_DATA = -1, ERR_TRUNCATED = -2 };

  void set_input(const uint8_t* p, const uint8_t* end) {
    in_ = p;
    in_end_ = end;
    bitbuf_ = 0;
    bitcnt_ = 0;
    state_ = ST_BLOCK_HEADER;
    last_block_ = false;
  }
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
  // SLACK bytes writable behind out_limit.  `hist` = bytes of this member already produced (window check).
  // Returns NEED_OUTPUT (out reached out_limit), MEMBER_END (final block done, byte aligned) or an error.
  static constexpr int SLACK = 320;
  Status run(uint8_t*& out, uint8_t* out_limit, uint64_t hist) {
    uint8_t* const out_begin = out;
    for (;;) {
      switch (state_) {
        case ST_BLOCK_HEADER: {
          if (last_block_) {
            align_drop();
            return MEMBER_END;
          }
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
            const Status s = read_dynamic();
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
            memcpy(out, in_, n);
            out += n;
            in_ += n;
            stored_left_ -= (unsigned)n;
          }
          state_ = ST_BLOCK_HEADER;
          break;
        }
        case ST_HUFF: {
          const Status s = huff(out, out_limit, hist, out_begin);
          if (s != MEMBER_END) return s;  // MEMBER_END here means "end of block"
          state_ = ST_BLOCK_HEADER;
          break;
        }
      }
    }
  }

 private:
  enum State { ST_BLOCK_HEADER, ST_STORED, ST_HUFF };
  static constexpr int LL_BITS = 12, D_BITS = 8;
  // entry: [4:0] bits to consume, [7:5] kind, [12:8] extra bits / sub-table bits, [31:16] value
  enum Kind { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4, K_DIST = 5, K_LIT2 = 6 };
  static uint32_t entry(unsigned nbits, Kind k, unsigned extra, unsigned value) {
    return nbits | ((unsigned)k << 5) | (extra << 8) | (value << 16);
  }

  const uint8_t* in_ = nullptr;
  const uint8_t* in_end_ = nullptr;
  uint64_t bitbuf_ = 0;
  int bitcnt_ = 0;
  State state_ = ST_BLOCK_HEADER;
  bool last_block_ = false;
  unsigned stored_left_ = 0;
  const char* err_ = "";
  uint32_t ll_[(1 << LL_BITS) + 288 * 8];
  uint32_t ll2_[1 << LL_BITS];   // literal pairs (built lazily, once a block has proven long enough to pay for it)
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
  static bool build(const uint8_t* lens, int n, int pbits, uint32_t* tab, int tab_cap, F sym_entry) {
    int count[16] = {0};
    for (int i = 0; i < n; ++i) ++count[lens[i]];
    count[0] = 0;
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
      left <<= 1;
      left -= count[l];
      if (left < 0) return false;
    }
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
      uint8_t sub_bits[1 << 12];
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

  // FASTQ text is almost all literals with short codes (2-3 bits for bases, 4-6 for qualities), and the chain
  // "look up -> shift by the code length -> look up" is what bounds a table decoder (~6 cycles per symbol).
  // Where two literal codes fit the primary index together, the pair table delivers both: half the chain per
  // byte.  Building it costs one pass over the primary table, so a block earns it only after PAIR_AFTER symbols
  // (fast compressors emit a block every ~16 k symbols).
  static constexpr unsigned PAIR_AFTER = 3000;
  void pair_literals() {
    for (unsigned i = 0; i < (1u << LL_BITS); ++i) {
      const uint32_t e1 = ll_[i];
      uint32_t r = e1;
      if (((e1 >> 5) & 7u) == K_LIT) {
        const unsigned l1 = e1 & 31u;
        const uint32_t e2 = ll_[i >> l1];
        const unsigned l2 = e2 & 31u;
        if (((e2 >> 5) & 7u) == K_LIT && l1 + l2 <= (unsigned)LL_BITS)
          r = entry(l1 + l2, K_LIT2, 0, (e1 >> 16) | ((e2 >> 16) << 8));
      }
      ll2_[i] = r;
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

  Status read_dynamic() {
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
    if (!build(cl, 19, 7, ct, 1 << 7, [](int s, int) { return entry(0, K_LIT, 0, (unsigned)s); }))
      return fail("invalid code lengths set");
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
  Status huff(uint8_t*& out_ref, uint8_t* out_limit, uint64_t hist0, const uint8_t* out_begin) {
    uint8_t* out = out_ref;
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
      const uint32_t* const lt = pairs_ready_ ? ll2_ : ll_;
      uint32_t e = lt[bb & ll_mask];
      // literal pairs / literals: after a refill at least 56 bits are valid whenever 8 input bytes were left, so up
      // to four entries (<= 11 bits each) are consumed without re-checking the count; near the end of the input the
      // count is checked per entry
      if (in <= in_fast_end) {
        int n = 0;
        while (n < 4) {
          const unsigned k = (e >> 5) & 7u;
          if (k == K_LIT2) {
            out[0] = (uint8_t)(e >> 16);
            out[1] = (uint8_t)(e >> 24);
            out += 2;
          } else if (k == K_LIT) {
            *out++ = (uint8_t)(e >> 16);
          } else {
            break;
          }
          bb >>= e & 31u;
          bc -= (int)(e & 31u);
          e = lt[bb & ll_mask];
          ++n;
        }
        block_syms_ += (unsigned)n;
        if (n == 4) continue;  // refill, go on (e is looked up again: cheap, keeps the loop simple)
        if (n > 0 && in <= in_fast_end) {  // a length / distance pair needs up to 48 bits: top up (idempotent)
          bb |= load64(in) << bc;
          in += (63 - bc) >> 3;
          bc |= 56;
        }
      }
      if (((e >> 5) & 7u) == K_SUB) {
        const unsigned sb = (e >> 8) & 31u;
        e = ll_[(e >> 16) + ((bb >> LL_BITS) & ((1u << sb) - 1))];
        if (((e >> 5) & 7u) != K_BAD) {
          bb >>= LL_BITS;
          bc -= LL_BITS;
        }
      }
      unsigned kind = (e >> 5) & 7u;
      if (kind == K_LIT || kind == K_LIT2) {
        if ((int)(e & 31u) > bc) { ret = fail_trunc(); break; }
        bb >>= e & 31u;
        bc -= (int)(e & 31u);
        *out++ = (uint8_t)(e >> 16);
        if (kind == K_LIT2) *out++ = (uint8_t)(e >> 24);
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
        const uint8_t* src = out - dist;
        uint8_t* const end = out + len;
        if (dist >= 8) {
          do {
            memcpy(out, src, 8);
            out += 8;
            src += 8;
          } while (out < end);
        } else if (dist == 1) {
          memset(out, *src, len);
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
