// One-shot raw DEFLATE (RFC 1951) of one BGZF payload (< 64 KiB) for the chop stage's output
// (bgzf::io::MultithreadedWriter in the reference: src/output/writefq.rs:252-276, src/bin/predict.rs:266-270; noodles deflates
// every block with flate2 at its default level).
//
// zlib at level 6 spends 0.093 CPU-seconds per 1 000 nanopore reads here -- 2.3 cores per GPU rank, the largest host cost of
// config 5 and what bounds `deepchopper chop` at the reference's `-t 2`.  On FASTQ text its time goes into walking full hash
// chains: in a 32 KiB window every 3- or 4-mer of ACGT has occurred a hundred times, every chain is as long as it is allowed
// to be (128), and what the walk finds are 3 .. 6-byte matches that cost MORE bits than the 2-bit literals they replace.
// This encoder is built for that text:
//   * two match finders: a 4-byte hash with SHORT chains for nearby short matches, and a one-slot 8-byte hash that finds the
//     long matches (headers, poly-A tails, repeats) directly, however deep they would sit in a 4-byte chain;
//   * a match is taken only if it is cheaper than its literals: per-byte literal costs from the block's own byte histogram
//     (the first approximation of the Huffman code it will get), match cost from its length / distance class;
//   * one step of lazy evaluation, as zlib's levels 4+;
//   * one dynamic-Huffman block per payload (length-limited codes: Huffman by two queues, lengths folded to 15 / 7 bits by
//     Kraft-sum repair), or a stored block when that is smaller.
// Any inflater reads the result; tests/test_fast_deflate.py round-trips it through zlib and through this library's own inflater.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace dcdeflate {

constexpr int kHashBits = 15, kHashSize = 1 << kHashBits;
constexpr int kMaxMatch = 258, kMaxDist = 32768;
#ifndef DC_DEFLATE_TUNE
constexpr int kHashBytes = 4;    // bytes hashed into the chains = the shortest match looked for
constexpr int kChainDepth = 12;  // short chains: nearby short matches only (the 8-byte table finds the long ones)
constexpr int kLazyBelow = 48;   // a match this long is taken at once
constexpr int kNice = 64;        // ... and ends the search
// what a match costs: the length symbol and the distance symbol of a block that uses many matches (4 and 3 bits, measured on
// nanopore FASTQ) + their extra bits
constexpr float kLenBits = 4.0f, kDistBits = 3.0f, kGainMin = 0.0f, kLazyFrac = 1.0f;
constexpr int kColdAfter = 24, kColdDepth = 3, kGoodLen = 8;
#else
static int kHashBytes = 4, kChainDepth = 12, kLazyBelow = 48, kNice = 64;
static float kLenBits = 4.0f, kDistBits = 3.0f, kGainMin = 0.0f, kLazyFrac = 1.0f;
static int kColdAfter = 24, kColdDepth = 3, kGoodLen = 8;
#endif

struct Workspace {
  uint16_t head4[kHashSize];   // position + 1 of the latest occurrence (0 = none); a payload has < 65 535 bytes
  uint16_t head8[kHashSize];
  uint16_t prev4[65536];
  uint32_t syms[65536];        // literal: byte; match: 0x80000000 | (len - 3) << 16 | (dist - 1)
  uint32_t lfreq[288], dfreq[32];
  float lit_cost[256];
  float cum[65536 + 1];        // prefix sums of the literal costs
};

static inline uint32_t ld32(const uint8_t* p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}
static inline uint64_t ld64(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}
static inline uint32_t hash4(const uint8_t* p) {  // the first kHashBytes (4 .. 6) bytes
  const uint64_t v = ld64(p) & (kHashBytes >= 8 ? ~0ull : ((1ull << (8 * kHashBytes)) - 1));
  return (uint32_t)((v * 0x9E3779B97F4A7C15ull) >> (64 - kHashBits));
}
static inline uint32_t hash8(const uint8_t* p) { return (uint32_t)((ld64(p) * 0x9E3779B97F4A7C15ull) >> (64 - kHashBits)); }

static inline int match_len(const uint8_t* a, const uint8_t* b, int max) {
  int l = 0;
  while (l + 8 <= max) {
    const uint64_t x = ld64(a + l) ^ ld64(b + l);
    if (x) return l + (__builtin_ctzll(x) >> 3);
    l += 8;
  }
  while (l < max && a[l] == b[l]) ++l;
  return l;
}

// length / distance symbol tables of RFC 1951, 3.2.5
struct Tables {
  uint8_t len_sym[259];   // length -> symbol - 257
  uint8_t len_extra[29];
  uint16_t len_base[29];
  uint8_t dist_extra[30];
  uint16_t dist_base[30];
  uint8_t dist_sym_lo[512];   // dist - 1 < 512
  uint8_t dist_sym_hi[256];   // (dist - 1) >> 7 for the rest
  Tables() {
    static const uint8_t le[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint8_t de[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    int b = 3;
    for (int s = 0; s < 28; ++s) {
      len_extra[s] = le[s];
      len_base[s] = (uint16_t)b;
      for (int k = 0; k < (1 << le[s]); ++k, ++b) len_sym[b] = (uint8_t)s;
    }
    len_extra[28] = 0;
    len_base[28] = 258;
    len_sym[258] = 28;
    int d = 1;
    for (int s = 0; s < 30; ++s) {
      dist_extra[s] = de[s];
      dist_base[s] = (uint16_t)d;
      for (int k = 0; k < (1 << de[s]); ++k, ++d) {
        const int z = d - 1;
        if (z < 512) dist_sym_lo[z] = (uint8_t)s;
        else dist_sym_hi[z >> 7] = (uint8_t)s;
      }
    }
  }
  int dsym(int dist) const {
    const int z = dist - 1;
    return z < 512 ? dist_sym_lo[z] : dist_sym_hi[z >> 7];
  }
};
static const Tables& tables() {
  static const Tables t;
  return t;
}

struct BitWriter {
  uint8_t* p;
  uint8_t* end;
  uint64_t acc = 0;
  int nb = 0;
  bool overflow = false;
  void put(uint32_t v, int n) {  // n <= 32, nb < 32 on entry
    acc |= (uint64_t)v << nb;
    nb += n;
    if (nb >= 32) {
      if (p + 4 <= end) {
        const uint32_t lo = (uint32_t)acc;
        memcpy(p, &lo, 4);  // (little-endian host, as the rest of this library)
        p += 4;
      } else {
        overflow = true;
      }
      acc >>= 32;
      nb -= 32;
    }
  }
  void flush() {
    while (nb > 0) {
      if (p < end) *p++ = (uint8_t)acc;
      else overflow = true;
      acc >>= 8;
      nb -= 8;
    }
    nb = 0;
  }
};

// Length-limited Huffman code lengths for n symbols (freq 0 -> length 0); at least two codes get a length so that any
// decoder accepts the tree.
static void code_lengths(const uint32_t* freq, int n, int max_len, uint8_t* len) {
  struct Node {
    uint64_t w;
    int l, r;
  };
  int idx[288];
  int m = 0;
  for (int i = 0; i < n; ++i) {
    len[i] = 0;
    if (freq[i]) idx[m++] = i;
  }
  if (m == 0) {
    len[0] = len[1] = 1;
    return;
  }
  if (m == 1) {
    len[idx[0]] = 1;
    len[idx[0] ? 0 : 1] = 1;
    return;
  }
  std::sort(idx, idx + m, [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
  Node nodes[2 * 288];
  for (int i = 0; i < m; ++i) nodes[i] = {freq[idx[i]], -1, -1};
  int leaf = 0, inner = m, made = m;  // two queues: leaves (sorted), inner nodes (made in non-decreasing weight)
  auto take = [&]() {
    if (leaf < m && (inner >= made || nodes[leaf].w <= nodes[inner].w)) return leaf++;
    return inner++;
  };
  while ((m - leaf) + (made - inner) > 1) {
    const int a = take(), b = take();
    nodes[made] = {nodes[a].w + nodes[b].w, a, b};
    ++made;
  }
  // depths: walk down from the root (the last node made)
  int depth[2 * 288];
  depth[made - 1] = 0;
  for (int i = made - 1; i >= m; --i) {
    depth[nodes[i].l] = depth[i] + 1;
    depth[nodes[i].r] = depth[i] + 1;
  }
  int count[64] = {0};
  for (int i = 0; i < m; ++i) count[std::min(depth[i], 63)]++;
  // fold lengths above max_len into it and repair the Kraft sum
  for (int i = max_len + 1; i < 64; ++i) {
    count[max_len] += count[i];
    count[i] = 0;
  }
  uint64_t total = 0;
  for (int i = max_len; i >= 1; --i) total += (uint64_t)count[i] << (max_len - i);
  while (total > (1ull << max_len)) {
    count[max_len]--;
    for (int i = max_len - 1; i >= 1; --i)
      if (count[i]) {
        count[i]--;
        count[i + 1] += 2;
        break;
      }
    --total;
  }
  // rarest symbols get the longest codes
  int k = 0;
  for (int l = max_len; l >= 1; --l)
    for (int c = 0; c < count[l]; ++c) len[idx[k++]] = (uint8_t)l;
}

static void canonical_codes(const uint8_t* len, int n, uint16_t* code) {  // bit-reversed: DEFLATE packs codes MSB first into an LSB-first stream
  int count[16] = {0}, next[16];
  for (int i = 0; i < n; ++i) count[len[i]]++;
  count[0] = 0;
  int c = 0;
  for (int l = 1; l <= 15; ++l) {
    c = (c + count[l - 1]) << 1;
    next[l] = c;
  }
  for (int i = 0; i < n; ++i) {
    const int l = len[i];
    if (!l) {
      code[i] = 0;
      continue;
    }
    uint32_t v = (uint32_t)next[l]++, r = 0;
    for (int b = 0; b < l; ++b) r |= ((v >> b) & 1u) << (l - 1 - b);
    code[i] = (uint16_t)r;
  }
}

// Raw-deflate `n` (< 65 535) bytes into out[0 .. cap); returns the bytes written, 0 if they do not fit.
static size_t compress_block(const uint8_t* in, size_t n, uint8_t* out, size_t cap, Workspace& w) {
  const Tables& T = tables();
  if (n >= 65535) return 0;
  auto stored = [&]() -> size_t {
    if (cap < n + 5) return 0;
    out[0] = 1;  // BFINAL, BTYPE 00
    out[1] = (uint8_t)n;
    out[2] = (uint8_t)(n >> 8);
    out[3] = (uint8_t)~n;
    out[4] = (uint8_t)(~n >> 8);
    if (n) memcpy(out + 5, in, n);
    return n + 5;
  };
  if (n < 16) return stored();

  // ---- literal costs from the byte histogram ----
  uint32_t hist[256] = {0};
  for (size_t i = 0; i < n; ++i) hist[in[i]]++;
  for (int b = 0; b < 256; ++b) w.lit_cost[b] = hist[b] ? std::min(15.0f, std::max(1.0f, log2f((float)n / (float)hist[b]))) : 15.0f;
  w.cum[0] = 0.f;
  for (size_t i = 0; i < n; ++i) w.cum[i + 1] = w.cum[i] + w.lit_cost[in[i]];
  auto match_cost = [](int len, int dist) -> float {  // length symbol + extra bits, distance symbol + extra bits
    const int lx = len < 11 ? 0 : len < 19 ? 1 : len < 35 ? 2 : len < 67 ? 3 : len < 131 ? 4 : 5;
    const int dx = dist < 5 ? 0 : 30 - __builtin_clz((unsigned)(dist - 1));
    return kLenBits + lx + kDistBits + dx;
  };

  memset(w.head4, 0, sizeof w.head4);
  memset(w.head8, 0, sizeof w.head8);
  uint32_t* const syms = w.syms;
  size_t ns = 0;  // (one symbol per byte at most)
  const size_t hash_end = n >= 8 ? n - 8 : 0;  // positions with 8 readable bytes

  struct M {
    int len, dist;
    float gain;
  };
  auto find = [&](size_t i, int depth) -> M {
    M best{0, 0, kGainMin};
    if (i > hash_end) return best;
    const int max = (int)std::min<size_t>(kMaxMatch, n - i);
    const uint64_t here = ld64(in + i);
    const float* cum = w.cum + i;
    // Candidates come nearest first: one that is no longer than the longest seen so far is also farther away -- it cannot
    // win, whether or not that longest one paid
    int longest = kHashBytes - 1;
    auto consider = [&](size_t c) {
      const uint64_t x = ld64(in + c) ^ here;
      int l;
      if (x) {
        l = __builtin_ctzll(x) >> 3;
        if (l <= longest) return;
      } else {
        l = 8 + match_len(in + c + 8, in + i + 8, max - 8);
      }
      if (l > max) l = max;
      if (l <= longest) return;
      longest = l;
      const int dist = (int)(i - c);
      const float gain = cum[l] - match_cost(l, dist);
      if (gain > best.gain) best = {l, dist, gain};
    };
    const uint32_t h8 = hash8(in + i), h4 = hash4(in + i);
    if (const uint32_t c8 = w.head8[h8]) {
      if (i - (c8 - 1) <= (size_t)kMaxDist) consider(c8 - 1);
    }
    const uint32_t first = w.head4[h4];
    uint32_t c = first;
    for (int d = 0; c && d < depth && longest < kNice; ++d) {
      const size_t cp = c - 1;
      if (i - cp > (size_t)kMaxDist) break;
      consider(cp);
      c = w.prev4[cp];
    }
    // ... and position i becomes a candidate itself (its hashes are at hand)
    w.prev4[i] = (uint16_t)first;
    w.head4[h4] = (uint16_t)(i + 1);
    w.head8[h8] = (uint16_t)(i + 1);
    return best;
  };
  auto insert = [&](size_t i) {
    if (i > hash_end) return;
    const uint32_t h4 = hash4(in + i);
    w.prev4[i] = w.head4[h4];
    w.head4[h4] = (uint16_t)(i + 1);
    w.head8[hash8(in + i)] = (uint16_t)(i + 1);
  };

  // Search effort follows the text: inside a stretch that has not produced a match for a while (a sequence line: its 4-mers
  // have all been seen, none of the finds pays) the chains are walked kColdDepth deep; a match brings the full depth back.  The
  // lazy second look behind a match that is already long gets a quarter of the depth, as zlib's good_length does.
  int cold = 0;
  auto depth_now = [&]() { return cold > kColdAfter ? kColdDepth : kChainDepth; };
  size_t i = 0;
  M cur = find(0, kChainDepth);
  while (i < n) {
    if (cur.len >= kHashBytes && cur.gain > kGainMin) {
      cold = 0;
      // one step of lazy evaluation: a better match one byte on wins over this one + makes this byte a literal
      if (cur.len < kLazyBelow && i + 1 < n) {
        const M nxt = find(i + 1, cur.len >= kGoodLen ? std::max(1, kChainDepth / 4) : kChainDepth);
        if (nxt.len >= kHashBytes && nxt.gain > cur.gain + w.lit_cost[in[i]] * kLazyFrac) {
          syms[ns++] = in[i];
          ++i;
          cur = nxt;
          continue;
        }
        syms[ns++] = 0x80000000u | (uint32_t)(cur.len - 3) << 16 | (uint32_t)(cur.dist - 1);
        for (size_t k = i + 2; k < i + (size_t)cur.len; ++k) insert(k);  // (i and i + 1 went in with their searches)
      } else {
        syms[ns++] = 0x80000000u | (uint32_t)(cur.len - 3) << 16 | (uint32_t)(cur.dist - 1);
        for (size_t k = i + 1; k < i + (size_t)cur.len; ++k) insert(k);
      }
      i += (size_t)cur.len;
      cur = i < n ? find(i, kChainDepth) : M{0, 0, 0.f};
    } else {
      syms[ns++] = in[i];
      ++i;
      ++cold;
      cur = i < n ? find(i, depth_now()) : M{0, 0, 0.f};
    }
  }

  // ---- Huffman codes ----
  memset(w.lfreq, 0, sizeof w.lfreq);
  memset(w.dfreq, 0, sizeof w.dfreq);
  for (size_t k = 0; k < ns; ++k) {
    const uint32_t s = syms[k];
    if (s & 0x80000000u) {
      w.lfreq[257 + T.len_sym[((s >> 16) & 0x7fff) + 3]]++;
      w.dfreq[T.dsym((int)(s & 0xffff) + 1)]++;
    } else {
      w.lfreq[s]++;
    }
  }
  w.lfreq[256] = 1;
  uint8_t llen[288], dlen[32];
  uint16_t lcode[288], dcode[32];
  code_lengths(w.lfreq, 286, 15, llen);
  code_lengths(w.dfreq, 30, 15, dlen);
  canonical_codes(llen, 286, lcode);
  canonical_codes(dlen, 30, dcode);
  int hlit = 286, hdist = 30;
  while (hlit > 257 && llen[hlit - 1] == 0) --hlit;
  while (hdist > 1 && dlen[hdist - 1] == 0) --hdist;
  // code lengths of both alphabets, run-length coded with the symbols 16 / 17 / 18
  uint8_t seq[286 + 30];
  memcpy(seq, llen, (size_t)hlit);
  memcpy(seq + hlit, dlen, (size_t)hdist);
  const int nseq = hlit + hdist;
  struct CL {
    uint8_t sym, extra;
  };
  CL cl[286 + 30];
  int ncl = 0;
  uint32_t clfreq[19] = {0};
  for (int k = 0; k < nseq;) {
    const int v = seq[k];
    int run = 1;
    while (k + run < nseq && seq[k + run] == v) ++run;
    int left = run;
    if (v == 0) {
      while (left >= 11) {
        const int r = std::min(left, 138);
        cl[ncl++] = {18, (uint8_t)(r - 11)};
        left -= r;
      }
      if (left >= 3) {
        cl[ncl++] = {17, (uint8_t)(left - 3)};
        left = 0;
      }
    } else {
      cl[ncl++] = {(uint8_t)v, 0};
      --left;
      while (left >= 3) {
        const int r = std::min(left, 6);
        cl[ncl++] = {16, (uint8_t)(r - 3)};
        left -= r;
      }
    }
    while (left-- > 0) cl[ncl++] = {(uint8_t)v, 0};
    k += run;
  }
  for (int k = 0; k < ncl; ++k) clfreq[cl[k].sym]++;
  uint8_t cllen[19];
  uint16_t clcode[19];
  code_lengths(clfreq, 19, 7, cllen);
  canonical_codes(cllen, 19, clcode);
  static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  int hclen = 19;
  while (hclen > 4 && cllen[order[hclen - 1]] == 0) --hclen;

  BitWriter bw{out, out + cap};
  bw.put(1, 1);  // BFINAL
  bw.put(2, 2);  // dynamic Huffman
  bw.put((uint32_t)(hlit - 257), 5);
  bw.put((uint32_t)(hdist - 1), 5);
  bw.put((uint32_t)(hclen - 4), 4);
  for (int k = 0; k < hclen; ++k) bw.put(cllen[order[k]], 3);
  for (int k = 0; k < ncl; ++k) {
    bw.put(clcode[cl[k].sym], cllen[cl[k].sym]);
    if (cl[k].sym == 16) bw.put(cl[k].extra, 2);
    else if (cl[k].sym == 17) bw.put(cl[k].extra, 3);
    else if (cl[k].sym == 18) bw.put(cl[k].extra, 7);
  }
  for (size_t k = 0; k < ns; ++k) {
    const uint32_t s = syms[k];
    if (s & 0x80000000u) {
      const int len = (int)((s >> 16) & 0x7fff) + 3, dist = (int)(s & 0xffff) + 1;
      const int ls = T.len_sym[len], ds = T.dsym(dist);
      bw.put(lcode[257 + ls], llen[257 + ls]);
      if (T.len_extra[ls]) bw.put((uint32_t)(len - T.len_base[ls]), T.len_extra[ls]);
      bw.put(dcode[ds], dlen[ds]);
      if (T.dist_extra[ds]) bw.put((uint32_t)(dist - T.dist_base[ds]), T.dist_extra[ds]);
    } else {
      bw.put(lcode[s], llen[s]);
    }
  }
  bw.put(lcode[256], llen[256]);
  bw.flush();
  const size_t used = (size_t)(bw.p - out);
  if (bw.overflow || used >= n + 5) return stored();
  return used;
}

}  // namespace dcdeflate
