// Smoothing / interval path (S0-S5) as a wavefront scan.  One workgroup of 64 / 128 / 256 threads per read.
//
// Labels are held as bit-vectors in LDS (one 64-bit word per wavefront ballot).  A windowed
// majority needs the number of ones in [start, end); that is P(end) - P(start) with
// P(x) = word_prefix[x / 64] + popcount(word[x / 64] & low_bits(x % 64)); word_prefix is an
// exclusive scan of per-word popcounts (wave shuffles + one LDS hop).  Run detection, the
// length filter and the ordered compaction of the surviving intervals reuse the same
// ballot -> word-prefix machinery, so the only global traffic is the logits (8 B/base) in and
// the labels / a handful of intervals out.
#include "dc_common.h"

namespace dc {

constexpr int SM_MAX_LEN = 65536;              // bases per read supported by the LDS bit-vectors
constexpr int SM_MAX_WORDS = SM_MAX_LEN / 64;  // 1024
constexpr int SM_LDS_IV = 32;                  // kept intervals mirrored in LDS (the default approved count is 20)

struct SmoothArgs {
  // input (exactly one of logits / labels)
  const float* logits;    // [n_rows, row_stride, 2]
  int64_t row_stride;
  const int32_t* first;   // [n] first valid column (logits mode)
  const int32_t* len;     // [n] number of bases   (logits mode)
  const int8_t* labels;   // ragged at in_offsets  (labels mode)
  const int64_t* in_offsets;
  int n_reads;
  // parameters
  int window;             // already forced odd
  int min_interval;
  int approved;
  int apply_approved_rule;
  int max_intervals;
  // outputs
  const int64_t* out_offsets;  // ragged offsets for pred / smooth (labels mode: == in_offsets)
  int8_t* pred;
  int8_t* smooth;
  int32_t* intervals;
  int32_t* n_intervals;
  int32_t* keep;
  int32_t* n_keep;
};

// exclusive scan of popcounts of `words[0..nw)` into pref[0..nw]; pref[nw] = total.
template <int BS, int CAPW>
__device__ __forceinline__ void word_prefix(const unsigned long long* words, int nw, unsigned* pref,
                                            unsigned* scratch) {
  constexpr int WPT = (CAPW + BS - 1) / BS;  // consecutive words per thread
  const int w0 = threadIdx.x * WPT;
  unsigned c[WPT];
  unsigned tsum = 0;
#pragma unroll
  for (int j = 0; j < WPT; ++j) {
    c[j] = (w0 + j < nw) ? (unsigned)__popcll(words[w0 + j]) : 0u;
    tsum += c[j];
  }
  unsigned total;
  unsigned ex = block_exclusive_scan<BS>(tsum, scratch, &total);
#pragma unroll
  for (int j = 0; j < WPT; ++j) {
    if (w0 + j < nw) pref[w0 + j] = ex;
    ex += c[j];
  }
  if (threadIdx.x == 0) pref[nw] = total;
  __syncthreads();
}

__device__ __forceinline__ unsigned ones_before(const unsigned long long* words, const unsigned* pref, int x) {
  const int w = x >> 6, b = x & 63;
  unsigned p = pref[w];
  if (b) p += (unsigned)__popcll(words[w] & ((1ull << b) - 1ull));
  return p;
}

__device__ __forceinline__ bool bit_at(const unsigned long long* words, int x) {
  return (words[x >> 6] >> (x & 63)) & 1ull;
}

// start of the maximal run of ones that contains position i (bit i must be set)
__device__ __forceinline__ int run_start(const unsigned long long* words, int i) {
  int w = i >> 6;
  const int b = i & 63;
  unsigned long long zeros = ~words[w] & (b == 63 ? ~0ull >> 1 : ((1ull << b) - 1ull));
  // (bits strictly below b)
  if (b == 0) zeros = 0;
  while (true) {
    if (zeros) return (w << 6) + (63 - __clzll((long long)zeros)) + 1;
    if (w == 0) return 0;
    --w;
    zeros = ~words[w];
  }
}

#ifndef SM_OCC64
#define SM_OCC64 6  // waves per SIMD asked of the 64-thread form (79 VGPRs, no spill; the larger forms would spill)
#endif
#ifndef SM_S0_BATCH
#define SM_S0_BATCH 4
#endif

// 16 label bits -> 16 bytes of 0 / 1
__device__ __forceinline__ uint4 bits_to_bytes16(unsigned m) {
  auto q = [](unsigned x) { return (x & 1u) | ((x & 2u) << 7) | ((x & 4u) << 14) | ((x & 8u) << 21); };
  return make_uint4(q(m), q(m >> 4), q(m >> 8), q(m >> 12));
}

// BS threads per read, LDS bit-vectors of CAPW words.  Global traffic moves in chunks of 16 consecutive positions per
// thread: 8 x 16-byte logit loads, one 16-byte store of raw labels, one of smoothed labels (round 3; one position per
// lane meant 64-byte label stores per wave: 1.1 TB/s).  Short reads take the 64-thread form (<1 KiB of LDS, a "barrier"
// is a wave-local no-op, 28 reads resident per CU; a capacity of 16 words made hipcc turn the single-trip chunk loops into
// straight-line code with 197 VGPRs); the 256-thread form holds reads up to 65 536 bases.
template <bool FROM_LOGITS, int BS, int CAPW>
__global__ __launch_bounds__(BS, BS == 64 ? SM_OCC64 : 1) void smooth_kernel(SmoothArgs a) {
  __shared__ unsigned long long s_raw[CAPW];   // raw labels, later "kept run end" flags
  __shared__ unsigned long long s_sm[CAPW];    // smoothed labels
  __shared__ unsigned s_pref[CAPW + 1];
  __shared__ unsigned s_scratch[8];
  __shared__ int s_iv[2 * SM_LDS_IV];  // the first kept intervals, for the complement step (no global round trip)
  unsigned short* raw16 = reinterpret_cast<unsigned short*>(s_raw);  // chunk c = bits [16 (c & 3), +16) of word c >> 2
  unsigned short* sm16 = reinterpret_cast<unsigned short*>(s_sm);
  // logits mode: per wave, the 2-bit results of one stripe of 1 024 positions on their way from the lanes that loaded
  // them to the lanes that own their chunks
  __shared__ __attribute__((aligned(8))) unsigned char s_xb[FROM_LOGITS ? BS / 64 : 1][512];

  for (int r = blockIdx.x; r < a.n_reads; r += gridDim.x) {
    int n;
    const float* lg = nullptr;
    const int8_t* lb = nullptr;
    if (FROM_LOGITS) {
      n = a.len[r];
      lg = a.logits + ((int64_t)r * a.row_stride + a.first[r]) * 2;
    } else {
      n = (int)(a.in_offsets[r + 1] - a.in_offsets[r]);
      lb = a.labels + a.in_offsets[r];
    }
    // Contract (include/deepchopper_hip.h): 0 <= first[r], first[r] + len[r] <= row_stride, and no read longer than this
    // form's bit-vectors.  dc_tokenize_batch guarantees it; another caller's own lengths may not, and a silent clamp
    // would hand back truncated labels and intervals that look valid (and the stripe loop would read past the row).
    // Such a read is FLAGGED -- n_intervals[r] = n_keep[r] = -1, nothing else written for it -- and skipped.
    bool bad = n > CAPW * 64;
    if (FROM_LOGITS) bad = bad || a.first[r] < 0 || (int64_t)a.first[r] + n > a.row_stride;
    if (bad) {  // block-uniform
      if (threadIdx.x == 0) {
        a.n_intervals[r] = -1;
        if (a.n_keep) a.n_keep[r] = -1;
      }
      continue;
    }
    if (n < 0) n = 0;
    const int nw = (n + 63) >> 6;
    const int n_pad = nw << 6;
    const int nch = n_pad >> 4;
    const int64_t out_off = a.out_offsets ? a.out_offsets[r] : 0;

    // ---- S0: argmax (first max wins) -> label bits ------------------------------------
    // Logits: a chunk of 16 positions is 128 bytes, so "one chunk per lane" meant eight 16-byte loads per lane with a
    // lane stride of 128 bytes -- every instruction touched 64 lines for 16 bytes each, and with 28 waves per CU the lines
    // did not survive in the 32 KiB L1 until their next 16 bytes were asked for: L2 -> L1 moved each line up to eight
    // times (ablating the loads took 37 % off the kernel).  Now a wave walks a stripe of 1 024 positions with eight
    // loads of 1 KiB each (lane l of load j: positions 128 j + 2 l, + 1), and the two label bits of a lane and load go
    // through a 512-byte LDS array to the lane that owns their chunk: byte 64 j + l = 8 c + t holds bits 2 t, 2 t + 1 of
    // chunk c, which reads its eight bytes back as one 64-bit word and squeezes them to 16 bits with two multiplies.
    if (FROM_LOGITS && n >= 2) {
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
      unsigned char* xb = s_xb[wv];
      for (int st = wv; st * 1024 < n; st += BS / 64) {  // wave-uniform
        const int P = st * 1024;
        __builtin_amdgcn_wave_barrier();  // the previous stripe's bytes have been read (one wave: LDS operations are in order)
#pragma unroll
        for (int jb = 0; jb < 8; jb += SM_S0_BATCH) {  // loads in flight together (all eight cost 32 registers and a wave per SIMD)
          float4 v[SM_S0_BATCH];
#pragma unroll
          for (int j = 0; j < SM_S0_BATCH; ++j) {
            const int pos = P + 128 * (jb + j) + 2 * lane;
            const int pa = pos < n - 2 ? pos : n - 2;  // clamped into the read (results of clamped lanes are masked below)
            __builtin_memcpy(&v[j], lg + 2 * (int64_t)pa, 16);
          }
#pragma unroll
          for (int j = 0; j < SM_S0_BATCH; ++j) {
            const int pos = P + 128 * (jb + j) + 2 * lane;
            const unsigned a0 = v[j].y > v[j].x ? 1u : 0u, a1 = v[j].w > v[j].z ? 1u : 0u;  // class 1 only if strictly larger: ties -> 0
            // pos + 1 < n: both positions are this lane's; pos == n - 1: the clamped load holds it in its second half
            const unsigned two = pos + 1 < n ? (a0 | a1 << 1) : pos < n ? a1 : 0u;
            xb[64 * (jb + j) + lane] = (unsigned char)two;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned long long x = *reinterpret_cast<const unsigned long long*>(xb + 8 * lane);
        // bytes b0 .. b3 (two bits each) of a 32-bit half -> b0 | b1 << 2 | b2 << 4 | b3 << 6 in the product's top byte
        const unsigned m = (((unsigned)x * 0x01041040u) >> 24) | ((((unsigned)(x >> 32) * 0x01041040u) >> 24) << 8);
        const int c = 64 * st + lane, p0 = 16 * c;
        if (c < nch) raw16[c] = (unsigned short)m;
        if (a.pred) {
          if (p0 + 16 <= n) {
            const uint4 o = bits_to_bytes16(m);
            __builtin_memcpy(a.pred + out_off + p0, &o, 16);
          } else {
            for (int k = 0; k < 16 && p0 + k < n; ++k) a.pred[out_off + p0 + k] = (m >> k) & 1u;
          }
        }
      }
    } else
    for (int c = threadIdx.x; c < nch; c += BS) {
      const int p0 = 16 * c;
      unsigned m = 0;
      if (p0 + 16 <= n) {
        if (FROM_LOGITS) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float4 v;
            __builtin_memcpy(&v, lg + 2 * (int64_t)p0 + 4 * j, 16);
            m |= (v.y > v.x ? 1u : 0u) << (2 * j);  // class 1 only if strictly larger: ties -> 0
            m |= (v.w > v.z ? 1u : 0u) << (2 * j + 1);
          }
        } else {
          uint4 v;
          __builtin_memcpy(&v, lb + p0, 16);
          const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < 16; ++k) m |= (((wv[k >> 2] >> (8 * (k & 3))) & 255u) == 1u ? 1u : 0u) << k;
        }
        if (FROM_LOGITS && a.pred) {
          const uint4 o = bits_to_bytes16(m);
          __builtin_memcpy(a.pred + out_off + p0, &o, 16);
        }
      } else {
        for (int k = 0; k < 16 && p0 + k < n; ++k) {
          bool lab;
          if (FROM_LOGITS) {
            const float2 v = *reinterpret_cast<const float2*>(lg + 2 * (int64_t)(p0 + k));
            lab = v.y > v.x;
          } else {
            lab = lb[p0 + k] == 1;
          }
          m |= (lab ? 1u : 0u) << k;
          if (FROM_LOGITS && a.pred) a.pred[out_off + p0 + k] = lab ? 1 : 0;
        }
      }
      raw16[c] = (unsigned short)m;
    }
    __syncthreads();
    const int w = a.window, h = w >> 1;
    const bool short_window = h >= 1 && w <= 31;
    if (!short_window) word_prefix<BS, CAPW>(s_raw, nw, s_pref, s_scratch);  // the general window form counts through prefixes

    // ---- S1: majority_voting (smooth/utils.rs:48-97) -----------------------------------
    if (short_window) {
      // Windows of at most 31 positions.  Away from the read's ends the window of position i is [i - h, i + h]: 2 h + 1
      // positions, odd, so no tie and no clamping -- a funnel shift, an and, a popcount and one compare per position.
      // Every lane runs that form on all 16 positions of its chunk (zero bits stand in beyond the ends); the at most
      // 2 h positions whose window is cut by an end of the read (i < h, i >= n - h: clamped start / end, the
      // "end == n" re-anchoring, ties) are then redone with the general rule, ONE position per lane of the first wave,
      // and patched into the bit-vector -- instead of every lane dragging the general rule through all its positions
      // for the sake of one or two lanes (~20 instructions per position, a sixth of the kernel).
      const unsigned mask_w = (1u << (2 * h + 1)) - 1u;
      for (int c = threadIdx.x; c < nch; c += BS) {
        // positions [16 c - 32, 16 c + 32) of the bit-vector as two 32-bit halves, shifted so that bit k is position 16 c + k - h
        const unsigned lo = (c >= 2 ? (unsigned)raw16[c - 2] : 0u) | (c >= 1 ? (unsigned)raw16[c - 1] << 16 : 0u);
        const unsigned hi = (unsigned)raw16[c] | (c + 1 < nch ? (unsigned)raw16[c + 1] << 16 : 0u);
        const unsigned x_lo = __builtin_amdgcn_alignbit(hi, lo, 32u - (unsigned)h), x_hi = hi >> (32u - (unsigned)h);
        unsigned m = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const unsigned win = k == 0 ? x_lo : __builtin_amdgcn_alignbit(x_hi, x_lo, (unsigned)k);
          m |= (__popc(win & mask_w) > h ? 1u : 0u) << k;
        }
        sm16[c] = (unsigned short)m;
      }
      __syncthreads();
      if (threadIdx.x < 64) {
        const int l = threadIdx.x;
        const int n_left = h < n ? h : n;                      // positions 0 .. n_left - 1
        const int r0 = n - h > n_left ? n - h : n_left;        // positions r0 .. n - 1
        const int n_right = n - r0;
        const int i = l < n_left ? l : r0 + (l - n_left);
        bool sm = false;
        if (l < n_left + n_right) {
          int start = i - h < 0 ? 0 : i - h;
          int end = i + h + 1 > n ? n : i + h + 1;
          if (end == n && (end - start) < w) start = end - w < 0 ? 0 : end - w;
          const int ln = end - start;  // 1 .. 31
          const unsigned* raw32 = reinterpret_cast<const unsigned*>(s_raw);
          const int wi = start >> 5;
          const unsigned b_lo = raw32[wi], b_hi = wi + 1 < 2 * CAPW ? raw32[wi + 1] : 0u;
          const unsigned win = __builtin_amdgcn_alignbit(b_hi, b_lo, (unsigned)(start & 31));
          const int ones = __popc(win & ((1u << ln) - 1u));
          if (2 * ones > ln) sm = true;
          else if (2 * ones < ln) sm = false;
          else sm = bit_at(s_raw, i);
        }
        const unsigned long long bal = __ballot(sm);
        if (l == 0 && n > 0) {
          const unsigned long long m_left = (1ull << n_left) - 1ull, m_right = (1ull << n_right) - 1ull;  // < 16 bits each
          s_sm[0] = (s_sm[0] & ~m_left) | (bal & m_left);
          if (n_right > 0) {
            const unsigned long long bits = (bal >> n_left) & m_right;
            const int wr = r0 >> 6, sr = r0 & 63;
            s_sm[wr] = (s_sm[wr] & ~(m_right << sr)) | (bits << sr);
            if (sr + n_right > 64) s_sm[wr + 1] = (s_sm[wr + 1] & ~(m_right >> (64 - sr))) | (bits >> (64 - sr));
          }
        }
      }
    } else {
      for (int c = threadIdx.x; c < nch; c += BS) {
        const int p0 = 16 * c;
        unsigned m = 0;
        for (int k = 0; k < 16 && p0 + k < n; ++k) {
          const int i = p0 + k;
          int start = i - h < 0 ? 0 : i - h;
          int end = i + h + 1 > n ? n : i + h + 1;
          if (end == n && (end - start) < w) start = end - w < 0 ? 0 : end - w;
          const int ones = (int)(ones_before(s_raw, s_pref, end) - ones_before(s_raw, s_pref, start));
          const int ln = end - start;
          bool sm;
          if (2 * ones > ln) sm = true;
          else if (2 * ones < ln) sm = false;
          else sm = bit_at(s_raw, i);
          if (sm) m |= 1u << k;
        }
        sm16[c] = (unsigned short)m;
      }
    }
    __syncthreads();

    // ---- S2 + S3: run ends -> (start, end) with the index-0 quirk -> length filter -----
    // flag word: bit i set iff a kept interval ends at i (inclusive end).  Per chunk of 16 positions: run ends are the
    // set bits whose successor is clear (positions >= n are clear) -- a handful per read, so only those walk back to
    // their run's start
    for (int c = threadIdx.x; c < nch; c += BS) {
      const unsigned m = sm16[c];
      if (a.smooth) {
        const int p0 = 16 * c;
        if (p0 + 16 <= n) {
          const uint4 o = bits_to_bytes16(m);
          __builtin_memcpy(a.smooth + out_off + p0, &o, 16);
        } else {
          for (int k = 0; k < 16 && p0 + k < n; ++k) a.smooth[out_off + p0 + k] = (m >> k) & 1u;
        }
      }
      const unsigned nxt = c + 1 < nch ? (sm16[c + 1] & 1u) : 0u;
      unsigned ends = m & ~((m >> 1) | (nxt << 15));
      unsigned kept = 0;
      while (ends) {
        const int k = __builtin_ctz(ends);
        ends &= ends - 1;
        const int i = 16 * c + k;
        int s = run_start(s_sm, i);
        const int e = i + 1;
        if (s == 0) s = 1;  // utils.rs:680-683: `start == 0` doubles as "no open run"
        if ((s < e) && (e - s >= a.min_interval)) kept |= 1u << k;
      }
      raw16[c] = (unsigned short)kept;
    }
    __syncthreads();
    word_prefix<BS, CAPW>(s_raw, nw, s_pref, s_scratch);
    const int total = (int)s_pref[nw];
    const bool rejected = a.apply_approved_rule && total > a.approved;  // predict.rs:204-206
    const int n_out = rejected ? 0 : total;
    int32_t* iv = a.intervals + (int64_t)r * a.max_intervals * 2;
    if (!rejected) {
      for (int c = threadIdx.x; c < nch; c += BS) {
        unsigned m = raw16[c];
        while (m) {
          const int k = __builtin_ctz(m);
          m &= m - 1;
          const int i = 16 * c + k;
          const int rank = (int)ones_before(s_raw, s_pref, i);
          if (rank < a.max_intervals) {
            int s = run_start(s_sm, i);
            if (s == 0) s = 1;
            iv[2 * rank] = s;
            iv[2 * rank + 1] = i + 1;
            if (rank < SM_LDS_IV) {
              s_iv[2 * rank] = s;
              s_iv[2 * rank + 1] = i + 1;
            }
          }
        }
      }
    }
    if (threadIdx.x == 0) a.n_intervals[r] = n_out;
    __syncthreads();  // interval stores by this block visible to thread 0 below (same block: LDS not involved,
                      // global writes are ordered by the barrier + the fence)
    __threadfence_block();

    // ---- S5: generate_unmaped_intervals (output/split.rs:260-292) ----------------------
    if (a.keep && threadIdx.x == 0) {
      int32_t* kp = a.keep + (int64_t)r * (a.max_intervals + 1) * 2;
      int cnt = 0;
      const int m = n_out < a.max_intervals ? n_out : a.max_intervals;
      if (m == 0) {
        kp[0] = 0;
        kp[1] = n;
        cnt = 1;
      } else {
        int cur = 0;
        for (int j = 0; j < m; ++j) {
          const int s = j < SM_LDS_IV ? s_iv[2 * j] : iv[2 * j], e = j < SM_LDS_IV ? s_iv[2 * j + 1] : iv[2 * j + 1];
          if (cur < s) {
            kp[2 * cnt] = cur;
            kp[2 * cnt + 1] = s;
            ++cnt;
          }
          cur = e;
        }
        if (cur < n - 1) {  // the tail segment ends at n-1: the last base is dropped (split.rs:287-289)
          kp[2 * cnt] = cur;
          kp[2 * cnt + 1] = n - 1;
          ++cnt;
        }
      }
      if (a.n_keep) a.n_keep[r] = cnt;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void summary_predict_kernel(const int8_t* __restrict__ pred,
                                                              const int8_t* __restrict__ labels, int n_rows,
                                                              int64_t row_stride, int ignore,
                                                              const int64_t* __restrict__ out_offsets,
                                                              int8_t* __restrict__ out_pred,
                                                              int8_t* __restrict__ out_labels,
                                                              int32_t* __restrict__ counts) {
  __shared__ unsigned scratch[8];
  for (int r = blockIdx.x; r < n_rows; r += gridDim.x) {
    const int8_t* p = pred + (int64_t)r * row_stride;
    const int8_t* l = labels + (int64_t)r * row_stride;
    const int64_t off = out_offsets ? out_offsets[r] : 0;
    unsigned written = 0;
    for (int64_t base = 0; base < row_stride; base += 256) {
      const int64_t i = base + threadIdx.x;
      int8_t lv = 0, pv = 0;
      unsigned keep = 0;
      if (i < row_stride) {
        lv = l[i];
        pv = p[i];
        keep = lv != (int8_t)ignore;
      }
      unsigned tot;
      const unsigned pos = block_exclusive_scan_256(keep, scratch, &tot);
      if (keep && out_pred) {
        out_pred[off + written + pos] = pv;
        if (out_labels) out_labels[off + written + pos] = lv;
      }
      written += tot;
    }
    if (threadIdx.x == 0 && counts) counts[r] = (int32_t)written;
  }
}

static int launch_smooth(void* stream, SmoothArgs& a, bool from_logits) {
  DC_REQUIRE(a.window >= 1, "window must be >= 1");
  DC_REQUIRE(a.max_intervals >= 1, "max_intervals must be >= 1");
  DC_REQUIRE(!a.apply_approved_rule || a.max_intervals >= a.approved, "max_intervals must be >= approved");
  DC_REQUIRE(a.intervals && a.n_intervals, "intervals / n_intervals must not be NULL");
  if ((a.window & 1) == 0) a.window += 1;  // smooth/utils.rs:50-54
  int grid = a.n_reads < 65535 ? a.n_reads : 65535;
  // logits mode: no read is longer than its row, so the row stride picks the form; ragged label input (the `.pt`
  // route's chop) has its lengths on the device only and takes the general form
  const int64_t max_len = from_logits ? a.row_stride : SM_MAX_LEN;
  if (from_logits && max_len <= 2048)
    hipLaunchKernelGGL((smooth_kernel<true, 64, 32>), dim3(grid), dim3(64), 0, as_stream(stream), a);
  else if (from_logits && max_len <= 4096)
    hipLaunchKernelGGL((smooth_kernel<true, 128, 64>), dim3(grid), dim3(128), 0, as_stream(stream), a);
  else if (from_logits)
    hipLaunchKernelGGL((smooth_kernel<true, 256, SM_MAX_WORDS>), dim3(grid), dim3(256), 0, as_stream(stream), a);
  else
    hipLaunchKernelGGL((smooth_kernel<false, 256, SM_MAX_WORDS>), dim3(grid), dim3(256), 0, as_stream(stream), a);
  DC_LAUNCH_CHECK();
  return DC_OK;
}

}  // namespace dc

using namespace dc;

extern "C" int dc_smooth_intervals_from_logits(void* stream, const float* d_logits, int64_t row_stride,
                                               const int32_t* d_first, const int32_t* d_len, int32_t n_reads,
                                               const int64_t* d_out_offsets, int32_t window, int32_t min_interval,
                                               int32_t approved, int32_t max_intervals, int8_t* d_pred,
                                               int8_t* d_smooth, int32_t* d_intervals, int32_t* d_n_intervals,
                                               int32_t* d_keep, int32_t* d_n_keep) {
  if (n_reads <= 0) return DC_OK;
  DC_REQUIRE(d_logits && d_first && d_len, "null pointer");
  DC_REQUIRE((d_pred == nullptr && d_smooth == nullptr) || d_out_offsets, "label outputs need d_out_offsets");
  SmoothArgs a{};
  a.logits = d_logits;
  a.row_stride = row_stride;
  a.first = d_first;
  a.len = d_len;
  a.n_reads = n_reads;
  a.window = window;
  a.min_interval = min_interval;
  a.approved = approved;
  a.apply_approved_rule = 1;
  a.max_intervals = max_intervals;
  a.out_offsets = d_out_offsets;
  a.pred = d_pred;
  a.smooth = d_smooth;
  a.intervals = d_intervals;
  a.n_intervals = d_n_intervals;
  a.keep = d_keep;
  a.n_keep = d_n_keep;
  return launch_smooth(stream, a, true);
}

extern "C" int dc_smooth_intervals_from_labels(void* stream, const int8_t* d_labels, const int64_t* d_offsets,
                                               int32_t n_reads, int32_t window, int32_t min_interval,
                                               int32_t approved, int32_t apply_approved_rule,
                                               int32_t max_intervals, int8_t* d_smooth, int32_t* d_intervals,
                                               int32_t* d_n_intervals, int32_t* d_keep, int32_t* d_n_keep) {
  if (n_reads <= 0) return DC_OK;
  DC_REQUIRE(d_labels && d_offsets, "null pointer");
  SmoothArgs a{};
  a.labels = d_labels;
  a.in_offsets = d_offsets;
  a.n_reads = n_reads;
  a.window = window;
  a.min_interval = min_interval;
  a.approved = approved;
  a.apply_approved_rule = apply_approved_rule;
  a.max_intervals = max_intervals;
  a.out_offsets = d_offsets;
  a.pred = nullptr;
  a.smooth = d_smooth;
  a.intervals = d_intervals;
  a.n_intervals = d_n_intervals;
  a.keep = d_keep;
  a.n_keep = d_n_keep;
  return launch_smooth(stream, a, false);
}

extern "C" int dc_summary_predict(void* stream, const int8_t* d_pred, const int8_t* d_labels, int32_t n_rows,
                                  int64_t row_stride, int32_t ignore_label, const int64_t* d_out_offsets,
                                  int8_t* d_out_pred, int8_t* d_out_labels, int32_t* d_counts) {
  if (n_rows <= 0) return DC_OK;
  DC_REQUIRE(d_pred && d_labels, "null pointer");
  DC_REQUIRE(d_out_pred == nullptr || d_out_offsets, "compacted output needs d_out_offsets");
  int grid = n_rows < 65535 ? n_rows : 65535;
  hipLaunchKernelGGL(summary_predict_kernel, dim3(grid), dim3(256), 0, as_stream(stream), d_pred, d_labels, n_rows,
                     row_stride, ignore_label, d_out_offsets, d_out_pred, d_out_labels, d_counts);
  DC_LAUNCH_CHECK();
  return DC_OK;
}
