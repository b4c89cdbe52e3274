// Smoothing / interval path (S0-S5) as a wavefront scan.  One 256-thread workgroup per read.
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
__device__ __forceinline__ void word_prefix(const unsigned long long* words, int nw, unsigned* pref,
                                            unsigned* scratch) {
  // 256 threads x 4 consecutive words = 1024 words max
  const int w0 = threadIdx.x * 4;
  unsigned c[4];
  unsigned tsum = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    c[j] = (w0 + j < nw) ? (unsigned)__popcll(words[w0 + j]) : 0u;
    tsum += c[j];
  }
  unsigned total;
  unsigned ex = block_exclusive_scan_256(tsum, scratch, &total);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
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

template <bool FROM_LOGITS>
__global__ __launch_bounds__(256) void smooth_kernel(SmoothArgs a) {
  __shared__ unsigned long long s_raw[SM_MAX_WORDS];   // raw labels, later "kept run end" flags
  __shared__ unsigned long long s_sm[SM_MAX_WORDS];    // smoothed labels
  __shared__ unsigned s_pref[SM_MAX_WORDS + 1];
  __shared__ unsigned s_scratch[8];

  const int lane = threadIdx.x & 63;
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
    if (n > SM_MAX_LEN) n = SM_MAX_LEN;  // host rejects longer reads; defensive clamp
    if (n < 0) n = 0;
    const int nw = (n + 63) >> 6;
    const int n_pad = nw << 6;
    const int64_t out_off = a.out_offsets ? a.out_offsets[r] : 0;

    // ---- S0: argmax (first max wins) -> label bits ------------------------------------
    for (int i = threadIdx.x; i < n_pad; i += 256) {
      bool lab = false;
      if (i < n) {
        if (FROM_LOGITS) {
          const float2 v = *reinterpret_cast<const float2*>(lg + 2 * (int64_t)i);
          lab = v.y > v.x;  // class 1 only if strictly larger: ties -> 0
        } else {
          lab = lb[i] == 1;
        }
        if (FROM_LOGITS && a.pred) a.pred[out_off + i] = lab ? 1 : 0;
      }
      const unsigned long long m = __ballot(lab);
      if (lane == 0) s_raw[i >> 6] = m;
    }
    __syncthreads();
    word_prefix(s_raw, nw, s_pref, s_scratch);

    // ---- S1: majority_voting (smooth/utils.rs:48-97) -----------------------------------
    const int w = a.window, h = w >> 1;
    for (int i = threadIdx.x; i < n_pad; i += 256) {
      bool sm = false;
      if (i < n) {
        int start = i - h < 0 ? 0 : i - h;
        int end = i + h + 1 > n ? n : i + h + 1;
        if (end == n && (end - start) < w) start = end - w < 0 ? 0 : end - w;
        const int ones = (int)(ones_before(s_raw, s_pref, end) - ones_before(s_raw, s_pref, start));
        const int ln = end - start;
        if (2 * ones > ln) sm = true;
        else if (2 * ones < ln) sm = false;
        else sm = bit_at(s_raw, i);
        if (a.smooth) a.smooth[out_off + i] = sm ? 1 : 0;
      }
      const unsigned long long m = __ballot(sm);
      if (lane == 0) s_sm[i >> 6] = m;
    }
    __syncthreads();

    // ---- S2 + S3: run ends -> (start, end) with the index-0 quirk -> length filter -----
    // flag word: bit i set iff a kept interval ends at i (inclusive end)
    for (int i = threadIdx.x; i < n_pad; i += 256) {
      bool kept = false;
      if (i < n && bit_at(s_sm, i) && (i == n - 1 || !bit_at(s_sm, i + 1))) {
        int s = run_start(s_sm, i);
        const int e = i + 1;
        if (s == 0) s = 1;  // utils.rs:680-683: `start == 0` doubles as "no open run"
        kept = (s < e) && (e - s >= a.min_interval);
      }
      const unsigned long long m = __ballot(kept);
      if (lane == 0) s_raw[i >> 6] = m;
    }
    __syncthreads();
    word_prefix(s_raw, nw, s_pref, s_scratch);
    const int total = (int)s_pref[nw];
    const bool rejected = a.apply_approved_rule && total > a.approved;  // predict.rs:204-206
    const int n_out = rejected ? 0 : total;
    int32_t* iv = a.intervals + (int64_t)r * a.max_intervals * 2;
    if (!rejected) {
      for (int i = threadIdx.x; i < n; i += 256) {
        if (bit_at(s_raw, i)) {
          const int rank = (int)ones_before(s_raw, s_pref, i);
          if (rank < a.max_intervals) {
            int s = run_start(s_sm, i);
            if (s == 0) s = 1;
            iv[2 * rank] = s;
            iv[2 * rank + 1] = i + 1;
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
          const int s = iv[2 * j], e = iv[2 * j + 1];
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
  if (from_logits)
    hipLaunchKernelGGL(smooth_kernel<true>, dim3(grid), dim3(256), 0, as_stream(stream), a);
  else
    hipLaunchKernelGGL(smooth_kernel<false>, dim3(grid), dim3(256), 0, as_stream(stream), a);
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
