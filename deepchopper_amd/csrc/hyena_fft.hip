// O(L log L) long convolution of the Hyena operator for long reads (the reference's `fftconv`,
// HF-Hub remote code called at deepchopper/models/llm/hyena.py:34-41; restated in oracle/forward_ref.py:68-77).
//
// hyena3_kernel (forward_kernels.hip) evaluates the causal convolution as block-Toeplitz MFMA products: O(L^2) per
// (read, channel), unbeatable at 1 kb but 43 % of the step on a 100 bp - 10 kb length mix.  Reads above a length
// threshold take this path instead: one workgroup per (read, channel) runs
//     z = shortconv(x1) * shortconv(v)                      (gated input, as the z tiles of hyena3)
//     c[m] = z[2m] + i z[2m+1]                              (real sequence of 2M samples as M complex points)
//     C = FFT_M(c)   ->  Z[k] (unpack)  ->  Y = Z * K       (K = spectrum of the layer's filter taps, built once)
//     C'[k] (repack)  ->  c' = IFFT_M(C')                   (IFFT as conj . FFT . conj)
//     y[t] = c'-sample[t] * shortconv(x0)[t],  t < L
// entirely in LDS: M = 2^LOGM >= L complex points (2M real samples >= 2L: the circular convolution is the linear
// one on [0, L)), fp32 throughout like the reference.  The FFT is a Stockham autosort radix-16 decimation in time
// with the butterflies in registers (M / 16 threads, 16 points each per pass; the last pass has radix M / 16^k);
// LDS points carry one pad slot per 16 (the first pass writes with a stride of 16 points: 17 slots = 34 dwords
// is conflict free for ds_write_b64's 16-lane groups).  Twiddles come from one exact table (fp64 sincos once per
// process) through four lookups w, w^2, w^4, w^8 and at most three products.
// No rocFFT / hipFFT: the transform, its packing tricks and the gating are one kernel.
#include "dc_common.h"
#include "forward.h"

namespace dc {

namespace {

constexpr int FFT_TW_N = 65536;  // twiddle table: exp(-2 pi i n / 65536) (the 2^15-point form needs the 2^16-th roots)
// The split transform of 3 x 4096 points needs roots the 2^16 table does not hold: the 24 576-th roots (W_2M for M = 12 288)
// stand behind it in the same buffer.
constexpr int FFT_TW3_N = 24576;
constexpr int FFT_TW3_OFF = FFT_TW_N, FFT_TW_TOTAL = FFT_TW_N + FFT_TW3_N;

__global__ __launch_bounds__(256) void fft_twiddle_kernel(float2* __restrict__ tw, int count) {  // tw[n] = exp(-2 pi i n / count)
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= count) return;
  double s, c;
  sincospi(-2.0 * (double)n / (double)count, &s, &c);
  tw[n] = make_float2((float)c, (float)s);
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
// two-lane fp32 vectors: the pointwise phases are written on these so that they compile to v_pk_* instructions
// (left as scalar float2 code the SLP vectoriser packed the butterflies but none of the pointwise loops)
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f V(float2 a) { return v2f{a.x, a.y}; }
__device__ __forceinline__ float2 F(v2f a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ v2f vfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f vcmul(v2f a, v2f b) {  // complex product
  return vfma(v2f{-a.y, a.y}, b.yx, a.xx * b);
}
__device__ __forceinline__ v2f vconj(v2f a) { return v2f{a.x, -a.y}; }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

constexpr int bitrev(int x, int bits) {
  int r = 0;
  for (int b = 0; b < bits; ++b) r |= ((x >> b) & 1) << (bits - 1 - b);
  return r;
}
constexpr int ilog2(int x) { return x <= 1 ? 0 : 1 + ilog2(x >> 1); }

// exp(-2 pi i m / n) for the in-register butterflies (n <= 16): exact to fp32 rounding
__device__ __forceinline__ float2 unit_root(int m, int n) {
  // multiples of pi/8: cos / sin of k * 22.5 degrees
  constexpr float C[5] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.0f};
  const int k = (m * 16 / n) & 15;  // angle = -2 pi k / 16
  const int q = k & 3, quad = k >> 2;
  float c = C[q], s = C[4 - q];  // cos, sin of q * 22.5 deg
  // rotate by quadrant: angle = -(quad * 90 + q * 22.5) degrees
  float cr, sr;
  switch (quad) {
    case 0: cr = c; sr = -s; break;
    case 1: cr = -s; sr = -c; break;
    case 2: cr = -c; sr = s; break;
    default: cr = s; sr = c; break;
  }
  return make_float2(cr, sr);
}

// In-register DFT of R points (decimation in frequency, radix 2): frequency f ends up at position bitrev(f).
template <int R>
__device__ __forceinline__ void dft_regs(float2 (&u)[R]) {
#pragma unroll
  for (int s = R / 2; s >= 1; s >>= 1) {
#pragma unroll
    for (int blk = 0; blk < R; blk += 2 * s) {
#pragma unroll
      for (int m = 0; m < s; ++m) {
        const float2 a = u[blk + m], b = u[blk + m + s];
        u[blk + m] = cadd(a, b);
        const float2 d = csub(a, b);
        u[blk + m + s] = m == 0 ? d : cmul(d, unit_root(m, 2 * s));
      }
    }
  }
}

__device__ __forceinline__ int pad16(int idx) { return idx + (idx >> 4); }

// Twiddles of one pass for one thread: w = exp(-2 pi i k / (p R)) for its NB butterflies, as w, w^2, w^4, w^8 (the
// other powers are at most three products away).  They come from the exact table in global memory (L2): a pass
// requests the twiddles of the NEXT pass before it starts its own work, so their round trip (several hundred cycles,
// longer than a pass's LDS reads + barrier) is hidden behind a whole pass instead of parking every wave once per pass.
template <int LOGM, int R>
struct PassTw {
  static constexpr int NB = 16 / R;
  float2 w1[NB], w2[NB], w4[NB], w8[NB];
  __device__ __forceinline__ void load(const float2* __restrict__ tw, int p, int tid) {
    constexpr int T = (1 << LOGM) / 16;
    if (p <= 1) return;
    const int step = FFT_TW_N / (p * R);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int k = (tid + b * T) & (p - 1);
      w1[b] = tw[(k * step) & (FFT_TW_N - 1)];
      if (R > 2) w2[b] = tw[(2 * k * step) & (FFT_TW_N - 1)];
      if (R > 4) w4[b] = tw[(4 * k * step) & (FFT_TW_N - 1)];
      if (R > 8) w8[b] = tw[(8 * k * step) & (FFT_TW_N - 1)];
    }
  }
};

// One Stockham pass of radix R over the M points in LDS; p = product of the radices of the passes before it.
// T = M / 16 threads; every thread owns 16 / R butterflies.  Loads of all threads precede all stores (two
// barriers), so the pass runs in place.  `w` holds this pass's twiddles (requested during the previous pass).
template <int LOGM, int R>
__device__ __forceinline__ void fft_pass(float2* __restrict__ s, const PassTw<LOGM, R>& w, int p, int tid) {
  constexpr int M = 1 << LOGM, T = M / 16, NB = 16 / R, LOGR = ilog2(R);
  float2 u[NB][R];
  // M / R and b T are multiples of 16, so the pad of a point is that of (tid) plus a constant: one base address,
  // immediate offsets
  const float2* sr = s + pad16(tid);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int t = 0; t < R; ++t) {
      constexpr int S = M / R;
      u[b][t] = sr[(b * T + t * S) + ((b * T + t * S) >> 4)];
    }
  }
  __syncthreads();
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int i = tid + b * T;
    const int k = i & (p - 1);
    if (p > 1) {
      float2 wp[16];
      wp[1] = w.w1[b];
      if (R > 2) wp[2] = w.w2[b];
      if (R > 4) wp[4] = w.w4[b];
      if (R > 8) wp[8] = w.w8[b];
#pragma unroll
      for (int t = 3; t < R; ++t) {
        if ((t & (t - 1)) != 0) {  // not a power of two: highest set bit times the rest
          const int hb = 1 << ilog2(t);
          wp[t] = cmul(wp[hb], wp[t - hb]);
        }
      }
#pragma unroll
      for (int t = 1; t < R; ++t) u[b][t] = cmul(u[b][t], wp[t]);
    }
    dft_regs<R>(u[b]);
    const int j = ((i - k) << LOGR) + k;
    if (p >= 16) {  // p is a multiple of 16: constant offsets from one padded base
      float2* sw = s + pad16(j);
#pragma unroll
      for (int t = 0; t < R; ++t) sw[bitrev(t, LOGR) * (p + (p >> 4))] = u[b][t];
    } else {        // first pass (p = 1, j = R i with R = 16): slot 17 i + t
      float2* sw = s + pad16(j);
#pragma unroll
      for (int t = 0; t < R; ++t) sw[bitrev(t, LOGR)] = u[b][t];
    }
  }
  __syncthreads();
}

template <int LOGM>
__device__ __forceinline__ void fft_lds(float2* __restrict__ s, const float2* __restrict__ tw, int tid) {
  // radices: 16 while they fit, then the rest (2, 4 or 8).  Twiddle requests run one pass ahead.
  constexpr int R3 = LOGM == 10 ? 4 : LOGM == 11 ? 8 : 16;   // third pass
  constexpr int R4 = LOGM == 13 ? 2 : LOGM == 14 ? 4 : 0;    // fourth pass (2^13, 2^14 only)
  PassTw<LOGM, 16> w1, w2;
  PassTw<LOGM, R3> w3;
  w2.load(tw, 16, tid);
  fft_pass<LOGM, 16>(s, w1, 1, tid);
  w3.load(tw, 256, tid);
  fft_pass<LOGM, 16>(s, w2, 16, tid);
  if constexpr (R4 != 0) {
    PassTw<LOGM, R4> w4;
    w4.load(tw, 4096, tid);
    fft_pass<LOGM, R3>(s, w3, 256, tid);
    fft_pass<LOGM, R4>(s, w4, 4096, tid);
  } else {
    fft_pass<LOGM, R3>(s, w3, 256, tid);
  }
}

// Spectrum of a real sequence r[0 .. 2M) from C = FFT_M(r[2m] + i r[2m+1]), evaluated pairwise:
//   Z[k] = E + W^k O,  Z[M-k] = conj(E - W^k O),  E = (C[k] + conj(C[M-k])) / 2,  O = -i (C[k] - conj(C[M-k])) / 2
template <int LOGM>
__device__ __forceinline__ void unpack_pair(float2 ck, float2 cmk, float2 wk, float2& zk, float2& zmk) {
  const float2 b = cconj(cmk);
  const float2 e = make_float2(0.5f * (ck.x + b.x), 0.5f * (ck.y + b.y));
  const float2 d = make_float2(0.5f * (ck.x - b.x), 0.5f * (ck.y - b.y));
  const float2 o = make_float2(d.y, -d.x);  // -i d
  const float2 wo = cmul(wk, o);
  zk = cadd(e, wo);
  zmk = cconj(csub(e, wo));
}

// ---------------------------------------------------------------------------------------------------
// filter spectra: K[c][0 .. M] = rfft(kf[c][0 .. min(M, taps)), n = 2M)
// ---------------------------------------------------------------------------------------------------
template <int LOGM>
__global__ __launch_bounds__((1 << LOGM) / 16) void hyena_fft_filter_kernel(const float* __restrict__ kf, int64_t kf_stride,
                                                                              int kf_taps, const float2* __restrict__ tw,
                                                                              float2* __restrict__ spec) {
  constexpr int M = 1 << LOGM, T = M / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  float2* s = reinterpret_cast<float2*>(s_raw);
  const int c = blockIdx.x, tid = threadIdx.x;
  const float* k = kf + (int64_t)c * kf_stride;
  for (int m = tid; m < M; m += T) {
    const int t = 2 * m;
    s[pad16(m)] = make_float2(t < kf_taps && t < M ? k[t] : 0.f, t + 1 < kf_taps && t + 1 < M ? k[t + 1] : 0.f);
  }
  __syncthreads();
  fft_lds<LOGM>(s, tw, tid);
  float2* out = spec + (int64_t)c * (M + 1);
  for (int kk = tid; kk <= M / 2; kk += T) {
    const float2 ck = s[pad16(kk)], cmk = s[pad16((M - kk) & (M - 1))];
    float2 zk, zmk;
    unpack_pair<LOGM>(ck, cmk, tw[kk * (FFT_TW_N / (2 * M))], zk, zmk);
    if (kk == 0) {
      out[0] = make_float2(ck.x + ck.y, 0.f);  // E[0] + O[0]
      out[M] = make_float2(ck.x - ck.y, 0.f);  // Nyquist: E[0] - O[0]
    } else {
      out[kk] = zk;
      out[M - kk] = zmk;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// the operator for one (read, channel)
// ---------------------------------------------------------------------------------------------------
template <int LOGM, bool TAB, int SP>  // SP: 1 = reads with a shared [PAD] prefix (HyenaArgs::pv / padoff), 2 = its linear form (+ dbase / zc / Fc)
__global__ __launch_bounds__((1 << LOGM) / 16, 4) void hyena_fft_kernel(HyenaArgs a, const int32_t* __restrict__ reads,
                                                                       const float2* __restrict__ tw,
                                                                       const float2* __restrict__ spec) {
  constexpr int M = 1 << LOGM, T = M / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  float2* s = reinterpret_cast<float2*>(s_raw);
  float* s_tab = reinterpret_cast<float*>(s + (M + M / 16));  // TAB: x0 | x1 | v of this channel for ids 0..15, [16] = 0
  const int r = reads[blockIdx.x], c = blockIdx.y, tid = threadIdx.x;
  const int toff = a.toff[r];
  const int L = a.ntok[r];
  const int D = a.D;
  constexpr bool DL = SP == 2;  // y = (k * (z - z_pad) + F) * gate: the transform covers the read's own positions only
  const int pv = SP ? a.pv[r] : 0, padoff = SP ? a.padoff[r] : 0;  // positions < pv live in the shared [PAD] row (pv: a multiple of 4)
  auto slot = [&](int t) -> int { return (SP && t < pv ? padoff : toff) + t; };
  const float* zcrow = DL ? a.zc + (int64_t)c * a.dstride + a.dbase[r] : nullptr;  // the shared row's z / convolution at the read's positions
  const float* fcrow = DL ? a.Fc + (int64_t)c * a.dstride + a.dbase[r] : nullptr;
  const float* ux0 = a.u + (int64_t)c * a.Tp;
  const float* ux1 = a.u + (int64_t)(D + c) * a.Tp;
  const float* uv = a.u + (int64_t)(2 * D + c) * a.Tp;
  const unsigned char* tok = TAB ? a.tok : nullptr;
  if (TAB) {
    if (tid < 3 * 17) {
      const int k3 = tid / 17, id = tid - 17 * k3;
      s_tab[tid] = id < a.vocab && id < 16 ? a.tab[(int64_t)id * 3 * D + k3 * D + c] : 0.f;
    }
    __syncthreads();
  }
  // short causal filter (order 3, weight index 0 = oldest tap) of row group k3 (0 x0, 1 x1, 2 v) at tokens t, t + 1
  // (t even).  Loading and evaluating are separate steps: every pointwise loop below first issues the loads of UB
  // iterations and only then consumes them -- written as one load-then-use body per iteration, each iteration paid a
  // full memory round trip (16 + 8 + 16 of them in sequence were ~80 % of a workgroup's life).
  //   non-TAB: u[t-2 .. t+1] as two aligned 8-byte loads (reads start at multiples of 32 tokens; slots between L and
  //            the end of the read's last 32-block are readable and masked here)
  //   TAB:     the four token ids as two 16-bit loads; values from the 17-entry table in LDS (0xFF -> entry 16 = 0)
  struct Raw {
    float2 lo, hi;  // non-TAB: u[t-2], u[t-1] | u[t], u[t+1].   TAB: lo.x carries the four ids
  };
  // Branch-free: the addresses are clamped into the read and the values masked afterwards -- a load inside a
  // divergent branch is followed by its own s_waitcnt vmcnt(0), which would serialise the batch again.
  const int t_last = L > 0 ? (L - 1) & ~1 : 0;  // last even token position of the read
  auto raw_load = [&](const float* row, int t) -> Raw {
    Raw r;
    const int tc = t < t_last ? t : t_last;
    const int tl = tc >= 2 ? tc - 2 : 0;
    const int sc = slot(tc), sl = slot(tl);
    if (TAB) {
      const uint32_t hi16 = *reinterpret_cast<const unsigned short*>(tok + sc);
      const uint32_t lo16 = *reinterpret_cast<const unsigned short*>(tok + sl);
      uint32_t ids = (t >= 2 ? lo16 : 0xFFFFu) | (hi16 << 16);
      if (t + 1 >= L) ids |= 0xFF000000u;
      if (t >= L) ids = 0xFFFFFFFFu;
      r.lo = make_float2(__builtin_bit_cast(float, ids), 0.f);
      r.hi = make_float2(0.f, 0.f);
      return r;
    }
    const float2 lo = *reinterpret_cast<const float2*>(row + sl);
    const float2 hi = *reinterpret_cast<const float2*>(row + sc);
    const bool in = t < L;
    r.lo = make_float2(in && t >= 2 ? lo.x : 0.f, in && t >= 2 ? lo.y : 0.f);
    r.hi = make_float2(in ? hi.x : 0.f, in && t + 1 < L ? hi.y : 0.f);
    return r;
  };
  float sw_[3][3], sb_[3];
#pragma unroll
  for (int k3 = 0; k3 < 3; ++k3) {
#pragma unroll
    for (int j = 0; j < 3; ++j) sw_[k3][j] = a.sw[(k3 * D + c) * 3 + j];
    sb_[k3] = a.sb[k3 * D + c];
  }
  auto sc_eval = [&](int k3, const Raw& r) -> v2f {  // the filter's outputs at t, t + 1
    v2f lo, hi;
    if (TAB) {
      const uint32_t ids = __builtin_bit_cast(uint32_t, r.lo.x);
      auto val = [&](int byte) -> float {
        const uint32_t id = (ids >> (8 * byte)) & 255u;
        return s_tab[k3 * 17 + (id < 16u ? id : 16u)];
      };
      lo = v2f{val(0), val(1)};
      hi = v2f{val(2), val(3)};
    } else {
      lo = V(r.lo);
      hi = V(r.hi);
    }
    const v2f mid = v2f{lo.y, hi.x};
    const v2f w0 = sw_[k3][0], w1 = sw_[k3][1], w2 = sw_[k3][2], bb = sb_[k3];  // splats
    return vfma(w0, lo, vfma(w1, mid, vfma(w2, hi, bb)));
  };
  constexpr int NIT = M / T;  // 16 points per thread
  constexpr int UB = 4;       // iterations whose loads are in flight together
  constexpr int PSTEP = T + T / 16;  // pad16(tid + j T) = pad16(tid) + j PSTEP (T is a multiple of 16)
  float2* const s_me = s + pad16(tid);

  // ---- gated input, packed: c[m] = z[2m] + i z[2m+1] ----
#pragma unroll 1
  for (int j0 = 0; j0 < NIT; j0 += UB) {
    if (2 * (j0 * T) >= L) {  // block-uniform: this batch lies in the zero padding (at least half of the M points do)
#pragma unroll
      for (int u = 0; u < UB; ++u) s_me[(j0 + u) * PSTEP] = make_float2(0.f, 0.f);
      continue;
    }
    Raw r1[UB], rv[UB];
    float2 zsub[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int t = 2 * (tid + (j0 + u) * T);
      r1[u] = raw_load(ux1, t);
      if (!TAB) rv[u] = raw_load(uv, t);
      if (DL) zsub[u] = *reinterpret_cast<const float2*>(zcrow + (t < t_last ? t : t_last));
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int t = 2 * (tid + (j0 + u) * T);
      v2f z = sc_eval(1, r1[u]) * sc_eval(2, TAB ? r1[u] : rv[u]);
      if (DL) z = t < pv ? v2f{0.f, 0.f} : z - V(zsub[u]);  // (over the shared row's positions z - z_pad is zero by definition)
      if (t >= L) z = v2f{0.f, 0.f};
      else if (t + 1 >= L) z.y = 0.f;
      s_me[(j0 + u) * PSTEP] = F(z);
    }
  }
  __syncthreads();
  fft_lds<LOGM>(s, tw, tid);

  // ---- spectrum product and repacking for the inverse transform (stored conjugated: IFFT = conj . FFT . conj) ----
  const float2* K = spec + (int64_t)c * (M + 1);
  constexpr int TWS = FFT_TW_N / (2 * M);
  auto product = [&](int kk, float2 kf_, float2 kb_, float2 wk_) {  // kf = K[kk], kb = K[M - kk]
    const v2f ck = V(s[pad16(kk)]), cmk = V(s[pad16((M - kk) & (M - 1))]);
    const v2f kf = V(kf_), kb = V(kb_), wk = V(wk_);
    if (kk == 0) {
      const float y0 = (ck.x + ck.y) * kf.x;   // Y[0]  (real)
      const float ym = (ck.x - ck.y) * kb.x;   // Y[M]  (real)
      // Ye[0] = (Y[0] + Y[M]) / 2, Yo[0] = (Y[0] - Y[M]) / 2; C'[0] = Ye + i Yo, stored conjugated
      s[pad16(0)] = make_float2(0.5f * (y0 + ym), -0.5f * (y0 - ym));
      return;
    }
    // unpack: Z[k] = E + W^k O, Z[M-k] = conj(E - W^k O), E = (C[k] + conj(C[M-k])) / 2, O = -i (C[k] - conj(C[M-k])) / 2
    const v2f half = 0.5f;
    const v2f bc = vconj(cmk);
    const v2f e = half * (ck + bc), d = half * (ck - bc);
    const v2f wo = vcmul(wk, v2f{d.y, -d.x});
    const v2f zk = e + wo, zmk = vconj(e - wo);
    const v2f yk = vcmul(zk, kf), ymk = vcmul(zmk, kb);
    // repack: Ye[k] = (Y[k] + conj(Y[M-k])) / 2,  Yo[k] = conj(W^k) (Y[k] - conj(Y[M-k])) / 2
    const v2f bq = vconj(ymk);
    const v2f ye = half * (yk + bq), d2 = half * (yk - bq);
    const v2f yo = vcmul(vconj(wk), d2);
    // C'[k] = Ye + i Yo ;  C'[M-k] = conj(Ye) + i conj(Yo); both stored conjugated
    const v2f iyo = v2f{-yo.y, yo.x};
    const v2f cp = ye + iyo;
    const v2f cq = vconj(ye) + v2f{yo.y, yo.x};
    s[pad16(kk)] = F(vconj(cp));
    if (kk != M / 2) s[pad16(M - kk)] = F(vconj(cq));
  };
  // pairs (kk, M - kk) for kk = tid + j T, j < NIT / 2: touched by exactly one thread each, in place
#pragma unroll 1
  for (int j0 = 0; j0 < NIT / 2; j0 += UB) {
    float2 kf[UB], kb[UB], wk[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int kk = tid + (j0 + u) * T;
      kf[u] = K[kk];
      kb[u] = K[M - kk];
      wk[u] = tw[kk * TWS];
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) product(tid + (j0 + u) * T, kf[u], kb[u], wk[u]);
  }
  if (tid == 0) product(M / 2, K[M / 2], K[M / 2], tw[(M / 2) * TWS]);  // the self-paired middle bin
  __syncthreads();
  fft_lds<LOGM>(s, tw, tid);

  // ---- y[t] = conv[t] * shortconv(x0)[t]: conv[2m] = Re r[m] / M, conv[2m+1] = -Im r[m] / M ----
  const float inv_m = 1.0f / (float)M;
  float* y = a.y + (int64_t)c * a.Tp + toff;
#pragma unroll 1
  for (int j0 = 0; j0 < NIT; j0 += UB) {
    if (2 * (j0 * T) >= L) break;  // block-uniform: nothing of the read is left in this batch
    Raw r0[UB];
    float2 fadd[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int t = 2 * (tid + (j0 + u) * T);
      r0[u] = raw_load(ux0, t);
      if (DL) fadd[u] = *reinterpret_cast<const float2*>(fcrow + (t < t_last ? t : t_last));
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int t = 2 * (tid + (j0 + u) * T);
      const v2f rv = V(s_me[(j0 + u) * PSTEP]);
      v2f cv = rv * v2f{inv_m, -inv_m};
      if (DL) cv += V(fadd[u]);  // + the shared row's own convolution (the read's z was taken relative to it)
      const v2f o = cv * sc_eval(0, r0[u]);
      if (t < L && t >= pv) {  // (SP: own positions only)
        if (t + 1 < L) {
          *reinterpret_cast<float2*>(y + t) = F(o);
        } else {
          y[t] = o.x;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Reads of 16 385 .. 32 769 tokens (the model's context is 32 768 bases + [SEP], deepchopper/models/llm/components.py:
// 8-14): M = 2^15 complex points do not fit the LDS, so the transform is split once, M = 16 x 2048 (four-step FFT):
//   forward   c[m], m = n2 + 2048 n1:   a_k1[n2] = sum_n1 c[n2 + 2048 n1] W16^(n1 k1)      (<= 9 non-zero terms: the
//             b_k1[n2] = a_k1[n2] W_M^(n2 k1);   C[k1 + 16 k2] = FFT_2048(b_k1)[k2]          input fills half of M)
//   product   pairs k <-> M - k of the packed real transform: rows k1 and 16 - k1 (rows 0 and 8 pair with themselves)
//   inverse   R_k1[b] = FFT_2048(conj Y'[k1 + 16 k2])[b] W_M^(k1 b)   ->   scratch[k1][b]    (one HBM round trip)
//             r[b + 2048 a] = sum_k1 R_k1[b] W16^(k1 a)                                     (second kernel, + output gate)
// hyena_fft15_kernel: one workgroup per (read, channel, row pair): two 2048-point transforms side by side in LDS (the
// in-LDS radix-16 code above, 128 threads each), everything between the input rows and the scratch rows in LDS.  The
// gated input is recomputed by the 8 workgroups of a (read, channel) (L2 hits, ~40 instructions per sample pair).
// The filter spectrum keeps taps 0 .. M-1; a read of exactly M + 1 = 32 769 tokens is missing the single product
// k'[M] z[0] in y[M], which the second kernel adds.
// ---------------------------------------------------------------------------------------------------
constexpr int F15_LOGM = 15, F15_M = 1 << F15_LOGM, F15_N2 = 2048, F15_ROW = F15_N2 + F15_N2 / 16;  // padded row (float2)

// short causal filter + gating helpers of one (read, channel) (the same arithmetic as hyena_fft_kernel's lambdas)
template <bool TAB, int SP>
struct GateSrc {
  static constexpr bool DL = SP == 2;
  const float *zcrow, *fcrow;  // DL: the shared row's z / convolution at the read's positions
  const float *ux0, *ux1, *uv;
  const unsigned char* tok;
  const float* s_tab;
  int L, t_last;
  int toff, pv, padoff;  // SP: positions < pv (a multiple of 4) live in the shared [PAD] row at slot padoff + t
  __device__ __forceinline__ int slot(int t) const { return (SP && t < pv ? padoff : toff) + t; }
  float sw_[3][3], sb_[3];
  struct Raw {
    float2 lo, hi;
  };
  __device__ __forceinline__ void init(const HyenaArgs& a, int r, int c, const float* s_tab_) {
    toff = a.toff[r];
    pv = SP ? a.pv[r] : 0;
    padoff = SP ? a.padoff[r] : 0;
    zcrow = DL ? a.zc + (int64_t)c * a.dstride + a.dbase[r] : nullptr;
    fcrow = DL ? a.Fc + (int64_t)c * a.dstride + a.dbase[r] : nullptr;
    L = a.ntok[r];
    t_last = L > 0 ? (L - 1) & ~1 : 0;
    ux0 = a.u + (int64_t)c * a.Tp;
    ux1 = a.u + (int64_t)(a.D + c) * a.Tp;
    uv = a.u + (int64_t)(2 * a.D + c) * a.Tp;
    tok = TAB ? a.tok : nullptr;
    s_tab = s_tab_;
#pragma unroll
    for (int k3 = 0; k3 < 3; ++k3) {
#pragma unroll
      for (int j = 0; j < 3; ++j) sw_[k3][j] = a.sw[(k3 * a.D + c) * 3 + j];
      sb_[k3] = a.sb[k3 * a.D + c];
    }
  }
  __device__ __forceinline__ Raw raw_load(const float* row, int t) const {
    Raw r;
    const int tc = t < t_last ? t : t_last;
    const int tl = tc >= 2 ? tc - 2 : 0;
    const int sc = slot(tc), sl = slot(tl);
    if (TAB) {
      const uint32_t hi16 = *reinterpret_cast<const unsigned short*>(tok + sc);
      const uint32_t lo16 = *reinterpret_cast<const unsigned short*>(tok + sl);
      uint32_t ids = (t >= 2 ? lo16 : 0xFFFFu) | (hi16 << 16);
      if (t + 1 >= L) ids |= 0xFF000000u;
      if (t >= L) ids = 0xFFFFFFFFu;
      r.lo = make_float2(__builtin_bit_cast(float, ids), 0.f);
      r.hi = make_float2(0.f, 0.f);
      return r;
    }
    const float2 lo = *reinterpret_cast<const float2*>(row + sl);
    const float2 hi = *reinterpret_cast<const float2*>(row + sc);
    const bool in = t < L;
    r.lo = make_float2(in && t >= 2 ? lo.x : 0.f, in && t >= 2 ? lo.y : 0.f);
    r.hi = make_float2(in ? hi.x : 0.f, in && t + 1 < L ? hi.y : 0.f);
    return r;
  }
  __device__ __forceinline__ v2f sc_eval(int k3, const Raw& r) const {  // the filter's outputs at t, t + 1
    v2f lo, hi;
    if (TAB) {
      const uint32_t ids = __builtin_bit_cast(uint32_t, r.lo.x);
      auto val = [&](int byte) -> float {
        const uint32_t id = (ids >> (8 * byte)) & 255u;
        return s_tab[k3 * 17 + (id < 16u ? id : 16u)];
      };
      lo = v2f{val(0), val(1)};
      hi = v2f{val(2), val(3)};
    } else {
      lo = V(r.lo);
      hi = V(r.hi);
    }
    const v2f mid = v2f{lo.y, hi.x};
    const v2f w0 = sw_[k3][0], w1 = sw_[k3][1], w2 = sw_[k3][2], bb = sb_[k3];
    return vfma(w0, lo, vfma(w1, mid, vfma(w2, hi, bb)));
  }
  // gated input pair z[t], z[t+1] (t even), zero beyond the read
  __device__ __forceinline__ v2f z_pair(int t) const {
    const Raw r1 = raw_load(ux1, t);
    const Raw rv = TAB ? r1 : raw_load(uv, t);
    v2f z = sc_eval(1, r1) * sc_eval(2, rv);
    if (DL) z = t < pv ? v2f{0.f, 0.f} : z - V(*reinterpret_cast<const float2*>(zcrow + (t < t_last ? t : t_last)));
    if (t >= L) z = v2f{0.f, 0.f};
    else if (t + 1 >= L) z.y = 0.f;
    return z;
  }
};

// ---------------------------------------------------------------------------------------------------
// Split transforms (round 5): M = Q x N2 complex points, N2 = 2^LOGN, Q = 2, 3, 4 (and Q = 1: the same kernel around one
// row) -- one outer step of the four-step factorisation around the in-LDS radix-16 code, all of it in LDS (Q rows of N2
// points, one group of N2 / 16 threads each):
//   forward   m = n2 + N2 n1:   b_k1[n2] = W_M^(n2 k1) sum_n1 c[n2 + N2 n1] W_Q^(n1 k1);   C[k1 + Q k2] = FFT_N2(b_k1)[k2]
//   product   pairs k <-> M - k of the packed real transform; C[k] lives in row k mod Q at index k div Q
//   inverse   R_k1 = FFT_N2(row k1);   r[b + N2 a] = sum_k1 R_k1[b] W_M^(k1 b) W_Q^(k1 a)
// What it buys over one 2^LOGM transform of the next power of two:
//   * the packed input fills at most HALF of the M points (2M real samples >= 2L), so the outer forward step has
//     ceil(Q / 2) terms instead of Q (Q = 2: none -- row 1 is row 0 times a twiddle), and only the outputs r[n], n < M / 2,
//     are wanted, so the outer inverse step is evaluated for a < ceil(Q / 2) only: the zero half of the first pass and the
//     discarded half of the last one are never computed;
//   * N2 = 4096 transforms in THREE radix-16 passes where 2^13 / 2^14 points take four;
//   * M need not be a power of two: 3 x 4096 points serve reads of 8 193 .. 12 288 tokens that paid for 16 384.
// And the kernel around the transforms is organised by memory round trips, which is what a workgroup's life consisted of
// (two workgroups per CU, 8 dependent global round trips of ~3 us each in the round-2 form): the channel's x1 | v rows are
// staged in the transform's own LDS with 16-byte loads, all in flight at once; the spectrum loads of the product phase and
// the output gate's loads are each one batch; the twiddles of the outer steps and of the product phase are one table look-up
// per thread times compile-time roots of unity.  Measured on the config-4 length mix, per launch of a non-table layer
// (profiles/r05b_rocprofv3_kernel_stats_config4_mixed.csv): 2^13 class 2.16 -> 1.58 ms, 2^14 class 1.39 -> 0.95 ms (as
// 3 x 4096), 2^12 class 0.63 -> 0.50 ms (the one-row form); FFT stage -27 %.  Size classes that were built and measured
// and did NOT pay -- 3 x 1024, 3 x 2048, 5 x 2048 points: fewer points, but 3 / 6 / 10 waves per workgroup fill a CU worse
// than 8 / 12 / 16 -- are not instantiated (profiles/r05b_fft_class_sweep.jsonl).
// W_M / W_2M come from `twq` (W_2M^n = twq[n * qstep]: the 2^16 table for Q = 1, 2, 4; the 24 576-th roots for Q = 3), the
// rows' own twiddles from the 2^16 table as before.
// ---------------------------------------------------------------------------------------------------
template <int LOGN, int Q>
struct SplitGeo {
  static constexpr int N2 = 1 << LOGN, M = Q * N2, TG = N2 / 16, T = Q * TG;
  static constexpr int NZ = (Q + 1) / 2;      // blocks of N2 points that can hold input / wanted output (points < M / 2)
  static constexpr int NJ = (16 + Q - 1) / Q;  // "one n2 per thread" iterations that cover a row (T = Q N2 / 16 threads)
  // Padded row + an offset that spreads the Q rows over the LDS banks for the product phase (lane -> row k mod Q)
  static constexpr int ROW = N2 + N2 / 16 + (Q == 2 ? 16 : Q == 3 ? 11 : Q == 4 ? 8 : 0);
  static constexpr size_t smem = (size_t)Q * ROW * sizeof(float2) + 64 * sizeof(float);
};

// x * exp(-2 pi i e / Q); e is a compile-time constant wherever this is called (fully unrolled loops)
template <int Q>
__device__ __forceinline__ v2f mul_wq(v2f x, int e) {
  e %= Q;
  if (e == 0) return x;
  if (Q % 2 == 0 && e == Q / 2) return -x;
  if (Q % 4 == 0 && e == Q / 4) return v2f{x.y, -x.x};       // * (-i)
  if (Q % 4 == 0 && e == 3 * Q / 4) return v2f{-x.y, x.x};   // * (+i)
  float c = 1.f, s = 0.f;
  if (Q == 3) {
    c = -0.5f;
    s = e == 1 ? -0.86602540378443865f : 0.86602540378443865f;
  }
  return vcmul(x, v2f{c, s});
}

// LDS slot of bin k of the split layout
template <int LOGN, int Q>
__device__ __forceinline__ int split_at(int k) {
  const int q = k / Q;
  return (k - q * Q) * SplitGeo<LOGN, Q>::ROW + pad16(q);
}

// Outer forward step for one n2: z[n1] = c[n2 + N2 n1] (n1 < NZ) -> the Q row entries b_k1[n2]; wf[k1] = W_M^(n2 k1), k1 >= 1
template <int LOGN, int Q>
__device__ __forceinline__ void split_forward_store(float2* __restrict__ col, const v2f (&z)[SplitGeo<LOGN, Q>::NZ],
                                                    const float2 (&wf)[Q]) {  // col = row 0's slot of n2
  using G = SplitGeo<LOGN, Q>;
#pragma unroll
  for (int k1 = 0; k1 < Q; ++k1) {
    v2f acc = z[0];
#pragma unroll
    for (int n1 = 1; n1 < G::NZ; ++n1) acc += mul_wq<Q>(z[n1], n1 * k1);
    col[k1 * G::ROW] = F(k1 == 0 ? acc : vcmul(acc, V(wf[k1])));
  }
}

template <int LOGN, int Q>
__global__ __launch_bounds__(Q * (1 << LOGN) / 16) void hyena_fftq_filter_kernel(const float* __restrict__ kf, int64_t kf_stride,
                                                                                  int kf_taps, const float2* __restrict__ tw,
                                                                                  const float2* __restrict__ twq, int qstep,
                                                                                  float2* __restrict__ spec) {
  using G = SplitGeo<LOGN, Q>;
  constexpr int N2 = G::N2, M = G::M, T = G::T, TG = G::TG, NZ = G::NZ, NJ = G::NJ;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_rawq[];
  float2* s = reinterpret_cast<float2*>(s_rawq);
  const int c = blockIdx.x, tid = threadIdx.x;
  const float* k = kf + (int64_t)c * kf_stride;
  const int taps = kf_taps < M ? kf_taps : M;
#pragma unroll 1
  for (int j = 0; j < NJ; ++j) {
    const int n2 = tid + j * T;
    if (n2 >= N2) break;
    v2f z[NZ];
    float2 wf[Q];
#pragma unroll
    for (int n1 = 0; n1 < NZ; ++n1) {
      const int t = 2 * (n2 + N2 * n1);
      z[n1] = v2f{t < taps ? k[t] : 0.f, t + 1 < taps ? k[t + 1] : 0.f};
    }
#pragma unroll
    for (int k1 = 1; k1 < Q; ++k1) wf[k1] = twq[2 * n2 * k1 * qstep];
    split_forward_store<LOGN, Q>(s + pad16(n2), z, wf);
  }
  __syncthreads();
  const int grp = tid / TG;
  fft_lds<LOGN>(s + grp * G::ROW, tw, tid - grp * TG);
  float2* out = spec + (int64_t)c * (M + 1);
  for (int kk = tid; kk <= M / 2; kk += T) {
    const float2 ck = s[split_at<LOGN, Q>(kk)], cmk = s[split_at<LOGN, Q>(kk ? M - kk : 0)];
    float2 zk, zmk;
    unpack_pair<0>(ck, cmk, twq[kk * qstep], zk, zmk);
    if (kk == 0) {
      out[0] = make_float2(ck.x + ck.y, 0.f);
      out[M] = make_float2(ck.x - ck.y, 0.f);
    } else {
      out[kk] = zk;
      out[M - kk] = zmk;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Pair maps (round 6).  Between the two transforms the packed-real convolution does, per conjugate pair (k, M - k):
// unpack (C[k], C[M-k]) -> (Z[k], Z[M-k]), multiply by (K[k], K[M-k]), repack -> (C'[k], C'[M-k]).  All three steps are linear
// in a = C[k], b = conj(C[M-k]), so the whole thing is ONE 2 x 2 complex map per pair.  With w = W_2M^k = c + i s,
// P = K[k], Qc = conj(K[M-k]):
//       u = C'[k]        = A a + B b          A  = ((P + Qc) + s (P - Qc)) / 2
//       v = conj(C'[M-k]) = -B a + A' b       A' = ((P + Qc) - s (P - Qc)) / 2        B = i c (P - Qc) / 2
// (alpha = (1 - i w) / 2, beta = (1 + i w) / 2: Z[k] = alpha a + beta b, conj Z[M-k] = beta a + alpha b, u = conj(alpha) P Z[k] +
//  conj(beta) Qc conj Z[M-k], v likewise with alpha and beta swapped; |alpha|^2 = (1 + s) / 2, |beta|^2 = (1 - s) / 2,
//  conj(alpha) beta = i c / 2.)  A, A', B are built once per filter and size class, in double, with the inverse transform's
// 1 / M folded in; the kernel's product phase is then 8 packed FMAs per pair with no twiddle, no halving and no conjugation
// left (round 5: 59 instructions per pair, 470 of a thread's ~2 000).  Bin 0 (Y[0] and Y[M] are real and pair with each
// other) keeps its own two real factors in A; the self-paired middle bin has B = 0.
// coef layout per channel: A[0 .. M/2] | A'[0 .. M/2] | B[0 .. M/2]  (float2 each).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hyena_fft_pairmap_kernel(const float2* __restrict__ spec, const float2* __restrict__ twq, int qstep,
                                                               int M, float2* __restrict__ coef) {
  const int kk = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
  const int H = M / 2 + 1;
  if (kk >= H) return;
  const float2* K = spec + (int64_t)c * (M + 1);
  float2* out = coef + (int64_t)c * 3 * H;
  const double inv_m = 1.0 / (double)M;
  if (kk == 0) {
    out[0] = make_float2((float)((double)K[0].x * inv_m), (float)((double)K[M].x * inv_m));
    out[H] = make_float2(0.f, 0.f);
    out[2 * H] = make_float2(0.f, 0.f);
    return;
  }
  const float2 p = K[kk], q = K[M - kk], w = twq[kk * qstep];
  const double px = p.x, py = p.y, qx = q.x, qy = -(double)q.y;  // Qc = conj(K[M-k])
  const double sx = 0.5 * (px + qx), sy = 0.5 * (py + qy), dx = 0.5 * (px - qx), dy = 0.5 * (py - qy);
  const double cw = w.x, sw = w.y;
  out[kk] = make_float2((float)((sx + sw * dx) * inv_m), (float)((sy + sw * dy) * inv_m));
  out[H + kk] = make_float2((float)((sx - sw * dx) * inv_m), (float)((sy - sw * dy) * inv_m));
  out[2 * H + kk] = make_float2((float)(-cw * dy * inv_m), (float)(cw * dx * inv_m));  // i c (dx + i dy)
}

// Complex multiply-accumulate on packed fp32 with every swizzle and sign as an operand modifier of v_pk_fma_f32 (op_sel picks the
// half of a source per result lane, neg_lo / neg_hi negate a source per lane): two instructions per complex product, no moves.
// (Written as vector code hipcc built the rotated operand (-x.y, x.x) with a v_xor + v_mov per product: 7 of 20 instructions
// of a pair.)   r = acc + a x:   lo = a.x x.x - a.y x.y + acc.x,   hi = a.x x.y + a.y x.x + acc.y
#define DC_PK_FMA(dst, s0, s1, s2, mods) asm("v_pk_fma_f32 %0, %1, %2, %3 " mods : "=v"(dst) : "v"(s0), "v"(s1), "v"(s2))
__device__ __forceinline__ v2f pm_mul(v2f a, v2f x) {  // a x
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(x));  // a.xx * x
  DC_PK_FMA(r, a, x, r, "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]");     // + a.yy * (-x.y, x.x)
  return r;
}
__device__ __forceinline__ v2f pm_nmul(v2f a, v2f x) {  // -(a x)
  v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(x));
  DC_PK_FMA(r, a, x, r, "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]");
  return r;
}
__device__ __forceinline__ v2f pm_fma_conj(v2f cm, v2f x, v2f acc) {  // acc + conj(cm) x
  v2f r;
  DC_PK_FMA(r, cm, x, acc, "op_sel_hi:[0,1,1]");                                  // + cm.xx * x
  DC_PK_FMA(r, cm, x, r, "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]");      // - cm.yy * (-x.y, x.x)
  return r;
}
__device__ __forceinline__ v2f pm_fma_conj_out_conj(v2f cm, v2f x, v2f acc) {  // conj(acc + conj(cm) x): the last step negates its hi lane
  v2f r;
  DC_PK_FMA(r, cm, x, acc, "op_sel_hi:[0,1,1]");
  DC_PK_FMA(r, cm, x, r, "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,0,1]");      // hi: -(-cm.y x.x + r.y) = cm.y x.x - r.y
  return r;
}

template <int LOGN, int Q, bool TAB, int SP, bool PM>
__global__ __launch_bounds__(Q * (1 << LOGN) / 16, 4) void hyena_fftq_kernel(HyenaArgs a, const int32_t* __restrict__ reads,
                                                                          const float2* __restrict__ tw,
                                                                          const float2* __restrict__ twq, int qstep,
                                                                          const float2* __restrict__ spec) {  // PM: the pair maps
  using G = SplitGeo<LOGN, Q>;
  constexpr int N2 = G::N2, M = G::M, T = G::T, TG = G::TG, NZ = G::NZ, ROW = G::ROW;
  // "one n2 per thread" iterations.  Q == 1 (one row, no outer step): input and wanted output fill the first half of the row
  constexpr int NJ = Q == 1 ? 8 : G::NJ;
  constexpr bool DL = SP == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_rawq[];
  float2* s = reinterpret_cast<float2*>(s_rawq);
  float* s_tab = reinterpret_cast<float*>(s + Q * ROW);
  const int r = reads[blockIdx.x], c = blockIdx.y, tid = threadIdx.x;
  if (TAB) {
    if (tid < 3 * 17) {
      const int k3 = tid / 17, id = tid - 17 * k3;
      s_tab[tid] = id < a.vocab && id < 16 ? a.tab[(int64_t)id * 3 * a.D + k3 * a.D + c] : 0.f;
    }
  }
  using Gate = GateSrc<TAB, SP>;
  using Raw = typename Gate::Raw;
  Gate g;
  g.init(a, r, c, s_tab);
  const int L = g.L, t_last = g.t_last;

  // W_M^(tid k1): the outer steps' twiddle of n2 = tid + j T is this times W_16^(j k1) (T = M / 16), a compile-time constant
  float2 wb[Q];
#pragma unroll
  for (int k1 = 1; k1 < Q; ++k1) wb[k1] = twq[2 * tid * k1 * qstep];

  // ---- stage the channel's x1 | v rows (TAB: the token ids) in the transform's LDS: 16-byte loads, all of them in flight at
  //      once -- one memory round trip for the whole gated input (the per-iteration loads of the first version were 3 of 8
  //      dependent round trips of a workgroup's life)
  float* const sf = reinterpret_cast<float*>(s);          // x1[t] at sf[t], v[t] at sf[M + t]
  unsigned char* const sb = reinterpret_cast<unsigned char*>(s);  // TAB: id of position t at sb[t]
  constexpr int NS = (M / 4 + T - 1) / T;  // groups of 4 positions per thread
  static_assert(NS == 4, "staging is written out for four groups per thread");
  float2 zsub[NJ][NZ];
  {
    // (four named groups, not an array: hipcc kept an indexed array of these in scratch memory)
    float4 a1_0, a1_1, a1_2, a1_3, av_0, av_1, av_2, av_3;
    uint32_t aw_0 = 0, aw_1 = 0, aw_2 = 0, aw_3 = 0;
#define DC_STAGE_LOAD(i)                                                  \
  {                                                                       \
    const int t4 = 4 * (tid + i * T);                                     \
    const int sl = g.slot(t4 < L ? t4 : 0);                               \
    if (TAB) {                                                            \
      aw_##i = *reinterpret_cast<const uint32_t*>(g.tok + sl);            \
    } else {                                                              \
      a1_##i = *reinterpret_cast<const float4*>(g.ux1 + sl);              \
      av_##i = *reinterpret_cast<const float4*>(g.uv + sl);               \
    }                                                                     \
  }
    DC_STAGE_LOAD(0) DC_STAGE_LOAD(1) DC_STAGE_LOAD(2) DC_STAGE_LOAD(3)
#undef DC_STAGE_LOAD
    if (DL) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int n1 = 0; n1 < NZ; ++n1) {
          const int t = 2 * (tid + j * T + N2 * n1);
          zsub[j][n1] = *reinterpret_cast<const float2*>(g.zcrow + (t < t_last ? t : t_last));
        }
      }
    }
#define DC_STAGE_STORE(i)                                                 \
  {                                                                       \
    const int t4 = 4 * (tid + i * T);                                     \
    if (t4 < L) {                                                         \
      if (TAB) {                                                          \
        *reinterpret_cast<uint32_t*>(sb + t4) = aw_##i;                   \
      } else {                                                            \
        *reinterpret_cast<float4*>(sf + t4) = a1_##i;                     \
        *reinterpret_cast<float4*>(sf + M + t4) = av_##i;                 \
      }                                                                   \
    }                                                                     \
  }
    DC_STAGE_STORE(0) DC_STAGE_STORE(1) DC_STAGE_STORE(2) DC_STAGE_STORE(3)
#undef DC_STAGE_STORE
  }
  __syncthreads();

  // ---- gated input pairs of this thread, from the staged rows ----
  v2f z[NJ][NZ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n2 = tid + j * T;
#pragma unroll
    for (int n1 = 0; n1 < NZ; ++n1) {
      const int t = 2 * (n2 + N2 * n1);
      v2f zz = v2f{0.f, 0.f};
      if (t < L && (Q == 1 || (16 % Q) == 0 || n2 < N2)) {
        Raw r1, rv;
        if (TAB) {
          const uint32_t lo16 = t >= 2 ? *reinterpret_cast<const unsigned short*>(sb + t - 2) : 0xFFFFu;
          uint32_t ids = lo16 | ((uint32_t)*reinterpret_cast<const unsigned short*>(sb + t) << 16);
          if (t + 1 >= L) ids |= 0xFF000000u;
          r1.lo = make_float2(__builtin_bit_cast(float, ids), 0.f);
          r1.hi = make_float2(0.f, 0.f);
          rv = r1;
        } else {
          const int tl = t >= 2 ? t - 2 : 0;
          r1.lo = *reinterpret_cast<const float2*>(sf + tl);
          r1.hi = *reinterpret_cast<const float2*>(sf + t);
          rv.lo = *reinterpret_cast<const float2*>(sf + M + tl);
          rv.hi = *reinterpret_cast<const float2*>(sf + M + t);
          if (t < 2) r1.lo = rv.lo = make_float2(0.f, 0.f);
        }
        zz = g.sc_eval(1, r1) * g.sc_eval(2, rv);
        if (DL) zz = t < g.pv ? v2f{0.f, 0.f} : zz - V(zsub[j][n1]);
        if (t + 1 >= L) zz.y = 0.f;
      }
      z[j][n1] = zz;
    }
    __builtin_amdgcn_sched_barrier(0);  // (all 4 NJ NZ LDS reads hoisted to the front cost ~90 VGPRs)
  }
  __syncthreads();

  // ---- outer forward step -> rows ----
  // (T is a multiple of 16: the padded slot of n2 = tid + j T is that of tid plus j (T + T / 16) -- one address, constant offsets)
  constexpr int PSTEP = T + T / 16;
  float2* const s_me = s + pad16(tid);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n2 = tid + j * T;
    if (Q == 1) {
      s_me[j * PSTEP] = F(z[j][0]);
      s_me[j * PSTEP + (N2 / 2 + N2 / 32)] = make_float2(0.f, 0.f);
    } else if ((16 % Q) == 0 || n2 < N2) {
      float2 wf[Q];
#pragma unroll
      for (int k1 = 1; k1 < Q; ++k1) {
        const int e = (j * k1) & 15;
        wf[k1] = e == 0 ? wb[k1] : cmul(wb[k1], unit_root(e, 16));
      }
      split_forward_store<LOGN, Q>(s_me + j * PSTEP, z[j], wf);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  const int grp = tid / TG;
  fft_lds<LOGN>(s + grp * ROW, tw, tid - grp * TG);

  // ---- spectrum product and repacking for the inverse transform (stored conjugated: IFFT = conj . FFT . conj) ----
  // Bin kk_j = tid + j T lives in row kk mod Q = tid mod Q (T is a multiple of Q) at index kk / Q = tid / Q + j T / Q, and
  // T / Q = N2 / 16 is a multiple of 16: its padded slot is that of tid plus j (T / Q) 17 / 16, its partner's (M - kk_j) that of
  // M - tid minus the same -- two addresses per thread, constant offsets
  constexpr int KSTEP = T / Q + T / Q / 16;
  static_assert(T % Q == 0 && (T / Q) % 16 == 0, "bin slots by constant strides");
  float2* const pk0 = s + split_at<LOGN, Q>(tid);
  float2* const pmk0 = s + split_at<LOGN, Q>(M - tid);  // (thread 0: bin M, one slot behind row 0 -- never dereferenced, see product)
  if constexpr (PM) {
    // One 2 x 2 complex map per pair (hyena_fft_pairmap_kernel): u = A a + B b -> conj(u) to bin k, v = -B a + A' b to bin M - k,
    // a = C[k], b = conj(C[M-k]).  All twenty-four coefficient loads of the thread in flight at once.
    constexpr int H = M / 2 + 1;
    const float2* CA = spec + (int64_t)c * 3 * H;
    float2 ca[8], cp[8], cb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kk = tid + j * T;
      ca[j] = CA[kk];
      cp[j] = CA[H + kk];
      cb[j] = CA[2 * H + kk];
    }
    const float2 am = CA[M / 2];  // the self-paired middle bin: B = 0, u = A a
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float2* const pk = pk0 + j * KSTEP;
      float2* const pmk = pmk0 - j * KSTEP;
      const v2f a = V(*pk);
      if (j == 0 && tid == 0) {  // bin 0: Y[0] = (Re + Im) K[0], Y[M] = (Re - Im) K[M], both real; C'[0] = (Y0 + YM) / 2 + i (Y0 - YM) / 2
        const float y0 = (a.x + a.y) * ca[0].x, ym = (a.x - a.y) * ca[0].y;
        *pk = make_float2(0.5f * (y0 + ym), -0.5f * (y0 - ym));
      } else {
        const v2f cm = V(*pmk);  // b = conj(cm)
        const v2f A = V(ca[j]), A2 = V(cp[j]), B = V(cb[j]);
        *pk = F(pm_fma_conj_out_conj(cm, B, pm_mul(a, A)));   // conj(u), u = A a + B b
        *pmk = F(pm_fma_conj(cm, A2, pm_nmul(a, B)));         // v = -B a + A' b
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (tid == 0) {
      float2* const pm = s + split_at<LOGN, Q>(M / 2);
      const v2f u = pm_mul(V(*pm), V(am));
      *pm = make_float2(u.x, -u.y);
    }
  } else {
  const float2* K = spec + (int64_t)c * (M + 1);
  auto product = [&](int kk, float2* pk, float2* pmk, float2 kf_, float2 kb_, float2 wk_) {  // kf = K[kk], kb = K[M - kk], wk = W_2M^kk
    const v2f ck = V(*pk);
    const v2f kf = V(kf_), kb = V(kb_), wk = V(wk_);
    if (kk == 0) {
      const float y0 = (ck.x + ck.y) * kf.x;   // Y[0]  (real)
      const float ym = (ck.x - ck.y) * kb.x;   // Y[M]  (real)
      *pk = make_float2(0.5f * (y0 + ym), -0.5f * (y0 - ym));
      return;
    }
    const v2f cmk = V(*pmk);
    const v2f half = 0.5f;
    const v2f bc = vconj(cmk);
    const v2f e = half * (ck + bc), d = half * (ck - bc);
    const v2f wo = vcmul(wk, v2f{d.y, -d.x});
    const v2f zk = e + wo, zmk = vconj(e - wo);
    const v2f yk = vcmul(zk, kf), ymk = vcmul(zmk, kb);
    const v2f bq = vconj(ymk);
    const v2f ye = half * (yk + bq), d2 = half * (yk - bq);
    const v2f yo = vcmul(vconj(wk), d2);
    const v2f cp = ye + v2f{-yo.y, yo.x};
    const v2f cq = vconj(ye) + v2f{yo.y, yo.x};
    *pk = F(vconj(cp));
    if (2 * kk != M) *pmk = F(vconj(cq));
  };
  {
    // kk = tid + j T, j < 8 (kk < 8 T = M / 2: every pair (kk, M - kk) belongs to exactly one thread).  All sixteen spectrum
    // loads in flight at once; W_2M^kk = W_2M^tid W_32^j (T = M / 16) needs one table look-up instead of eight gathers.
    float2 kf[8], kb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kk = tid + j * T;
      kf[j] = K[kk];
      kb[j] = K[M - kk];
    }
    const float2 wk0 = twq[tid * qstep];
    const float2 km = K[M / 2], wm = twq[(M / 2) * qstep];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      constexpr float C32[9] = {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f,
                                0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f, 0.0f};
      const float2 wk = j == 0 ? wk0 : cmul(wk0, make_float2(C32[j], -C32[8 - j]));  // exp(-2 pi i j / 32)
      product(tid + j * T, pk0 + j * KSTEP, pmk0 - j * KSTEP, kf[j], kb[j], wk);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (tid == 0) product(M / 2, s + split_at<LOGN, Q>(M / 2), s + split_at<LOGN, Q>(M / 2), km, km, wm);  // the self-paired middle bin
  }
  }
  __syncthreads();

  fft_lds<LOGN>(s + grp * ROW, tw, tid - grp * TG);
  // (opaque copies: left alone, hipcc keeps the NJ (Q - 1) products wb W_16^(j k1) of the forward step alive -- spilled --
  //  across both transforms to reuse them below; recomputing them is a dozen packed multiplies)
#pragma unroll
  for (int k1 = 1; k1 < Q; ++k1) asm volatile("" : "+v"(wb[k1].x), "+v"(wb[k1].y));

  // the output gate's inputs: every load of the output phase in flight at once -- one round trip
  Raw r0[NJ][NZ];
  float2 fadd[NJ][NZ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
#pragma unroll
    for (int aa = 0; aa < NZ; ++aa) {
      const int t = 2 * (tid + j * T + N2 * aa);
      r0[j][aa] = g.raw_load(g.ux0, t);
      if (DL) fadd[j][aa] = *reinterpret_cast<const float2*>(g.fcrow + (t < t_last ? t : t_last));
    }
  }

  // ---- outer inverse step for the wanted half, output gate: conv[2n] = Re r[n] / M, conv[2n+1] = -Im r[n] / M ----
  const float inv_m = PM ? 1.0f : 1.0f / (float)M;  // (PM: folded into the pair maps)
  float* y = a.y + (int64_t)c * a.Tp + g.toff;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int b = tid + j * T;
    if (Q != 1 && (16 % Q) != 0 && b >= N2) continue;
    const float2* const col = s_me + j * PSTEP;
    v2f tk[Q];
    tk[0] = V(col[0]);
#pragma unroll
    for (int k1 = 1; k1 < Q; ++k1) {
      const int e = (j * k1) & 15;
      const float2 wf = e == 0 ? wb[k1] : cmul(wb[k1], unit_root(e, 16));
      tk[k1] = vcmul(V(col[k1 * ROW]), V(wf));
    }
#pragma unroll
    for (int aa = 0; aa < NZ; ++aa) {
      const int t = 2 * (b + N2 * aa);
      v2f rr = tk[0];
#pragma unroll
      for (int k1 = 1; k1 < Q; ++k1) rr += mul_wq<Q>(tk[k1], k1 * aa);
      v2f cv = rr * v2f{inv_m, -inv_m};
      if (DL) cv += V(fadd[j][aa]);  // + the shared row's own convolution (the read's z was taken relative to it)
      const v2f o = cv * g.sc_eval(0, r0[j][aa]);
      if (t < L && t >= g.pv) {  // (SP: own positions only)
        if (t + 1 < L) *reinterpret_cast<float2*>(y + t) = F(o);
        else y[t] = o.x;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// rows of workgroup p (0..7): (0, 8) for p = 0, (p, 16 - p) otherwise
__device__ __forceinline__ void f15_rows(int p, int& ka, int& kb) {
  ka = p;
  kb = p == 0 ? 8 : 16 - p;
}

// forward half shared by the operator and the filter-spectrum kernel: fills the two rows from `sample(m)` (complex point
// m of the packed input, zero beyond n_points) and transforms them.  s0 / s1: padded rows; tid in [0, 256).
template <class Sample>
__device__ __forceinline__ void f15_forward_rows(float2* s0, float2* s1, int ka, int kb, int n_points, const float2* __restrict__ tw,
                                                 int tid, Sample sample) {
  const int n1_used = (n_points + F15_N2 - 1) / F15_N2;  // blocks of 2048 points that hold input (<= 9)
  float2 w16a[9];                                        // W16^(n1 ka); row kb uses the conjugate (kb = 16 - ka) or (-1)^n1 (kb = 8)
#pragma unroll
  for (int n1 = 0; n1 < 9; ++n1) w16a[n1] = tw[((n1 * ka) & 15) * (FFT_TW_N / 16)];
#pragma unroll 1
  for (int j = 0; j < F15_N2 / 256; ++j) {
    const int n2 = tid + 256 * j;
    v2f acc_a = v2f{0.f, 0.f}, acc_b = v2f{0.f, 0.f};
#pragma unroll
    for (int n1 = 0; n1 < 9; ++n1) {
      if (n1 < n1_used) {  // block-uniform
        const v2f cpt = sample(n2 + F15_N2 * n1);
        const v2f wa = V(w16a[n1]);
        acc_a += vcmul(cpt, wa);
        acc_b += ka == 0 ? ((n1 & 1) ? -cpt : cpt) : vcmul(cpt, vconj(wa));
      }
    }
    // b_k1[n2] = a_k1[n2] W_M^(n2 k1)
    const v2f ta = V(tw[((n2 * ka) & (F15_M - 1)) * (FFT_TW_N / F15_M)]);
    const v2f tb = V(tw[((n2 * kb) & (F15_M - 1)) * (FFT_TW_N / F15_M)]);
    s0[pad16(n2)] = F(vcmul(acc_a, ta));
    s1[pad16(n2)] = F(vcmul(acc_b, tb));
  }
  __syncthreads();
  fft_lds<11>(tid < 128 ? s0 : s1, tw, tid & 127);
}

template <bool TAB, int SP>
__global__ __launch_bounds__(256) void hyena_fft15_kernel(HyenaArgs a, const int32_t* __restrict__ reads, const float2* __restrict__ tw,
                                                          const float2* __restrict__ spec, float2* __restrict__ scratch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw15[];
  float2* s0 = reinterpret_cast<float2*>(s_raw15);
  float2* s1 = s0 + F15_ROW;
  float* s_tab = reinterpret_cast<float*>(s1 + F15_ROW);
  const int slot = blockIdx.x >> 3, p = blockIdx.x & 7;
  const int r = reads[slot], c = blockIdx.y, tid = threadIdx.x;
  if (TAB) {
    if (tid < 3 * 17) {
      const int k3 = tid / 17, id = tid - 17 * k3;
      s_tab[tid] = id < a.vocab && id < 16 ? a.tab[(int64_t)id * 3 * a.D + k3 * a.D + c] : 0.f;
    }
    __syncthreads();
  }
  GateSrc<TAB, SP> g;
  g.init(a, r, c, s_tab);
  int ka, kb;
  f15_rows(p, ka, kb);
  const int n_points = (g.L + 1) >> 1;  // complex points that hold input
  f15_forward_rows(s0, s1, ka, kb, n_points < F15_M ? n_points : F15_M, tw, tid, [&](int m) -> v2f { return g.z_pair(2 * m); });

  // ---- spectrum product + repacking, pair (k, M - k); results stored conjugated (IFFT = conj . FFT . conj) ----
  const float2* K = spec + (int64_t)c * (F15_M + 1);
  constexpr int TW2 = FFT_TW_N / (2 * F15_M);  // W_2M^k = tw[k * TW2]
  auto product = [&](float2* pk, float2* pmk, int k) {  // *pk = C[k], *pmk = C[M - k], 0 < k < M / 2 or k == M / 2 (pk == pmk)
    const v2f ck = V(*pk), cmk = V(*pmk);
    const v2f kf = V(K[k]), kbk = V(K[F15_M - k]), wk = V(tw[k * TW2]);
    const v2f half = 0.5f;
    const v2f bc = vconj(cmk);
    const v2f e = half * (ck + bc), d = half * (ck - bc);
    const v2f wo = vcmul(wk, v2f{d.y, -d.x});
    const v2f zk = e + wo, zmk = vconj(e - wo);
    const v2f yk = vcmul(zk, kf), ymk = vcmul(zmk, kbk);
    const v2f bq = vconj(ymk);
    const v2f ye = half * (yk + bq), d2 = half * (yk - bq);
    const v2f yo = vcmul(vconj(wk), d2);
    const v2f cp = ye + v2f{-yo.y, yo.x};
    const v2f cq = vconj(ye) + v2f{yo.y, yo.x};
    *pk = F(vconj(cp));
    if (pk != pmk) *pmk = F(vconj(cq));
  };
  if (p != 0) {  // row ka index k2  <->  row kb index 2047 - k2
#pragma unroll 1
    for (int j = 0; j < F15_N2 / 256; ++j) {
      const int k2 = tid + 256 * j;
      product(s0 + pad16(k2), s1 + pad16(F15_N2 - 1 - k2), ka + 16 * k2);
    }
  } else {
    // row 0: k = 16 k2 <-> row 0 index 2048 - k2 (k2 = 1 .. 1023), the real pair k = 0 / M, the self-paired k = M / 2
    // row 8: k = 8 + 16 k2 <-> row 8 index 2047 - k2 (k2 = 0 .. 1023)
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
      const int k2 = tid + 256 * j;  // 0 .. 1023
      product(s1 + pad16(k2), s1 + pad16(F15_N2 - 1 - k2), 8 + 16 * k2);
      if (k2 >= 1) product(s0 + pad16(k2), s0 + pad16(F15_N2 - k2), 16 * k2);
    }
    if (tid == 0) {
      const float2 c0 = s0[pad16(0)];
      const float y0 = (c0.x + c0.y) * K[0].x;       // Y[0]  (real)
      const float ym = (c0.x - c0.y) * K[F15_M].x;   // Y[M]  (real)
      s0[pad16(0)] = make_float2(0.5f * (y0 + ym), -0.5f * (y0 - ym));
      product(s0 + pad16(F15_N2 / 2), s0 + pad16(F15_N2 / 2), F15_M / 2);
    }
  }
  __syncthreads();
  fft_lds<11>(tid < 128 ? s0 : s1, tw, tid & 127);

  // ---- R_k1[b] W_M^(k1 b) -> scratch[slot][c][k1][b] ----
  float2* out = scratch + ((int64_t)slot * a.D + c) * F15_M;
#pragma unroll 1
  for (int j = 0; j < F15_N2 / 256; ++j) {
    const int b = tid + 256 * j;
    const v2f ta = V(tw[((b * ka) & (F15_M - 1)) * (FFT_TW_N / F15_M)]);
    const v2f tb = V(tw[((b * kb) & (F15_M - 1)) * (FFT_TW_N / F15_M)]);
    out[ka * F15_N2 + b] = F(vcmul(V(s0[pad16(b)]), ta));
    out[kb * F15_N2 + b] = F(vcmul(V(s1[pad16(b)]), tb));
  }
}

// second kernel: r[b + 2048 a] = sum_k1 scratch[k1][b] W16^(k1 a); y[t] = conv[t] * shortconv(x0)[t]
template <bool TAB, int SP>
__global__ __launch_bounds__(256) void hyena_fft15_out_kernel(HyenaArgs a, const int32_t* __restrict__ reads,
                                                              const float2* __restrict__ scratch) {
  __shared__ float s_tab[64];
  const int slot = blockIdx.x >> 3, bblk = blockIdx.x & 7;
  const int r = reads[slot], c = blockIdx.y, tid = threadIdx.x;
  if (TAB) {
    if (tid < 3 * 17) {
      const int k3 = tid / 17, id = tid - 17 * k3;
      s_tab[tid] = id < a.vocab && id < 16 ? a.tab[(int64_t)id * 3 * a.D + k3 * a.D + c] : 0.f;
    }
    __syncthreads();
  }
  GateSrc<TAB, SP> g;
  g.init(a, r, c, s_tab);
  const int b = bblk * 256 + tid;
  const float2* in = scratch + ((int64_t)slot * a.D + c) * F15_M + b;
  float2 u[16];
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) u[k1] = in[k1 * F15_N2];
  dft_regs<16>(u);  // frequency a at position bitrev(a)
  const float inv_m = 1.0f / (float)F15_M;
  float* y = a.y + (int64_t)c * a.Tp + a.toff[r];
  const int L = g.L;
#pragma unroll
  for (int aa = 0; aa < 9; ++aa) {  // t = 2 (b + 2048 aa) < 32 770
    const int t = 2 * (b + F15_N2 * aa);
    if (t < L && t >= g.pv) {  // (SP: own positions only)
      const v2f rv = V(u[bitrev(aa, 4)]);
      v2f conv = rv * v2f{inv_m, -inv_m};
      if (t == F15_M) {  // L = M + 1: the tap the M-tap spectrum leaves out, y[M] += k'[M] z[0]
        const v2f z0 = g.z_pair(0);
        conv.x = fmaf(a.kf[(int64_t)c * a.kf_stride + F15_M], z0.x, conv.x);
      }
      if (SP == 2) conv += V(*reinterpret_cast<const float2*>(g.fcrow + t));  // + the shared row's own convolution
      const v2f o = conv * g.sc_eval(0, g.raw_load(g.ux0, t));
      if (t + 1 < L) *reinterpret_cast<float2*>(y + t) = F(o);
      else y[t] = o.x;
    }
  }
}

// filter spectrum for the 2^15-point form: K[c][0 .. M] = rfft(kf[c][0 .. M), n = 2M)
__global__ __launch_bounds__(256) void hyena_fft15_filter_kernel(const float* __restrict__ kf, int64_t kf_stride, int kf_taps,
                                                                 const float2* __restrict__ tw, float2* __restrict__ spec) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw15[];
  float2* s0 = reinterpret_cast<float2*>(s_raw15);
  float2* s1 = s0 + F15_ROW;
  const int p = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
  const float* k = kf + (int64_t)c * kf_stride;
  int ka, kb;
  f15_rows(p, ka, kb);
  const int taps = kf_taps < F15_M ? kf_taps : F15_M;
  f15_forward_rows(s0, s1, ka, kb, (taps + 1) >> 1, tw, tid, [&](int m) -> v2f {
    const int t = 2 * m;
    return v2f{t < taps ? k[t] : 0.f, t + 1 < taps ? k[t + 1] : 0.f};
  });
  float2* out = spec + (int64_t)c * (F15_M + 1);
  constexpr int TW2 = FFT_TW_N / (2 * F15_M);
  auto unpack = [&](const float2* pk, const float2* pmk, int kk) {
    float2 zk, zmk;
    unpack_pair<F15_LOGM>(*pk, *pmk, tw[kk * TW2], zk, zmk);
    out[kk] = zk;
    out[F15_M - kk] = zmk;
  };
  if (p != 0) {
    for (int j = 0; j < F15_N2 / 256; ++j) {
      const int k2 = tid + 256 * j;
      unpack(s0 + pad16(k2), s1 + pad16(F15_N2 - 1 - k2), ka + 16 * k2);
    }
  } else {
    for (int j = 0; j < 4; ++j) {
      const int k2 = tid + 256 * j;
      unpack(s1 + pad16(k2), s1 + pad16(F15_N2 - 1 - k2), 8 + 16 * k2);
      if (k2 >= 1) unpack(s0 + pad16(k2), s0 + pad16(F15_N2 - k2), 16 * k2);
    }
    if (tid == 0) {
      const float2 c0 = s0[pad16(0)];
      out[0] = make_float2(c0.x + c0.y, 0.f);
      out[F15_M] = make_float2(c0.x - c0.y, 0.f);
      unpack(s0 + pad16(F15_N2 / 2), s0 + pad16(F15_N2 / 2), F15_M / 2);
    }
  }
}

constexpr size_t f15_smem() { return (size_t)2 * F15_ROW * sizeof(float2) + 64 * sizeof(float); }

template <int LOGM>
constexpr size_t fft_smem() {
  return (size_t)((1 << LOGM) + (1 << LOGM) / 16) * sizeof(float2) + 64 * sizeof(float);
}

template <int LOGM>
int launch_filter_t(hipStream_t st, const float* kf, int64_t kf_stride, int kf_taps, int D, const float2* tw, float2* spec) {
  constexpr int T = (1 << LOGM) / 16;
  DC_SET_MAX_SMEM_ONCE_PER_DEVICE((&hyena_fft_filter_kernel<LOGM>), fft_smem<LOGM>());
  hipLaunchKernelGGL(hyena_fft_filter_kernel<LOGM>, dim3(D), dim3(T), fft_smem<LOGM>(), st, kf, kf_stride, kf_taps, tw, spec);
  DC_LAUNCH_CHECK();
  return DC_OK;
}

template <int LOGM, bool TAB, int SP>
int launch_fft_t2(hipStream_t st, const HyenaArgs& a, const int32_t* reads, int n_reads, const float2* tw, const float2* spec) {
  constexpr int T = (1 << LOGM) / 16;
  DC_SET_MAX_SMEM_ONCE_PER_DEVICE((&hyena_fft_kernel<LOGM, TAB, SP>), fft_smem<LOGM>());
  hipLaunchKernelGGL((hyena_fft_kernel<LOGM, TAB, SP>), dim3(n_reads, a.D), dim3(T), fft_smem<LOGM>(), st, a, reads, tw, spec);
  DC_LAUNCH_CHECK();
  return DC_OK;
}
template <int LOGM, bool TAB>
int launch_fft_t(hipStream_t st, const HyenaArgs& a, const int32_t* reads, int n_reads, const float2* tw, const float2* spec) {
  if (a.pv && a.dbase) return launch_fft_t2<LOGM, TAB, 2>(st, a, reads, n_reads, tw, spec);
  return a.pv ? launch_fft_t2<LOGM, TAB, 1>(st, a, reads, n_reads, tw, spec) : launch_fft_t2<LOGM, TAB, 0>(st, a, reads, n_reads, tw, spec);
}


template <int Q>
constexpr int split_tw_off() { return Q == 3 ? FFT_TW3_OFF : 0; }
template <int LOGN, int Q>
constexpr int split_tw_step() { return (Q == 3 ? FFT_TW3_N : FFT_TW_N) / (2 * SplitGeo<LOGN, Q>::M); }

template <int LOGN, int Q>
int launch_filter_q(hipStream_t st, const float* kf, int64_t kf_stride, int kf_taps, int D, const float2* tw, float2* spec) {
  using G = SplitGeo<LOGN, Q>;
  DC_SET_MAX_SMEM_ONCE_PER_DEVICE((&hyena_fftq_filter_kernel<LOGN, Q>), G::smem);
  hipLaunchKernelGGL((hyena_fftq_filter_kernel<LOGN, Q>), dim3(D), dim3(G::T), G::smem, st, kf, kf_stride, kf_taps, tw,
                     tw + split_tw_off<Q>(), split_tw_step<LOGN, Q>(), spec);
  DC_LAUNCH_CHECK();
  return DC_OK;
}

// the pair maps stand behind the spectra in a class's buffer (hyena_fft_spec_floats); DC_FFT_PAIRMAP=0 keeps round 5's product (A/B)
bool fft_pairmap() {  // (read per launch: a test flips it between two forwards of one process)
  const char* e = getenv("DC_FFT_PAIRMAP");
  return !(e && atoi(e) == 0);
}

template <int LOGN, int Q, bool TAB, int SP>
int launch_fftq_t2(hipStream_t st, const HyenaArgs& a, const int32_t* reads, int n_reads, const float2* tw, const float2* spec) {
  using G = SplitGeo<LOGN, Q>;
  if (fft_pairmap()) {
    DC_SET_MAX_SMEM_ONCE_PER_DEVICE((&hyena_fftq_kernel<LOGN, Q, TAB, SP, true>), G::smem);
    hipLaunchKernelGGL((hyena_fftq_kernel<LOGN, Q, TAB, SP, true>), dim3(n_reads, a.D), dim3(G::T), G::smem, st, a, reads, tw,
                       tw + split_tw_off<Q>(), split_tw_step<LOGN, Q>(), spec + (int64_t)a.D * (G::M + 1));
  } else {
    DC_SET_MAX_SMEM_ONCE_PER_DEVICE((&hyena_fftq_kernel<LOGN, Q, TAB, SP, false>), G::smem);
    hipLaunchKernelGGL((hyena_fftq_kernel<LOGN, Q, TAB, SP, false>), dim3(n_reads, a.D), dim3(G::T), G::smem, st, a, reads, tw,
                       tw + split_tw_off<Q>(), split_tw_step<LOGN, Q>(), spec);
  }
  DC_LAUNCH_CHECK();
  return DC_OK;
}
template <int LOGN, int Q, bool TAB>
int launch_fftq_t(hipStream_t st, const HyenaArgs& a, const int32_t* reads, int n_reads, const float2* tw, const float2* spec) {
  if (a.pv && a.dbase) return launch_fftq_t2<LOGN, Q, TAB, 2>(st, a, reads, n_reads, tw, spec);
  return a.pv ? launch_fftq_t2<LOGN, Q, TAB, 1>(st, a, reads, n_reads, tw, spec)
              : launch_fftq_t2<LOGN, Q, TAB, 0>(st, a, reads, n_reads, tw, spec);
}
template <int LOGN, int Q>
int launch_fftq(hipStream_t st, const HyenaArgs& a, const int32_t* reads, int n_reads, const float2* tw, const float2* spec) {
  return a.tok ? launch_fftq_t<LOGN, Q, true>(st, a, reads, n_reads, tw, spec) : launch_fftq_t<LOGN, Q, false>(st, a, reads, n_reads, tw, spec);
}
template <int LOGM>
int launch_fft_direct(hipStream_t st, const HyenaArgs& a, const int32_t* reads, int n_reads, const float2* tw, const float2* spec) {
  return a.tok ? launch_fft_t<LOGM, true>(st, a, reads, n_reads, tw, spec) : launch_fft_t<LOGM, false>(st, a, reads, n_reads, tw, spec);
}

}  // namespace

int launch_fft_twiddles(hipStream_t st, float* tw) {
  float2* t2 = reinterpret_cast<float2*>(tw);
  hipLaunchKernelGGL(fft_twiddle_kernel, dim3(FFT_TW_N / 256), dim3(256), 0, st, t2, FFT_TW_N);
  hipLaunchKernelGGL(fft_twiddle_kernel, dim3(FFT_TW3_N / 256), dim3(256), 0, st, t2 + FFT_TW3_OFF, FFT_TW3_N);
  DC_LAUNCH_CHECK();
  return DC_OK;
}
int fft_twiddle_count() { return FFT_TW_TOTAL; }

// classes the staged / split kernel serves (hyena_fftq_kernel): their buffers carry the pair maps behind the spectra
static bool fftq_class(const FftClassInfo& ci) { return ci.q >= 2 || (ci.q == 1 && ci.points == 4096); }

int64_t hyena_fft_spec_floats(int cls, int D) {
  const FftClassInfo& ci = kFftClass[cls];
  int64_t n = (int64_t)D * ((int64_t)ci.points + 1) * 2;
  if (fftq_class(ci)) n += (int64_t)D * 3 * ((int64_t)ci.points / 2 + 1) * 2;
  return n;
}

static int launch_hyena_fft_filter_spectra(hipStream_t st, int cls, const float* kf, int64_t kf_stride, int kf_taps, int D,
                                           const float* tw, float* spec);

int launch_hyena_fft_filter(hipStream_t st, int cls, const float* kf, int64_t kf_stride, int kf_taps, int D,
                            const float* tw, float* spec) {
  DC_REQUIRE(cls >= 0 && cls < kFftClasses, "FFT long convolution: no such size class");
  int rc = launch_hyena_fft_filter_spectra(st, cls, kf, kf_stride, kf_taps, D, tw, spec);
  const FftClassInfo& ci = kFftClass[cls];
  if (rc || !fftq_class(ci)) return rc;
  const int M = ci.points;
  const float2* t2 = reinterpret_cast<const float2*>(tw);
  float2* s2 = reinterpret_cast<float2*>(spec);
  const bool q3 = ci.q == 3;  // W_2M^k: the 24 576-th roots for 3 x 4096 points, the 2^16 table otherwise
  hipLaunchKernelGGL(hyena_fft_pairmap_kernel, dim3((M / 2 + 1 + 255) / 256, D), dim3(256), 0, st, (const float2*)s2,
                     t2 + (q3 ? FFT_TW3_OFF : 0), (q3 ? FFT_TW3_N : FFT_TW_N) / (2 * M), M, s2 + (int64_t)D * (M + 1));
  DC_LAUNCH_CHECK();
  return DC_OK;
}

static int launch_hyena_fft_filter_spectra(hipStream_t st, int cls, const float* kf, int64_t kf_stride, int kf_taps, int D,
                                           const float* tw, float* spec) {
  const FftClassInfo& ci = kFftClass[cls];
  const float2* t2 = reinterpret_cast<const float2*>(tw);
  float2* s2 = reinterpret_cast<float2*>(spec);
  switch (ci.q * 100000 + ci.points) {
    case 100000 + 1024: return launch_filter_t<10>(st, kf, kf_stride, kf_taps, D, t2, s2);
    case 100000 + 2048: return launch_filter_t<11>(st, kf, kf_stride, kf_taps, D, t2, s2);
    case 100000 + 4096: return launch_filter_t<12>(st, kf, kf_stride, kf_taps, D, t2, s2);
    case 100000 + 8192: return launch_filter_t<13>(st, kf, kf_stride, kf_taps, D, t2, s2);
    case 100000 + 16384: return launch_filter_t<14>(st, kf, kf_stride, kf_taps, D, t2, s2);
    case 200000 + 8192: return launch_filter_q<12, 2>(st, kf, kf_stride, kf_taps, D, t2, s2);
    case 300000 + 12288: return launch_filter_q<12, 3>(st, kf, kf_stride, kf_taps, D, t2, s2);
    case 400000 + 16384: return launch_filter_q<12, 4>(st, kf, kf_stride, kf_taps, D, t2, s2);
    case 32768:
      hipLaunchKernelGGL(hyena_fft15_filter_kernel, dim3(8, D), dim3(256), f15_smem(), st, kf, kf_stride, kf_taps, t2, s2);
      DC_LAUNCH_CHECK();
      return DC_OK;
  }
  set_error("FFT long convolution: unsupported size class %d", cls);
  return DC_ERR_UNSUPPORTED;
}

int64_t hyena_fft15_scratch_bytes(int n_reads, int D) { return (int64_t)n_reads * D * F15_M * (int64_t)sizeof(float2); }

int launch_hyena_fft15(hipStream_t st, const HyenaArgs& a, const int32_t* reads, int n_reads, const float* tw, const float* spec,
                       float* scratch) {
  if (n_reads <= 0) return DC_OK;
  DC_REQUIRE(a.ntok && a.toff && a.u && a.y && a.sw && a.sb && a.kf && reads && tw && spec && scratch,
             "FFT long convolution (2^15): null argument");
  // (tap 32 768 of a.kf is read only for a read of 32 769 tokens, whose own length made the host materialise it)
  const float2* t2 = reinterpret_cast<const float2*>(tw);
  const float2* s2 = reinterpret_cast<const float2*>(spec);
  float2* sc = reinterpret_cast<float2*>(scratch);
  const dim3 grid((unsigned)(8 * n_reads), (unsigned)a.D), block(256);
#define DC_F15_LAUNCH(TAB, SP)                                                                                          \
  do {                                                                                                                  \
    hipLaunchKernelGGL((hyena_fft15_kernel<TAB, SP>), grid, block, f15_smem(), st, a, reads, t2, s2, sc);               \
    DC_LAUNCH_CHECK();                                                                                                  \
    hipLaunchKernelGGL((hyena_fft15_out_kernel<TAB, SP>), grid, block, 0, st, a, reads, (const float2*)sc);             \
  } while (0)
  const int sp = a.pv == nullptr ? 0 : (a.dbase == nullptr ? 1 : 2);
  if (a.tok) {
    if (sp == 2) DC_F15_LAUNCH(true, 2);
    else if (sp == 1) DC_F15_LAUNCH(true, 1);
    else DC_F15_LAUNCH(true, 0);
  } else {
    if (sp == 2) DC_F15_LAUNCH(false, 2);
    else if (sp == 1) DC_F15_LAUNCH(false, 1);
    else DC_F15_LAUNCH(false, 0);
  }
#undef DC_F15_LAUNCH
  DC_LAUNCH_CHECK();
  return DC_OK;
}

int launch_hyena_fft(hipStream_t st, const HyenaArgs& a, int cls, const int32_t* reads, int n_reads, const float* tw,
                     const float* spec) {
  if (n_reads <= 0) return DC_OK;
  DC_REQUIRE(a.ntok && a.toff && a.u && a.y && a.sw && a.sb && reads && tw && spec, "FFT long convolution: null argument");
  DC_REQUIRE(cls >= 0 && cls < kFftClasses, "FFT long convolution: no such size class");
  const FftClassInfo& ci = kFftClass[cls];
  const float2* t2 = reinterpret_cast<const float2*>(tw);
  const float2* s2 = reinterpret_cast<const float2*>(spec);
  // 4 096 points: the staged kernel's one-row form (Q = 1) measured 10 % faster than the round-2 kernel on this class
  // (profiles/r05b_fft_class_sweep.jsonl); DC_FFT_STAGED1=0 keeps the latter (A/B)
  static const bool staged1 = !(getenv("DC_FFT_STAGED1") && atoi(getenv("DC_FFT_STAGED1")) == 0);
  if (staged1 && ci.q == 1 && ci.points == 4096) return launch_fftq<12, 1>(st, a, reads, n_reads, t2, s2);
  switch (ci.q * 100000 + ci.points) {
    case 100000 + 1024: return launch_fft_direct<10>(st, a, reads, n_reads, t2, s2);
    case 100000 + 2048: return launch_fft_direct<11>(st, a, reads, n_reads, t2, s2);
    case 100000 + 4096: return launch_fft_direct<12>(st, a, reads, n_reads, t2, s2);
    case 100000 + 8192: return launch_fft_direct<13>(st, a, reads, n_reads, t2, s2);
    case 100000 + 16384: return launch_fft_direct<14>(st, a, reads, n_reads, t2, s2);
    case 200000 + 8192: return launch_fftq<12, 2>(st, a, reads, n_reads, t2, s2);
    case 300000 + 12288: return launch_fftq<12, 3>(st, a, reads, n_reads, t2, s2);
    case 400000 + 16384: return launch_fftq<12, 4>(st, a, reads, n_reads, t2, s2);
  }
  set_error("FFT long convolution: size class %d has no in-LDS kernel", cls);
  return DC_ERR_UNSUPPORTED;
}

}  // namespace dc
