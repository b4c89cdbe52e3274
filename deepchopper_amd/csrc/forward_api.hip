// Model handle, weight repacking and forward orchestration behind the C ABI (F1-F3).
#include <algorithm>
#include <cmath>
#include <map>
#include <numeric>
#include <vector>

#include "dc_common.h"
#include "forward.h"

using namespace dc;

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct LayerDev {
  float *Wt_in = nullptr, *b_in = nullptr;    // [D][3D], [3D]   (norm1 folded)
  float *sw = nullptr, *sb = nullptr;         // [3D][3], [3D]
  float *sw_unit = nullptr, *sb_unit = nullptr;  // the same with x0's taps zeroed and its bias 1: a unit output gate (linear shared prefix)
  float *Wt_out = nullptr, *b_out = nullptr;  // [D][D], [D]
  float *Wt_fc1 = nullptr, *b_fc1 = nullptr;  // [D][DI], [DI]  (norm2 folded)
  float *Wt_fc2 = nullptr, *b_fc2 = nullptr;  // [DI][D], [D]
  unsigned short *Wp_in = nullptr, *Wp_out = nullptr, *Wp_fc1 = nullptr, *Wp_fc2 = nullptr;  // bf16 planes
  unsigned short *Wq_in = nullptr, *Wq_out = nullptr, *Wq_fc1 = nullptr, *Wq_fc2 = nullptr;  // fp16 planes (scaled)
  unsigned short* Wq_fc2p = nullptr;  // fc2 planes with the k order of the fused MLP kernel (mlp_fused_perm16)
  unsigned short* Wq_inp = nullptr;   // in_proj planes in that k order (B operand = the previous block's MLP output)
  unsigned short *Wp_fc2p = nullptr, *Wp_inp = nullptr;  // the same two in the bf16x6 planes (strict mode's fused MLP kernel)
  float is_in = 1.f, is_out = 1.f, is_fc1 = 1.f, is_fc2 = 1.f;                              // 1 / weight scale
  // implicit filter parameters (device copies, checkpoint layout)
  float *pos_z = nullptr, *pos_t = nullptr, *fw0 = nullptr, *fb0 = nullptr, *fw1 = nullptr, *fb1 = nullptr,
        *fw2 = nullptr, *fb2 = nullptr, *fw3 = nullptr, *freq0 = nullptr, *freq1 = nullptr, *freq2 = nullptr,
        *deltas = nullptr, *fbias = nullptr;
  float* kf = nullptr;  // [D][kf_taps] materialised long filter
  unsigned short* kfp = nullptr;  // bf16 planes of the reversed filter, 8 shifted copies (hyena3)
  unsigned short* kfq = nullptr;  // fp16 planes (2) of the per-channel scaled filter
  float *kscale = nullptr, *kinv = nullptr;  // [D] power-of-two filter scale and its inverse
  float* kspec[kFftClasses] = {};  // FFT long convolution: filter spectra [D][points + 1] float2 per size class (forward.h: kFftClass)
};

// linear shared prefix: positions of the zero stretch in front of the shared row's in the zc / Fc rows (>= the longest sequence
// of a read without a pad run + the 32-token block its kernels may read past its end)
constexpr int kPrefixZero = 32768 + 128;

struct ProfEvent {
  hipEvent_t a, b;
  int tag;
};

}  // namespace

struct dc_model {
  dc_model_config cfg{};
  int device = 0;
  float* emb = nullptr;
  std::vector<LayerDev> layers;
  float *Wt_l1 = nullptr, *b_l1 = nullptr, *Wt_l2 = nullptr, *b_l2 = nullptr, *W3 = nullptr, *b3 = nullptr;
  unsigned short *Wp_l1 = nullptr, *Wp_l2 = nullptr;
  unsigned short *Wq_l1 = nullptr, *Wq_l2 = nullptr;
  unsigned short* Wq_l1p = nullptr;  // head linear1 planes in the fused MLP kernel's k order
  unsigned short* Wp_l1p = nullptr;  // ... as bf16x6 planes
  int mlp_chain = 1;                 // fused MLP kernel also runs the GEMM that consumes LayerNorm(its output)
  float is_l1 = 1.f, is_l2 = 1.f;
  int* d_overflow = nullptr;  // fp16x3 range guard (device counter)
  int* ovf_target = nullptr;  // caller-owned counter that replaces it (dc_model_set_overflow_target)
  float* u_table0 = nullptr;  // [vocab][3D] layer-0 in_proj(LayerNorm(emb[v])) (fp64 on the host)
  int mlp_fusion = 1;  // fp16x3: fc1 -> GELU -> fc2 in one kernel (dc_model_set_mlp_fusion)
  int gemm_mode = 2;  // 0 = fp32 MFMA, 1 = bf16x6 (error-free 3-way bf16 split, fp32-equivalent), 2 = fp16x3 (2-way fp16 split)
  int64_t kf_taps = 0;
  float* fft_tw = nullptr;          // twiddle table of the FFT long convolution
  bool kspec_valid[kFftClasses] = {};  // spectra match the materialised filters
  uint32_t fft_class_mask = 0;      // size classes in use (bit per kFftClass entry; dc_model_set_fft_split / DC_FFT_CLASS_MASK)
  int fft_min_tokens = 2560;        // reads with more tokens take the FFT long convolution (dc_model_set_fft_threshold)
  // FFT buckets of the cached geometry: read lists by transform size
  int32_t* d_fft_reads = nullptr;
  int fft_count[kFftClasses] = {}, fft_offset[kFftClasses] = {};  // per size class; the last = the 16 x 2048 four-step form
  DevBuf fft_scratch;               // 2^15-point form: one HBM round trip between its two kernels
  int max_P_mfma = 0;
  int n_groups_long = 0;            // leading groups with reads above 1 024 tokens (16-block chunks); the rest fit one 32-block chunk
  std::vector<void*> owned;  // every weight allocation
  int64_t weight_bytes = 0;

  // workspace (grown on demand)
  DevBuf x, u, y, hmid, obuf, mean, rstd, qpad, idx, part, tok;
  int64_t Tp_cap = 0;

  // index arrays: host mirror of the last geometry + pinned staging
  std::vector<int32_t> h_geom;  // key of the cached geometry
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  hipEvent_t staging_done = nullptr;
  bool staging_pending = false;
  // device views into idx.  "Units" = the caller's reads, then the shared [PAD] rows of a dc_forward_shared_prefix launch.
  // Slot map (embed / head scatter): unit u owns slots [d_toff[u], + ceil4(d_ntok[u])), its tokens come from d_src_base[u]
  // (< 0: a shared [PAD] row -- constant tokens in, logits out to padlog + (-1 - d_src_base[u])).  Long convolution: unit u
  // is a sequence of d_cntok[u] positions (d_nblk[u] 32-token blocks); positions < d_pv[u] live at slot d_padoff[u] + t, the
  // others at d_ctoff[u] + t (d_ctoff = d_toff - d_pv).  Without shared prefixes the two views coincide.
  int32_t *d_blk_read = nullptr, *d_toff = nullptr, *d_ntok = nullptr, *d_nblk = nullptr, *d_grp_reads = nullptr,
          *d_grp_P = nullptr;
  int32_t *d_ctoff = nullptr, *d_cntok = nullptr, *d_pv = nullptr, *d_padoff = nullptr;
  int64_t* d_src_base = nullptr;
  int64_t* d_fill_dst = nullptr;    // [n_reads] dense position of a row's first pad column
  int32_t *d_fill_n = nullptr, *d_fill_src = nullptr;  // [n_reads] pad columns of the row; position 0 of its shared row in padlog
  bool shared_prefix = false;       // the cached geometry has shared [PAD] rows
  int prefix_mode = 1;              // dc_forward_shared_prefix: 1 = exact form (a row per long-convolution class, reads walk their pad runs), 2 = linear form
  bool linear_prefix = false;       // ... and they are the linear form's ONE row
  int32_t* d_dbase = nullptr;       // [n_units] linear form: index of a read's position 0 in the zc / Fc rows
  DevBuf zc, Fc;                    // linear form: [D][dcap] gated input / long convolution of the shared row behind a zero stretch
  int64_t dcap = 0;                 // their row stride
  int lin_pad_cls = -1, lin_pad_len = 0, lin_pad_P = 0, lin_pad_toff = 0, lin_pad_fft_index = 0;  // the shared row: class, tokens, blocks, slot, entry in d_fft_reads
  int n_fill_rows = 0, pad_id = 0;
  DevBuf padlog;                    // float2 per position of the shared [PAD] rows: their logits
  int* d_prefix_bad = nullptr;      // tokens inside a declared pad prefix that were not ([PAD], quality 0)
  int64_t Tp = 0;
  int n_groups = 0, max_P = 0, min_P = 0;

  // profiling
  bool profiling = false;
  std::vector<ProfEvent> prof_events;
  std::vector<hipEvent_t> prof_pool;
  std::map<std::string, std::pair<double, int64_t>> prof_acc;
  std::vector<std::string> prof_tags;
};

namespace {

int dev_alloc(dc_model* m, size_t bytes, void** out) {
  DC_HIP_CHECK(hipMalloc(out, bytes ? bytes : 4));
  m->owned.push_back(*out);
  m->weight_bytes += (int64_t)bytes;
  return DC_OK;
}

int upload(dc_model* m, const std::vector<float>& h, float** out) {
  void* p = nullptr;
  int rc = dev_alloc(m, h.size() * sizeof(float), &p);
  if (rc) return rc;
  DC_HIP_CHECK(hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  *out = (float*)p;
  return DC_OK;
}

// bf16 round-to-nearest-even of an fp32 value (finite inputs)
inline unsigned short bf16_rne(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
inline float bf16_to_f32(unsigned short b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// Error-free 3-way split of Wt[K][N] into planes [3][K/8][N][8] (h | m | l), uploaded as bf16.
int upload_planes(dc_model* m, const std::vector<float>& Wt, int K, int N, unsigned short** out) {
  std::vector<unsigned short> pl((size_t)3 * K * N);
  const size_t plane = (size_t)K * N;
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < N; ++n) {
      const float x = Wt[(size_t)k * N + n];
      const unsigned short hh = bf16_rne(x);
      const float r1 = x - bf16_to_f32(hh);
      const unsigned short mm = bf16_rne(r1);
      const float r2 = r1 - bf16_to_f32(mm);
      const unsigned short ll = bf16_rne(r2);
      const size_t idx = ((size_t)(k >> 3) * N + n) * 8 + (k & 7);
      pl[idx] = hh;
      pl[plane + idx] = mm;
      pl[2 * plane + idx] = ll;
    }
  void* p = nullptr;
  int rc = dev_alloc(m, pl.size() * sizeof(unsigned short), &p);
  if (rc) return rc;
  DC_HIP_CHECK(hipMemcpy(p, pl.data(), pl.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  *out = (unsigned short*)p;
  return DC_OK;
}

// fp16 conversion on the host (round to nearest even, subnormals kept)
inline unsigned short f32_to_f16_rne(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7FFFFFFFu;
  if (x >= 0x47800000u) return (unsigned short)(sign | (x > 0x7F800000u ? 0x7E00u : 0x7C00u));
  if (x < 0x38800000u) {  // below 2^-14: half subnormal, value = r * 2^-24
    float fa;
    memcpy(&fa, &x, 4);
    const uint32_t r = (uint32_t)lrintf(fa * 16777216.0f);
    return (unsigned short)(sign | r);
  }
  const uint32_t mant = x & 0x7FFFFFu, e = (x >> 23) - 127 + 15;
  uint32_t hbits = (e << 10) | (mant >> 13);
  const uint32_t rem = mant & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (hbits & 1u))) ++hbits;
  return (unsigned short)(sign | hbits);
}
inline float f16_to_f32(unsigned short hb) {
  const uint32_t e = (hb >> 10) & 31u, mant = hb & 0x3FFu;
  float v;
  if (e == 0) v = (float)mant * 5.9604644775390625e-8f;  // 2^-24
  else if (e == 31) v = mant ? NAN : INFINITY;
  else v = std::ldexp((float)(mant | 0x400u), (int)e - 25);
  return (hb & 0x8000u) ? -v : v;
}

// fp16x3 planes: Wt * scale = h + l with h = fp16(Wt * scale), l = fp16(Wt * scale - h); scale is the power of
// two that puts max|Wt| in [1024, 2048).  Layout [2][K/8][N][8].
int upload_planes_f16(dc_model* m, const std::vector<float>& Wt, int K, int N, unsigned short** out,
                      float* inv_scale) {
  float mx = 0.f;
  for (float v : Wt) mx = std::max(mx, std::fabs(v));
  int e = 0;
  if (mx > 0.f) {
    (void)std::frexp(mx, &e);  // mx = f * 2^e, f in [0.5, 1)
    e = 11 - e;                // mx * 2^e in [1024, 2048)
  }
  const float scale = std::ldexp(1.0f, e);
  *inv_scale = std::ldexp(1.0f, -e);
  std::vector<unsigned short> pl((size_t)2 * K * N);
  const size_t plane = (size_t)K * N;
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < N; ++n) {
      const float x = Wt[(size_t)k * N + n] * scale;  // exact (power of two)
      const unsigned short hh = f32_to_f16_rne(x);
      const unsigned short ll = f32_to_f16_rne(x - f16_to_f32(hh));
      const size_t idx = ((size_t)(k >> 3) * N + n) * 8 + (k & 7);
      pl[idx] = hh;
      pl[plane + idx] = ll;
    }
  void* p = nullptr;
  int rc = dev_alloc(m, pl.size() * sizeof(unsigned short), &p);
  if (rc) return rc;
  DC_HIP_CHECK(hipMemcpy(p, pl.data(), pl.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  *out = (unsigned short*)p;
  return DC_OK;
}

int upload_raw(dc_model* m, const float* h, size_t n, float** out) {
  std::vector<float> v(h, h + n);
  return upload(m, v, out);
}

// Wt[k][n] = W[n][k] * gamma[k];  b'[n] = b[n] + sum_k W[n][k] * beta[k]   (gamma/beta may be null)
void fold_transpose(const float* W, const float* b, const float* gamma, const float* beta, int N, int K,
                    std::vector<float>& Wt, std::vector<float>& bo) {
  Wt.resize((size_t)K * N);
  bo.resize(N);
  for (int n = 0; n < N; ++n) {
    double acc = b ? (double)b[n] : 0.0;
    const float* wr = W + (size_t)n * K;
    for (int k = 0; k < K; ++k) {
      const float g = gamma ? gamma[k] : 1.0f;
      Wt[(size_t)k * N + n] = wr[k] * g;
      if (beta) acc += (double)wr[k] * (double)beta[k];
    }
    bo[n] = (float)acc;
  }
}

int ensure(DevBuf& b, size_t bytes, bool zero_fill = true) {
  if (b.bytes >= bytes) return DC_OK;
  if (b.p) DC_HIP_CHECK(hipFree(b.p));
  b.p = nullptr;
  b.bytes = 0;
  DC_HIP_CHECK(hipMalloc(&b.p, bytes));
  if (!zero_fill) {  // a buffer that is written in full before it is read (the 2^15-point FFT's scratch)
    b.bytes = bytes;
    return DC_OK;
  }
  // Token slots that belong to no read (the tail up to the 256-token tile) are computed but never meaningful;
  // the long-convolution kernel does not write them.  Starting from zeros keeps them finite for good, so they
  // can neither trip the fp16 range guard nor feed NaNs into the LayerNorm statistics of their tile.
  DC_HIP_CHECK(hipMemset(b.p, 0, bytes));
  // hipMemset runs on the NULL stream, which a non-blocking stream (torch side streams, any caller's own stream) does not
  // wait for: without this the zero-fill could land in the middle of the first forward that uses the buffer (found by the
  // lanes test of round 3: the first batch on every side stream came out wrong)
  DC_HIP_CHECK(hipStreamSynchronize(nullptr));
  b.bytes = bytes;
  return DC_OK;
}

int ensure_workspace(dc_model* m, int64_t Tp) {
  if (Tp <= m->Tp_cap) return DC_OK;
  const auto& c = m->cfg;
  const size_t f = sizeof(float);
  int rc;
  if ((rc = ensure(m->x, (size_t)c.d_model * Tp * f))) return rc;
  if ((rc = ensure(m->u, (size_t)3 * c.d_model * Tp * f))) return rc;
  if ((rc = ensure(m->y, (size_t)c.d_model * Tp * f))) return rc;
  const size_t hm = (size_t)std::max(c.d_inner, c.head_hidden);
  if ((rc = ensure(m->hmid, hm * Tp * f))) return rc;
  if ((rc = ensure(m->obuf, (size_t)c.head_hidden * Tp * f))) return rc;
  if ((rc = ensure(m->mean, (size_t)Tp * f))) return rc;
  if ((rc = ensure(m->rstd, (size_t)Tp * f))) return rc;
  if ((rc = ensure(m->qpad, (size_t)Tp * f))) return rc;
  if ((rc = ensure(m->tok, (size_t)Tp))) return rc;  // packed token ids for the layer-0 table gather
  if ((rc = ensure(m->part, (size_t)(c.head_hidden / 128) * Tp * 2 * f))) return rc;
  m->Tp_cap = Tp;
  return DC_OK;
}

int ensure_filters(dc_model* m, hipStream_t st, int64_t taps_needed) {
  if (taps_needed <= m->kf_taps) return DC_OK;
  int64_t taps = ((taps_needed + 1023) / 1024) * 1024;
  const auto& c = m->cfg;
  for (auto& L : m->layers) {
    if (L.kf) {
      DC_HIP_CHECK(hipStreamSynchronize(st));
      DC_HIP_CHECK(hipFree(L.kf));
      L.kf = nullptr;
      if (L.kfp) DC_HIP_CHECK(hipFree(L.kfp));
      if (L.kfq) DC_HIP_CHECK(hipFree(L.kfq));
      L.kfp = L.kfq = nullptr;
    }
    DC_HIP_CHECK(hipMalloc((void**)&L.kf, (size_t)c.d_model * taps * sizeof(float)));
    DC_HIP_CHECK(hipMalloc((void**)&L.kfp, (size_t)c.d_model * 3 * 8 * hyena3_units((int)taps) * 16));
    DC_HIP_CHECK(hipMalloc((void**)&L.kfq, (size_t)c.d_model * 2 * 8 * hyena3_units((int)taps) * 16));
    const bool need_scale = !L.kscale;
    if (need_scale) {
      DC_HIP_CHECK(hipMalloc((void**)&L.kscale, (size_t)c.d_model * sizeof(float)));
      DC_HIP_CHECK(hipMalloc((void**)&L.kinv, (size_t)c.d_model * sizeof(float)));
    }
    FilterArgs f{};
    f.pos_z = L.pos_z;
    f.pos_t = L.pos_t;
    f.w0 = L.fw0; f.b0 = L.fb0; f.w1 = L.fw1; f.b1 = L.fb1; f.w2 = L.fw2; f.b2 = L.fb2; f.w3 = L.fw3;
    f.freq0 = L.freq0; f.freq1 = L.freq1; f.freq2 = L.freq2;
    f.deltas = L.deltas;
    f.bias = L.fbias;
    f.fo = c.filter_order;
    f.emb = c.emb_dim;
    f.D = c.d_model;
    f.max_seq_len = c.max_seq_len;
    f.n_taps = (int)taps;
    f.shift = c.filter_shift;
    f.out = L.kf;
    f.out_stride = taps;
    int rc;
    if (need_scale) {
      // the fp16x3 scale of a channel's filter is a property of the whole filter: take it once from all max_seq_len taps
      // (a temporary full-length materialisation), so that results do not depend on the lengths the handle saw before
      const int64_t full = (((int64_t)c.max_seq_len + 1023) / 1024) * 1024;
      float* tmp = nullptr;
      DC_HIP_CHECK(hipMalloc((void**)&tmp, (size_t)c.d_model * full * sizeof(float)));
      FilterArgs ff = f;
      ff.n_taps = (int)full;
      ff.out = tmp;
      ff.out_stride = full;
      if ((rc = launch_filter(st, ff)) || (rc = launch_filter_scale(st, tmp, c.d_model, (int)full, L.kscale, L.kinv))) {
        (void)hipFree(tmp);
        return rc;
      }
      DC_HIP_CHECK(hipStreamSynchronize(st));
      DC_HIP_CHECK(hipFree(tmp));
    }
    if ((rc = launch_filter(st, f))) return rc;
    if ((rc = launch_filter_planes(st, L.kf, c.d_model, (int)taps, L.kfp, L.kfq, L.kscale))) return rc;
  }
  m->kf_taps = taps;
  for (bool& v : m->kspec_valid) v = false;
  return DC_OK;
}

// Size classes of the FFT long convolution in use: the split forms (default) or the power-of-two transforms of rounds 2-4.
uint32_t fft_mask_for(bool split) {
  uint32_t mask = 0;
  for (int k = 0; k < kFftClasses; ++k) {
    const FftClassInfo& ci = kFftClass[k];
    const bool shadowed = ci.q == 1 && ci.points >= 8192;  // direct 2^13 / 2^14: served by the split forms of the same size
    if (split ? !shadowed : ci.q <= 1) mask |= 1u << k;
  }
  return mask;
}

// Filter spectra of every layer for the transforms of size class b (built once per materialised filter).
int ensure_fft_spectra(dc_model* m, hipStream_t st, int b) {
  DC_REQUIRE(b >= 0 && b < kFftClasses, "FFT size class out of range");
  if (!m->fft_tw) {
    DC_HIP_CHECK(hipMalloc((void**)&m->fft_tw, (size_t)fft_twiddle_count() * 2 * sizeof(float)));
    int rc = launch_fft_twiddles(st, m->fft_tw);
    if (rc) return rc;
  }
  if (m->kspec_valid[b]) return DC_OK;
  const size_t bytes = (size_t)hyena_fft_spec_floats(b, m->cfg.d_model) * sizeof(float);
  for (auto& L : m->layers) {
    if (!L.kspec[b]) DC_HIP_CHECK(hipMalloc((void**)&L.kspec[b], bytes));
    int rc = launch_hyena_fft_filter(st, b, L.kf, m->kf_taps, (int)m->kf_taps, m->cfg.d_model, m->fft_tw, L.kspec[b]);
    if (rc) return rc;
  }
  m->kspec_valid[b] = true;
  return DC_OK;
}

// ---- profiling helpers -------------------------------------------------------------------
int prof_tag(dc_model* m, const char* name) {
  for (size_t i = 0; i < m->prof_tags.size(); ++i)
    if (m->prof_tags[i] == name) return (int)i;
  m->prof_tags.push_back(name);
  return (int)m->prof_tags.size() - 1;
}

hipEvent_t prof_get_event(dc_model* m) {
  if (!m->prof_pool.empty()) {
    hipEvent_t e = m->prof_pool.back();
    m->prof_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

struct ProfScope {
  dc_model* m;
  hipStream_t st;
  ProfEvent ev{};
  bool on;
  ProfScope(dc_model* m_, hipStream_t st_, const char* name) : m(m_), st(st_), on(m_->profiling) {
    if (!on) return;
    ev.a = prof_get_event(m);
    ev.b = prof_get_event(m);
    ev.tag = prof_tag(m, name);
    if (!ev.a || !ev.b) {
      on = false;
      return;
    }
    (void)hipEventRecord(ev.a, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(ev.b, st);
    m->prof_events.push_back(ev);
  }
};

struct WRef {  // the three device forms of one weight matrix
  const float* Wt;
  const unsigned short* Wp;  // bf16x6 planes
  const unsigned short* Wq;  // fp16x3 planes
  float inv_scale;           // fp16x3: 1 / weight scale
};

int run_gemm(dc_model* m, hipStream_t st, const char* name, const WRef& w, const float* X, float* C,
             const float* bias, const float* R, const float* q, bool ln, int epi, int K, int N,
             bool stats_are_sums = false) {
  GemmArgs g{};
  g.Wt = w.Wt;
  g.Wp = m->gemm_mode == 1 ? w.Wp : (m->gemm_mode == 2 ? w.Wq : nullptr);
  g.split_mode = m->gemm_mode == 2 ? 1 : 0;
  g.w_inv_scale = m->gemm_mode == 2 ? w.inv_scale : 1.0f;
  g.overflow = m->ovf_target ? m->ovf_target : m->d_overflow;
  g.X = X;
  g.C = C;
  g.bias = bias;
  g.R = R;
  g.q = q;
  g.mean = (const float*)m->mean.p;
  g.rstd = (const float*)m->rstd.p;
  g.K = K;
  g.N = N;
  g.Tp = m->Tp;
  g.gelu_tanh = m->cfg.gelu_tanh;
  g.w3 = m->W3;
  g.part = (float*)m->part.p;
  g.st_sum = (float*)m->mean.p;  // the statistics buffers double as sum / sum-of-squares accumulators
  g.st_sq = (float*)m->rstd.p;
  g.stats_are_sums = stats_are_sums ? 1 : 0;
  g.ln_eps = m->cfg.ln_eps;
  ProfScope ps(m, st, name);
  return launch_gemm(st, g, epi, ln);
}

// Build (or reuse) the packed-layout index arrays for a batch geometry.
// h_npad (nullable): read r is `[PAD] x h_npad[r]` followed by its own tokens (dc_forward_shared_prefix); the pad runs are not
// laid out per read -- one all-[PAD] row per long-convolution class carries them (see dc_model: "units").
int setup_geometry(dc_model* m, hipStream_t st, int n_reads, int64_t row_stride, const int32_t* h_first,
                   const int32_t* h_ntok, const int32_t* h_npad = nullptr, int pad_id = 0) {
  // geometry key
  std::vector<int32_t> key;
  key.reserve(3 * (size_t)n_reads + 6);
  key.push_back(n_reads);
  key.push_back((int32_t)(row_stride & 0x7fffffff));
  key.push_back((int32_t)(row_stride >> 31));
  key.push_back(m->fft_min_tokens);
  key.push_back((int32_t)m->fft_class_mask);
  key.push_back(h_npad ? 1 + pad_id + 1024 * m->prefix_mode : 0);
  key.insert(key.end(), h_first, h_first + n_reads);
  key.insert(key.end(), h_ntok, h_ntok + n_reads);
  if (h_npad) key.insert(key.end(), h_npad, h_npad + n_reads);
  if (key == m->h_geom && m->Tp > 0) return DC_OK;

  // long-convolution class of a sequence of nt positions: 0 = block-Toeplitz MFMA, 1 + b = FFT size class b (forward.h:
  // kFftClass, ordered by capacity).  Long reads (more than fft_min_tokens tokens) take the FFT long convolution, bucketed by
  // transform size: up to 16 384 tokens one in-LDS transform, up to 2^15 + 1 (the model's whole context) the four-step form.
  auto conv_class = [&](int nt) -> int {
    if (nt > m->fft_min_tokens && nt <= (1 << 15) + 1) {
      for (int b = 0; b < kFftClasses; ++b)
        if ((m->fft_class_mask >> b & 1u) && nt <= kFftClass[b].points) return 1 + b;
      return 1 + kFft15Class;  // 2^15 + 1 tokens still fit the 2^15 form (one tap added by hand)
    }
    return 0;
  };

  // Shared [PAD] rows.
  // Exact form (prefix_mode 1): one row per long-convolution class that holds reads with a pad run, as long as the longest run
  // (rounded down to a multiple of 4: a read owns the last <= 3 pads in front of its first base, so that the 16-byte accesses
  // of the long-convolution kernels never straddle the two sources).  A row per CLASS because a read's pad positions must see
  // what they would have seen inside the read's own sequence: the block-Toeplitz forms are exactly causal and bit-identical
  // to one another, so one row serves them all bit for bit; an FFT form's result depends (in the last bits) on the transform.
  // The long convolution of a read still walks its whole padded sequence.
  // Linear form (prefix_mode 2): ONE row as long as the longest padded row.  The convolution is linear: with z_pad and
  // F = k * z_pad of that row known, y = (k * (z - z_pad) + F) * gate, and z - z_pad is zero over a read's whole pad run -- a
  // read's sequence starts 4 positions in front of its own slots (the short filter's halo) and its long convolution
  // costs what it costs without padding; its class follows its own length.
  const bool linear = h_npad && m->prefix_mode == 2;
  int pad_len[kFftClasses + 1] = {};
  std::vector<int32_t> pv(n_reads, 0), t0(n_reads, 0);  // pv: positions of a read's sequence the shared row holds; t0: position of the shared row the read's sequence starts at
  int n_fill = 0;
  for (int r = 0; r < n_reads; ++r) {
    DC_REQUIRE(h_ntok[r] >= 1, "every read needs at least one token");
    DC_REQUIRE(h_first[r] >= 0 && (int64_t)h_first[r] + h_ntok[r] <= row_stride, "read exceeds its row");
    if (!h_npad) continue;
    DC_REQUIRE(h_npad[r] >= 0 && h_npad[r] < h_ntok[r], "a read's pad run must be shorter than the read");
    pv[r] = h_npad[r] & ~3;
    n_fill += pv[r] > 0;
    if (linear) {
      if (pv[r] > 0) {
        t0[r] = pv[r] - 4;
        pad_len[0] = std::max(pad_len[0], (int)h_ntok[r]);  // F is needed wherever a read that leans on the row has positions
      }
    } else {
      const int k = conv_class(h_ntok[r]);
      pad_len[k] = std::max(pad_len[k], (int)pv[r]);
    }
  }
  int pad_unit[kFftClasses + 1], n_units = n_reads;
  int64_t padlog_len = 0;
  int32_t padlog_off[kFftClasses + 1] = {};
  for (int k = 0; k < kFftClasses + 1; ++k) {
    pad_unit[k] = -1;
    if (pad_len[k] > 0) {
      pad_unit[k] = n_units++;
      padlog_off[k] = (int32_t)padlog_len;
      padlog_len += pad_len[k];
    }
  }
  const bool sp = n_units > n_reads;
  // linear form: zc / Fc rows = [a zero stretch every read without a pad run points at | the shared row's positions | slack]
  constexpr int kZero = kPrefixZero;
  const int64_t dstride = linear && sp ? (int64_t)kZero + ((pad_len[0] + 3) & ~3) + 64 : 0;

  // Packed layout: a unit starts at a multiple of 4 token slots (16-byte aligned fp32 rows for the float4 / 8-byte
  // accesses of the long-convolution kernels) and owns ceil4(own tokens) slots.  (Rounds 1-2 gave every read whole 32-token
  // blocks: 1 001 tokens took 1 024 slots, 2.2 % of every GEMM's work; 513 tokens 544, 6 %.)  The long-convolution
  // kernels still walk a read in 32-token blocks: the tail of its last block overlaps the next read's slots, which they
  // may READ (causal: a later position never reaches an earlier output) but must not write -- their stores are masked
  // by the read's slot count.  32 slots of slack behind the last read keep those reads inside the buffers.
  std::vector<int32_t> nblk(n_units), toff(n_units), nown(n_units), ctoff(n_units), cntok(n_units), upv(n_units, 0), padoff(n_units, 0),
      dbase(n_units, 0);
  std::vector<int32_t> ucls(n_units);
  std::vector<int64_t> src_base(n_units);
  int64_t slots = 0;
  int maxP = 0, minP = 1 << 30;
  for (int u = 0; u < n_units; ++u) {
    if (u < n_reads) {
      upv[u] = pv[u] - t0[u];           // exact form: the pad run (floor 4); linear form: 4 (or 0: a read that owns all of its positions)
      cntok[u] = h_ntok[u] - t0[u];
      nown[u] = h_ntok[u] - pv[u];
      src_base[u] = (int64_t)u * row_stride + h_first[u] + pv[u];
      ucls[u] = conv_class(cntok[u]);
      if (linear && pv[u] > 0) dbase[u] = kZero + t0[u];
    } else {
      int k = 0;
      while (pad_unit[k] != u) ++k;
      cntok[u] = nown[u] = pad_len[k];
      src_base[u] = -1 - (int64_t)padlog_off[k];
      ucls[u] = linear ? conv_class(pad_len[k]) : k;
    }
    nblk[u] = (cntok[u] + 31) / 32;
    DC_REQUIRE(slots < ((int64_t)1 << 31) - 64, "batch too large for 32-bit token offsets");
    toff[u] = (int32_t)slots;
    ctoff[u] = toff[u] - upv[u];
    slots += ((int64_t)nown[u] + 3) & ~(int64_t)3;
    maxP = std::max(maxP, nblk[u]);
    minP = std::min(minP, nblk[u]);
  }
  for (int r = 0; r < n_reads; ++r)
    if (pv[r] > 0) padoff[r] = toff[pad_unit[linear ? 0 : ucls[r]]] + t0[r];
  const int64_t Tp = ((slots + 32 + 255) / 256) * 256;  // + slack, then a multiple of every GEMM token tile
  DC_REQUIRE(Tp < ((int64_t)1 << 31), "batch too large (token count must fit int32)");
  DC_REQUIRE((int64_t)maxP * 32 <= m->cfg.max_seq_len + 31, "read longer than the model's max_seq_len");
  const int64_t n_tb = Tp / 4;  // entries of the slot -> unit map (one per 4 slots)

  // FFT classes by transform size; the others form groups of 32 units of similar length (longest first) for the
  // block-Toeplitz kernel.
  std::vector<int32_t> order(n_units);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return nblk[a] > nblk[b]; });
  // (linear form: the shared row runs in a launch of its own in front of the reads' -- they need its z and F -- as a group / a
  //  read list of one behind the others')
  const int lin_pad = linear && sp ? pad_unit[0] : -1;
  std::vector<int32_t> fft_lists[kFftClasses], mfma_order;
  for (int j = 0; j < n_units; ++j) {
    const int u = order[j];
    if (u == lin_pad) continue;
    if (ucls[u] > 0) fft_lists[ucls[u] - 1].push_back(u);
    else mfma_order.push_back(u);
  }
  const int n_mfma = (int)mfma_order.size();
  const int n_groups = (n_mfma + 31) / 32;
  std::vector<int32_t> grp_reads((size_t)(n_groups + 1) * 32, -1), grp_P(n_groups + 1, 0);
  int maxP_mfma = 0;
  for (int j = 0; j < n_mfma; ++j) {
    grp_reads[j] = mfma_order[j];
    grp_P[j / 32] = std::max(grp_P[j / 32], nblk[mfma_order[j]]);
    maxP_mfma = std::max(maxP_mfma, nblk[mfma_order[j]]);
  }
  std::vector<int32_t> fft_reads;
  for (int b = 0; b < kFftClasses; ++b) {
    m->fft_offset[b] = (int)fft_reads.size();
    m->fft_count[b] = (int)fft_lists[b].size();
    fft_reads.insert(fft_reads.end(), fft_lists[b].begin(), fft_lists[b].end());
  }
  m->lin_pad_cls = -1;
  if (lin_pad >= 0) {
    m->lin_pad_cls = ucls[lin_pad];
    m->lin_pad_len = pad_len[0];
    m->lin_pad_P = nblk[lin_pad];
    m->lin_pad_toff = toff[lin_pad];
    if (ucls[lin_pad] == 0) {
      grp_reads[(size_t)n_groups * 32] = lin_pad;
      grp_P[n_groups] = nblk[lin_pad];
    }
    m->lin_pad_fft_index = (int)fft_reads.size();
    fft_reads.push_back(lin_pad);
  }
  std::vector<int32_t> blk_read(n_tb, -1);  // unit that owns slots [4 i, 4 i + 4), -1 = none
  for (int u = 0; u < n_units; ++u) {
    const int64_t q0 = toff[u] / 4, q1 = q0 + (nown[u] + 3) / 4;
    std::fill(blk_read.begin() + q0, blk_read.begin() + q1, u);
  }
  // pad columns of the caller's dense logits <- the shared rows' logits
  std::vector<int64_t> fill_dst(std::max(n_fill, 1));
  std::vector<int32_t> fill_n(std::max(n_fill, 1)), fill_src(std::max(n_fill, 1));
  if (h_npad) {
    int j = 0;
    for (int r = 0; r < n_reads; ++r)
      if (pv[r] > 0) {  // (the last <= 3 pads of a run are the read's own: their logits come from its own slots)
        fill_dst[j] = (int64_t)r * row_stride + h_first[r];
        fill_n[j] = pv[r];
        fill_src[j] = linear ? 0 : padlog_off[ucls[r]];
        ++j;
      }
  }

  // layout inside one staging / device buffer (all 8-byte aligned)
  auto al8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
  const size_t nu = (size_t)n_units, i4 = sizeof(int32_t);
  size_t o_src = 0;
  size_t o_fdst = al8(o_src + sizeof(int64_t) * nu);
  size_t o_blk = al8(o_fdst + sizeof(int64_t) * fill_dst.size());
  size_t o_toff = al8(o_blk + i4 * n_tb);
  size_t o_ntok = al8(o_toff + i4 * nu);
  size_t o_nblk = al8(o_ntok + i4 * nu);
  size_t o_ctoff = al8(o_nblk + i4 * nu);
  size_t o_cntok = al8(o_ctoff + i4 * nu);
  size_t o_pv = al8(o_cntok + i4 * nu);
  size_t o_pad = al8(o_pv + i4 * nu);
  size_t o_db = al8(o_pad + i4 * nu);
  size_t o_fn = al8(o_db + i4 * nu);
  size_t o_fs = al8(o_fn + i4 * fill_n.size());
  size_t o_gr = al8(o_fs + i4 * fill_src.size());
  size_t o_gp = al8(o_gr + i4 * grp_reads.size());
  size_t o_fr = al8(o_gp + i4 * grp_P.size());
  size_t total = al8(o_fr + i4 * std::max<size_t>(fft_reads.size(), 1));

  if (m->staging_pending) {
    DC_HIP_CHECK(hipEventSynchronize(m->staging_done));
    m->staging_pending = false;
  }
  if (m->pinned_bytes < total) {
    if (m->pinned) DC_HIP_CHECK(hipHostFree(m->pinned));
    m->pinned = nullptr;
    DC_HIP_CHECK(hipHostMalloc(&m->pinned, total, hipHostMallocDefault));
    m->pinned_bytes = total;
  }
  if (m->idx.bytes < total) {
    DC_HIP_CHECK(hipStreamSynchronize(st));  // an earlier forward may still read the old arrays
    int rc = ensure(m->idx, total);
    if (rc) return rc;
  }
  if (sp && m->padlog.bytes < (size_t)padlog_len * 2 * sizeof(float)) {
    DC_HIP_CHECK(hipStreamSynchronize(st));
    int rc = ensure(m->padlog, (size_t)padlog_len * 2 * sizeof(float));
    if (rc) return rc;
  }
  char* hp = (char*)m->pinned;
  memcpy(hp + o_src, src_base.data(), sizeof(int64_t) * nu);
  memcpy(hp + o_fdst, fill_dst.data(), sizeof(int64_t) * fill_dst.size());
  memcpy(hp + o_blk, blk_read.data(), i4 * n_tb);
  memcpy(hp + o_toff, toff.data(), i4 * nu);
  memcpy(hp + o_ntok, nown.data(), i4 * nu);
  memcpy(hp + o_nblk, nblk.data(), i4 * nu);
  memcpy(hp + o_ctoff, ctoff.data(), i4 * nu);
  memcpy(hp + o_cntok, cntok.data(), i4 * nu);
  memcpy(hp + o_pv, upv.data(), i4 * nu);
  memcpy(hp + o_pad, padoff.data(), i4 * nu);
  memcpy(hp + o_db, dbase.data(), i4 * nu);
  memcpy(hp + o_fn, fill_n.data(), i4 * fill_n.size());
  memcpy(hp + o_fs, fill_src.data(), i4 * fill_src.size());
  memcpy(hp + o_gr, grp_reads.data(), i4 * grp_reads.size());
  memcpy(hp + o_gp, grp_P.data(), i4 * grp_P.size());
  if (!fft_reads.empty()) memcpy(hp + o_fr, fft_reads.data(), i4 * fft_reads.size());
  DC_HIP_CHECK(hipMemcpyAsync(m->idx.p, m->pinned, total, hipMemcpyHostToDevice, st));
  DC_HIP_CHECK(hipEventRecord(m->staging_done, st));
  m->staging_pending = true;

  char* dp = (char*)m->idx.p;
  m->d_src_base = (int64_t*)(dp + o_src);
  m->d_fill_dst = (int64_t*)(dp + o_fdst);
  m->d_blk_read = (int32_t*)(dp + o_blk);
  m->d_toff = (int32_t*)(dp + o_toff);
  m->d_ntok = (int32_t*)(dp + o_ntok);
  m->d_nblk = (int32_t*)(dp + o_nblk);
  m->d_ctoff = (int32_t*)(dp + o_ctoff);
  m->d_cntok = (int32_t*)(dp + o_cntok);
  m->d_pv = (int32_t*)(dp + o_pv);
  m->d_padoff = (int32_t*)(dp + o_pad);
  m->d_dbase = (int32_t*)(dp + o_db);
  m->d_fill_n = (int32_t*)(dp + o_fn);
  m->d_fill_src = (int32_t*)(dp + o_fs);
  m->d_grp_reads = (int32_t*)(dp + o_gr);
  m->d_grp_P = (int32_t*)(dp + o_gp);
  m->d_fft_reads = (int32_t*)(dp + o_fr);
  m->shared_prefix = sp;
  m->linear_prefix = linear && sp;
  if (dstride > m->dcap) {
    // The rows keep the stride they were allocated with (a new stride would move the zero stretches onto old data): grow in
    // steps, zero-filled -- the stretch in front of the shared row's positions is what reads without a pad run subtract / add
    const int64_t cap = ((dstride + 8191) / 8192) * 8192;
    const size_t need = (size_t)m->cfg.d_model * cap * sizeof(float);
    DC_HIP_CHECK(hipStreamSynchronize(st));
    int rc;
    if ((rc = ensure(m->zc, need)) || (rc = ensure(m->Fc, need))) return rc;
    m->dcap = cap;
  }
  m->n_fill_rows = n_fill;
  m->pad_id = pad_id;
  m->max_P_mfma = maxP_mfma;
  // units are ordered longest first: the groups of more than 32 blocks (1 024 tokens) are a prefix
  m->n_groups_long = 0;
  for (int g = 0; g < n_groups; ++g) m->n_groups_long += grp_P[g] > 32;
  m->Tp = Tp;
  m->n_groups = n_groups;
  m->max_P = maxP;
  m->min_P = minP;
  m->h_geom.swap(key);
  return DC_OK;
}

// Final LayerNorm (folded into linear1) + token-classification head on the residual stream in m->x.
// head_sums: mean / rstd hold the per-token sum / sum of squares left by the last fc2 epilogue;
// head1_ready: linear1 already ran inside the last block's fused MLP kernel (its output is in m->hmid).
int run_head(dc_model* m, hipStream_t st, bool head_sums, bool head1_ready, float* d_logits) {
  const auto& c = m->cfg;
  const int D = c.d_model, H = c.head_hidden;
  float* x = (float*)m->x.p;
  float* hm = (float*)m->hmid.p;
  float* ob = (float*)m->obuf.p;
  float* mean = (float*)m->mean.p;
  float* rstd = (float*)m->rstd.p;
  float* qpad = (float*)m->qpad.p;
  const int64_t Tp = m->Tp;
  int rc;
  if (!head_sums && !head1_ready) {
    ProfScope ps(m, st, "ln_stats");
    if ((rc = launch_ln_stats(st, x, D, Tp, c.ln_eps, mean, rstd))) return rc;
  }
  if (!head1_ready &&
      (rc = run_gemm(m, st, "gemm_head1", WRef{m->Wt_l1, m->Wp_l1, m->Wq_l1, m->is_l1}, x, hm, m->b_l1, nullptr, qpad, true, EPI_BIAS_RELU_Q, D, H,
                     head_sums)))
    return rc;
  const bool no_head_fuse = getenv("DC_NO_HEAD_FUSE") != nullptr;  // A/B switch
  if (m->gemm_mode >= 1 && !no_head_fuse) {
    // linear2 + residual + ReLU + linear3 in one kernel: the 1024-wide activation never reaches HBM
    if ((rc = run_gemm(m, st, "gemm_head2", WRef{m->Wt_l2, m->Wp_l2, m->Wq_l2, m->is_l2}, hm, ob, m->b_l2, hm, nullptr, false, EPI_HEAD2_FUSED, H,
                       H)))
      return rc;
    ProfScope ps(m, st, "head_out");
    if ((rc = launch_head_reduce(st, (const float*)m->part.p, H / gemm6_channel_block(), m->b3, Tp, m->d_blk_read, m->d_toff,
                                 m->d_src_base, m->d_ntok, d_logits, (float*)m->padlog.p)))
      return rc;
  } else {
    if ((rc = run_gemm(m, st, "gemm_head2", WRef{m->Wt_l2, m->Wp_l2, m->Wq_l2, m->is_l2}, hm, ob, m->b_l2, hm, nullptr, false, EPI_BIAS_RES_RELU,
                       H, H)))
      return rc;
    ProfScope ps(m, st, "head_out");
    if ((rc = launch_head_out(st, ob, m->W3, m->b3, H, c.num_class, Tp, m->d_blk_read, m->d_toff, m->d_src_base,
                              m->d_ntok, d_logits, (float*)m->padlog.p)))
      return rc;
  }
  return DC_OK;
}

int forward_core(dc_model* m, hipStream_t st, const void* d_ids, int ids_i64, const float* d_quals, int n_reads,
                 int64_t row_stride, const int32_t* h_first, const int32_t* h_ntok, float* d_logits,
                 const int32_t* h_npad = nullptr, int pad_id = 0) {
  const auto& c = m->cfg;
  int rc;
  if ((rc = setup_geometry(m, st, n_reads, row_stride, h_first, h_ntok, h_npad, pad_id))) return rc;
  if ((rc = ensure_workspace(m, m->Tp))) return rc;
  // taps: the longest read, and every tap an FFT size class in use transforms (its spectrum is built from min(taps, M)
  // taps: materialising all M makes it -- and with it the last bits of the result -- independent of what the handle saw before)
  int64_t taps_needed = (int64_t)m->max_P * 32;
  const int pad_fft = m->linear_prefix && m->lin_pad_cls > 0 ? m->lin_pad_cls - 1 : -1;  // the linear form's shared row on an FFT class
  for (int b = 0; b < kFftClasses; ++b)
    if (m->fft_count[b] || b == pad_fft) taps_needed = std::max<int64_t>(taps_needed, (int64_t)kFftClass[b].points);
  if ((rc = ensure_filters(m, st, taps_needed))) return rc;
  for (int b = 0; b < kFftClasses; ++b)
    if ((m->fft_count[b] || b == pad_fft) && (rc = ensure_fft_spectra(m, st, b))) return rc;
  // reads per launch pair of the 2^15 form.  The scratch is D x 2^15 complex = 64 MiB per read and stays with the handle, so
  // the batch is what bounds a handle's footprint: 4 reads = 256 MiB at most (16 would hold 1 GiB for good, per handle --
  // lanes, or ranks sharing a GPU), and a launch of 4 reads is already 8 x 4 x 256 = 8 192 workgroups
  constexpr int kFft15Batch = 4;
  if (m->fft_count[kFft15Class] || pad_fft == kFft15Class) {
    const size_t need = (size_t)hyena_fft15_scratch_bytes(std::max(1, std::min(m->fft_count[kFft15Class], kFft15Batch)), c.d_model);
    if (m->fft_scratch.bytes < need) {
      DC_HIP_CHECK(hipStreamSynchronize(st));  // earlier launches may still read the buffer about to be freed
      // every (k1, b) of a (read, channel) is stored by the first kernel before the second one loads it: no zero fill
      if ((rc = ensure(m->fft_scratch, need, /*zero_fill=*/false))) return rc;
    }
  }

  const int D = c.d_model, DI = c.d_inner, H = c.head_hidden;
  float* x = (float*)m->x.p;
  float* u = (float*)m->u.p;
  float* y = (float*)m->y.p;
  float* hm = (float*)m->hmid.p;
  float* mean = (float*)m->mean.p;
  float* rstd = (float*)m->rstd.p;
  float* qpad = (float*)m->qpad.p;
  const int64_t Tp = m->Tp;

  ProfScope whole(m, st, "forward");
  const bool no_fused_stats = getenv("DC_NO_FUSED_STATS") != nullptr;  // A/B switch
  // bf16x6 mode with the layer-0 table: every LayerNorm input is produced by an out_proj / fc2 GEMM whose
  // epilogue also accumulates the per-token sum and sum of squares -> no separate statistics pass
  const bool fused_stats = !no_fused_stats && m->gemm_mode >= 1;
  const bool no_l0_table = getenv("DC_NO_L0_TABLE") != nullptr;  // A/B switch
  const bool l0_table = !no_l0_table && m->u_table0 != nullptr;
  // layer 0 with the 16-bit long convolution: its kernel gathers u from the per-token-id table itself, so the
  // embed kernel writes the packed token ids (1 B/token) instead of u (3 KB/token)
  const bool l0_gather = l0_table && getenv("DC_NO_L0_GATHER") == nullptr && m->gemm_mode >= 1 && c.vocab <= 16 &&
                         !(getenv("DC_HYENA") && atoi(getenv("DC_HYENA")) != 3);
  unsigned char* tokp = (unsigned char*)m->tok.p;
  {
    ProfScope ps(m, st, "embed");
    rc = l0_table ? launch_embed_l0(st, d_ids, ids_i64, d_quals, m->emb, m->u_table0, c.vocab, D, m->d_blk_read,
                                    m->d_toff, m->d_src_base, m->d_ntok, Tp, x, l0_gather ? nullptr : u, qpad,
                                    l0_gather ? tokp : nullptr, m->pad_id)
                  : launch_embed(st, d_ids, ids_i64, d_quals, m->emb, c.vocab, D, m->d_blk_read, m->d_toff,
                                 m->d_src_base, m->d_ntok, Tp, x, qpad, m->pad_id);
    if (rc) return rc;
  }
  const bool no_mlp_chain = getenv("DC_NO_MLP_CHAIN") != nullptr;  // A/B switch
  bool u_ready = false;     // the previous block's fused MLP kernel already produced this block's in_proj output
  bool head1_ready = false;  // ... or the head's linear1 output (last block)
  for (int li = 0; li < c.n_layer; ++li) {
    LayerDev& L = m->layers[li];
    const bool sums = fused_stats && l0_table;  // statistics of x come from the previous producer's epilogue
    if (!(li == 0 && l0_table) && !u_ready) {  // layer 0: u was gathered from the per-token-id table by the embed kernel
      if (!sums) {
        ProfScope ps(m, st, "ln_stats");
        if ((rc = launch_ln_stats(st, x, D, Tp, c.ln_eps, mean, rstd))) return rc;
      }
      if ((rc = run_gemm(m, st, "gemm_inproj", WRef{L.Wt_in, L.Wp_in, L.Wq_in, L.is_in}, x, u, L.b_in, nullptr, nullptr, true, EPI_BIAS, D,
                         3 * D, sums)))
        return rc;
    }
    auto zero_stats = [&]() -> int {
      DC_HIP_CHECK(hipMemsetAsync(mean, 0, (size_t)Tp * sizeof(float), st));
      DC_HIP_CHECK(hipMemsetAsync(rstd, 0, (size_t)Tp * sizeof(float), st));
      return DC_OK;
    };
    {
      HyenaArgs a{};
      a.u = u;
      a.y = y;
      a.kf = L.kf;
      a.kf_stride = m->kf_taps;
      a.sw = L.sw;
      a.sb = L.sb;
      a.grp_reads = m->d_grp_reads;
      a.grp_P = m->d_grp_P;
      a.toff = m->d_ctoff;   // the long convolution's view of a unit: its whole sequence, shared [PAD] prefix included
      a.nblk = m->d_nblk;
      a.ntok = m->d_cntok;
      if (m->shared_prefix) {
        a.pv = m->d_pv;
        a.padoff = m->d_padoff;
      }
      a.Tp = Tp;
      a.D = D;
      a.kfp = L.kfp;
      a.kfp_nu = hyena3_units((int)m->kf_taps);
      a.kf_taps = (int)m->kf_taps;
      a.kfp_f16 = L.kfq;
      a.ch_inv_scale = L.kinv;
      a.split_mode = m->gemm_mode == 2 ? 1 : 0;
      a.overflow = m->ovf_target ? m->ovf_target : m->d_overflow;
      if (li == 0 && l0_gather) {
        a.tok = tokp;
        a.tab = m->u_table0;
        a.vocab = c.vocab;
      }
      ProfScope ps(m, st, "hyena");
      // long-convolution kernel: 3 = 16-bit split MFMA (with the split GEMM modes), 2 = fp32 MFMA
      const int hy_env = getenv("DC_HYENA") ? atoi(getenv("DC_HYENA")) : 0;
      const int hy = hy_env ? hy_env : (m->gemm_mode >= 1 ? 3 : 2);
      if (m->linear_prefix) {
        // the shared [PAD] row first, alone, with a unit output gate: F = k * z_pad into its y slots; then z_pad, F -> zc / Fc
        // and y = F * gate (launch_pad_row_finish).  The reads' launches below subtract z_pad and add F.
        ProfScope pp(m, st, "hyena_pad_row");
        HyenaArgs ap = a;
        ap.pv = ap.padoff = ap.dbase = nullptr;
        ap.sw = L.sw_unit;
        ap.sb = L.sb_unit;
        const int k = m->lin_pad_cls;
        if (k == 0) {
          ap.grp_reads = a.grp_reads + (size_t)m->n_groups * 32;
          ap.grp_P = a.grp_P + m->n_groups;
          rc = hy == 2 ? launch_hyena2(st, ap, 1, m->lin_pad_P) : launch_hyena3(st, ap, 1, m->lin_pad_P);
        } else if (k - 1 < kFft15Class) {
          rc = launch_hyena_fft(st, ap, k - 1, m->d_fft_reads + m->lin_pad_fft_index, 1, m->fft_tw, L.kspec[k - 1]);
        } else {
          rc = launch_hyena_fft15(st, ap, m->d_fft_reads + m->lin_pad_fft_index, 1, m->fft_tw, L.kspec[kFft15Class],
                                  (float*)m->fft_scratch.p);
        }
        if (rc) return rc;
        if ((rc = launch_pad_row_finish(st, u, a.tok ? a.tab : nullptr, m->pad_id, D, Tp, m->lin_pad_toff, m->lin_pad_len, L.sw, L.sb, y,
                                        (float*)m->zc.p, (float*)m->Fc.p, m->dcap, kPrefixZero)))
          return rc;
        a.dbase = m->d_dbase;
        a.zc = (const float*)m->zc.p;
        a.Fc = (const float*)m->Fc.p;
        a.dstride = m->dcap;
      }
      if (m->n_groups > 0) {
        const bool one_launch = getenv("DC_H3_ONE_LAUNCH") != nullptr;  // A/B
        if (hy == 3 && !one_launch && m->n_groups_long > 0 && m->n_groups_long < m->n_groups) {
          // mixed lengths: the 32-block form (one filter window, measured 6 % faster at 1 kb) for the groups that fit
          // it, the 16-block form only for the longer ones -- two launches over disjoint group ranges
          rc = launch_hyena3(st, a, m->n_groups_long, m->max_P_mfma);
          if (rc) return rc;
          HyenaArgs b = a;
          b.grp_reads = a.grp_reads + (size_t)m->n_groups_long * 32;
          b.grp_P = a.grp_P + m->n_groups_long;
          rc = launch_hyena3(st, b, m->n_groups - m->n_groups_long, 32);
        } else {
          rc = hy == 2 ? launch_hyena2(st, a, m->n_groups, m->max_P_mfma) : launch_hyena3(st, a, m->n_groups, m->max_P_mfma);
        }
        if (rc) return rc;
      }
      // long reads: O(L log L) FFT convolution, one launch per transform size present in the batch
      for (int b = 0; b < kFft15Class; ++b) {
        if (m->fft_count[b] == 0) continue;
        ProfScope pf(m, st, "hyena_fft");
        if ((rc = launch_hyena_fft(st, a, b, m->d_fft_reads + m->fft_offset[b], m->fft_count[b], m->fft_tw, L.kspec[b])))
          return rc;
      }
      for (int r0 = 0; r0 < m->fft_count[kFft15Class]; r0 += kFft15Batch) {  // 16 385 .. 32 769 tokens: 16 x 2048 four-step form
        ProfScope pf(m, st, "hyena_fft15");
        if ((rc = launch_hyena_fft15(st, a, m->d_fft_reads + m->fft_offset[kFft15Class] + r0,
                                     std::min(kFft15Batch, m->fft_count[kFft15Class] - r0), m->fft_tw, L.kspec[kFft15Class],
                                     (float*)m->fft_scratch.p)))
          return rc;
      }
    }
    const bool no_mlp_fuse = getenv("DC_NO_MLP_FUSE") != nullptr;  // A/B switch
    // fp16x3 (mode 2) and, since round 6, the strict bf16x6 arithmetic (mode 1: mlp_fused6_kernel; DC_NO_MLP_FUSE6 keeps its
    // three plain GEMM launches, A/B)
    const bool strict = m->gemm_mode == 1;
    const bool fused_mlp = (m->gemm_mode == 2 || (strict && getenv("DC_NO_MLP_FUSE6") == nullptr)) && sums && !no_mlp_fuse &&
                           m->mlp_fusion && (strict ? L.Wp_fc2p : L.Wq_fc2p) && D == 256 && DI % 128 == 0 && DI <= 1024;
    if (sums && (rc = zero_stats())) return rc;
    if ((rc = run_gemm(m, st, "gemm_outproj", WRef{L.Wt_out, L.Wp_out, L.Wq_out, L.is_out}, y, x, L.b_out, x, nullptr, false,
                       sums ? EPI_BIAS_RES_STATS : EPI_BIAS_RES, D, D)))
      return rc;
    if (!sums) {
      ProfScope ps(m, st, "ln_stats");
      if ((rc = launch_ln_stats(st, x, D, Tp, c.ln_eps, mean, rstd))) return rc;
    }
    if (fused_mlp) {
      // fc1 -> GELU -> fc2 -> +residual -> LayerNorm statistics in one kernel; the d_inner-wide
      // activation stays in registers (fp16x3 only)
      GemmArgs g{};
      g.Wp = strict ? L.Wp_fc1 : L.Wq_fc1;
      g.Wp2 = strict ? L.Wp_fc2p : L.Wq_fc2p;
      g.split_mode = strict ? 0 : 1;
      g.w_inv_scale = strict ? 1.0f : L.is_fc1;  // (bf16 planes hold the weights themselves: no power-of-two scale to undo)
      g.w_inv_scale2 = strict ? 1.0f : L.is_fc2;
      g.bias = L.b_fc1;
      g.bias2 = L.b_fc2;
      g.overflow = m->ovf_target ? m->ovf_target : m->d_overflow;
      g.X = x;
      g.R = x;
      g.C = x;
      g.mean = mean;
      g.rstd = rstd;
      g.st_sum = mean;
      g.st_sq = rstd;
      g.stats_are_sums = 1;
      g.ln_eps = c.ln_eps;
      g.K = D;
      g.N = DI;
      g.Tp = Tp;
      g.gelu_tanh = c.gelu_tanh;
      // chain: the GEMM that consumes LayerNorm(this block's output) runs inside the same kernel, its B operand
      // taken from the accumulator registers: the next block's in_proj, or the head's linear1 after the last block
      u_ready = false;
      const bool chain = !no_mlp_chain && m->mlp_chain;
      if (chain && li + 1 < c.n_layer && (strict ? m->layers[li + 1].Wp_inp : m->layers[li + 1].Wq_inp) && (3 * D) % 128 == 0) {
        LayerDev& Ln = m->layers[li + 1];
        g.Wp3 = strict ? Ln.Wp_inp : Ln.Wq_inp;
        g.bias3 = Ln.b_in;
        g.w_inv_scale3 = strict ? 1.0f : Ln.is_in;
        g.N3 = 3 * D;
        g.epi3 = EPI_BIAS;
        g.C3 = u;
        u_ready = true;
      } else if (chain && li + 1 == c.n_layer && (strict ? m->Wp_l1p : m->Wq_l1p) && H % 128 == 0) {
        g.Wp3 = strict ? m->Wp_l1p : m->Wq_l1p;
        g.bias3 = m->b_l1;
        g.w_inv_scale3 = strict ? 1.0f : m->is_l1;
        g.N3 = H;
        g.epi3 = EPI_BIAS_RELU_Q;
        g.C3 = hm;
        g.q = qpad;
        head1_ready = true;
      }
      ProfScope ps(m, st, g.Wp3 ? (g.epi3 == EPI_BIAS ? "mlp_inproj_fused" : "mlp_head1_fused") : "mlp_fused");
      if ((rc = launch_mlp_fused(st, g))) return rc;
    } else {
      u_ready = false;
      if ((rc = run_gemm(m, st, "gemm_fc1", WRef{L.Wt_fc1, L.Wp_fc1, L.Wq_fc1, L.is_fc1}, x, hm, L.b_fc1, nullptr, nullptr, true,
                         EPI_BIAS_GELU, D, DI, sums)))
        return rc;
      if (sums && (rc = zero_stats())) return rc;
      if ((rc = run_gemm(m, st, "gemm_fc2", WRef{L.Wt_fc2, L.Wp_fc2, L.Wq_fc2, L.is_fc2}, hm, x, L.b_fc2, x, nullptr, false,
                         sums ? EPI_BIAS_RES_STATS : EPI_BIAS_RES, DI, D)))
        return rc;
    }
  }
  if ((rc = run_head(m, st, fused_stats && l0_table, head1_ready, d_logits))) return rc;
  if (m->n_fill_rows > 0) {  // shared [PAD] rows: their logits into every row's pad columns (+ the contract check)
    ProfScope ps(m, st, "pad_fill");
    return launch_fill_pad_logits(st, d_ids, ids_i64, d_quals, m->n_fill_rows, m->max_P * 32, m->d_fill_dst, m->d_fill_n,
                                  m->d_fill_src, (const float*)m->padlog.p, m->pad_id, d_logits, m->d_prefix_bad);
  }
  return DC_OK;
}

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" int dc_model_create(const dc_model_config* cfg, const dc_model_weights* w, dc_model** out) {
  DC_REQUIRE(cfg && w && out, "null pointer");
  DC_REQUIRE(cfg->d_model == 256, "this build is specialised for d_model == 256 (dc_hg.py:96,147)");
  DC_REQUIRE(cfg->d_inner % 128 == 0 && cfg->head_hidden % 128 == 0, "d_inner / head_hidden must be multiples of 128");
  DC_REQUIRE(cfg->num_class == 2, "num_class must be 2");
  DC_REQUIRE(cfg->short_filter_order == 3, "short_filter_order must be 3");
  DC_REQUIRE(cfg->filter_order <= 64 && cfg->emb_dim <= 8, "implicit filter wider than supported");
  DC_REQUIRE(cfg->n_layer >= 1 && cfg->vocab >= 1 && cfg->vocab <= 256, "bad n_layer / vocab");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
    set_error("no HIP device available: libdeepchopper_hip has no CPU fallback");
    return DC_ERR_NO_DEVICE;
  }
  dc_model* m = new dc_model();
  m->cfg = *cfg;
  if (const char* e = getenv("DC_GEMM_MODE"))
    m->gemm_mode = (strcmp(e, "f32") == 0 || strcmp(e, "0") == 0) ? 0 : ((strcmp(e, "fp16x3") == 0 || strcmp(e, "2") == 0) ? 2 : 1);
  if (const char* e = getenv("DC_FFT_MIN")) m->fft_min_tokens = std::max(256, atoi(e));
  {
    const char* e = getenv("DC_FFT_SPLIT");
    m->fft_class_mask = fft_mask_for(!(e && atoi(e) == 0));
    // experiments: an explicit set of size classes (bit k = kFftClass[k]); the four-step class is always available
    if (const char* cm = getenv("DC_FFT_CLASS_MASK")) m->fft_class_mask = ((uint32_t)strtoul(cm, nullptr, 0) | 1u << kFft15Class) & ((1u << kFftClasses) - 1);
  }
  if (const char* e = getenv("DC_PREFIX_MODE")) m->prefix_mode = (strcmp(e, "linear") == 0 || strcmp(e, "2") == 0) ? 2 : 1;
  DC_HIP_CHECK(hipGetDevice(&m->device));
  DC_HIP_CHECK(hipEventCreateWithFlags(&m->staging_done, hipEventDisableTiming));
  {
    void* p = nullptr;
    int rc0 = dev_alloc(m, sizeof(int), &p);
    if (rc0) return rc0;
    m->d_overflow = (int*)p;
    DC_HIP_CHECK(hipMemset(p, 0, sizeof(int)));
    void* pb = nullptr;
    if ((rc0 = dev_alloc(m, sizeof(int), &pb))) return rc0;
    m->d_prefix_bad = (int*)pb;
    DC_HIP_CHECK(hipMemset(pb, 0, sizeof(int)));
    DC_HIP_CHECK(hipStreamSynchronize(nullptr));  // NULL-stream memset: complete before any caller stream can touch it
  }
  const int D = cfg->d_model, DI = cfg->d_inner, H = cfg->head_hidden, fo = cfg->filter_order, emb = cfg->emb_dim;
  int rc;
#define UP(expr)        \
  if ((rc = (expr))) {  \
    dc_model_free(m);   \
    return rc;          \
  }
  UP(upload_raw(m, w->embedding, (size_t)cfg->vocab * D, &m->emb));
  m->layers.resize(cfg->n_layer);
  std::vector<float> Wt, bo;
  for (int li = 0; li < cfg->n_layer; ++li) {
    const dc_layer_weights& lw = w->layers[li];
    LayerDev& L = m->layers[li];
    fold_transpose(lw.in_proj_w, lw.in_proj_b, lw.norm1_w, lw.norm1_b, 3 * D, D, Wt, bo);
    UP(upload(m, Wt, &L.Wt_in));
    UP(upload_planes(m, Wt, D, 3 * D, &L.Wp_in));
    UP(upload_planes_f16(m, Wt, D, 3 * D, &L.Wq_in, &L.is_in));
    if (D % 16 == 0) {  // k order of the fused MLP kernel's third GEMM
      std::vector<float> Wtp((size_t)D * 3 * D);
      for (int k = 0; k < D; ++k)
        memcpy(&Wtp[(size_t)k * 3 * D], &Wt[(size_t)(mlp_fused_perm(k)) * 3 * D], (size_t)3 * D * sizeof(float));
      float is_p = 0.f;
      UP(upload_planes_f16(m, Wtp, D, 3 * D, &L.Wq_inp, &is_p));
      UP(upload_planes(m, Wtp, D, 3 * D, &L.Wp_inp));
    }
    if (li == 0) {
      // u0[v][n] = b'[n] + sum_k Wt[k][n] * (emb[v][k] - mean_v) * rstd_v   (gamma / beta already folded), fp64
      std::vector<float> tab((size_t)cfg->vocab * 3 * D);
      for (int v = 0; v < cfg->vocab; ++v) {
        const float* e = w->embedding + (size_t)v * D;
        double mu = 0.0, var = 0.0;
        for (int k = 0; k < D; ++k) mu += e[k];
        mu /= D;
        for (int k = 0; k < D; ++k) var += ((double)e[k] - mu) * ((double)e[k] - mu);
        var /= D;
        const double rstd = 1.0 / std::sqrt(var + (double)cfg->ln_eps);
        for (int n = 0; n < 3 * D; ++n) {
          double acc = bo[n];
          for (int k = 0; k < D; ++k) acc += (double)Wt[(size_t)k * 3 * D + n] * (((double)e[k] - mu) * rstd);
          tab[(size_t)v * 3 * D + n] = (float)acc;
        }
      }
      UP(upload(m, tab, &m->u_table0));
    }
    UP(upload(m, bo, &L.b_in));
    UP(upload_raw(m, lw.short_w, (size_t)3 * D * 3, &L.sw));
    UP(upload_raw(m, lw.short_b, (size_t)3 * D, &L.sb));
    {
      std::vector<float> swu(lw.short_w, lw.short_w + (size_t)3 * D * 3), sbu(lw.short_b, lw.short_b + (size_t)3 * D);
      std::fill(swu.begin(), swu.begin() + (size_t)D * 3, 0.f);
      std::fill(sbu.begin(), sbu.begin() + D, 1.f);
      UP(upload(m, swu, &L.sw_unit));
      UP(upload(m, sbu, &L.sb_unit));
    }
    fold_transpose(lw.out_proj_w, lw.out_proj_b, nullptr, nullptr, D, D, Wt, bo);
    UP(upload(m, Wt, &L.Wt_out));
    UP(upload_planes(m, Wt, D, D, &L.Wp_out));
    UP(upload_planes_f16(m, Wt, D, D, &L.Wq_out, &L.is_out));
    UP(upload(m, bo, &L.b_out));
    fold_transpose(lw.fc1_w, lw.fc1_b, lw.norm2_w, lw.norm2_b, DI, D, Wt, bo);
    UP(upload(m, Wt, &L.Wt_fc1));
    UP(upload_planes(m, Wt, D, DI, &L.Wp_fc1));
    UP(upload_planes_f16(m, Wt, D, DI, &L.Wq_fc1, &L.is_fc1));
    UP(upload(m, bo, &L.b_fc1));
    fold_transpose(lw.fc2_w, lw.fc2_b, nullptr, nullptr, D, DI, Wt, bo);
    UP(upload(m, Wt, &L.Wt_fc2));
    UP(upload_planes(m, Wt, DI, D, &L.Wp_fc2));
    UP(upload_planes_f16(m, Wt, DI, D, &L.Wq_fc2, &L.is_fc2));
    if (DI % 16 == 0) {  // fused MLP: logical k index 16 b + kappa stands for hidden unit 16 b + perm16(kappa)
      std::vector<float> Wtp((size_t)DI * D);
      for (int k = 0; k < DI; ++k)
        memcpy(&Wtp[(size_t)k * D], &Wt[(size_t)(mlp_fused_perm(k)) * D], (size_t)D * sizeof(float));
      float is_p = 0.f;
      UP(upload_planes_f16(m, Wtp, DI, D, &L.Wq_fc2p, &is_p));  // same values, same power-of-two scale as Wq_fc2
      UP(upload_planes(m, Wtp, DI, D, &L.Wp_fc2p));
    }
    UP(upload(m, bo, &L.b_fc2));
    UP(upload_raw(m, lw.pos_z, (size_t)cfg->max_seq_len * emb, &L.pos_z));
    UP(upload_raw(m, lw.pos_t, (size_t)cfg->max_seq_len, &L.pos_t));
    UP(upload_raw(m, lw.filt_w0, (size_t)fo * emb, &L.fw0));
    UP(upload_raw(m, lw.filt_b0, fo, &L.fb0));
    UP(upload_raw(m, lw.filt_w1, (size_t)fo * fo, &L.fw1));
    UP(upload_raw(m, lw.filt_b1, fo, &L.fb1));
    UP(upload_raw(m, lw.filt_w2, (size_t)fo * fo, &L.fw2));
    UP(upload_raw(m, lw.filt_b2, fo, &L.fb2));
    UP(upload_raw(m, lw.filt_w3, (size_t)D * fo, &L.fw3));
    UP(upload_raw(m, lw.filt_freq0, fo, &L.freq0));
    UP(upload_raw(m, lw.filt_freq1, fo, &L.freq1));
    UP(upload_raw(m, lw.filt_freq2, fo, &L.freq2));
    UP(upload_raw(m, lw.deltas, D, &L.deltas));
    UP(upload_raw(m, lw.filt_bias, D, &L.fbias));
  }
  fold_transpose(w->head_w1, w->head_b1, w->ln_f_w, w->ln_f_b, H, D, Wt, bo);
  UP(upload(m, Wt, &m->Wt_l1));
  UP(upload_planes(m, Wt, D, H, &m->Wp_l1));
  UP(upload_planes_f16(m, Wt, D, H, &m->Wq_l1, &m->is_l1));
  if (D % 16 == 0) {
    std::vector<float> Wtp((size_t)D * H);
    for (int k = 0; k < D; ++k)
      memcpy(&Wtp[(size_t)k * H], &Wt[(size_t)(mlp_fused_perm(k)) * H], (size_t)H * sizeof(float));
    float is_p = 0.f;
    UP(upload_planes_f16(m, Wtp, D, H, &m->Wq_l1p, &is_p));
    UP(upload_planes(m, Wtp, D, H, &m->Wp_l1p));
  }
  UP(upload(m, bo, &m->b_l1));
  fold_transpose(w->head_w2, w->head_b2, nullptr, nullptr, H, H, Wt, bo);
  UP(upload(m, Wt, &m->Wt_l2));
  UP(upload_planes(m, Wt, H, H, &m->Wp_l2));
  UP(upload_planes_f16(m, Wt, H, H, &m->Wq_l2, &m->is_l2));
  UP(upload(m, bo, &m->b_l2));
  UP(upload_raw(m, w->head_w3, (size_t)cfg->num_class * H, &m->W3));
  UP(upload_raw(m, w->head_b3, cfg->num_class, &m->b3));
#undef UP
  *out = m;
  return DC_OK;
}

extern "C" void dc_model_free(dc_model* m) {
  if (!m) return;
  (void)hipDeviceSynchronize();
  for (void* p : m->owned) (void)hipFree(p);
  for (auto& L : m->layers) {
    if (L.kf) (void)hipFree(L.kf);
    if (L.kfp) (void)hipFree(L.kfp);
    if (L.kfq) (void)hipFree(L.kfq);
    if (L.kscale) (void)hipFree(L.kscale);
    if (L.kinv) (void)hipFree(L.kinv);
    for (float* p : L.kspec)
      if (p) (void)hipFree(p);
  }
  if (m->fft_tw) (void)hipFree(m->fft_tw);
  for (DevBuf* b : {&m->x, &m->u, &m->y, &m->hmid, &m->obuf, &m->mean, &m->rstd, &m->qpad, &m->idx, &m->part, &m->tok, &m->padlog,
                    &m->fft_scratch, &m->zc, &m->Fc})
    if (b->p) (void)hipFree(b->p);
  if (m->pinned) (void)hipHostFree(m->pinned);
  if (m->staging_done) (void)hipEventDestroy(m->staging_done);
  for (auto& e : m->prof_events) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  for (auto e : m->prof_pool) (void)hipEventDestroy(e);
  delete m;
}

extern "C" int dc_forward(dc_model* m, void* stream, const void* d_input_ids, int32_t ids_i64,
                          const float* d_input_quals, int32_t batch, int32_t seq_len, float* d_logits) {
  DC_REQUIRE(m && d_input_ids && d_input_quals && d_logits, "null pointer");
  DC_REQUIRE(batch >= 1 && seq_len >= 1, "batch and seq_len must be positive");
  std::vector<int32_t> first(batch, 0), ntok(batch, seq_len);
  return forward_core(m, as_stream(stream), d_input_ids, ids_i64, d_input_quals, batch, seq_len, first.data(),
                      ntok.data(), d_logits);
}

extern "C" int dc_forward_ragged(dc_model* m, void* stream, const void* d_input_ids, int32_t ids_i64,
                                 const float* d_input_quals, int32_t n_reads, int64_t row_stride,
                                 const int32_t* h_first, const int32_t* h_n_tokens, float* d_logits) {
  DC_REQUIRE(m && d_input_ids && d_input_quals && d_logits && h_first && h_n_tokens, "null pointer");
  DC_REQUIRE(n_reads >= 1 && row_stride >= 1, "n_reads and row_stride must be positive");
  return forward_core(m, as_stream(stream), d_input_ids, ids_i64, d_input_quals, n_reads, row_stride, h_first,
                      h_n_tokens, d_logits);
}

extern "C" int dc_forward_shared_prefix(dc_model* m, void* stream, const void* d_input_ids, int32_t ids_i64,
                                        const float* d_input_quals, int32_t n_reads, int64_t row_stride,
                                        const int32_t* h_first, const int32_t* h_n_tokens, const int32_t* h_n_pad,
                                        int32_t pad_token_id, float* d_logits) {
  DC_REQUIRE(m && d_input_ids && d_input_quals && d_logits && h_first && h_n_tokens && h_n_pad, "null pointer");
  DC_REQUIRE(n_reads >= 1 && row_stride >= 1, "n_reads and row_stride must be positive");
  DC_REQUIRE(pad_token_id >= 0 && pad_token_id < m->cfg.vocab, "pad_token_id outside the vocabulary");
  return forward_core(m, as_stream(stream), d_input_ids, ids_i64, d_input_quals, n_reads, row_stride, h_first,
                      h_n_tokens, d_logits, h_n_pad, pad_token_id);
}

extern "C" int dc_model_set_prefix_mode(dc_model* m, int32_t mode) {
  DC_REQUIRE(m, "null pointer");
  DC_REQUIRE(mode == 1 || mode == 2, "prefix mode must be 1 (exact form) or 2 (linear form)");
  m->prefix_mode = mode;
  return DC_OK;
}

extern "C" int dc_model_prefix_violations(dc_model* m, void* stream, int32_t* count) {
  DC_REQUIRE(m && count, "null pointer");
  hipStream_t st = as_stream(stream);
  int v = 0;
  DC_HIP_CHECK(hipMemcpyAsync(&v, m->d_prefix_bad, sizeof(int), hipMemcpyDeviceToHost, st));
  DC_HIP_CHECK(hipStreamSynchronize(st));
  if (v) DC_HIP_CHECK(hipMemsetAsync(m->d_prefix_bad, 0, sizeof(int), st));
  *count = v;
  return DC_OK;
}

extern "C" int dc_head_forward(dc_model* m, void* stream, const float* d_hidden, const float* d_input_quals,
                               int32_t batch, int32_t seq_len, float* d_logits) {
  DC_REQUIRE(m && d_hidden && d_input_quals && d_logits, "null pointer");
  DC_REQUIRE(batch >= 1 && seq_len >= 1, "batch and seq_len must be positive");
  hipStream_t st = as_stream(stream);
  std::vector<int32_t> first(batch, 0), ntok(batch, seq_len);
  int rc;
  if ((rc = setup_geometry(m, st, batch, seq_len, first.data(), ntok.data()))) return rc;
  if ((rc = ensure_workspace(m, m->Tp))) return rc;
  if ((rc = launch_load_hidden(st, d_hidden, d_input_quals, m->cfg.d_model, m->d_blk_read, m->d_toff, m->d_src_base,
                               m->d_ntok, m->Tp, (float*)m->x.p, (float*)m->qpad.p)))
    return rc;
  return run_head(m, st, /*head_sums=*/false, /*head1_ready=*/false, d_logits);
}

extern "C" int64_t dc_model_device_bytes(const dc_model* m) {
  if (!m) return 0;
  int64_t b = m->weight_bytes;
  b += (int64_t)m->layers.size() * m->cfg.d_model * m->kf_taps * (int64_t)sizeof(float);
  b += (int64_t)m->layers.size() * m->cfg.d_model * 3 * 8 * hyena3_units((int)m->kf_taps) * 16;
  for (const DevBuf* d : {&m->x, &m->u, &m->y, &m->hmid, &m->obuf, &m->mean, &m->rstd, &m->qpad, &m->idx, &m->part})
    b += (int64_t)d->bytes;
  return b;
}

extern "C" int dc_model_filter(dc_model* m, void* stream, int32_t layer, int32_t seq_len, float* d_out) {
  DC_REQUIRE(m && d_out, "null pointer");
  DC_REQUIRE(layer >= 0 && layer < m->cfg.n_layer, "layer out of range");
  DC_REQUIRE(seq_len >= 1, "seq_len must be positive");
  hipStream_t st = as_stream(stream);
  int rc = ensure_filters(m, st, seq_len);
  if (rc) return rc;
  DC_HIP_CHECK(hipMemcpy2DAsync(d_out, (size_t)seq_len * sizeof(float), m->layers[layer].kf,
                                (size_t)m->kf_taps * sizeof(float), (size_t)seq_len * sizeof(float), m->cfg.d_model,
                                hipMemcpyDeviceToDevice, st));
  return DC_OK;
}

extern "C" int dc_model_set_gemm_mode(dc_model* m, int32_t mode) {
  DC_REQUIRE(m, "null pointer");
  DC_REQUIRE(mode >= 0 && mode <= 2, "gemm mode must be 0 (fp32 MFMA), 1 (bf16x6) or 2 (fp16x3)");
  m->gemm_mode = mode;
  return DC_OK;
}

extern "C" int dc_model_set_mlp_fusion(dc_model* m, int32_t on) {
  DC_REQUIRE(m, "null pointer");
  m->mlp_fusion = on ? 1 : 0;
  m->mlp_chain = on >= 2 || on < 0 ? 1 : 0;  // 1: MLP only; 2 (default state): MLP + the GEMM after it
  return DC_OK;
}

extern "C" int dc_model_overflow(dc_model* m, void* stream, int32_t* count) {
  DC_REQUIRE(m && count, "null pointer");
  hipStream_t st = as_stream(stream);
  int v = 0;
  DC_HIP_CHECK(hipMemcpyAsync(&v, m->d_overflow, sizeof(int), hipMemcpyDeviceToHost, st));
  DC_HIP_CHECK(hipStreamSynchronize(st));
  if (v) DC_HIP_CHECK(hipMemsetAsync(m->d_overflow, 0, sizeof(int), st));
  *count = v;
  return DC_OK;
}

extern "C" int dc_model_set_fft_split(dc_model* m, int32_t on) {
  DC_REQUIRE(m, "null model");
  m->fft_class_mask = fft_mask_for(on != 0);
  return DC_OK;
}

extern "C" int dc_model_set_fft_threshold(dc_model* m, int32_t min_tokens) {
  DC_REQUIRE(m, "null pointer");
  DC_REQUIRE(min_tokens >= 256, "the FFT long convolution takes reads of more than min_tokens >= 256 tokens (a value above the model's context disables it)");
  m->fft_min_tokens = min_tokens;
  return DC_OK;
}

extern "C" int dc_model_set_overflow_target(dc_model* m, int32_t* d_counter) {
  DC_REQUIRE(m, "null pointer");
  m->ovf_target = d_counter;
  return DC_OK;
}

extern "C" int dc_model_get_gemm_mode(const dc_model* m, int32_t* mode) {
  DC_REQUIRE(m && mode, "null pointer");
  *mode = m->gemm_mode;
  return DC_OK;
}

extern "C" int dc_model_set_profiling(dc_model* m, int32_t enabled) {
  DC_REQUIRE(m, "null pointer");
  m->profiling = enabled != 0;
  return DC_OK;
}

extern "C" int dc_model_profile_reset(dc_model* m) {
  DC_REQUIRE(m, "null pointer");
  for (auto& e : m->prof_events) {
    m->prof_pool.push_back(e.a);
    m->prof_pool.push_back(e.b);
  }
  m->prof_events.clear();
  m->prof_acc.clear();
  return DC_OK;
}

// Fold finished event pairs into per-stage totals.  The caller must have synchronised the stream.
static int prof_collect(dc_model* m) {
  for (auto& e : m->prof_events) {
    float ms = 0.f;
    DC_HIP_CHECK(hipEventElapsedTime(&ms, e.a, e.b));
    auto& acc = m->prof_acc[m->prof_tags[e.tag]];
    acc.first += ms;
    acc.second += 1;
    m->prof_pool.push_back(e.a);
    m->prof_pool.push_back(e.b);
  }
  m->prof_events.clear();
  return DC_OK;
}

extern "C" int dc_model_profile_get(dc_model* m, const char* stage, double* total_ms, int64_t* launches) {
  DC_REQUIRE(m && stage && total_ms && launches, "null pointer");
  int rc = prof_collect(m);
  if (rc) return rc;
  auto it = m->prof_acc.find(stage);
  if (it == m->prof_acc.end()) {
    *total_ms = 0.0;
    *launches = 0;
    return DC_OK;
  }
  *total_ms = it->second.first;
  *launches = it->second.second;
  return DC_OK;
}

extern "C" int dc_model_profile_stages(dc_model* m, char* buf, int32_t buf_len) {
  DC_REQUIRE(m && buf && buf_len > 0, "bad arguments");
  int rc = prof_collect(m);
  if (rc) return rc;
  std::string s;
  for (auto& kv : m->prof_acc) {
    if (!s.empty()) s += ",";
    s += kv.first;
  }
  snprintf(buf, (size_t)buf_len, "%s", s.c_str());
  return DC_OK;
}
