// Argument blocks shared between forward_kernels.hip (device code + launchers) and
// forward_api.hip (model handle, weight repacking, orchestration).
#pragma once
#include "dc_common.h"

namespace dc {

enum GemmEpilogue {
  EPI_BIAS = 0,       // C = acc + bias                                   (in_proj)
  EPI_BIAS_RES,       // C = acc + bias + R                               (out_proj, fc2: residual stream)
  EPI_BIAS_GELU,      // C = gelu(acc + bias)                             (fc1)
  EPI_BIAS_RELU_Q,    // C = relu(acc + bias) + q[t]                      (head linear1 + quality residual)
  EPI_BIAS_RES_RELU,  // C = relu(acc + bias + R)                         (head linear2 + residual)
  EPI_HEAD2_FUSED,    // o = relu(acc + bias + R) is not stored: part[nb][t][:] = W3[:, n-block] . o   (linear3 fused)
  EPI_BIAS_RES_STATS, // EPI_BIAS_RES + per-token sum / sum of squares of the result added into st_sum / st_sq
                      // (LayerNorm statistics of the next layer fused into the producer; two channel blocks per
                      // token -> two commutative atomic addends -> deterministic)
};

struct GemmArgs {
  const float* Wt;    // [K][N]  pre-transposed (and LayerNorm-gamma folded) weights (fp32 MFMA path)
  const unsigned short* Wp;  // split planes of Wt, [NP][K/8][N][8] 16-bit: bf16 h|m|l (split_mode 0) or fp16 h|l of
                             // Wt * w_scale (split_mode 1); nullptr selects the fp32-MFMA kernel
  int split_mode;            // 0 = bf16x6, 1 = fp16x3
  float w_inv_scale;         // fp16x3: 1 / w_scale (a power of two), applied to the accumulator in the epilogue
  int* overflow;             // fp16x3: device counter bumped when an activation is outside the fp16 range
  const float* X;     // [K][Tp]
  float* C;           // [N][Tp]
  const float* bias;  // [N]     (LayerNorm-beta folded)
  const float* R;     // [N][Tp] residual or nullptr
  const float* q;     // [Tp]    quality or nullptr
  const float* mean;  // [Tp]    LayerNorm statistics of X's tokens (LN variants)
  const float* rstd;  // [Tp]
  float* st_sum;      // [Tp] out: sum over channels of the result (EPI_BIAS_RES_STATS), zeroed by the caller
  float* st_sq;       // [Tp] out: sum of squares
  int stats_are_sums; // LN prologue: mean[] / rstd[] hold sum / sum of squares over K channels (fused statistics)
  float ln_eps;
  const float* w3;    // [2][N]  head linear3 weight (EPI_HEAD2_FUSED)
  float* part;        // [N/128][Tp][2] partial logits per 128-channel block (EPI_HEAD2_FUSED)
  int K, N;
  int64_t Tp;
  int gelu_tanh;
  // fused MLP (launch_mlp_fused): second matrix = fc2, split planes with the k order permuted inside every
  // aligned group of 16 hidden units so that fc1's accumulator registers are fc2's B fragments as they stand
  const unsigned short* Wp2;  // [2][N/8][K][8] fp16 h|l of fc2^T * w_scale2 (N hidden -> K outputs)
  const float* bias2;         // [K]
  float w_inv_scale2;
  int stats_store;            // EPI_BIAS_RES_STATS: 1 = this workgroup owns all channels of its tokens: plain stores
  // fused MLP, optional third matrix: the GEMM that consumes LayerNorm(new residual stream) -- the next block's
  // in_proj (epi3 = EPI_BIAS) or the head's linear1 (epi3 = EPI_BIAS_RELU_Q, with q) -- K -> N3 outputs, k order
  // permuted like Wp2 (the new residual stream is its B operand straight from the accumulator registers)
  const unsigned short* Wp3;  // [2][K/8][N3][8] or nullptr
  const float* bias3;         // [N3] (LayerNorm-beta folded)
  float w_inv_scale3;
  int N3, epi3;
  float* C3;                  // [N3][Tp]
};

struct FilterArgs {
  const float *pos_z, *pos_t;
  const float *w0, *b0, *w1, *b1, *w2, *b2, *w3;
  const float *freq0, *freq1, *freq2;
  const float *deltas, *bias;
  int fo, emb, D, max_seq_len, n_taps;
  float shift;
  float* out;          // [D][out_stride]
  int64_t out_stride;
};

struct HyenaArgs {
  const float* u;      // [3D][Tp] in_proj output (x0 | x1 | v)
  float* y;            // [D][Tp]
  const float* kf;     // [D][kf_stride] long filter of this layer (skip folded into tap 0)
  int64_t kf_stride;
  const float* sw;     // [3D][3] short filter
  const float* sb;     // [3D]
  const int32_t* grp_reads;  // [n_groups][32] read index or -1
  const int32_t* grp_P;      // [n_groups] max 32-token blocks over the group's reads
  const int32_t* toff;       // [n_reads] first token column
  const int32_t* nblk;       // [n_reads] 32-token blocks
  const int32_t* ntok;       // [n_reads] tokens (FFT long convolution: exact read length)
  int64_t Tp;
  int D;
  const unsigned short* kfp;  // hyena3: bf16 filter planes [D][3][8][kfp_nu][8] (reversed, 8 shifted copies)
  int kfp_nu;          // 16-byte units per (plane, shift) copy
  int kf_taps;         // taps of the materialised filter (multiple of 1024)
  const unsigned short* kfp_f16;  // fp16x3: 2 fp16 planes of the per-channel scaled filter, same unit layout
  const float* ch_inv_scale;      // fp16x3: [D] 1 / (power-of-two scale of the channel's filter)
  int split_mode;                 // 0 = bf16x6, 1 = fp16x3
  int* overflow;                  // fp16x3 range guard counter
  // layer 0 (hyena3 only): u is a function of the token id alone -- gather it from the per-id table instead
  // of reading a materialised [3D][Tp] buffer (the embed kernel then does not write one)
  const unsigned char* tok;       // [Tp] token id per packed slot, 255 = no token; nullptr = read `u`
  const float* tab;               // [vocab][3D] in_proj(LayerNorm(emb[id])) (x0 | x1 | v)
  int vocab;                      // <= 16
  // Shared [PAD] prefix (dc_forward_shared_prefix; nullptr = every read owns all of its positions).  A row of a left-padded
  // reference batch is `[PAD] x p ++ read`; HyenaDNA is causal, so what the model holds over the pad run does not depend on
  // the row and ONE all-[PAD] row per launch carries it.  A read's positions t < pv[r] (a multiple of 4: the 16-byte accesses
  // below never straddle) live in that row's slots, padoff[r] + t; positions t >= pv[r] in its own slots, toff[r] + t
  // (toff[r] = first own slot - pv[r]).  Only own positions are stored.
  const int32_t* pv;              // [n_reads] positions in front of the read's own slots
  const int32_t* padoff;          // [n_reads] slot of the shared row that stands for the read's position 0
  // Linear form of the shared prefix (dbase != nullptr): the convolution is linear, so with z_pad / F = k * z_pad of the ONE
  // shared row known, a read's output is  y[t] = ((k * (z - z_pad))[t] + F[t]) * gate[t]  and z - z_pad vanishes over the whole
  // pad run: the read's sequence starts 4 positions in front of its own slots (pv = 4: the short filter's halo) instead of
  // at the head of its pad run -- the long convolution costs what it costs without padding.
  const int32_t* dbase;           // [n_reads] index of the read's position 0 in the rows of zc / Fc (a multiple of 4)
  const float* zc;                // [D][dstride] gated input z of the shared row (0 in front of it: reads without a pad run)
  const float* Fc;                // [D][dstride] its long convolution k * z_pad, before the output gate
  int64_t dstride;
};

int launch_gemm(hipStream_t st, const GemmArgs& g, int epi, bool ln);
// x <- x + fc2(gelu(fc1(LN(x)))) in one kernel (fp16x3 only): g.Wp / bias / w_inv_scale = fc1 (K = D -> N = d_inner),
// g.Wp2 / bias2 / w_inv_scale2 = fc2 (permuted planes), X = R = C = the residual stream, mean / rstd = sum / sum of
// squares in, st_sum / st_sq = the new statistics out (same buffers, overwritten)
int launch_mlp_fused(hipStream_t st, const GemmArgs& g);
int mlp_fused_perm16(int kappa);
int mlp_fused_perm(int k);       // k order of the fused MLP kernel's register-fed GEMMs: real row that logical k stands for
int gemm6_channel_block();  // output channels per block of the bf16x6 kernel (partials of EPI_HEAD2_FUSED)
int launch_filter(hipStream_t st, const FilterArgs& f);
int launch_hyena2(hipStream_t st, const HyenaArgs& a, int n_groups, int max_P);
int launch_hyena3(hipStream_t st, const HyenaArgs& a, int n_groups, int max_P);
// O(L log L) long convolution for long reads (hyena_fft.hip), one workgroup per (read, channel).  Size classes, ordered by
// capacity: `points` = M complex points of the transform = the longest sequence (tokens) the class takes.
//   q == 1: one in-LDS radix-16 transform of 2^logn points
//   q >= 2: split transform, q rows of 2^logn points (three radix-16 passes each; the zero half of the input and the unused
//           half of the output are never computed; 3 x 4096 points for 8 193 .. 12 288 tokens)
//   q == 0: the four-step 2^15-point form (two kernels, launch_hyena_fft15)
// Classes 4 and 7 (direct 2^13 / 2^14: four passes) are the round 2-4 kernels for the sizes the split forms 3 and 6 now serve:
// off by default, kept as the A/B reference (dc_model_set_fft_split(m, 0) / DC_FFT_SPLIT=0 restores the power-of-two classes).
struct FftClassInfo {
  int points, q, logn;
};
constexpr int kFftClasses = 9;
constexpr int kFft15Class = kFftClasses - 1;
inline constexpr FftClassInfo kFftClass[kFftClasses] = {{1024, 1, 10},  {2048, 1, 11},  {4096, 1, 12},  {8192, 2, 12}, {8192, 1, 13},
                                                        {12288, 3, 12}, {16384, 4, 12}, {16384, 1, 14}, {32768, 0, 15}};
int fft_twiddle_count();                           // float2 entries of the twiddle tables (one buffer)
int launch_fft_twiddles(hipStream_t st, float* tw);
int64_t hyena_fft_spec_floats(int cls, int D);     // floats of a size class's spectrum buffer (spectra + the split kernel's pair maps)
int launch_hyena_fft_filter(hipStream_t st, int cls, const float* kf, int64_t kf_stride, int kf_taps, int D,
                            const float* tw, float* spec);  // spec: [D][points + 1] float2 (+ pair maps, hyena_fft_spec_floats)
int launch_hyena_fft(hipStream_t st, const HyenaArgs& a, int cls, const int32_t* reads, int n_reads, const float* tw,
                     const float* spec);
// reads of 16 385 .. 32 769 tokens: 2^15 complex points as 16 x 2048 (two kernels, one scratch round trip of
// hyena_fft15_scratch_bytes(n_reads, D)); the filter spectrum comes from launch_hyena_fft_filter(.., kFft15Class, ..)
int64_t hyena_fft15_scratch_bytes(int n_reads, int D);
int launch_hyena_fft15(hipStream_t st, const HyenaArgs& a, const int32_t* reads, int n_reads, const float* tw, const float* spec,
                       float* scratch);
int hyena3_units(int kf_taps);  // 16-byte units per (plane, shift) copy of the bf16 filter planes
int launch_filter_scale(hipStream_t st, const float* kf, int D, int kf_taps, float* ch_scale, float* ch_inv_scale);
int launch_filter_planes(hipStream_t st, const float* kf, int D, int kf_taps, unsigned short* kfp_bf16,
                         unsigned short* kfp_f16, const float* ch_scale);
int launch_embed(hipStream_t st, const void* ids, int ids_i64, const float* quals, const float* emb, int vocab, int D,
                 const int32_t* blk_read, const int32_t* toff, const int64_t* src_base, const int32_t* ntok,
                 int64_t Tp, float* x, float* qpad, int pad_id);  // src_base[r] < 0: a shared [PAD] row (token pad_id, quality 0)
int launch_load_hidden(hipStream_t st, const float* hidden, const float* quals, int D, const int32_t* blk_read,
                       const int32_t* toff, const int64_t* src_base, const int32_t* ntok, int64_t Tp, float* x,
                       float* qpad);
int launch_ln_stats(hipStream_t st, const float* x, int D, int64_t Tp, float eps, float* mean, float* rstd);
// linear form of the shared [PAD] prefix: y slots of the shared row hold F = k * z_pad (unit-gate pass) -> zc / Fc rows at z0 + t, y = F * gate
// (tab != nullptr: layer 0, u is the table row of pad_id)
int launch_pad_row_finish(hipStream_t st, const float* u, const float* tab, int pad_id, int D, int64_t Tp, int pad_toff, int L,
                          const float* sw, const float* sb, float* y, float* zc, float* Fc, int64_t dstride, int z0);
int launch_head_reduce(hipStream_t st, const float* part, int n_part, const float* b3, int64_t Tp,
                       const int32_t* blk_read, const int32_t* toff, const int64_t* src_base, const int32_t* ntok,
                       float* logits, float* padlog);  // src_base[r] < 0: logits of a shared [PAD] row -> padlog + (-1 - src_base[r])
int launch_embed_l0(hipStream_t st, const void* ids, int ids_i64, const float* quals, const float* emb,
                    const float* u_table, int vocab, int D, const int32_t* blk_read, const int32_t* toff,
                    const int64_t* src_base, const int32_t* ntok, int64_t Tp, float* x, float* u, float* qpad,
                    unsigned char* tok, int pad_id);  // u == nullptr: only x, qpad and the packed token ids `tok` are written
int launch_head_out(hipStream_t st, const float* o, const float* w3, const float* b3, int H, int C, int64_t Tp,
                    const int32_t* blk_read, const int32_t* toff, const int64_t* src_base, const int32_t* ntok,
                    float* logits, float* padlog);
// dc_forward_shared_prefix: row j of the fill list gets logits[fill_dst[j] + i] = padlog[fill_src[j] + i], i < fill_n[j]; `bad` counts
// wavefronts that met a position not holding (pad_id, quality 0)
int launch_fill_pad_logits(hipStream_t st, const void* ids, int ids_i64, const float* quals, int n_rows, int max_run,
                           const int64_t* fill_dst, const int32_t* fill_n, const int32_t* fill_src, const float* padlog,
                           int pad_id, float* logits, int* bad);

}  // namespace dc
