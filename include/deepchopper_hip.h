/* deepchopper_hip.h -- C ABI of libdeepchopper_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the `deepchopper predict` hot path of ylab-hi/DeepChopper.  Every entry
 * point replaces one function (or one fused group) that the reference reaches through its
 * PyO3 module `deepchopper.deepchopper` (src/python.rs:879-958) or through its PyTorch model
 * (deepchopper/models).  The reference-side binding a maintainer would add is shown in
 * INTEGRATION.md.
 *
 * Conventions
 *  - plain pointers and sizes only; no torch / HIP types in signatures (`stream` is a
 *    hipStream_t passed as void*, NULL = the default stream).
 *  - pointers prefixed d_ are DEVICE pointers, h_ are HOST pointers.
 *  - every function returns 0 on success, non-zero on failure; `dc_last_error()` returns a
 *    thread-local message (the PyO3 layer turns Rust errors into Python exceptions,
 *    src/error.rs:28-53 -- the shim raises Exception(dc_last_error())).
 *  - result buffers are caller-owned; only opaque handles (`dc_model`) are library-owned and
 *    have an explicit free.
 *  - entry points are re-entrant per (device, stream); a `dc_model` handle must not be used
 *    from two threads at once.
 *  - ragged batches: read r occupies bytes/labels [offsets[r], offsets[r+1]) of a
 *    concatenated buffer (`offsets` has n_reads+1 int64 entries, on the device).
 */
#ifndef DEEPCHOPPER_HIP_H
#define DEEPCHOPPER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DC_OK 0
#define DC_ERR_INVALID_ARG 1
#define DC_ERR_HIP 2
#define DC_ERR_NO_DEVICE 3
#define DC_ERR_UNSUPPORTED 4

/* per-read status bits written by the tokenise / k-mer kernels */
#define DC_READ_OK 0
#define DC_READ_TRUNCATED 1        /* len(seq) >= max_length: tokenizer.py:154-163                     */
#define DC_READ_DROPPED_CHARS 2    /* normalize() removed whitespace: seq/qual lengths now differ     */
#define DC_READ_QUAL_UNDERFLOW 4   /* a quality byte < offset (u8 subtraction wrapped)                 */
#define DC_READ_INVALID_BASE 8     /* k-mer: a window holds a byte outside `bases` -> read dropped    */
#define DC_READ_SHORTER_THAN_K 16  /* k-mer: read shorter than k -> skipped                            */

/* token ids of the HyenaDNA character tokenizer, as pinned by src/smooth/utils.rs:6-25 */
#define DC_TOK_SEP 1
#define DC_TOK_PAD 4
#define DC_TOK_UNK 6
#define DC_IGNORE_LABEL (-100)

const char* dc_last_error(void);
int dc_version(void);
/* number of HIP devices visible (0 when there is none; never fails) */
int dc_device_count(void);
int dc_set_device(int device);

/* ------------------------------------------------------------------------------------------
 * Tokenise (T1-T5)
 * ------------------------------------------------------------------------------------------ */

/* normalize_seq(seq, iupac)                    replaces src/python.rs:272-275 (needletail
 * Sequence::normalize).  One launch for a ragged batch.  Output is written compacted per read
 * into d_out at the same offsets; d_out_len[r] receives the normalised length. */
int dc_normalize_seq(void* stream, const uint8_t* d_seq, const int64_t* d_offsets, int32_t n_reads,
                     int32_t iupac, uint8_t* d_out, int32_t* d_out_len);

/* encode_qual(qual, offset)                    replaces src/python.rs:25-35
 * phred[i] = qual[i] - offset (u8, wrapping).  n_bytes = total bytes. */
int dc_encode_qual(void* stream, const uint8_t* d_qual, int64_t n_bytes, int32_t qual_offset,
                   uint8_t* d_phred);

/* Fused per-read tokenisation = parse_fastq_file body (deepchopper/data/only_fq.py:36-72:
 * upper-case, encode_qual, normalize_seq) + HyenaDNA character tokenizer + trailing [SEP]
 * (tokenizer.py:148) + tokenize_and_align_labels_and_quals_ids (tokenizer.py:145-178: labels
 * 0 / -100, quals = phred ++ [0] L2-normalised in fp32) + the LEFT-padding collator
 * (tokenizer.py:34-93).
 *
 *  d_seq, d_qual : raw FASTQ sequence / quality bytes, concatenated; read r = [off[r], off[r+1])
 *  max_length    : tokenizer.max_len_single_sentence (reads with len >= max_length keep
 *                  max_length-1 bases and are flagged DC_READ_TRUNCATED)
 *  row_stride    : tokens per output row (>= longest tokenised read of the batch, i.e. the
 *                  batch's Lmax; rows are left-padded to it when pad_left != 0)
 *  ids_i64       : 1 -> d_input_ids is int64[n_reads,row_stride] (the .pt-compatible layout),
 *                  0 -> uint8[n_reads,row_stride] (compact, what dc_forward consumes fastest)
 *  d_labels      : int8[n_reads,row_stride] or NULL
 *  d_n_tokens    : int32[n_reads] tokens per read incl. [SEP] (or NULL)
 *  d_status      : int32[n_reads] DC_READ_* bits
 */
int dc_tokenize_batch(void* stream, const uint8_t* d_seq, const uint8_t* d_qual,
                      const int64_t* d_offsets, int32_t n_reads, int32_t max_length,
                      int32_t qual_offset, int64_t row_stride, int32_t pad_left, int32_t ids_i64,
                      void* d_input_ids, float* d_input_quals, int8_t* d_labels,
                      int32_t* d_n_tokens, int32_t* d_status);

/* ------------------------------------------------------------------------------------------
 * k-mer encoder (K1-K5)        replaces TensorEncoder::encode, src/fq_encode/tensor_encoder.rs
 * ------------------------------------------------------------------------------------------ */

/* Pass 1: per-read validity (DC_READ_INVALID_BASE / DC_READ_SHORTER_THAN_K) -- the reference drops
 * such reads (tensor_encoder.rs:144-146,242-250,294-299).  `h_bases` is a host string. */
int dc_kmer_validate(void* stream, const uint8_t* d_seq, const int64_t* d_offsets, int32_t n_reads,
                     int32_t k, const char* h_bases, int32_t n_bases, int32_t* d_status);

/* Pass 2: sliding-window encode.  d_out_row[r] = output row of read r, or -1 to skip it.
 *  d_input  : int32[n_rows,2,width]   row 0 k-mer ids, row 1 integer-mean k-mer quality, pad -1
 *  d_qual_out: int32[n_rows,max_seq_len] phred, pad -1
 *  id = sum_j idx(b_j) * n_bases^(k-1-j)  (kmer.rs:213-228); mean = (u8 wrapping sum)/k
 *  (triat.rs:102-134). */
int dc_kmer_encode(void* stream, const uint8_t* d_seq, const uint8_t* d_qual,
                   const int64_t* d_offsets, const int32_t* d_out_row, int32_t n_reads, int32_t k,
                   const char* h_bases, int32_t n_bases, int32_t qual_offset, int32_t width,
                   int32_t max_seq_len, int32_t* d_input, int32_t* d_qual_out);

/* Vectorised target rows (tensor_encoder.rs:185-203): d_target int32[n_rows,n_targets,width],
 * d_ranges int32[n_rows,n_targets,2] k-mer target regions. */
int dc_kmer_vectorize_targets(void* stream, const int32_t* d_ranges, int32_t n_rows,
                              int32_t n_targets, int32_t width, int32_t* d_target);

/* ------------------------------------------------------------------------------------------
 * Smoothing / interval path (S0-S5)
 * ------------------------------------------------------------------------------------------ */

/* Fused: argmax over 2 classes (first max wins) restricted to the read's valid positions
 * (smooth/predict.rs:264-317 strips left pad and [SEP] through target != -100), majority_voting
 * (smooth/utils.rs:48-97), get_label_region incl. its index-0 quirk (utils.rs:671-695),
 * length filter + approved-count rule (smooth/predict.rs:186-209) and the complement
 * generate_unmaped_intervals (output/split.rs:260-292).  One workgroup per read.
 *
 *  d_logits     : float[n_rows,row_stride,2]
 *  d_first,d_len: int32[n_reads] first valid column of the read's row and number of bases
 *  d_pred,d_smooth: int8 outputs, ragged at d_out_offsets (int64[n_reads+1]) or NULL
 *  d_intervals  : int32[n_reads,max_intervals,2]; d_n_intervals int32[n_reads]
 *                 (0 when the filtered count exceeds `approved`, like the reference)
 *  d_keep       : int32[n_reads,max_intervals+1,2] kept segments; d_n_keep int32[n_reads]
 *                 (either may be NULL).  max_intervals must be >= approved.
 *
 *  Contract on the lengths (they live on the device, so the host cannot check them):
 *  0 <= d_first[r], d_first[r] + d_len[r] <= row_stride, d_len[r] <= 65 536 (the kernel form --
 *  2 048 / 4 096 / 65 536 positions -- is picked from row_stride alone).  dc_tokenize_batch's
 *  outputs satisfy it.  A read that violates it is not truncated silently: the kernel writes
 *  d_n_intervals[r] = -1 (and d_n_keep[r] = -1) and nothing else for that read; the other reads of
 *  the launch are unaffected.  The same flag marks a read longer than 65 536 positions in
 *  dc_smooth_intervals_from_labels. */
int dc_smooth_intervals_from_logits(void* stream, const float* d_logits, int64_t row_stride,
                                    const int32_t* d_first, const int32_t* d_len, int32_t n_reads,
                                    const int64_t* d_out_offsets, int32_t window,
                                    int32_t min_interval, int32_t approved, int32_t max_intervals,
                                    int8_t* d_pred, int8_t* d_smooth, int32_t* d_intervals,
                                    int32_t* d_n_intervals, int32_t* d_keep, int32_t* d_n_keep);

/* Same pipeline starting from int8 labels (ragged at d_offsets): majority_voting,
 * get_label_region, smooth_label_region (src/python.rs:667-690,816).
 *  apply_approved_rule = 0 -> report every region (get_label_region): d_n_intervals is the true
 *  count, which may exceed max_intervals (only the first max_intervals are stored). */
int dc_smooth_intervals_from_labels(void* stream, const int8_t* d_labels, const int64_t* d_offsets,
                                    int32_t n_reads, int32_t window, int32_t min_interval,
                                    int32_t approved, int32_t apply_approved_rule,
                                    int32_t max_intervals, int8_t* d_smooth, int32_t* d_intervals,
                                    int32_t* d_n_intervals, int32_t* d_keep, int32_t* d_n_keep);

/* summary_predict (src/utils.rs:9-55): keep positions whose label != ignore.  Dense
 * int8[n_rows,row_stride] inputs; compacted outputs at d_out_offsets (exclusive scan of the
 * kept counts, computed by the caller from d_counts of a first call with d_out == NULL). */
int dc_summary_predict(void* stream, const int8_t* d_pred, const int8_t* d_labels, int32_t n_rows,
                       int64_t row_stride, int32_t ignore_label, const int64_t* d_out_offsets,
                       int8_t* d_out_pred, int8_t* d_out_labels, int32_t* d_counts);

/* ------------------------------------------------------------------------------------------
 * Forward (F1-F3)            replaces TokenClassificationLit.forward / predict_step
 *                            (deepchopper/models/basic_module.py:90-100,197-207)
 * ------------------------------------------------------------------------------------------ */

typedef struct dc_model dc_model;

typedef struct {
  int32_t d_model;            /* 256   (dc_hg.py:96,147)                      */
  int32_t d_inner;            /* 1024                                          */
  int32_t n_layer;            /* 4                                             */
  int32_t vocab;              /* rows of the embedding table (12 padded to 16) */
  int32_t filter_order;       /* 64                                            */
  int32_t emb_dim;            /* 5                                             */
  int32_t short_filter_order; /* 3                                             */
  int32_t max_seq_len;        /* rows of pos_emb.z / .t (32770)                */
  int32_t head_hidden;        /* 1024  (dc_hg.py:148-149)                      */
  int32_t num_class;          /* 2                                             */
  int32_t gelu_tanh;          /* 1 = tanh approximation, 0 = erf               */
  float ln_eps;               /* 1e-5                                          */
  float filter_shift;         /* 0.05                                          */
} dc_model_config;

/* HOST pointers to fp32 tensors in CHECKPOINT layout (nn.Linear weight = [out, in]). */
typedef struct {
  const float *norm1_w, *norm1_b, *norm2_w, *norm2_b; /* [D]                       */
  const float *in_proj_w, *in_proj_b;                 /* [3D, D], [3D]             */
  const float *out_proj_w, *out_proj_b;               /* [D, D], [D]               */
  const float *short_w, *short_b;                     /* [3D, 1, 3], [3D]          */
  const float *filt_bias;                             /* [D]   filter_fn.bias      */
  const float *pos_z, *pos_t;                         /* [max_seq_len, emb], [max_seq_len] */
  const float *filt_w0, *filt_b0;                     /* [fo, emb], [fo]           */
  const float *filt_w1, *filt_b1;                     /* [fo, fo], [fo]            */
  const float *filt_w2, *filt_b2;                     /* [fo, fo], [fo]            */
  const float *filt_w3;                               /* [D, fo]  (no bias)        */
  const float *filt_freq0, *filt_freq1, *filt_freq2;  /* [fo] each                 */
  const float *deltas;                                /* [D]                       */
  const float *fc1_w, *fc1_b, *fc2_w, *fc2_b;         /* [DI, D],[DI],[D, DI],[D]  */
} dc_layer_weights;

typedef struct {
  const float* embedding;        /* [vocab, D]                 */
  const dc_layer_weights* layers; /* n_layer entries            */
  const float *ln_f_w, *ln_f_b;  /* [D]                        */
  const float *head_w1, *head_b1; /* [H, D], [H]               */
  const float *head_w2, *head_b2; /* [H, H], [H]               */
  const float *head_w3, *head_b3; /* [C, H], [C]               */
} dc_model_weights;

/* DeepChopper.from_pretrained / from_checkpoint (dc_hg.py:70-163): upload + repack weights on
 * the current device.  The implicit long-convolution filters are input independent and are
 * materialised once per handle, lazily, up to the longest read seen. */
int dc_model_create(const dc_model_config* cfg, const dc_model_weights* w, dc_model** out);
void dc_model_free(dc_model* m);

/* logits[b, l, :] = head(backbone(input_ids)[b, l], input_quals[b, l])  for a dense batch.
 *  d_input_ids : int64[B, L] (ids_i64 = 1) or uint8[B, L] (ids_i64 = 0); left pad ids are
 *                ordinary tokens to the model, exactly as in the reference (no attention mask)
 *  d_logits    : float[B, L, num_class] */
int dc_forward(dc_model* m, void* stream, const void* d_input_ids, int32_t ids_i64,
               const float* d_input_quals, int32_t batch, int32_t seq_len, float* d_logits);

/* Ragged variant: read r has n_tokens[r] tokens starting at row r, column first[r] of the dense
 * [n_reads,row_stride] id / quality arrays; only those tokens are run through the model (the
 * native zero-padding mode == the reference at --batch-size 1).  Logits are written to the
 * same dense positions of d_logits float[n_reads,row_stride,num_class]; other positions are
 * left untouched.  h_first / h_n_tokens are HOST arrays. */
int dc_forward_ragged(dc_model* m, void* stream, const void* d_input_ids, int32_t ids_i64,
                      const float* d_input_quals, int32_t n_reads, int64_t row_stride,
                      const int32_t* h_first, const int32_t* h_n_tokens, float* d_logits);

/* Reference batches without the cost of their padding.  The reference's collator left-pads every batch to its longest
 * read with [PAD] (deepchopper/models/llm/tokenizer.py:71-87, `deepchopper predict -b 12`, deepchopper/cli.py:70) and the pad
 * tokens run through the model like any other token (no attention mask), so a read's logits depend on how
 * many pads stand in front of it -- dc_forward / dc_forward_ragged reproduce that by running every row at its padded
 * length.  HyenaDNA is causal: what the model computes over a run of [PAD] at the head of a row is the same for every
 * row.  This entry point computes it ONCE per launch: arguments as dc_forward_ragged, plus h_n_pad[r] = how many of row
 * r's n_tokens[r] tokens, counted from first[r], are left padding -- the caller asserts that those positions hold
 * (pad_token_id, quality 0.0), as the collator writes them.  Per launch one all-[PAD] row per long-convolution form in use
 * stands for every row's pad run (rounded down to a multiple of 4; a row keeps the last <= 3 pads itself): embedding,
 * every GEMM, the fused MLP and the head see sum(n_tokens - n_pad) + a few shared rows instead of sum(n_tokens) tokens;
 * the long-convolution kernels read a row's positions in front of its own tokens from the shared row.
 * Results are those of dc_forward_ragged(first, n_tokens) on the same buffers: bit for bit while every row takes a
 * block-Toeplitz long convolution (<= fft threshold tokens), within fp32 rounding of the FFT forms above (whose circular
 * transform lets later tokens reach the pad positions as rounding noise, in the reference too).  Logits are written
 * for all n_tokens[r] positions of a row, pad columns included (copied from the shared row).
 * The pad positions themselves are never read by the model; the copy pass checks them and counts violations of the
 * contract (dc_model_prefix_violations).  Requires 0 <= h_n_pad[r] < h_n_tokens[r].  Host arrays as in dc_forward_ragged. */
int dc_forward_shared_prefix(dc_model* m, void* stream, const void* d_input_ids, int32_t ids_i64,
                             const float* d_input_quals, int32_t n_reads, int64_t row_stride,
                             const int32_t* h_first, const int32_t* h_n_tokens, const int32_t* h_n_pad,
                             int32_t pad_token_id, float* d_logits);

/* How dc_forward_shared_prefix shares the pad runs.
 *   1 (default) exact form: as described above -- one shared row per long-convolution form; the long convolution of a
 *     row still walks its whole padded sequence (reading the pad positions from the shared row).  Bit-identical to
 *     dc_forward_ragged on the padded rows wherever every row takes a block-Toeplitz form.
 *   2 linear form: the long convolution is linear, so with ONE shared row (as long as the longest padded row) giving
 *     z_pad (its gated input) and F = k * z_pad, a row's output is ((k * (z - z_pad)) + F) * gate and z - z_pad is zero
 *     over the row's whole pad run: the long convolution starts 4 positions in front of the row's own tokens and costs
 *     what it costs without padding (O(n^2) / O(n log n) in the READ's length, not the batch's longest).  Same
 *     mathematics, another summation order: logits agree with the exact form to fp32 rounding (~1e-5 on O(1) logits;
 *     tests hold it to 2e-4), not bit for bit.
 * $DC_PREFIX_MODE = exact | linear sets the initial mode of a handle.  Who defaults to what: a handle created through this ABI
 * starts in mode 1; the Python pipeline (deepchopper_amd/predict.py: `predict`, `predict --chop --reference-batching`, the
 * `.pt` writer) switches its handle to mode 2 unless $DC_PREFIX_MODE says exact -- measured over a 32 600-token pad run and
 * 20 seeds of nanopore-like batches, mode 2 is the closer of the two to a float64 evaluation (<= 1.0e-4 against <= 1.2e-4 for
 * the padded launch; they differ from each other by <= 1.5e-4: tests/test_gpu_forward.py::test_linear_prefix_form_*). */
int dc_model_set_prefix_mode(dc_model* m, int32_t mode);

/* Wavefronts of dc_forward_shared_prefix's copy pass that met a declared pad position not holding
 * (pad_token_id, 0.0) since the last call; synchronises `stream`, resets the counter.  Non-zero = the caller broke the
 * contract and the logits of that launch are not those of the padded batch. */
int dc_model_prefix_violations(dc_model* m, void* stream, int32_t* count);

/* Diagnostic: the tail of the forward alone -- final LayerNorm + TokenClassificationHead.forward
 * (deepchopper/models/llm/head.py:75-102) -- on a caller-supplied residual stream
 * d_hidden float[batch, seq_len, d_model] (the backbone's last hidden state BEFORE ln_f, token-major as the
 * reference holds it) and d_input_quals float[batch, seq_len]; logits float[batch, seq_len, num_class].
 * Runs the same kernels as dc_forward (LayerNorm statistics, linear1 + ReLU + quality, linear2 + residual +
 * ReLU fused with linear3) in the GEMM mode in force; lets the head be checked directly against vectors
 * produced by the reference's own head.py (tests/golden/make_head_golden.py). */
int dc_head_forward(dc_model* m, void* stream, const float* d_hidden, const float* d_input_quals, int32_t batch,
                    int32_t seq_len, float* d_logits);

/* Device bytes currently held by the handle (weights + filters + workspace). */
int64_t dc_model_device_bytes(const dc_model* m);

/* Materialise the long-convolution filter of layer `layer` for `seq_len` taps into
 * d_out float[D, seq_len] (filter + skip folded in: k'[0] += filt_bias).  For tests/diagnostics. */
int dc_model_filter(dc_model* m, void* stream, int32_t layer, int32_t seq_len, float* d_out);

/* Arithmetic of the forward GEMMs.  0 = fp32-input MFMA (v_mfma_f32_32x32x2_f32, an exact fp32 FMA
 * chain).  1 = "bf16x6": every fp32 operand is split error-free into three bf16 terms
 * (x = h + m + l, 24 significant bits) and the six leading cross products run on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- fp32-equivalent accuracy (dropped terms
 * <= 2^-24 relative) at 6/16 of the fp32-MFMA cost.  2 = "fp16x3": x = h + l in two fp16 terms (22 bits),
 * three cross products on v_mfma_f32_32x32x16_f16, weights pre-scaled by a power of two per tensor; guarded by
 * dc_model_overflow.  Default: 2; $DC_GEMM_MODE = "f32" | "bf16x6" | "fp16x3". */
int dc_model_set_gemm_mode(dc_model* m, int32_t mode);

/* fp16x3 only: kernel fusion level of the MLP half of each block (HyenaDNA block `mlp` + residual, see
 * oracle/forward_ref.py backbone_forward).
 *   0  fc1 and fc2 as two GEMM launches;
 *   1  fc1 -> GELU -> fc2 -> +residual -> LayerNorm statistics as ONE kernel that keeps the d_inner-wide
 *      activation in registers;
 *   2  (default) that kernel also runs the GEMM that consumes LayerNorm(its output): the next block's in_proj,
 *      or the head's linear1 (+ReLU + quality) after the last block -- the new residual stream is that GEMM's
 *      B operand straight from the accumulator registers.
 * Needs d_model 256, d_inner a multiple of 128 and <= 1024 (head_hidden a multiple of 128 for the head), otherwise
 * the plain GEMM launches run.  Same arithmetic at every level
 * (tests/test_gpu_forward.py::test_mlp_fusion_matches_two_kernel_form). */
int dc_model_set_mlp_fusion(dc_model* m, int32_t level);

/* Long convolution of the Hyena operator (the reference's `fftconv`, HF-Hub code called at
 * deepchopper/models/llm/hyena.py:34-41).  Reads of up to `min_tokens` tokens run the block-Toeplitz MFMA kernel
 * (O(L^2), measured faster below ~2.5k tokens); longer reads (up to 16 384 tokens) run an in-LDS radix-16 FFT
 * convolution (O(L log L), fp32 like the reference; padded sizes 1 024 .. 16 384 complex points, see
 * dc_model_set_fft_split).  Default 2560
 * ($DC_FFT_MIN): inside the 2^12 bucket the FFT costs the same for every read while the Toeplitz cost grows with
 * L^2, they cross near 2 550 tokens; min_tokens >= 256 (DC_ERR_ARG below); a value above the model's context length
 * disables the FFT path.  Same results within the forward's tolerance either way
 * (tests/test_gpu_forward.py::test_fft_long_convolution_matches_toeplitz_and_oracle). */
int dc_model_set_fft_threshold(dc_model* m, int32_t min_tokens);

/* Size classes of that FFT convolution.  on != 0 (default; $DC_FFT_SPLIT=0 for the other): transforms above 4 096 points are
 * split once, M = Q x 4096 (Q = 2, 3, 4 rows in LDS, three radix-16 passes each instead of four; the zero half of the packed
 * input and the unused half of the output are never computed): classes of 1 024, 2 048, 4 096, 8 192, 12 288, 16 384 points.
 * on == 0: the power-of-two transforms of rounds 2-4 (1 024 .. 16 384 points, four passes for 2^13 / 2^14).  Same results to
 * fp32 rounding (tests/test_gpu_forward.py::test_split_fft_classes_match_the_power_of_two_transforms). */
int dc_model_set_fft_split(dc_model* m, int32_t on);

/* The predict -> chop wire format, device half (the reference: CustomWriter.write_on_batch_end = torch.save of
 * {prediction f32[B,L,2], target i64[B,L], seq i64[B,L], qual f32[B,L], id i64[B,256]} per batch,
 * deepchopper/models/callbacks.py:12-25).  The reference batches of a fused launch lie in the launch's dense
 * [n_rows][row_stride] arrays (logits f32 x 2, labels i8, token ids u8 or i64, qualities f32), batch i in rows
 * [row0, row0 + B) from column c0 on (its own left padding starts there; L = row_stride - c0).  dc_pt_pack writes, for
 * every file i, the storages `data/0..3` exactly as the `.pt` file holds them -- prediction f32[B][L][2] | target i64[B][L] |
 * seq i64[B][L] | qual f32[B][L], 28 B per position, ids and labels widened here -- back to back at d_out + offset_i.
 *   file table  int64[n_files][4] = {row0, B, c0, offset_i (bytes, a multiple of 64)}: the HOST copy is validated
 *               (rows / columns inside the launch, payload inside out_bytes, < 2^31 positions per file), the DEVICE copy
 *               is what the kernel reads.
 * One launch on `stream`; d_out then goes to pinned host memory in one copy and into the files without another pass over
 * it (include/deepchopper_io.h: dc_pt_submit_packed).  On a nanopore-like length mix the dense arrays are four times
 * what the files hold; cutting the batches out on the host moved 12.6 GB over PCIe for 5.8 GB of files. */
int dc_pt_pack(void* stream, const float* d_logits, const int8_t* d_labels, const void* d_ids, int32_t ids_i64,
               const float* d_quals, int64_t n_rows, int64_t row_stride, int32_t n_files, const int64_t* h_file_table,
               const int64_t* d_file_table, void* d_out, int64_t out_bytes);

/* fp16x3 range guard.  Mode 2 of dc_model_set_gemm_mode ("fp16x3") splits operands into two fp16 terms
 * (22 significant bits, three MFMAs); an activation whose magnitude exceeds 65504 cannot be represented
 * and bumps a device counter.  Synchronises `stream`, returns the number of workgroups that saw such a
 * value since the last call and clears the counter; callers re-run the batch in mode 1 when it is non-zero
 * (deepchopper_amd/predict.py does). */
int dc_model_overflow(dc_model* m, void* stream, int32_t* count);

/* Per-batch attribution of the range guard for pipelined callers.  While a non-NULL caller-owned device
 * counter is set, every forward launched afterwards bumps THAT int32 (the caller zeroes it, and reads it
 * back once the batch has finished -- e.g. with the batch's other results on a copy stream) instead of
 * the handle's own counter, so a batch launched while an earlier one is still being collected cannot
 * have its overflow consumed, or cleared, by the earlier batch's check.  NULL restores the handle's
 * counter (dc_model_overflow). */
int dc_model_set_overflow_target(dc_model* m, int32_t* d_counter);

/* The GEMM mode in force (0 / 1 / 2 as dc_model_set_gemm_mode): the one place $DC_GEMM_MODE is parsed is
 * dc_model_create; bindings read the result here instead of parsing the variable again. */
int dc_model_get_gemm_mode(const dc_model* m, int32_t* mode);

/* Per-stage device timing with HIP events recorded on the stream the kernels are launched on.
 * While enabled, every kernel launch of dc_forward / dc_forward_ragged is bracketed by an event
 * pair.  After synchronising the stream, dc_model_profile_get returns the summed milliseconds and
 * the number of launches of one stage since the last reset.  Stage names: "forward" (whole call),
 * "embed", "ln_stats", "gemm_inproj", "hyena", "gemm_outproj", "gemm_fc1", "gemm_fc2" (or "mlp_fused" / "mlp_inproj_fused" / "mlp_head1_fused"),
 * "gemm_head1", "gemm_head2", "head_out".  bench.py builds its `roofline` object from these. */
int dc_model_set_profiling(dc_model* m, int32_t enabled);
int dc_model_profile_reset(dc_model* m);
int dc_model_profile_get(dc_model* m, const char* stage, double* total_ms, int64_t* launches);
/* comma-separated list of the stages recorded so far */
int dc_model_profile_stages(dc_model* m, char* buf, int32_t buf_len);

#ifdef __cplusplus
}
#endif
#endif /* DEEPCHOPPER_HIP_H */
