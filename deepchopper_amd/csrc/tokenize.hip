// Tokenise kernels (T1-T5): byte work, HBM-bound.  One 256-thread workgroup per read; lanes walk
// the read's bytes with unit stride so every wave access is one contiguous 64-byte segment.
#include "dc_common.h"

namespace dc {

// needletail 0.7.3 Sequence::normalize (called at src/python.rs:272-275).  0 = drop the byte.
__device__ __forceinline__ uint8_t normalize_char(uint8_t c, bool iupac) {
  switch (c) {
    case 'A': case 'C': case 'G': case 'T': case 'N': case '-': return c;
    case 'a': return 'A';
    case 'c': return 'C';
    case 'g': return 'G';
    case 't': case 'u': case 'U': return 'T';
    case '.': case '~': return '-';
    case ' ': case '\t': case '\r': case '\n': return 0;
    default: break;
  }
  if (iupac) {
    uint8_t u = (c >= 'a' && c <= 'z') ? uint8_t(c - 32) : c;
    switch (u) {
      case 'B': case 'D': case 'H': case 'V': case 'R': case 'Y': case 'S': case 'W': case 'K': case 'M':
        return u;
      default: break;
    }
  }
  return 'N';
}

// pyfastx uppercase=True (only_fq.py:34) -> normalize_seq(.., false) -> character tokenizer
// ids pinned by src/smooth/utils.rs:6-25.  0xFF = byte removed by normalize.
__device__ __forceinline__ uint8_t token_of_raw(uint8_t c) {
  uint8_t u = (c >= 'a' && c <= 'z') ? uint8_t(c - 32) : c;
  uint8_t n = normalize_char(u, false);
  switch (n) {
    case 0: return 0xFF;
    case 'A': return 7;
    case 'C': return 8;
    case 'G': return 9;
    case 'T': return 10;
    case 'N': return 11;
    default: return DC_TOK_UNK;  // '-' is not in the vocabulary
  }
}

__global__ __launch_bounds__(256) void normalize_seq_kernel(const uint8_t* __restrict__ seq,
                                                            const int64_t* __restrict__ offsets, int n_reads,
                                                            int iupac, uint8_t* __restrict__ out,
                                                            int32_t* __restrict__ out_len) {
  __shared__ unsigned scratch[8];
  for (int r = blockIdx.x; r < n_reads; r += gridDim.x) {
    const int64_t beg = offsets[r];
    const int64_t len = offsets[r + 1] - beg;
    unsigned written = 0;
    for (int64_t base = 0; base < len; base += 256) {
      const int64_t i = base + threadIdx.x;
      uint8_t c = 0;
      if (i < len) c = normalize_char(seq[beg + i], iupac != 0);
      unsigned keep = c != 0;
      unsigned tot;
      unsigned pos = block_exclusive_scan_256(keep, scratch, &tot);
      if (keep) out[beg + written + pos] = c;
      written += tot;
    }
    if (threadIdx.x == 0) out_len[r] = (int32_t)written;
  }
}

__global__ void encode_qual_kernel(const uint8_t* __restrict__ q, int64_t n, uint8_t off,
                                   uint8_t* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = uint8_t(q[i] - off);
}

template <typename IdT>
__global__ __launch_bounds__(256) void tokenize_batch_kernel(
    const uint8_t* __restrict__ seq, const uint8_t* __restrict__ qual, const int64_t* __restrict__ offsets,
    int n_reads, int max_length, uint8_t qual_offset, int64_t row_stride, int pad_left,
    IdT* __restrict__ ids, float* __restrict__ quals, int8_t* __restrict__ labels,
    int32_t* __restrict__ n_tokens, int32_t* __restrict__ status) {
  __shared__ unsigned long long s_sumsq[4];
  __shared__ unsigned s_flags;
  for (int r = blockIdx.x; r < n_reads; r += gridDim.x) {
    const int64_t beg = offsets[r];
    const int64_t raw_len = offsets[r + 1] - beg;
    int st = 0;
    // tokenizer.py:154-163: len(seq) >= max_length keeps max_length-1 characters.
    int64_t len = raw_len;
    if (raw_len >= max_length) {
      len = max_length - 1;
      st |= DC_READ_TRUNCATED;
    }
    const int64_t ntok = len + 1;
    if (threadIdx.x == 0) s_flags = 0;
    __syncthreads();

    // pass 1: exact integer sum of squares of the phred values (quals = phred ++ [0]).
    unsigned long long acc = 0;
    unsigned flags = 0;
    for (int64_t i = threadIdx.x; i < len; i += 256) {
      const uint8_t qb = qual[beg + i];
      if (qb < qual_offset) flags |= DC_READ_QUAL_UNDERFLOW;
      const unsigned p = uint8_t(qb - qual_offset);
      acc += (unsigned long long)(p * p);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) s_sumsq[threadIdx.x >> 6] = acc;
    if (flags) atomicOr(&s_flags, flags);
    __syncthreads();
    const unsigned long long sumsq = s_sumsq[0] + s_sumsq[1] + s_sumsq[2] + s_sumsq[3];
    // F.normalize(dim=0): x / max(||x||_2, 1e-12), fp32 (tokenizer.py:166).
    const float nrm = fmaxf(sqrtf((float)sumsq), 1e-12f);  // correctly rounded (hipcc default)

    const int64_t pad = pad_left ? (row_stride - ntok) : 0;
    IdT* id_row = ids + (int64_t)r * row_stride;
    float* q_row = quals + (int64_t)r * row_stride;
    int8_t* l_row = labels ? labels + (int64_t)r * row_stride : nullptr;

    // pass 2: write the whole row: pad | tokens | [SEP] (| pad on the right when !pad_left)
    for (int64_t c = threadIdx.x; c < row_stride; c += 256) {
      const int64_t i = c - pad;
      IdT tok = (IdT)DC_TOK_PAD;
      float qv = 0.0f;
      int8_t lab = (int8_t)DC_IGNORE_LABEL;
      if (i >= 0 && i < len) {
        const uint8_t t = token_of_raw(seq[beg + i]);
        if (t == 0xFF) {
          flags |= DC_READ_DROPPED_CHARS;
          tok = (IdT)11;
        } else {
          tok = (IdT)t;
        }
        const float p = (float)uint8_t(qual[beg + i] - qual_offset);
        qv = p / nrm;  // IEEE division (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt)
        lab = 0;
      } else if (i == len) {
        tok = (IdT)DC_TOK_SEP;
      }
      id_row[c] = tok;
      q_row[c] = qv;
      if (l_row) l_row[c] = lab;
    }
    if (flags) atomicOr(&s_flags, flags);
    __syncthreads();
    if (threadIdx.x == 0) {
      if (n_tokens) n_tokens[r] = (int32_t)ntok;
      status[r] = st | (int)s_flags;
    }
    __syncthreads();
  }
}

}  // namespace dc

using namespace dc;

extern "C" int dc_normalize_seq(void* stream, const uint8_t* d_seq, const int64_t* d_offsets, int32_t n_reads,
                                int32_t iupac, uint8_t* d_out, int32_t* d_out_len) {
  if (n_reads <= 0) return DC_OK;
  DC_REQUIRE(d_seq && d_offsets && d_out && d_out_len, "null pointer");
  int grid = n_reads < 65535 ? n_reads : 65535;
  hipLaunchKernelGGL(normalize_seq_kernel, dim3(grid), dim3(256), 0, as_stream(stream), d_seq, d_offsets, n_reads,
                     iupac, d_out, d_out_len);
  DC_LAUNCH_CHECK();
  return DC_OK;
}

extern "C" int dc_encode_qual(void* stream, const uint8_t* d_qual, int64_t n_bytes, int32_t qual_offset,
                              uint8_t* d_phred) {
  if (n_bytes <= 0) return DC_OK;
  DC_REQUIRE(d_qual && d_phred, "null pointer");
  DC_REQUIRE(qual_offset >= 0 && qual_offset < 256, "qual_offset out of range");
  int64_t blocks = (n_bytes + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(encode_qual_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), d_qual, n_bytes,
                     (uint8_t)qual_offset, d_phred);
  DC_LAUNCH_CHECK();
  return DC_OK;
}

extern "C" int dc_tokenize_batch(void* stream, const uint8_t* d_seq, const uint8_t* d_qual,
                                 const int64_t* d_offsets, int32_t n_reads, int32_t max_length,
                                 int32_t qual_offset, int64_t row_stride, int32_t pad_left, int32_t ids_i64,
                                 void* d_input_ids, float* d_input_quals, int8_t* d_labels, int32_t* d_n_tokens,
                                 int32_t* d_status) {
  if (n_reads <= 0) return DC_OK;
  DC_REQUIRE(d_seq && d_qual && d_offsets && d_input_ids && d_input_quals && d_status, "null pointer");
  DC_REQUIRE(max_length >= 2, "max_length must be >= 2");
  DC_REQUIRE(row_stride >= 1, "row_stride must be >= 1");
  DC_REQUIRE(qual_offset >= 0 && qual_offset < 256, "qual_offset out of range");
  int grid = n_reads < 65535 ? n_reads : 65535;
  if (ids_i64) {
    hipLaunchKernelGGL(tokenize_batch_kernel<int64_t>, dim3(grid), dim3(256), 0, as_stream(stream), d_seq, d_qual,
                       d_offsets, n_reads, max_length, (uint8_t)qual_offset, row_stride, pad_left,
                       (int64_t*)d_input_ids, d_input_quals, d_labels, d_n_tokens, d_status);
  } else {
    hipLaunchKernelGGL(tokenize_batch_kernel<uint8_t>, dim3(grid), dim3(256), 0, as_stream(stream), d_seq, d_qual,
                       d_offsets, n_reads, max_length, (uint8_t)qual_offset, row_stride, pad_left,
                       (uint8_t*)d_input_ids, d_input_quals, d_labels, d_n_tokens, d_status);
  }
  DC_LAUNCH_CHECK();
  return DC_OK;
}
