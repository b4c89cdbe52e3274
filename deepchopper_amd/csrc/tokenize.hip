// Tokenise kernels (T1-T5): byte work, HBM-bound.  dc_tokenize_batch: one wave per read, 16 bases per lane and step
// (16-byte accesses); the small helpers (normalize_seq, encode_qual) walk bytes with unit stride.
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

// One WAVE per read, 16 bases per lane and step (round 3; the first version gave every lane one byte: 64-byte wave
// accesses, 1.0 TB/s).  A lane loads 16 sequence and 16 quality bytes with one 16-byte access each (gfx950 runs with
// unaligned access enabled: a read starts at an arbitrary byte of the packed FASTQ buffers) and stores 16 tokens /
// labels (16 B) and 16 fp32 qualities (4 x 16 B).  Loads that would leave the buffer (first / last read of a batch)
// and the partial chunk at the end of a row go through a byte-wise path.  Arithmetic is unchanged: exact integer sum
// of squares -> sqrtf -> IEEE division per base.
constexpr int TK_WAVES = 4;  // reads per 256-thread workgroup
#ifndef TK_OCC
#define TK_OCC 7  // waves per SIMD asked of the compiler (72 VGPRs, no spill; eight spill 48 bytes)
#endif

// bit b of a nibble -> byte b of a word = 0xFF
__device__ __forceinline__ uint32_t nibble_to_bytes(uint32_t n) {
  const uint32_t m = (n * 0x00204081u) & 0x01010101u;
  return (m << 8) - m;
}

template <typename IdT>
__global__ __launch_bounds__(64 * TK_WAVES, TK_OCC) void tokenize_batch_kernel(
    const uint8_t* __restrict__ seq, const uint8_t* __restrict__ qual, const int64_t* __restrict__ offsets,
    int n_reads, int max_length, uint8_t qual_offset, int64_t row_stride, int pad_left,
    IdT* __restrict__ ids, float* __restrict__ quals, int8_t* __restrict__ labels,
    int32_t* __restrict__ n_tokens, int32_t* __restrict__ status) {
  __shared__ uint8_t s_tok[256];  // raw byte -> token id (| 0x80: removed by normalize, see below)
  // phred value -> phred / norm of the wave's current read: a read has at most 256 distinct quotients, so the IEEE
  // division (a dozen VALU instructions) runs 4 times per lane and read instead of once per base
  __shared__ float s_quot[TK_WAVES][256];
  {
    const uint8_t t = token_of_raw((uint8_t)threadIdx.x);
    s_tok[threadIdx.x] = t == 0xFF ? uint8_t(0x80 | 11) : t;  // bit 7: removed by normalize, counted as token 11
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float* const quot = s_quot[threadIdx.x >> 6];
  __shared__ uint32_t s_xch[TK_WAVES][256];  // a wave's phred bytes of one step, changing lanes for the quality stores
  uint32_t* const xch = s_xch[threadIdx.x >> 6];
  const int64_t total_bytes = offsets[n_reads];
  const int64_t wave0 = (int64_t)blockIdx.x * TK_WAVES + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * TK_WAVES;
  for (int64_t r = wave0; r < n_reads; r += n_waves) {
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

    // (32-bit indices inside a row: a tokenised read is shorter than max_length, rows are checked on the host)
    const int len32 = (int)len, stride32 = (int)row_stride;
    const int pad = pad_left ? stride32 - (int)ntok : 0;
    IdT* id_row = ids + r * row_stride;
    float* q_row = quals + r * row_stride;
    int8_t* l_row = labels ? labels + r * row_stride : nullptr;
    unsigned flags = 0;

    // 16 sequence / quality bytes for source indices [i0, i0 + 16): bytes of the neighbouring reads inside the window
    // are loaded as they are (masked by the users); a window that leaves the packed buffers (first / last read of a
    // batch) is read byte-wise
    auto load16 = [&](const uint8_t* __restrict__ src, int i0) -> uint4 {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (i0 + 16 > 0 && i0 < len32) {
        const int64_t a0 = beg + i0;
        if (a0 >= 0 && a0 + 16 <= total_bytes) {
          __builtin_memcpy(&v, src + a0, 16);
        } else {
          uint32_t w[4] = {0, 0, 0, 0};
          for (int k = 0; k < 16; ++k)
            if (i0 + k >= 0 && i0 + k < len32) w[k >> 2] |= (uint32_t)src[beg + i0 + k] << (8 * (k & 3));
          v = make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
      return v;
    };
    // Word-wise ("SWAR") handling of a 16-byte chunk.  Only the lane that holds a read's end (or, with left padding, its
    // start) has partly valid chunks, but a SIMD runs every lane's instructions: per-byte compares and selects were
    // most of this kernel (~55 instructions per base; without its stores it still took 0.115 of 0.165 ms).  Instead the
    // validity of a chunk's bytes becomes four byte masks, computed once, and invalid bytes are replaced up front by
    // values that produce the right result on their own: quality byte = qual_offset (phred 0 -> quotient 0, nothing
    // added to the sum of squares, no underflow), token word merged with the pad word by one v_bfi, label word = ~valid
    // & 0x9C9C9C9C (-100).
    struct Chunk {
      uint32_t valid[4];  // byte masks: source index inside [0, len)
      uint32_t sep[4];    //             source index == len  ([SEP])
    };
    auto chunk_masks = [&](int i0) -> Chunk {
      int k_lo = -i0, k_hi = len32 - i0;  // valid k: [k_lo, k_hi) /\ [0, 16)
      k_lo = k_lo < 0 ? 0 : k_lo > 16 ? 16 : k_lo;
      const int k_sep = k_hi;
      k_hi = k_hi < k_lo ? k_lo : k_hi > 16 ? 16 : k_hi;
      const uint32_t vbits = ((1u << k_hi) - 1u) & ~((1u << k_lo) - 1u);
      const uint32_t sbits = (unsigned)k_sep < 16u ? 1u << k_sep : 0u;
      Chunk c;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c.valid[j] = nibble_to_bytes((vbits >> (4 * j)) & 15u);
        c.sep[j] = nibble_to_bytes((sbits >> (4 * j)) & 15u);
      }
      return c;
    };
    const uint32_t qoff4 = 0x01010101u * qual_offset;
    auto mask_quals = [&](const Chunk& c, const uint4& qv, uint32_t (&qm)[4]) {  // invalid bytes -> qual_offset; underflow flag
      const uint32_t q[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) qm[j] = (q[j] & c.valid[j]) | (qoff4 & ~c.valid[j]);
      if (qual_offset <= 128) {  // "a byte of x is below n" for n <= 128, word-wise
        uint32_t any = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) any |= (qm[j] - qoff4) & ~qm[j] & 0x80808080u;
        if (any) flags |= DC_READ_QUAL_UNDERFLOW;
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k)
          if (((qm[k >> 2] >> (8 * (k & 3))) & 255u) < qual_offset) flags |= DC_READ_QUAL_UNDERFLOW;
      }
    };
    auto phred = [&](const uint32_t (&qm)[4], int k) -> unsigned { return (((qm[k >> 2] >> (8 * (k & 3))) & 255u) - qual_offset) & 255u; };
    auto sumsq16 = [&](int i0, const uint4& qv) -> unsigned {  // over [i0, i0 + 16) /\ [0, len)
      uint32_t qm[4];
      mask_quals(chunk_masks(i0), qv, qm);
      unsigned sacc = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const unsigned ph = phred(qm, k);
        sacc += ph * ph;  // 16 x 255^2 < 2^32
      }
      return sacc;
    };

    // A row of at most 1 024 columns is one step of the wave: its chunk is requested up front and, when the row carries
    // no left-pad shift, also is the lane's share of the sum of squares -- the read's bytes are fetched ONCE and the only
    // dependent memory round trip is load -> reduce -> store.  Otherwise pass 1 walks the quality bytes on its own.
    const bool one_step = stride32 <= 16 * 64;
    uint4 sv = make_uint4(0, 0, 0, 0), qv = make_uint4(0, 0, 0, 0);
    if (one_step) {
      sv = load16(seq, 16 * lane - pad);
      qv = load16(qual, 16 * lane - pad);
    }
    // pass 1: exact integer sum of squares of the phred values (quals = phred ++ [0]).
    unsigned long long acc = 0;
    if (one_step && pad == 0) {
      acc = sumsq16(16 * lane, qv);
    } else {
#pragma unroll 1
      for (int i0 = 16 * lane; i0 < len32; i0 += 16 * 64) acc += sumsq16(i0, load16(qual, i0));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    // F.normalize(dim=0): x / max(||x||_2, 1e-12), fp32 (tokenizer.py:166).
    const float nrm = fmaxf(sqrtf((float)acc), 1e-12f);  // correctly rounded (hipcc default)
    // IEEE division (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), once per phred value
    {
      float4 qq;
      qq.x = (float)(4 * lane) / nrm;
      qq.y = (float)(4 * lane + 1) / nrm;
      qq.z = (float)(4 * lane + 2) / nrm;
      qq.w = (float)(4 * lane + 3) / nrm;
      __builtin_amdgcn_wave_barrier();  // the previous read's lookups are done (one wave: LDS operations are in order)
      *reinterpret_cast<float4*>(quot + 4 * lane) = qq;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }

    // pass 2: write the whole row: pad | tokens | [SEP] (| pad on the right when !pad_left), 1 024 columns per step.
    // Tokens and labels are a byte per column: a lane's 16 and the wave's 1 024 bytes are contiguous.  The fp32
    // qualities of a lane's 16 columns are 64 bytes -- four stores of 16 bytes each, which from every lane at once are
    // four stores with a stride of 64 bytes, a quarter of every line per instruction.  So the phred bytes change lanes
    // first (a 1 KiB exchange through LDS: lane l hands over its 16, takes 4 x 4 from lanes 16 k + l / 4): store k of
    // lane l then covers columns 256 k + 4 l .. + 3, and one instruction of the wave writes 1 KiB in a piece.
#pragma unroll 1
    for (int cbase = 0; cbase < stride32; cbase += 16 * 64) {
      const int c0 = cbase + 16 * lane;
      const int i0 = c0 - pad;  // source index of the chunk's first column
      if (!one_step) {
        sv = load16(seq, i0);
        qv = load16(qual, i0);
      }
      const Chunk cm = chunk_masks(i0);
      uint32_t qm[4];
      mask_quals(cm, qv, qm);
      __builtin_amdgcn_wave_barrier();  // (one wave: LDS operations are in order; the barriers only pin the compiler)
      *reinterpret_cast<uint4*>(xch + 4 * lane) = make_uint4(qm[0], qm[1], qm[2], qm[3]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint32_t sw[4] = {sv.x, sv.y, sv.z, sv.w};
      uint32_t tokw[4], labw[4];
      uint32_t dropped = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) w |= (uint32_t)s_tok[(sw[j] >> (8 * b)) & 255u] << (8 * b);
        dropped |= w & cm.valid[j] & 0x80808080u;
        w &= 0x7F7F7F7Fu;
        w = (w & cm.valid[j]) | ((0x01010101u * DC_TOK_PAD) & ~cm.valid[j]);
        tokw[j] = (w & ~cm.sep[j]) | ((0x01010101u * DC_TOK_SEP) & cm.sep[j]);
        labw[j] = ~cm.valid[j] & (0x01010101u * (uint32_t)(uint8_t)DC_IGNORE_LABEL);
      }
      if (dropped) flags |= DC_READ_DROPPED_CHARS;
      if (c0 + 16 <= stride32) {
        if (sizeof(IdT) == 1) {
          const uint4 o = make_uint4(tokw[0], tokw[1], tokw[2], tokw[3]);
          __builtin_memcpy(id_row + c0, &o, 16);
        } else {
#pragma unroll
          for (int k = 0; k < 16; k += 2) {
            const ulonglong2 o = make_ulonglong2((tokw[k >> 2] >> (8 * (k & 3))) & 255u, (tokw[k >> 2] >> (8 * ((k + 1) & 3))) & 255u);
            __builtin_memcpy(id_row + c0 + k, &o, 16);
          }
        }
        if (l_row) {
          const uint4 o = make_uint4(labw[0], labw[1], labw[2], labw[3]);
          __builtin_memcpy(l_row + c0, &o, 16);
        }
      } else {  // the row ends inside this chunk (the next row belongs to another wave), or in front of it
        for (int k = 0; k < 16 && c0 + k < stride32; ++k) {
          id_row[c0 + k] = (IdT)((tokw[k >> 2] >> (8 * (k & 3))) & 255u);
          if (l_row) l_row[c0 + k] = (int8_t)((labw[k >> 2] >> (8 * (k & 3))) & 255u);
        }
      }
      // qualities: invalid bytes carry qual_offset, i.e. phred 0 -> 0 / norm = 0
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t w = xch[64 * k + lane];
        const int col = cbase + 256 * k + 4 * lane;
        float4 o;
        o.x = quot[((w & 255u) - qual_offset) & 255u];
        o.y = quot[(((w >> 8) & 255u) - qual_offset) & 255u];
        o.z = quot[(((w >> 16) & 255u) - qual_offset) & 255u];
        o.w = quot[((w >> 24) - qual_offset) & 255u];
        if (col + 4 <= stride32) {
          __builtin_memcpy(q_row + col, &o, 16);
        } else {
          const float e[4] = {o.x, o.y, o.z, o.w};
          for (int b = 0; b < 4 && col + b < stride32; ++b) q_row[col + b] = e[b];
        }
      }
    }
    // status flags of the read: OR over the wave
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) flags |= __shfl_xor(flags, d, 64);
    if (lane == 0) {
      if (n_tokens) n_tokens[r] = (int32_t)ntok;
      status[r] = st | (int)flags;
    }
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
  DC_REQUIRE(row_stride >= 1 && row_stride < ((int64_t)1 << 31), "row_stride must be in [1, 2^31)");
  DC_REQUIRE(qual_offset >= 0 && qual_offset < 256, "qual_offset out of range");
  int64_t blocks = ((int64_t)n_reads + TK_WAVES - 1) / TK_WAVES;
  int grid = (int)(blocks < 65535 ? blocks : 65535);
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
