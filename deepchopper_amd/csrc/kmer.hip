// k-mer encoder (K1-K5): sliding-window integer kernel.  One 256-thread workgroup per read; the
// read's base digits and phred bytes are staged once in LDS tiles (tile + k-1 halo) and each lane
// evaluates four consecutive windows from LDS, so every input byte is fetched from HBM exactly once
// and every store is 16 bytes per lane.
#include "dc_common.h"

namespace dc {

constexpr int KMER_TILE = 2048;   // windows per LDS tile (a multiple of 1 024: 256 lanes x 4 windows)
constexpr int KMER_MAX_K = 16;    // ids must fit int32: n_bases^k < 2^31 is checked on the host

struct BaseLut {
  int8_t v[256];
};

__global__ __launch_bounds__(256) void kmer_validate_kernel(const uint8_t* __restrict__ seq,
                                                            const int64_t* __restrict__ offsets, int n_reads, int k,
                                                            BaseLut lut, int32_t* __restrict__ status) {
  __shared__ unsigned s_bad;
  for (int r = blockIdx.x; r < n_reads; r += gridDim.x) {
    const int64_t beg = offsets[r];
    const int64_t len = offsets[r + 1] - beg;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    unsigned bad = 0;
    for (int64_t i = threadIdx.x; i < len; i += 256) bad |= (lut.v[seq[beg + i]] < 0);
    if (bad) atomicOr(&s_bad, 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
      int st = 0;
      if (len < k) st |= DC_READ_SHORTER_THAN_K;
      else if (s_bad) st |= DC_READ_INVALID_BASE;
      status[r] = st;
    }
    __syncthreads();
  }
}

// Round 3: FOUR windows per lane and step, so that every store is 16 bytes per lane and 1 KiB per wave instruction
// (the first version gave each lane one window: 4-byte stores, 3.5 TB/s); the staging loads are 4 bytes per lane and
// the base -> digit table sits in LDS (indexing the by-value kernel argument was a dependent global load per base).
__global__ __launch_bounds__(256) void kmer_encode_kernel(
    const uint8_t* __restrict__ seq, const uint8_t* __restrict__ qual, const int64_t* __restrict__ offsets,
    const int32_t* __restrict__ out_row, int n_reads, int k, int n_bases, BaseLut lut, uint8_t qual_offset,
    int width, int max_seq_len, int32_t* __restrict__ input, int32_t* __restrict__ qual_out) {
  __shared__ int8_t s_lut[256];
  __shared__ __attribute__((aligned(4))) int8_t s_digit[KMER_TILE + KMER_MAX_K + 4];
  __shared__ __attribute__((aligned(4))) uint8_t s_phred[KMER_TILE + KMER_MAX_K + 4];
  const int tid = threadIdx.x;
  s_lut[tid] = lut.v[tid];
  // 4 values at row[w0 .. w0 + 3], the part of them below `lim`
  auto store4 = [](int32_t* __restrict__ row, int64_t w0, int64_t lim, int32_t a, int32_t b, int32_t c, int32_t d) {
    if (w0 + 4 <= lim) {
      const int4 o = make_int4(a, b, c, d);
      __builtin_memcpy(row + w0, &o, 16);  // (rows start at multiples of `width` elements: 4-byte alignment is all there is)
    } else {
      const int32_t e[4] = {a, b, c, d};
      for (int x = 0; x < 4 && w0 + x < lim; ++x) row[w0 + x] = e[x];
    }
  };
  // 4 bytes at src[i .. i + 3], the part of them below `lim` (others 0)
  auto load4 = [](const uint8_t* __restrict__ src, int64_t i, int64_t lim) -> uint32_t {
    uint32_t v = 0;
    if (i + 4 <= lim) {
      __builtin_memcpy(&v, src + i, 4);
    } else {
      for (int x = 0; x < 4 && i + x < lim; ++x) v |= (uint32_t)src[i + x] << (8 * x);
    }
    return v;
  };
  for (int r = blockIdx.x; r < n_reads; r += gridDim.x) {
    const int row = out_row[r];
    if (row < 0) continue;  // block-uniform
    const int64_t beg = offsets[r];
    const int64_t len = offsets[r + 1] - beg;
    const int64_t nwin = len >= k ? len - k + 1 : 0;
    int32_t* ids_row = input + ((int64_t)row * 2) * width;
    int32_t* kq_row = ids_row + width;
    int32_t* q_row = qual_out + (int64_t)row * max_seq_len;
    const int64_t n_out = nwin < width ? nwin : width;  // windows that land in the row

    for (int64_t t0 = 0; t0 < nwin; t0 += KMER_TILE) {
      const int64_t tile_w = (nwin - t0) < KMER_TILE ? (nwin - t0) : KMER_TILE;
      const int64_t tile_b = tile_w + k - 1;  // bytes needed
      __syncthreads();
      for (int64_t j = 4 * tid; j < tile_b; j += 4 * 256) {
        const uint32_t sb = load4(seq + beg + t0, j, tile_b), qb = load4(qual + beg + t0, j, tile_b);
        uint32_t dg = 0, ph = 0;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          dg |= (uint32_t)(uint8_t)s_lut[(sb >> (8 * x)) & 255u] << (8 * x);
          ph |= ((((qb >> (8 * x)) & 255u) - qual_offset) & 255u) << (8 * x);
        }
        *reinterpret_cast<uint32_t*>(s_digit + j) = dg;  // (bytes past tile_b are scratch: the arrays carry 4 spare)
        *reinterpret_cast<uint32_t*>(s_phred + j) = ph;
      }
      __syncthreads();
      for (int64_t j = 4 * tid; j < tile_w; j += 4 * 256) {
        int32_t id[4];
        int32_t kq[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          int32_t v = 0;
          uint8_t qs = 0;  // u8 wrapping sum, triat.rs:124
          if (j + x < tile_w) {
            for (int m = 0; m < k; ++m) {
              v = v * n_bases + s_digit[j + x + m];
              qs = uint8_t(qs + s_phred[j + x + m]);
            }
          }
          id[x] = v;
          kq[x] = (int32_t)(qs / (uint8_t)k);
        }
        const int64_t w = t0 + j;
        store4(ids_row, w, n_out, id[0], id[1], id[2], id[3]);
        store4(kq_row, w, n_out, kq[0], kq[1], kq[2], kq[3]);
      }
    }
    // pad rows with -1 (Vec::resize(.., -1), tensor_encoder.rs:260-266): from the first multiple of 4 at or above nwin
    // in whole quads, the few columns in front of it one by one
    {
      const int64_t q0 = (nwin + 3) & ~(int64_t)3;
      if (tid < q0 - nwin && nwin + tid < width) {
        ids_row[nwin + tid] = -1;
        kq_row[nwin + tid] = -1;
      }
      for (int64_t w = q0 + 4 * tid; w < width; w += 4 * 256) {
        store4(ids_row, w, width, -1, -1, -1, -1);
        store4(kq_row, w, width, -1, -1, -1, -1);
      }
    }
    // full phred row (written on its own: storing it from the staging loop, where the bytes are at hand, puts stores in
    // front of the tile barrier and measured 15 % slower)
    for (int64_t i = 4 * tid; i < max_seq_len; i += 4 * 256) {
      const uint32_t qb = load4(qual + beg, i, len);
      int32_t e[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) e[x] = i + x < len ? (int32_t)((((qb >> (8 * x)) & 255u) - qual_offset) & 255u) : -1;
      store4(q_row, i, max_seq_len, e[0], e[1], e[2], e[3]);
    }
  }
}

__global__ void kmer_vectorize_targets_kernel(const int32_t* __restrict__ ranges, int64_t n_rows_targets, int width,
                                              int32_t* __restrict__ target) {
  // one block per (row, target)
  for (int64_t rt = blockIdx.x; rt < n_rows_targets; rt += gridDim.x) {
    const int s = ranges[rt * 2], e = ranges[rt * 2 + 1];
    const bool fill = s < e && e <= width;  // tensor_encoder.rs:197-199
    int32_t* row = target + rt * width;
    for (int w = threadIdx.x; w < width; w += blockDim.x) row[w] = (fill && w >= s && w < e) ? 1 : 0;
  }
}

static int make_lut(const char* bases, int n_bases, int k, BaseLut* lut) {
  DC_REQUIRE(bases && n_bases > 0 && n_bases <= 127, "bases must hold 1..127 symbols");
  DC_REQUIRE(k >= 1 && k <= KMER_MAX_K, "k must be in 1..16");
  double lim = 1.0;
  for (int i = 0; i < k; ++i) lim *= n_bases;
  DC_REQUIRE(lim < 2147483648.0, "n_bases^k must fit int32");
  memset(lut->v, 0xFF, sizeof(lut->v));
  for (int i = 0; i < n_bases; ++i) lut->v[(uint8_t)bases[i]] = (int8_t)i;
  return DC_OK;
}

}  // namespace dc

using namespace dc;

extern "C" int dc_kmer_validate(void* stream, const uint8_t* d_seq, const int64_t* d_offsets, int32_t n_reads,
                                int32_t k, const char* h_bases, int32_t n_bases, int32_t* d_status) {
  if (n_reads <= 0) return DC_OK;
  DC_REQUIRE(d_seq && d_offsets && d_status, "null pointer");
  BaseLut lut;
  int rc = make_lut(h_bases, n_bases, k, &lut);
  if (rc) return rc;
  int grid = n_reads < 65535 ? n_reads : 65535;
  hipLaunchKernelGGL(kmer_validate_kernel, dim3(grid), dim3(256), 0, as_stream(stream), d_seq, d_offsets, n_reads, k,
                     lut, d_status);
  DC_LAUNCH_CHECK();
  return DC_OK;
}

extern "C" int dc_kmer_encode(void* stream, const uint8_t* d_seq, const uint8_t* d_qual, const int64_t* d_offsets,
                              const int32_t* d_out_row, int32_t n_reads, int32_t k, const char* h_bases,
                              int32_t n_bases, int32_t qual_offset, int32_t width, int32_t max_seq_len,
                              int32_t* d_input, int32_t* d_qual_out) {
  if (n_reads <= 0) return DC_OK;
  DC_REQUIRE(d_seq && d_qual && d_offsets && d_out_row && d_input && d_qual_out, "null pointer");
  DC_REQUIRE(width >= 1 && max_seq_len >= 1, "width / max_seq_len must be positive");
  DC_REQUIRE(qual_offset >= 0 && qual_offset < 256, "qual_offset out of range");
  BaseLut lut;
  int rc = make_lut(h_bases, n_bases, k, &lut);
  if (rc) return rc;
  int grid = n_reads < 65535 ? n_reads : 65535;
  hipLaunchKernelGGL(kmer_encode_kernel, dim3(grid), dim3(256), 0, as_stream(stream), d_seq, d_qual, d_offsets,
                     d_out_row, n_reads, k, n_bases, lut, (uint8_t)qual_offset, width, max_seq_len, d_input,
                     d_qual_out);
  DC_LAUNCH_CHECK();
  return DC_OK;
}

extern "C" int dc_kmer_vectorize_targets(void* stream, const int32_t* d_ranges, int32_t n_rows, int32_t n_targets,
                                         int32_t width, int32_t* d_target) {
  const int64_t n = (int64_t)n_rows * n_targets;
  if (n <= 0) return DC_OK;
  DC_REQUIRE(d_ranges && d_target && width >= 1, "bad arguments");
  int grid = n < 65535 ? (int)n : 65535;
  hipLaunchKernelGGL(kmer_vectorize_targets_kernel, dim3(grid), dim3(256), 0, as_stream(stream), d_ranges, n, width,
                     d_target);
  DC_LAUNCH_CHECK();
  return DC_OK;
}
