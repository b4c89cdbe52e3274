// dc_pt_pack: the storages of the predict -> chop wire format, laid out on the device.
//
// `deepchopper predict` writes one `.pt` per reference batch (CustomWriter.write_on_batch_end, deepchopper/models/
// callbacks.py:12-25: torch.save of {prediction f32[B,L,2], target i64[B,L], seq i64[B,L], qual f32[B,L], id i64[B,256]}).  The
// batches of a fused launch sit in the launch's dense [n, row_stride] arrays, each from the column its own left padding
// starts at: on a nanopore-like length mix the dense arrays are 4x what the files hold, and bringing them to the host to
// cut the batches out there moved 12.6 GB where the files are 5.8 GB.  This kernel cuts and converts on the device: for every
// file, its rows from its column on, as the four storages `data/0..3` exactly as the file holds them (token ids and labels
// widened to int64), back to back at the file's offset of one output buffer -- which goes to pinned host memory in one
// copy and from there into the files without another pass (include/deepchopper_io.h: dc_pt_submit_packed).
// HBM-bound, 14 B read + 28 B written per position; coalesced on both sides (one position per lane and step): 115 us per
// launch of ~5.6 M positions, 0.2 % of the GPU time of a run that writes every file.
#include "dc_common.h"

namespace dc {
namespace {

constexpr int kPackThreads = 256, kPackSteps = 8, kPackBlock = kPackThreads * kPackSteps;

template <typename IdT>
__global__ __launch_bounds__(kPackThreads) void pt_pack_kernel(const float2* __restrict__ logits, const int8_t* __restrict__ labels,
                                                               const IdT* __restrict__ ids, const float* __restrict__ quals,
                                                               int64_t stride, const int64_t* __restrict__ table,
                                                               unsigned char* __restrict__ out) {
  const int64_t* t = table + 4 * (int64_t)blockIdx.y;
  const int64_t row0 = t[0], c0 = t[2];
  const uint32_t B = (uint32_t)t[1], L = (uint32_t)(stride - c0), n = B * L;  // (n < 2^31: checked on the host)
  const uint32_t p0 = blockIdx.x * (uint32_t)kPackBlock;
  if (p0 >= n) return;
  unsigned char* base = out + t[3];
  float2* o0 = reinterpret_cast<float2*>(base);                   // prediction f32 [B][L][2]
  int64_t* o1 = reinterpret_cast<int64_t*>(base) + n;             // target i64 [B][L]
  int64_t* o2 = o1 + n;                                           // seq i64 [B][L]
  float* o3 = reinterpret_cast<float*>(o2 + n);                   // qual f32 [B][L]
#pragma unroll
  for (int k = 0; k < kPackSteps; ++k) {
    const uint32_t p = p0 + k * kPackThreads + threadIdx.x;
    if (p < n) {
      const uint32_t r = p / L, c = p - r * L;
      const int64_t src = (row0 + r) * stride + c0 + c;
      o0[p] = logits[src];
      o1[p] = (int64_t)labels[src];
      o2[p] = (int64_t)ids[src];
      o3[p] = quals[src];
    }
  }
}

}  // namespace
}  // namespace dc

extern "C" int dc_pt_pack(void* stream, const float* d_logits, const int8_t* d_labels, const void* d_ids, int32_t ids_i64,
                          const float* d_quals, int64_t n_rows, int64_t row_stride, int32_t n_files,
                          const int64_t* h_file_table, const int64_t* d_file_table, void* d_out, int64_t out_bytes) {
  using namespace dc;
  if (n_files == 0) return DC_OK;
  DC_REQUIRE(d_logits && d_labels && d_ids && d_quals && h_file_table && d_file_table && d_out, "null pointer");
  DC_REQUIRE(n_files > 0 && n_files <= 65535 && n_rows > 0 && row_stride > 0, "bad launch shape");
  int64_t max_n = 0;
  for (int i = 0; i < n_files; ++i) {
    const int64_t row0 = h_file_table[4 * i], B = h_file_table[4 * i + 1], c0 = h_file_table[4 * i + 2], off = h_file_table[4 * i + 3];
    DC_REQUIRE(row0 >= 0 && B >= 1 && row0 + B <= n_rows && c0 >= 0 && c0 < row_stride, "a file's rows / column lie outside the launch");
    const int64_t n = B * (row_stride - c0);
    DC_REQUIRE(n < ((int64_t)1 << 31), "a file of 2^31 positions or more");
    DC_REQUIRE(off >= 0 && off % 64 == 0 && off + n * 28 <= out_bytes, "a file's payload does not fit the output buffer");
    max_n = n > max_n ? n : max_n;
  }
  const dim3 grid((unsigned)((max_n + kPackBlock - 1) / kPackBlock), (unsigned)n_files), block(kPackThreads);
  hipStream_t st = as_stream(stream);
  if (ids_i64)
    hipLaunchKernelGGL((pt_pack_kernel<int64_t>), grid, block, 0, st, reinterpret_cast<const float2*>(d_logits), d_labels,
                       static_cast<const int64_t*>(d_ids), d_quals, row_stride, d_file_table, static_cast<unsigned char*>(d_out));
  else
    hipLaunchKernelGGL((pt_pack_kernel<unsigned char>), grid, block, 0, st, reinterpret_cast<const float2*>(d_logits), d_labels,
                       static_cast<const unsigned char*>(d_ids), d_quals, row_stride, d_file_table, static_cast<unsigned char*>(d_out));
  DC_LAUNCH_CHECK();
  return DC_OK;
}
