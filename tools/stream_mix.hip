// What a plain streaming kernel reaches on this GPU for the read : write mixes of the integer kernels -- the yardstick
// next to which `tokenize_batch_kernel` (2 B read : 6 B written per base), `smooth_kernel` (8 : 1) and
// `kmer_encode_kernel` (2 : 8) are quoted in DESIGN.md section 4.  Not part of the product library.
//
//   hipcc -O3 --offload-arch=gfx950 tools/stream_mix.hip -o tools/bin/stream_mix && tools/bin/stream_mix
//
// Every lane moves 16 bytes per access (the widest global access), grid-stride, no arithmetic worth the name; buffers
// are 64 Mi elements (the size of the 64 k-read x 1 kb launches of tools/bench_intkernels.py), timed with HIP events
// over 20 launches after 3 warm-ups.  Output: one JSON line per mix.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// RD / WR: 16-byte vectors read / written per lane and step, from / to separate streams
template <int RD, int WR>
__global__ __launch_bounds__(256) void mix_kernel(const u4* __restrict__ src, u4* __restrict__ dst, int64_t n_steps,
                                                  unsigned* __restrict__ sink) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  u4 acc = {0u, 0u, 0u, 0u};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_steps; i += stride) {
    u4 v[RD > 0 ? RD : 1];
#pragma unroll
    for (int r = 0; r < RD; ++r) v[r] = src[(int64_t)r * n_steps + i];
#pragma unroll
    for (int r = 0; r < RD; ++r) acc ^= v[r];
#pragma unroll
    for (int w = 0; w < WR; ++w) {
      u4 o = acc;
      o.x += (unsigned)w;
      dst[(int64_t)w * n_steps + i] = o;
    }
  }
  if (WR == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;  // keeps the loads alive
}

template <int RD, int WR>
static void run(const char* name, const u4* src, u4* dst, int64_t n_steps, unsigned* sink) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grid = 256 * 16;  // 16 workgroups per CU
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((mix_kernel<RD, WR>), dim3(grid), dim3(256), 0, 0, src, dst, n_steps, sink);
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((mix_kernel<RD, WR>), dim3(grid), dim3(256), 0, 0, src, dst, n_steps, sink);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double bytes = 16.0 * (RD + WR) * (double)n_steps;
  printf("{\"mix\": \"%s\", \"read_vectors\": %d, \"write_vectors\": %d, \"MB_per_launch\": %.1f, \"ms\": %.4f, \"TB_per_s\": %.3f}\n", name, RD,
         WR, bytes / 1e6, ms, bytes / ms / 1e9);
}

int main() {
  // 8 vectors x n_steps x 16 B per side at most; n_steps chosen so that a launch moves about what the integer kernels
  // move (0.5 - 0.65 GB) when RD + WR = 8 .. 10
  const int64_t n_steps = 4 << 20;  // 4 Mi steps x 16 B = 64 MiB per stream
  u4 *src, *dst;
  unsigned* sink;
  CK(hipMalloc(&src, 16 * n_steps * 8));
  CK(hipMalloc(&dst, 16 * n_steps * 8));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(src, 1, 16 * n_steps * 8));
  CK(hipMemset(dst, 0, 16 * n_steps * 8));
  CK(hipDeviceSynchronize());
  run<8, 0>("read only", src, dst, n_steps, sink);
  run<0, 8>("write only", src, dst, n_steps, sink);
  run<4, 4>("copy 1:1", src, dst, n_steps, sink);
  run<2, 6>("tokenize-like 2:6", src, dst, n_steps, sink);
  run<8, 1>("smooth-like 8:1", src, dst, n_steps, sink);
  run<2, 8>("kmer-like 2:8", src, dst, n_steps, sink);
  run<2, 1>("out_proj-like 2:1", src, dst, n_steps, sink);
  return 0;
}
