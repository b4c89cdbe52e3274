// Internal helpers shared by the HIP translation units of libdeepchopper_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/deepchopper_hip.h"

namespace dc {

void set_error(const char* fmt, ...);

#define DC_HIP_CHECK(expr)                                                                     \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      dc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return DC_ERR_HIP;                                                                       \
    }                                                                                          \
  } while (0)

#define DC_REQUIRE(cond, msg)                                            \
  do {                                                                   \
    if (!(cond)) {                                                       \
      dc::set_error("invalid argument: %s (%s)", msg, #cond);            \
      return DC_ERR_INVALID_ARG;                                         \
    }                                                                    \
  } while (0)

#define DC_LAUNCH_CHECK()                                                                   \
  do {                                                                                      \
    hipError_t _e = hipGetLastError();                                                      \
    if (_e != hipSuccess) {                                                                 \
      dc::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return DC_ERR_HIP;                                                                    \
    }                                                                                       \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int WAVE = 64;

// Exclusive prefix over the 256 threads of a block of one unsigned value per thread.
// `scratch` holds >= 5 unsigned.  Returns the exclusive prefix; *total receives the block sum.
__device__ __forceinline__ unsigned block_exclusive_scan_256(unsigned v, unsigned* scratch, unsigned* total) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  unsigned inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    unsigned o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) scratch[wave] = inc;
  __syncthreads();
  unsigned base = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    unsigned t = scratch[w];
    if (w < wave) base += t;
  }
  if (total) *total = scratch[0] + scratch[1] + scratch[2] + scratch[3];
  __syncthreads();
  return base + inc - v;
}

// The same for a block of BS = 64 / 128 / 256 threads (BS / 64 waves).
template <int BS>
__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned* scratch, unsigned* total) {
  static_assert(BS % 64 == 0 && BS <= 256, "block of 1..4 waves");
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  unsigned inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    unsigned o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (BS == 64) {
    if (total) *total = __shfl(inc, 63, 64);
    return inc - v;
  }
  if (lane == 63) scratch[wave] = inc;
  __syncthreads();
  unsigned base = 0, sum = 0;
#pragma unroll
  for (int w = 0; w < BS / 64; ++w) {
    unsigned t = scratch[w];
    if (w < wave) base += t;
    sum += t;
  }
  if (total) *total = sum;
  __syncthreads();
  return base + inc - v;
}

}  // namespace dc
