// Internal helpers shared by the HIP translation units of libdeepchopper_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
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

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE and one process may drive several, so a process-wide
// "done" flag is wrong -- and a driver call per launch is host time on a path that is host-bound for small batches.
// One bit per device and call site (a `static std::atomic<uint64_t>` inside the launch function template): the attribute
// is set BEFORE the bit, so a thread that sees the bit may launch; two threads racing on a fresh device both set it.
#define DC_SET_MAX_SMEM_ONCE_PER_DEVICE(kernel_ptr, bytes)                                                   \
  do {                                                                                                       \
    static std::atomic<uint64_t> _seen{0};                                                                   \
    int _dev = -1;                                                                                           \
    DC_HIP_CHECK(hipGetDevice(&_dev));                                                                       \
    const uint64_t _bit = (_dev >= 0 && _dev < 64) ? (1ull << _dev) : 0;                                     \
    if (!_bit || !(_seen.load(std::memory_order_acquire) & _bit)) {                                          \
      DC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_ptr),                            \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)));           \
      _seen.fetch_or(_bit, std::memory_order_release);                                                       \
    }                                                                                                        \
  } while (0)

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
