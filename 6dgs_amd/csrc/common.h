// common.h -- launch helpers shared by the HIP translation units of lib6dgs_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sixdgs.h"

#define SDG_CHECK_ARG(cond) \
  do {                      \
    if (!(cond)) return SIXDGS_E_BADARG; \
  } while (0)

#define SDG_LAUNCH_OK()                       \
  do {                                        \
    hipError_t _e = hipGetLastError();        \
    if (_e != hipSuccess) return (int)_e;     \
  } while (0)

static inline hipStream_t sdg_stream(sixdgs_stream_t s) { return (hipStream_t)s; }
static inline size_t sdg_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int64_t sdg_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- wave / block primitives (wave = 64 lanes on gfx950) ----------------------------------------
__device__ __forceinline__ int sdg_lane() { return threadIdx.x & 63; }
__device__ __forceinline__ int sdg_wave() { return threadIdx.x >> 6; }

template <typename T>
__device__ __forceinline__ T sdg_wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float sdg_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// inclusive scan across the 64 lanes of a wave
template <typename T>
__device__ __forceinline__ T sdg_wave_inclusive_scan(T v) {
  const int lane = sdg_lane();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    T n = __shfl_up(v, o, 64);
    if (lane >= o) v += n;
  }
  return v;
}
// Block-wide exclusive scan for blockDim.x == 64*NW; `smem` needs NW+1 elements of T.
// Returns the exclusive prefix of `v`; *total receives the block sum.  Contains two barriers.
template <typename T, int NW>
__device__ __forceinline__ T sdg_block_exclusive_scan(T v, T* smem, T* total) {
  T inc = sdg_wave_inclusive_scan(v);
  if (sdg_lane() == 63) smem[sdg_wave()] = inc;
  __syncthreads();
  T off = T(0), tot = T(0);
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    T s = smem[w];
    if (w < sdg_wave()) off += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return off + inc - v;
}

// ---- caller-owned kernel timing (include/sixdgs.h: sixdgs_profile) -----------------------------------
struct SdgProfileScope {
  sixdgs_profile* p;
  hipStream_t s;
  int slot;
  SdgProfileScope(sixdgs_profile* prof, hipStream_t stream, double flops, double bytes) : p(prof), s(stream), slot(-1) {
    if (!p || p->count >= SIXDGS_PROFILE_SLOTS) return;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess) return;
    if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return; }
    slot = p->count++;
    p->start[slot] = a; p->stop[slot] = b; p->flops[slot] = flops; p->bytes[slot] = bytes;
    (void)hipEventRecord(a, s);
  }
  ~SdgProfileScope() {
    if (slot >= 0) (void)hipEventRecord((hipEvent_t)p->stop[slot], s);
  }
};
