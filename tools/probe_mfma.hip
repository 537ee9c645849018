// Early toolchain probe: ctypes + torch-owned pointers + f32 MFMA fragment layouts on gfx950.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_mfma32(const float* A, const float* B, float* D, int K) {
  // A [32][K] row-major, B [K][32] row-major, D [32][32]
  int l = threadIdx.x;
  f32x16 acc = {0};
  for (int k0 = 0; k0 < K; k0 += 2) {
    float a = A[(l & 31) * K + k0 + (l >> 5)];
    float b = B[(k0 + (l >> 5)) * 32 + (l & 31)];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    D[row * 32 + (l & 31)] = acc[r];
  }
}
__global__ void k_mfma16(const float* A, const float* B, float* D, int K) {
  // A [16][K], B [K][16], D [16][16]
  int l = threadIdx.x;
  f32x4 acc = {0};
  for (int k0 = 0; k0 < K; k0 += 4) {
    float a = A[(l & 15) * K + k0 + (l >> 4)];
    float b = B[(k0 + (l >> 4)) * 16 + (l & 15)];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}
extern "C" int probe_mfma32(const float* A, const float* B, float* D, int K, void* stream) {
  hipLaunchKernelGGL(k_mfma32, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, D, K);
  return (int)hipGetLastError();
}
extern "C" int probe_mfma16(const float* A, const float* B, float* D, int K, void* stream) {
  hipLaunchKernelGGL(k_mfma16, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, D, K);
  return (int)hipGetLastError();
}
