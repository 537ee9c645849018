// Probe: fp32 GEMM emulated with three bf16 planes per operand and the 6 leading cross terms on
// v_mfma_f32_32x32x16_bf16, versus the exact-fp32 MFMA chain; also pins the bf16 operand layout.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned short f2bf(float x) {  // round-to-nearest-even
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ inline float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ inline void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  h = f2bf(x); float r = x - bf2f(h);
  m = f2bf(r); r = r - bf2f(m);
  l = f2bf(r);
}
// D[32][32] = A[32][K] * B[32][K]^T  (both k-contiguous), K % 16 == 0
__global__ void k_bf16x6(const float* A, const float* B, float* D, int K, int terms) {
  int l = threadIdx.x;
  f32x16 acc = {0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    bf16x8 a[3], b[3];
    for (int e = 0; e < 8; ++e) {
      int k = k0 + 8 * (l >> 5) + e;
      unsigned short h, m, lo;
      split3(A[(l & 31) * K + k], h, m, lo); a[0][e] = h; a[1][e] = m; a[2][e] = lo;
      split3(B[(l & 31) * K + k], h, m, lo); b[0][e] = h; b[1][e] = m; b[2][e] = lo;
    }
    // smallest terms first
    if (terms >= 6) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    }
    if (terms >= 3) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
// same product with separate accumulators per magnitude class, summed small-to-large at the end
__global__ void k_bf16x6_sep(const float* A, const float* B, float* D, int K) {
  int l = threadIdx.x;
  f32x16 a0 = {0}, a1 = {0}, a2 = {0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    bf16x8 a[3], b[3];
    for (int e = 0; e < 8; ++e) {
      int k = k0 + 8 * (l >> 5) + e;
      unsigned short h, m, lo;
      split3(A[(l & 31) * K + k], h, m, lo); a[0][e] = h; a[1][e] = m; a[2][e] = lo;
      split3(B[(l & 31) * K + k], h, m, lo); b[0][e] = h; b[1][e] = m; b[2][e] = lo;
    }
    a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], a2, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], a2, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], a2, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], a1, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], a1, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], a0, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = (a2[r] + a1[r]) + a0[r];
}
__global__ void k_f32(const float* A, const float* B, float* D, int K) {
  int l = threadIdx.x;
  f32x16 acc = {0};
  for (int k0 = 0; k0 < K; k0 += 2) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * K + k0 + (l >> 5)], B[(l & 31) * K + k0 + (l >> 5)], acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
extern "C" int probe_run(const float* A, const float* B, float* D, int K, int mode, void* s) {
  if (mode == 0) hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), 0, (hipStream_t)s, A, B, D, K);
  else if (mode == 7) hipLaunchKernelGGL(k_bf16x6_sep, dim3(1), dim3(64), 0, (hipStream_t)s, A, B, D, K);
  else hipLaunchKernelGGL(k_bf16x6, dim3(1), dim3(64), 0, (hipStream_t)s, A, B, D, K, mode);
  return (int)hipGetLastError();
}

// ---- fp16 x 3: two fp16 planes per operand (x*2^s = h + l), cross terms l*h + h*l + h*h ----------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k_f16x3(const float* A, const float* B, float* D, int K, float sa, float sb, int terms) {
  int l = threadIdx.x;
  f32x16 acc = {0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    f16x8 ah, al, bh, bl;
    for (int e = 0; e < 8; ++e) {
      int k = k0 + 8 * (l >> 5) + e;
      float x = A[(l & 31) * K + k] * sa;
      _Float16 h = (_Float16)x; ah[e] = h; al[e] = (_Float16)(x - (float)h);
      float y = B[(l & 31) * K + k] * sb;
      _Float16 g = (_Float16)y; bh[e] = g; bl[e] = (_Float16)(y - (float)g);
    }
    if (terms >= 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, acc, 0, 0, 0);
    if (terms >= 3) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
  }
  float inv = 1.0f / (sa * sb);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r] * inv;
}
extern "C" int probe_run_f16(const float* A, const float* B, float* D, int K, float sa, float sb, int terms, void* s) {
  hipLaunchKernelGGL(k_f16x3, dim3(1), dim3(64), 0, (hipStream_t)s, A, B, D, K, sa, sb, terms);
  return (int)hipGetLastError();
}
