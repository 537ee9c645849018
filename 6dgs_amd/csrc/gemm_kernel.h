// gemm_kernel.h -- fp32-in / fp32-accumulate MFMA tile kernel for gfx950 (MI355X, CDNA4).
//
//   D[i][j] = sum_k A[i][k] * B[j][k]        (both operands row-major with k contiguous: "NT")
//
// used for every dense contraction on the 6DGS pose path:
//   ray MLP layers / k_proj / q_proj:  A = activations [rows, K], B = Linear.weight [N, K]
//   scorer logits:                      A = q [tokens, 384],       B = key cache [rays, 384]
//
// Why fp32 MFMA: the reference computes in fp32 and the top-k ray indices must be reproduced
// (north_star: "ray indices/top-k bit-exact"); v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain at
// the fp32 vector peak (157 TFLOP/s dense on MI355X) and leaves the VALU free for the epilogues.
//
// Tiling (64-wide wavefronts): workgroup = WM x 2 waves, each wave owns a 64x64 sub-tile = 2x2
// MFMA 32x32 accumulators (64 VGPRs).  K is consumed in slabs of 32: both operand slabs are staged
// global -> registers (float4, 8 lanes per 128-B row segment) -> LDS rows padded to 36 floats, which
// makes the ds_read_b128 fragment reads bank-conflict free (row stride 36 dwords: 16 rows tile the 64
// banks exactly).  Within a slab the k index is permuted between the two 32-lane halves of the wave
// (half g takes k = 16 g + s at MFMA step s) so that every lane fetches its 16 fragment values with four
// 16-byte LDS reads; the permutation is applied to A and B alike, so the sum is unchanged.
// Double-buffered LDS, one barrier per slab, next slab's global loads in flight during the MFMAs.
#pragma once
#include "common.h"

namespace sdg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBK = 32;        // k per slab
constexpr int kLdsRow = 36;    // padded LDS row (floats)
constexpr int kBN = 128;       // columns (j) per workgroup: 2 wave columns x 64

struct GemmOperands {
  const float* a0;   // [M, K0] first A segment
  const float* a1;   // [M, K-K0] second A segment (concatenated along k) or null
  const float* b;    // [N, K]
  int64_t lda0, lda1, ldb;
  int64_t m, n;      // valid rows of A / rows of B
  int k, k0;         // total K (multiple of 4), K0 (multiple of 32 when a1 != null, else == k)
};

// bijective XCD-aware remap: consecutive work items land on the same XCD (block b runs on XCD b % 8),
// so workgroups that share an operand panel hit in that XCD's L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u;
  const unsigned xcd = bid & 7u, slot = bid >> 3;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

template <int WM>
struct GemmSmem {
  static constexpr int BM = WM * 64;
  static constexpr int kStageFloats = (BM + kBN) * kLdsRow;
  static constexpr int kBytes = 2 * kStageFloats * 4;
};

// The mainloop.  acc[tm][tn] is the 32x32 accumulator of sub-tile (tm, tn) of this wave's 64x64.
// C/D layout of v_mfma_f32_32x32x2_f32: lane l, register r -> row (r&3) + 8*(r>>2) + 4*(l>>5),
// column l&31.
template <int WM>
__device__ __forceinline__ void gemm_mainloop(const GemmOperands& g, int64_t row0, int64_t col0, float* smem,
                                              f32x16 (&acc)[2][2]) {
  constexpr int BM = WM * 64;
  constexpr int NT = WM * 128;
  constexpr int APASS = BM * 8 / NT;   // float4 loads per thread per slab for A
  constexpr int BPASS = kBN * 8 / NT;
  constexpr int kStage = GemmSmem<WM>::kStageFloats;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = tid >> 3, lchunk = tid & 7;   // loader: 8 lanes cover one 128-B row segment

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[APASS], rb[BPASS];
  const int nslab = (g.k + kBK - 1) / kBK;

  auto load_slab = [&](int s) {
    const int kk = s * kBK + lchunk * 4;
    const bool kin = kk < g.k;
    const bool seg1 = g.a1 != nullptr && kk >= g.k0;
    const float* abase = seg1 ? g.a1 : g.a0;
    const int64_t lda = seg1 ? g.lda1 : g.lda0;
    const int ka = seg1 ? kk - g.k0 : kk;
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
      const int64_t r = row0 + p * (NT / 8) + lrow;
      ra[p] = (kin && r < g.m) ? *reinterpret_cast<const float4*>(abase + r * lda + ka) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < BPASS; ++p) {
      const int64_t r = col0 + p * (NT / 8) + lrow;
      rb[p] = (kin && r < g.n) ? *reinterpret_cast<const float4*>(g.b + r * g.ldb + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_slab = [&](int buf) {
    float* sa = smem + buf * kStage;
    float* sb = sa + BM * kLdsRow;
#pragma unroll
    for (int p = 0; p < APASS; ++p)
      *reinterpret_cast<float4*>(sa + (p * (NT / 8) + lrow) * kLdsRow + lchunk * 4) = ra[p];
#pragma unroll
    for (int p = 0; p < BPASS; ++p)
      *reinterpret_cast<float4*>(sb + (p * (NT / 8) + lrow) * kLdsRow + lchunk * 4) = rb[p];
  };

  load_slab(0);
  store_slab(0);
  __syncthreads();

  const int frow = lane & 31, fk = (lane >> 5) * 16;
  for (int s = 0; s < nslab; ++s) {
    const int buf = s & 1;
    if (s + 1 < nslab) load_slab(s + 1);
    const float* sa = smem + buf * kStage + (wm * 64 + frow) * kLdsRow + fk;
    const float* sb = smem + buf * kStage + BM * kLdsRow + (wn * 64 + frow) * kLdsRow + fk;
#pragma unroll
    for (int h = 0; h < 2; ++h) {   // two halves of 8 k-steps keep the fragment registers at 32
      float4 a[2][2], b[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          a[t][c] = *reinterpret_cast<const float4*>(sa + t * 32 * kLdsRow + h * 8 + c * 4);
          b[t][c] = *reinterpret_cast<const float4*>(sb + t * 32 * kLdsRow + h * 8 + c * 4);
        }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a0 = e == 0 ? a[0][c].x : e == 1 ? a[0][c].y : e == 2 ? a[0][c].z : a[0][c].w;
          const float a1 = e == 0 ? a[1][c].x : e == 1 ? a[1][c].y : e == 2 ? a[1][c].z : a[1][c].w;
          const float b0 = e == 0 ? b[0][c].x : e == 1 ? b[0][c].y : e == 2 ? b[0][c].z : b[0][c].w;
          const float b1 = e == 0 ? b[1][c].x : e == 1 ? b[1][c].y : e == 2 ? b[1][c].z : b[1][c].w;
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
      }
    }
    if (s + 1 < nslab) store_slab(buf ^ 1);
    __syncthreads();
  }
}

// row / column (relative to the workgroup tile) of accumulator element (tm, tn, r) of this lane
__device__ __forceinline__ int acc_row(int wm, int tm, int r, int lane) {
  return wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int acc_col(int wn, int tn, int lane) { return wn * 64 + tn * 32 + (lane & 31); }

}  // namespace sdg
