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
  // scaled-fp16 x 3 variant only: an upper bound of max|.| of every operand row (within 2x of the true maximum)
  const float* amax0;   // [M] for segment a0
  const float* amax1;   // [M] for segment a1 (or null)
  const float* bmax;    // [N]
};

// bijective XCD-aware remap: consecutive work items land on the same XCD (block b runs on XCD b % 8),
// so workgroups that share an operand panel hit in that XCD's L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u;
  const unsigned xcd = bid & 7u, slot = bid >> 3;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

// Guarded 16-byte operand load without a branch: the address is clamped into the matrix (row -> last valid
// row, k -> 0) so the load is always legal, and the value is zeroed by a select.  A conditional load would be
// scalarised by hipcc into four exec-masked global_load_dword (measured: 3x slower kernel).
__device__ __forceinline__ float4 load4_guarded(const float* base, int64_t row, int64_t rows, int64_t ld, int k, bool k_ok) {
  const bool ok = k_ok && row < rows;
  const int64_t rc = row < rows ? row : rows - 1;
  float4 v = *reinterpret_cast<const float4*>(base + rc * ld + (k_ok ? k : 0));
  v.x = ok ? v.x : 0.f;
  v.y = ok ? v.y : 0.f;
  v.z = ok ? v.z : 0.f;
  v.w = ok ? v.w : 0.f;
  return v;
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
    for (int p = 0; p < APASS; ++p) ra[p] = load4_guarded(abase, row0 + p * (NT / 8) + lrow, g.m, lda, ka, kin);
#pragma unroll
    for (int p = 0; p < BPASS; ++p) rb[p] = load4_guarded(g.b, col0 + p * (NT / 8) + lrow, g.n, g.ldb, kk, kin);
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

// ------------------------------------------------------------------------------------------------
// bf16x6 variant: the same fp32 product on the bf16 matrix pipe (16x the fp32 MFMA rate).
// Every fp32 operand value is split on the fly into three bf16 planes x = h + m + l (round-to-nearest,
// exact: 3 x 8 significant bits cover the 24-bit mantissa) and the six leading cross terms
//   l*h + h*l + m*m + m*h + h*m + h*h       (dropped: m*l, l*m, l*l <= 2^-26 |x||y|)
// are accumulated in fp32 by v_mfma_f32_32x32x16_bf16, smallest terms first.  Each bf16 x bf16 product is
// exact in fp32, so the only rounding is the accumulation: measured error 1.3e-7 * sum|a||b| at K = 384
// versus 1.6e-7 for the fp32 MFMA chain (tools/probe_bf16x6.*).  6 MFMAs of 32 cycles per
// 32x32x16 block against 8 x 64 cycles for v_mfma_f32_32x32x2_f32: 2.67x less matrix-pipe time.
//
// LDS image of a slab (32 k): per row 3 planes x 32 bf16 = 3 x 64 B, row stride 208 B (13 x 16 B: odd
// multiple of the 16-byte slot, so the 16-lane groups of ds_read_b128 hit 16 distinct slots).  Operand
// layout of v_mfma_f32_32x32x16_bf16: lane l holds row l&31, k = 8*(l>>5) .. +7 -> one 16-byte read per
// (row block, plane, k-step).  Single LDS buffer (53 KB -> up to 3 workgroups per CU) with the next slab's
// global loads held in registers during the MFMAs; two barriers per slab.
// ------------------------------------------------------------------------------------------------
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
constexpr int kB6Row = 208;
constexpr int kB6Bytes = (128 + kBN) * kB6Row;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  bf16x2_t v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}
// two fp32 values -> one packed dword per plane
__device__ __forceinline__ void split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  h = pack_bf16(a, b);
  const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = pack_bf16(ra, rb);
  l = pack_bf16(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ void split_store8(const float4& lo, const float4& hi, char* dst) {
  uint4 h, m, l;
  split_pair(lo.x, lo.y, h.x, m.x, l.x);
  split_pair(lo.z, lo.w, h.y, m.y, l.y);
  split_pair(hi.x, hi.y, h.z, m.z, l.z);
  split_pair(hi.z, hi.w, h.w, m.w, l.w);
  *reinterpret_cast<uint4*>(dst) = h;
  *reinterpret_cast<uint4*>(dst + 64) = m;
  *reinterpret_cast<uint4*>(dst + 128) = l;
}

__device__ __forceinline__ void gemm_mainloop_b6(const GemmOperands& g, int64_t row0, int64_t col0, char* smem,
                                                 f32x16 (&acc)[2][2]) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = tid >> 2, lchunk = tid & 3;   // loader: 4 lanes x 32 B cover one 128-B row segment
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[2][2], rb[2][2];
  const int nslab = (g.k + kBK - 1) / kBK;
  auto load_slab = [&](int s) {
    const int kk = s * kBK + lchunk * 8;
    const bool seg1 = g.a1 != nullptr && kk >= g.k0;
    const float* abase = seg1 ? g.a1 : g.a0;
    const int64_t lda = seg1 ? g.lda1 : g.lda0;
    const int ka = seg1 ? kk - g.k0 : kk;
    const bool k0ok = kk < g.k, k1ok = kk + 4 < g.k;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      ra[p][0] = load4_guarded(abase, row0 + p * 64 + lrow, g.m, lda, ka, k0ok);
      ra[p][1] = load4_guarded(abase, row0 + p * 64 + lrow, g.m, lda, ka + 4, k1ok);
      rb[p][0] = load4_guarded(g.b, col0 + p * 64 + lrow, g.n, g.ldb, kk, k0ok);
      rb[p][1] = load4_guarded(g.b, col0 + p * 64 + lrow, g.n, g.ldb, kk + 4, k1ok);
    }
  };
  auto store_slab = [&]() {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      split_store8(ra[p][0], ra[p][1], smem + (p * 64 + lrow) * kB6Row + lchunk * 16);
      split_store8(rb[p][0], rb[p][1], smem + (128 + p * 64 + lrow) * kB6Row + lchunk * 16);
    }
  };

  load_slab(0);
  const char* sa = smem + (wm * 64 + (lane & 31)) * kB6Row + (lane >> 5) * 16;
  const char* sb = smem + (128 + wn * 64 + (lane & 31)) * kB6Row + (lane >> 5) * 16;
  for (int s = 0; s < nslab; ++s) {
    store_slab();
    __syncthreads();
    if (s + 1 < nslab) load_slab(s + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[t][p] = *reinterpret_cast<const bf16x8*>(sa + t * 32 * kB6Row + p * 64 + ks * 32);
          b[t][p] = *reinterpret_cast<const bf16x8*>(sb + t * 32 * kB6Row + p * 64 + ks * 32);
        }
      // (A plane, B plane) pairs, smallest magnitude first; 4 independent accumulators per pair
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
      constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][PA[q]], b[0][PB[q]], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][PA[q]], b[1][PB[q]], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][PA[q]], b[0][PB[q]], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][PA[q]], b[1][PB[q]], acc[1][1], 0, 0, 0);
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// scaled fp16 x 3 variant (the dense layers of the ray MLP, round 2): the arithmetic of the scorer's logits kernel -- every operand
// ROW multiplied by the power of two that puts its largest magnitude in [2^13, 2^14), split into two fp16 planes x 2^s = h + l
// (22 significant bits for everything within 2^-17 of the row maximum, an absolute error <= 2^-38 of the row maximum below that),
// three cross terms l*h + h*l + h*h on v_mfma_f32_32x32x16_f16 -- with the split done on the fly like the bf16 x 6 variant.  Half
// the MFMA instructions and two thirds of the LDS bytes of bf16 x 6; measured error 1.0e-7 * sum|a||b| (tools/probe_f16x3.py).
// The row maxima come with the operands (GemmOperands::amax0 / amax1 / bmax): the producing layer's epilogue leaves them
// (k_linear's rowmax output), the weights' are computed when they are packed.  The caller undoes the scales per output element:
// f3_inv_scale(row maximum of A) * f3_inv_scale(row maximum of B).
// LDS image of a slab: per row 2 planes x 32 fp16 = 2 x 64 B, row stride 144 B (9 x 16 B: odd multiple of the 16-byte slot).
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
constexpr int kF3Row = 144;
constexpr int kF3Bytes = (128 + kBN) * kF3Row;

// sh with m * 2^sh in [2^13, 2^14) for a normal m > 0 (clamped to +-100; m == 0 gives 2^100, harmless)
__device__ __forceinline__ int f3_shift(float m) {
  const int eb = (int)((__float_as_uint(m) >> 23) & 0xffu);
  const int sh = 140 - eb;
  return sh > 100 ? 100 : (sh < -100 ? -100 : sh);
}
__device__ __forceinline__ float f3_scale(float m) { return __uint_as_float((unsigned)(127 + f3_shift(m)) << 23); }
__device__ __forceinline__ float f3_inv_scale(float m) { return __uint_as_float((unsigned)(127 - f3_shift(m)) << 23); }

__device__ __forceinline__ void f3_split_store8(const float4& lo, const float4& hi, float sc, char* dst) {
  const float x[8] = {lo.x * sc, lo.y * sc, lo.z * sc, lo.w * sc, hi.x * sc, hi.y * sc, hi.z * sc, hi.w * sc};
  f16x8_t h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const _Float16 hh = (_Float16)x[e];
    h[e] = hh;
    l[e] = (_Float16)(x[e] - (float)hh);
  }
  *reinterpret_cast<f16x8_t*>(dst) = h;
  *reinterpret_cast<f16x8_t*>(dst + 64) = l;
}

__device__ __forceinline__ void gemm_mainloop_f3(const GemmOperands& g, int64_t row0, int64_t col0, char* smem, f32x16 (&acc)[2][2]) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = tid >> 2, lchunk = tid & 3;   // loader: 4 lanes x 32 B cover one 128-B row segment
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // the four operand rows this thread stages (two of A, two of B) keep their scale for the whole contraction
  float sca[2], scb[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int64_t ra_ = min(row0 + p * 64 + lrow, g.m - 1), rb_ = min(col0 + p * 64 + lrow, g.n - 1);
    float ma = g.amax0[ra_];
    if (g.amax1 != nullptr) ma = fmaxf(ma, g.amax1[ra_]);
    sca[p] = f3_scale(ma);
    scb[p] = f3_scale(g.bmax[rb_]);
  }

  float4 ra[2][2], rb[2][2];
  const int nslab = (g.k + kBK - 1) / kBK;
  auto load_slab = [&](int s) {
    const int kk = s * kBK + lchunk * 8;
    const bool seg1 = g.a1 != nullptr && kk >= g.k0;
    const float* abase = seg1 ? g.a1 : g.a0;
    const int64_t lda = seg1 ? g.lda1 : g.lda0;
    const int ka = seg1 ? kk - g.k0 : kk;
    const bool k0ok = kk < g.k, k1ok = kk + 4 < g.k;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      ra[p][0] = load4_guarded(abase, row0 + p * 64 + lrow, g.m, lda, ka, k0ok);
      ra[p][1] = load4_guarded(abase, row0 + p * 64 + lrow, g.m, lda, ka + 4, k1ok);
      rb[p][0] = load4_guarded(g.b, col0 + p * 64 + lrow, g.n, g.ldb, kk, k0ok);
      rb[p][1] = load4_guarded(g.b, col0 + p * 64 + lrow, g.n, g.ldb, kk + 4, k1ok);
    }
  };
  auto store_slab = [&]() {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      f3_split_store8(ra[p][0], ra[p][1], sca[p], smem + (p * 64 + lrow) * kF3Row + lchunk * 16);
      f3_split_store8(rb[p][0], rb[p][1], scb[p], smem + (128 + p * 64 + lrow) * kF3Row + lchunk * 16);
    }
  };

  load_slab(0);
  const char* sa = smem + (wm * 64 + (lane & 31)) * kF3Row + (lane >> 5) * 16;
  const char* sb = smem + (128 + wn * 64 + (lane & 31)) * kF3Row + (lane >> 5) * 16;
  for (int s = 0; s < nslab; ++s) {
    store_slab();
    __syncthreads();
    if (s + 1 < nslab) load_slab(s + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8_t a[2][2], b[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          a[t][p] = *reinterpret_cast<const f16x8_t*>(sa + t * 32 * kF3Row + p * 64 + ks * 32);
          b[t][p] = *reinterpret_cast<const f16x8_t*>(sb + t * 32 * kF3Row + p * 64 + ks * 32);
        }
      constexpr int PA[3] = {1, 0, 0};   // (A plane, B plane): l*h, h*l, h*h -- smallest magnitude first
      constexpr int PB[3] = {0, 1, 0};
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][PA[q]], b[0][PB[q]], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][PA[q]], b[1][PB[q]], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][PA[q]], b[0][PB[q]], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][PA[q]], b[1][PB[q]], acc[1][1], 0, 0, 0);
      }
    }
    __syncthreads();
  }
}

constexpr int kMmaF32 = 0, kMmaBf16x6 = 1, kMmaF16x3 = 2;
template <int MMA>
struct TileSmem {
  static constexpr int kBytes = MMA == kMmaBf16x6 ? kB6Bytes : (MMA == kMmaF16x3 ? kF3Bytes : GemmSmem<2>::kBytes);
};
template <int MMA>
__device__ __forceinline__ void gemm_tile(const GemmOperands& g, int64_t row0, int64_t col0, char* smem, f32x16 (&acc)[2][2]) {
  if constexpr (MMA == kMmaBf16x6) gemm_mainloop_b6(g, row0, col0, smem, acc);
  else if constexpr (MMA == kMmaF16x3) gemm_mainloop_f3(g, row0, col0, smem, acc);
  else gemm_mainloop<2>(g, row0, col0, reinterpret_cast<float*>(smem), acc);
}

// row / column (relative to the workgroup tile) of accumulator element (tm, tn, r) of this lane
__device__ __forceinline__ int acc_row(int wm, int tm, int r, int lane) {
  return wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int acc_col(int wn, int tn, int lane) { return wn * 64 + tn * 32 + (lane & 31); }

}  // namespace sdg
