// gemm.hip -- dense layers of the scorer: ray encoding, the ray MLP + k_proj chain (key cache,
// once per scene), q_proj (per image), weight packing.  All contractions run on the fp32 MFMA tile
// kernel of gemm_kernel.h with bias/ReLU fused into the epilogue.
#include "gemm_kernel.h"
#include "device_math.h"
#include "dense.h"

using namespace sdg;

namespace {

// ------------------------------------------------------------------------------------------------
// y[M,N] = act(A . W^T + b)
// ------------------------------------------------------------------------------------------------
template <int MMA, bool RELU>
__global__ void __launch_bounds__(256, 2) k_linear(GemmOperands g, const float* __restrict__ bias, float* __restrict__ y,
                                                    int64_t ldy, unsigned n_tiles, unsigned total_tiles, float* __restrict__ out_rowmax) {
  __shared__ __attribute__((aligned(16))) char smem[TileSmem<MMA>::kBytes];
  const unsigned w = xcd_remap(blockIdx.x, total_tiles);
  const int64_t row0 = (int64_t)(w / n_tiles) * 128;
  const int64_t col0 = (int64_t)(w % n_tiles) * kBN;
  f32x16 acc[2][2];
  gemm_tile<MMA>(g, row0, col0, smem, acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  float ib[2] = {1.f, 1.f};
  if (MMA == kMmaF16x3) {      // undo the per-row operand scales (exact powers of two)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) ib[tn] = f3_inv_scale(g.bmax[min(col0 + acc_col(wn, tn, lane), g.n - 1)]);
  }
  float rmx[2][16];            // rowmax output: max |y| of this lane's 2 columns for each of its 32 rows
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = row0 + acc_row(wm, tm, r, lane);
      float ia = 1.f;
      if (MMA == kMmaF16x3) {
        const int64_t rc = min(row, g.m - 1);
        float ma = g.amax0[rc];
        if (g.amax1 != nullptr) ma = fmaxf(ma, g.amax1[rc]);
        ia = f3_inv_scale(ma);
      }
      float mx = 0.f;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int64_t col = col0 + acc_col(wn, tn, lane);
        const float bv = (bias != nullptr && col < g.n) ? bias[col] : 0.f;
        if (row < g.m && col < g.n) {
          float v = (MMA == kMmaF16x3 ? acc[tm][tn][r] * (ia * ib[tn]) : acc[tm][tn][r]) + bv;
          if (RELU) v = fmaxf(v, 0.f);
          y[row * ldy + col] = v;
          mx = fmaxf(mx, fabsf(v));
        }
      }
      rmx[tm][r] = mx;
    }
  if (out_rowmax != nullptr) {
    // Row maxima for the NEXT layer's operand scales: the 32 lanes of a half wave hold 32 columns of the same 32 rows; a halving
    // butterfly (lane bit i <-> row-slot bit 4 - i; DPP inside quads, ds_swizzle across) leaves every lane with the maximum of
    // ONE row over the wave's 64 columns, and one atomic max per lane (non-negative floats order like their bit patterns) merges
    // the wave columns and column tiles.  out_rowmax must be zeroed before the launch.
    const bool lb0 = lane & 1, lb1 = lane & 2, lb2 = lane & 4, lb3 = lane & 8, lb4 = lane & 16;
    float v16[16], v8[8], v4[4], v2[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {       // slot = tm * 16 + r; bit 4 (tm) <-> lane bit 0
      const float keep = lb0 ? rmx[1][i] : rmx[0][i], send = lb0 ? rmx[0][i] : rmx[1][i];
      v16[i] = fmaxf(keep, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true)));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float keep = lb1 ? v16[i + 8] : v16[i], send = lb1 ? v16[i] : v16[i + 8];
      v8[i] = fmaxf(keep, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0x4E, 0xF, 0xF, true)));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float keep = lb2 ? v8[i + 4] : v8[i], send = lb2 ? v8[i] : v8[i + 4];
      v4[i] = fmaxf(keep, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, send), 0x101F)));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float keep = lb3 ? v4[i + 2] : v4[i], send = lb3 ? v4[i] : v4[i + 2];
      v2[i] = fmaxf(keep, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, send), 0x201F)));
    }
    const float keep = lb4 ? v2[1] : v2[0], send = lb4 ? v2[0] : v2[1];
    const float mrow = fmaxf(keep, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, send), 0x401F)));
    const int slot = ((lane & 1) << 4) | ((lane & 2) << 2) | (lane & 4) | ((lane & 8) >> 2) | ((lane & 16) >> 4);
    const int64_t row = row0 + acc_row(wm, slot >> 4, slot & 15, lane);
    if (row < g.m) atomicMax(reinterpret_cast<unsigned*>(out_rowmax + row), __float_as_uint(mrow));
  }
}

// max |x| of every row of a [rows, ld] matrix (the static operand scales of the weights; one wave per row)
__global__ void __launch_bounds__(256) k_row_absmax(const float* __restrict__ src, int rows, int k, int64_t ld, float* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float m = 0.f;
  for (int i = lane; i < k; i += 64) m = fmaxf(m, fabsf(src[(int64_t)row * ld + i]));
  m = sdg_wave_max(m);
  if (lane == 0) out[row] = m;
}

// upper bound of max |x| over the encoded ray input (a12): the raw coordinates, and 1 for the sin / cos features
__global__ void __launch_bounds__(256) k_ray_input_bound(const float* __restrict__ ori, const float* __restrict__ dir, const float* __restrict__ rgb,
                                                        int64_t R, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  float m = 1.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) m = fmaxf(m, fmaxf(fabsf(ori[3 * i + c]), fmaxf(fabsf(dir[3 * i + c]), fabsf(rgb[3 * i + c]))));
  out[i] = m;
}

// Split-K for GEMMs with few output tiles and a long K (the camera-up CNN as im2col: M = images x positions <= a few hundred,
// K = 9600): grid.y slices K, every slice writes its partial tile to part[slice][M][N] and k_splitk_finish adds the slices in
// ascending order (deterministic), then bias / ReLU.
template <int MMA>
__global__ void __launch_bounds__(256, 2) k_linear_splitk(GemmOperands g, int k_slice, float* __restrict__ part, unsigned n_tiles,
                                                          unsigned total_tiles) {
  __shared__ __attribute__((aligned(16))) char smem[TileSmem<MMA>::kBytes];
  const unsigned w = xcd_remap(blockIdx.x, total_tiles);
  const int64_t row0 = (int64_t)(w / n_tiles) * 128;
  const int64_t col0 = (int64_t)(w % n_tiles) * kBN;
  const int kb = (int)blockIdx.y * k_slice;
  const int kl = min(k_slice, g.k - kb);
  g.a0 += kb;
  g.b += kb;
  g.k = kl;
  g.k0 = kl;
  f32x16 acc[2][2];
  gemm_tile<MMA>(g, row0, col0, smem, acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  float* out = part + (int64_t)blockIdx.y * g.m * g.n;
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int64_t col = col0 + acc_col(wn, tn, lane);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + acc_row(wm, tm, r, lane);
        if (row < g.m && col < g.n) out[row * g.n + col] = acc[tm][tn][r];
      }
  }
}
__global__ void __launch_bounds__(256) k_splitk_finish(const float* __restrict__ part, int slices, int64_t m, int n,
                                                       const float* __restrict__ bias, int relu, float* __restrict__ y, int64_t ldy) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m * n) return;
  const int64_t row = i / n;
  const int col = (int)(i - row * n);
  float v = 0.f;
  for (int s = 0; s < slices; ++s) v += part[(int64_t)s * m * n + i];
  if (bias) v += bias[col];
  if (relu) v = fmaxf(v, 0.f);
  y[row * ldy + col] = v;
}

int resolve_mma(int mode) {
  if (mode == SIXDGS_MMA_F32) return mode;
  return SIXDGS_MMA_BF16X6;  // DEFAULT, BF16X6 and F16X3: generic GEMMs carry no row maxima, so no scaled-fp16 variant for them
}

// scaled fp16 x 3 (operands with row maxima: the ray MLP chain); out_rowmax (zeroed by the caller) receives max |y| per row
int launch_linear_f3(const GemmOperands& g, const float* bias, bool relu, float* y, int64_t ldy, float* out_rowmax, hipStream_t s) {
  const int64_t m_tiles = sdg_cdiv(g.m, 128), n_tiles = sdg_cdiv(g.n, kBN);
  const int64_t total = m_tiles * n_tiles;
  if (total <= 0) return 0;
  if (total > 0x7fffffffLL || !g.amax0 || !g.bmax) return SIXDGS_E_BADARG;
  const dim3 grid((unsigned)total), blk(256);
  if (relu) hipLaunchKernelGGL((k_linear<kMmaF16x3, true>), grid, blk, 0, s, g, bias, y, ldy, (unsigned)n_tiles, (unsigned)total, out_rowmax);
  else hipLaunchKernelGGL((k_linear<kMmaF16x3, false>), grid, blk, 0, s, g, bias, y, ldy, (unsigned)n_tiles, (unsigned)total, out_rowmax);
  SDG_LAUNCH_OK();
  return 0;
}

int launch_linear(const GemmOperands& g, const float* bias, bool relu, float* y, int64_t ldy, hipStream_t s, int mma) {
  const int64_t m_tiles = sdg_cdiv(g.m, 128), n_tiles = sdg_cdiv(g.n, kBN);
  const int64_t total = m_tiles * n_tiles;
  if (total <= 0) return 0;
  if (total > 0x7fffffffLL) return SIXDGS_E_BADARG;
  const dim3 grid((unsigned)total), blk(256);
  const unsigned nt = (unsigned)n_tiles, tt = (unsigned)total;
  if (resolve_mma(mma) == SIXDGS_MMA_BF16X6) {
    if (relu) hipLaunchKernelGGL((k_linear<kMmaBf16x6, true>), grid, blk, 0, s, g, bias, y, ldy, nt, tt, (float*)nullptr);
    else hipLaunchKernelGGL((k_linear<kMmaBf16x6, false>), grid, blk, 0, s, g, bias, y, ldy, nt, tt, (float*)nullptr);
  } else {
    if (relu) hipLaunchKernelGGL((k_linear<kMmaF32, true>), grid, blk, 0, s, g, bias, y, ldy, nt, tt, (float*)nullptr);
    else hipLaunchKernelGGL((k_linear<kMmaF32, false>), grid, blk, 0, s, g, bias, y, ldy, nt, tt, (float*)nullptr);
  }
  SDG_LAUNCH_OK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// a12: x[R,144]
// ------------------------------------------------------------------------------------------------
// One workgroup = 64 rays; one sincosf per (component, frequency) -- 66 per ray instead of 132 separate sinf / cosf calls with their own
// argument reductions -- staged in LDS [ray][145] and written as the block's 64 x 144 contiguous floats.
__global__ void __launch_bounds__(256) k_ray_encode(const float* __restrict__ ori, const float* __restrict__ dir,
                                                     const float* __restrict__ rgb, int64_t R, float* __restrict__ x) {
  constexpr int kRays = 64, kLd = SIXDGS_RAY_IN_PAD;
  __shared__ float X[kRays][kLd + 1];
  __shared__ float src[kRays][9];      // p, d, c
  const int tid = threadIdx.x;
  const int64_t ray0 = (int64_t)blockIdx.x * kRays;
  const int n = (int)min((int64_t)kRays, R - ray0);
  for (int i = tid; i < kRays * 9; i += 256) {
    const int r = i / 9, a = i - r * 9;
    const int64_t gr = ray0 + min(r, n - 1);
    const float* q = a < 3 ? ori : (a < 6 ? dir : rgb);
    const float v = q[3 * gr + (a % 3)];
    src[r][a] = v;
    X[r][a] = v;
  }
  for (int i = tid; i < kRays * (kLd - 141); i += 256) X[i / (kLd - 141)][141 + i % (kLd - 141)] = 0.f;
  __syncthreads();
  for (int i = tid; i < kRays * 66; i += 256) {
    const int r = i % kRays, k = i / kRays;                   // k: p 0..23 (8 frequencies x 3), d 24..47, c 48..65 (6 x 3)
    const int grp = k < 24 ? 0 : (k < 48 ? 1 : 2);
    const int F = grp == 2 ? 6 : 8, o = k - grp * 24;
    const int comp = o / F, f = o - comp * F;
    const float v = src[r][grp * 3 + comp] * (float)(1 << f);
    float sn, cs;
    sincosf(v, &sn, &cs);
    const int col = 9 + grp * 48 + comp * F + f;              // same columns as ray_input_element (device_math.h)
    X[r][col] = sn;
    X[r][col + 3 * F] = cs;
  }
  __syncthreads();
  float* dst = x + ray0 * kLd;
  for (int i = tid; i < n * kLd; i += 256) dst[i] = X[i / kLd][i % kLd];
}

// ------------------------------------------------------------------------------------------------
// weight packing: dst[n][kd] = k < ks ? src[n][k] : 0 ;  transposed variant dst[k][n]
// ------------------------------------------------------------------------------------------------
__global__ void k_pad_rows(const float* __restrict__ src, int n, int ks, int kd, float* __restrict__ dst) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * kd) return;
  int r = i / kd, k = i - r * kd;
  dst[i] = k < ks ? src[(int64_t)r * ks + k] : 0.f;
}
__global__ void k_pad_transpose(const float* __restrict__ src, int n, int ks, int kd, float* __restrict__ dst) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;  // dst index [k][n]
  if (i >= n * kd) return;
  int k = i / n, r = i - k * n;
  dst[i] = k < ks ? src[(int64_t)r * ks + k] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// q_proj: q[b][t][n] = sum_k tok[b][t][k] wq_t[k][n] + bq[n]; 398-wide token rows are not 16-B
// aligned, so this small layer (78 MFLOP per image) runs on the VALU from an LDS copy of the rows.
// ------------------------------------------------------------------------------------------------
constexpr int kQTok = 8;
__global__ void __launch_bounds__(SIXDGS_D) k_q_proj(const float* __restrict__ tokens, const int* __restrict__ n_tok,
                                                      const float* __restrict__ wq_t, const float* __restrict__ bq,
                                                      float* __restrict__ q) {
  __shared__ float tk[kQTok][SIXDGS_TOK_IN + 2];
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * kQTok;
  const int nt = n_tok[b];
  const int n = threadIdx.x;
  float* qo = q + ((int64_t)b * SIXDGS_MAX_TOKENS + t0) * SIXDGS_D;
  if (t0 >= nt) {  // rows beyond the image's token count: defined (zero) but never consumed
    for (int t = 0; t < kQTok; ++t) qo[(int64_t)t * SIXDGS_D + n] = 0.f;
    return;
  }
  const float* src = tokens + ((int64_t)b * SIXDGS_MAX_TOKENS + t0) * SIXDGS_TOK_IN;
  for (int i = threadIdx.x; i < kQTok * SIXDGS_TOK_IN; i += blockDim.x) {
    int t = i / SIXDGS_TOK_IN, k = i - t * SIXDGS_TOK_IN;
    tk[t][k] = (t0 + t < nt) ? src[i] : 0.f;
  }
  __syncthreads();
  float acc[kQTok];
#pragma unroll
  for (int t = 0; t < kQTok; ++t) acc[t] = 0.f;
  for (int k = 0; k < SIXDGS_TOK_IN; ++k) {
    const float w = wq_t[(int64_t)k * SIXDGS_D + n];
#pragma unroll
    for (int t = 0; t < kQTok; ++t) acc[t] = fmaf(tk[t][k], w, acc[t]);
  }
  const float bv = bq[n];
#pragma unroll
  for (int t = 0; t < kQTok; ++t) qo[(int64_t)t * SIXDGS_D + n] = (t0 + t < nt) ? acc[t] + bv : 0.f;
}

// Layer 4 of the ray MLP has no non-linearity behind it (ray_preprocessor.py:27-31,46) and k_proj follows at once
// (our_multihead_attention.py:74): K = Wk (W4 h3 + b4) + bk = (Wk W4) h3 + (Wk b4 + bk).  The composite is formed ONCE per set of weights,
// in fp64 (products of two fp32 values are exact in fp64; 384 of them accumulate to ~1e-16 relative), and rounded to fp32 once.
//   w4k[i][j] = sum_m wk[i][m] w4[m][j]     b4k[i] = sum_m wk[i][m] b4[m] + bk[i]
__global__ void __launch_bounds__(256) k_compose_w4k(const float* __restrict__ wk, const float* __restrict__ bk, const float* __restrict__ w4,
                                                     const float* __restrict__ b4, float* __restrict__ w4k, float* __restrict__ b4k) {
  __shared__ float wrow[SIXDGS_D];
  const int i = blockIdx.x;                        // output row
  for (int m = threadIdx.x; m < SIXDGS_D; m += 256) wrow[m] = wk[(int64_t)i * SIXDGS_D + m];
  __syncthreads();
  for (int j = threadIdx.x; j < SIXDGS_HID + 1; j += 256) {      // column SIXDGS_HID: the bias
    double acc = 0.0;
    if (j < SIXDGS_HID) {
      for (int m = 0; m < SIXDGS_D; ++m) acc = fma((double)wrow[m], (double)w4[(int64_t)m * SIXDGS_HID + j], acc);
      w4k[(int64_t)i * SIXDGS_HID + j] = (float)acc;
    } else {
      for (int m = 0; m < SIXDGS_D; ++m) acc = fma((double)wrow[m], (double)b4[m], acc);
      b4k[i] = (float)(acc + (double)bk[i]);
    }
  }
}

constexpr size_t pad64(size_t x) { return (x + 63) / 64 * 64; }
constexpr size_t kOffW1 = 0;
constexpr size_t kOffB1 = kOffW1 + pad64(512 * 144);
constexpr size_t kOffW2 = kOffB1 + pad64(512);
constexpr size_t kOffB2 = kOffW2 + pad64(512 * 512);
constexpr size_t kOffW3 = kOffB2 + pad64(512);
constexpr size_t kOffB3 = kOffW3 + pad64(512 * 656);
constexpr size_t kOffW4 = kOffB3 + pad64(512);
constexpr size_t kOffB4 = kOffW4 + pad64(384 * 512);
constexpr size_t kOffWk = kOffB4 + pad64(384);
constexpr size_t kOffBk = kOffWk + pad64(384 * 384);
constexpr size_t kOffWq = kOffBk + pad64(384);
constexpr size_t kOffBq = kOffWq + pad64(400 * 384);
constexpr size_t kOffM1 = kOffBq + pad64(384);      // max |w| of every row of W1 .. Wk (operand scales of the scaled-fp16 GEMMs)
constexpr size_t kOffM2 = kOffM1 + pad64(512);
constexpr size_t kOffM3 = kOffM2 + pad64(512);
constexpr size_t kOffM4 = kOffM3 + pad64(512);
constexpr size_t kOffMk = kOffM4 + pad64(384);
constexpr size_t kOffW4k = kOffMk + pad64(384);     // Wk W4 [384][512], Wk b4 + bk [384] and its row maxima: k_proj folded into layer 4 (round 5)
constexpr size_t kOffB4k = kOffW4k + pad64(384 * 512);
constexpr size_t kOffM4k = kOffB4k + pad64(384);
constexpr size_t kOffPlanes = kOffM4k + pad64(384);  // scaled fp16 planes of W1 .. Wk, Wk W4 for the plane-to-plane chain (dense.hip)
constexpr size_t kPlaneFloats = (512 * 5 + 512 * 16 + 512 * 21 + 384 * 16 + 384 * 12 + 384 * 16) * 32;
constexpr size_t kPackedFloats = kOffPlanes + pad64(kPlaneFloats);

// per ray of a chunk: fp32-operand chain x, h1, h2 + three row-maximum arrays = 1171 floats; plane-to-plane chain fp32 keys (384 floats)
// + planes and shifts (4776 B): the larger of the two
constexpr size_t kChunkFloatsPerRay = 1640;      // 1536 B of fp32 keys + dense_chain_bytes_per_ray() (asserted in sixdgs_ray_keys_ex)
constexpr int64_t kChunkSlackRays = 128;      // the plane-to-plane chain keeps whole granules of 128 rays: a ragged last chunk rounds up

}  // namespace

extern "C" {

int sixdgs_abi_version(void) { return SIXDGS_ABI_VERSION; }

const char* sixdgs_error_string(int status) {
  if (status == 0) return "ok";
  if (status == SIXDGS_E_BADARG) return "sixdgs: bad argument";
  if (status == SIXDGS_E_WORKSPACE) return "sixdgs: workspace too small";
  if (status == SIXDGS_E_UNSUPPORTED) return "sixdgs: unsupported configuration";
  if (status > 0) return hipGetErrorString((hipError_t)status);
  return "sixdgs: unknown error";
}

size_t sixdgs_packed_weights_floats(void) { return kPackedFloats; }

int sixdgs_pack_weights(const float* mlp0_w, const float* mlp0_b, const float* mlp2_w, const float* mlp2_b, const float* mlp2_0_w,
                        const float* mlp2_0_b, const float* mlp2_2_w, const float* mlp2_2_b, const float* kproj_w,
                        const float* kproj_b, const float* qproj_w, const float* qproj_b, float* packed,
                        sixdgs_scorer_weights* out, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(mlp0_w && mlp0_b && mlp2_w && mlp2_b && mlp2_0_w && mlp2_0_b && mlp2_2_w && mlp2_2_b && kproj_w && kproj_b &&
                qproj_w && qproj_b && packed && out);
  hipStream_t s = sdg_stream(stream);
  auto pad = [&](const float* src, int n, int ks, int kd, size_t off) {
    hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)sdg_cdiv((int64_t)n * kd, 256)), dim3(256), 0, s, src, n, ks, kd, packed + off);
  };
  pad(mlp0_w, 512, 141, 144, kOffW1);
  pad(mlp0_b, 1, 512, 512, kOffB1);
  pad(mlp2_w, 512, 512, 512, kOffW2);
  pad(mlp2_b, 1, 512, 512, kOffB2);
  pad(mlp2_0_w, 512, 653, 656, kOffW3);
  pad(mlp2_0_b, 1, 512, 512, kOffB3);
  pad(mlp2_2_w, 384, 512, 512, kOffW4);
  pad(mlp2_2_b, 1, 384, 384, kOffB4);
  pad(kproj_w, 384, 384, 384, kOffWk);
  pad(kproj_b, 1, 384, 384, kOffBk);
  hipLaunchKernelGGL(k_pad_transpose, dim3((unsigned)sdg_cdiv(384 * 400, 256)), dim3(256), 0, s, qproj_w, 384, 398, 400,
                     packed + kOffWq);
  pad(qproj_b, 1, 384, 384, kOffBq);
  auto rowmax = [&](size_t woff, int rows, int k, size_t moff) {
    hipLaunchKernelGGL(k_row_absmax, dim3((unsigned)sdg_cdiv(rows, 4)), dim3(256), 0, s, packed + woff, rows, k, (int64_t)k, packed + moff);
  };
  rowmax(kOffW1, 512, 144, kOffM1);
  rowmax(kOffW2, 512, 512, kOffM2);
  rowmax(kOffW3, 512, 656, kOffM3);
  rowmax(kOffW4, 384, 512, kOffM4);
  rowmax(kOffWk, 384, 384, kOffMk);
  hipLaunchKernelGGL(k_compose_w4k, dim3(SIXDGS_D), dim3(256), 0, s, packed + kOffWk, packed + kOffBk, packed + kOffW4, packed + kOffB4, packed + kOffW4k,
                     packed + kOffB4k);
  rowmax(kOffW4k, 384, 512, kOffM4k);
  SDG_LAUNCH_OK();
  out->w1 = packed + kOffW1; out->b1 = packed + kOffB1;
  out->w2 = packed + kOffW2; out->b2 = packed + kOffB2;
  out->w3 = packed + kOffW3; out->b3 = packed + kOffB3;
  out->w4 = packed + kOffW4; out->b4 = packed + kOffB4;
  out->wk = packed + kOffWk; out->bk = packed + kOffBk;
  out->wq = packed + kOffWq; out->bq = packed + kOffBq;
  out->m1 = packed + kOffM1; out->m2 = packed + kOffM2; out->m3 = packed + kOffM3; out->m4 = packed + kOffM4; out->mk = packed + kOffMk;
  out->w4k = packed + kOffW4k; out->b4k = packed + kOffB4k; out->m4k = packed + kOffM4k;
  out->planes = packed + kOffPlanes;
  return dense_pack_weight_planes(out, reinterpret_cast<char*>(packed + kOffPlanes), s);
}

int sixdgs_ray_encode(const float* ori, const float* dir, const float* rgb, int64_t r, float* x, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(r >= 0);
  if (r == 0) return 0;
  SDG_CHECK_ARG(ori && dir && rgb && x);
  hipLaunchKernelGGL(k_ray_encode, dim3((unsigned)sdg_cdiv(r, 64)), dim3(256), 0, sdg_stream(stream), ori, dir,
                     rgb, r, x);
  SDG_LAUNCH_OK();
  return 0;
}

int sixdgs_linear(const float* x, int64_t m, int k, int64_t ldx, const float* w, int64_t ldw, const float* b, int n, int relu,
                  float* y, int64_t ldy, sixdgs_stream_t stream) {
  return sixdgs_linear_ex(x, m, k, ldx, w, ldw, b, n, relu, y, ldy, stream, SIXDGS_MMA_DEFAULT);
}

int sixdgs_linear_ex(const float* x, int64_t m, int k, int64_t ldx, const float* w, int64_t ldw, const float* b, int n, int relu,
                     float* y, int64_t ldy, sixdgs_stream_t stream, int mma_mode) {
  SDG_CHECK_ARG(m >= 0 && n > 0 && k > 0 && (k % 4) == 0 && (ldx % 4) == 0 && (ldw % 4) == 0 && ldx >= k && ldw >= k && ldy >= n);
  if (m == 0) return 0;
  SDG_CHECK_ARG(x && w && y);
  SDG_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0);
  GemmOperands g = {x, nullptr, w, ldx, 0, ldw, m, n, k, k};
  return launch_linear(g, b, relu != 0, y, ldy, sdg_stream(stream), mma_mode);
}

size_t sixdgs_linear_splitk_workspace_bytes(int64_t m, int n, int slices) {
  return sdg_align((size_t)(m > 0 ? m : 1) * (size_t)(n > 0 ? n : 1) * (size_t)(slices > 0 ? slices : 1) * sizeof(float));
}

int sixdgs_linear_splitk(const float* x, int64_t m, int k, int64_t ldx, const float* w, int64_t ldw, const float* b, int n, int relu,
                         float* y, int64_t ldy, int slices, void* ws, size_t ws_bytes, sixdgs_stream_t stream, int mma_mode) {
  SDG_CHECK_ARG(m >= 0 && n > 0 && k > 0 && (k % 4) == 0 && (ldx % 4) == 0 && (ldw % 4) == 0 && ldx >= k && ldw >= k && ldy >= n);
  SDG_CHECK_ARG(slices >= 1 && slices <= 1024);
  if (m == 0) return 0;
  SDG_CHECK_ARG(x && w && y && ws && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)ws % 16) == 0);
  if (ws_bytes < sixdgs_linear_splitk_workspace_bytes(m, n, slices)) return SIXDGS_E_WORKSPACE;
  int k_slice = (int)sdg_cdiv(sdg_cdiv(k, slices), 32) * 32;        // whole 32-wide k-slabs per slice
  const int used = (int)sdg_cdiv(k, k_slice);
  GemmOperands g = {x, nullptr, w, ldx, 0, ldw, m, n, k, k};
  const int64_t m_tiles = sdg_cdiv(m, 128), n_tiles = sdg_cdiv(n, kBN);
  const int64_t total = m_tiles * n_tiles;
  if (total > 0x7fffffffLL) return SIXDGS_E_BADARG;
  hipStream_t s = sdg_stream(stream);
  const dim3 grid((unsigned)total, (unsigned)used), blk(256);
  if (resolve_mma(mma_mode) == SIXDGS_MMA_BF16X6)
    hipLaunchKernelGGL((k_linear_splitk<kMmaBf16x6>), grid, blk, 0, s, g, k_slice, (float*)ws, (unsigned)n_tiles, (unsigned)total);
  else
    hipLaunchKernelGGL((k_linear_splitk<kMmaF32>), grid, blk, 0, s, g, k_slice, (float*)ws, (unsigned)n_tiles, (unsigned)total);
  hipLaunchKernelGGL(k_splitk_finish, dim3((unsigned)sdg_cdiv(m * n, 256)), dim3(256), 0, s, (const float*)ws, used, m, n, b, relu, y, ldy);
  SDG_LAUNCH_OK();
  return 0;
}

size_t sixdgs_ray_keys_workspace_bytes(int64_t r, int64_t max_chunk) {
  if (max_chunk <= 0) max_chunk = 262144;
  int64_t c = r < max_chunk ? r : max_chunk;
  if (c < 1) c = 1;
  return sdg_align((size_t)(c + kChunkSlackRays) * kChunkFloatsPerRay * sizeof(float));
}

int sixdgs_ray_keys(const float* ori, const float* dir, const float* rgb, int64_t r, const sixdgs_scorer_weights* w, float* feat,
                    float* key, void* ws, size_t ws_bytes, sixdgs_stream_t stream) {
  return sixdgs_ray_keys_ex(ori, dir, rgb, r, w, feat, key, nullptr, nullptr, nullptr, ws, ws_bytes, stream, nullptr, SIXDGS_MMA_DEFAULT);
}

int sixdgs_ray_keys_ex(const float* ori, const float* dir, const float* rgb, int64_t r, const sixdgs_scorer_weights* w, float* feat,
                       float* key, void* key_planes, float* key_inv_scale, float* d_key_norm_max, void* ws, size_t ws_bytes,
                       sixdgs_stream_t stream, sixdgs_profile* prof, int mma_mode) {
  SDG_CHECK_ARG(r >= 0 && w);
  if (r == 0) return 0;
  SDG_CHECK_ARG(ori && dir && rgb && ws && (feat || key || key_planes));
  const bool f16 = key_planes && (mma_mode == SIXDGS_MMA_F16X3 || mma_mode == SIXDGS_MMA_F16X3_L32 || mma_mode == SIXDGS_MMA_DEFAULT);
  SDG_CHECK_ARG(!f16 || key_inv_scale);
  if (key_planes && !f16) return SIXDGS_E_UNSUPPORTED;      // key planes exist in one format: scaled fp16 (the three-plane bf16 format went with its scorer kernel, round 6)
  if (kChunkFloatsPerRay * sizeof(float) < (size_t)SIXDGS_D * sizeof(float) + dense_chain_bytes_per_ray()) return SIXDGS_E_WORKSPACE;
  const int64_t chunk_cap = (int64_t)(ws_bytes / (kChunkFloatsPerRay * sizeof(float))) - kChunkSlackRays;
  if (chunk_cap < 1) return SIXDGS_E_WORKSPACE;
  const int64_t chunk = chunk_cap < r ? (chunk_cap >= 128 ? chunk_cap / 128 * 128 : chunk_cap) : r;
  if (f16 && chunk < r && (chunk % 128) != 0) return SIXDGS_E_WORKSPACE;   // scale tiles must not straddle chunks
  hipStream_t s = sdg_stream(stream);
  float* x = (float*)ws;
  float* h1 = x + chunk * SIXDGS_RAY_IN_PAD;
  float* h2 = h1 + chunk * SIXDGS_HID;
  float* rmx = h2 + chunk * SIXDGS_HID;        // row maxima: encoded input, and two buffers that alternate through the layers
  float* rma = rmx + chunk;
  float* rmb = rma + chunk;
  // the dense layers run scaled fp16 x 3 (row-scaled operands, 3 MFMA terms) in the default / F16X3 modes, bf16 x 6 or fp32 otherwise
  const bool f3 = mma_mode == SIXDGS_MMA_DEFAULT || mma_mode == SIXDGS_MMA_F16X3 || mma_mode == SIXDGS_MMA_F16X3_L32;
  for (int64_t r0 = 0; r0 < r; r0 += chunk) {
    const int64_t m = (r - r0) < chunk ? (r - r0) : chunk;
    // algorithmic work per ray: ray MLP 1 730 560 + k_proj 294 912 FLOP; 36 B in, 1536 B key out
    SdgProfileScope scope(prof, s, (double)m * ((key || key_planes) ? 2025472.0 : 1730560.0), (double)m * (36.0 + 1536.0));
    int st = 0;
    float* f = feat ? feat + r0 * SIXDGS_D : h2;
    // without a caller buffer for fp32 keys the chunk lands in h1 (free after layer 4) and only the planes persist
    float* kdst = key ? key + r0 * SIXDGS_D : h1;     // (the plane-to-plane chain below redirects it)
    const bool want_key = key || key_planes;
    bool fused_planes = false;
    if (f3 && want_key && !feat && w->planes) {
      // plane-to-plane chain (dense.hip): no fp32 activation through HBM, nothing split in a main loop; fp32 keys land in the
      // caller's buffer or at the head of the workspace, the planes and shifts of the layers behind them
      float* kd = key ? key + r0 * SIXDGS_D : x;
      char* pws = reinterpret_cast<char*>(x + (size_t)chunk * SIXDGS_D);
      fused_planes = f16 && !key;      // only the planes are wanted: k_proj writes them itself (identical to the split of its fp32 keys)
      if ((st = dense_chain(ori + 3 * r0, dir + 3 * r0, rgb + 3 * r0, m, w, reinterpret_cast<const char*>(w->planes), kd,
                            fused_planes ? (char*)key_planes + (size_t)r0 * 1536 : nullptr, fused_planes ? key_inv_scale + r0 / 128 : nullptr,
                            fused_planes ? d_key_norm_max : nullptr, pws, s)))
        return st;
      kdst = kd;
    } else if (f3) {
      if ((st = sixdgs_ray_encode(ori + 3 * r0, dir + 3 * r0, rgb + 3 * r0, m, x, stream))) return st;
      hipLaunchKernelGGL(k_ray_input_bound, dim3((unsigned)sdg_cdiv(m, 256)), dim3(256), 0, s, ori + 3 * r0, dir + 3 * r0, rgb + 3 * r0, m, rmx);
      auto zero = [&](float* p) { return hipMemsetAsync(p, 0, (size_t)m * sizeof(float), s); };
      if (zero(rma) != hipSuccess) return (int)hipGetLastError();
      GemmOperands g1 = {x, nullptr, w->w1, SIXDGS_RAY_IN_PAD, 0, SIXDGS_RAY_IN_PAD, m, SIXDGS_HID, SIXDGS_RAY_IN_PAD, SIXDGS_RAY_IN_PAD, rmx, nullptr, w->m1};
      if ((st = launch_linear_f3(g1, w->b1, true, h1, SIXDGS_HID, rma, s))) return st;
      if (zero(rmb) != hipSuccess) return (int)hipGetLastError();
      GemmOperands g2 = {h1, nullptr, w->w2, SIXDGS_HID, 0, SIXDGS_HID, m, SIXDGS_HID, SIXDGS_HID, SIXDGS_HID, rma, nullptr, w->m2};
      if ((st = launch_linear_f3(g2, w->b2, true, h2, SIXDGS_HID, rmb, s))) return st;
      // layer 3 consumes the concatenation [h2, x] without materialising it (two A segments, one scale per row for both)
      if (zero(rma) != hipSuccess) return (int)hipGetLastError();
      GemmOperands g3 = {h2, x, w->w3, SIXDGS_HID, SIXDGS_RAY_IN_PAD, SIXDGS_HID + SIXDGS_RAY_IN_PAD, m, SIXDGS_HID,
                         SIXDGS_HID + SIXDGS_RAY_IN_PAD, SIXDGS_HID, rmb, rmx, w->m3};
      if ((st = launch_linear_f3(g3, w->b3, true, h1, SIXDGS_HID, rma, s))) return st;
      if (zero(rmb) != hipSuccess) return (int)hipGetLastError();
      GemmOperands g4 = {h1, nullptr, w->w4, SIXDGS_HID, 0, SIXDGS_HID, m, SIXDGS_D, SIXDGS_HID, SIXDGS_HID, rma, nullptr, w->m4};
      if ((st = launch_linear_f3(g4, w->b4, false, f, SIXDGS_D, want_key ? rmb : nullptr, s))) return st;
      if (want_key) {
        GemmOperands g5 = {f, nullptr, w->wk, SIXDGS_D, 0, SIXDGS_D, m, SIXDGS_D, SIXDGS_D, SIXDGS_D, rmb, nullptr, w->mk};
        if ((st = launch_linear_f3(g5, w->bk, false, kdst, SIXDGS_D, nullptr, s))) return st;
      }
    } else {
      if ((st = sixdgs_ray_encode(ori + 3 * r0, dir + 3 * r0, rgb + 3 * r0, m, x, stream))) return st;
      GemmOperands g1 = {x, nullptr, w->w1, SIXDGS_RAY_IN_PAD, 0, SIXDGS_RAY_IN_PAD, m, SIXDGS_HID, SIXDGS_RAY_IN_PAD, SIXDGS_RAY_IN_PAD};
      if ((st = launch_linear(g1, w->b1, true, h1, SIXDGS_HID, s, mma_mode))) return st;
      GemmOperands g2 = {h1, nullptr, w->w2, SIXDGS_HID, 0, SIXDGS_HID, m, SIXDGS_HID, SIXDGS_HID, SIXDGS_HID};
      if ((st = launch_linear(g2, w->b2, true, h2, SIXDGS_HID, s, mma_mode))) return st;
      // layer 3 consumes the concatenation [h2, x] without materialising it (two A segments)
      GemmOperands g3 = {h2, x, w->w3, SIXDGS_HID, SIXDGS_RAY_IN_PAD, SIXDGS_HID + SIXDGS_RAY_IN_PAD, m, SIXDGS_HID,
                         SIXDGS_HID + SIXDGS_RAY_IN_PAD, SIXDGS_HID};
      if ((st = launch_linear(g3, w->b3, true, h1, SIXDGS_HID, s, mma_mode))) return st;
      GemmOperands g4 = {h1, nullptr, w->w4, SIXDGS_HID, 0, SIXDGS_HID, m, SIXDGS_D, SIXDGS_HID, SIXDGS_HID};
      if ((st = launch_linear(g4, w->b4, false, f, SIXDGS_D, s, mma_mode))) return st;
      if (want_key) {
        GemmOperands g5 = {f, nullptr, w->wk, SIXDGS_D, 0, SIXDGS_D, m, SIXDGS_D, SIXDGS_D, SIXDGS_D};
        if ((st = launch_linear(g5, w->bk, false, kdst, SIXDGS_D, s, mma_mode))) return st;
      }
    }
    if (want_key && !fused_planes) {
      if (f16) {
        st = sixdgs_split_planes_f16(kdst, m, SIXDGS_D, (char*)key_planes + (size_t)r0 * 1536, key_inv_scale + r0 / 128, stream);
        if (!st && d_key_norm_max)      // (the fused path above takes the norms from the k_proj epilogue; here: one pass over the planes)
          st = sixdgs_key_planes_norm_max((char*)key_planes + (size_t)r0 * 1536, key_inv_scale + r0 / 128, m, d_key_norm_max, stream);
      }
      if (st) return st;
    }
  }
  return 0;
}

int sixdgs_q_proj(const float* tokens, const int32_t* d_n_tok, int batch, const sixdgs_scorer_weights* w, float* q,
                  sixdgs_stream_t stream) {
  SDG_CHECK_ARG(batch >= 0 && w);
  if (batch == 0) return 0;
  SDG_CHECK_ARG(tokens && d_n_tok && q);
  hipLaunchKernelGGL(k_q_proj, dim3(SIXDGS_MAX_TOKENS / kQTok, (unsigned)batch), dim3(SIXDGS_D), 0, sdg_stream(stream), tokens,
                     d_n_tok, w->wq, w->bq, q);
  SDG_LAUNCH_OK();
  return 0;
}

}  // extern "C"
