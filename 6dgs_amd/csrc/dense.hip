// dense.hip -- the ray MLP + k_proj chain on PRE-SPLIT operands (round 2): every layer reads scaled fp16 planes and writes scaled
// fp16 planes, so that nothing is split in a main loop and no fp32 activation goes through HBM.
//   replaces RayPreprocessor.forward (ray_preprocessor.py:36-46) + k_proj (our_multihead_attention.py:74) for the key cache
//   (sixdgs_ray_keys_ex with key planes and no feature output; the fp32-operand kernels of gemm.hip serve every other caller).
//
// Arithmetic: the scorer's -- x 2^s = h + l in two fp16 planes, three cross terms l*h + h*l + h*h on v_mfma_f32_32x32x16_f16, fp32
// accumulation (measured 1.0e-7 * sum|a||b|, below the fp32 MFMA chain).  Scales: weights one power of two per ROW (static, set when
// the weights are packed); activations one per (ray, block of 128 features) -- exactly what ONE workgroup produces, so the epilogue
// knows the block's maximum without any cross-workgroup traffic.  A block of 128 features is 4 k-slabs of the next layer: when its
// contraction crosses into a block with another scale the accumulators are multiplied by the (exact) power of two between the two.
//
// Orientation: C[feature][ray] = sum_k W[feature][k] A[ray][k] (weights = MFMA rows, rays = MFMA columns): a lane then owns ONE ray
// per column tile -- the per-ray rescale is one factor per lane -- and 4 consecutive features per register group, i.e. 8-byte pieces
// of a ray's plane row.
//
// Tile 128 features x 128 rays, 4 waves (2 x 2), K in slabs of 32: both operand slabs (2 planes x 64 B per row) go global ->
// registers -> LDS (row stride 144 B, conflict-free ds_read_b128) with NO arithmetic on the way; double-buffered, one barrier per
// slab.  Epilogue through LDS (the main loop's buffers are free by then): bias, ReLU, per-ray block maximum (lane, l ^ 32 partner,
// the two feature waves), scale, split, staged as [ray][4 slabs][2 planes][32] = 512 contiguous bytes per ray and written to the
// next layer's plane array as 16-byte pieces, 1 KiB contiguous per wave instruction.  The last layer (k_proj) leaves fp32 rows
// instead, which k_split_tiles_f16 turns into the scorer's per-128-RAY-tile planes.
#include "gemm_kernel.h"
#include "device_math.h"
#include "dense.h"

using namespace sdg;

namespace {

constexpr int kSlabB = 128;           // bytes of one (row, slab): plane h 64 B, plane l 64 B
constexpr int kPRow = 144;            // LDS row stride of a staged slab
constexpr int kPStage = 256 * kPRow;  // 128 weight rows + 128 ray rows
constexpr int kStRow = 528;           // staging row of the epilogue: 512 B of a ray + 16 B (with 512 the 32 lanes of a write hit one bank: 32-way conflict)
constexpr int kShMax = 40;            // activation shifts are clamped to +-40: the rescale between blocks stays far from overflow

__device__ __forceinline__ int p_shift(float m) {
  const int sh = f3_shift(m);
  return sh > kShMax ? kShMax : (sh < -kShMax ? -kShMax : sh);
}
__device__ __forceinline__ float pow2i(int e) { return __uint_as_float((unsigned)(127 + e) << 23); }

struct DenseArgs {
  const char* wp;        // weight planes [N][KS][128 B]
  const float* wmax;     // [N] max |w| per row (the weight row's scale is f3_scale(wmax[n]))
  const float* bias;     // [N]
  const char* a0;        // activation planes, segment 0: [M][ks0][128 B]
  const int* s0;         // shifts [M][g0] (g0 = ceil(ks0 / 4))
  const char* a1;        // segment 1 (or null): [M][ks1][128 B]
  const int* s1;         // [M][g1]
  int ks0, ks1, g0, g1;  // slabs / groups per segment (ks0 a multiple of 4 when a1 != null)
  int64_t m;             // rays
  int n;                 // features (multiple of 128)
  char* out_planes;      // [M][N/32][128 B] or null
  int* out_shift;        // [M][N/128]
  float* out_f32;        // [M][ldo] or null (exactly one of out_planes / out_f32)
  int64_t ldo;
  int relu;
};

constexpr int kMaxGroups = 6;
// bias, ReLU, per-(ray, block) scale, split, staging, coalesced stores of one 128-feature block of one 128-ray tile
__device__ __forceinline__ void dense_epilogue(const DenseArgs& A, char* smem, float (*wmaxs)[128], const float* cwb, const f32x16 (&acc)[2][2],
                                               const int (&shg)[2][kMaxGroups], int f0, int64_t ray0, int ks, int lane, int tid, int wm, int wn) {
  // ---- epilogue.  Lane: rays (tn) x features f0 + wm*64 + tm*32 + 8*(r>>2) + 4*(lane>>5) + (r&3).
  float v[2][2][16];
  float rmax[2] = {0.f, 0.f};
  const int glast = (ks - 1) >> 2;
  float ib[2];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    int shl = 0;
#pragma unroll
    for (int gg = 0; gg < kMaxGroups; ++gg) shl = gg == glast ? shg[tn][gg] : shl;
    ib[tn] = pow2i(-shl);
  }
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      // 4 consecutive features: their constants (reciprocal weight-row scale, bias) from the block's LDS copy.  Read from global memory
      // here, the 16 loads were waited for two at a time behind their issue: ~4 us of exposed L2 latency per block.
      const int fl4 = wm * 64 + tm * 32 + 8 * rg + 4 * (lane >> 5);
      const float4 iw4 = *reinterpret_cast<const float4*>(cwb + fl4), b4 = *reinterpret_cast<const float4*>(cwb + 128 + fl4);
      const float iw[4] = {iw4.x, iw4.y, iw4.z, iw4.w};
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          float x = acc[tm][tn][4 * rg + j] * (iw[j] * ib[tn]) + bb[j];
          if (A.relu) x = fmaxf(x, 0.f);
          v[tn][tm][4 * rg + j] = x;
          rmax[tn] = fmaxf(rmax[tn], fabsf(x));
        }
    }
  if (A.out_f32 != nullptr) {
    // fp32 rows through the staging tile [ray][128 features + pad] (66 KiB), then 16-byte pieces: 512 contiguous bytes per ray
    float* st = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int ray = wn * 64 + tn * 32 + (lane & 31), fl = wm * 64 + tm * 32 + 8 * rg + 4 * (lane >> 5);
          *reinterpret_cast<float4*>(st + ray * (kStRow / 4) + fl) = float4{v[tn][tm][4 * rg], v[tn][tm][4 * rg + 1], v[tn][tm][4 * rg + 2], v[tn][tm][4 * rg + 3]};
        }
    __syncthreads();
    for (int i = tid; i < 128 * 32; i += 256) {
      const int ray = i >> 5, c = i & 31;
      if (ray0 + ray < A.m) *reinterpret_cast<float4*>(A.out_f32 + (ray0 + ray) * A.ldo + f0 + c * 4) = reinterpret_cast<const float4*>(st + ray * (kStRow / 4))[c];
    }
  } else {
    // per-ray maximum of this block: the partner lane l ^ 32 holds the other features of the same ray, the other feature wave the rest
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) rmax[tn] = fmaxf(rmax[tn], __shfl_xor(rmax[tn], 32, 64));
    if (lane < 32) {
      wmaxs[wm][wn * 64 + lane] = rmax[0];
      wmaxs[wm][wn * 64 + 32 + lane] = rmax[1];
    }
    __syncthreads();
    char* stp = smem;       // staging [ray 128][slab 4][plane 2][32 fp16] = 512 B (+ 16 B pad) per ray
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int ray = wn * 64 + tn * 32 + (lane & 31);
      const int sh = p_shift(fmaxf(wmaxs[0][ray], wmaxs[1][ray]));
      const float sc = pow2i(sh);
      if (wm == 0 && lane < 32 && ray0 + ray < A.m) A.out_shift[(ray0 + ray) * (A.n >> 7) + (f0 >> 7)] = sh;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
          f16x4 h, l;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float x = v[tn][tm][4 * rg + j] * sc;
            const _Float16 hh = (_Float16)x;
            h[j] = hh;
            l[j] = (_Float16)(x - (float)hh);
          }
          const int fl = wm * 64 + tm * 32 + 8 * rg + 4 * (lane >> 5);     // feature within the block: slab fl >> 5, position fl & 31
          char* d = stp + ray * kStRow + (fl >> 5) * kSlabB + (fl & 31) * 2;
          *reinterpret_cast<f16x4*>(d) = h;
          *reinterpret_cast<f16x4*>(d + 64) = l;
        }
    }
    __syncthreads();
    const int nslab_out = A.n >> 5;
    for (int i = tid; i < 128 * 32; i += 256) {
      const int ray = i >> 5, c = i & 31;
      if (ray0 + ray < A.m)
        *reinterpret_cast<uint4*>(A.out_planes + ((ray0 + ray) * nslab_out + (f0 >> 5)) * kSlabB + c * 16) = reinterpret_cast<const uint4*>(stp + ray * kStRow)[c];
    }
  }
}

// One workgroup = one tile of 128 rays, ALL feature blocks of the layer one after the other (the per-workgroup set-up is paid once per
// ray tile).  Operand slabs are fetched TWO slabs ahead into two sets of staging registers: with one slab of look-ahead a load had to
// come back within the 24 MFMAs (0.4 us) of the slab in front of it, which L2 does not do -- the counters showed the matrix pipe busy
// 32 % of the cycles at an un-throttled 2.1 GHz, i.e. a latency-bound kernel.  The look-ahead runs across feature blocks, so a block's
// first slabs arrive during the previous block's epilogue.
__global__ void __launch_bounds__(256, 2) k_dense_planes(DenseArgs A, unsigned n_blocks, unsigned total_tiles) {
  __shared__ __attribute__((aligned(16))) char smem[2 * kPStage];      // 73 728 B: two slab stages; the epilogue staging aliases them
  __shared__ float wmaxs[2][128];                                      // per-ray maxima of the two feature waves
  __shared__ __attribute__((aligned(16))) float cwb[256];              // the current block's reciprocal weight-row scales [128] and biases [128]
  const unsigned w = xcd_remap(blockIdx.x, total_tiles);
  const int64_t ray0 = (int64_t)w * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;               // wm: feature half (MFMA rows), wn: ray half (MFMA columns)
  const int ks = A.ks0 + A.ks1;
  f32x16 acc[2][2];

  // loader: 8 lanes x 16 B cover the 128 bytes of one (row, slab) -- every wave instruction reads 8 FULL cache lines (a lane reading 64
  // contiguous bytes of its own row touches 32 lines per instruction); 4 instructions x 32 rows per operand.  Row offsets are 32-bit
  // (a chunk's planes stay below 4 GB) against uniform bases; the loads are unconditional (past the end the last slab is re-read):
  // with a conditional load hipcc kept the staging registers in scratch and waited for every load right behind its issue.
  const int lrow = tid >> 3, lc8 = tid & 7;
  unsigned woff[4], aoff0[4], aoff1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned lray = (unsigned)min((int64_t)(i * 32 + lrow), A.m - 1 - ray0);        // ray within the tile, clamped to the last valid one
    woff[i] = (unsigned)((i * 32 + lrow) * ks) * kSlabB + lc8 * 16;
    aoff0[i] = (lray * (unsigned)A.ks0) * kSlabB + lc8 * 16;
    aoff1[i] = (lray * (unsigned)A.ks1) * kSlabB + lc8 * 16;
  }
  const char* abase0 = A.a0 + (ray0 * A.ks0) * kSlabB;
  const char* abase1 = A.a1 ? A.a1 + (ray0 * A.ks1) * kSlabB : abase0;
  unsigned lb = 0;      // load cursor: feature block and slab of the next fetch
  int ls = 0;
  uint4 p0, p1, p2, p3, p4, p5, p6, p7;      // slab t + 1 (weights 0..3, rays 4..7)
  uint4 q0, q1, q2, q3, q4, q5, q6, q7;      // slab t + 2
#define SDG_FETCH(R0, R1, R2, R3, R4, R5, R6, R7)                                                            \
  {                                                                                                          \
    const bool seg1_ = ls >= A.ks0;                                                                          \
    const char* wb_ = A.wp + ((int64_t)lb * 128 * ks + ls) * kSlabB;                                         \
    const char* ab_ = seg1_ ? abase1 + (unsigned)(ls - A.ks0) * kSlabB : abase0 + (unsigned)ls * kSlabB;     \
    R0 = *reinterpret_cast<const uint4*>(wb_ + woff[0]);                                                     \
    R1 = *reinterpret_cast<const uint4*>(wb_ + woff[1]);                                                     \
    R2 = *reinterpret_cast<const uint4*>(wb_ + woff[2]);                                                     \
    R3 = *reinterpret_cast<const uint4*>(wb_ + woff[3]);                                                     \
    R4 = *reinterpret_cast<const uint4*>(ab_ + (seg1_ ? aoff1[0] : aoff0[0]));                               \
    R5 = *reinterpret_cast<const uint4*>(ab_ + (seg1_ ? aoff1[1] : aoff0[1]));                               \
    R6 = *reinterpret_cast<const uint4*>(ab_ + (seg1_ ? aoff1[2] : aoff0[2]));                               \
    R7 = *reinterpret_cast<const uint4*>(ab_ + (seg1_ ? aoff1[3] : aoff0[3]));                               \
    if (++ls == ks) {                                                                                        \
      ls = 0;                                                                                                \
      lb = lb + 1 < n_blocks ? lb + 1 : lb;                                                                  \
    }                                                                                                        \
  }
#define SDG_STAGE(BUF, R0, R1, R2, R3, R4, R5, R6, R7)                         \
  {                                                                             \
    char* dw_ = smem + (BUF) * kPStage + lrow * kPRow + lc8 * 16;               \
    *reinterpret_cast<uint4*>(dw_) = R0;                                        \
    *reinterpret_cast<uint4*>(dw_ + 32 * kPRow) = R1;                           \
    *reinterpret_cast<uint4*>(dw_ + 64 * kPRow) = R2;                           \
    *reinterpret_cast<uint4*>(dw_ + 96 * kPRow) = R3;                           \
    *reinterpret_cast<uint4*>(dw_ + 128 * kPRow) = R4;                          \
    *reinterpret_cast<uint4*>(dw_ + 160 * kPRow) = R5;                          \
    *reinterpret_cast<uint4*>(dw_ + 192 * kPRow) = R6;                          \
    *reinterpret_cast<uint4*>(dw_ + 224 * kPRow) = R7;                          \
  }
// the set holding slab t + 1 goes to the idle stage and takes slab t + 3; the two sets swap roles (par), no register is copied (a copy
// p = q would have to wait for q's loads: the look-ahead would be gone)
// One half of the two-fold unrolled slab loop: rescale if a new input block starts, the slab's 24 MFMAs, the block's epilogue when it was the
// block's last slab; then the set holding slab t + 1 (R...) goes to the other stage and takes slab t + 3.  The loop alternates HALF(p...)
// and HALF(q...) in straight-line code, so that the waits in front of the staging stores are counted (vmcnt(8): the other set's loads
// stay in flight); with a run-time choice of the set the compiler's counter analysis merges both orders and drains the queue.
#define SDG_HALF(R0, R1, R2, R3, R4, R5, R6, R7)                                                                       \
  {                                                                                                                     \
    if (s == 0) creg = tid < 128 ? A.wmax[blk * 128 + tid] : A.bias[blk * 128 + tid - 128];   /* in flight for the whole block */ \
    if ((s & 3) == 0 && s > 0) {                                                                                        \
      const int g = s >> 2;                                                                                             \
      _Pragma("unroll") for (int tn = 0; tn < 2; ++tn) {                                                                \
        int d = 0;                                                                                                      \
        _Pragma("unroll") for (int gg = 1; gg < kMaxGroups; ++gg) d = gg == g ? shg[tn][gg] - shg[tn][gg - 1] : d;      \
        const float fac = pow2i(d);                                                                                     \
        _Pragma("unroll") for (int tm = 0; tm < 2; ++tm)                                                                \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[tm][tn][r] *= fac;                                         \
      }                                                                                                                 \
    }                                                                                                                   \
    const char* sw = smem + buf * kPStage + (wm * 64 + frow) * kPRow + fk;                                              \
    const char* sr = smem + buf * kPStage + (128 + wn * 64 + frow) * kPRow + fk;                                        \
    _Pragma("unroll") for (int kstep = 0; kstep < 2; ++kstep) {                                                         \
      f16x8_t a[2][2], b[2][2];                                                                                         \
      _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                     \
        _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {                                                              \
          a[t][pl] = *reinterpret_cast<const f16x8_t*>(sw + t * 32 * kPRow + pl * 64 + kstep * 32);                     \
          b[t][pl] = *reinterpret_cast<const f16x8_t*>(sr + t * 32 * kPRow + pl * 64 + kstep * 32);                     \
        }                                                                                                               \
      /* (weight plane, ray plane): l*h, h*l, h*h -- smallest magnitude first */                                        \
      _Pragma("unroll") for (int qq = 0; qq < 3; ++qq) {                                                                \
        const int pa = qq == 0 ? 1 : 0, pb = qq == 1 ? 1 : 0;                                                           \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][pa], b[0][pb], acc[0][0], 0, 0, 0);                     \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][pa], b[1][pb], acc[0][1], 0, 0, 0);                     \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][pa], b[0][pb], acc[1][0], 0, 0, 0);                     \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][pa], b[1][pb], acc[1][1], 0, 0, 0);                     \
      }                                                                                                                 \
    }                                                                                                                   \
    int nbuf = buf ^ 1;                                                                                                 \
    if (s + 1 == ks) {     /* the block is complete: its epilogue borrows both stages */                                \
      cwb[tid] = tid < 128 ? f3_inv_scale(creg) : creg;                                                                 \
      __syncthreads();                                                                                                  \
      dense_epilogue(A, smem, wmaxs, cwb, acc, shg, (int)blk * 128, ray0, ks, lane, tid, wm, wn);                       \
      __syncthreads();                                                                                                  \
      _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                                  \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                                \
          _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) acc[i_][j_][r_] = 0.f;                                      \
      s = -1;                                                                                                           \
      ++blk;                                                                                                            \
      nbuf = 0;                                                                                                         \
    }                                                                                                                   \
    SDG_STAGE(nbuf, R0, R1, R2, R3, R4, R5, R6, R7)                                                                     \
    SDG_FETCH(R0, R1, R2, R3, R4, R5, R6, R7)                                                                           \
    __syncthreads();                                                                                                    \
    buf = nbuf;                                                                                                         \
    ++s;                                                                                                                \
  }

  // this lane's two rays (column tiles tn = 0, 1) and the shifts of all their input blocks (<= 6: fetched once, not inside the loop)
  int shg[2][kMaxGroups];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int64_t cray = min(ray0 + wn * 64 + tn * 32 + (lane & 31), A.m - 1);
#pragma unroll
    for (int g = 0; g < kMaxGroups; ++g)
      shg[tn][g] = g < A.g0 ? A.s0[cray * A.g0 + g] : (g < A.g0 + A.g1 ? A.s1[cray * A.g1 + (g - A.g0)] : 0);
  }

  SDG_FETCH(p0, p1, p2, p3, p4, p5, p6, p7)        // slab 0
  SDG_STAGE(0, p0, p1, p2, p3, p4, p5, p6, p7)
  SDG_FETCH(p0, p1, p2, p3, p4, p5, p6, p7)        // slab 1
  SDG_FETCH(q0, q1, q2, q3, q4, q5, q6, q7)        // slab 2
  __syncthreads();
  const int frow = lane & 31, fk = (lane >> 5) * 16;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int buf = 0, s = 0;
  unsigned blk = 0;
  float creg = 0.f;
  while (true) {                     // flattened over (feature block, slab); after the very last slab one surplus stage / fetch happens (harmless)
    SDG_HALF(p0, p1, p2, p3, p4, p5, p6, p7)
    if (blk >= n_blocks) break;
    SDG_HALF(q0, q1, q2, q3, q4, q5, q6, q7)
    if (blk >= n_blocks) break;
  }
#undef SDG_HALF
#undef SDG_FETCH
#undef SDG_STAGE
}

// a12 as planes: x[R][5 slabs][2 planes][32] (141 inputs, zero padded to 160), one shift per ray from the bound max(1, |coordinates|)
__global__ void __launch_bounds__(256) k_ray_encode_planes(const float* __restrict__ ori, const float* __restrict__ dir, const float* __restrict__ rgb,
                                                           int64_t R, char* __restrict__ xp, int* __restrict__ xs) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (ray, group of 8 inputs): 20 groups per ray
  if (i >= R * 20) return;
  const int64_t ray = i / 20;
  const int g8 = (int)(i - ray * 20);
  const float p[3] = {ori[3 * ray], ori[3 * ray + 1], ori[3 * ray + 2]};
  const float d[3] = {dir[3 * ray], dir[3 * ray + 1], dir[3 * ray + 2]};
  const float c[3] = {rgb[3 * ray], rgb[3 * ray + 1], rgb[3 * ray + 2]};
  float m = 1.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) m = fmaxf(m, fmaxf(fabsf(p[a]), fmaxf(fabsf(d[a]), fabsf(c[a]))));
  const int sh = p_shift(m);
  const float sc = pow2i(sh);
  f16x8_t h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int col = g8 * 8 + e;
    const float x = (col < SIXDGS_RAY_IN_PAD ? ray_input_element(p, d, c, col) : 0.f) * sc;
    const _Float16 hh = (_Float16)x;
    h[e] = hh;
    l[e] = (_Float16)(x - (float)hh);
  }
  char* dst = xp + (ray * 5 + (g8 >> 2)) * kSlabB + (g8 & 3) * 16;
  *reinterpret_cast<f16x8_t*>(dst) = h;
  *reinterpret_cast<f16x8_t*>(dst + 64) = l;
  if (g8 == 0) { xs[2 * ray] = sh; xs[2 * ray + 1] = sh; }
}

// weights fp32 [n][ld] (columns c0 .. c0 + kcols of every row; zero beyond) -> planes [n][ks_total][128 B] at slab offset s_off, scaled by
// the row's power of two f3_scale(wmax[row])
__global__ void __launch_bounds__(256) k_weight_planes(const float* __restrict__ src, int n, int64_t ld, int c0, int kcols, int kslabs, const float* __restrict__ wmax,
                                                       char* __restrict__ dst, int ks_total, int s_off) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;        // (row, group of 8 columns)
  if (i >= n * kslabs * 4) return;
  const int row = i / (kslabs * 4), g8 = i - row * (kslabs * 4);
  const float sc = f3_scale(wmax[row]);
  f16x8_t h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int col = g8 * 8 + e;
    const float x = (col < kcols ? src[(int64_t)row * ld + c0 + col] : 0.f) * sc;
    const _Float16 hh = (_Float16)x;
    h[e] = hh;
    l[e] = (_Float16)(x - (float)hh);
  }
  char* d = dst + ((int64_t)row * ks_total + s_off + (g8 >> 2)) * kSlabB + (g8 & 3) * 16;
  *reinterpret_cast<f16x8_t*>(d) = h;
  *reinterpret_cast<f16x8_t*>(d + 64) = l;
}

int launch_dense(const DenseArgs& A, hipStream_t s) {
  const int64_t m_tiles = sdg_cdiv(A.m, 128);
  if (m_tiles <= 0) return 0;
  if (m_tiles > 0x7fffffffLL) return SIXDGS_E_BADARG;
  hipLaunchKernelGGL(k_dense_planes, dim3((unsigned)m_tiles), dim3(256), 0, s, A, (unsigned)(A.n / 128), (unsigned)m_tiles);
  SDG_LAUNCH_OK();
  return 0;
}

}  // namespace

namespace sdg {

size_t dense_weight_plane_bytes() { return (size_t)(512 * 5 + 512 * 16 + 512 * 21 + 384 * 16 + 384 * 12) * kSlabB; }

int dense_pack_weight_planes(const sixdgs_scorer_weights* w, char* planes, hipStream_t s) {
  struct L { const float* src; int n; int64_t ld; int c0, kcols, kslabs; const float* wmax; int ks_total, s_off; size_t off; };
  const size_t o1 = 0, o2 = o1 + (size_t)512 * 5 * kSlabB, o3 = o2 + (size_t)512 * 16 * kSlabB, o4 = o3 + (size_t)512 * 21 * kSlabB,
               ok = o4 + (size_t)384 * 16 * kSlabB;
  const L layers[] = {
      {w->w1, 512, SIXDGS_RAY_IN_PAD, 0, SIXDGS_RAY_IN_PAD, 5, w->m1, 5, 0, o1},
      {w->w2, 512, SIXDGS_HID, 0, SIXDGS_HID, 16, w->m2, 16, 0, o2},
      {w->w3, 512, SIXDGS_HID + SIXDGS_RAY_IN_PAD, 0, SIXDGS_HID, 16, w->m3, 21, 0, o3},                     // [h2 | x]: the h2 columns ...
      {w->w3, 512, SIXDGS_HID + SIXDGS_RAY_IN_PAD, SIXDGS_HID, SIXDGS_RAY_IN_PAD, 5, w->m3, 21, 16, o3},     // ... then the x columns, padded to 5 slabs
      {w->w4, 384, SIXDGS_HID, 0, SIXDGS_HID, 16, w->m4, 16, 0, o4},
      {w->wk, 384, SIXDGS_D, 0, SIXDGS_D, 12, w->mk, 12, 0, ok},
  };
  for (const L& l : layers)
    hipLaunchKernelGGL(k_weight_planes, dim3((unsigned)sdg_cdiv((int64_t)l.n * l.kslabs * 4, 256)), dim3(256), 0, s, l.src, l.n, l.ld, l.c0, l.kcols, l.kslabs,
                       l.wmax, planes + l.off, l.ks_total, l.s_off);
  SDG_LAUNCH_OK();
  return 0;
}

size_t dense_chain_bytes_per_ray() { return 5 * kSlabB + 2 * 16 * kSlabB + (2 + 4 + 4) * sizeof(int); }

// ori/dir/rgb of m rays -> fp32 keys kdst [m][384] (row stride 384).  ws: dense_chain_bytes_per_ray() * m bytes, 256-B aligned.
int dense_chain(const float* ori, const float* dir, const float* rgb, int64_t m, const sixdgs_scorer_weights* w, const char* wplanes, float* kdst, char* ws,
                hipStream_t s) {
  char* xp = ws;
  char* hp1 = xp + (size_t)m * 5 * kSlabB;
  char* hp2 = hp1 + (size_t)m * 16 * kSlabB;
  int* xs = reinterpret_cast<int*>(hp2 + (size_t)m * 16 * kSlabB);
  int* sa = xs + 2 * m;
  int* sb = sa + 4 * m;
  const size_t o1 = 0, o2 = o1 + (size_t)512 * 5 * kSlabB, o3 = o2 + (size_t)512 * 16 * kSlabB, o4 = o3 + (size_t)512 * 21 * kSlabB,
               ok = o4 + (size_t)384 * 16 * kSlabB;
  hipLaunchKernelGGL(k_ray_encode_planes, dim3((unsigned)sdg_cdiv(m * 20, 256)), dim3(256), 0, s, ori, dir, rgb, m, xp, xs);
  int st;
  DenseArgs l1 = {wplanes + o1, w->m1, w->b1, xp, xs, nullptr, nullptr, 5, 0, 2, 0, m, 512, hp1, sa, nullptr, 0, 1};
  if ((st = launch_dense(l1, s))) return st;
  DenseArgs l2 = {wplanes + o2, w->m2, w->b2, hp1, sa, nullptr, nullptr, 16, 0, 4, 0, m, 512, hp2, sb, nullptr, 0, 1};
  if ((st = launch_dense(l2, s))) return st;
  DenseArgs l3 = {wplanes + o3, w->m3, w->b3, hp2, sb, xp, xs, 16, 5, 4, 2, m, 512, hp1, sa, nullptr, 0, 1};
  if ((st = launch_dense(l3, s))) return st;
  // layer 4 has 384 outputs: 3 blocks; its planes reuse hp2 with 12 slabs per ray
  DenseArgs l4 = {wplanes + o4, w->m4, w->b4, hp1, sa, nullptr, nullptr, 16, 0, 4, 0, m, 384, hp2, sb, nullptr, 0, 0};
  if ((st = launch_dense(l4, s))) return st;
  DenseArgs l5 = {wplanes + ok, w->mk, w->bk, hp2, sb, nullptr, nullptr, 12, 0, 3, 0, m, 384, nullptr, nullptr, kdst, SIXDGS_D, 0};
  return launch_dense(l5, s);
}

}  // namespace sdg
