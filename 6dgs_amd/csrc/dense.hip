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
// Tiles, staging, barrier placement, persistence and the epilogue: see k_dense_planes below.  The last layer (k_proj) leaves either fp32
// rows or -- on the key-cache path -- the scorer's per-128-RAY-tile key planes and reciprocal tile scales themselves.
#include "gemm_kernel.h"
#include "device_math.h"
#include "dense.h"
#include "dense_layout.h"
#include <cstdlib>

using namespace sdg;

namespace {

// Layout of the chain's ACTIVATION planes in HBM: two forms, chosen per process (SIXDGS_DENSE_CM, see dense_chunk_major()).
//   ray-major   [ray][slab][plane h 64 B | plane l 64 B] (rounds 2-3): an operand load is a set of 128-byte pieces 640 B .. 2 KB apart, and
//               a lane of the MFMA result (one ray, 4 consecutive features) owns 8-byte pieces of its ray's row, so the epilogue goes through
//               an LDS staging tile (two barriers per (block, 128-ray half)) to write 512-byte pieces.
//   chunk-major [granule of 128 rays][slab][plane 2][16-byte chunk 4][ray 128][16 B] (round 3, last change): the same 16 KB per (granule,
//               slab), with the 8 features of a chunk of 128 consecutive rays adjacent.  With the rows of the layer's weights permuted so
//               that a lane's registers hold 8 CONSECUTIVE features (dl::row_perm), a lane owns whole 16-byte chunks and a wave
//               instruction of the epilogue writes 2 x 512 consecutive bytes STRAIGHT from the accumulator registers: no staging tile, no
//               barriers behind the block maxima.  The consumer's loads are 1 KB of consecutive bytes per wave instruction (64 rays of one
//               chunk), written into the same LDS image as before.
// (Round 3 also measured a tile-major form [granule][slab][ray][128 B] -- same time as ray-major to 0.3 %, removed again)
// (constants and index arithmetic of both layouts: dense_layout.h, shared with the CPU test-suite)
using dl::kGran;
using dl::kGranSlab;
using dl::kChunkRun;
using dl::kSlabB;
using dl::kPRow;
constexpr int kStRow = 528;           // staging row of the epilogue: 512 B of a ray + 16 B (with 512 the 32 lanes of a write hit one bank: 32-way conflict)
constexpr int kShMax = 40;            // activation shifts are clamped to +-40: the rescale between blocks stays far from overflow

__device__ __forceinline__ int p_shift(float m) {
  const int sh = f3_shift(m);
  return sh > kShMax ? kShMax : (sh < -kShMax ? -kShMax : sh);
}
__device__ __forceinline__ float pow2i(int e) { return __uint_as_float((unsigned)(127 + e) << 23); }

// Activation-plane loads and stores of the chain are NON-TEMPORAL (round 6): activations stream through the L2 once per pass (the sibling pass reads the same
// lines at the same time) while the layer's weight planes are re-read from it for every tile.  Alternating processes on one box, 8 M rays (profiles/
// r06_chain_nt_ab.log): 365.3 -> 369.9 TFLOP/s with both hints (loads alone 368.9, stores alone 365.1 -- round 2's "non-temporal output stores: no change"
// stands on its own); L2 <-> fabric bytes unchanged (FETCH_SIZE +2 %, WRITE_SIZE the same): what improves is the weight planes' stay in L2, not the traffic.
// Same keys bit for bit (cache hints).  -DSDG_ACT_NT=0 -DSDG_OUT_NT=0: the accesses of rounds 2-5.
#ifndef SDG_ACT_NT
#define SDG_ACT_NT 1
#endif
typedef unsigned dense_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_act(const char* p) {
#if SDG_ACT_NT
  const dense_u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const dense_u32x4_t*>(p));
  return uint4{v.x, v.y, v.z, v.w};
#else
  return *reinterpret_cast<const uint4*>(p);
#endif
}

#ifndef SDG_OUT_NT
#define SDG_OUT_NT 1
#endif
typedef float dense_f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_act(char* p, float a, float b, float c, float d) {
#if SDG_OUT_NT
  __builtin_nontemporal_store(dense_f32x4_t{a, b, c, d}, reinterpret_cast<dense_f32x4_t*>(p));
#else
  *reinterpret_cast<float4*>(p) = float4{a, b, c, d};
#endif
}

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
  unsigned* out_norm_max; // or null (with out_tile_inv): *out_norm_max = max(*out_norm_max, bit pattern of max over the rays of |key row|, rounded up)
  int cm_in, cm_out;     // the input / output planes are chunk-major (cm_out: with out_planes and without out_tile_inv; the weight rows are permuted)
  float* out_tile_inv;   // or null.  Non-null (with out_planes, N = 384): the planes are the SCORER's key planes -- one power-of-two scale per
                         // tile of 128 rays (largest magnitude in [2^13, 2^14)), out_tile_inv[tile] = its reciprocal; out_shift is not written
};

constexpr int kMaxGroups = 6;
constexpr int kMaxN = 512;                      // widest layer
#ifdef SDG_DENSE_PROF      // developer build (SIXDGS_EXTRA_FLAGS=-DSDG_DENSE_PROF): cycle stamps of one workgroup's loop sections
__device__ long long g_dense_prof[8 * 16];
#define SDG_T(V) const long long V = clock64();
#define SDG_ACC(K, A_, B_) prof_t[K] += (B_) - (A_);
#else
#define SDG_T(V)
#define SDG_ACC(K, A_, B_)
#endif
constexpr int kWRows = 512;                    // rows of a staged slab: the pass's weight rows + the tile's ray rows
constexpr int kWStage = kWRows * kPRow;        // 73 728 B

// One layer of the plane-to-plane chain.  A workgroup (8 waves: 4 feature waves wm x 2 ray waves wn) is PERSISTENT: it walks over ray
// tiles blockIdx.x, blockIdx.x + gridDim.x, ...; per tile the layer's features come in passes of FP = 128 * NTM features, a wave holding
// NTM x NTN accumulator tiles of 32 x 32: feature tile tm of wave wm is features tm * 128 + wm * 32 .. + 31 of the pass (so accumulator
// row tm IS the 128-feature output block tm, whose four 32-feature slabs sit in the four feature waves), ray tile tn is rays
// wn * 32 * NTN + tn * 32 .. + 31 of the tile's RT = 64 * NTN rays.  Two shapes are used, both with 512 staged rows per slab:
//   <2, 4>: 256 features x 256 rays (N = 512 layers, two passes), 48 MFMAs per 12 fragment reads;
//   <3, 2>: 384 features x 128 rays (N = 384 layers, ONE pass: the input planes are read once), 36 MFMAs per 10 fragment reads.
// Operand slabs go global -> registers -> LDS one slab ahead, one register / one LDS write / one load per group of MFMAs (slots), with a
// load cursor that runs on across passes and TILES: the next tile's first slabs arrive during this tile's last epilogue, so the start-up
// latency of a tile (10 us of 60..90 when every tile was its own workgroup) is paid once per workgroup.
// The slab's ONE barrier sits behind the last slot and behind the second half's fragment reads: a wave passing it has issued all its
// writes of the next slab and finished reading this slab's stage, so behind it the next stage is complete and this one is free for the
// slab after next -- and a wave runs from a slab's last MFMA straight into the next slab's fragment reads.
// Epilogue per pass: bias, ReLU, per-(ray, 128-feature block) power-of-two scale, fp16 split (all waves, in place), then per (block,
// 128-ray half) LDS staging and coalesced 16-byte stores through the stage just consumed.
// INCM / OUTCM: the input / output activation planes are chunk-major (above); OUTCM also means the layer's weight rows are permuted.
template <int NTM, int NTN, bool INCM, bool OUTCM>
__global__ void __launch_bounds__(512, 1) k_dense_planes(DenseArgs A, unsigned n_pass, unsigned total_tiles, unsigned split) {
  constexpr int FP = 128 * NTM;          // features per pass
  constexpr int RT = 64 * NTN;           // rays per tile
  constexpr int kWL = 2 * NTM;           // weight loads per thread and slab (64 rows each); ray loads: 8 - kWL
  constexpr int kBarrierGroup = (3 * NTM + 1) > 8 ? (3 * NTM + 1) : 8;      // first MFMA group behind the barrier
  constexpr int kNS = (kMaxGroups * RT + 511) / 512;                         // input shifts per thread
  static_assert(FP + RT == kWRows, "a staged slab is 512 rows");
  __shared__ __attribute__((aligned(16))) char smem[2 * kWStage];      // two slab stages; the epilogue staging [128 rays][528 B] aliases one
  __shared__ unsigned wmaxb[NTM][RT];                                  // per-(block, ray) maxima (bit patterns of non-negative floats)
  __shared__ __attribute__((aligned(16))) float cwb[2 * kMaxN];        // the layer's reciprocal weight-row scales [n] and biases [n] (loaded once per workgroup)
  __shared__ int shl[kMaxGroups][RT];                                  // input shifts of the tile's rays, per 128-input block
  __shared__ unsigned tile_max;                                        // key-plane mode: the tile's largest magnitude (bit pattern)
  __shared__ float rsq[NTM == 3 ? 4 : 1][NTM == 3 ? RT : 1];           // key-plane mode: per feature wave, the rays' partial sums of squares
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ks = A.ks0 + A.ks1;
  const int nb_all = A.n >> 7;
  f32x16 acc[NTM][NTN];
  // split (two-pass layers, round 3): the two passes of a ray tile run in TWO sibling workgroups at the same time instead of one
  // after the other in one workgroup -- work items w, w ^ 1 of the XCD remap sit on one XCD, start together and do identical work, so
  // the sibling's reads of the tile's input planes hit that XCD's L2 (one workgroup's second pass came a full pass later, after 16 MB
  // of other tiles had gone through the 4 MB L2: the input planes of the N = 512 layers were read from HBM twice).
  const unsigned w = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned pbase = split ? (w & 1u) : 0u;              // first pass of this workgroup ...
  const unsigned npl = split ? 1u : n_pass;                  // ... and how many it runs per tile
  const unsigned tstride = split ? gridDim.x >> 1 : gridDim.x;
  unsigned tile = split ? w >> 1 : w;
  int64_t ray0 = (int64_t)tile * RT;

  int shn[kNS];      // the shifts this thread moves into the table (first tile: now; later tiles: fetched at the tile switch, stored one slab later)
#define SDG_SHIFT_FETCH()                                                                                                  \
  _Pragma("unroll") for (int k_ = 0; k_ < kNS; ++k_) {                                                                     \
    const int i_ = tid + 512 * k_;                                                                                         \
    shn[k_] = 0;                                                                                                           \
    if (i_ < kMaxGroups * RT) {                                                                                            \
      const int g_ = i_ / RT;                                                                                              \
      const int64_t cray_ = min(ray0 + (i_ % RT), A.m - 1);                                                                \
      if (g_ < A.g0) shn[k_] = A.s0[cray_ * A.g0 + g_];                                                                    \
      else if (g_ < A.g0 + A.g1) shn[k_] = A.s1[cray_ * A.g1 + (g_ - A.g0)];                                               \
    }                                                                                                                      \
  }
#define SDG_SHIFT_STORE()                                                                                                  \
  _Pragma("unroll") for (int k_ = 0; k_ < kNS; ++k_) {                                                                     \
    const int i_ = tid + 512 * k_;                                                                                         \
    if (i_ < kMaxGroups * RT) (&shl[0][0])[i_] = shn[k_];                                                                  \
  }
  SDG_SHIFT_FETCH()
  SDG_SHIFT_STORE()

  // loader: 8 lanes x 16 B cover the 128 bytes of one (row, slab): every wave instruction reads 8 full cache lines; 8 instructions x 64
  // rows per slab.  A load's address is a uniform base (scalar unit) + a 32-bit offset, three VALU operations per load; loads are
  // unconditional with clamped rows.
  const unsigned lrow = dl::load_ray(false, (unsigned)tid, 0), lc16 = dl::load_chunk8(false, (unsigned)tid) * 16u;      // weights and ray-major rays
  const unsigned rm_dst = dl::lds_offset(lrow, 0, 0) + lc16;                            // LDS offsets of this thread's pieces within a stage
  const unsigned cm_lane = dl::load_ray(true, (unsigned)tid, 0), cm_chunk = dl::load_chunk8(true, (unsigned)tid) * (unsigned)kChunkRun;
  const unsigned cm_dst = dl::lds_offset(cm_lane, 0, 0) + dl::load_chunk8(true, (unsigned)tid) * 16u;
  unsigned lt = tile, lb = 0;      // load cursor: tile, pass and slab of the next fetch
  int ls = 0;
  const char *abase0, *abase1;
  unsigned lrmax;                  // last valid ray of the cursor's tile
#define SDG_TILEBASE()                                                                   \
  {                                                                                      \
    const int64_t r0_ = (int64_t)lt * RT;                                                \
    abase0 = INCM ? A.a0 + (r0_ >> 7) * A.ks0 * kGranSlab : A.a0 + (r0_ * A.ks0) * kSlabB;                                   \
    abase1 = A.a1 ? (INCM ? A.a1 + (r0_ >> 7) * A.ks1 * kGranSlab : A.a1 + (r0_ * A.ks1) * kSlabB) : abase0;                 \
    lrmax = (unsigned)min((int64_t)(RT - 1), A.m - 1 - r0_);                             \
  }
  SDG_TILEBASE()
  uint4 p0, p1, p2, p3, p4, p5, p6, p7;      // the slab in flight (weights 0 .. kWL-1, rays kWL .. 7); named: an array ended up in scratch
#define SDG_BASES()                                                                                                       \
  const char* wbase = A.wp + (unsigned)ls * kSlabB;                                                                       \
  const bool seg1_ = ls >= A.ks0;                                                                                         \
  const unsigned sstep_ = INCM ? (unsigned)kGranSlab : (unsigned)kSlabB;                                                  \
  const char* abase = seg1_ ? abase1 + (unsigned)(ls - A.ks0) * sstep_ : abase0 + (unsigned)ls * sstep_;                  \
  const unsigned astride = (unsigned)(seg1_ ? A.ks1 : A.ks0) * sstep_, wstride = (unsigned)ks * kSlabB;                   \
  const unsigned wrow0 = (pbase + lb) * (unsigned)FP, wrmax = (unsigned)A.n - 1u;
// ray rows of a slab.  Ray-major: thread (row lrow of 64, chunk tid & 7) reads row * astride (astride = bytes per ray), clamped to the tile's
// last valid ray.  Chunk-major: wave w reads chunk (plane w >> 2, k-chunk w & 3) of the piece's 64 rays, lane = ray: 1 KB of consecutive bytes
// per wave instruction; astride = bytes per granule; a ray beyond the tile's last valid one reads that one instead (same run of bytes).
#define SDG_LOAD(J, P)                                                                                                    \
  if ((J) < kWL) {                                                                                                        \
    P = *reinterpret_cast<const uint4*>(wbase + (min(wrow0 + 64u * (J) + lrow, wrmax) * wstride + lc16));                 \
  } else if (!INCM) {                                                                                                     \
    P = ld_act(abase + (min(64u * ((J) - kWL) + lrow, lrmax) * astride + lc16));                                          \
  } else {                                                                                                                \
    const unsigned rl_ = min(64u * ((J) - kWL) + cm_lane, lrmax);                                                         \
    P = ld_act(abase + dl::cm_src_offset(rl_, cm_chunk, astride));                                                        \
  }
#define SDG_LOAD_ALL() SDG_LOAD(0, p0) SDG_LOAD(1, p1) SDG_LOAD(2, p2) SDG_LOAD(3, p3) SDG_LOAD(4, p4) SDG_LOAD(5, p5) SDG_LOAD(6, p6) SDG_LOAD(7, p7)
// LDS image of a slab (both layouts): row r (weights 0 .. FP-1, rays FP ..) at r * kPRow: [plane h 64 B | plane l 64 B].  DST = the stage's
// base; weights and ray-major rays: thread (lrow, chunk tid & 7); chunk-major rays: thread (ray = lane, chunk = wave)
#define SDG_WRITE(DST, J, P)                                                                                              \
  *reinterpret_cast<uint4*>((DST) + (((J) >= kWL && INCM) ? cm_dst + (FP + 64 * ((J) - kWL)) * kPRow : rm_dst + (J) * 64 * kPRow)) = P;
#define SDG_WRITE_ALL(DST) SDG_WRITE(DST, 0, p0) SDG_WRITE(DST, 1, p1) SDG_WRITE(DST, 2, p2) SDG_WRITE(DST, 3, p3) SDG_WRITE(DST, 4, p4) SDG_WRITE(DST, 5, p5) SDG_WRITE(DST, 6, p6) SDG_WRITE(DST, 7, p7)
// past the last tile the cursor stays on the last tile's last pass (harmless re-reads)
#define SDG_ADVANCE()                                     \
  if (++ls == ks) {                                       \
    ls = 0;                                               \
    if (++lb == npl) {                                    \
      if (lt + tstride < total_tiles) {                   \
        lb = 0;                                           \
        lt += tstride;                                    \
        SDG_TILEBASE()                                    \
      } else {                                            \
        lb = npl - 1;                                     \
      }                                                   \
    }                                                     \
  }
  {
    SDG_BASES()
    SDG_LOAD_ALL()       // slab 0
    SDG_ADVANCE()
  }
  {
    char* const dw0 = smem;
    SDG_WRITE_ALL(dw0)
  }
  {
    SDG_BASES()
    SDG_LOAD_ALL()       // slab 1
    SDG_ADVANCE()
  }
  __syncthreads();
  const int frow = lane & 31, fk = (lane >> 5) * 16;
#pragma unroll
  for (int i = 0; i < NTM; ++i)
#pragma unroll
    for (int j = 0; j < NTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int buf = 0, s = 0;
  unsigned pass = 0;
  bool shifts_pending = false;
  for (int i = tid; i < A.n; i += 512) {
    cwb[i] = f3_inv_scale(A.wmax[i]);
    cwb[kMaxN + i] = A.bias[i];
  }
#ifdef SDG_DENSE_PROF
  long long prof_t[4] = {0, 0, 0, 0};
  const long long prof_start = clock64();
  const long long prof_rt = wall_clock64();
#endif
// one slot: staging register P of the slab loaded during the previous iteration goes to the idle stage and is re-loaded with its piece of
// the slab after next (a load has one full slab, ~2 us, to come back); pinned between two groups of MFMAs
#define SDG_SLOT(J, P)                     \
  {                                        \
    __builtin_amdgcn_sched_barrier(0);     \
    SDG_WRITE(dw, J, P)                    \
    SDG_LOAD(J, P)                         \
    __builtin_amdgcn_sched_barrier(0);     \
  }
  while (true) {                     // flattened over (tile, pass, slab)
    const bool last = s + 1 == ks;
    SDG_T(t0_)
    if (s == 1 && shifts_pending) {      // the new tile's shifts (fetched at the switch): visible behind this slab's barrier, first used at slab 4
      SDG_SHIFT_STORE()
      shifts_pending = false;
    }
    SDG_BASES()
    char* dw = smem + (buf ^ 1) * kWStage;
    if ((s & 3) == 0 && s > 0) {       // a new 128-input block: accumulators to its scale (exact powers of two)
      unsigned t_ = threadIdx.x;      // opaque copy: computed from it, the address is formed HERE (hoisted out of the loop it was spilled, and
      asm volatile("" : "+v"(t_));    // the scratch reload's wait drained the operand loads in flight every fourth slab)
      const int* sp = &shl[0][0] + (s >> 2) * RT + ((t_ >> 6) & 1) * 32 * NTN + (t_ & 31);
#pragma unroll
      for (int tn = 0; tn < NTN; ++tn) {
        const float fac = pow2i(sp[32 * tn] - sp[32 * tn - RT]);
#pragma unroll
        for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[tm][tn][r] *= fac;
      }
    }
    {
      const char* sw = smem + buf * kWStage + (wm * 32 + frow) * kPRow + fk;
      const char* sr = smem + buf * kWStage + (FP + wn * 32 * NTN + frow) * kPRow + fk;
      f16x8_t a[NTM][2], b[NTN][2];
#pragma unroll
      for (int kstep = 0; kstep < 2; ++kstep) {
        // (fragment reads in plane order; reading the planes of the first term l*h first was measured in round 4 -- 305.9 / 304.8 vs 301.9 / 305.7 TFLOP/s,
        //  profiles/r04_chain_lds_dma_ab.log: no difference -- and dropped)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
          for (int t = 0; t < NTM; ++t) a[t][pl] = *reinterpret_cast<const f16x8_t*>(sw + t * 128 * kPRow + pl * 64 + kstep * 32);
#pragma unroll
          for (int t = 0; t < NTN; ++t) b[t][pl] = *reinterpret_cast<const f16x8_t*>(sr + t * 32 * kPRow + pl * 64 + kstep * 32);
        }
        // (weight plane, ray plane): l*h, h*l, h*h -- smallest magnitude first
#pragma unroll
        for (int qq = 0; qq < 3; ++qq) {
          const int pa = qq == 0 ? 1 : 0, pb = qq == 1 ? 1 : 0;
#pragma unroll
          for (int tm = 0; tm < NTM; ++tm) {
            const int g = (kstep * 3 + qq) * NTM + tm;      // MFMA group (compile-time after unrolling)
            if (g == kBarrierGroup) __syncthreads();        // the slab's one barrier
#pragma unroll
            for (int tn = 0; tn < NTN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tm][pa], b[tn][pb], acc[tm][tn], 0, 0, 0);
            switch (g) {
              case 0: SDG_SLOT(0, p0) break;
              case 1: SDG_SLOT(1, p1) break;
              case 2: SDG_SLOT(2, p2) break;
              case 3: SDG_SLOT(3, p3) break;
              case 4: SDG_SLOT(4, p4) break;
              case 5: SDG_SLOT(5, p5) break;
              case 6: SDG_SLOT(6, p6) break;
              case 7: SDG_SLOT(7, p7) break;
              default: break;
            }
          }
        }
      }
    }
#ifdef SDG_FUSE_L1_PROBE
    // Timing probe (developer build, profiles/r06_chain_l1_fused_probe.md): the MATRIX WORK a loader that recomputes h1 = ReLU(W1 x) would add to a layer-2 pass -- per
    // slab of 32 h1 features 32 x 256 rays x 160 inputs x 3 terms = 240 MFMAs per workgroup, 30 per wave -- issued here on a zero weight fragment (the sums, hence the keys,
    // are unchanged), for the 16-slab layer only.  Everything else such a loader needs (x streamed 16 x from L2, 64 more accumulators, the block scale of h1) is NOT in it.
    if (NTM == 2 && NTN == 4 && ks == 16) {
      f16x8_t z;
#pragma unroll
      for (int e_ = 0; e_ < 8; ++e_) z[e_] = (_Float16)0.f;
      asm volatile("" : "+v"(z));
      const f16x8_t bz = *reinterpret_cast<const f16x8_t*>(smem + buf * kWStage + (FP + wn * 32 * NTN + frow) * kPRow + fk);
#pragma unroll
      for (int i_ = 0; i_ < 30; ++i_) acc[i_ & 1][(i_ >> 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(z, bz, acc[i_ & 1][(i_ >> 1) & 3], 0, 0, 0);
    }
#endif
    SDG_ADVANCE()
    SDG_T(t1_)
    SDG_ACC(0, t0_, t1_)
    if (last) {     // the pass is complete: its epilogue borrows the stage just consumed (the other one holds the next pass's first slab)
      // Opaque copies of the thread coordinates for the whole epilogue: formed HERE.  As loop invariants of the (tile, pass, slab) loop
      // the epilogue's dozens of staging / store addresses were hoisted to the kernel's start, spilled, and partly reloaded inside the
      // slab loop (tile-major build: 112 spilled registers; with the copies: the epilogue recomputes them, a few dozen VALU per pass).
      unsigned te = threadIdx.x;
      asm volatile("" : "+v"(te));
      const int lane_e = (int)(te & 63u), wave_e = __builtin_amdgcn_readfirstlane((int)(te >> 6));
      const int wm_e = wave_e >> 1, wn_e = wave_e & 1;
      const int rayl_e = wn_e * 32 * NTN + (lane_e & 31);
      char* const stg = smem + buf * kWStage;        // staging [128 rays][528 B]: one (128-feature block, 128-ray half) at a time
      for (int i = te; i < NTM * RT; i += 512) (&wmaxb[0][0])[i] = 0u;
      const bool tile_mode = NTM == 3 && NTN == 2 && A.out_tile_inv != nullptr;       // the one-pass 384 x 128 shape only (compiled out of the other: its register budget is spent)
      if (te == 0) tile_max = 0u;
      __syncthreads();
      const int f0 = (int)(pbase + pass) * FP;
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
      const bool planes_out = A.out_f32 == nullptr;
      {
        // lane_e: rays rayl_e + 32 tn, features (of the pass) tm*128 + wm_e*32 + 8*(r>>2) + 4*(lane_e>>5) + (r&3) -- with permuted weight rows
        // (OUTCM) tm*128 + wm_e*32 + dl::row_perm(the same) = .. + 16*(r>>3) + 8*(lane_e>>5) + 4*((r>>2)&1) + (r&3); acc becomes the layer output in
        // place.  All factors are powers of two (exact): value = fma(acc * 2^-shift_in, 1 / weight-row scale, bias), one rounding.  Packed
        // fp32 multiplies / fmas, the ReLU as a maximum with 0 or -inf, the row maximum as max3: 2.5 VALU operations per value.
        const int glast = (ks - 1) >> 2;
        const float relu_floor = A.relu ? 0.f : -__builtin_inff();
        float ib[NTN];
#pragma unroll
        for (int tn = 0; tn < NTN; ++tn) ib[tn] = pow2i(-shl[glast][rayl_e + 32 * tn]);
#pragma unroll
        for (int tm = 0; tm < NTM; ++tm) {
          float rmax[NTN];
#pragma unroll
          for (int tn = 0; tn < NTN; ++tn) rmax[tn] = 0.f;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int fl4 = f0 + tm * 128 + wm_e * 32 + dl::acc_feature(OUTCM, lane_e, 4 * rg);      // registers 4 rg .. 4 rg + 3: 4 consecutive features
            const float4 iw4 = *reinterpret_cast<const float4*>(cwb + fl4), b4 = *reinterpret_cast<const float4*>(cwb + kMaxN + fl4);
            const f32x2_t iw[2] = {{iw4.x, iw4.y}, {iw4.z, iw4.w}};
            const f32x2_t bb[2] = {{b4.x, b4.y}, {b4.z, b4.w}};
#pragma unroll
            for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
              for (int jp = 0; jp < 2; ++jp) {
                f32x2_t v = {acc[tm][tn][4 * rg + 2 * jp], acc[tm][tn][4 * rg + 2 * jp + 1]};
                v = v * f32x2_t{ib[tn], ib[tn]};
                v = __builtin_elementwise_fma(v, iw[jp], bb[jp]);
                const float x0 = fmaxf(v.x, relu_floor), x1 = fmaxf(v.y, relu_floor);
                acc[tm][tn][4 * rg + 2 * jp] = x0;
                acc[tm][tn][4 * rg + 2 * jp + 1] = x1;
                rmax[tn] = fmaxf(fmaxf(rmax[tn], fabsf(x0)), fabsf(x1));
              }
          }
          if (tile_mode) {
            float r2 = rmax[0];
#pragma unroll
            for (int tn = 1; tn < NTN; ++tn) r2 = fmaxf(r2, rmax[tn]);
            r2 = sdg_wave_max(r2);
            if (lane_e == 0) atomicMax(&tile_max, __float_as_uint(r2));
          } else if (planes_out) {
            // per-(block, ray) maximum: the partner lane_e l ^ 32 holds the wave's other features of the ray, the other three feature waves
            // the block's other slabs (LDS maximum on the bit patterns: the values are non-negative)
#pragma unroll
            for (int tn = 0; tn < NTN; ++tn) {
              const float r2 = fmaxf(rmax[tn], __shfl_xor(rmax[tn], 32, 64));
              if (lane_e < 32) atomicMax(&wmaxb[tm][rayl_e + 32 * tn], __float_as_uint(r2));
            }
          }
        }
      }
      if (NTM == 3 && tile_mode && A.out_norm_max != nullptr) {
        // |key row| of the tile's rays (the select path derives its slack from the largest one, sixdgs.h): this lane's 48 features of
        // each of its rays, + the partner lane's, one slot per feature wave -- summed in a fixed order behind the barrier (deterministic)
#pragma unroll
        for (int tn = 0; tn < NTN; ++tn) {
          f32x2_t q2 = {0.f, 0.f};
#pragma unroll
          for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const f32x2_t v = {acc[tm][tn][r], acc[tm][tn][r + 1]};
              q2 = __builtin_elementwise_fma(v, v, q2);
            }
          float rs = q2.x + q2.y;
          rs += __shfl_xor(rs, 32, 64);
          if (lane_e < 32) rsq[NTM == 3 ? wm_e : 0][NTM == 3 ? rayl_e + 32 * tn : 0] = rs;
        }
      }
      __syncthreads();
      if (NTM == 3 && tile_mode && A.out_norm_max != nullptr && te < (unsigned)RT) {
        const int ri = NTM == 3 ? (int)te : 0;
        float n2 = (rsq[0][ri] + rsq[NTM == 3 ? 1 : 0][ri]) + (rsq[NTM == 3 ? 2 : 0][ri] + rsq[NTM == 3 ? 3 : 0][ri]);
        n2 = ray0 + (int64_t)te < A.m ? n2 : 0.f;
        n2 = sdg_wave_max(n2);
        if (lane_e == 0 && n2 > 0.f) atomicMax(A.out_norm_max, __float_as_uint(sqrtf(n2) * 1.000244140625f));
      }
      if (planes_out) {
        // every wave splits its values now (in place: 4 values -> 2 registers of h, 2 of l): the unit loop below only moves bytes
#pragma unroll
        for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
          for (int tn = 0; tn < NTN; ++tn) {
            const int ray = rayl_e + 32 * tn;
            int sh;
            if (tile_mode) {      // the scale rule of k_split_tiles_f16 (score.hip), on the same fp32 values: identical planes
              const float m = __uint_as_float(tile_max);
              sh = 0;
              if (m > 0.f && m < INFINITY) {
                int e;
                frexpf(m, &e);
                sh = 14 - e;
                sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
              }
              if (te == 0 && tm == 0 && tn == 0) A.out_tile_inv[ray0 / RT] = ldexpf(1.f, -sh);
            } else {
              sh = p_shift(__uint_as_float(wmaxb[tm][ray]));
              if (wm_e == 0 && lane_e < 32 && ray0 + ray < A.m) A.out_shift[(ray0 + ray) * nb_all + (f0 >> 7) + tm] = sh;
            }
            const float sc = tile_mode ? ldexpf(1.f, sh) : pow2i(sh);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              f16x4 h, l;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float x = acc[tm][tn][4 * rg + j] * sc;
                const _Float16 hh = (_Float16)x;
                h[j] = hh;
                l[j] = (_Float16)(x - (float)hh);
              }
              const f32x2_t hb = __builtin_bit_cast(f32x2_t, h), lb2 = __builtin_bit_cast(f32x2_t, l);
              acc[tm][tn][4 * rg] = hb.x;
              acc[tm][tn][4 * rg + 1] = hb.y;
              acc[tm][tn][4 * rg + 2] = lb2.x;
              acc[tm][tn][4 * rg + 3] = lb2.y;
            }
          }
      }
      if constexpr (OUTCM) {
        // chunk-major planes, straight from the registers: register groups 2 p and 2 p + 1 of a lane hold the 8 features of chunk 2 p + h of the
        // wave's slab (permuted weight rows), as 16 B of plane h and 16 B of plane l; lanes 0..31 / 32..63 of a store cover 32 consecutive rays
        // of chunk 2 p / 2 p + 1: two runs of 512 consecutive bytes per wave instruction.  No staging tile, no further barrier.
        const int nslab_out = A.n >> 5;
#pragma unroll
        for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
          for (int tn = 0; tn < NTN; ++tn) {
            const int64_t gr = ray0 + rayl_e + 32 * tn;
            if (gr < A.m) {
              char* const ob = A.out_planes + dl::cm_offset(gr, nslab_out, (f0 >> 5) + tm * 4 + wm_e, 0, lane_e >> 5);      // chunk 2 pp + h: + 2 pp runs
#pragma unroll
              for (int pp = 0; pp < 2; ++pp) {
                st_act(ob + (2 * pp) * kChunkRun, acc[tm][tn][8 * pp], acc[tm][tn][8 * pp + 1], acc[tm][tn][8 * pp + 4], acc[tm][tn][8 * pp + 5]);
                st_act(ob + (4 + 2 * pp) * kChunkRun, acc[tm][tn][8 * pp + 2], acc[tm][tn][8 * pp + 3], acc[tm][tn][8 * pp + 6], acc[tm][tn][8 * pp + 7]);
              }
            }
          }
      } else {
#pragma unroll
      for (int tm = 0; tm < NTM; ++tm) {                    // unit = (128-feature block tm, 128-ray half): its waves fill the staging tile
        for (int hsel = 0; hsel < RT / 128; ++hsel) {
          if (((wn_e * 32 * NTN) >> 7) == hsel) {
            const int ru = (rayl_e & 127);                    // ray within the unit: ru + 32 * tn
            if (!planes_out) {
              float* st = reinterpret_cast<float*>(stg);
#pragma unroll
              for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                  *reinterpret_cast<float4*>(st + (ru + 32 * tn) * (kStRow / 4) + wm_e * 32 + 8 * rg + 4 * (lane_e >> 5)) =
                      float4{acc[tm][tn][4 * rg], acc[tm][tn][4 * rg + 1], acc[tm][tn][4 * rg + 2], acc[tm][tn][4 * rg + 3]};
            } else {
#pragma unroll
              for (int tn = 0; tn < NTN; ++tn)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                  // feature wm_e*32 + 8 rg + 4 (lane_e>>5) + j of the block: slab wm_e, position 8 rg + 4 (lane_e>>5)
                  char* d = stg + (ru + 32 * tn) * kStRow + wm_e * kSlabB + (8 * rg + 4 * (lane_e >> 5)) * 2;
                  *reinterpret_cast<f32x2_t*>(d) = f32x2_t{acc[tm][tn][4 * rg], acc[tm][tn][4 * rg + 1]};
                  *reinterpret_cast<f32x2_t*>(d + 64) = f32x2_t{acc[tm][tn][4 * rg + 2], acc[tm][tn][4 * rg + 3]};
                }
            }
          }
          __syncthreads();
          const int fb = f0 + tm * 128;
          const int64_t rbase = ray0 + hsel * 128;
          if (!planes_out) {
#pragma unroll
            for (int i = (int)te; i < 128 * 32; i += 512) {
              const int ray = i >> 5, c = i & 31;
              if (rbase + ray < A.m)
                *reinterpret_cast<float4*>(A.out_f32 + (rbase + ray) * A.ldo + fb + c * 4) = reinterpret_cast<const float4*>(stg + ray * kStRow)[c];
            }
          } else {      // ray-major rows: the round-2 activation layout, and the scorer's key planes [ray][12 slabs][128 B]
            const int nslab_out = A.n >> 5;
#pragma unroll
            for (int i = (int)te; i < 128 * 32; i += 512) {
              const int ray = i >> 5, c = i & 31;
              if (rbase + ray < A.m)
                *reinterpret_cast<uint4*>(A.out_planes + ((rbase + ray) * nslab_out + (fb >> 5)) * kSlabB + c * 16) = reinterpret_cast<const uint4*>(stg + ray * kStRow)[c];
            }
          }
          __syncthreads();
        }
      }
      }
      SDG_T(te_)
      SDG_ACC(1, t1_, te_)
      if (++pass == npl) {      // next tile of this workgroup
        pass = 0;
        tile += tstride;
        if (tile >= total_tiles) break;
        ray0 = (int64_t)tile * RT;
        SDG_SHIFT_FETCH()
        shifts_pending = true;
      }
#pragma unroll
      for (int i_ = 0; i_ < NTM; ++i_)
#pragma unroll
        for (int j_ = 0; j_ < NTN; ++j_)
#pragma unroll
          for (int r_ = 0; r_ < 16; ++r_) acc[i_][j_][r_] = 0.f;
      s = -1;
    }
    buf ^= 1;
    ++s;
  }
#undef SDG_SLOT
#undef SDG_LOAD
#undef SDG_LOAD_ALL
#undef SDG_WRITE
#undef SDG_WRITE_ALL
#undef SDG_BASES
#undef SDG_ADVANCE
#undef SDG_TILEBASE
#undef SDG_SHIFT_FETCH
#undef SDG_SHIFT_STORE
#ifdef SDG_DENSE_PROF
  if (blockIdx.x == gridDim.x / 2 && lane == 0) {
    g_dense_prof[wave * 16 + 0] = prof_t[0];
    g_dense_prof[wave * 16 + 1] = prof_t[1];
    g_dense_prof[wave * 16 + 8] = clock64() - prof_start;
    g_dense_prof[wave * 16 + 9] = wall_clock64() - prof_rt;      // 100 MHz
  }
#endif
}

// a12 as planes: x[R][5 slabs][2 planes][32] (141 inputs, zero padded to 160), one shift per ray from the bound max(1, |coordinates|).
// One workgroup = 64 rays.  The 66 (component, frequency) arguments of a ray give sine AND cosine with one argument reduction (sincosf:
// the thread-per-8-inputs form this replaces called sinf / cosf 132 times per ray, each with its own reduction -- coordinates times 2^7
// reach 1e5 rad), staged as fp32 in LDS [ray][161] (odd stride: consecutive rays on consecutive banks), then split 8 inputs at a time.
constexpr int kEncRays = 64;
__global__ void __launch_bounds__(256) k_ray_encode_planes(const float* __restrict__ ori, const float* __restrict__ dir, const float* __restrict__ rgb,
                                                           int64_t R, char* __restrict__ xp, int* __restrict__ xs, int chunk_major) {
  __shared__ float X[kEncRays][161];
  __shared__ float src[kEncRays][9];      // p, d, c
  const int tid = threadIdx.x;
  const int64_t ray0 = (int64_t)blockIdx.x * kEncRays;
  const int n = (int)min((int64_t)kEncRays, R - ray0);
  for (int i = tid; i < kEncRays * 9; i += 256) {
    const int r = i / 9, a = i - r * 9;
    const int64_t gr = ray0 + min(r, n - 1);
    const float* q = a < 3 ? ori : (a < 6 ? dir : rgb);
    const float v = q[3 * gr + (a % 3)];
    src[r][a] = v;
    X[r][a] = v;                                                   // columns 0..8: the raw coordinates (ray_preprocessor.py:36-44)
  }
  for (int i = tid; i < kEncRays * 19; i += 256) X[i / 19][141 + i % 19] = 0.f;      // padding columns 141..159
  __syncthreads();
  for (int i = tid; i < kEncRays * 66; i += 256) {
    const int r = i % kEncRays, k = i / kEncRays;                   // k: p 0..23 (8 frequencies x 3), d 24..47, c 48..65 (6 x 3)
    const int grp = k < 24 ? 0 : (k < 48 ? 1 : 2);
    const int F = grp == 2 ? 6 : 8, o = k - grp * 24;
    const int comp = o / F, f = o - comp * F;
    const float v = src[r][grp * 3 + comp] * (float)(1 << f);
    float sn, cs;
    sincosf(v, &sn, &cs);
    const int col = 9 + grp * 48 + comp * F + f;                    // sines, then (3 F further) the cosines of the group
    X[r][col] = sn;
    X[r][col + 3 * F] = cs;
  }
  __syncthreads();
  // 8 inputs -> one 16-byte chunk of plane h and one of plane l.  Ray-major: consecutive threads take the 20 chunks of a ray; chunk-major:
  // consecutive threads take the same chunk of consecutive rays (their 16-byte pieces are adjacent in that layout)
  for (int i = tid; i < kEncRays * 20; i += 256) {
    const int r = chunk_major ? i % kEncRays : i / 20, g8 = chunk_major ? i / kEncRays : i % 20;
    if (r >= n) continue;
    float m = 1.f;
#pragma unroll
    for (int a = 0; a < 9; ++a) m = fmaxf(m, fabsf(src[r][a]));
    const int sh = p_shift(m);
    const float sc = pow2i(sh);
    f16x8_t h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = X[r][g8 * 8 + e] * sc;
      const _Float16 hh = (_Float16)x;
      h[e] = hh;
      l[e] = (_Float16)(x - (float)hh);
    }
    const int64_t gr = ray0 + r;
    char* const dst = xp + (chunk_major ? dl::cm_offset(gr, 5, g8 >> 2, 0, g8 & 3) : dl::rm_offset(gr, 5, g8 >> 2, 0, g8 & 3));
    *reinterpret_cast<f16x8_t*>(dst) = h;
    *reinterpret_cast<f16x8_t*>(dst + (chunk_major ? 4 * kChunkRun : 64)) = l;
    if (g8 == 0) { xs[2 * (ray0 + r)] = sh; xs[2 * (ray0 + r) + 1] = sh; }
  }
}

// weights fp32 [n][ld] (columns c0 .. c0 + kcols of every row; zero beyond) -> planes [n][ks_total][128 B] at slab offset s_off, scaled by
// the row's power of two f3_scale(wmax[row]); perm: plane row (row & ~31) + m holds feature (row & ~31) + dl::row_perm(m)
__global__ void __launch_bounds__(256) k_weight_planes(const float* __restrict__ src, int n, int64_t ld, int c0, int kcols, int kslabs, const float* __restrict__ wmax,
                                                       char* __restrict__ dst, int ks_total, int s_off, int perm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;        // (plane row, group of 8 columns)
  if (i >= n * kslabs * 4) return;
  const int prow = i / (kslabs * 4), g8 = i - prow * (kslabs * 4);
  const int row = perm ? (prow & ~31) | dl::row_perm(prow & 31) : prow;
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
  char* d = dst + ((int64_t)prow * ks_total + s_off + (g8 >> 2)) * kSlabB + (g8 & 3) * 16;
  *reinterpret_cast<f16x8_t*>(d) = h;
  *reinterpret_cast<f16x8_t*>(d + 64) = l;
}

// The chain's activation layout of this process: chunk-major unless SIXDGS_DENSE_CM=0 (ray-major, rounds 2-3).  One per process, because the
// weight planes are packed once (sixdgs_pack_scorer_weights) with or without the row permutation that goes with it.  Both layouts give the
// same keys bit for bit (same MFMA order per output, same scales; tools/cm_check.py, test_chain_layouts_give_identical_keys); chunk-major is
// 3-4 % faster (290 -> 302 TFLOP/s fp32-equivalent, 8 M rays, alternating runs on one box; profiles/r03_chain_chunk_major.log).
constexpr bool kChunkMajorDefault = true;
bool dense_chunk_major() {
  static const bool on = [] { const char* e = getenv("SIXDGS_DENSE_CM"); return e ? atoi(e) != 0 : kChunkMajorDefault; }();
  return on;
}

int dense_grid(int64_t tiles) {
  static int cus = 0;      // one persistent workgroup per compute unit (the kernel's 156 KB of LDS allow one)
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) return 0;
    cus = prop.multiProcessorCount;
  }
  return (int)(tiles < cus ? tiles : cus);
}

int launch_dense(const DenseArgs& A, hipStream_t s) {
  if (A.m <= 0) return 0;
  if (A.ks0 + A.ks1 < 2 || (A.g0 + A.g1) > kMaxGroups || A.n > kMaxN) return SIXDGS_E_BADARG;
  if (A.out_tile_inv && (A.n != 384 || !A.out_planes)) return SIXDGS_E_BADARG;      // a key tile = one workgroup tile of the 384 x 128 shape
  // N = 384 layers: one pass of 384 features over 128-ray tiles; N = 512: two passes of 256 features over 256-ray tiles
  const bool wide = A.n % 384 == 0;
  if (!wide && A.n % 256 != 0) return SIXDGS_E_BADARG;
  const int64_t tiles = sdg_cdiv(A.m, wide ? 128 : 256);
  if (tiles > 0x7fffffffLL) return SIXDGS_E_BADARG;
  const unsigned n_pass = (unsigned)(wide ? A.n / 384 : A.n / 256);
  const unsigned split = (!wide && n_pass == 2) ? 1u : 0u;      // (round 2 ran both passes in one workgroup: the N = 512 layers read their input twice)
  int grid = dense_grid(split ? 2 * tiles : tiles);
  if (grid <= 0) return (int)hipErrorInvalidDevice;
  if (split) grid &= ~1;                       // sibling pairs
  if (A.cm_out && (!A.out_planes || A.out_tile_inv || !A.cm_in)) return SIXDGS_E_BADARG;
  const dim3 g((unsigned)grid), b(512);
  const unsigned nt = (unsigned)tiles;
  if (wide) {
    if (A.cm_out) hipLaunchKernelGGL((k_dense_planes<3, 2, true, true>), g, b, 0, s, A, n_pass, nt, 0u);
    else if (A.cm_in) hipLaunchKernelGGL((k_dense_planes<3, 2, true, false>), g, b, 0, s, A, n_pass, nt, 0u);
    else hipLaunchKernelGGL((k_dense_planes<3, 2, false, false>), g, b, 0, s, A, n_pass, nt, 0u);
  } else {
    if (A.cm_in != A.cm_out) return SIXDGS_E_BADARG;      // (the N = 512 layers are inner layers)
    if (A.cm_out) hipLaunchKernelGGL((k_dense_planes<2, 4, true, true>), g, b, 0, s, A, n_pass, nt, split);
    else hipLaunchKernelGGL((k_dense_planes<2, 4, false, false>), g, b, 0, s, A, n_pass, nt, split);
  }
  SDG_LAUNCH_OK();
  return 0;
}

}  // namespace

#ifdef SDG_DENSE_PROF
extern "C" int sixdgs_debug_dense_prof(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dense_prof), sizeof(long long) * 128); }
#endif

namespace sdg {

size_t dense_weight_plane_bytes() { return (size_t)(512 * 5 + 512 * 16 + 512 * 21 + 384 * 16 + 384 * 12 + 384 * 16) * kSlabB; }

// k_proj folded into layer 4 (round 5): layer 4 has no non-linearity behind it (ray_preprocessor.py:27-31,46) and k_proj follows
// (our_multihead_attention.py:74), so K = (Wk W4) h3 + (Wk b4 + bk) -- four launches per chunk instead of five, 294 912 of the 2 025 472
// FLOP per ray and the 3.1 KB/ray round trip of layer 4's planes gone.  The composite weights come from sixdgs_pack_weights (fp64 composition,
// one rounding).  On unless SIXDGS_FOLD_KPROJ=0 (read at every call: the five-layer form stays reachable for A/B runs and the parity test).
bool dense_fold_kproj() {
  const char* e = getenv("SIXDGS_FOLD_KPROJ");
  return e ? atoi(e) != 0 : true;
}

int dense_pack_weight_planes(const sixdgs_scorer_weights* w, char* planes, hipStream_t s) {
  struct L { const float* src; int n; int64_t ld; int c0, kcols, kslabs; const float* wmax; int ks_total, s_off; size_t off; int perm; };
  const int pm = dense_chunk_major() ? 1 : 0;      // the layers that WRITE chunk-major planes (all but k_proj) have their rows permuted
  const size_t o1 = 0, o2 = o1 + (size_t)512 * 5 * kSlabB, o3 = o2 + (size_t)512 * 16 * kSlabB, o4 = o3 + (size_t)512 * 21 * kSlabB,
               ok = o4 + (size_t)384 * 16 * kSlabB, o4k = ok + (size_t)384 * 12 * kSlabB;
  const L layers[] = {
      {w->w1, 512, SIXDGS_RAY_IN_PAD, 0, SIXDGS_RAY_IN_PAD, 5, w->m1, 5, 0, o1, pm},
      {w->w2, 512, SIXDGS_HID, 0, SIXDGS_HID, 16, w->m2, 16, 0, o2, pm},
      {w->w3, 512, SIXDGS_HID + SIXDGS_RAY_IN_PAD, 0, SIXDGS_HID, 16, w->m3, 21, 0, o3, pm},                     // [h2 | x]: the h2 columns ...
      {w->w3, 512, SIXDGS_HID + SIXDGS_RAY_IN_PAD, SIXDGS_HID, SIXDGS_RAY_IN_PAD, 5, w->m3, 21, 16, o3, pm},     // ... then the x columns, padded to 5 slabs
      {w->w4, 384, SIXDGS_HID, 0, SIXDGS_HID, 16, w->m4, 16, 0, o4, pm},
      {w->wk, 384, SIXDGS_D, 0, SIXDGS_D, 12, w->mk, 12, 0, ok, 0},
      {w->w4k, 384, SIXDGS_HID, 0, SIXDGS_HID, 16, w->m4k, 16, 0, o4k, 0},                                       // Wk W4: writes keys, rows not permuted
  };
  for (const L& l : layers)
    hipLaunchKernelGGL(k_weight_planes, dim3((unsigned)sdg_cdiv((int64_t)l.n * l.kslabs * 4, 256)), dim3(256), 0, s, l.src, l.n, l.ld, l.c0, l.kcols, l.kslabs,
                       l.wmax, planes + l.off, l.ks_total, l.s_off, l.perm);
  SDG_LAUNCH_OK();
  return 0;
}

size_t dense_chain_bytes_per_ray() { return 5 * kSlabB + 2 * 16 * kSlabB + (2 + 4 + 4) * sizeof(int); }

// ori/dir/rgb of m rays -> fp32 keys kdst [m][384] (row stride 384).  ws: dense_chain_bytes_per_ray() * (m rounded up to 128) bytes, 256-B aligned.
int dense_chain(const float* ori, const float* dir, const float* rgb, int64_t m, const sixdgs_scorer_weights* w, const char* wplanes, float* kdst, char* kplanes,
                float* kinv, float* knorm_max, char* ws, hipStream_t s) {
  const size_t mp = (size_t)sdg_cdiv(m, kGran) * kGran;      // the plane buffers hold whole granules of 128 rays (either layout: 128 B per ray and slab)
  const size_t ng = mp / kGran;
  const int cm = dense_chunk_major() ? 1 : 0;
  char* xp = ws;
  char* hp1 = xp + ng * (size_t)(5 * kGranSlab);
  char* hp2 = hp1 + ng * (size_t)(16 * kGranSlab);
  int* xs = reinterpret_cast<int*>(hp2 + ng * (size_t)(16 * kGranSlab));
  int* sa = xs + 2 * mp;
  int* sb = sa + 4 * mp;
  const size_t o1 = 0, o2 = o1 + (size_t)512 * 5 * kSlabB, o3 = o2 + (size_t)512 * 16 * kSlabB, o4 = o3 + (size_t)512 * 21 * kSlabB,
               ok = o4 + (size_t)384 * 16 * kSlabB, o4k = ok + (size_t)384 * 12 * kSlabB;
  hipLaunchKernelGGL(k_ray_encode_planes, dim3((unsigned)sdg_cdiv(m, kEncRays)), dim3(256), 0, s, ori, dir, rgb, m, xp, xs, cm);
  int st;
#ifdef SDG_DENSE_PROF      // developer build: stop behind layer n, so that the cycle stamps (one set, overwritten by every launch) are that layer's
  const char* stop_e = getenv("SIXDGS_DENSE_PROF_LAYER");
  const int stop_at = stop_e ? atoi(stop_e) : 0;
#define SDG_PROF_STOP(N) if (stop_at == (N)) return 0;
#else
#define SDG_PROF_STOP(N)
#endif
  DenseArgs l1 = {wplanes + o1, w->m1, w->b1, xp, xs, nullptr, nullptr, 5, 0, 2, 0, m, 512, hp1, sa, nullptr, 0, 1, nullptr, cm, cm, nullptr};
  if ((st = launch_dense(l1, s))) return st;
  SDG_PROF_STOP(1)
  DenseArgs l2 = {wplanes + o2, w->m2, w->b2, hp1, sa, nullptr, nullptr, 16, 0, 4, 0, m, 512, hp2, sb, nullptr, 0, 1, nullptr, cm, cm, nullptr};
  if ((st = launch_dense(l2, s))) return st;
  SDG_PROF_STOP(2)
  DenseArgs l3 = {wplanes + o3, w->m3, w->b3, hp2, sb, xp, xs, 16, 5, 4, 2, m, 512, hp1, sa, nullptr, 0, 1, nullptr, cm, cm, nullptr};
  if ((st = launch_dense(l3, s))) return st;
  SDG_PROF_STOP(3)
  if (dense_fold_kproj()) {
    // layer 4 and k_proj as ONE layer on h3: fp32 keys, or (kplanes) the scorer's key planes, tile scales and key-norm maximum
    DenseArgs l4k = {wplanes + o4k, w->m4k, w->b4k, hp1, sa, nullptr, nullptr, 16, 0, 4, 0, m, 384, kplanes, nullptr, kplanes ? nullptr : kdst, SIXDGS_D, 0,
                     kplanes ? reinterpret_cast<unsigned*>(knorm_max) : nullptr, cm, 0, kplanes ? kinv : nullptr};
    return launch_dense(l4k, s);
  }
  // layer 4 has 384 outputs: 3 blocks; its planes reuse hp2 with 12 slabs per ray
  DenseArgs l4 = {wplanes + o4, w->m4, w->b4, hp1, sa, nullptr, nullptr, 16, 0, 4, 0, m, 384, hp2, sb, nullptr, 0, 0, nullptr, cm, cm, nullptr};
  if ((st = launch_dense(l4, s))) return st;
  // k_proj: fp32 keys, or (kplanes) straight the scorer's key planes -- a 128-ray tile of the 384 x 128 shape IS a key tile, so the split
  // kernel and the fp32 keys' trip through HBM drop out
  DenseArgs l5 = {wplanes + ok, w->mk, w->bk, hp2, sb, nullptr, nullptr, 12, 0, 3, 0, m, 384, kplanes, nullptr, kplanes ? nullptr : kdst, SIXDGS_D, 0,
                  kplanes ? reinterpret_cast<unsigned*>(knorm_max) : nullptr, cm, 0, kplanes ? kinv : nullptr};
  return launch_dense(l5, s);
}

}  // namespace sdg
