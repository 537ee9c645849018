// score.hip -- the per-image part of the scorer:
//   logits[t][r] = q[t] . key[r] / sqrt(384)           (matrix-core tile kernels, stored once + online row statistics)
//   score[r]     = sum_t exp(logits[t][r] - max_t) / sumexp_t      (HBM-streaming second pass)
//   top-k        = radix select over the score bits + ordered gather + small sort
// replaces MultiHeadAttention.forward (our_multihead_attention.py:70-79,4-12), the column sum of
// IdentificationModule.run_attention (identification_module.py:80-82) and torch.topk (:131).
//
// The softmax runs over the RAY axis (rows = image tokens): its row statistics must be complete before any column sum can
// be formed, so the [T, R] logits are written once to a caller-provided workspace and streamed back, instead of recomputing
// the contraction in a second pass.  Two logits kernels, selected by the operands and SIXDGS_MMA_* (include/sixdgs.h):
//   k_logits_f16x   (default; key PLANES)  scaled fp16 planes x 3 MFMA terms, 256 x 256 tiles, 24-bit (or fp32) blocked logits; the select sweep
//   k_logits<MMA>   (fp32 KEYS: sixdgs_score_topk, the plain entry point on a [R,384] key matrix; scenes below 4 M rays keep one)  bf16 x 6 with the
//                   split done on the fly, or -- SIXDGS_MMA_F32 -- the fp32 MFMA chain, which the parity tests use as the independent arithmetic
// followed by k_merge_stats, k_score_reduce(_blocked, _blocked24) and the top-k kernels.  DESIGN.md 3a / 3b tell how the
// default kernel got its shape and what bounds it.
#include <stdlib.h>

#include <cstdlib>
#include <cstdio>
#include "gemm_kernel.h"
#include "sweep_plan.h"

using namespace sdg;

namespace {

constexpr int kT = SIXDGS_MAX_TOKENS;
constexpr float kSqrtD = 19.595917942265423f;  // math.sqrt(384) rounded to fp32, as torch does for `/ python_float`

// ------------------------------------------------------------------------------------------------
// pass 1: logits tile + online row statistics.  One workgroup = 128 tokens x a contiguous group of
// 128-ray tiles of one image.
// ------------------------------------------------------------------------------------------------
struct LogitsArgs {
  const float* q;        // [B,256,384]
  const int* n_tok;      // [B]
  const float* key;      // [R,384]
  float* logits;         // [Bg,256,ldl]
  float* partial;        // [Bg,G,256,2]
  int64_t r, ldl;
  int tiles_per_group, n_tiles, n_groups;
  int b0;                // first image of this group in q / n_tok
};

template <int MMA>
__global__ void __launch_bounds__(256, 2) k_logits(LogitsArgs A) {
  __shared__ __attribute__((aligned(16))) char smem[TileSmem<MMA>::kBytes];
  __shared__ float part[2][128][2];
  const int bl = blockIdx.y;            // image within the group
  const int b = A.b0 + bl;
  const unsigned w = xcd_remap(blockIdx.x, gridDim.x);
  const int grp = (int)(w >> 1), m_tile = (int)(w & 1u);
  const int M = A.n_tok[b];
  const int row0 = m_tile * 128;
  float* pout = A.partial + (((int64_t)bl * A.n_groups + grp) * kT + row0) * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  float m_run = -INFINITY, s_run = 0.f;   // owned by thread tid < 128 for token row0 + tid
  if (row0 < M) {
    GemmOperands g = {A.q + (int64_t)b * kT * SIXDGS_D, nullptr, A.key, SIXDGS_D, 0, SIXDGS_D, M, A.r, SIXDGS_D, SIXDGS_D};
    float* lg = A.logits + (int64_t)bl * kT * A.ldl;
    const int t_begin = grp * A.tiles_per_group;
    const int t_end = min(t_begin + A.tiles_per_group, A.n_tiles);
    for (int tile = t_begin; tile < t_end; ++tile) {
      const int64_t col0 = (int64_t)tile * kBN;
      f32x16 acc[2][2];
      gemm_tile<MMA>(g, row0, col0, smem, acc);
      const int64_t c0 = col0 + acc_col(wn, 0, lane), c1 = col0 + acc_col(wn, 1, lane);
      const bool v0 = c0 < A.r, v1 = c1 < A.r;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + acc_row(wm, tm, r, lane);
          const float l0 = acc[tm][0][r] / kSqrtD, l1 = acc[tm][1][r] / kSqrtD;
          if (row < M) {
            if (v0) lg[(int64_t)row * A.ldl + c0] = l0;
            if (v1) lg[(int64_t)row * A.ldl + c1] = l1;
          }
          float mx = fmaxf(v0 ? l0 : -INFINITY, v1 ? l1 : -INFINITY);
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
          float sm = 0.f;
          if (mx > -INFINITY) sm = (v0 ? expf(l0 - mx) : 0.f) + (v1 ? expf(l1 - mx) : 0.f);
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
          if ((lane & 31) == 0) {
            const int lr = acc_row(wm, tm, r, lane);
            part[wn][lr][0] = mx;
            part[wn][lr][1] = sm;
          }
        }
      __syncthreads();
      if (tid < 128) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float mt = part[h][tid][0], st = part[h][tid][1];
          if (mt > -INFINITY) {
            const float mn = fmaxf(m_run, mt);
            s_run = s_run * expf(m_run - mn) + st * expf(mt - mn);
            m_run = mn;
          }
        }
      }
      // the next tile's mainloop starts with a barrier before `part` can be overwritten again
    }
  }
  if (tid < 128) {
    pout[2 * tid] = m_run;
    pout[2 * tid + 1] = s_run;
  }
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
constexpr int kRowBytes = 12 * 3 * 64;   // 2304 B: the q-plane slot of the two-pass workspace keeps the size round 1's bf16 x 6 plane scorer gave it (that kernel and its
                                         // three-plane key format were removed in round 6: nothing launched them outside their own tests; workspace sizes stay what callers know)
constexpr float kInvSqrtD = 0.05103103630798288f;   // (float)(1/sqrt(384))

// ------------------------------------------------------------------------------------------------
// pass 1, fp16x3 variant (SIXDGS_MMA_F16X3): TWO scaled fp16 planes per operand, THREE cross terms.
// x * 2^s = h + l with h = fp16(x 2^s), l = fp16(x 2^s - h): fp16 carries 11 significant bits, so h + l reproduces
// x to 2^-23 and l*h + h*l + h*h to ~2^-22 per product (every fp16 x fp16 product is exact in fp32).  The power-of-two
// scale (one per 128-row tile of either operand, chosen so that the tile's max|x| 2^s is in [2^13, 2^14)) keeps h and l
// in fp16's normal range for everything within 2^-17 of the tile's largest value and is undone exactly by the epilogue
// constant.  Measured (tools/probe_f16x3.py, K = 384): 1.0e-7 * sum|a||b|, below both the fp32 MFMA chain (1.6e-7) and
// bf16x6.  Half the MFMA instructions of bf16x6 and 2/3 of its operand bytes (1536 B per row, the size of fp32).
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int kRowF = 12 * 2 * 64;        // 1536 B of planes per operand row
constexpr int kSlabF = 128;               // bytes per row and slab

struct LogitsF16Args {
  const char* qp;        // [B][256][1536 B]
  const int* n_tok;
  const char* kp;        // [R][1536 B]
  const float* qinv;     // [B][2] reciprocal power-of-two scale of each 128-token half, indexed like qp
  const float* kinv;     // [ceil(R/128)] reciprocal power-of-two scale of each 128-ray tile
  float* logits;
  float* partial;
  int64_t r, ldl;
  int tiles_per_group, n_tiles, n_groups, b0, nb;
  // OUT == kOutUB only (top-k without materialised logits, see "select path" below)
  const float* ctok;     // [B][256] per-token exponent offset: -ref_t log2e - log2(f Z~_t), -inf for tokens >= n_tok
  float* ub;             // [nb][4 token quarters wm][ub_stride] partial upper-bound sums of every ray
  int64_t ub_stride;
  // sibling lock-step (round 3; 0 / null = off): the grid is n_sets x nb PERSISTENT workgroups, sibling set s = the nb workgroups that
  // score the nb images of a launch against the same ray-tile groups s, s + n_sets, ...; after every tile they meet at sib_sync[s]
  int n_sets;
  unsigned* sib_sync;    // [n_sets] zeroed before the launch
  int sib_period;        // tiles between two meetings of a sibling set (>= 1)
  unsigned sib_extra;    // arrivals a set waits for beyond its members: 0.  (1 = SIXDGS_SIBLING_SYNC=3, the test of the bounded wait: a sibling never shows up)
  int q_quarter_scales;  // 0: qinv [B][2], one scale per 128-token half (k_split_tiles_f16); 1: qinv [B][4], one per 64-token quarter (k_split_q_slots:
                         // the select sweep's packed slots, where the quarters of a tile belong to different images)
};
// what the kernel leaves behind for each tile
constexpr int kOutF32 = 0;     // logits as fp32 (blocked layout) + running (max, sumexp)
constexpr int kOutL24 = 1;     // logits as 24-bit fixed point + running (max, sumexp)
constexpr int kOutStats = 2;   // running (max, sumexp) only (the sample pre-pass of the select path)
constexpr int kAblOneTerm = 8192;   // ABL bit (a product mode, not an ablation): only the h*h term of the three -- logits to ~2^-11 of |q||k| / sqrt(384).  For the sample
                                    // pre-pass alone: its (max, sumexp) set the exponent offsets ref_t and the scale Z~_t of the sweep, whose g_t = Z_t / (f Z~_t) is EXACT
                                    // relative to whatever Z~_t it was given -- the pre-pass decides how wide the bounds are, never the answer (include/sixdgs.h)
constexpr int kOutUB = 3;      // per ray: sum_t exp(l_tr - ref_t) / (f Z~_t) (4 token-quarter partials); per token: the exact sum over rays

// A wave-uniform global load through the scalar cache.  As a plain load hipcc emits global_load_dword (the kernel also
// stores to global memory, so it cannot prove the location unclobbered) followed by s_waitcnt vmcnt(0) -- which drains
// every LDS-DMA prefetch in flight.  The constant address space forces s_load_dword (lgkmcnt).
__device__ __forceinline__ float load_uniform(const float* p) {
  return *reinterpret_cast<const __attribute__((address_space(4))) float*>((uintptr_t)p);
}

__device__ __forceinline__ f16x8 lds_read_frag_h_off(unsigned addr, const int imm) {
  f16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(imm));
  return v;
}

#ifdef SIXDGS_ABLATION
__device__ unsigned long long g_dbg_cycles[8][8];   // [wave][phase] summed over the blocks with grp == 0 (timing variant)
#endif
// ------------------------------------------------------------------------------------------------
// fp16x3 logits kernel, 256 tokens x 256 rays per tile, one workgroup of 8 waves per CU.
// How it got this shape (all measured on MI355X, tools/ablate_logits.py; DESIGN.md section 3b has the numbers):
//  * 128 x 128 tiles / 4 waves (the bf16x6 structure): every non-MFMA cycle is exposed with one wave per SIMD;
//  * the reference orientation Q K^T leaves a lane with one ray column and 16 token rows: 64 four-byte stores per lane
//    and two 31-step cross-lane reductions per tile for the row statistics.  Computing K Q^T instead gives a lane ONE
//    token and runs of 4 consecutive rays: 16-byte stores that are 1 KiB contiguous per wave instruction, and running
//    (max, sumexp) per lane with no cross-lane traffic at all until the end of the run;
//  * cycle stamps then showed the CU's vector-memory issue port as the limiter (~40 cycles per 1-KiB DMA piece or
//    store; waves queue on it while the matrix pipe idles), so the tile grew to 256 x 256: 64 DMA pieces per 32-k slab
//    for 65 536 outputs (256 x 128: 48 for 32 768), 12 ds_read_b128 per 24 MFMA (8 per 12);
//  * with that the matrix pipe is busy ~85 % of the cycles of the slab loop and the chip is at its power limit
//    (GRBM_GUI_ACTIVE / duration = 1.36 GHz effective clock under this kernel).
//   wave (wm 0..3, wn 0..1): tokens [64 wm, +64) x rays [128 wn, +128) of the tile: acc[2][4] = 128 registers
//   LDS (160 KiB exactly): q slabs  [plane 2][stage 2][256 rows][64 B] at 0       (from L2: 2-stage ring)
//                          key slabs [stage 3][plane 2][256 rows][64 B] at 64 KiB (from HBM: 3-stage ring)
//   per slab and wave: 4 steps of 12 MFMA (k-step x ray half); fragments for the next step are read during the current
//   one (A: 2 x 2 planes per k-step; B: 2 ray blocks x 2 planes per step); the 8 DMA pieces (q of slab+2, key of slab+3)
//   are issued in step 3, after the slab barrier.
// ------------------------------------------------------------------------------------------------
constexpr int kBNX = 256;                // rays per tile
// How the select sweep's grid is laid out when a launch scores several images (SIXDGS_SIBLING_SYNC overrides).  The nb workgroups that
// score the nb images against the same ray-tile group ("siblings", one XCD) share the key tiles through that XCD's 4 MB L2 only while
// they stay within about a third of a tile of each other (32 workgroups x 384 KB per tile go through the L2 at once).
//   0  one-shot grid, one workgroup per (ray-tile group, image) -- round 2.  A sibling starts whenever a CU frees up;
//   2  persistent sibling sets: n_sets = CUs / nb sets of nb workgroups, all resident from the start, set s walks the groups
//      s, s + n_sets, ... -- the siblings start together ONCE and then drift at their CUs' pace;
//   1  (default) persistent sets that meet after every tile (an atomic arrival + a spin on a scalar load per sibling set).
// HBM bytes per launch / algorithmic bytes (FETCH_SIZE x 2 + WRITE_SIZE, headline workload), two boxes, `tools/sib_ab.sh` / `sib_ab2.sh`:
//   mode 0: 1.40x / 1.19x      mode 2: 1.10x / 1.32x (1.15x and 1.47x in two profile runs)      mode 1: 1.05x / 1.14x      (meeting every 4th / 8th tile: 1.24x / 1.25x)
// -- only the per-tile meeting is better on every box.  Launch time by HIP events: 57.77 / 58.16 / 58.69 ms on the first box, 59.9 / 60.3 / 60.1 ms on the
// second (under the PMC pass): the meeting costs 0.2-1.6 % (a tile takes as long as its slowest sibling), less than the boxes differ (56.5-61 ms).  The
// re-reads themselves never cost time -- the kernel is matrix-pipe / power bound at 1.0-1.3 TB/s -- the default takes the bytes down where that is nearly free.
// No deadlock: within ONE launch a set's resident members spin only until the set's other members are dispatched, which needs a free CU, which the sets
// whose members are all resident provide by finishing (workgroups are dispatched in blockIdx order: the resident prefix consists of whole sets but one
// per XCD).  Two or more sweeps in flight on one device (streams, processes) could each hold half-resident sets on all of an XCD's CUs; for that the spin
// is bounded (kSibSpinLimit) and a set whose member gives up is released for the rest of the launch -- the launch then runs like mode 2.
constexpr int kSiblingSyncDefault = 1;
constexpr int kSibSpinLimit = 1 << 14;          // polls of ~0.3-1 us each
constexpr unsigned kSibReleased = 0x40000000u;  // OR-ed into a set's arrival counter: every later target compares as reached
constexpr int kSweepMaxImages = 8;      // 256-token SLOTS per sweep launch (the last launch of a batch: up to 12); see sixdgs_select_sweep.  SIXDGS_SWEEP_MAX_IMAGES=n overrides (the name is
                                        // round 4's, when a slot held one image); n <= 0 means "as many as the slot table holds": launches of 21, a last one of up to 31 (sweep_plan.h) -- NOT one launch for any batch
constexpr bool kPrepassOneTermDefault = true;    // profiles/r06_prepass_one_term.md
constexpr int kPrepassReserveCusOneSlot = 160;
constexpr int kPrepassReserveCus = 64;   // CUs the sample pre-pass leaves to other streams (-1: one-shot grid on all of them); see sixdgs_select_sample_stats
constexpr int kSibPeriod = 1;            // tiles between two meetings of a sibling set (mode 1).  Round 3 measured 4 and 8: 1.24x / 1.25x the algorithmic bytes against 1.05-1.14x
// 24-bit logits: per 128-ray tile [token group 8][ray quad 32][token 32][4 x 24 bit = 12 B] = 96 KiB, followed by the
// references [token group 8][ray half of the quad 2][token 32] fp32 = 2 KiB (the maximum of the 64 logits a lane produced
// for that token in this tile).  0.766 x the bytes of fp32 logits, written once and read once per image.
constexpr int kTileBytes24 = 98304 + 2048;
constexpr int kQStageX = 256 * 64;       // one plane of one stage
constexpr int kKBaseX = 4 * kQStageX;    // 64 KiB
constexpr int kLdsX = kKBaseX + 6 * kQStageX;   // 160 KiB

// OUT: what leaves the kernel (kOut*): fp32 or 24-bit fixed-point logits (see kTileBytes24), statistics only, or the
// upper-bound column sums of the select path
template <int ABL, int OUT, bool PERS = false>
__global__ void __launch_bounds__(512, 1) k_logits_f16x(LogitsF16Args A) {
  constexpr bool L24 = OUT == kOutL24;
  constexpr bool kOneTerm = (ABL & kAblOneTerm) != 0;
  __shared__ __attribute__((aligned(1024))) char lds[kLdsX];
  const unsigned w = xcd_remap(blockIdx.x, gridDim.x);
  const int bl = (int)(w % (unsigned)A.nb);
  const int set = (int)(w / (unsigned)A.nb);        // the ray-tile group (one-shot grid) or the sibling set (persistent grid)
  const int b = A.b0 + bl;
  const int M = A.n_tok[b];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const float cq = A.q_quarter_scales ? A.qinv[4 * b + wm] : A.qinv[2 * b + (wm >> 1)];
  const bool active = wm * 64 < M;
  float ct[2] = {0.f, 0.f};
  if (OUT == kOutUB) {
    ct[0] = A.ctok[(int64_t)b * kT + wm * 64 + (lane & 31)];
    ct[1] = A.ctok[(int64_t)b * kT + wm * 64 + 32 + (lane & 31)];
  }
  // A.n_tiles counts 256-ray tiles here; the groups take floor(n_tiles / n_groups) tiles, the first n_tiles % n_groups one more
  const int t_base = A.n_tiles / A.n_groups, t_rem = A.n_tiles - t_base * A.n_groups;
  const int n_tiles128 = (int)((A.r + 127) >> 7);
  unsigned sib_target = PERS ? A.sib_extra : 0u;    // arrivals at sib_sync[set] after the tiles walked so far (nb per tile)
  int grp = set;
  do {                                              // PERS: the ray-tile groups set, set + n_sets, ...; otherwise the one group `set`
  float* pout = A.partial + ((int64_t)bl * A.n_groups + grp) * kT * 2;
  float m_run[2] = {-INFINITY, -INFINITY}, s_run[2] = {0.f, 0.f};
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 zs[2][2] = {{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}};      // [token block][r & 2]: pairs (r & 3) = (0, 1), (2, 3)
  const int t_begin = grp * t_base + min(grp, t_rem);
  const int t_end = t_begin + t_base + (grp < t_rem ? 1 : 0);
  if (PERS && !(M > 0) && A.sib_sync && t_begin < t_end) {      // an image without tokens walks no tiles: its arrivals all at once
    const unsigned meets = (unsigned)((t_end - t_begin + A.sib_period - 1) / A.sib_period);
    if (tid == 0) atomicAdd(A.sib_sync + set, meets);
    sib_target += (unsigned)A.nb * meets;
  }
  if (M > 0 && t_begin < t_end) {
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)lds;
    // ---- DMA pieces: 1 KiB = 8 rows x 128 B = the two 64-byte planes of a (row, slab), i.e. 8 FULL 128-byte lines per
    //      instruction (16 half lines with a plane-major mapping -- the vector-memory port is the limiter, and its cost
    //      follows the number of lines touched).  LDS image [256 rows][128 B] per operand and stage, lane-linear; the eight
    //      16-byte chunks of a row (plane * 4 + k-chunk) are XOR-swizzled with (row >> 1) & 7 on the source side, which
    //      makes the 16 lanes of every ds_read_b128 group hit 16 distinct bank groups.
    //      piece i = 0..3 of either operand: rows 8 (4 wave + i) ..
    int lane_p = lane;      // (PERS: an opaque copy, so that the piece offsets below are formed per GROUP instead of being hoisted out of the
    if (PERS) asm volatile("" : "+v"(lane_p));      // group loop and spilled across the whole kernel)
    const int prow = lane_p >> 3, pos8 = lane_p & 7;
    unsigned offQ[4], offK[4];
    int rowK[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave * 4 + i) * 8 + prow;
      const unsigned inrow = (unsigned)((pos8 ^ ((row >> 1) & 7)) << 4);   // (plane * 64 + chunk * 16) of the source row-slab
      offQ[i] = (unsigned)min(row, M - 1) * kRowF + inrow;
      offK[i] = inrow;
      rowK[i] = row;
    }
    const char* qbase = A.qp + (int64_t)b * kT * kRowF;
    // Cache policy of the key stream's loads (gfx950 aux bits: 1 = sc0, 2 = nt, 16 = sc1): `nt`.  A key tile is read once per launch by the workgroups of ONE
    // sibling set, which meet after every tile -- all its readers pass within a tile's time -- while the q planes of the launch's slots are re-read for every
    // tile by every set of the XCD.  With 8 slots those q planes are 3.1 MB of the XCD's 4 MB L2, and key lines without the hint displaced them: L2 <-> fabric
    // traffic 1.44 x the algorithmic bytes at 8 tiles per launch; with it 1.02 x, the sweep the same or 0.2 % faster (profiles/r06_key_stream_nt.md; rounds
    // 2-4 measured the hint at 2-4 slots per launch, where the q planes fit anyway: no difference, as again now).  -DSDG_KEY_AUX=0: the loads of rounds 1-5.
#ifndef SDG_KEY_AUX
#define SDG_KEY_AUX 2
#endif
    auto issue_q = [&](const int s, const int qstage, const int i) {
      if (ABL & 1) return;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(qbase + (offQ[i] + (unsigned)(s * kSlabF))),
                                       (lds_ptr_t)(lds + qstage * (2 * kQStageX) + (wave * 4 + i) * 1024), 16, 0, 0);
    };
    auto issue_k = [&](const char* kbase, int lim, const int s, const int kstage, const int i) {
      if (ABL & 8) return;
      const unsigned ob = (unsigned)min(rowK[i], lim) * kRowF + offK[i] + (unsigned)(s * kSlabF);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(kbase + ob),
                                       (lds_ptr_t)(lds + kKBaseX + kstage * (2 * kQStageX) + (wave * 4 + i) * 1024), 16, 0, SDG_KEY_AUX);
    };
    auto tile_lim = [&](int tile) {
      const int64_t left = A.r - (int64_t)tile * kBNX - 1;
      return left < kBNX - 1 ? (int)left : kBNX - 1;
    };
    // fragment addresses per (k-step, plane): row block t adds 32 rows = 4096 B (the swizzle term (row >> 1) & 7 does not
    // depend on t), so row block and stage go into the 16-bit offset field of ds_read; fb2 = fb + 64 KiB (key stage 2)
    unsigned fa[2][2], fb[2][2], fb2[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const int c8 = pl * 4 + 2 * ks + (lane_p >> 5);
        const int ra = wm * 64 + (lane_p & 31), rb = wn * 128 + (lane_p & 31);
        fa[ks][pl] = lds0 + ra * 128 + ((c8 ^ ((ra >> 1) & 7)) << 4);
        fb[ks][pl] = lds0 + kKBaseX + rb * 128 + ((c8 ^ ((rb >> 1) & 7)) << 4);
        fb2[ks][pl] = fb[ks][pl] + 65536u;
      }
    auto read_a = [&](const int t, const int ks, const int pl, const int qstage) {
      return lds_read_frag_h_off(fa[ks][pl], qstage * (2 * kQStageX) + t * 4096);
    };
    auto read_b = [&](const int t, const int ks, const int pl, const int kstage) {
      const int off = kstage * (2 * kQStageX) + t * 4096;
      return off < 65536 ? lds_read_frag_h_off(fb[ks][pl], off & 65535) : lds_read_frag_h_off(fb2[ks][pl], off & 65535);
    };
    // logits of image bl, blocked by 128-ray tiles: [tile128][token group g = t / 32][ray quad][t % 32][r % 4]
    float* lg = A.logits + (int64_t)bl * kT * A.ldl + ((wm * 2) * 4096 + lane_p * 4);
    char* lg24 = reinterpret_cast<char*>(A.logits) + (int64_t)bl * kT * A.ldl * 4;   // same per-image region, 24-bit layout
    const char* kcur = A.kp + (int64_t)t_begin * kBNX * kRowF;
    int lim_cur = tile_lim(t_begin);

    f16x8 a0[2][2], a1[2][2], b0[2][2], b1[2][2];   // [row block][plane]
    auto wait_lds = [&]() {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    // timing variant (ABL & 2048): phase p accumulates the cycles since the previous stamp (0 loop overhead, 1 steps 0-2,
    // 2 vmcnt wait, 3 barrier, 4 step 3, 5 epilogue)
    unsigned long long t_prev = 0, t_acc[6] = {0, 0, 0, 0, 0, 0};
    auto tstamp = [&](const int ph) {
      const unsigned long long now = __builtin_readcyclecounter();
      t_acc[ph] += now - t_prev;
      t_prev = now;
    };
    if (ABL & 2048) t_prev = __builtin_readcyclecounter();
    // ---- prologue: key slabs 0, 1, 2 and q slabs 0, 1 in the steady-state order (q of a batch before its key) ----------
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_k(kcur, lim_cur, 0, 0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_q(0, 0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_k(kcur, lim_cur, 1, 1, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_q(1, 1, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_k(kcur, lim_cur, 2, 2, i);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        if (kOneTerm && pl == 1) continue;      // (a fragment read whose result is never used must not be issued: its register is free for reuse before the data lands)
        a0[t][pl] = read_a(t, 0, pl, 0);
        b0[t][pl] = read_b(t, 0, pl, 0);
      }
    wait_lds();
    if (ABL & 16) {      // (ablation "no fragment reads": the second fragment set must still be DEFINED, or the compiler drops the MFMAs that use it --
#pragma unroll          //  the r1-r4 "MFMA only" figures above 2.5 PFLOP/s came from that)
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          a1[t][pl] = a0[t][pl];
          b1[t][pl] = b0[t][pl];
          asm volatile("" : "+v"(a1[t][pl]), "+v"(b1[t][pl]));
        }
    }

    for (int tile = t_begin; tile < t_end; ++tile) {
      // after the last tile of the run the prefetch simply re-reads the current tile (harmless, keeps the slab loop and
      // its vmcnt bookkeeping free of branches)
      const bool has_next = tile + 1 < t_end;
      const char* knext = has_next ? kcur + (int64_t)kBNX * kRowF : kcur;
      const int lim_next = has_next ? tile_lim(tile + 1) : lim_cur;
      // the 12 slabs are fully unrolled; keep the per-slab source addresses from being hoisted out of the tile loop
      asm volatile("" : "+v"(offQ[0]), "+v"(offQ[1]), "+v"(offQ[2]), "+v"(offQ[3]), "+v"(offK[0]), "+v"(offK[1]), "+v"(offK[2]), "+v"(offK[3]));
      // ---- token-aware (round 4): a wave whose 64-token row block lies at or beyond the image's token count (masked Tanks&Temples / Blender views
      // keep 56-140 of 256 tokens, backbone.py:86-114) has nothing to compute: it skips the tile's 576 MFMAs, fragment reads and accumulators and only
      // does its share of the workgroup's data movement -- per slab the same counted wait, the same barrier and the same 8 DMA pieces in the same
      // order as the waves that compute (vmcnt bookkeeping unchanged), issued right behind the barrier.  Wave-uniform branch; waves 0-3 (token rows
      // 0-127) sit on the four SIMDs, so an image of <= 128 tokens leaves every SIMD with ONE computing wave and half the matrix work.
      if (!active && !(ABL & 4096)) {
        int ks3 = 0;                                 // sl % 3 (a ROLLED loop with run-time stages: unrolled, its 96 piece addresses cost the computing
#pragma unroll 1                                     // waves registers -- 41 spilled VGPRs in the first build)
        for (int sl = 0; sl < 12; ++sl) {
          const int qs = sl & 1;
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          const int sq = sl + 2 < 12 ? sl + 2 : sl + 2 - 12;
#pragma unroll
          for (int i = 0; i < 4; ++i) issue_q(sq, qs, i);
          const bool nxt = sl + 3 >= 12;
          const char* kb = nxt ? knext : kcur;
          const int kl = nxt ? lim_next : lim_cur, sk = nxt ? sl + 3 - 12 : sl + 3;
#pragma unroll
          for (int i = 0; i < 4; ++i) issue_k(kb, kl, sk, ks3, i);
          ks3 = ks3 == 2 ? 0 : ks3 + 1;
        }
        kcur = knext;
        lim_cur = lim_next;
        continue;
      }
      f32x16 acc[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

      // One step: 12 MFMAs (l*h, h*l, h*h for 2 token blocks x 2 ray blocks of ray half `h`) on A fragments xa and B
      // fragments xb (C^T = K Q^T: rows = rays, columns = tokens); `side(slot)` places the reads / DMA of the step.
      auto mfma_step = [&](f16x8 (&xa)[2][2], f16x8 (&xb)[2][2], const int h, auto side) {
        constexpr int PA[3] = {1, 0, 0};
        constexpr int PB[3] = {0, 1, 0};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
          for (int z = 0; z < 4; ++z) {
            const int tm = z >> 1, tn = z & 1;
            if (!(ABL & 4) && !((ABL & kAblOneTerm) && q < 2))
              acc[tm][2 * h + tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xb[tn][PB[q]], xa[tm][PA[q]], acc[tm][2 * h + tn], 0, 0, 0);
            side(q * 4 + z);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };

#pragma unroll
      for (int sl = 0; sl < 12; ++sl) {
        const int qs = sl & 1, ks3 = sl % 3;              // stages of this slab
        const int qn = (sl + 1) & 1, kn = (sl + 1) % 3;    // stages of the next slab
        if (ABL & 2048) tstamp(0);
        // step 0: (k-step 0, ray half 0); read B(k-step 0, half 1)
        mfma_step(a0, b0, 0, [&](const int slot) {
          if (kOneTerm && (slot & 1)) return;
          if (!(ABL & 16) && slot < 4) b1[slot >> 1][slot & 1] = read_b(2 + (slot >> 1), 0, slot & 1, ks3);
        });
        wait_lds();
        // step 1: (k-step 0, half 1); read A(k-step 1), B(k-step 1, half 0)
        mfma_step(a0, b1, 1, [&](const int slot) {
          if (kOneTerm && (slot & 1)) return;
          if (!(ABL & 16) && slot < 4) a1[slot >> 1][slot & 1] = read_a(slot >> 1, 1, slot & 1, qs);
          else if (!(ABL & 16) && slot < 8) b0[(slot - 4) >> 1][slot & 1] = read_b((slot - 4) >> 1, 1, slot & 1, ks3);
        });
        wait_lds();
        // step 2: (k-step 1, half 0); read B(k-step 1, half 1) -- the last reads of this slab
        mfma_step(a1, b0, 0, [&](const int slot) {
          if (kOneTerm && (slot & 1)) return;
          if (!(ABL & 16) && slot < 4) b1[slot >> 1][slot & 1] = read_b(2 + (slot >> 1), 1, slot & 1, ks3);
        });
        wait_lds();
        // q(sl+1) was issued first in the previous batch: only that batch's 4 key pieces are younger.  After the barrier
        // slab sl+1 is visible to every wave and the stages of slab sl are free.
        if (ABL & 2048) tstamp(1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (ABL & 2048) tstamp(2);
        if (!(ABL & 32)) __builtin_amdgcn_s_barrier();
        if (ABL & 2048) tstamp(3);
        // step 3: (k-step 1, half 1); read A / B(half 0) of slab sl+1; issue q(sl+2) -> q stage qs, key(sl+3) -> key stage ks3
        mfma_step(a1, b1, 1, [&](const int slot) {
          if (!(kOneTerm && (slot & 1))) {
            if (!(ABL & 16) && slot < 4) a0[slot >> 1][slot & 1] = read_a(slot >> 1, 0, slot & 1, qn);
            else if (!(ABL & 16) && slot < 8) b0[(slot - 4) >> 1][slot & 1] = read_b((slot - 4) >> 1, 0, slot & 1, kn);
          }
          if (slot < 4) issue_q((sl + 2) % 12, qs, slot);
          else if (slot < 8) {
            if (sl + 3 < 12) issue_k(kcur, lim_cur, sl + 3, ks3, slot - 4);
            else issue_k(knext, lim_next, sl + 3 - 12, ks3, slot - 4);
          }
        });
        wait_lds();
        if (ABL & 2048) tstamp(4);
      }

      // ---- sibling lock-step: the nb workgroups of this sibling set (same XCD: consecutive work items of the remap) meet here after
      // every sib_period-th tile (and after a group's last one), so that they stay within a fraction of a tile of each other and the key slabs one of them pulls from HBM are still in the XCD's
      // L2 when the others ask for them (round 2 measured 1.40x the algorithmic bytes: the siblings started together and drifted).
      // Wave 0 announces (one atomic without return) and spins on a SCALAR load (lgkmcnt: the ring's vmcnt bookkeeping is untouched); the
      // other waves run on into the epilogue and the next tile's first slab, where the slab barrier holds them for wave 0.
      if (PERS && A.sib_sync != nullptr && (((tile - t_begin) % A.sib_period) == A.sib_period - 1 || tile + 1 == t_end)) {
        sib_target += (unsigned)A.nb;
        if (wave == 0) {
          unsigned* const sp = A.sib_sync + set;
          if (lane == 0) __hip_atomic_fetch_add(sp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // The wait is BOUNDED: the meeting is a cache-locality measure, nothing reads what a sibling wrote.  A sibling that does not show up within
          // kSibSpinLimit polls (milliseconds; a tile takes ~30 us) is not resident -- another kernel holds its CU, e.g. a second sweep on another
          // stream or from another process, whose own half-resident sets could be waiting for OUR CUs -- and then the first member to notice releases
          // the set for the rest of the launch (kSibReleased is beyond every target: < 2^24 arrivals per set).
          for (int polls = 0;; ++polls) {
            unsigned seen;
            asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(seen) : "s"(sp) : "memory");
            if ((int)(seen - sib_target) >= 0) break;
            if (polls == kSibSpinLimit) {
              if (lane == 0) __hip_atomic_fetch_or(sp, kSibReleased, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (OR: several members may give up at once)
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
        }
      }
      if (ABL & 2) {
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
        if (sacc == 12345.678f) lg[lane] = sacc;
        kcur = knext;
        lim_cur = lim_next;
        continue;
      }
      // ---- epilogue.  Lane l holds token (l & 31) and, per accumulator register group rg, the four consecutive rays
      // 8 rg + 4 (l >> 5) + {0..3} of its 32-ray block: one 16-byte store per group, 1 KiB contiguous per wave instruction.
      // The constant undoes both power-of-two operand scales (exactly) and applies 1/sqrt 384.
      if (OUT == kOutUB) {
        // ---- select path.  e'[t][r] = exp2(acc * cfl + ct[t]) = exp(logit - ref_t) / (f Z~_t); the per-token sums over the rays
        // accumulate in zs (exact softmax denominators relative to ref_t); the per-ray sums over this wave's 64 tokens are formed
        // by a halving butterfly over the 32 lanes of each half wave (lane bit i <-> ray-register bit 3 - i, DPP for the two
        // in-quad steps, ds_swizzle for the rest) and leave as TWO 4-byte stores per lane and tile: ray
        // 64 b4 + 32 x + 8 rg + 4 h + j of the wave's 128-ray half with 4 rg + j = bitrev4(lane & 15), h = lane >> 5.
        if (active) {
          const int t128 = min(2 * tile + wn, n_tiles128 - 1);
          const float cfl = ((cq * load_uniform(A.kinv + t128)) * kInvSqrtD) * 1.4426950408889634f;
          const bool ragged = lim_cur < kBNX - 1;
#ifdef SDG_UB_OPAQUE
          int ray0 = wn * 128 + 4 * (lane >> 5);
          asm volatile("" : "+v"(ray0));        // formed here, not hoisted (see the two-pass branch below)
#else
          const int ray0 = wn * 128 + 4 * (lane >> 5);
#endif
          const bool lb0 = lane & 1, lb1 = lane & 2, lb2 = lane & 4, lb3 = lane & 8, lb4 = lane & 16;
          float fin[4];
#pragma unroll
          for (int tn = 0; tn < 4; ++tn) {
            // two rays per instruction (round 3): the whole epilogue is VALU time the matrix pipe waits for -- 128 scalar fmas, 256 + 128
            // scalar adds and 128 v_exp_f32 per lane and tile were 27 % of a tile; as v_pk_fma_f32 / v_pk_add_f32 on (r, r + 1) pairs the
            // fmas and adds halve.  Same operations in the same order per value: the results are the same bits.
            float u[16];
            const f32x2 cfl2 = {cfl, cfl}, ct0 = {ct[0], ct[0]}, ct1 = {ct[1], ct[1]};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const f32x2 x0 = __builtin_elementwise_fma(f32x2{acc[0][tn][r], acc[0][tn][r + 1]}, cfl2, ct0);
              const f32x2 x1 = __builtin_elementwise_fma(f32x2{acc[1][tn][r], acc[1][tn][r + 1]}, cfl2, ct1);
              f32x2 e0 = {__builtin_amdgcn_exp2f(x0.x), __builtin_amdgcn_exp2f(x0.y)};
              f32x2 e1 = {__builtin_amdgcn_exp2f(x1.x), __builtin_amdgcn_exp2f(x1.y)};
              if (ragged) {
                const int rr = ray0 + tn * 32 + 8 * (r >> 2) + (r & 3);                 // clamped duplicates of the last ray count as 0
                const bool oka = rr <= lim_cur, okb = rr + 1 <= lim_cur;
                e0 = f32x2{oka ? e0.x : 0.f, okb ? e0.y : 0.f};
                e1 = f32x2{oka ? e1.x : 0.f, okb ? e1.y : 0.f};
              }
              zs[0][(r & 3) >> 1] += e0;
              zs[1][(r & 3) >> 1] += e1;
              const f32x2 uu = e0 + e1;
              u[r] = uu.x;
              u[r + 1] = uu.y;
            }
            float v8[8], v4[4], v2[2];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const float keep = lb0 ? u[i + 8] : u[i], send = lb0 ? u[i] : u[i + 8];
              v8[i] = keep + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float keep = lb1 ? v8[i + 4] : v8[i], send = lb1 ? v8[i] : v8[i + 4];
              v4[i] = keep + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const float keep = lb2 ? v4[i + 2] : v4[i], send = lb2 ? v4[i] : v4[i + 2];
              v2[i] = keep + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, send), 0x101F));   // lane ^ 4
            }
            {
              const float keep = lb3 ? v2[1] : v2[0], send = lb3 ? v2[0] : v2[1];
              fin[tn] = keep + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, send), 0x201F));   // lane ^ 8
            }
          }
          float o[2];
#pragma unroll
          for (int x = 0; x < 2; ++x) {
            const float keep = lb4 ? fin[2 + x] : fin[x], send = lb4 ? fin[x] : fin[2 + x];
            o[x] = keep + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, send), 0x401F));      // lane ^ 16
          }
          const int rrev = ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3);
          const int inhalf = ((lane & 16) << 2) + 8 * (rrev >> 2) + 4 * (lane >> 5) + (rrev & 3);
          float* up = A.ub + ((int64_t)bl * 4 + wm) * A.ub_stride + ((int64_t)tile * kBNX + wn * 128 + inhalf);
          __builtin_nontemporal_store(o[0], up);
          __builtin_nontemporal_store(o[1], up + 32);
        }
      } else if (active) {
        const int t128 = min(2 * tile + wn, n_tiles128 - 1);
        const float cf = (cq * load_uniform(A.kinv + t128)) * kInvSqrtD;
        float* tb = lg + (int64_t)(2 * tile + wn) * (kT * 128);
        const bool ragged = lim_cur < kBNX - 1;           // only the last ray tile of the scene
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          float mx = -INFINITY;
#pragma unroll
          for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              float4 v;
              v.x = acc[tm][tn][4 * rg + 0] * cf;
              v.y = acc[tm][tn][4 * rg + 1] * cf;
              v.z = acc[tm][tn][4 * rg + 2] * cf;
              v.w = acc[tm][tn][4 * rg + 3] * cf;
              acc[tm][tn][4 * rg + 0] = v.x;
              acc[tm][tn][4 * rg + 1] = v.y;
              acc[tm][tn][4 * rg + 2] = v.z;
              acc[tm][tn][4 * rg + 3] = v.w;
              if (OUT == kOutF32 && !(ABL & 64)) {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(tb + tm * 4096 + tn * 1024 + rg * 256));
              }
              mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));   // clamped duplicate rays cannot raise the max
            }
          if (L24 && !(ABL & 64)) {
            // 24-bit fixed point of (this lane's tile maximum - logit), resolution 2^-19, clamped at 32 (e^-32 of the
            // largest term): an absolute error <= 2^-20 per logit, below the fp32 rounding of a logit of magnitude >= 16
            typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
            char* t24 = lg24 + (int64_t)(2 * tile + wn) * kTileBytes24 + ((wm * 2 + tm) * 12288 + lane * 12);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
              for (int rg = 0; rg < 4; ++rg) {
                unsigned d[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                  d[j] = (unsigned)__float2uint_rn(fminf((mx - acc[tm][tn][4 * rg + j]) * 524288.f, 16777215.f));
                const u32x3 w3 = {d[0] | (d[1] << 24), (d[1] >> 8) | (d[2] << 16), (d[2] >> 16) | (d[3] << 8)};
                __builtin_nontemporal_store(w3, reinterpret_cast<u32x3*>(t24 + (tn * 8 + 2 * rg) * 384));
                __builtin_amdgcn_sched_barrier(0);      // keeps the 64 encodes from being hoisted above the stores (spills)
              }
            reinterpret_cast<float*>(lg24 + (int64_t)(2 * tile + wn) * kTileBytes24 + 98304)[((wm * 2 + tm) * 2 + (lane >> 5)) * 32 + (lane & 31)] = mx;
          }
          const float mn = fmaxf(m_run[tm], mx);
          float sum[4] = {0.f, 0.f, 0.f, 0.f};
          if (!ragged) {
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
              for (int r = 0; r < 16; ++r) sum[r & 3] += __expf(acc[tm][tn][r] - mn);
          } else {
            // (only the scene's last tile comes here.  The 64 ray indices are formed from an OPAQUE copy: as loop invariants they were
            // hoisted to the kernel's start and spilled -- 344 B of scratch in round 2 -- for a branch taken once per launch)
            int ray0 = wn * 128 + 4 * (lane >> 5);
            asm volatile("" : "+v"(ray0));
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
              for (int r = 0; r < 16; ++r)
                sum[r & 3] += (ray0 + tn * 32 + 8 * (r >> 2) + (r & 3) <= lim_cur) ? __expf(acc[tm][tn][r] - mn) : 0.f;
          }
          s_run[tm] = s_run[tm] * __expf(m_run[tm] - mn) + ((sum[0] + sum[1]) + (sum[2] + sum[3]));
          m_run[tm] = mn;
        }
      }
      kcur = knext;
      lim_cur = lim_next;
      if (ABL & 2048) tstamp(5);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing prefetch
#ifdef SIXDGS_ABLATION
    if ((ABL & 2048) && lane == 0 && grp < 64) {
#pragma unroll
      for (int ph = 0; ph < 6; ++ph) atomicAdd(&g_dbg_cycles[wave][ph], t_acc[ph]);
      if (wave == 0) atomicAdd(&g_dbg_cycles[0][7], 1ull);
    }
#endif
  }
  // merge the four partials of every token (2 ray halves of the lane layout x 2 waves wn) through the (now idle) ring
  __syncthreads();
  float(*part)[256][2] = reinterpret_cast<float(*)[256][2]>(lds);   // [wn * 2 + (lane >> 5)][token][max, sum]
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    float* pp = part[wn * 2 + (lane >> 5)][wm * 64 + tm * 32 + (lane & 31)];
    if (OUT == kOutUB) {     // sums relative to the fixed reference: merged as (max 0, sum)
      pp[0] = 0.f;
      pp[1] = (zs[tm][0].x + zs[tm][0].y) + (zs[tm][1].x + zs[tm][1].y);
    } else {
      pp[0] = m_run[tm];
      pp[1] = s_run[tm];
    }
  }
  __syncthreads();
  if (tid < 256) {
    float m = -INFINITY, sres = 0.f;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float mt = part[h][tid][0], st2 = part[h][tid][1];
      if (mt > -INFINITY) {
        const float mn = fmaxf(m, mt);
        sres = sres * __expf(m - mn) + st2 * __expf(mt - mn);
        m = mn;
      }
    }
    pout[2 * tid] = m;
    pout[2 * tid + 1] = sres;
  }
  if (PERS) __syncthreads();      // the merge buffer is the ring: done with it before the next group's prologue refills it
  } while (PERS && (grp += A.n_sets) < A.n_groups);
}

// fp32 rows [rows][384] (row stride ld) -> fp16 planes [row][12][2][32] of x * 2^s, one power-of-two scale per 128-row
// tile chosen so that the tile's largest magnitude lands in [2^13, 2^14); inv_scale[tile] = 2^-s.  One block per tile; the
// second sweep over the tile's 192 KiB hits L2.
__global__ void __launch_bounds__(256) k_split_tiles_f16(const float* __restrict__ src, int64_t rows, int64_t ld, char* __restrict__ dst,
                                                         float* __restrict__ inv_scale) {
  __shared__ float wmax[4];
  const int64_t row0 = (int64_t)blockIdx.x * 128;
  const int n = (int)min((int64_t)128, rows - row0);
  const float* base = src + row0 * ld;
  float m = 0.f;
  for (int i = threadIdx.x; i < n * 96; i += 256) {
    const int row = i / 96;
    const float4 v = *reinterpret_cast<const float4*>(base + (int64_t)row * ld + (i - row * 96) * 4);
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));   // NaN operands are skipped by fmaxf
  }
  m = sdg_wave_max(m);
  if (sdg_lane() == 0) wmax[sdg_wave()] = m;
  __syncthreads();
  m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  int sh = 0;
  if (m > 0.f && m < INFINITY) {
    int e;
    frexpf(m, &e);                 // m = f * 2^e, f in [0.5, 1)  ->  m * 2^(14 - e) in [2^13, 2^14)
    sh = 14 - e;
    sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
  }
  const float sc = ldexpf(1.f, sh);
  if (threadIdx.x == 0) inv_scale[blockIdx.x] = ldexpf(1.f, -sh);
  for (int i = threadIdx.x; i < n * 48; i += 256) {
    const int row = i / 48, k8 = i - row * 48;
    const float* sp = base + (int64_t)row * ld + k8 * 8;
    const float4 lo = *reinterpret_cast<const float4*>(sp);
    const float4 hi = *reinterpret_cast<const float4*>(sp + 4);
    const float x[8] = {lo.x * sc, lo.y * sc, lo.z * sc, lo.w * sc, hi.x * sc, hi.y * sc, hi.z * sc, hi.w * sc};
    f16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const _Float16 hh = (_Float16)x[e];
      h[e] = hh;
      l[e] = (_Float16)(x[e] - (float)hh);
    }
    char* d = dst + (row0 + row) * kRowF + (k8 >> 2) * kSlabF + (k8 & 3) * 16;
    *reinterpret_cast<f16x8*>(d) = h;
    *reinterpret_cast<f16x8*>(d + 64) = l;
  }
}


// merge the per-group partial statistics: stats[b][t] = (max, sumexp).  Four threads per token take every fourth group
// (independent load chains), then one thread folds the four partials in a fixed order.
// 16 lanes per token (round 3; was 4: 256 partials merged one after the other per thread made this 0.14 ms -- 2 % of a single-image step):
// block (image, token quarter) = 64 tokens x 16 lanes, lane j merges the groups j, j + 16, ..., the 16 results in ascending j.
__global__ void __launch_bounds__(1024) k_merge_stats(const float* __restrict__ partial, int n_groups, float* __restrict__ stats) {
  __shared__ float sm[16][64][2];
  const int bl = blockIdx.x, tl = threadIdx.x & 63, j = threadIdx.x >> 6, t = blockIdx.y * 64 + tl;
  const float* p = partial + (int64_t)bl * n_groups * kT * 2;
  float m = -INFINITY, s = 0.f;
  for (int g = j; g < n_groups; g += 16) {
    const float mt = p[((int64_t)g * kT + t) * 2], st = p[((int64_t)g * kT + t) * 2 + 1];
    if (mt > -INFINITY) {
      const float mn = fmaxf(m, mt);
      s = s * expf(m - mn) + st * expf(mt - mn);
      m = mn;
    }
  }
  sm[j][tl][0] = m;
  sm[j][tl][1] = s;
  __syncthreads();
  if (j == 0) {
    m = -INFINITY;
    s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float mt = sm[q][tl][0], st = sm[q][tl][1];
      if (mt > -INFINITY) {
        const float mn = fmaxf(m, mt);
        s = s * expf(m - mn) + st * expf(mt - mn);
        m = mn;
      }
    }
    stats[((int64_t)bl * kT + t) * 2] = m;
    stats[((int64_t)bl * kT + t) * 2 + 1] = s;
  }
}

// pass 2: score[r] = sum_{t < T} exp(l[t][r] - max_t) / sumexp_t   (softmax over rays, summed over tokens)
__global__ void __launch_bounds__(256) k_score_reduce(const float* __restrict__ logits, int64_t ldl, const float* __restrict__ stats,
                                                       const int* __restrict__ n_tok, int b0, int64_t R, float* __restrict__ scores,
                                                       int64_t score_stride) {
  __shared__ float st[kT][2];
  const int bl = blockIdx.y;
  const int T = n_tok[b0 + bl];
  for (int i = threadIdx.x; i < kT * 2; i += blockDim.x) {   // (max, 1 / sumexp): one multiply per logit instead of a division
    const float v = stats[(int64_t)bl * kT * 2 + i];
    (&st[0][0])[i] = (i & 1) ? 1.f / v : v;
  }
  __syncthreads();
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= R) return;
  const float* l = logits + (int64_t)bl * kT * ldl + j;
  float s = 0.f;
#pragma unroll 8
  for (int t = 0; t < T; ++t) s += __expf(l[(int64_t)t * ldl] - st[t][0]) * st[t][1];
  scores[(int64_t)bl * score_stride + j] = s;
}

#define FASTEXP(x) __expf(x)   // v_exp_f32 path (2 ulp); with the reciprocal above the pass becomes HBM-bound instead of VALU-bound
// pass 2 on the blocked logits of k_logits_f16x ([tile][token group 8][ray quad 32][token 32][ray 4]): a wave owns a pair of
// ray quads (8 rays): lanes 0..31 / 32..63 hold the 32 tokens of a group for quad 2p / 2p+1, every load instruction reads
// 1 KiB contiguous, the 8 token groups accumulate in registers and one butterfly over 32 lanes finishes the column sums.
__global__ void __launch_bounds__(256) k_score_reduce_blocked(const float* __restrict__ logits, int64_t ldl, const float* __restrict__ stats,
                                                               const int* __restrict__ n_tok, int b0, int64_t R,
                                                               float* __restrict__ scores, int64_t score_stride) {
  const int bl = blockIdx.y;
  const int T = n_tok[b0 + bl];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31;
  float mt[8], st[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int t = g * 32 + l31;
    mt[g] = stats[((int64_t)bl * kT + t) * 2];
    st[g] = 1.f / stats[((int64_t)bl * kT + t) * 2 + 1];     // reciprocal once per token: 1 multiply per logit instead of a division
  }
  const int64_t tile = blockIdx.x;
  const float* tb = logits + (int64_t)bl * kT * ldl + tile * (kT * kBN) + lane * 4;
  float* out = scores + (int64_t)bl * score_stride + tile * kBN;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int p = u * 4 + wave;                 // quad pair
    float4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 8; ++g)
      if (g * 32 + l31 < T) {
        const float4 v = *reinterpret_cast<const float4*>(tb + g * 4096 + p * 256);
        acc.x += FASTEXP(v.x - mt[g]) * st[g];
        acc.y += FASTEXP(v.y - mt[g]) * st[g];
        acc.z += FASTEXP(v.z - mt[g]) * st[g];
        acc.w += FASTEXP(v.w - mt[g]) * st[g];
      }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      acc.x += __shfl_xor(acc.x, o, 64);
      acc.y += __shfl_xor(acc.y, o, 64);
      acc.z += __shfl_xor(acc.z, o, 64);
      acc.w += __shfl_xor(acc.w, o, 64);
    }
    if (l31 == 0) {
      const int64_t r0 = tile * kBN + p * 8 + (lane >> 5) * 4;
      float* o4 = out + p * 8 + (lane >> 5) * 4;
      if (r0 + 3 < R && ((uintptr_t)o4 & 15) == 0) *reinterpret_cast<float4*>(o4) = acc;
      else {   // ragged end of the scene, or a caller buffer whose row stride R is not a multiple of 4
        if (r0 < R) o4[0] = acc.x;
        if (r0 + 1 < R) o4[1] = acc.y;
        if (r0 + 2 < R) o4[2] = acc.z;
        if (r0 + 3 < R) o4[3] = acc.w;
      }
    }
  }
}

// the same on the 24-bit fixed-point logits (kTileBytes24)
__global__ void __launch_bounds__(256) k_score_reduce_blocked24(const float* __restrict__ logits, int64_t ldl, const float* __restrict__ stats,
                                                               const int* __restrict__ n_tok, int b0, int64_t R,
                                                               float* __restrict__ scores, int64_t score_stride) {
  const int bl = blockIdx.y;
  const int T = n_tok[b0 + bl];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31;
  float mt[8], st[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int t = g * 32 + l31;
    mt[g] = stats[((int64_t)bl * kT + t) * 2];
    st[g] = 1.f / stats[((int64_t)bl * kT + t) * 2 + 1];     // reciprocal once per token: 1 multiply per logit instead of a division
  }
  const int64_t tile = blockIdx.x;
  const char* tb = reinterpret_cast<const char*>(logits) + (int64_t)bl * kT * ldl * 4 + tile * kTileBytes24;
  float refv[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) refv[g] = reinterpret_cast<const float*>(tb + 98304)[(g * 2 + (lane >> 5)) * 32 + l31];
  float* out = scores + (int64_t)bl * score_stride + tile * kBN;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int p = u * 4 + wave;                 // quad pair
    float4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 8; ++g)
      if (g * 32 + l31 < T) {
        typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
        const u32x3 w3 = *reinterpret_cast<const u32x3*>(tb + g * 12288 + p * 768 + lane * 12);
        float4 v;
        v.x = refv[g] - (float)(w3.x & 0xffffffu) * 1.9073486328125e-06f;
        v.y = refv[g] - (float)((w3.x >> 24) | ((w3.y & 0xffffu) << 8)) * 1.9073486328125e-06f;
        v.z = refv[g] - (float)((w3.y >> 16) | ((w3.z & 0xffu) << 16)) * 1.9073486328125e-06f;
        v.w = refv[g] - (float)(w3.z >> 8) * 1.9073486328125e-06f;
        acc.x += FASTEXP(v.x - mt[g]) * st[g];
        acc.y += FASTEXP(v.y - mt[g]) * st[g];
        acc.z += FASTEXP(v.z - mt[g]) * st[g];
        acc.w += FASTEXP(v.w - mt[g]) * st[g];
      }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      acc.x += __shfl_xor(acc.x, o, 64);
      acc.y += __shfl_xor(acc.y, o, 64);
      acc.z += __shfl_xor(acc.z, o, 64);
      acc.w += __shfl_xor(acc.w, o, 64);
    }
    if (l31 == 0) {
      const int64_t r0 = tile * kBN + p * 8 + (lane >> 5) * 4;
      float* o4 = out + p * 8 + (lane >> 5) * 4;
      if (r0 + 3 < R && ((uintptr_t)o4 & 15) == 0) *reinterpret_cast<float4*>(o4) = acc;
      else {   // ragged end of the scene, or a caller buffer whose row stride R is not a multiple of 4
        if (r0 < R) o4[0] = acc.x;
        if (r0 + 1 < R) o4[1] = acc.y;
        if (r0 + 2 < R) o4[2] = acc.z;
        if (r0 + 3 < R) o4[3] = acc.w;
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// top-k: 4-pass MSB radix select on an order-preserving key, ordered gather, bitonic sort
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned score_key(float v) {
  unsigned u = __float_as_uint(v);
  if (u == 0x80000000u) u = 0u;          // -0.0 == +0.0: one key, so that the tie goes to the lower index
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_score(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

struct SelectState {
  unsigned prefix;   // high bits of the k-th largest key decided so far
  unsigned remain;   // how many are still needed among keys matching the prefix
};
// Re-derive the selection after `passes` histogram passes (each block does this redundantly: <= 3x256 bins).
__device__ SelectState select_state(const unsigned* hist /*[4][256] of this image*/, int passes, unsigned k, unsigned* sm /*[256]*/) {
  SelectState st = {0u, k};
  for (int p = 0; p < passes; ++p) {
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) sm[i] = hist[p * 256 + i];
    __syncthreads();
    // every thread scans from the top bin (256 adds; uniform)
    unsigned acc = 0, digit = 0, rem = st.remain;
    for (int d = 255; d >= 0; --d) {
      const unsigned c = sm[d];
      if (acc + c >= st.remain) { digit = (unsigned)d; rem = st.remain - acc; break; }
      acc += c;
    }
    st.prefix |= digit << (24 - 8 * p);
    st.remain = rem;
  }
  return st;
}

template <int PASS>
__global__ void __launch_bounds__(256) k_topk_hist(const float* __restrict__ scores, int64_t stride, int64_t R, unsigned k,
                                                    unsigned* __restrict__ hist_all, const int* __restrict__ only) {
  __shared__ unsigned sm[256];
  __shared__ unsigned lh[256];
  const int b = blockIdx.y;
  if (only != nullptr && only[b] == 0) return;      // (an image this selection is not needed for: the whole block leaves)
  unsigned* hist = hist_all + (int64_t)b * 4 * 256;
  const SelectState st = select_state(hist, PASS, k, sm);
  for (int i = threadIdx.x; i < 256; i += blockDim.x) lh[i] = 0;
  __syncthreads();
  const float* s = scores + (int64_t)b * stride;
  constexpr unsigned hi_mask = PASS == 0 ? 0u : (0xffffffffu << (32 - 8 * PASS));
  constexpr int shift = 24 - 8 * PASS;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < R; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned key = score_key(s[i]);
    if ((key & hi_mask) == st.prefix) atomicAdd(&lh[(key >> shift) & 255u], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += blockDim.x)
    if (lh[i]) atomicAdd(&hist[PASS * 256 + i], lh[i]);
}

// ordered gather.  Block c owns the contiguous index range [c*span, (c+1)*span).
// COUNT: counts[b][c] = (#greater, #equal) in the range.  WRITE: emits candidates in index order:
// greater ones at their global rank, equal ones after all greater ones while rank_eq < remain.
template <bool WRITE>
__global__ void __launch_bounds__(256) k_topk_gather(const float* __restrict__ scores, int64_t stride, int64_t R, unsigned k,
                                                      const unsigned* __restrict__ hist_all, int64_t span,
                                                      unsigned* __restrict__ counts /*[B][NB][2]*/,
                                                      const unsigned* __restrict__ offs /*[B][NB][2] exclusive*/,
                                                      unsigned* __restrict__ cand_key, int64_t* __restrict__ cand_idx, int topk,
                                                      const int* __restrict__ only) {
  __shared__ unsigned sm[256];
  __shared__ int sc[2][5];
  const int b = blockIdx.y;
  if (only != nullptr && only[b] == 0) return;
  const SelectState st = select_state(hist_all + (int64_t)b * 4 * 256, 4, k, sm);
  const unsigned thr = st.prefix;
  const float* s = scores + (int64_t)b * stride;
  const int64_t i0 = (int64_t)blockIdx.x * span, i1 = min(i0 + span, R);
  unsigned run_g = 0, run_e = 0;
  unsigned base_g = 0, base_e = 0, n_greater = 0;
  if (WRITE) {
    base_g = offs[((int64_t)b * gridDim.x + blockIdx.x) * 2];
    base_e = offs[((int64_t)b * gridDim.x + blockIdx.x) * 2 + 1];
    n_greater = k - st.remain;
  }
  for (int64_t c0 = i0; c0 < i1; c0 += 256) {
    const int64_t i = c0 + threadIdx.x;
    unsigned key = 0;
    bool gt = false, eq = false;
    if (i < i1) {
      key = score_key(s[i]);
      gt = key > thr;
      eq = key == thr;
    }
    const unsigned long long bg = __ballot(gt), be = __ballot(eq);
    const int lane = sdg_lane(), wv = sdg_wave();
    if (lane == 0) { sc[0][wv] = __popcll(bg); sc[1][wv] = __popcll(be); }
    __syncthreads();
    unsigned og = 0, oe = 0, tg = 0, te = 0;
    for (int w2 = 0; w2 < 4; ++w2) {
      if (w2 < wv) { og += sc[0][w2]; oe += sc[1][w2]; }
      tg += sc[0][w2];
      te += sc[1][w2];
    }
    __syncthreads();
    if (WRITE) {
      const unsigned long long below = (1ull << lane) - 1ull;
      if (gt) {
        const unsigned pos = base_g + run_g + og + __popcll(bg & below);
        if (pos < (unsigned)topk) { cand_key[(int64_t)b * topk + pos] = key; cand_idx[(int64_t)b * topk + pos] = i; }
      } else if (eq) {
        const unsigned re = base_e + run_e + oe + __popcll(be & below);
        if (re < st.remain) {
          const unsigned pos = n_greater + re;
          if (pos < (unsigned)topk) { cand_key[(int64_t)b * topk + pos] = key; cand_idx[(int64_t)b * topk + pos] = i; }
        }
      }
    }
    run_g += tg;
    run_e += te;
  }
  if (!WRITE && threadIdx.x == 0) {
    counts[((int64_t)b * gridDim.x + blockIdx.x) * 2] = run_g;
    counts[((int64_t)b * gridDim.x + blockIdx.x) * 2 + 1] = run_e;
  }
}

__global__ void __launch_bounds__(1024) k_topk_scan(const unsigned* __restrict__ counts, int nb, unsigned* __restrict__ offs, const int* __restrict__ only) {
  __shared__ unsigned sm[17];
  const int b = blockIdx.x;
  if (only != nullptr && only[b] == 0) return;
  for (int comp = 0; comp < 2; ++comp) {
    unsigned base = 0;
    for (int c0 = 0; c0 < nb; c0 += 1024) {
      const int i = c0 + threadIdx.x;
      const unsigned v = i < nb ? counts[((int64_t)b * nb + i) * 2 + comp] : 0u;
      unsigned tot;
      const unsigned ex = sdg_block_exclusive_scan<unsigned, 16>(v, sm, &tot);
      if (i < nb) offs[((int64_t)b * nb + i) * 2 + comp] = base + ex;
      base += tot;
    }
  }
}

// sort the <= 1024 candidates by (key desc, index asc) and emit idx/val; pads with (-1, NaN)
__global__ void __launch_bounds__(1024) k_topk_sort(const unsigned* __restrict__ cand_key, const int64_t* __restrict__ cand_idx,
                                                     int topk, int k_eff, int64_t* __restrict__ idx, float* __restrict__ val, const int* __restrict__ only) {
  __shared__ unsigned sk[1024];
  __shared__ long long si[1024];
  const int b = blockIdx.x, t = threadIdx.x;
  if (only != nullptr && only[b] == 0) return;
  if (t < k_eff) { sk[t] = cand_key[(int64_t)b * topk + t]; si[t] = cand_idx[(int64_t)b * topk + t]; }
  else { sk[t] = 0u; si[t] = 0x7fffffffffffffffll; }
  __syncthreads();
  for (int size = 2; size <= 1024; size <<= 1)
    for (int strd = size >> 1; strd > 0; strd >>= 1) {
      const int p = t ^ strd;
      if (p > t) {
        const bool up = (t & size) == 0;   // "up" block: best-first order
        const unsigned ka = sk[t], kb = sk[p];
        const long long ia = si[t], ib = si[p];
        const bool a_first = ka > kb || (ka == kb && ia < ib);   // a ranks before b
        if (up ? !a_first : a_first) { sk[t] = kb; sk[p] = ka; si[t] = ib; si[p] = ia; }
      }
      __syncthreads();
    }
  if (t < topk) {
    if (t < k_eff) { idx[(int64_t)b * topk + t] = si[t]; val[(int64_t)b * topk + t] = key_score(sk[t]); }
    else { idx[(int64_t)b * topk + t] = -1; val[(int64_t)b * topk + t] = NAN; }
  }
}

// ------------------------------------------------------------------------------------------------
// Select path: the top-k WITHOUT the [T, R] logits ever reaching HBM (inference: no score vector is asked for).
//   score[r] = sum_t e[t][r] / Z_t,  e = exp(logit - ref_t),  Z_t = sum_r e[t][r].
// A pre-pass over a 1/f ray sample gives ref_t (the sample maximum) and Z~_t (so that f Z~_t ~ Z_t).  The main sweep forms
//   U[r] = sum_t e[t][r] / (f Z~_t)   (4 B per ray and image instead of 784)   and the EXACT   g_t = Z_t / (f Z~_t),
// so that score[r] = sum_t e'[t][r] / g_t lies in [U[r] / g_max, U[r] / g_min].  Every ray of the true top-k therefore has
// U[r] >= U_(k) g_min / g_max (U_(k) = k-th largest U): those few rays are re-scored exactly from their key planes and the
// top-k of the exact scores is the result -- the sample only decides how many candidates there are, never the answer.
// ------------------------------------------------------------------------------------------------
constexpr float kLog2e = 1.4426950408889634f;

// ctok[b][t] = -ref_t log2e - log2(f Z~_t)   (-inf for padding tokens: their e' is exactly 0)
__global__ void __launch_bounds__(kT) k_sel_prepare(const float* __restrict__ stats_s, const int* __restrict__ n_tok, int b0, float log2_frac,
                                                    float* __restrict__ ctok) {
  const int bl = blockIdx.x, t = threadIdx.x;
  const float m = stats_s[((int64_t)bl * kT + t) * 2], z = stats_s[((int64_t)bl * kT + t) * 2 + 1];
  const bool ok = t < n_tok[b0 + bl] && z > 0.f && m > -INFINITY && m < INFINITY;
  ctok[(int64_t)bl * kT + t] = ok ? -(m * kLog2e) - (log2f(z) + log2_frac) : -INFINITY;
}

// ---- token packing (round 5; sweep_plan.h): the small kernels around the sweep that know which image sits in which quarter of which slot ----------
// The sweep kernel itself does not: it sees `n_slots` "images" of up to 256 token rows, their planes, per-row ctok and row counts.
//
// q rows of the launch's slots -> scaled fp16 planes [slot][256 rows][1536 B] with one power-of-two scale per QUARTER (the scale rule of
// k_split_tiles_f16 on 64 rows: an image's planes do not depend on the quarter it lands in, nor on its neighbours), inv_scale [slot][4];
// rows at or beyond the image's token count and unassigned quarters are zero.  Also the slot's ctok row (or null: the sample pre-pass has none) and
// its number of token rows (what the sweep kernel takes as the "image's" token count).  One workgroup per (slot, quarter).
__global__ void __launch_bounds__(256) k_split_q_slots(const float* __restrict__ q, const int* __restrict__ n_tok, const SweepSlots T, char* __restrict__ dst,
                                                       float* __restrict__ inv_scale, const float* __restrict__ ctok, float* __restrict__ ctok_slot,
                                                       int* __restrict__ slot_rows) {
  __shared__ float wmax[4];
  const int slot = blockIdx.x >> 2, w = blockIdx.x & 3;
  const int img = T.q_img[slot][w], lq = T.q_lq[slot][w];
  const int n = img < 0 ? 0 : min(64, max(0, n_tok[img] - 64 * lq));
  const float* base = q + ((int64_t)(img < 0 ? 0 : img) * kT + 64 * (lq < 0 ? 0 : lq)) * SIXDGS_D;
  float m = 0.f;
  for (int i = threadIdx.x; i < n * 96; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(base + (int64_t)i * 4);
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  m = sdg_wave_max(m);
  if (sdg_lane() == 0) wmax[sdg_wave()] = m;
  __syncthreads();
  m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  int sh = 0;
  if (m > 0.f && m < INFINITY) {
    int e;
    frexpf(m, &e);
    sh = 14 - e;
    sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
  }
  const float sc = ldexpf(1.f, sh);
  if (threadIdx.x == 0) inv_scale[blockIdx.x] = ldexpf(1.f, -sh);
  char* const drow = dst + ((int64_t)slot * kT + 64 * w) * kRowF;
  for (int i = threadIdx.x; i < 64 * 48; i += 256) {
    const int row = i / 48, k8 = i - row * 48;
    f16x8 h, l;
    if (row < n) {
      const float* sp = base + (int64_t)row * SIXDGS_D + k8 * 8;
      const float4 lo = *reinterpret_cast<const float4*>(sp);
      const float4 hi = *reinterpret_cast<const float4*>(sp + 4);
      const float x[8] = {lo.x * sc, lo.y * sc, lo.z * sc, lo.w * sc, hi.x * sc, hi.y * sc, hi.z * sc, hi.w * sc};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const _Float16 hh = (_Float16)x[e];
        h[e] = hh;
        l[e] = (_Float16)(x[e] - (float)hh);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = l[e] = (_Float16)0.f;
    }
    char* d = drow + (int64_t)row * kRowF + (k8 >> 2) * kSlabF + (k8 & 3) * 16;
    *reinterpret_cast<f16x8*>(d) = h;
    *reinterpret_cast<f16x8*>(d + 64) = l;
  }
  if (ctok_slot != nullptr && threadIdx.x < 64) {
    const int t = threadIdx.x;
    ctok_slot[(int64_t)slot * kT + 64 * w + t] = t < n ? ctok[(int64_t)img * kT + 64 * lq + t] : -INFINITY;
  }
  if (w == 0 && threadIdx.x == 0) {      // token rows of the slot: up to the last row of the last quarter that holds any
    int rows = 0;
    for (int x = 0; x < kSlotQuarters; ++x) {
      const int im = T.q_img[slot][x];
      const int nx = im < 0 ? 0 : min(64, max(0, n_tok[im] - 64 * T.q_lq[slot][x]));
      if (nx > 0) rows = 64 * x + nx;
    }
    slot_rows[slot] = rows;
  }
}

// k_merge_stats per IMAGE of a packed launch: image T.img[blockIdx.x], its token quarter blockIdx.y (local tokens 64 y ..) sits in (slot, tile quarter)
// T.img_q[..][y] of the launch's partials.  The same merge order as k_merge_stats (16 lanes per token over the groups, then ascending).  gsum != null:
// gsum[img][t] += the merged sum (the sweep: partials are (0, sum) pairs); otherwise stats[img][t] = (max, sumexp) (the sample pre-pass).  Quarters the
// image does not have: (-inf, 0) / nothing added.
__global__ void __launch_bounds__(1024) k_merge_stats_slots(const float* __restrict__ partial, int n_groups, const SweepSlots T, float* __restrict__ stats,
                                                            float* __restrict__ gsum) {
  __shared__ float sm[16][64][2];
  const int k = blockIdx.x, y = blockIdx.y, tl = threadIdx.x & 63, j = threadIdx.x >> 6;
  const int img = T.img[k], sq = y < T.img_nq[k] ? T.img_q[k][y] : -1;
  float m = -INFINITY, s = 0.f;
  if (sq >= 0) {
    const float* p = partial + (int64_t)(sq >> 2) * n_groups * kT * 2;
    const int row = 64 * (sq & 3) + tl;
    for (int g = j; g < n_groups; g += 16) {
      const float mt = p[((int64_t)g * kT + row) * 2], st = p[((int64_t)g * kT + row) * 2 + 1];
      if (mt > -INFINITY) {
        const float mn = fmaxf(m, mt);
        s = s * expf(m - mn) + st * expf(mt - mn);
        m = mn;
      }
    }
  }
  sm[j][tl][0] = m;
  sm[j][tl][1] = s;
  __syncthreads();
  if (j == 0) {
    m = -INFINITY;
    s = 0.f;
#pragma unroll
    for (int qd = 0; qd < 16; ++qd) {
      const float mt = sm[qd][tl][0], st = sm[qd][tl][1];
      if (mt > -INFINITY) {
        const float mn = fmaxf(m, mt);
        s = s * expf(m - mn) + st * expf(mt - mn);
        m = mn;
      }
    }
    const int64_t o = (int64_t)img * kT + 64 * y + tl;
    if (gsum != nullptr) {
      gsum[o] += s;
    } else {
      stats[o * 2] = m;
      stats[o * 2 + 1] = s;
    }
  }
}

// k_sel_finish per IMAGE of a packed launch: U[img][r] = sum over the image's quarters, in quarter order, of their rows of partial sums (row 4 slot +
// tile quarter of ub, wherever the quarter was laid).  The quarters read are those the table gave the image AND its device token count fills (the others
// were never written).  The table is planned from the caller's HOST copy of the token counts: should that copy promise fewer quarters than the image
// has tokens for, the tokens beyond them get no sum (k_merge_stats_slots adds nothing: their gsum stays 0) and k_sel_bounds reports the image
// undecidable (g_t = 0 for a token that exists) -- it is then scored by the two-pass path instead of silently without some of its tokens.
// tmax (or null): the largest U of every 256-ray tile = of the 4 x 64 rays one wave of this kernel handles.  The k-th largest tile maximum is a lower
// bound of the k-th largest U (k tiles hold a ray that large) and, with the top rays scattered over 10^5 tiles, almost equal to it: the candidate stage
// takes its threshold from the tile maxima (r / 256 values) instead of a radix select over all r values of U.
__global__ void __launch_bounds__(256) k_sel_finish_slots(const float* __restrict__ ub, int64_t stride, int64_t u_stride, const int* __restrict__ n_tok, const SweepSlots T,
                                                          int64_t R, float* __restrict__ U, float* __restrict__ tmax, int64_t tmax_stride) {
  const int k = blockIdx.y, img = T.img[k];
  const int nq_dev = (min(max(n_tok[img], 0), kT) + 63) >> 6;
  const int nq = min(nq_dev, (int)T.img_nq[k]);
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  float m = -INFINITY;
  if (i < R) {
    float4 a = {0.f, 0.f, 0.f, 0.f};
    const int64_t row[kSlotQuarters] = {T.img_q[k][0], T.img_q[k][1], T.img_q[k][2], T.img_q[k][3]};      // (scalar loads, once per thread)
#pragma unroll
    for (int y = 0; y < kSlotQuarters; ++y)
      if (y < nq) {
        const float4 v = *reinterpret_cast<const float4*>(ub + row[y] * stride + i);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    *reinterpret_cast<float4*>(U + (int64_t)img * u_stride + i) = a;     // rows are padded to whole 256-ray tiles: the tail past R is scratch
    m = a.x;
    if (i + 1 < R) m = fmaxf(m, a.y);
    if (i + 2 < R) m = fmaxf(m, a.z);
    if (i + 3 < R) m = fmaxf(m, a.w);
  }
  if (tmax != nullptr) {
    m = sdg_wave_max(m);
    const int64_t tile = (i - 4 * sdg_lane()) >> 8;      // the wave's first ray / 256
    if (sdg_lane() == 0 && tile * 256 < R) tmax[(int64_t)img * tmax_stride + tile] = m;
  }
}

// max over the rows of |x_row| for the values the scaled fp16 planes hold, x = (h + l) * inv_scale[row / 128]: one wave per row (48 lanes
// x one 16-byte chunk of h and its partner of l).  *out is UPDATED (bit pattern of a non-negative float: atomicMax): callers that feed
// the scene in chunks accumulate.  The result is rounded UP (x (1 + 2^-12) covers the fp32 sum of 384 squares) -- it is used as a bound.
__global__ void __launch_bounds__(256) k_plane_norm_max(const char* __restrict__ planes, const float* __restrict__ inv_scale, int64_t rows,
                                                        unsigned* __restrict__ out) {
  const int lane = sdg_lane();
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + sdg_wave(), nw = (int64_t)gridDim.x * 4;
  float best = 0.f;
  for (int64_t row = wave0; row < rows; row += nw) {
    float ss = 0.f;
    if (lane < 48) {
      const char* p = planes + row * kRowF + (lane >> 2) * kSlabF + (lane & 3) * 16;
      const f16x8 h = *reinterpret_cast<const f16x8*>(p), l = *reinterpret_cast<const f16x8*>(p + 64);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = (float)h[e] + (float)l[e];       // exact: 22 significant bits
        ss = __builtin_fmaf(x, x, ss);
      }
    }
    ss = sdg_wave_sum(ss);
    best = fmaxf(best, sqrtf(ss) * inv_scale[row >> 7]);
  }
  if (lane == 0 && best > 0.f) atomicMax(out, __float_as_uint(best * 1.000244140625f));
}

// per image: g_min / g_max over its tokens, the candidate threshold on U and the validity of the bounds.
// info[bl][4] = {threshold on U, flag, eps, x}: flag 0 ok, 1 = no tokens (all scores are exactly 0), 2 = bounds unusable (overflow / NaN)
//
// The slack of the threshold is DERIVED, not tuned.  What is compared are two evaluations of the same quantity: U[r]
// from the sweep (matrix-core accumulation, fp32 epilogue) and the candidates' exact re-score (fp64 accumulation of the same plane
// products, the same epilogue constants).  With x = max_t |q_t| * max_r |k_r| / sqrt(384) -- a bound on sum_i |q_i k_i| / sqrt(384),
// hence on every |logit|, by Cauchy-Schwarz -- one e'[t][r] of the sweep differs from the re-score's by a relative
//   eps = x (1151 * 2u (1 + 2^-9) + 2^-22 + u)    fp32 accumulation of 1152 products inside the MFMAs (worst case, any order, 2u per
//                                                 addition: covers a truncating as well as a round-to-nearest adder), the dropped
//                                                 l x l term, the rounding of the exact sum to fp32                          <= 1.38e-4 x
//       + 2 * 128 u ln2                           the fma forming the exp2 argument (|argument| <= 128 or e' is 0 / inf), both sides
//       + 2 * 2 u                                 v_exp_f32 (1 ulp), both sides
//       + 2 * 10 u                                the sums over the 256 tokens (trees of depth <= 10), both sides,    u = 2^-24
// and so does U[r] (a sum of non-negative e').  A ray of the true top-k has S = sum_t e'_exact / g_t >= U_(k) / ((1 + eps) g_max) and
// S <= U[r] / ((1 - eps) g_min):   U[r] >= U_(k) (g_min / g_max) (1 - eps) / (1 + eps).
constexpr float kEpsPerX = 1.4e-4f, kEpsConst = 1.3e-5f;
__global__ void __launch_bounds__(kT) k_sel_bounds(const float* __restrict__ gsum, const int* __restrict__ n_tok, const float* __restrict__ valU,
                                                   const float* __restrict__ q, const float* __restrict__ key_norm_max, int topk, int k_eff,
                                                   float* __restrict__ info, const int* __restrict__ only) {
  __shared__ float smin[4], smax[4], sq[4];
  const int bl = blockIdx.x, t = threadIdx.x, M = n_tok[bl];
  if (only != nullptr && only[bl] == 0) return;
  const float g = gsum[(int64_t)bl * kT + t];
  float lo = t < M ? g : INFINITY, hi = t < M ? g : -INFINITY;
  bool bad = t < M && !(g > 0.f && g < INFINITY);
  float qn = 0.f;
  if (t < M) {
    const float4* qr = reinterpret_cast<const float4*>(q + ((int64_t)bl * kT + t) * SIXDGS_D);
    float ss = 0.f;
    for (int i = 0; i < SIXDGS_D / 4; ++i) {
      const float4 v = qr[i];
      ss = __builtin_fmaf(v.x, v.x, ss); ss = __builtin_fmaf(v.y, v.y, ss); ss = __builtin_fmaf(v.z, v.z, ss); ss = __builtin_fmaf(v.w, v.w, ss);
    }
    qn = sqrtf(ss) * 1.000244140625f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o, 64));
    hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    qn = fmaxf(qn, __shfl_xor(qn, o, 64));
  }
  const unsigned long long anybad = __ballot(bad);
  if (sdg_lane() == 0) { smin[sdg_wave()] = anybad ? NAN : lo; smax[sdg_wave()] = hi; sq[sdg_wave()] = qn; }
  __syncthreads();
  if (t == 0) {
    float flag = 0.f, thr = INFINITY, eps = 0.f, x = 0.f;
    if (M <= 0) flag = 1.f;
    else {
      float a = smin[0], b = smax[0], qm = sq[0];
      bool nan = a != a;
      for (int w = 1; w < 4; ++w) { nan = nan || smin[w] != smin[w]; a = fminf(a, smin[w]); b = fmaxf(b, smax[w]); qm = fmaxf(qm, sq[w]); }
      const float uk = valU[(int64_t)bl * topk + (k_eff - 1)];
      x = qm * key_norm_max[0] * kInvSqrtD * 1.0000002f;
      eps = x * kEpsPerX + kEpsConst;
      if (nan || !(uk >= 0.f && uk < INFINITY) || !(a > 0.f) || !(eps < 0.25f)) flag = 2.f;
      else thr = (uk * (a / b)) * ((1.f - eps) / (1.f + eps)) * 0.9999998f;      // the last factor: the roundings of this very line
    }
    info[bl * 4 + 0] = thr;
    info[bl * 4 + 1] = flag;
    info[bl * 4 + 2] = eps;
    info[bl * 4 + 3] = x;
  }
}

// need[bl] = 1 when the threshold taken from the TILE maxima admitted more than cmax candidates (the top rays sit in few tiles, the k-th
// largest tile maximum is then far below the k-th largest U): that image gets the exact k-th largest U after all
__global__ void k_sel_need_exact(const int* __restrict__ total, const float* __restrict__ info, int nb, int cmax, int* __restrict__ need) {
  const int bl = blockIdx.x * blockDim.x + threadIdx.x;
  if (bl >= nb) return;
  need[bl] = (info[bl * 4 + 1] == 0.f && total[bl] > cmax) ? 1 : 0;
}

// d_count[bl] = number of candidates (may exceed max_candidates: the re-score stage then refuses), -1 bounds unusable, -2 no tokens
__global__ void k_sel_count(const int* __restrict__ total, const float* __restrict__ info, int nb, int* __restrict__ d_count) {
  const int bl = blockIdx.x * blockDim.x + threadIdx.x;
  if (bl >= nb) return;
  const float flag = info[bl * 4 + 1];
  d_count[bl] = flag == 1.f ? -2 : (flag != 0.f ? -1 : total[bl]);
}

// ordered compaction of {r : U[r] >= threshold}: COUNT -> counts[bl][blk]; WRITE -> cand[bl][pos] = r in ascending r for pos < cmax
template <bool WRITE>
__global__ void __launch_bounds__(256) k_sel_candidates(const float* __restrict__ U, int64_t stride, int64_t R, const float* __restrict__ info,
                                                        int64_t span, unsigned* __restrict__ counts, const unsigned* __restrict__ offs,
                                                        int64_t* __restrict__ cand, int cmax, const int* __restrict__ only,
                                                        const float* __restrict__ tmax_all) {
  __shared__ int sc[4];
  const int bl = blockIdx.y;
  if (only != nullptr && only[bl] == 0) return;
  const float thr = info[bl * 4 + 0];
  const float* u = U + (int64_t)bl * stride;
  const float* tmax = tmax_all ? tmax_all + (int64_t)bl * (stride >> 8) : nullptr;
  const int64_t i0 = (int64_t)blockIdx.x * span, i1 = min(i0 + span, R);
  unsigned run = 0;
  const unsigned base = WRITE ? offs[(int64_t)bl * gridDim.x + blockIdx.x] : 0u;
  for (int64_t c0 = i0; c0 < i1; c0 += 256) {
    if (tmax != nullptr && !(tmax[c0 >> 8] >= thr)) continue;      // no ray of this 256-ray tile reaches the threshold: 4 B read instead of 1 KB
    const int64_t i = c0 + threadIdx.x;
    const bool in = i < i1 && u[i] >= thr;
    const unsigned long long bm = __ballot(in);
    const int lane = sdg_lane(), wv = sdg_wave();
    if (lane == 0) sc[wv] = __popcll(bm);
    __syncthreads();
    unsigned before = 0, tot = 0;
    for (int w2 = 0; w2 < 4; ++w2) {
      if (w2 < wv) before += sc[w2];
      tot += sc[w2];
    }
    __syncthreads();
    if (WRITE && in) {
      const unsigned pos = base + run + before + __popcll(bm & ((1ull << lane) - 1ull));
      if (pos < (unsigned)cmax) cand[(int64_t)bl * cmax + pos] = i;
    }
    run += tot;
  }
  if (!WRITE && threadIdx.x == 0) counts[(int64_t)bl * gridDim.x + blockIdx.x] = run;
}

// exclusive scan of the per-block candidate counts; total[bl] = number of candidates
__global__ void __launch_bounds__(1024) k_sel_scan(const unsigned* __restrict__ counts, int nb, unsigned* __restrict__ offs, int* __restrict__ total) {
  __shared__ unsigned sm[17];
  const int bl = blockIdx.x;
  unsigned base = 0;
  for (int c0 = 0; c0 < nb; c0 += 1024) {
    const int i = c0 + threadIdx.x;
    const unsigned v = i < nb ? counts[(int64_t)bl * nb + i] : 0u;
    unsigned tot;
    const unsigned ex = sdg_block_exclusive_scan<unsigned, 16>(v, sm, &tot);
    if (i < nb) offs[(int64_t)bl * nb + i] = base + ex;
    base += tot;
  }
  if (threadIdx.x == 0) total[bl] = (int)min(base, 0x7fffffffu);
}

// exact scores of the candidates: 8 candidates per block, thread t = token t.  The same operands as the matrix-core kernel
// (scaled fp16 planes, three cross terms) accumulated with fp32 FMAs.
__global__ void __launch_bounds__(kT) k_sel_rescore(const char* __restrict__ qp, const float* __restrict__ qinv, const char* __restrict__ kp,
                                                    const float* __restrict__ kinv, const int* __restrict__ n_tok, int b0,
                                                    const float* __restrict__ ctok, const float* __restrict__ gsum, const int64_t* __restrict__ cand,
                                                    const int* __restrict__ total, int cmax, int compact, float* __restrict__ cscore) {
  __shared__ __attribute__((aligned(16))) char krow[8][kRowF];
  __shared__ float kscale[8];
  __shared__ float red[8][4];
  const int bl = blockIdx.y, b = b0 + bl, t = threadIdx.x;
  const int n = total[bl] > cmax ? 0 : max(total[bl], 0);      // refused images (flags, overflow of the candidate list) cost nothing
  const int c0 = blockIdx.x * 8;
  if (c0 >= n) {
    if (t < 8 && c0 + t < cmax) cscore[(int64_t)bl * cmax + c0 + t] = -INFINITY;
    return;
  }
  const int nc = min(8, n - c0);
  for (int i = t; i < 8 * (kRowF / 16); i += kT) {
    const int c = i / (kRowF / 16), o = i - c * (kRowF / 16);
    // compact: kp holds the planes of this image's candidates in candidate order ([nb][cmax] rows, own 128-row tile scales)
    const int64_t r = compact ? (int64_t)bl * cmax + c0 + min(c, nc - 1) : cand[(int64_t)bl * cmax + c0 + min(c, nc - 1)];
    reinterpret_cast<float4*>(krow[c])[o] = reinterpret_cast<const float4*>(kp + r * kRowF)[o];
    if (o == 0) kscale[c] = kinv[r >> 7];
  }
  __syncthreads();
  const int M = n_tok[b];
  const char* qrow = qp + ((int64_t)b * kT + min(t, max(M - 1, 0))) * kRowF;
  // fp64 accumulation: every fp16 x fp16 product is exact and 1152 of them sum without rounding that matters, so a candidate's
  // logit is the exactly rounded contraction of the plane values (the vector fp64 rate makes these 8 x 1152 FMAs per thread free)
  double acc[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
  for (int sl = 0; sl < 12; ++sl) {
    f16x8 qh[4], ql[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      qh[j] = *reinterpret_cast<const f16x8*>(qrow + sl * kSlabF + j * 16);
      ql[j] = *reinterpret_cast<const f16x8*>(qrow + sl * kSlabF + 64 + j * 16);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      double a = acc[c];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f16x8 kh = *reinterpret_cast<const f16x8*>(krow[c] + sl * kSlabF + j * 16);
        const f16x8 kl = *reinterpret_cast<const f16x8*>(krow[c] + sl * kSlabF + 64 + j * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const double h = (double)(float)kh[e], l = (double)(float)kl[e], qhh = (double)(float)qh[j][e], qll = (double)(float)ql[j][e];
          a = __builtin_fma(h, qll, a);
          a = __builtin_fma(l, qhh, a);
          a = __builtin_fma(h, qhh, a);
        }
      }
      acc[c] = a;
    }
  }
  const float cq = qinv[2 * b + (t >> 7)], ct = t < M ? ctok[(int64_t)bl * kT + t] : -INFINITY;
  const float gt = gsum[(int64_t)bl * kT + t], rgt = (t < M && gt > 0.f) ? 1.f / gt : 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float cfl = ((cq * kscale[c]) * kInvSqrtD) * kLog2e;
    float e = t < M ? __builtin_amdgcn_exp2f(__builtin_fmaf((float)acc[c], cfl, ct)) * rgt : 0.f;
    e = sdg_wave_sum(e);
    if (sdg_lane() == 0) red[c][sdg_wave()] = e;
  }
  __syncthreads();
  if (t < 8 && c0 + t < cmax)
    cscore[(int64_t)bl * cmax + c0 + t] = t < nc ? (red[t][0] + red[t][1]) + (red[t][2] + red[t][3]) : -INFINITY;
}

// result: top-k of the exact candidate scores mapped back to ray indices; status[b] = candidates examined, or -1 when the
// select path cannot answer for this image (bounds unusable, more than cmax candidates, fewer than k) -- the caller falls back.
__global__ void __launch_bounds__(1024) k_sel_emit(const int64_t* __restrict__ lidx, const float* __restrict__ lval, const int64_t* __restrict__ cand,
                                                   const int* __restrict__ count, int cmax, int topk, int k_eff, int allow_fewer,
                                                   int64_t* __restrict__ idx, float* __restrict__ val, int* __restrict__ status) {
  const int bl = blockIdx.x, j = threadIdx.x;
  const int n = count[bl];
  int st = n;
  if (n == -2) st = 0;
  else if (n < 0 || n > cmax || (n < k_eff && !allow_fewer)) st = -1;      // a shard of a ray-sharded scene may hold fewer than k candidates
  if (allow_fewer && n >= 0 && n < k_eff) k_eff = n;
  if (j == 0) status[bl] = st;
  if (j >= topk) return;
  if (n == -2) {                           // no tokens: every score is exactly 0 and the k lowest indices win
    idx[(int64_t)bl * topk + j] = j < k_eff ? j : -1;
    val[(int64_t)bl * topk + j] = j < k_eff ? 0.f : NAN;
  } else if (st < 0 || j >= k_eff) {
    idx[(int64_t)bl * topk + j] = -1;
    val[(int64_t)bl * topk + j] = NAN;
  } else {
    idx[(int64_t)bl * topk + j] = cand[(int64_t)bl * cmax + lidx[(int64_t)bl * topk + j]];
    val[(int64_t)bl * topk + j] = lval[(int64_t)bl * topk + j];
  }
}

// ---- short lists: the whole selection in ONE workgroup per image (round 4).  The k-th largest TILE maximum of U (r / 256 values: 75 k at 19 M rays,
// 125 k at 32 M) and the top-100 of the <= 4096 re-scored candidates went through the same nine launches as a 32 M-element list (memset, 4 histogram
// passes, count, scan, write, sort: ~215 us per selection at one image per step, where nothing overlaps them -- 0.43 ms of a 13.5 ms step).  Here: radix
// select with the histogram in LDS (4 coalesced sweeps over the values, from L2), ONE ordered gather (a single workgroup walks the list in index order,
// so the ranks are running counts: no count / scan / write split), bitonic sort.  Same definition as the multi-kernel path -- the k largest by (value
// descending, index ascending) -- and the same bits (test_topk_small_lists_equal_the_multi_kernel_path).
constexpr int64_t kTopkSmallMax = 1 << 17;
__global__ void __launch_bounds__(1024) k_topk_small(const float* __restrict__ scores, int64_t stride, int n, int topk, int k_eff,
                                                      int64_t* __restrict__ idx, float* __restrict__ val, const int* __restrict__ only) {
  __shared__ unsigned hist[256];
  __shared__ unsigned sel[3];              // prefix of the k-th largest key decided so far, how many are still needed among the keys matching it, how many match it
  __shared__ unsigned slot;                // next free position of the unordered gather
  __shared__ unsigned sk[1024];
  __shared__ long long si[1024];
  __shared__ int wc[2][16];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (only != nullptr && only[b] == 0) return;
  const float* s = scores + (int64_t)b * stride;
  sk[t] = 0u;
  si[t] = 0x7fffffffffffffffll;
  unsigned prefix = 0u, remain = (unsigned)k_eff;
  if (k_eff > 0) {
    for (int p = 0; p < 4; ++p) {
      if (t < 256) hist[t] = 0u;
      __syncthreads();
      const unsigned hi_mask = p == 0 ? 0u : (0xffffffffu << (32 - 8 * p));
      const int shift = 24 - 8 * p;
      // The first passes see (nearly) ONE digit for every element -- scores of one image share their sign and leading exponent bits -- and 64 lanes
      // adding to one LDS word serialise (the first build spent 160 us per call on exactly that, profiles/r04_bench_headline_kernel_trace.md): the lanes
      // that share the first active lane's digit are counted with a ballot and added once; the others (later passes: few, scattered) add for themselves.
      // (eight values per thread requested before the first is looked at: with one load per iteration the workgroup waited ~0.7 us for each of 122 -- 85 us per pass)
      for (int c0 = 0; c0 < n; c0 += 8 * 1024) {
        float v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = c0 + u * 1024 + t;
          v8[u] = i < n ? s[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = c0 + u * 1024 + t;
          if (c0 + u * 1024 >= n) break;      // (uniform)
          unsigned key = 0u;
          bool act = false;
          if (i < n) {
            key = score_key(v8[u]);
            act = (key & hi_mask) == prefix;
          }
          const unsigned bin = (key >> shift) & 255u;
          const unsigned long long m = __ballot(act);
          if (m != 0ull) {
            const int leader = __ffsll((long long)m) - 1;
            const unsigned lb = (unsigned)__shfl((int)bin, leader);
            const unsigned long long same = __ballot(act && bin == lb);
            if (lane == leader) atomicAdd(&hist[lb], (unsigned)__popcll(same));
            else if (act && bin != lb) atomicAdd(&hist[bin], 1u);
          }
        }
      }
      __syncthreads();
      if (t < 64) {      // the digit of the k-th largest: lane l owns bins 255 - 4 l .. 252 - 4 l (descending), a wave scan finds the lane that crosses `remain`
        unsigned c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = hist[255 - (4 * t + j)];
        const unsigned loc = (c[0] + c[1]) + (c[2] + c[3]);
        const unsigned inc = sdg_wave_inclusive_scan(loc), exc = inc - loc;
        if (inc >= remain && exc < remain) {      // exactly one lane: the matching keys number at least `remain`
          unsigned acc = exc, digit = 0u, rem = 0u;
          bool found = false;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (!found && acc + c[j] >= remain) { digit = (unsigned)(255 - (4 * t + j)); rem = remain - acc; found = true; sel[2] = c[j]; }
            acc += c[j];
          }
          sel[0] = prefix | (digit << shift);
          sel[1] = rem;
        }
      }
      __syncthreads();
      prefix = sel[0];
      remain = sel[1];
    }
    const unsigned thr = prefix, n_greater = (unsigned)k_eff - remain, n_equal = sel[2];      // (after the last pass the bin count is the number of keys EQUAL to the threshold)
    if (n_greater + n_equal <= 1024u) {
      // Everything at or above the threshold fits the sort buffer (the usual case: k keys above a threshold that few share): gather it in ANY order -- one
      // sweep, a wave-aggregated atomic for the positions, no barrier -- and let the sort below put it into (key descending, index ascending) order; the first
      // k_eff entries are then the same as the ordered gather's (which pays two workgroup barriers per 1024 values for an order the sort establishes anyway).
      if (t == 0) slot = 0u;
      __syncthreads();
      const unsigned long long below = (1ull << lane) - 1ull;
      for (int c0 = 0; c0 < n; c0 += 8 * 1024) {
        float v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = c0 + u * 1024 + t;
          v8[u] = i < n ? s[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = c0 + u * 1024 + t;
          if (c0 + u * 1024 >= n) break;      // (uniform)
          unsigned key = 0u;
          bool hit = false;
          if (i < n) {
            key = score_key(v8[u]);
            hit = key >= thr;
          }
          const unsigned long long m = __ballot(hit);
          if (m != 0ull) {
            const int leader = __ffsll((long long)m) - 1;
            unsigned base = 0u;
            if (lane == leader) base = atomicAdd(&slot, (unsigned)__popcll(m));
            base = (unsigned)__shfl((int)base, leader);
            if (hit) {
              const unsigned pos = base + (unsigned)__popcll(m & below);
              if (pos < 1024u) { sk[pos] = key; si[pos] = (long long)i; }
            }
          }
        }
      }
    } else {
    // ordered gather: keys above the threshold at their running rank, keys equal to it (in index order) behind them while `remain` lasts
    unsigned run_g = 0u, run_e = 0u;
    for (int c0 = 0; c0 < n; c0 += 1024) {
      const int i = c0 + t;
      unsigned key = 0u;
      bool gt = false, eq = false;
      if (i < n) {
        key = score_key(s[i]);
        gt = key > thr;
        eq = key == thr;
      }
      const unsigned long long bg = __ballot(gt), be = __ballot(eq);
      if (lane == 0) { wc[0][wv] = __popcll(bg); wc[1][wv] = __popcll(be); }
      __syncthreads();
      unsigned og = 0u, oe = 0u, tg = 0u, te = 0u;
#pragma unroll
      for (int w2 = 0; w2 < 16; ++w2) {
        const unsigned g = (unsigned)wc[0][w2], e = (unsigned)wc[1][w2];
        if (w2 < wv) { og += g; oe += e; }
        tg += g;
        te += e;
      }
      __syncthreads();
      const unsigned long long below = (1ull << lane) - 1ull;
      if (gt) {
        const unsigned pos = run_g + og + (unsigned)__popcll(bg & below);
        if (pos < 1024u) { sk[pos] = key; si[pos] = (long long)i; }
      } else if (eq) {
        const unsigned re = run_e + oe + (unsigned)__popcll(be & below);
        if (re < remain && n_greater + re < 1024u) { sk[n_greater + re] = key; si[n_greater + re] = (long long)i; }
      }
      run_g += tg;
      run_e += te;
      if (run_e >= remain && run_g >= n_greater) break;      // (uniform: everything wanted has been seen)
    }
    }
  }
  __syncthreads();
  for (int size = 2; size <= 1024; size <<= 1)
    for (int strd = size >> 1; strd > 0; strd >>= 1) {
      const int p = t ^ strd;
      if (p > t) {
        const bool up = (t & size) == 0;
        const unsigned ka = sk[t], kb = sk[p];
        const long long ia = si[t], ib = si[p];
        const bool a_first = ka > kb || (ka == kb && ia < ib);
        if (up ? !a_first : a_first) { sk[t] = kb; sk[p] = ka; si[t] = ib; si[p] = ia; }
      }
      __syncthreads();
    }
  if (t < topk) {
    if (t < k_eff) { idx[(int64_t)b * topk + t] = si[t]; val[(int64_t)b * topk + t] = key_score(sk[t]); }
    else { idx[(int64_t)b * topk + t] = -1; val[(int64_t)b * topk + t] = NAN; }
  }
}

struct TopkPlan {
  int nb;          // gather blocks per image
  int64_t span;    // indices per gather block
  size_t off_hist, off_counts, off_offs, off_ckey, off_cidx, bytes;
};
TopkPlan topk_plan(int64_t r, int batch, int topk) {
  TopkPlan p;
  int64_t nb = sdg_cdiv(r, 256 * 16);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  p.nb = (int)nb;
  p.span = sdg_cdiv(sdg_cdiv(r, nb), 256) * 256;
  if (p.span < 256) p.span = 256;
  size_t o = 0;
  p.off_hist = o;   o += sdg_align((size_t)batch * 4 * 256 * sizeof(unsigned));
  p.off_counts = o; o += sdg_align((size_t)batch * nb * 2 * sizeof(unsigned));
  p.off_offs = o;   o += sdg_align((size_t)batch * nb * 2 * sizeof(unsigned));
  p.off_ckey = o;   o += sdg_align((size_t)batch * topk * sizeof(unsigned));
  p.off_cidx = o;   o += sdg_align((size_t)batch * topk * sizeof(int64_t));
  p.bytes = o;
  return p;
}

// only (device, [batch], or null = all): images with only[b] == 0 are skipped -- their idx / val rows keep what they held
int run_topk(const float* scores, int64_t stride, int64_t r, int batch, int topk, int64_t* idx, float* val, char* ws,
             hipStream_t s, const int* only = nullptr) {
  const TopkPlan p = topk_plan(r, batch, topk);
  unsigned* hist = (unsigned*)(ws + p.off_hist);
  unsigned* counts = (unsigned*)(ws + p.off_counts);
  unsigned* offs = (unsigned*)(ws + p.off_offs);
  unsigned* ckey = (unsigned*)(ws + p.off_ckey);
  int64_t* cidx = (int64_t*)(ws + p.off_cidx);
  const int k_eff = (int)(r < topk ? r : topk);
  static const bool small_ok = getenv("SIXDGS_TOPK_SMALL") == nullptr || atoi(getenv("SIXDGS_TOPK_SMALL")) != 0;      // (=0: the multi-kernel path for every size; test hook)
  if (small_ok && r <= kTopkSmallMax && topk <= 1024) {
    hipLaunchKernelGGL(k_topk_small, dim3((unsigned)batch), dim3(1024), 0, s, scores, stride, (int)r, topk, k_eff, idx, val, only);
    SDG_LAUNCH_OK();
    return 0;
  }
  if (k_eff > 0) {
    hipError_t e = hipMemsetAsync(hist, 0, (size_t)batch * 4 * 256 * sizeof(unsigned), s);
    if (e != hipSuccess) return (int)e;
    int64_t hb = sdg_cdiv(r, 256 * 8);
    if (hb > 1024) hb = 1024;
    const dim3 hg((unsigned)hb, (unsigned)batch);
    hipLaunchKernelGGL(k_topk_hist<0>, hg, dim3(256), 0, s, scores, stride, r, (unsigned)k_eff, hist, only);
    hipLaunchKernelGGL(k_topk_hist<1>, hg, dim3(256), 0, s, scores, stride, r, (unsigned)k_eff, hist, only);
    hipLaunchKernelGGL(k_topk_hist<2>, hg, dim3(256), 0, s, scores, stride, r, (unsigned)k_eff, hist, only);
    hipLaunchKernelGGL(k_topk_hist<3>, hg, dim3(256), 0, s, scores, stride, r, (unsigned)k_eff, hist, only);
    const dim3 gg((unsigned)p.nb, (unsigned)batch);
    hipLaunchKernelGGL(k_topk_gather<false>, gg, dim3(256), 0, s, scores, stride, r, (unsigned)k_eff, hist, p.span, counts, offs,
                       ckey, cidx, topk, only);
    hipLaunchKernelGGL(k_topk_scan, dim3((unsigned)batch), dim3(1024), 0, s, counts, p.nb, offs, only);
    hipLaunchKernelGGL(k_topk_gather<true>, gg, dim3(256), 0, s, scores, stride, r, (unsigned)k_eff, hist, p.span, counts, offs,
                       ckey, cidx, topk, only);
  }
  hipLaunchKernelGGL(k_topk_sort, dim3((unsigned)batch), dim3(1024), 0, s, ckey, cidx, topk, k_eff, idx, val, only);
  SDG_LAUNCH_OK();
  return 0;
}

struct ScorePlan {
  int64_t ldl;
  int n_tiles, tiles_per_group, n_groups;
  size_t per_image_logits, per_image_partial, per_image_stats, per_image_scores, per_image_qplanes, topk_bytes;
  size_t per_image() const { return per_image_logits + per_image_partial + per_image_stats + per_image_scores + per_image_qplanes; }
};
ScorePlan score_plan(int64_t r, int batch, int topk, bool logits24 = false) {
  ScorePlan p;
  p.ldl = sdg_cdiv(r > 0 ? r : 1, 256) * 256;   // whole 256-ray tiles (k_logits_f16x)
  p.n_tiles = (int)sdg_cdiv(r > 0 ? r : 1, kBN);
  p.tiles_per_group = (int)sdg_cdiv(p.n_tiles, 2048);
  p.n_groups = (int)sdg_cdiv(p.n_tiles, p.tiles_per_group);
  p.per_image_logits = logits24 ? sdg_align((size_t)(p.ldl / 128) * kTileBytes24, 1024)     // 24-bit tiles + references
                                : sdg_align((size_t)kT * p.ldl * sizeof(float));
  p.per_image_partial = sdg_align((size_t)p.n_groups * kT * 2 * sizeof(float));
  p.per_image_stats = sdg_align((size_t)kT * 2 * sizeof(float));
  p.per_image_scores = sdg_align((size_t)p.ldl * sizeof(float));
  p.per_image_qplanes = sdg_align((size_t)kT * kRowBytes, 1024) + 256;   // + absmax / scale / epilogue constant of the image
  p.topk_bytes = topk_plan(r, batch, topk).bytes;
  return p;
}

}  // namespace

extern "C" {

size_t sixdgs_topk_workspace_bytes(int64_t r, int batch, int topk) { return topk_plan(r, batch < 1 ? 1 : batch, topk).bytes; }

int sixdgs_topk(const float* scores, int64_t r, int batch, int topk, int64_t* idx, float* val, void* ws, size_t ws_bytes,
                sixdgs_stream_t stream) {
  SDG_CHECK_ARG(r >= 0 && batch >= 0 && topk >= 1 && topk <= 1024);
  if (batch == 0) return 0;
  SDG_CHECK_ARG((scores || r == 0) && idx && val && ws);
  if (ws_bytes < topk_plan(r, batch, topk).bytes) return SIXDGS_E_WORKSPACE;
  return run_topk(scores, r, r, batch, topk, idx, val, (char*)ws, sdg_stream(stream));
}

size_t sixdgs_score_topk_workspace_bytes(int64_t r, int batch, int topk) {
  if (batch < 1) batch = 1;
  const ScorePlan p = score_plan(r, batch, topk);
  return p.topk_bytes + (size_t)batch * p.per_image();
}

size_t sixdgs_score_topk_workspace_bytes_ex(int64_t r, int batch, int topk, int mma_mode, int with_key_planes) {
  if (batch < 1) batch = 1;
  const bool l24 = with_key_planes && (mma_mode == SIXDGS_MMA_F16X3 || mma_mode == SIXDGS_MMA_DEFAULT);
  const ScorePlan p = score_plan(r, batch, topk, l24);
  return p.topk_bytes + (size_t)batch * p.per_image();
}


size_t sixdgs_key_planes_f16_bytes(int64_t r) { return (size_t)(r > 0 ? r : 0) * kRowF; }

int sixdgs_split_planes_f16(const float* src, int64_t rows, int64_t ld, void* planes, float* d_inv_scale, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(rows >= 0 && ld >= SIXDGS_D && (ld % 4) == 0);
  if (rows == 0) return 0;
  SDG_CHECK_ARG(src && planes && d_inv_scale && ((uintptr_t)src % 16) == 0 && ((uintptr_t)planes % 16) == 0);
  hipLaunchKernelGGL(k_split_tiles_f16, dim3((unsigned)sdg_cdiv(rows, 128)), dim3(256), 0, sdg_stream(stream), src, rows, ld,
                     (char*)planes, d_inv_scale);
  SDG_LAUNCH_OK();
  return 0;
}


}  // extern "C"

namespace {
// phase 0: the whole scorer.  Phases 1 / 2 split it at the row statistics for ray-sharded scoring: phase 1 leaves the
// logits of ALL `batch` images in the workspace and returns the local statistics in row_stats; phase 2 takes the global
// statistics from row_stats and finishes (column sums, top-k).  `planes`: the logits are in the blocked fp16x3 layout.
int score_impl(int phase, bool planes, const float* q, const int32_t* d_n_tok, const int32_t* h_n_tok, int batch, const float* key,
               const void* key_planes, const float* d_key_scale, int64_t r, int topk, float* scores, int64_t* idx, float* val,
               float* row_stats, void* ws, size_t ws_bytes, sixdgs_stream_t stream, sixdgs_profile* prof, int mma_mode) {
  SDG_CHECK_ARG(r >= 0 && batch >= 0 && topk >= 1 && topk <= 1024);
  if (batch == 0) return 0;
  const bool f16_mode = mma_mode == SIXDGS_MMA_F16X3 || mma_mode == SIXDGS_MMA_F16X3_L32 || mma_mode == SIXDGS_MMA_DEFAULT;
  const bool use_f16 = (phase == 2 ? planes : key_planes != nullptr) && f16_mode;
  SDG_CHECK_ARG(phase == 2 || !use_f16 || d_key_scale != nullptr);
  // key planes are an operand of the fp16 x 3 kernels only.  (Round 1's bf16 x 6 plane scorer was removed in round 6: nothing outside its own tests
  // launched it.  SIXDGS_MMA_BF16X6 / _F32 score on fp32 keys -- k_logits<MMA> below -- and need `key`.)
  if ((phase == 2 ? planes : key_planes != nullptr) && !f16_mode && !(phase != 2 && key)) return SIXDGS_E_UNSUPPORTED;
  const bool logits24 = mma_mode != SIXDGS_MMA_F16X3_L32;         // 24-bit fixed-point logits between the passes (fp16x3 path)
  SDG_CHECK_ARG(d_n_tok && ws && (phase == 2 || (q && (key || use_f16 || r == 0))) && (phase == 1 || (idx && val)) &&
                (phase == 0 || row_stats));
  SDG_CHECK_ARG(((uintptr_t)key_planes % 16) == 0);
  SDG_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)key % 16) == 0 && ((uintptr_t)ws % 256) == 0);
  hipStream_t s = sdg_stream(stream);
  ScorePlan p = score_plan(r, batch, topk, use_f16 && logits24);
  const size_t per_image = p.per_image();
  // largest image group whose logits + top-k scratch fit the caller's workspace
  int64_t bg = batch > 65535 ? 65535 : batch;
  while (bg >= 1 && topk_plan(r, (int)bg, topk).bytes + (size_t)bg * per_image > ws_bytes) --bg;
  if (bg < 1 || (phase != 0 && bg < batch)) return SIXDGS_E_WORKSPACE;   // the split phases keep every image resident
  bg = sdg_cdiv(batch, sdg_cdiv(batch, bg));      // equal groups (6 images, room for 5: 3 + 3, not 5 + 1 -- the images of a group share the key stream)
  p.topk_bytes = topk_plan(r, (int)bg, topk).bytes;
  char* base = (char*)ws;
  char* topk_ws = base;
  float* logits = (float*)(base + p.topk_bytes);
  float* partial = (float*)((char*)logits + (size_t)bg * p.per_image_logits);
  float* stats = (float*)((char*)partial + (size_t)bg * p.per_image_partial);
  float* sc_ws = (float*)((char*)stats + (size_t)bg * p.per_image_stats);
  char* qplanes = (char*)sc_ws + (size_t)bg * p.per_image_scores;
  const int64_t ldl_img = (int64_t)(p.per_image_logits / sizeof(float)) / kT;  // == p.ldl (alignment keeps it)
  (void)ldl_img;
  for (int b0 = 0; b0 < batch; b0 += (int)bg) {
    const int nb = (int)((batch - b0) < bg ? (batch - b0) : bg);
    float* sc = scores ? scores + (int64_t)b0 * r : sc_ws;
    const int64_t sc_stride = scores ? r : (int64_t)(p.per_image_scores / sizeof(float));
    if (r > 0) {
      int n_groups_used = p.n_groups;
      LogitsArgs A = {q, d_n_tok, key, logits, partial, r, (int64_t)(p.per_image_logits / sizeof(float) / kT),
                      p.tiles_per_group, p.n_tiles, p.n_groups, b0};
      if (phase != 2) {
        double tok = 0.0;  // algorithmic work of this launch: 2*T*d FLOP and d*4 (key) + T*4 (logit) bytes per ray and image
        for (int i = 0; i < nb; ++i) tok += h_n_tok ? (double)h_n_tok[b0 + i] : (double)kT;
        // operand bytes per ray: 1536 B fp32 key or 2304 B of bf16 planes per image; the fp16x3 kernel streams its 1536 B of
        // fp16 planes once per LAUNCH (the images of a launch share every key tile through L2) and writes 3 (+1/16 for the
        // references) or 4 bytes per logit
        SdgProfileScope scope(prof, s, 2.0 * tok * SIXDGS_D * (double)r,
                              use_f16 ? (double)r * (kRowF + tok * (logits24 ? 3.0 + 8.0 / 128.0 : 4.0))
                                      : (double)r * (nb * (SIXDGS_D * 4.0) + tok * 4.0));
        if (use_f16) {
          // scaled fp16 planes of q (one power-of-two scale per 128-token half), then the fp16x3 kernel
          float* qinv = (float*)(qplanes + (size_t)bg * (p.per_image_qplanes - 256));
          hipLaunchKernelGGL(k_split_tiles_f16, dim3((unsigned)(2 * nb)), dim3(256), 0, s, q + (int64_t)b0 * kT * SIXDGS_D,
                             (int64_t)nb * kT, (int64_t)SIXDGS_D, qplanes, qinv);
          LogitsF16Args V = {qplanes - (int64_t)b0 * kT * kRowF, d_n_tok, (const char*)key_planes, qinv - 2 * b0, d_key_scale, logits,
                             partial, r, A.ldl, p.tiles_per_group, p.n_tiles, p.n_groups, b0, nb};
          // 256-ray tiles in groups: the group count is a multiple of the CU count (equal-length runs, no partial last
          // round for any number of images) and does NOT depend on the batch, so an image scores to the same bits alone
          // or inside a batch; runs of at least 16 tiles when the scene is large enough
          const int n_tiles_x = (p.n_tiles + 1) / 2;
          int n_groups_x = 1024;
          while (n_groups_x > 256 && n_tiles_x / n_groups_x < 16) n_groups_x -= 256;
          if (n_groups_x > p.n_groups) n_groups_x = p.n_groups;
          if (n_groups_x > n_tiles_x) n_groups_x = n_tiles_x;
          n_groups_used = n_groups_x;             // exactly this many runs (a whole number of rounds of the 256 CUs), lengths differ by <= 1
          V.tiles_per_group = (int)sdg_cdiv(n_tiles_x, n_groups_x);
          V.n_tiles = n_tiles_x;
          V.n_groups = n_groups_used;
          auto kern = logits24 ? k_logits_f16x<0, kOutL24> : k_logits_f16x<0, kOutF32>;
#ifdef SIXDGS_ABLATION   // timing experiments only (tools/ablate_logits.py builds a private copy of the library with it)
          if (const char* ab = getenv("SIXDGS_DEBUG_ABLATE")) {
#define SDG_ABL_CASE(n) case n: kern = logits24 ? k_logits_f16x<n, kOutL24> : k_logits_f16x<n, kOutF32>; break;
            switch (atoi(ab)) {
              SDG_ABL_CASE(1) SDG_ABL_CASE(8) SDG_ABL_CASE(9) SDG_ABL_CASE(2) SDG_ABL_CASE(11) SDG_ABL_CASE(27) SDG_ABL_CASE(59)
              SDG_ABL_CASE(64) SDG_ABL_CASE(2048)
              default: break;
            }
#undef SDG_ABL_CASE
          }
#endif
          hipLaunchKernelGGL(kern, dim3((unsigned)(n_groups_used * nb)), dim3(512), 0, s, V);
        } else if (mma_mode == SIXDGS_MMA_F32) {
          hipLaunchKernelGGL(k_logits<kMmaF32>, dim3((unsigned)(p.n_groups * 2), (unsigned)nb), dim3(256), 0, s, A);
        } else {
          hipLaunchKernelGGL(k_logits<kMmaBf16x6>, dim3((unsigned)(p.n_groups * 2), (unsigned)nb), dim3(256), 0, s, A);
        }
      }
      if (phase != 2) hipLaunchKernelGGL(k_merge_stats, dim3((unsigned)nb, 4), dim3(1024), 0, s, partial, n_groups_used, stats);
      if (phase == 2) {        // the caller's global statistics replace the local ones
        hipError_t e = hipMemcpyAsync(stats, row_stats + (int64_t)b0 * kT * 2, (size_t)nb * kT * 2 * sizeof(float),
                                      hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return (int)e;
      }
      if (phase == 1) {
        hipError_t e = hipMemcpyAsync(row_stats + (int64_t)b0 * kT * 2, stats, (size_t)nb * kT * 2 * sizeof(float),
                                      hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return (int)e;
        SDG_LAUNCH_OK();
        continue;
      }
      if (use_f16 && logits24)
        hipLaunchKernelGGL(k_score_reduce_blocked24, dim3((unsigned)p.n_tiles, (unsigned)nb), dim3(256), 0, s, logits, A.ldl, stats,
                           d_n_tok, b0, r, sc, sc_stride);
      else if (use_f16)
        hipLaunchKernelGGL(k_score_reduce_blocked, dim3((unsigned)p.n_tiles, (unsigned)nb), dim3(256), 0, s, logits, A.ldl, stats, d_n_tok,
                           b0, r, sc, sc_stride);
      else
        hipLaunchKernelGGL(k_score_reduce, dim3((unsigned)sdg_cdiv(r, 256), (unsigned)nb), dim3(256), 0, s, logits, A.ldl, stats,
                           d_n_tok, b0, r, sc, sc_stride);
      SDG_LAUNCH_OK();
      if (row_stats && phase == 0) {
        hipError_t e = hipMemcpyAsync(row_stats + (int64_t)b0 * kT * 2, stats, (size_t)nb * kT * 2 * sizeof(float),
                                      hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return (int)e;
      }
    }
    if (phase == 1) continue;
    int st = run_topk(sc, sc_stride, r, nb, topk, idx + (int64_t)b0 * topk, val + (int64_t)b0 * topk, topk_ws, s);
    if (st) return st;
  }
  return 0;
}
}  // namespace

extern "C" {

int sixdgs_score_topk_ex(const float* q, const int32_t* d_n_tok, const int32_t* h_n_tok, int batch, const float* key,
                         const void* key_planes, const float* d_key_scale, int64_t r, int topk, float* scores, int64_t* idx,
                         float* val, float* row_stats, void* ws, size_t ws_bytes, sixdgs_stream_t stream, sixdgs_profile* prof,
                         int mma_mode) {
  return score_impl(0, key_planes != nullptr, q, d_n_tok, h_n_tok, batch, key, key_planes, d_key_scale, r, topk, scores, idx, val,
                    row_stats, ws, ws_bytes, stream, prof, mma_mode);
}

int sixdgs_score_pass1(const float* q, const int32_t* d_n_tok, const int32_t* h_n_tok, int batch, const float* key,
                       const void* key_planes, const float* d_key_scale, int64_t r, int topk, float* row_stats, void* ws,
                       size_t ws_bytes, sixdgs_stream_t stream, sixdgs_profile* prof, int mma_mode) {
  return score_impl(1, key_planes != nullptr, q, d_n_tok, h_n_tok, batch, key, key_planes, d_key_scale, r, topk, nullptr, nullptr,
                    nullptr, row_stats, ws, ws_bytes, stream, prof, mma_mode);
}

int sixdgs_score_pass2(const float* row_stats, const int32_t* d_n_tok, int batch, int used_planes, int64_t r, int topk, float* scores,
                       int64_t* idx, float* val, void* ws, size_t ws_bytes, sixdgs_stream_t stream, int mma_mode) {
  return score_impl(2, used_planes != 0, nullptr, d_n_tok, nullptr, batch, nullptr, nullptr, nullptr, r, topk, scores, idx, val,
                    const_cast<float*>(row_stats), ws, ws_bytes, stream, nullptr, mma_mode);
}

int sixdgs_score_topk(const float* q, const int32_t* d_n_tok, int batch, const float* key, int64_t r, int topk, float* scores,
                      int64_t* idx, float* val, float* row_stats, void* ws, size_t ws_bytes, sixdgs_stream_t stream) {
  return sixdgs_score_topk_ex(q, d_n_tok, nullptr, batch, key, nullptr, nullptr, r, topk, scores, idx, val, row_stats, ws, ws_bytes,
                              stream, nullptr, SIXDGS_MMA_DEFAULT);
}

#ifdef SIXDGS_ABLATION
int sixdgs_debug_cycles(unsigned long long* host64, int reset) {   // ablation builds only (tools/ablate_logits.py)
  hipError_t e = hipMemcpyFromSymbol(host64, HIP_SYMBOL(g_dbg_cycles), sizeof(unsigned long long) * 64);
  if (e == hipSuccess && reset) {
    unsigned long long z[64] = {0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_cycles), z, sizeof(z));
  }
  return (int)e;
}
#endif

int sixdgs_profile_collect(sixdgs_profile* prof, double* ms_total, double* flops_total, double* bytes_total, int* launches) {
  SDG_CHECK_ARG(prof);
  double ms = 0.0, fl = 0.0, by = 0.0;
  int n = 0, rc = 0;
  for (int i = 0; i < prof->count; ++i) {
    hipEvent_t a = (hipEvent_t)prof->start[i], b = (hipEvent_t)prof->stop[i];
    float t = 0.f;
    hipError_t e = hipEventSynchronize(b);
    if (e == hipSuccess) e = hipEventElapsedTime(&t, a, b);
    if (e == hipSuccess) { ms += t; fl += prof->flops[i]; by += prof->bytes[i]; ++n; } else rc = (int)e;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
  }
  prof->count = 0;
  if (ms_total) *ms_total = ms;
  if (flops_total) *flops_total = fl;
  if (bytes_total) *bytes_total = by;
  if (launches) *launches = n;
  return rc;
}

}  // extern "C"

namespace {
// ---- select path, host side ------------------------------------------------------------------------------------------------
// Four stages, each an entry point of its own so that the main sweep can run chunk by chunk over a scene whose key planes do not
// fit the GPU (begin once, sweep per chunk, candidates once, re-score on the keys of the candidates alone); sixdgs_score_select
// chains them for the resident case.  Every stage takes all `nb` images at once; its scratch comes from one workspace.
struct SelectPlan {
  int64_t stride;      // floats per image row of the ub partials (whole 256-ray tiles)
  int nbc;             // candidate-compaction blocks per image
  int64_t span;
  size_t o_partial, o_stats, o_qpl, o_ub, o_idxu, o_valu, o_lidx, o_lval, o_info, o_counts, o_offs, o_total, o_cscore, o_slot, per_image, topk_bytes;
};
SelectPlan select_plan(int64_t r, int batch, int topk, int cmax, bool with_ub = true) {
  SelectPlan p;
  p.stride = sdg_cdiv(r > 0 ? r : 1, 256) * 256;
  const TopkPlan tp = topk_plan(r, batch, topk);
  p.nbc = tp.nb;
  p.span = tp.span;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += sdg_align(bytes); return at; };
  p.o_partial = take((size_t)1024 * kT * 2 * sizeof(float));
  p.o_stats = take((size_t)kT * 2 * sizeof(float));
  p.o_qpl = take(sdg_align((size_t)kT * kRowF, 1024) + 256);
  p.o_ub = take(with_ub ? (size_t)4 * p.stride * sizeof(float) : 16);
  p.o_idxu = take((size_t)topk * sizeof(int64_t));
  p.o_valu = take((size_t)topk * sizeof(float));
  p.o_lidx = take((size_t)topk * sizeof(int64_t));
  p.o_lval = take((size_t)topk * sizeof(float));
  p.o_info = take(16);
  p.o_counts = take((size_t)p.nbc * sizeof(unsigned));
  p.o_offs = take((size_t)p.nbc * sizeof(unsigned));
  p.o_total = take(16);
  p.o_cscore = take((size_t)cmax * sizeof(float));
  p.o_slot = take((size_t)kT * sizeof(float) + 256);      // packed sweep: the slot's ctok row [256] + its row count
  p.per_image = o;
  const size_t t2 = topk_plan(cmax, batch, topk).bytes;
  p.topk_bytes = tp.bytes > t2 ? tp.bytes : t2;
  return p;
}
struct SelectWs {
  SelectPlan p;
  char *topk_ws, *qplanes;
  float *partial, *stats, *qinv, *ub, *valU, *lval, *info, *cscore, *ctok_slot;
  int* slot_rows;
  int64_t *idxU, *lidx;
  unsigned *counts, *offs;
  int* total;
  size_t qpl_img;
};
// fields are [nb][...]: field f of image bl lives at base + f_offset * nb + bl * size_f (contiguous per field)
bool select_ws(void* ws, size_t ws_bytes, int64_t r, int nb, int topk, int cmax, SelectWs* w, bool with_ub = true) {
  w->p = select_plan(r, nb, topk, cmax, with_ub);
  if (w->p.topk_bytes + (size_t)nb * w->p.per_image > ws_bytes) return false;
  w->topk_ws = (char*)ws;
  char* img0 = (char*)ws + w->p.topk_bytes;
  auto field = [&](size_t off) { return img0 + off * (size_t)nb; };
  w->partial = (float*)field(w->p.o_partial);
  w->stats = (float*)field(w->p.o_stats);
  w->qplanes = field(w->p.o_qpl);
  w->qpl_img = sdg_align((size_t)kT * kRowF, 1024);
  w->qinv = (float*)(w->qplanes + (size_t)nb * w->qpl_img);
  w->ub = (float*)field(w->p.o_ub);
  w->idxU = (int64_t*)field(w->p.o_idxu);
  w->valU = (float*)field(w->p.o_valu);
  w->lidx = (int64_t*)field(w->p.o_lidx);
  w->lval = (float*)field(w->p.o_lval);
  w->info = (float*)field(w->p.o_info);
  w->counts = (unsigned*)field(w->p.o_counts);
  w->offs = (unsigned*)field(w->p.o_offs);
  w->total = (int*)field(w->p.o_total);
  w->cscore = (float*)field(w->p.o_cscore);
  w->ctok_slot = (float*)field(w->p.o_slot);
  w->slot_rows = (int*)(field(w->p.o_slot) + (size_t)nb * kT * sizeof(float));
  return true;
}

// run length of the fp16x3 kernel: a multiple of the CU count of equal-length runs, independent of the batch (batch invariance)
int f16x_groups(int64_t r) {
  const int n_tiles_x = (int)sdg_cdiv(r > 0 ? r : 1, kBNX);
  int g = 1024;
  while (g > 256 && n_tiles_x / g < 16) g -= 256;
  if (g > n_tiles_x) g = n_tiles_x;
  return g;
}

// CUs to spread persistent sibling sets over, or 0 = one-shot grid (SIXDGS_SIBLING_SYNC=0/1 overrides the default)
int sibling_sync_mode() {      // 0 one-shot grid, 1 persistent sibling sets in lock-step, 2 persistent without the lock-step (measurement), 3 lock-step with a sibling that never arrives (test)
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("SIXDGS_SIBLING_SYNC");
    mode = e ? atoi(e) : kSiblingSyncDefault;
  }
  return mode;
}
int sibling_sync_cus() {
  static int cus = -1;
  if (cus < 0) {
    const bool on = sibling_sync_mode() != 0;
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (on && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 0;
  }
  return cus;
}

// scaled fp16 planes of q (one power-of-two scale per 128-token half) for images [0, nb)
void select_q_planes(const float* q, int nb, const SelectWs& w, hipStream_t s) {
  hipLaunchKernelGGL(k_split_tiles_f16, dim3((unsigned)(2 * nb)), dim3(256), 0, s, q, (int64_t)nb * kT, (int64_t)SIXDGS_D, w.qplanes, w.qinv);
}
LogitsF16Args select_args(const int32_t* d_n_tok, int nb, const SelectWs& w, const void* planes, const float* scale, int64_t r) {
  LogitsF16Args V = {w.qplanes, d_n_tok, (const char*)planes, w.qinv, scale, nullptr, w.partial, r, w.p.stride, 0, 0, 0, 0, nb, nullptr, nullptr, 0};
  V.n_tiles = (int)sdg_cdiv(r, kBNX);
  V.n_groups = f16x_groups(r);
  V.tiles_per_group = (int)sdg_cdiv(V.n_tiles, V.n_groups);
  return V;
}
}  // namespace

extern "C" {

size_t sixdgs_select_workspace_bytes(int64_t r, int batch, int topk, int max_candidates) {
  if (batch < 1) batch = 1;
  const SelectPlan p = select_plan(r, batch, topk, max_candidates);
  return p.topk_bytes + (size_t)batch * p.per_image;
}

size_t sixdgs_select_candidates_workspace_bytes(int64_t r, int batch, int topk, int max_candidates) {
  if (batch < 1) batch = 1;
  const SelectPlan p = select_plan(r, batch, topk, max_candidates, false);
  return p.topk_bytes + (size_t)batch * p.per_image;
}

// slots per launch of the packed sweep (SIXDGS_SWEEP_MAX_IMAGES; before round 5 it counted images -- a slot then held one image)
static int sweep_slot_cap() {
  static const int cap = [] {
    const char* e = getenv("SIXDGS_SWEEP_MAX_IMAGES");
    const int v = e ? atoi(e) : kSweepMaxImages;
    if (e && v <= 0) fprintf(stderr, "6dgs_amd: SIXDGS_SWEEP_MAX_IMAGES=%s: launches of %d slots (the slot table's limit), a last one of up to %d -- not one launch per batch\n", e, kSweepMaxSlots * 2 / 3, kSweepMaxSlots * 2 / 3 + kSweepMaxSlots / 3);
    return v;
  }();
  return cap;
}

// The sample pre-pass with ONE of the three MFMA terms (kAblOneTerm) -- for every launch, whatever its slot count: an image's statistics, candidates and
// values must not depend on what it was batched with (test_token_packing_is_invisible..., test_sweep_launch_grouping_is_invisible).
// SIXDGS_PREPASS_TERMS=3: all three (rounds 2-5).
static bool prepass_one_term(int /*n_slots*/) {
  static const int forced = [] { const char* e = getenv("SIXDGS_PREPASS_TERMS"); return e ? atoi(e) : 0; }();
  return forced == 1 || (forced != 3 && kPrepassOneTermDefault);
}

int sixdgs_select_sweep_plan(const int32_t* h_n_tok, int batch, int32_t* slots_per_launch, int32_t* images_per_launch, int max_launches) {
  if (batch < 0 || batch > 32767 || max_launches < 0 || (max_launches > 0 && (!slots_per_launch || !images_per_launch))) return SIXDGS_E_BADARG;
  const std::vector<SweepSlots> plan = sweep_pack(h_n_tok, batch, sweep_slot_cap());
  for (size_t l = 0; l < plan.size() && (int)l < max_launches; ++l) {
    slots_per_launch[l] = plan[l].n_slots;
    images_per_launch[l] = plan[l].n_images;
  }
  return (int)plan.size();
}

int sixdgs_select_sample_stats(const float* q, const int32_t* d_n_tok, const int32_t* h_n_tok, int batch, const void* sample_planes,
                               const float* d_sample_scale, int64_t r_sample, float* row_stats, void* ws, size_t ws_bytes, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(batch >= 0 && batch <= 32767 && r_sample >= 1);
  if (batch == 0) return 0;
  SDG_CHECK_ARG(q && d_n_tok && sample_planes && d_sample_scale && row_stats && ws && ((uintptr_t)sample_planes % 16) == 0 &&
                ((uintptr_t)q % 16) == 0 && ((uintptr_t)ws % 256) == 0);
  hipStream_t s = sdg_stream(stream);
  SelectWs w;
  if (!select_ws(ws, ws_bytes, r_sample, batch, 1, 8, &w)) return SIXDGS_E_WORKSPACE;
  const int n_groups = f16x_groups(r_sample);
  for (const SweepSlots& T : sweep_pack(h_n_tok, batch, sweep_slot_cap())) {      // token packing: sweep_plan.h
    if (T.n_slots > 0) {
      hipLaunchKernelGGL(k_split_q_slots, dim3((unsigned)(4 * T.n_slots)), dim3(256), 0, s, q, d_n_tok, T, w.qplanes, w.qinv, (const float*)nullptr,
                         (float*)nullptr, w.slot_rows);
      LogitsF16Args V = select_args(w.slot_rows, T.n_slots, w, sample_planes, d_sample_scale, r_sample);
      V.q_quarter_scales = 1;
      // (round 6) PERSISTENT on fewer workgroups than compute units: as a one-shot grid the pre-pass holds every CU for its ~1 ms, and the image side of
      // the NEXT batch -- ~90 small dependent launches on another stream, which must finish inside the same window between two sweeps -- stands still.
      // With `reserve` CUs left out (kPrepassReserveCus; SIXDGS_PREPASS_RESERVE_CUS overrides, -1 = the one-shot grid) the same groups are walked by
      // (CUs - reserve) / slots sets: same partial statistics per group, hence the same bits.  Measured (profiles/r06_pipeline_ab.md): headline step - sweep
      // 2.84 -> 2.61 ms with 64 of 256 left out (16: no gain, 128: less), cfg-2 11.58 -> 11.39 ms between poses, cfg-3 unchanged, one batch at a time +0.09 ms.
      static const int reserve_env = [] { const char* e = getenv("SIXDGS_PREPASS_RESERVE_CUS"); return e ? atoi(e) : -2; }();
      // one slot (a single query per step: cfg-2): the pre-pass is short and not matrix-bound, and the window it opens is what the NEXT query's image side (1 ms of
      // small dependent launches) lives on -- 160 CUs left out instead of 64 (profiles/r06_prepass_one_term.md: with the one-term pre-pass 11.24 / 11.25 / 11.20 /
      // 11.16 ms between poses at 64 / 96 / 128 / 160, three terms at 64: 11.14).  The reserve never changes a bit: the same groups, walked by fewer sets.
      const int reserve = reserve_env != -2 ? reserve_env : (T.n_slots == 1 ? kPrepassReserveCusOneSlot : kPrepassReserveCus);
      const int cus = sibling_sync_cus();
      if (reserve >= 0 && cus > reserve && (cus - reserve) / T.n_slots >= 1) {
        V.n_sets = (cus - reserve) / T.n_slots < V.n_groups ? (cus - reserve) / T.n_slots : V.n_groups;
        V.sib_sync = nullptr;
        V.sib_extra = 0u;
        V.sib_period = kSibPeriod;
        if (prepass_one_term(T.n_slots)) hipLaunchKernelGGL((k_logits_f16x<kAblOneTerm, kOutStats, true>), dim3((unsigned)(V.n_sets * T.n_slots)), dim3(512), 0, s, V);
        else hipLaunchKernelGGL((k_logits_f16x<0, kOutStats, true>), dim3((unsigned)(V.n_sets * T.n_slots)), dim3(512), 0, s, V);
      } else {
        if (prepass_one_term(T.n_slots)) hipLaunchKernelGGL((k_logits_f16x<kAblOneTerm, kOutStats>), dim3((unsigned)(V.n_groups * T.n_slots)), dim3(512), 0, s, V);
        else hipLaunchKernelGGL((k_logits_f16x<0, kOutStats>), dim3((unsigned)(V.n_groups * T.n_slots)), dim3(512), 0, s, V);
      }
    }
    hipLaunchKernelGGL(k_merge_stats_slots, dim3((unsigned)T.n_images, 4), dim3(1024), 0, s, w.partial, n_groups, T, row_stats, (float*)nullptr);
  }
  SDG_LAUNCH_OK();
  return 0;
}

int sixdgs_select_prepare(const float* row_stats, const int32_t* d_n_tok, int batch, int64_t r_sample, int64_t r_total, float* ctok, float* gsum,
                          sixdgs_stream_t stream) {
  SDG_CHECK_ARG(batch >= 0 && r_sample >= 1 && r_total >= r_sample);
  if (batch == 0) return 0;
  SDG_CHECK_ARG(row_stats && d_n_tok && ctok && gsum);
  hipStream_t s = sdg_stream(stream);
  hipLaunchKernelGGL(k_sel_prepare, dim3((unsigned)batch), dim3(kT), 0, s, row_stats, d_n_tok, 0, log2f((float)((double)r_total / (double)r_sample)), ctok);
  hipError_t e = hipMemsetAsync(gsum, 0, (size_t)batch * kT * sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  SDG_LAUNCH_OK();
  return 0;
}

int sixdgs_select_begin(const float* q, const int32_t* d_n_tok, const int32_t* h_n_tok, int batch, const void* sample_planes, const float* d_sample_scale,
                        int64_t r_sample, int64_t r_total, float* ctok, float* gsum, void* ws, size_t ws_bytes, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(batch >= 0 && r_sample >= 1 && r_total >= r_sample);
  if (batch == 0) return 0;
  SDG_CHECK_ARG(ws && ((uintptr_t)ws % 256) == 0);
  SelectWs w;
  if (!select_ws(ws, ws_bytes, r_sample, batch, 1, 8, &w)) return SIXDGS_E_WORKSPACE;
  const int st = sixdgs_select_sample_stats(q, d_n_tok, h_n_tok, batch, sample_planes, d_sample_scale, r_sample, w.stats, ws, ws_bytes, stream);
  if (st) return st;
  return sixdgs_select_prepare(w.stats, d_n_tok, batch, r_sample, r_total, ctok, gsum, stream);
}

int sixdgs_select_sweep(const float* q, const int32_t* d_n_tok, const int32_t* h_n_tok, int batch, const void* key_planes,
                        const float* d_key_scale, int64_t r, const float* ctok, float* gsum, float* u, int64_t u_stride, float* u_tile_max,
                        void* ws, size_t ws_bytes, sixdgs_stream_t stream, sixdgs_profile* prof) {
  SDG_CHECK_ARG(batch >= 0 && batch <= 32767 && r >= 1 && u_stride >= sdg_cdiv(r, 256) * 256 && (u_stride % 4) == 0 && (!u_tile_max || (u_stride % 256) == 0));
  if (batch == 0) return 0;
  SDG_CHECK_ARG(q && d_n_tok && key_planes && d_key_scale && ctok && gsum && u && ws && ((uintptr_t)key_planes % 16) == 0 &&
                ((uintptr_t)q % 16) == 0 && ((uintptr_t)u % 16) == 0 && ((uintptr_t)ws % 256) == 0);
  // Slots per launch are capped (round 4, then per image): every sibling streams its OWN q planes (393 KB per slot) once per tile, and the slots of a launch
  // share an XCD's 4 MB L2 with the key tiles in flight -- with many of them the q planes fall out of it.  Measured (tools/time_sweep.py B 8388608,
  // profiles/r04_sweep_images_per_launch.log; TFLOP/s fp32-equivalent): one launch for the whole batch 415 / 422 / 413 / 401 / 354 / 278 at 4 / 8 / 12 / 16 / 24 /
  // 32 images; launches of 8: 423 / 424 / 426 at 16 / 24 / 32.  So a batch goes in launches of kSweepMaxImages slots, the last one taking what is left up to
  // 1.5 x that (a launch of one or two slots would pull every key tile from HBM for itself: 0.455).  Every launch streams the key planes once.
  // Round 5: the images of a launch are PACKED into the slots by their token counts (sweep_plan.h) -- two views of <= 128 tokens or four of <= 64 share a
  // 256-row tile -- so the matrix work AND the key stream per image follow the tokens the mask kept, as the reference's cost does
  // (backbone.py:86-114, identification_module.py:80-82).  Results do not depend on grouping or packing (tests/test_gpu_select.py, test_gpu_full_size.py).
  hipStream_t s = sdg_stream(stream);
  SelectWs w;
  if (!select_ws(ws, ws_bytes, r, batch, 1, 8, &w)) return SIXDGS_E_WORKSPACE;
  const int n_groups = f16x_groups(r);
  for (const SweepSlots& T : sweep_pack(h_n_tok, batch, sweep_slot_cap())) {
    const int ns = T.n_slots;
    double tok = 0.0;
    for (int k = 0; k < T.n_images; ++k) tok += h_n_tok ? (double)(h_n_tok[T.img[k]] < 0 ? 0 : (h_n_tok[T.img[k]] > kT ? kT : h_n_tok[T.img[k]])) : (double)kT;
    if (ns > 0) {
      hipLaunchKernelGGL(k_split_q_slots, dim3((unsigned)(4 * ns)), dim3(256), 0, s, q, d_n_tok, T, w.qplanes, w.qinv, ctok, w.ctok_slot, w.slot_rows);
      LogitsF16Args V = select_args(w.slot_rows, ns, w, key_planes, d_key_scale, r);
      V.q_quarter_scales = 1;
      V.ctok = w.ctok_slot;
      V.ub = w.ub;
      V.ub_stride = w.p.stride;
      // algorithmic work: 2*T*384 FLOP per ray and image; bytes: the key planes once per launch + 16 B of partial sums per ray and image
      unsigned grid = (unsigned)(V.n_groups * ns);
      const int cus = sibling_sync_cus();
      // (persistent sets leave cus % ns CUs idle: only when that is at most 1/16 of the chip -- e.g. not for 100 slots per launch)
      if (cus > 0 && ns >= 2 && cus / ns >= 1 && (cus % ns) * 16 <= cus && w.p.topk_bytes >= (size_t)(cus / ns) * sizeof(unsigned)) {
        // persistent sibling sets in lock-step (see the kernel): at most one workgroup per CU, so that every sibling is resident
        V.n_sets = cus / ns < V.n_groups ? cus / ns : V.n_groups;
        V.sib_sync = sibling_sync_mode() == 2 ? nullptr : reinterpret_cast<unsigned*>(w.topk_ws);      // the top-k scratch is idle during the sweep
        V.sib_extra = sibling_sync_mode() == 3 ? 1u : 0u;
        V.sib_period = kSibPeriod;
        if (V.sib_sync && hipMemsetAsync(V.sib_sync, 0, (size_t)V.n_sets * sizeof(unsigned), s) != hipSuccess) return (int)hipGetLastError();
        grid = (unsigned)(V.n_sets * ns);
      }
      SdgProfileScope scope(prof, s, 2.0 * tok * SIXDGS_D * (double)r, (double)r * (kRowF + T.n_images * 16.0));
      auto kern = V.n_sets > 0 ? k_logits_f16x<0, kOutUB, true> : k_logits_f16x<0, kOutUB, false>;
#ifdef SIXDGS_ABLATION   // timing / power experiments only (tools/power_trace.py abl<N> on a private -DSIXDGS_ABLATION build): the sweep with parts compiled out
      if (const char* ab = getenv("SIXDGS_DEBUG_ABLATE")) {
#define SDG_ABL_CASE(n) case n: kern = V.n_sets > 0 ? k_logits_f16x<n, kOutUB, true> : k_logits_f16x<n, kOutUB, false>; break;
        switch (atoi(ab)) {
          SDG_ABL_CASE(2) SDG_ABL_CASE(18) SDG_ABL_CASE(11) SDG_ABL_CASE(27) SDG_ABL_CASE(59) SDG_ABL_CASE(4096)
          default: break;
        }
#undef SDG_ABL_CASE
      }
#endif
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, s, V);
    }
    hipLaunchKernelGGL(k_merge_stats_slots, dim3((unsigned)T.n_images, 4), dim3(1024), 0, s, w.partial, n_groups, T, (float*)nullptr, gsum);
    hipLaunchKernelGGL(k_sel_finish_slots, dim3((unsigned)sdg_cdiv(r, 1024), (unsigned)T.n_images), dim3(256), 0, s, w.ub, w.p.stride, u_stride, d_n_tok, T, r, u,
                       u_tile_max, u_stride / 256);
  }
  SDG_LAUNCH_OK();
  return 0;
}

int sixdgs_key_planes_norm_max(const void* planes, const float* d_scale, int64_t rows, float* d_norm_max, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(rows >= 0);
  if (rows == 0) return 0;
  SDG_CHECK_ARG(planes && d_scale && d_norm_max && ((uintptr_t)planes % 16) == 0);
  const int64_t blocks = sdg_cdiv(rows, 4);
  hipLaunchKernelGGL(k_plane_norm_max, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, sdg_stream(stream), (const char*)planes, d_scale,
                     rows, reinterpret_cast<unsigned*>(d_norm_max));
  SDG_LAUNCH_OK();
  return 0;
}

// the k largest values of U -- or, given the tile maxima, of THOSE: a lower bound of the k-th largest U is all the candidate stage needs
int sixdgs_select_topk_u(const float* u, int64_t u_stride, int64_t r, const float* u_tile_max, int batch, int topk, float* val, void* ws,
                         size_t ws_bytes, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(batch >= 0 && r >= 1 && u_stride >= r && topk >= 1 && topk <= 1024);
  if (batch == 0) return 0;
  SDG_CHECK_ARG(u && val && ws && ((uintptr_t)ws % 256) == 0);
  SelectWs w;
  if (!select_ws(ws, ws_bytes, r, batch, topk, 8, &w, false)) return SIXDGS_E_WORKSPACE;
  const int64_t nt = sdg_cdiv(r, 256);
  if (u_tile_max && nt >= 2 * (int64_t)topk) return run_topk(u_tile_max, u_stride / 256, nt, batch, topk, w.idxU, val, w.topk_ws, sdg_stream(stream));
  return run_topk(u, u_stride, r, batch, topk, w.idxU, val, w.topk_ws, sdg_stream(stream));
}

int sixdgs_select_candidates(const float* u, int64_t u_stride, int64_t r, const float* u_tile_max, const float* q, const int32_t* d_n_tok, int batch,
                             const float* gsum, const float* d_key_norm_max, const float* d_uk, int topk, int max_candidates, int64_t* cand,
                             int32_t* d_count, void* ws, size_t ws_bytes, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(batch >= 0 && r >= 1 && u_stride >= r && topk >= 1 && topk <= 1024 && max_candidates >= topk && (max_candidates % 8) == 0);
  if (batch == 0) return 0;
  SDG_CHECK_ARG(u && q && d_n_tok && gsum && d_key_norm_max && cand && d_count && ws && ((uintptr_t)ws % 256) == 0 && ((uintptr_t)q % 16) == 0);
  hipStream_t s = sdg_stream(stream);
  SelectWs w;
  if (!select_ws(ws, ws_bytes, r, batch, topk, max_candidates, &w, false)) return SIXDGS_E_WORKSPACE;
  const int k_eff = (int)(r < topk ? r : topk);
  const dim3 cg((unsigned)w.p.nbc, (unsigned)batch);
  const int64_t nt = sdg_cdiv(r, 256);
  const bool tiles = !d_uk && u_tile_max && nt >= 2 * (int64_t)topk;
  const int* none = nullptr;
  if (d_uk) {        // ray-sharded: U_(k) -- or a lower bound of it -- over the rays of ALL shards (the caller merged the shards' sixdgs_select_topk_u lists)
    hipLaunchKernelGGL(k_sel_bounds, dim3((unsigned)batch), dim3(kT), 0, s, gsum, d_n_tok, d_uk, q, d_key_norm_max, 1, 1, w.info, none);
  } else if (!tiles) {
    int st = run_topk(u, u_stride, r, batch, topk, w.idxU, w.valU, w.topk_ws, s);      // U_(k): the k-th largest upper bound
    if (st) return st;
    hipLaunchKernelGGL(k_sel_bounds, dim3((unsigned)batch), dim3(kT), 0, s, gsum, d_n_tok, w.valU, q, d_key_norm_max, topk, k_eff, w.info, none);
  } else {
    // A lower bound of U_(k) is enough for a threshold, and the k-th largest TILE maximum is one (k tiles hold a ray at least that large):
    // a top-k over r / 256 values instead of six passes over the r values of U.  With the top rays scattered over the scene it is
    // practically U_(k) (headline: 113-120 candidates against 107-112).  When they are not -- all of the top rays in a handful of tiles --
    // the bound is far too low; that shows as more than max_candidates candidates, and exactly those images get the exact U_(k): every
    // kernel of the second selection leaves at once for the other images (`need`), so the common case pays only empty launches.
    int* need = w.total + batch;
    int st = run_topk(u_tile_max, u_stride / 256, nt, batch, topk, w.idxU, w.valU, w.topk_ws, s);
    if (st) return st;
    hipLaunchKernelGGL(k_sel_bounds, dim3((unsigned)batch), dim3(kT), 0, s, gsum, d_n_tok, w.valU, q, d_key_norm_max, topk, topk, w.info, none);
    hipLaunchKernelGGL(k_sel_candidates<false>, cg, dim3(256), 0, s, u, u_stride, r, w.info, w.p.span, w.counts, w.offs, cand, max_candidates, none, u_tile_max);
    hipLaunchKernelGGL(k_sel_scan, dim3((unsigned)batch), dim3(1024), 0, s, w.counts, w.p.nbc, w.offs, w.total);
    hipLaunchKernelGGL(k_sel_need_exact, dim3((unsigned)sdg_cdiv(batch, 64)), dim3(64), 0, s, w.total, w.info, batch, max_candidates, need);
    st = run_topk(u, u_stride, r, batch, topk, w.idxU, w.valU, w.topk_ws, s, need);
    if (st) return st;
    hipLaunchKernelGGL(k_sel_bounds, dim3((unsigned)batch), dim3(kT), 0, s, gsum, d_n_tok, w.valU, q, d_key_norm_max, topk, k_eff, w.info, need);
    hipLaunchKernelGGL(k_sel_candidates<false>, cg, dim3(256), 0, s, u, u_stride, r, w.info, w.p.span, w.counts, w.offs, cand, max_candidates, need, u_tile_max);
    hipLaunchKernelGGL(k_sel_scan, dim3((unsigned)batch), dim3(1024), 0, s, w.counts, w.p.nbc, w.offs, w.total);
    hipLaunchKernelGGL(k_sel_candidates<true>, cg, dim3(256), 0, s, u, u_stride, r, w.info, w.p.span, w.counts, w.offs, cand, max_candidates, none, u_tile_max);
  }
  if (!tiles) {
    hipLaunchKernelGGL(k_sel_candidates<false>, cg, dim3(256), 0, s, u, u_stride, r, w.info, w.p.span, w.counts, w.offs, cand, max_candidates, none, u_tile_max);
    hipLaunchKernelGGL(k_sel_scan, dim3((unsigned)batch), dim3(1024), 0, s, w.counts, w.p.nbc, w.offs, w.total);
    hipLaunchKernelGGL(k_sel_candidates<true>, cg, dim3(256), 0, s, u, u_stride, r, w.info, w.p.span, w.counts, w.offs, cand, max_candidates, none, u_tile_max);
  }
  hipLaunchKernelGGL(k_sel_count, dim3((unsigned)sdg_cdiv(batch, 64)), dim3(64), 0, s, w.total, w.info, batch, d_count);
  SDG_LAUNCH_OK();
  return 0;
}

int sixdgs_select_rescore(const float* q, const int32_t* d_n_tok, int batch, const void* planes, const float* d_scale, int compact,
                          const float* ctok, const float* gsum, const int64_t* cand, const int32_t* d_count, int64_t r, int topk,
                          int max_candidates, int allow_fewer, int64_t* idx, float* val, int32_t* d_status, void* ws, size_t ws_bytes,
                          sixdgs_stream_t stream) {
  SDG_CHECK_ARG(batch >= 0 && r >= 1 && topk >= 1 && topk <= 1024 && max_candidates >= topk && (max_candidates % 8) == 0);
  if (batch == 0) return 0;
  SDG_CHECK_ARG(q && d_n_tok && planes && d_scale && ctok && gsum && cand && d_count && idx && val && d_status && ws &&
                ((uintptr_t)planes % 16) == 0 && ((uintptr_t)q % 16) == 0 && ((uintptr_t)ws % 256) == 0);
  hipStream_t s = sdg_stream(stream);
  const int cmax = max_candidates;
  SelectWs w;
  if (!select_ws(ws, ws_bytes, cmax, batch, topk, cmax, &w, false)) return SIXDGS_E_WORKSPACE;
  const int k_eff = (int)(r < topk ? r : topk);
  select_q_planes(q, batch, w, s);
  hipLaunchKernelGGL(k_sel_rescore, dim3((unsigned)(cmax / 8), (unsigned)batch), dim3(kT), 0, s, w.qplanes, w.qinv, (const char*)planes, d_scale,
                     d_n_tok, 0, ctok, gsum, cand, d_count, cmax, compact, w.cscore);
  const int st = run_topk(w.cscore, cmax, cmax, batch, topk, w.lidx, w.lval, w.topk_ws, s);
  if (st) return st;
  hipLaunchKernelGGL(k_sel_emit, dim3((unsigned)batch), dim3(1024), 0, s, w.lidx, w.lval, cand, d_count, cmax, topk, k_eff, allow_fewer, idx, val, d_status);
  SDG_LAUNCH_OK();
  return 0;
}

size_t sixdgs_score_select_workspace_bytes(int64_t r, int batch, int topk, int max_candidates) {
  if (batch < 1) batch = 1;
  // stage scratch + ctok, gsum [B,256], U [B, stride], cand [B, cmax], count [B]
  const size_t stride = (size_t)sdg_cdiv(r > 0 ? r : 1, 256) * 256;
  return sdg_align(sixdgs_select_workspace_bytes(r, batch, topk, max_candidates)) +
         (size_t)batch * (2 * sdg_align(kT * sizeof(float)) + sdg_align(stride * sizeof(float)) + sdg_align(stride / 256 * sizeof(float)) +
                          sdg_align((size_t)max_candidates * sizeof(int64_t)) + 256);
}

int sixdgs_score_select(const float* q, const int32_t* d_n_tok, const int32_t* h_n_tok, int batch, const void* key_planes,
                        const float* d_key_scale, const float* d_key_norm_max, int64_t r, const void* sample_planes, const float* d_sample_scale,
                        int64_t r_sample, int topk, int max_candidates, int64_t* idx, float* val, int32_t* d_status, void* ws, size_t ws_bytes,
                        sixdgs_stream_t stream, sixdgs_profile* prof) {
  SDG_CHECK_ARG(r >= 1 && r_sample >= 1 && r_sample <= r && batch >= 0 && topk >= 1 && topk <= 1024 && max_candidates >= topk &&
                max_candidates <= (1 << 20) && (max_candidates % 8) == 0);
  if (batch == 0) return 0;
  SDG_CHECK_ARG(q && d_n_tok && d_key_norm_max && idx && val && d_status && ws && ((uintptr_t)ws % 256) == 0);
  int64_t bg = batch > 4096 ? 4096 : batch;
  while (bg >= 1 && sixdgs_score_select_workspace_bytes(r, (int)bg, topk, max_candidates) > ws_bytes) --bg;
  if (bg < 1) return SIXDGS_E_WORKSPACE;
  bg = sdg_cdiv(batch, sdg_cdiv(batch, bg));
  const size_t stride = (size_t)sdg_cdiv(r, 256) * 256;
  const size_t stage = sdg_align(sixdgs_select_workspace_bytes(r, (int)bg, topk, max_candidates));
  char* base = (char*)ws + stage;
  float* ctok = (float*)base;                        base += (size_t)bg * sdg_align(kT * sizeof(float));
  float* gsum = (float*)base;                        base += (size_t)bg * sdg_align(kT * sizeof(float));
  float* u = (float*)base;                           base += (size_t)bg * sdg_align(stride * sizeof(float));
  float* utm = (float*)base;                         base += (size_t)bg * sdg_align(stride / 256 * sizeof(float));
  int64_t* cand = (int64_t*)base;                    base += (size_t)bg * sdg_align((size_t)max_candidates * sizeof(int64_t));
  int32_t* count = (int32_t*)base;
  for (int b0 = 0; b0 < batch; b0 += (int)bg) {
    const int nb = (int)((batch - b0) < bg ? (batch - b0) : bg);
    const float* qg = q + (int64_t)b0 * kT * SIXDGS_D;
    const int32_t* ng = d_n_tok + b0;
    int st = sixdgs_select_begin(qg, ng, h_n_tok ? h_n_tok + b0 : nullptr, nb, sample_planes, d_sample_scale, r_sample, r, ctok, gsum, ws, stage, stream);
    if (st) return st;
    st = sixdgs_select_sweep(qg, ng, h_n_tok ? h_n_tok + b0 : nullptr, nb, key_planes, d_key_scale, r, ctok, gsum, u, (int64_t)stride, utm, ws, stage,
                             stream, prof);
    if (st) return st;
    st = sixdgs_select_candidates(u, (int64_t)stride, r, utm, qg, ng, nb, gsum, d_key_norm_max, nullptr, topk, max_candidates, cand, count, ws, stage, stream);
    if (st) return st;
    st = sixdgs_select_rescore(qg, ng, nb, key_planes, d_key_scale, 0, ctok, gsum, cand, count, r, topk, max_candidates, 0,
                               idx + (int64_t)b0 * topk, val + (int64_t)b0 * topk, d_status + b0, ws, stage, stream);
    if (st) return st;
  }
  return 0;
}

}  // extern "C"
