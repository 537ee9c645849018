// vit.hip -- the dense products of the image side (the backbone stage; pose_estimation/backbone.py:82-114 runs DINOv2 ViT-S/14 on every
// query image) with the neighbouring elementwise work folded in: LayerNorm in front of a product, bias / GELU / residual + LayerScale behind it.  A ViT
// block is five launches (LN1+QKV, attention, proj+residual, LN2+FC1+GELU, FC2+residual) instead of twelve.
//
// Why a second GEMM kernel: the token matrix of a batch is M = 257 x images rows (257 .. 4112 for 1 .. 16 images) by K = 384 or 1536.  The 128 x 128 tile
// of gemm.hip leaves most compute units without a workgroup at these sizes and splits BOTH operands in its main loop; the library's fp32-MFMA kernels run
// the four products of a block at 60 .. 115 TFLOP/s at 16 images and at ~8 us each (launch-bound) at one (profiles/r06_vit_stages.md).
//
// Shape of k_tok_gemm (round 6, fourth form; what the first three lost to is in profiles/r06_vit_stages.md):
//   * WEIGHTS are constants: split ONCE (sixdgs_tok_pack) into two scaled fp16 planes, one power-of-two scale per row, and stored in the MFMA's own
//     operand order -- [32-feature block][k-slab][plane][k-step][lane][16 B] -- so that a wave's fragment load is 1 KB of consecutive bytes straight
//     from L2 into registers: no LDS, no barrier and no split work for the weight operand, PFW slabs in flight per wave, loads unconditional (a cursor
//     that stops on the last set) so that the compiler counts them (a load under a branch made it drain the queue every ring revolution).
//   * TOKENS (the activation rows) are staged once per workgroup: a tile of 64 token rows x 384 columns is read as fp32 (16 lanes per row: the row's
//     LayerNorm statistics and its largest magnitude are DPP butterflies over registers the wave holds anyway), normalised if asked, scaled by the
//     row's power of two, split into two fp16 planes and left in LDS (110 KB) for the whole contraction.  K = 1536 (FC2) = four such chunks, each with
//     its own row scale, accumulated in fp32.
//   * a wave owns 32 features x 64 tokens (two 32 x 32 accumulators); a workgroup is EIGHT waves = 256 features x 64 tokens, two waves per SIMD: with
//     four (one per SIMD) every latency of the staging, of the fragment reads and of the epilogue stood exposed (cycle stamps: 3.4 us of matrix work in
//     a 9 us tile).  A workgroup walks over several feature tiles of its token tile when there are more tiles than compute units.
//   * the epilogue's per-feature constants are requested before the slab loop, the residual rows all at once behind it; the GELU's erf is a rational
//     form good to 1.5e-7 (12 VALU operations instead of libm's ~40: the erf was 2 us of a 5 us epilogue).
//   * arithmetic as everywhere in this library: x 2^s = h + l, three cross terms l*h + h*l + h*h on v_mfma_f32_32x32x16_f16, fp32 accumulation.
// Orientation: C[feature][token] (weights = MFMA rows): a lane owns ONE token per accumulator and 4 consecutive features per register group, so the
// token's scale is one factor per lane and the epilogue writes 16-byte pieces of a token's output row.
#include "gemm_kernel.h"
#include <cstdlib>

using namespace sdg;

namespace {

constexpr int kTT = 64;                       // token rows per workgroup tile
constexpr int kWaves = 8;
constexpr int kFT = 32 * kWaves;              // features per workgroup tile (the last tile of a layer may be half full: N is a multiple of 128)
constexpr int kCK = 384, kCS = kCK / 32;      // columns / k-slabs per staged chunk
constexpr int kRow = 144;                     // LDS row of a (token, slab): plane h 64 B | plane l 64 B | 16 B (odd multiple of the 16-byte slot: conflict-free fragment reads)
constexpr int kSlabLds = kTT * kRow;
constexpr int kFragSet = 4096;                // bytes of one (32-feature block, k-slab): [plane 2][k-step 2][lane 64][16 B]

enum { kAPlain = 0, kALayerNorm = 1 };
enum { kEpiBias = 0, kEpiGelu = 1, kEpiResid = 2 };

struct TokArgs {
  const float* x;        // [M][lda] rows
  const char* wp;        // packed weight planes (sixdgs_tok_pack): [N / 32][K / 32][kFragSet]
  const float* winv;     // [N] reciprocal row scales of the packed weights
  const float* bias;     // [N] or null
  const float* ln_g;     // [K] LayerNorm weight / bias (kALayerNorm)
  const float* ln_b;
  const float* res;      // [M][ldr] residual stream (kEpiResid)
  const float* gamma;    // [N] LayerScale (kEpiResid; null = 1)
  float* y;              // [M][ldy]
  int64_t m, lda, ldy, ldr;
  int n, k;
  float ln_eps;
  int ft_per_wg;         // feature tiles a workgroup walks over
};

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7): 1 - (a1 t + .. + a5 t^5) exp(-x^2), t = 1 / (1 + p |x|)
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  return copysignf(1.f - p * t * e, x);
}
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erf_as(v * 0.70710678118654752f)); }

// butterflies over the 16 lanes of a DPP row (a token row's 16 lanes): rotations by 8 and 4 within the row, then the two quad permutations -- four VALU
// operations with a DPP operand each, no LDS traffic (the generic __shfl_xor goes through ds_bpermute_b32)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float red16_sum(float v) {
  v += dpp_f<0x128>(v);      // row_ror:8
  v += dpp_f<0x124>(v);      // row_ror:4
  v += dpp_f<0x4e>(v);       // quad_perm:[2,3,0,1]
  v += dpp_f<0xb1>(v);       // quad_perm:[1,0,3,2]
  return v;
}
__device__ __forceinline__ float red16_max(float v) {
  v = fmaxf(v, dpp_f<0x128>(v));
  v = fmaxf(v, dpp_f<0x124>(v));
  v = fmaxf(v, dpp_f<0x4e>(v));
  v = fmaxf(v, dpp_f<0xb1>(v));
  return v;
}

typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));

#ifdef SDG_TOK_PROF      // developer build (SIXDGS_EXTRA_FLAGS=-DSDG_TOK_PROF, tools/prof_tok.py): cycle stamps of workgroup (0, 0)'s sections
__device__ long long g_tok_prof[16];
#define TOK_T(K) if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_tok_prof[K] = wall_clock64();
#else
#define TOK_T(K)
#endif

// MULTI: the contraction has more than one chunk of 384 columns (FC2): every chunk is staged with its own row scales and folded into a second
// accumulator set; single-chunk layers scale their one accumulator set in the epilogue.
template <int AMODE, int EPI, bool MULTI, int PFW>
__global__ void __launch_bounds__(64 * kWaves) k_tok_gemm(TokArgs A) {
  static_assert(kCS % PFW == 0, "the ring of weight fragments in flight divides the slabs of a chunk");
  __shared__ __attribute__((aligned(16))) char sa[kCS * kSlabLds];                    // the token tile's planes: [slab][token][kRow]
  __shared__ __attribute__((aligned(16))) float lnp[AMODE == kALayerNorm ? 2 * kCK : 4];
  __shared__ float inv_as[kTT];                                                        // reciprocal scale of every token row of the staged chunk
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row0 = (int64_t)blockIdx.x * kTT;
  const int n_fb = A.n >> 5;                                                            // 32-feature blocks of the layer
  const int n_ft = (A.n + kFT - 1) / kFT;
  const int ft0 = (int)blockIdx.y * A.ft_per_wg;
  const int ft1 = ft0 + A.ft_per_wg < n_ft ? ft0 + A.ft_per_wg : n_ft;
  const int KS = A.k >> 5, KC = MULTI ? A.k / kCK : 1;
  TOK_T(0)

  // ---- weight fragments: PFW (block, slab) sets in flight; the cursor runs on over chunks and feature tiles and stops on the wave's last set -----------
  // (a wave whose block lies beyond the layer in the last, half-full tile reads block 0 and stores nothing)
  uint4 wf[PFW][4];
  int pf_gs = 0, pf_ft = ft0;
  const char* const wlane = A.wp + lane * 16;
  auto blk_of = [&](const int ft) { const int b = ft * kWaves + wave; return b < n_fb ? b : 0; };
  size_t pf_blk = (size_t)blk_of(ft0) * KS;
  auto issue = [&](const int slot) {
    const char* p = wlane + (pf_blk + pf_gs) * kFragSet;
#pragma unroll
    for (int q = 0; q < 4; ++q) wf[slot][q] = *reinterpret_cast<const uint4*>(p + q * 1024);
    if (pf_gs + 1 < KS) {
      ++pf_gs;
    } else if (pf_ft + 1 < ft1) {
      ++pf_ft;
      pf_gs = 0;
      pf_blk = (size_t)blk_of(pf_ft) * KS;
    }
  };
#pragma unroll
  for (int s = 0; s < PFW; ++s) issue(s);

  if (AMODE == kALayerNorm) {
    for (int i = tid; i < kCK; i += 64 * kWaves) { lnp[i] = A.ln_g[i]; lnp[kCK + i] = A.ln_b[i]; }
    __syncthreads();
  }

  // ---- staging of chunk c of the token tile: 16 lanes per row, 4 rows per wave and pass, 2 passes ---------------------------------------------------------
  auto stage = [&](const int c) {
    const int sub = lane & 15, tq = lane >> 4;
#pragma unroll
    for (int p = 0; p < kTT / (4 * kWaves); ++p) {
      const int t = p * 4 * kWaves + wave * 4 + tq;
      const int64_t gr = row0 + t < A.m ? row0 + t : A.m - 1;                           // rows beyond M: a valid row is read, nothing is stored
      const float* src = A.x + gr * A.lda + c * kCK + 4 * sub;
      float4 v[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) v[i] = *reinterpret_cast<const float4*>(src + i * 64);
      if (AMODE == kALayerNorm) {      // torch.nn.LayerNorm: biased variance around the mean, two passes over the registers
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) s1 += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mean = red16_sum(s1) * (1.f / kCK);
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
          s2 += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
        const float rstd = rsqrtf(red16_sum(s2) * (1.f / kCK) + A.ln_eps);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const float4 g = *reinterpret_cast<const float4*>(lnp + 4 * (sub + 16 * i)), b = *reinterpret_cast<const float4*>(lnp + kCK + 4 * (sub + 16 * i));
          v[i].x = v[i].x * rstd * g.x + b.x; v[i].y = v[i].y * rstd * g.y + b.y; v[i].z = v[i].z * rstd * g.z + b.z; v[i].w = v[i].w * rstd * g.w + b.w;
        }
      }
      float mx = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
      mx = red16_max(mx);
      const float sc = f3_scale(mx);                                                     // row maximum to [2^13, 2^14): exact power of two
      if (sub == 0) inv_as[t] = f3_inv_scale(mx);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const float e[4] = {v[i].x * sc, v[i].y * sc, v[i].z * sc, v[i].w * sc};
        f16x4_t h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const _Float16 hh = (_Float16)e[j];
          h[j] = hh;
          l[j] = (_Float16)(e[j] - (float)hh);
        }
        char* d = sa + ((2 * i + (sub >> 3)) * kTT + t) * kRow + (sub & 7) * 8;         // columns 4 (sub + 16 i) .. + 3: slab 2 i + (sub >> 3), position 4 (sub & 7)
        *reinterpret_cast<f16x4_t*>(d) = h;
        *reinterpret_cast<f16x4_t*>(d + 64) = l;
      }
    }
  };

  const int tokl = lane & 31, half = lane >> 5;
  const char* fr = sa + tokl * kRow + half * 16;                                         // this lane's fragment pieces: + (slab * 64 + 32 tt) * kRow + plane * 64 + kstep * 32
  for (int ft = ft0; ft < ft1; ++ft) {
    const int fb = ft * kWaves + wave;
    const bool active = fb < n_fb;                                                       // (wave-uniform)
    const int fw = (active ? fb : 0) * 32 + 4 * half;                                    // this lane's features: fw + 8 rg + j
    // the epilogue's per-feature constants: requested now, used behind the slab loop
    float4 iw[4], bv[4], gv[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      iw[rg] = *reinterpret_cast<const float4*>(A.winv + fw + 8 * rg);
      bv[rg] = A.bias ? *reinterpret_cast<const float4*>(A.bias + fw + 8 * rg) : float4{0.f, 0.f, 0.f, 0.f};
      gv[rg] = (EPI == kEpiResid && A.gamma) ? *reinterpret_cast<const float4*>(A.gamma + fw + 8 * rg) : float4{1.f, 1.f, 1.f, 1.f};
    }
    f32x16 tot[2];
    f32x16 acc[2];
    if (MULTI) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { tot[0][i] = 0.f; tot[1][i] = 0.f; }
    }
#pragma unroll 1
    for (int c = 0; c < KC; ++c) {      // chunks of the contraction: ONE copy of the staging and of the slab loop in the code
      if (ft == ft0 || MULTI) {
        if (c > 0 || ft > ft0) __syncthreads();                                          // every wave is done with the planes about to be overwritten
        TOK_T(1)
        stage(c);
        __syncthreads();
        TOK_T(2)
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
#pragma unroll
      for (int s = 0; s < kCS; ++s) {
        const int slot = s % PFW;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f16x8_t b[2][2];
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) b[tt][pl] = *reinterpret_cast<const f16x8_t*>(fr + (s * kTT + 32 * tt) * kRow + pl * 64 + j * 32);
          const f16x8_t wh = __builtin_bit_cast(f16x8_t, wf[slot][j]), wl = __builtin_bit_cast(f16x8_t, wf[slot][2 + j]);
          // (weight plane, token plane): l*h, h*l, h*h -- smallest magnitude first
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, b[tt][0], acc[tt], 0, 0, 0);
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, b[tt][1], acc[tt], 0, 0, 0);
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, b[tt][0], acc[tt], 0, 0, 0);
        }
        issue(slot);
      }
      if (MULTI) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const float ia = inv_as[32 * tt + tokl];
#pragma unroll
          for (int i = 0; i < 16; ++i) tot[tt][i] += acc[tt][i] * ia;
        }
      }
      TOK_T(3)
    }

    // ---- epilogue: lane = token tokl (+ 32 tt), register 4 rg + j = feature 8 rg + 4 half + j of the wave's 32 ---------------------------------------------
    if (active) {
      float4 r4[2][4];
      bool live[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int64_t gr = row0 + 32 * tt + tokl;
        live[tt] = gr < A.m;
        if (EPI == kEpiResid) {
          const float* rp = A.res + (live[tt] ? gr : A.m - 1) * A.ldr + fw;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) r4[tt][rg] = *reinterpret_cast<const float4*>(rp + 8 * rg);
        }
      }
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const float ia = MULTI ? 1.f : inv_as[32 * tt + tokl];
        float* const yp = A.y + (row0 + 32 * tt + tokl) * A.ldy + fw;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const f32x16& a = MULTI ? tot[tt] : acc[tt];
          float4 v = {fmaf(a[4 * rg] * ia, iw[rg].x, bv[rg].x), fmaf(a[4 * rg + 1] * ia, iw[rg].y, bv[rg].y), fmaf(a[4 * rg + 2] * ia, iw[rg].z, bv[rg].z),
                      fmaf(a[4 * rg + 3] * ia, iw[rg].w, bv[rg].w)};
          if (EPI == kEpiGelu) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
          if (EPI == kEpiResid) {
            v.x = fmaf(gv[rg].x, v.x, r4[tt][rg].x); v.y = fmaf(gv[rg].y, v.y, r4[tt][rg].y); v.z = fmaf(gv[rg].z, v.z, r4[tt][rg].z); v.w = fmaf(gv[rg].w, v.w, r4[tt][rg].w);
          }
          if (live[tt]) *reinterpret_cast<float4*>(yp + 8 * rg) = v;
        }
      }
    }
    TOK_T(4)
  }
}

// fp32 weights [n][ldw] -> the packed planes of k_tok_gemm + reciprocal row scales.  One workgroup per block of 32 rows.
__global__ void __launch_bounds__(256) k_tok_pack(const float* __restrict__ w, int k, int64_t ldw, char* __restrict__ planes, float* __restrict__ winv) {
  __shared__ float rmax[32];
  const int tid = threadIdx.x, fb = blockIdx.x;
  {
    const int row = tid >> 3, part = tid & 7;
    const float* src = w + (int64_t)(fb * 32 + row) * ldw;
    float mx = 0.f;
    for (int c = part; c < k; c += 8) mx = fmaxf(mx, fabsf(src[c]));
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64)); mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
    if (part == 0) { rmax[row] = mx; winv[fb * 32 + row] = f3_inv_scale(mx); }
  }
  __syncthreads();
  const int KS = k >> 5;
  for (int idx = tid; idx < KS * 128; idx += 256) {
    const int gs = idx >> 7, j = (idx >> 6) & 1, l = idx & 63;
    const int row = l & 31, k0 = gs * 32 + j * 16 + 8 * (l >> 5);
    const float* src = w + (int64_t)(fb * 32 + row) * ldw + k0;
    const float sc = f3_scale(rmax[row]);
    f16x8_t h, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = src[e] * sc;
      const _Float16 hh = (_Float16)x;
      h[e] = hh;
      lo[e] = (_Float16)(x - (float)hh);
    }
    char* d = planes + ((size_t)(fb * KS + gs) * 4 + j) * 1024 + l * 16;
    *reinterpret_cast<f16x8_t*>(d) = h;
    *reinterpret_cast<f16x8_t*>(d + 2048) = lo;
  }
}

// ---- attention of a ViT block (dinov2 Attention.forward: softmax(q k^T / sqrt(64)) v per image and head) on the [M][3 * heads * 64] output of the QKV
// product, token-major in and out -- no head-major copies, one launch.  PyTorch's scaled_dot_product_attention (aotriton attn_fwd) takes 27 us for one
// image and 43 us for sixteen: 12 launches of it are a third of the ViT at one image.
// One workgroup = one (image, head, group of 128 queries); 4 waves, a wave owns 32 queries against ALL keys (T <= 288 = 9 key tiles):
//   S^T[key][query] = sum_d K[key][d] Q[query][d]: keys are MFMA rows, queries MFMA columns, so a lane owns ONE query (+ its partner lane 32 further the
//   other half of that query's keys): the softmax over keys is a reduction over the lane's own 144 registers and one lane exchange -- no LDS, no barrier;
//   the probabilities never leave the registers: the accumulator layout of S^T is, up to a permutation of the keys inside a tile, the B-operand layout
//   of O^T[d][query] = sum_key V^T[d][key] P^T[key][query]; the permutation is applied to the keys of V's fragments instead (two 8-byte reads).
// K (as [key][d]) and V (transposed, [d][key]) of the head are staged once per workgroup as two scaled fp16 planes each (one scale per head: its largest
// magnitude), Q stays in registers (one scale per query, the 1/8 folded in).  Arithmetic as everywhere: h + l planes, three cross terms, fp32 accumulation;
// exp2 on fp32 logits; P as h + l planes of p * 2^14.
constexpr int kAtD = 64;                      // head dimension
constexpr int kAtKT = 9;                      // key tiles of 32: tokens <= 288
constexpr int kAtKRow = 2 * kAtD * 2 + 16;    // LDS row of a key: plane h 128 B | plane l 128 B | 16 B (17 slots of 16 B: odd)
constexpr int kAtVPlane = kAtKT * 32 * 2;     // bytes of one plane of a V^T row: 288 keys
constexpr int kAtVRow = 2 * kAtVPlane + 8;    // 1160 B: 290 dwords, 290 mod 64 = 34: the 32 rows of a fragment read start on 32 distinct even banks

struct AttnArgs {
  const float* qkv;      // [images * tokens][ldq]: q | k | v, each heads * 64 wide
  float* y;              // [images * tokens][ldy]
  int64_t ldq, ldy;
  int tokens, heads;
};

__global__ void __launch_bounds__(256) k_tok_attn(AttnArgs A) {
  __shared__ __attribute__((aligned(16))) char Kp[kAtKT * 32 * kAtKRow];
  __shared__ __attribute__((aligned(16))) char Vt[kAtD * kAtVRow];
  __shared__ float wmax[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img = (int)blockIdx.x / A.heads, head = (int)blockIdx.x - img * A.heads;
  const int T = A.tokens, width = A.heads * kAtD;
  const float* base = A.qkv + (int64_t)img * T * A.ldq + head * kAtD;      // q of token 0; k at + width, v at + 2 width

  // ---- stage K and V^T: thread = (key t / 16 + 16 i, 4 consecutive d) ------------------------------------------------------------------------------------
  {
    constexpr int NI = (kAtKT * 32 + 15) / 16;      // 18 passes of 16 keys
    const int c4 = (tid & 15) * 4, k0 = tid >> 4;
    float4 kv[NI], vv[NI];
    float mk = 0.f, mv = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int key = k0 + 16 * i;
      if (key < T) {
        const float* p = base + (int64_t)key * A.ldq + width + c4;
        kv[i] = *reinterpret_cast<const float4*>(p);
        vv[i] = *reinterpret_cast<const float4*>(p + width);
      } else {
        kv[i] = float4{0.f, 0.f, 0.f, 0.f};
        vv[i] = float4{0.f, 0.f, 0.f, 0.f};
      }
      mk = fmaxf(fmaxf(mk, fmaxf(fabsf(kv[i].x), fabsf(kv[i].y))), fmaxf(fabsf(kv[i].z), fabsf(kv[i].w)));
      mv = fmaxf(fmaxf(mv, fmaxf(fabsf(vv[i].x), fabsf(vv[i].y))), fmaxf(fabsf(vv[i].z), fabsf(vv[i].w)));
    }
    mk = sdg_wave_max(mk);
    mv = sdg_wave_max(mv);
    if (lane == 0) { wmax[0][wave] = mk; wmax[1][wave] = mv; }
    __syncthreads();
    mk = fmaxf(fmaxf(wmax[0][0], wmax[0][1]), fmaxf(wmax[0][2], wmax[0][3]));
    mv = fmaxf(fmaxf(wmax[1][0], wmax[1][1]), fmaxf(wmax[1][2], wmax[1][3]));
    const float sk = f3_scale(mk), sv = f3_scale(mv);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int key = k0 + 16 * i;
      if (key >= kAtKT * 32) continue;
      const float ke[4] = {kv[i].x * sk, kv[i].y * sk, kv[i].z * sk, kv[i].w * sk};
      const float ve[4] = {vv[i].x * sv, vv[i].y * sv, vv[i].z * sv, vv[i].w * sv};
      f16x4_t kh, kl;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const _Float16 hh = (_Float16)ke[j];
        kh[j] = hh;
        kl[j] = (_Float16)(ke[j] - (float)hh);
        const _Float16 vh = (_Float16)ve[j];
        *reinterpret_cast<_Float16*>(Vt + (c4 + j) * kAtVRow + key * 2) = vh;
        *reinterpret_cast<_Float16*>(Vt + (c4 + j) * kAtVRow + kAtVPlane + key * 2) = (_Float16)(ve[j] - (float)vh);
      }
      *reinterpret_cast<f16x4_t*>(Kp + key * kAtKRow + c4 * 2) = kh;
      *reinterpret_cast<f16x4_t*>(Kp + key * kAtKRow + 2 * kAtD + c4 * 2) = kl;
    }
    __syncthreads();      // (the reciprocal scales are recomputed from mk / mv below: exact powers of two)
    // ---- this wave's query tiles of 32: tile (blockIdx.y * 4 + wave), then every (4 gridDim.y)-th (with 3 groups and 257 tokens one tile per wave; with
    //      2 groups -- chosen by the launcher when 3 would need a second round of workgroups -- one wave of the image takes the ninth tile too) -----------
    const int ql = lane & 31, half = lane >> 5;
#pragma unroll 1
    for (int q0 = ((int)blockIdx.y * 4 + wave) * 32; q0 < T; q0 += 128 * (int)gridDim.y) {
    const int q = q0 + ql < T ? q0 + ql : T - 1;                                  // queries beyond T: a valid row is computed, nothing is stored
    const float* qp = base + (int64_t)q * A.ldq + 8 * half;
    float4 qa[4][2];
    float mq = 0.f;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      qa[s4][0] = *reinterpret_cast<const float4*>(qp + 16 * s4);
      qa[s4][1] = *reinterpret_cast<const float4*>(qp + 16 * s4 + 4);
#pragma unroll
      for (int u = 0; u < 2; ++u) mq = fmaxf(fmaxf(mq, fmaxf(fabsf(qa[s4][u].x), fabsf(qa[s4][u].y))), fmaxf(fabsf(qa[s4][u].z), fabsf(qa[s4][u].w)));
    }
    mq = fmaxf(mq, __shfl_xor(mq, 32, 64));
    const float sq = f3_scale(mq);
    f16x8_t qh[4], qlo[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const float e[8] = {qa[s4][0].x, qa[s4][0].y, qa[s4][0].z, qa[s4][0].w, qa[s4][1].x, qa[s4][1].y, qa[s4][1].z, qa[s4][1].w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = e[j] * sq;
        const _Float16 hh = (_Float16)x;
        qh[s4][j] = hh;
        qlo[s4][j] = (_Float16)(x - (float)hh);
      }
    }
    // ---- S^T = K Q^T ------------------------------------------------------------------------------------------------------------------------------------------
    f32x16 acc[kAtKT];
    // (fragment addresses from an opaque copy of the lane index, formed HERE: K and V^T do not change between the query tiles of a wave, and as loop
    //  invariants every one of their ~200 fragment reads was hoisted out of the tile loop and spilled -- 434 registers)
    unsigned lo_ = (unsigned)lane;
    asm volatile("" : "+v"(lo_));
    const char* kfr = Kp + (lo_ & 31u) * kAtKRow + (lo_ >> 5) * 16;
#pragma unroll
    for (int kt = 0; kt < kAtKT; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[kt][i] = 0.f;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const f16x8_t kh = *reinterpret_cast<const f16x8_t*>(kfr + kt * 32 * kAtKRow + s4 * 32);
        const f16x8_t kl = *reinterpret_cast<const f16x8_t*>(kfr + kt * 32 * kAtKRow + 2 * kAtD + s4 * 32);
        acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s4], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qlo[s4], acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s4], acc[kt], 0, 0, 0);
      }
    }
    // ---- softmax over the keys of this lane's query: logits * log2(e) / 8, exp2 --------------------------------------------------------------------------------
    const float f = f3_inv_scale(mq) * f3_inv_scale(mk) * (0.125f * 1.4426950408889634f);
    float mx = -__builtin_inff();
    const int half_o = (int)(lo_ >> 5);      // (opaque: as invariants of the tile loop the 144 key masks were hoisted into 288 scalar registers and spilled)
#pragma unroll
    for (int kt = 0; kt < kAtKT; ++kt) {
      if ((kt + 1) * 32 <= T) {              // a full key tile (wave-uniform): no masks
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float t = acc[kt][i] * f;
          acc[kt][i] = t;
          mx = fmaxf(mx, t);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * half_o;
          const float t = key < T ? acc[kt][i] * f : -__builtin_inff();
          acc[kt][i] = t;
          mx = fmaxf(mx, t);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < kAtKT; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float p = __builtin_amdgcn_exp2f(acc[kt][i] - mx);
        acc[kt][i] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32, 64);
    // ---- O^T = V^T P^T: the lane's probabilities of a key tile are the B operand as they lie (keys (e & 3) + 8 (e >> 2) + 16 s2 + 4 half, e = 0..7) ---------
    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
    const char* vfr = Vt + (lo_ & 31u) * kAtVRow + (lo_ >> 5) * 8;
#pragma unroll
    for (int kt = 0; kt < kAtKT; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        f16x8_t ph, pl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = acc[kt][8 * s2 + e] * 16384.f;
          const _Float16 hh = (_Float16)x;
          ph[e] = hh;
          pl[e] = (_Float16)(x - (float)hh);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const char* vp = vfr + dt * 32 * kAtVRow + (kt * 32 + 16 * s2) * 2;
          const f16x4_t h0 = *reinterpret_cast<const f16x4_t*>(vp), h1 = *reinterpret_cast<const f16x4_t*>(vp + 16);
          const f16x4_t l0 = *reinterpret_cast<const f16x4_t*>(vp + kAtVPlane), l1 = *reinterpret_cast<const f16x4_t*>(vp + kAtVPlane + 16);
          const f16x8_t vh = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
          const f16x8_t vl = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[dt], 0, 0, 0);
        }
      }
    if (q0 + ql < T) {
      const float g = f3_inv_scale(mv) * (1.f / 16384.f) / sum;
      float* yp = A.y + ((int64_t)img * T + q0 + ql) * A.ldy + head * kAtD + 4 * half;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          *reinterpret_cast<float4*>(yp + dt * 32 + 8 * rg) = float4{o[dt][4 * rg] * g, o[dt][4 * rg + 1] * g, o[dt][4 * rg + 2] * g, o[dt][4 * rg + 3] * g};
    }
    }      // query tiles of this wave
  }
}

// a22's convolutions as GEMMs (camera_direction_network.py:29-36: valid k x k convolutions of the 16 x 16 feature map): the A matrix of a whole batch in
// ONE launch from a feature map with FREE strides, so the previous layer's GEMM output [B * ho * wo][C] is read where it lies (no permute copy).
// PyTorch's unfold launches one im2col kernel per IMAGE (64 launches of ~8 us for 16 images and four layers) and needed a transposing copy behind it.
// Row (b, oy, ox); column order 0: (c, ky, kx) -- conv.weight.view(out, -1) -- or 1: (ky, kx, c), which with channel-contiguous input makes every
// (row, tap) a plain copy of C consecutive floats (the caller permutes the weight's columns once).  Consecutive threads write consecutive outputs.
template <int ORDER, int VEC>
__global__ void __launch_bounds__(256) k_im2col(const float* __restrict__ x, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int C, int k, int ho, int wo,
                                                int64_t total, float* __restrict__ a) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;      // first output element of this thread
  if (i >= total) return;
  const int kk = k * k, cols = C * kk;
  const int64_t row = i / cols;
  const int j = (int)(i - row * cols);
  const int ox = (int)(row % wo);
  const int64_t t = row / wo;
  const int oy = (int)(t % ho);
  const int64_t b = t / ho;
  const float* src = x + b * sb + oy * sy + ox * sx;
  if (ORDER == 1) {
    const int tap = j / C, c = j - tap * C, ky = tap / k, kx = tap - ky * k;
    const float* p = src + ky * sy + kx * sx + c * sc;
    if (VEC == 4) *reinterpret_cast<float4*>(a + i) = *reinterpret_cast<const float4*>(p);      // (sc == 1, C % 4 == 0: checked by the launcher)
    else a[i] = *p;
  } else {
    const int c = j / kk, r = j - c * kk, ky = r / k, kx = r - ky * k;
    a[i] = src[c * sc + ky * sy + kx * sx];
  }
}

// a16's first step for a batch (pose_estimation/test.py:69-73: uint8 image / 255.0): [B][H][W][3] uint8 -> [B][3][H][W] fp32 through the caller's
// 256-entry table (built on the CPU with the reference's true division), planar so that the resize that follows reads contiguous rows.  Replaces
// stack + .long() + table lookup + stack + channels-last copy (five PyTorch kernels, 0.32 ms for 16 images of 800 x 800) with one pass.
__global__ void __launch_bounds__(256) k_u8_to_planar(const uint8_t* __restrict__ in, const float* __restrict__ lut, int64_t hw, int64_t total4, float* __restrict__ out) {
  __shared__ float tab[256];
  tab[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  // one thread = 4 consecutive pixels of one image: 12 bytes in, 3 x 16 bytes out
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t per = hw >> 2;
    const int64_t b = i / per, p4 = i - b * per;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(in + (b * hw + 4 * p4) * 3);
    const uint32_t w0 = src[0], w1 = src[1], w2 = src[2];
    const uint8_t px[12] = {(uint8_t)w0, (uint8_t)(w0 >> 8), (uint8_t)(w0 >> 16), (uint8_t)(w0 >> 24), (uint8_t)w1, (uint8_t)(w1 >> 8), (uint8_t)(w1 >> 16),
                            (uint8_t)(w1 >> 24), (uint8_t)w2, (uint8_t)(w2 >> 8), (uint8_t)(w2 >> 16), (uint8_t)(w2 >> 24)};
    float* dst = out + b * 3 * hw + 4 * p4;
#pragma unroll
    for (int c = 0; c < 3; ++c) *reinterpret_cast<float4*>(dst + c * hw) = float4{tab[px[c]], tab[px[3 + c]], tab[px[6 + c]], tab[px[9 + c]]};
  }
}

// a16 + the transform pipeline of backbone.py:52-77 for a batch of RGB images in ONE pass (round 6): uint8 [B][H][W][3] -> table lookup (value / 255, the caller's
// CPU-built table) -> antialiased bicubic resize to (nh, nw) -> centre crop S x S -> (x - mean) / std -> fp32 [B][3][S][S].  Replaces k_u8_to_planar + PyTorch's
// upsample_gen2d_aa (every output pixel re-forms its ~15 x 15 window: 0.29 ms per 16 images of 800 x 800) + the crop / normalise kernels.
// The arithmetic is that of the op it replaces -- PyTorch's antialias path (UpSampleBilinear2d.cu: _compute_weights_span, _compute_weights, the a = -0.5 cubic filter,
// interpolate_aa_single_dim), in its order: per output index i of the RESIZED grid, centre = scale (i + 0.5), span [xmin, xmin + xsize) = (int)(centre -+ support + 0.5)
// clipped to the input, weights filter((j + xmin - centre + 0.5) / scale) normalised by their sum in tap order; a window's rows are first reduced along x, then along y, each as
// a tap-order sum that starts from the first product.  What differs is only WHO computes a row's x-reduction: here once per (input row, output column) of a band of
// output rows, shared through LDS by the band's rows (separable), there once per output pixel.  The same products in the same order: equal up to the compiler's choice of
// fused multiply-adds (tests: <= 2e-6 of the op's result on [-2.2, 2.7]-ranged values).
// One workgroup = (band of rb output rows, image).  LDS: table | wx [S][kt] | x spans | wy [rb][kt] | y spans | x-reduced rows [nr][3][S].
struct PrepArgs {
  const uint8_t* in;
  const float* lut;
  float* out;
  int H, W, top, left, S, rb, kt, nrmax;
  float sh, sw, sup_h, sup_w;       // input / resized size per axis; support = 2 scale (scale >= 1), else 2
  float mean[3], stdv[3];
  int64_t last_dword;               // index of the last dword that holds image bytes (loads are clamped to it)
};

__device__ __forceinline__ float aa_cubic(float x) {      // upsample_antialias::BicubicFilterFunctor, a = -0.5
  const float a = -0.5f;
  x = fabsf(x);
  if (x < 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
  if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * a;
  return 0.f;
}

// span and normalised weights of output index i (of the resized grid) along an axis of n input samples; w[0 .. kt): zero beyond xsize
__device__ __forceinline__ void aa_weights(int i, int n, float scale, float support, int kt, float* w, int* meta) {
  const float center = scale * ((float)i + 0.5f);
  const int xmin = max((int)(center - support + 0.5f), 0);
  const int xsize = min((int)(center + support + 0.5f), n) - xmin;
  const float invscale = scale >= 1.f ? 1.f / scale : 1.f;
  const float xmc = (float)xmin - center;
  float total = 0.f;
  for (int j = 0; j < kt; ++j) {
    float v = 0.f;
    if (j < xsize) {
      v = aa_cubic(((float)j + xmc + 0.5f) * invscale);
      total += v;
    }
    w[j] = v;
  }
  if (total != 0.f)
    for (int j = 0; j < xsize && j < kt; ++j) w[j] /= total;
  meta[0] = xmin;
  meta[1] = xsize < kt ? xsize : kt;
}

__global__ void __launch_bounds__(256) k_image_prep(PrepArgs A) {
  extern __shared__ __attribute__((aligned(16))) float prep_sm[];
  float* const tab = prep_sm;
  float* const wx = tab + 256;
  int* const xmeta = reinterpret_cast<int*>(wx + A.S * A.kt);
  float* const wy = reinterpret_cast<float*>(xmeta + 2 * A.S);
  int* const ymeta = reinterpret_cast<int*>(wy + A.rb * A.kt);
  float* const hbuf = reinterpret_cast<float*>(ymeta + 2 * A.rb);
  const int tid = threadIdx.x, b = blockIdx.y, oy0 = blockIdx.x * A.rb;
  const int rows = min(A.rb, A.S - oy0);
  tab[tid] = A.lut[tid];
  for (int x = tid; x < A.S; x += 256) aa_weights(A.left + x, A.W, A.sw, A.sup_w, A.kt, wx + x * A.kt, xmeta + 2 * x);
  if (tid < rows) aa_weights(A.top + oy0 + tid, A.H, A.sh, A.sup_h, A.kt, wy + tid * A.kt, ymeta + 2 * tid);
  __syncthreads();
  const int r0 = ymeta[0];
  int r1 = r0;
  for (int ry = 0; ry < rows; ++ry) r1 = max(r1, ymeta[2 * ry] + ymeta[2 * ry + 1]);
  const int nr = min(r1 - r0, A.nrmax);
  // ---- along x: item = (input row r0 + r, output column x); its 3 * xsize bytes start anywhere: aligned dwords, shifted into place (v_alignbyte), 4 pixels = 3 dwords at a time
  const uint32_t* const base = reinterpret_cast<const uint32_t*>(A.in);
  const int groups = A.kt >> 2;
  for (int it = tid; it < nr * A.S; it += 256) {
    const int r = it / A.S, x = it - r * A.S;
    const int xmin = xmeta[2 * x];
    const int64_t byte0 = (((int64_t)b * A.H + (r0 + r)) * A.W + xmin) * 3;
    const unsigned sh8 = (unsigned)(byte0 & 3);
    const int64_t d0i = byte0 >> 2;
    const float* w = wx + x * A.kt;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    uint32_t carry = base[min(d0i, A.last_dword)];
    for (int g = 0; g < groups; ++g) {
      const uint32_t n0 = base[min(d0i + 3 * g + 1, A.last_dword)], n1 = base[min(d0i + 3 * g + 2, A.last_dword)], n2 = base[min(d0i + 3 * g + 3, A.last_dword)];
      const uint32_t e0 = __builtin_amdgcn_alignbyte(n0, carry, sh8), e1 = __builtin_amdgcn_alignbyte(n1, n0, sh8), e2 = __builtin_amdgcn_alignbyte(n2, n1, sh8);
      carry = n2;
      const float4 wv = *reinterpret_cast<const float4*>(w + 4 * g);
      a0 = __builtin_fmaf(tab[e0 & 255u], wv.x, a0);
      a1 = __builtin_fmaf(tab[(e0 >> 8) & 255u], wv.x, a1);
      a2 = __builtin_fmaf(tab[(e0 >> 16) & 255u], wv.x, a2);
      a0 = __builtin_fmaf(tab[e0 >> 24], wv.y, a0);
      a1 = __builtin_fmaf(tab[e1 & 255u], wv.y, a1);
      a2 = __builtin_fmaf(tab[(e1 >> 8) & 255u], wv.y, a2);
      a0 = __builtin_fmaf(tab[(e1 >> 16) & 255u], wv.z, a0);
      a1 = __builtin_fmaf(tab[e1 >> 24], wv.z, a1);
      a2 = __builtin_fmaf(tab[e2 & 255u], wv.z, a2);
      a0 = __builtin_fmaf(tab[(e2 >> 8) & 255u], wv.w, a0);
      a1 = __builtin_fmaf(tab[(e2 >> 16) & 255u], wv.w, a1);
      a2 = __builtin_fmaf(tab[e2 >> 24], wv.w, a2);
    }
    hbuf[(r * 3 + 0) * A.S + x] = a0;
    hbuf[(r * 3 + 1) * A.S + x] = a1;
    hbuf[(r * 3 + 2) * A.S + x] = a2;
  }
  __syncthreads();
  // ---- along y, normalisation, planar store (consecutive threads = consecutive x: conflict-free LDS reads, coalesced stores)
  for (int it = tid; it < rows * 3 * A.S; it += 256) {
    const int x = it % A.S, c = (it / A.S) % 3, ry = it / (3 * A.S);
    const int ymin = ymeta[2 * ry] - r0, ysize = ymeta[2 * ry + 1];
    const float* w = wy + ry * A.kt;
    float acc = 0.f;
    for (int j = 0; j < ysize; ++j) {
      const int rr = min(ymin + j, nr - 1);
      acc = __builtin_fmaf(hbuf[(rr * 3 + c) * A.S + x], w[j], acc);
    }
    A.out[(((int64_t)b * 3 + c) * A.S + (oy0 + ry)) * A.S + x] = (acc - A.mean[c]) / A.stdv[c];
  }
}

int tok_ft_per_wg(int64_t token_tiles, int n_ft, bool multi) {
  if (multi) return 1;                                     // every chunk re-stages the token tile: nothing to share between feature tiles
  static const int forced = [] { const char* e = getenv("SIXDGS_TOK_FTPW"); return e ? atoi(e) : 0; }();      // (developer hook: tools/time_vit_gemms.py)
  if (forced > 0) return forced < n_ft ? forced : n_ft;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  const int64_t per = sdg_cdiv(token_tiles * n_ft, cus);    // as many workgroups as there are compute units (one resident per unit: 110 KB of LDS), no more
  return (int)(per < 1 ? 1 : (per > n_ft ? n_ft : per));
}

template <int AMODE, bool MULTI>
int launch_tok(TokArgs& A, int epi, hipStream_t s) {
  const int64_t tt = sdg_cdiv(A.m, kTT);
  const int n_ft = (A.n + kFT - 1) / kFT;
  A.ft_per_wg = tok_ft_per_wg(tt, n_ft, MULTI);
  if (tt > 0x7fffffffLL) return SIXDGS_E_BADARG;
  const dim3 g((unsigned)tt, (unsigned)sdg_cdiv(n_ft, A.ft_per_wg)), b(64 * kWaves);
  constexpr int PFW = 4;
  switch (epi) {
    case kEpiBias: hipLaunchKernelGGL((k_tok_gemm<AMODE, kEpiBias, MULTI, PFW>), g, b, 0, s, A); break;
    case kEpiGelu: hipLaunchKernelGGL((k_tok_gemm<AMODE, kEpiGelu, MULTI, PFW>), g, b, 0, s, A); break;
    case kEpiResid: hipLaunchKernelGGL((k_tok_gemm<AMODE, kEpiResid, MULTI, PFW>), g, b, 0, s, A); break;
    default: return SIXDGS_E_BADARG;
  }
  SDG_LAUNCH_OK();
  return 0;
}

}  // namespace

#ifdef SDG_TOK_PROF
extern "C" int sixdgs_debug_tok_prof(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tok_prof), sizeof(long long) * 16); }
#endif

extern "C" {

size_t sixdgs_tok_pack_bytes(int n, int k) { return n > 0 && k > 0 ? (size_t)n * (size_t)k * 4 : 0; }

int sixdgs_tok_pack(const float* w, int n, int k, int64_t ldw, void* planes, float* inv_scale, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(w && planes && inv_scale && n > 0 && (n % 128) == 0 && k > 0 && (k % kCK) == 0 && ldw >= k);
  SDG_CHECK_ARG(((uintptr_t)planes % 16) == 0);
  hipLaunchKernelGGL(k_tok_pack, dim3((unsigned)(n / 32)), dim3(256), 0, sdg_stream(stream), w, k, ldw, static_cast<char*>(planes), inv_scale);
  SDG_LAUNCH_OK();
  return 0;
}

int sixdgs_tok_attention(const float* qkv, int64_t ldq, int images, int tokens, int heads, float* y, int64_t ldy, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(images >= 0 && tokens > 0 && heads > 0);
  if (images == 0) return 0;
  if (tokens > kAtKT * 32) return SIXDGS_E_UNSUPPORTED;                              // one wave holds a query's logits against ALL keys in registers
  SDG_CHECK_ARG(qkv && y && ((uintptr_t)qkv % 16) == 0 && ((uintptr_t)y % 16) == 0 && (ldq % 4) == 0 && ldq >= 3 * (int64_t)heads * kAtD && (ldy % 4) == 0 &&
                ldy >= (int64_t)heads * kAtD && (int64_t)images * heads <= 0x7fffffffLL);
  AttnArgs A = {qkv, y, ldq, ldy, tokens, heads};
  // groups of 4 query tiles per (image, head): as many as the tokens need while every workgroup finds a compute unit at once (152 KB of LDS: one per unit);
  // beyond that one group fewer -- a wave then walks a second tile instead of a second round of workgroups staging K and V again
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  int groups = (int)sdg_cdiv(tokens, 128);
  while (groups > 1 && (int64_t)images * heads * groups > cus) --groups;
  hipLaunchKernelGGL(k_tok_attn, dim3((unsigned)(images * heads), (unsigned)groups), dim3(256), 0, sdg_stream(stream), A);
  SDG_LAUNCH_OK();
  return 0;
}

int sixdgs_im2col(const float* x, int64_t stride_b, int64_t stride_c, int64_t stride_y, int64_t stride_x, int batch, int channels, int height, int width, int k,
                  int taps_major, float* a, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(batch >= 0 && channels > 0 && k > 0 && height >= k && width >= k);
  if (batch == 0) return 0;
  SDG_CHECK_ARG(x && a);
  const int ho = height - k + 1, wo = width - k + 1;
  const int64_t total = (int64_t)batch * ho * wo * channels * k * k;
  SDG_CHECK_ARG((int64_t)channels * k * k <= 0x7fffffffLL && sdg_cdiv(total, 256) <= 0x7fffffffLL);
  hipStream_t s = sdg_stream(stream);
  const bool vec = taps_major && stride_c == 1 && (channels % 4) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)a % 16) == 0 && (stride_b % 4) == 0 &&
                   (stride_y % 4) == 0 && (stride_x % 4) == 0;
  if (vec)
    hipLaunchKernelGGL((k_im2col<1, 4>), dim3((unsigned)sdg_cdiv(total / 4, 256)), dim3(256), 0, s, x, stride_b, stride_c, stride_y, stride_x, channels, k, ho, wo, total, a);
  else if (taps_major)
    hipLaunchKernelGGL((k_im2col<1, 1>), dim3((unsigned)sdg_cdiv(total, 256)), dim3(256), 0, s, x, stride_b, stride_c, stride_y, stride_x, channels, k, ho, wo, total, a);
  else
    hipLaunchKernelGGL((k_im2col<0, 1>), dim3((unsigned)sdg_cdiv(total, 256)), dim3(256), 0, s, x, stride_b, stride_c, stride_y, stride_x, channels, k, ho, wo, total, a);
  SDG_LAUNCH_OK();
  return 0;
}

int sixdgs_u8_to_planar(const uint8_t* images, int batch, int64_t pixels, const float* table256, float* out, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(batch >= 0 && pixels > 0 && (pixels % 4) == 0);
  if (batch == 0) return 0;
  SDG_CHECK_ARG(images && table256 && out && ((uintptr_t)images % 4) == 0 && ((uintptr_t)out % 16) == 0);
  const int64_t total4 = (int64_t)batch * (pixels / 4);
  const int64_t grid = sdg_cdiv(total4, 256);
  hipLaunchKernelGGL(k_u8_to_planar, dim3((unsigned)(grid < 8192 ? grid : 8192)), dim3(256), 0, sdg_stream(stream), images, table256, pixels, total4, out);
  SDG_LAUNCH_OK();
  return 0;
}

// images [batch][height][width][3] uint8 -> out [batch][3][out_size][out_size] fp32: see include/sixdgs.h
int sixdgs_image_prep(const uint8_t* images, int batch, int height, int width, const float* table256, int resized_h, int resized_w, int crop_top, int crop_left,
                      int out_size, const float* mean3, const float* std3, float* out, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(batch >= 0 && height > 0 && width > 0 && resized_h > 0 && resized_w > 0 && out_size > 0 && crop_top >= 0 && crop_left >= 0 &&
                crop_top + out_size <= resized_h && crop_left + out_size <= resized_w && mean3 && std3);
  if (batch == 0) return 0;
  SDG_CHECK_ARG(images && table256 && out && ((uintptr_t)images % 4) == 0 && (int64_t)batch * height * width * 3 < (int64_t)1 << 40 && batch <= 65535);
  PrepArgs A;
  A.in = images; A.lut = table256; A.out = out;
  A.H = height; A.W = width; A.top = crop_top; A.left = crop_left; A.S = out_size;
  A.sh = (float)height / (float)resized_h;                 // area_pixel_compute_scale, align_corners = false, no scale factor given
  A.sw = (float)width / (float)resized_w;
  A.sup_h = A.sh >= 1.f ? 2.f * A.sh : 2.f;                // (interp_size 4) / 2 * scale
  A.sup_w = A.sw >= 1.f ? 2.f * A.sw : 2.f;
  const float sup = A.sup_h > A.sup_w ? A.sup_h : A.sup_w;
  const int taps = (int)ceilf(sup) * 2 + 1;                // the op's own window size
  A.kt = (taps + 3) & ~3;
  if (A.kt > 64) return SIXDGS_E_UNSUPPORTED;              // (scale > 15: not a query image of this pipeline)
  for (int c = 0; c < 3; ++c) { A.mean[c] = mean3[c]; A.stdv[c] = std3[c]; }
  A.last_dword = ((int64_t)batch * height * width * 3 - 1) >> 2;
  // band height: 8 output rows for batches that fill the chip anyway, 4 below (more workgroups, shorter each); fewer when the x-reduced rows do not fit 150 KB of LDS
  size_t lds = 0;
  for (A.rb = (int64_t)batch * sdg_cdiv(out_size, 8) >= 256 ? 8 : 4; A.rb >= 1; A.rb >>= 1) {
    A.nrmax = (int)((float)A.rb * A.sh + 2.f * A.sup_h) + 3;
    lds = sizeof(float) * (256 + (size_t)out_size * A.kt + 2 * (size_t)out_size + (size_t)A.rb * A.kt + 2 * (size_t)A.rb + (size_t)A.nrmax * 3 * out_size);
    if (lds <= 150 * 1024) break;
  }
  if (A.rb < 1) return SIXDGS_E_UNSUPPORTED;
  static size_t lds_set = 0;
  if (lds > lds_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_image_prep), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return (int)e;
    lds_set = 150 * 1024;
  }
  hipLaunchKernelGGL(k_image_prep, dim3((unsigned)sdg_cdiv(out_size, A.rb), (unsigned)batch), dim3(256), lds, sdg_stream(stream), A);
  SDG_LAUNCH_OK();
  return 0;
}

// y = epilogue( prologue(x) . w^T + bias ): see include/sixdgs.h
int sixdgs_tok_linear(const float* x, int64_t m, int k, int64_t ldx, int a_mode, const float* ln_weight, const float* ln_bias, float ln_eps,
                      const void* w_planes, const float* w_inv_scale, const float* bias, int n, int epilogue, const float* residual, int64_t ldr,
                      const float* gamma, float* y, int64_t ldy, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(m >= 0 && n > 0 && (n % 128) == 0 && k > 0 && (k % kCK) == 0);
  if (m == 0) return 0;
  SDG_CHECK_ARG(x && w_planes && w_inv_scale && y && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w_planes % 16) == 0 && ((uintptr_t)y % 16) == 0);
  SDG_CHECK_ARG(((uintptr_t)w_inv_scale % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0) && (!gamma || ((uintptr_t)gamma % 16) == 0));
  SDG_CHECK_ARG(a_mode == kAPlain || a_mode == kALayerNorm);
  SDG_CHECK_ARG(epilogue >= kEpiBias && epilogue <= kEpiResid);
  SDG_CHECK_ARG((ldx % 4) == 0 && ldx >= k);
  SDG_CHECK_ARG(a_mode != kALayerNorm || (k == kCK && ln_weight && ln_bias));                    // the LayerNorm prologue owns whole rows of 384
  SDG_CHECK_ARG(epilogue != kEpiResid || (residual && ldr >= n && (ldr % 4) == 0 && ((uintptr_t)residual % 16) == 0));
  SDG_CHECK_ARG(ldy >= n && (ldy % 4) == 0);
  TokArgs A = {x, static_cast<const char*>(w_planes), w_inv_scale, bias, ln_weight, ln_bias, residual, gamma, y, m, ldx, ldy, ldr, n, k, ln_eps, 1};
  hipStream_t s = sdg_stream(stream);
  if (a_mode == kALayerNorm) return launch_tok<kALayerNorm, false>(A, epilogue, s);
  if (k == kCK) return launch_tok<kAPlain, false>(A, epilogue, s);
  return launch_tok<kAPlain, true>(A, epilogue, s);
}

}  // extern "C"
