// pose.hip -- per-image pose assembly from the top-k rays (one workgroup of four wavefronts per image: the per-ray parts in parallel, every sum in index order on one lane):
// duplicate-origin filter, least-squares line intersection, exclude_negatives reweighting, viewing
// direction, rotation assembly, NaN / singular fall-backs, pose errors.
// replaces pose_estimation/test.py:157-198,216-218,268-288 with line_intersection.py:5-34,75-154 and
// error_computation.py:3-8.  Batched over images so the loop's per-image device syncs disappear.
#include "common.h"
#include "device_math.h"

using namespace sdg;

namespace {

constexpr int kMaxK = 256;
#ifdef SDG_POSE_PROF      // developer build (tools/build_variant.py poseprof -DSDG_POSE_PROF): 100 MHz stamps of image 0's sections
__device__ long long g_pose_prof[16];
#define POSE_T(K) if (blockIdx.x == 0 && threadIdx.x == 0) g_pose_prof[K] = wall_clock64();
#else
#define POSE_T(K)
#endif

struct PoseArgs {
  const float* rays_ori;
  const float* rays_dir;
  const int64_t* idx;   // [B,k]
  const float* val;     // [B,k]
  const float* up;      // [B,3]
  const float* gt;      // [B,4,4] or null
  float* c2w;           // [B,4,4]
  int* status;          // [B]
  float* w_final;       // [B,k] or null
  int* n_kept;          // [B] or null
  float* centre;        // [B,3] or null
  float* errors;        // [B,2] or null
  int64_t r;
  int k;
};

// One workgroup per image.  Everything that is independent per ray (or per coordinate, or per pair) runs on all threads; every SUM runs in index order on one
// thread, like the oracle's loops, so the results are the bits of the sequential form this replaces (round 6: one wavefront walked every loop and one lane
// computed every ray's terms -- 133 us per batch, 83 of them in the membership test; now ~35).
constexpr int kPoseThreads = 1024;
constexpr int kPoseParts = 3;          // the membership test's inner loops are cut into this many interleaved parts per coordinate
__global__ void __launch_bounds__(kPoseThreads) k_solve_pose(PoseArgs A) {
  __shared__ float so[kMaxK * 3];
  __shared__ float sd[kMaxK * 3];
  __shared__ float sw[kMaxK];
  __shared__ float test[kMaxK * 3];
  __shared__ int cnt[kMaxK];
  __shared__ int keep[kMaxK];
  __shared__ int cpos[kMaxK];                                        // position of a kept ray in the compacted arrays
  __shared__ int anyf[kMaxK * 3];
  __shared__ float co[kMaxK * 3];                                    // the kept rays, in index order
  __shared__ float cd[kMaxK * 3];
  __shared__ float cw[kMaxK];
  __shared__ __attribute__((aligned(16))) float term[kMaxK * 12];    // per kept ray: I - d d^T (9) and (I - d d^T) o (3); later d * w (3)
  __shared__ int s_nt, s_m, s_ok, s_valid[kPoseThreads / 64];
  __shared__ float s_sum, s_ctr[3];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, k = A.k;
  const int64_t* idx = A.idx + (int64_t)b * k;

  POSE_T(0)
  // ---- gather the selected rays (negative / out-of-range indices = padding of a short top-k) -------
  {
    const int i = tid;      // k <= kMaxK <= kPoseThreads
    bool ok = false;
    if (i < k) {
      const int64_t id = idx[i];
      ok = id >= 0 && id < A.r;
      for (int c = 0; c < 3; ++c) {
        so[3 * i + c] = ok ? A.rays_ori[3 * id + c] : NAN;
        sd[3 * i + c] = ok ? A.rays_dir[3 * id + c] : NAN;
      }
      sw[i] = ok ? A.val[(int64_t)b * k + i] : 0.f;
      cnt[i] = 0;
    }
    for (int p = tid; p < 3 * k; p += kPoseThreads) anyf[p] = 0;
    const int nv = __popcll(__ballot(ok));                          // padding entries sit at the tail (sorted top-k): the valid ones are a prefix
    if (lane == 0) s_valid[tid >> 6] = nv;
  }
  __syncthreads();
  int kv = 0;
  for (int w = 0; w < kPoseThreads / 64; ++w) kv += s_valid[w];
  POSE_T(1)

  // ---- a17: torch.unique(rows, counts) + isin(assume_unique=True).any(dim=1)  (test.py:157-162) ------
  for (int it = tid; it < kv * 4; it += kPoseThreads) {             // (ray, quarter of the other rays)
    const int i = it >> 2, part = it & 3;
    int c = 0;
    for (int j = part; j < kv; j += 4)
      c += (so[3 * i] == so[3 * j] && so[3 * i + 1] == so[3 * j + 1] && so[3 * i + 2] == so[3 * j + 2]) ? 1 : 0;
    if (c) atomicAdd(&cnt[i], c);
  }
  __syncthreads();
  if (tid < 64) {
    // the origins that occur once, in index order (a wave scan instead of one lane walking the list)
    int nt = 0;
    for (int i0 = 0; i0 < kv; i0 += 64) {
      const int i = i0 + lane;
      const bool one = i < kv && cnt[i] == 1;
      const unsigned long long m = __ballot(one);
      if (one) {
        const int pos = nt + __popcll(m & ((1ull << lane) - 1ull));
        test[3 * pos] = so[3 * i];
        test[3 * pos + 1] = so[3 * i + 1];
        test[3 * pos + 2] = so[3 * i + 2];
      }
      nt += __popcll(m);
    }
    if (lane == 0) s_nt = 3 * nt;
  }
  __syncthreads();
  POSE_T(2)
  const int nt = s_nt, ne = 3 * kv;
  // torch picks the sort-based algorithm unless the test set is small:
  // test.numel() < (int64)(10.0f * pow(elements.numel(), 0.145))
  const long long small_thr = (long long)(10.0 * pow((double)ne, 0.145));
  const bool sorting = !((long long)nt < small_thr);
  for (int it = tid; it < ne * kPoseParts; it += kPoseThreads) {
    // one thread per (COORDINATE of the flattened origins, part of the lists); no early exits: every comparison is independent, so the LDS reads pipeline
    const int p = it / kPoseParts, part = it - p * kPoseParts;
    const float v = so[p];
    bool any = false;
    for (int t = part; t < nt; t += kPoseParts) any = any || (test[t] == v);
    if (sorting)
      for (int q2 = p + 1 + part; q2 < ne; q2 += kPoseParts) any = any || (so[q2] == v);
    if (any) atomicOr(&anyf[p], 1);
  }
  __syncthreads();
  if (tid < kv) keep[tid] = (anyf[3 * tid] | anyf[3 * tid + 1] | anyf[3 * tid + 2]) != 0 ? 1 : 0;
  __syncthreads();
  POSE_T(3)

  // ---- the kept rays in index order (wave scan), their per-ray terms in parallel -----------------------------------------------------------------------------
  if (tid < 64) {
    int mm = 0;
    for (int i0 = 0; i0 < kv; i0 += 64) {
      const int i = i0 + lane;
      const bool kp = i < kv && keep[i] != 0;
      const unsigned long long mk = __ballot(kp);
      if (kp) {
        const int pos = mm + __popcll(mk & ((1ull << lane) - 1ull));
        for (int c = 0; c < 3; ++c) { co[3 * pos + c] = so[3 * i + c]; cd[3 * pos + c] = sd[3 * i + c]; }
        cw[pos] = sw[i];
        cpos[i] = pos;
      }
      mm += __popcll(mk);
    }
    if (lane == 0) s_m = mm;
  }
  __syncthreads();
  const int m = s_m;
  if (tid < m) {
    const int i = tid;
    const float d0 = cd[3 * i], d1 = cd[3 * i + 1], d2 = cd[3 * i + 2];
    const float o0 = co[3 * i], o1 = co[3 * i + 1], o2 = co[3 * i + 2];
    const float P[9] = {1.f - d0 * d0, 0.f - d0 * d1, 0.f - d0 * d2, 0.f - d1 * d0, 1.f - d1 * d1,
                        0.f - d1 * d2, 0.f - d2 * d0, 0.f - d2 * d1, 1.f - d2 * d2};
    float* t = term + 12 * i;
    for (int a = 0; a < 9; ++a) t[a] = P[a];
    t[9] = (P[0] * o0 + P[1] * o1) + P[2] * o2;
    t[10] = (P[3] * o0 + P[4] * o1) + P[5] * o2;
    t[11] = (P[6] * o0 + P[7] * o1) + P[8] * o2;
  }
  __syncthreads();
  POSE_T(4)
  // ---- sums in index order on thread 0, the per-ray steps between them on all threads ------------------------------------------------------------------------
  if (tid == 0) {
    float sum = 0.f;
    for (int i = 0; i < m; ++i) sum += cw[i];
    s_sum = sum;
    // a18: R = sum (I - d d^T), q = sum (I - d d^T) o   (unweighted, test.py:169-171)
    float Rm[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, qv[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < m; ++i) {
      const float* t = term + 12 * i;
      for (int a = 0; a < 9; ++a) Rm[a] += t[a];
      qv[0] += t[9];
      qv[1] += t[10];
      qv[2] += t[11];
    }
    float ctr[3];
    s_ok = solve_centre(Rm, qv, ctr) ? 1 : 0;
    s_ctr[0] = ctr[0]; s_ctr[1] = ctr[1]; s_ctr[2] = ctr[2];
  }
  __syncthreads();
  POSE_T(5)
  // exclude_negatives (line_intersection.py:29-34) and renormalise; the second solve of the reference
  // (test.py:177-179) is unweighted too and therefore returns the same centre
  if (tid < m) {
    const int i = tid;
    const float w0 = cw[i] / s_sum;
    const float v0 = s_ctr[0] - co[3 * i], v1 = s_ctr[1] - co[3 * i + 1], v2 = s_ctr[2] - co[3 * i + 2];
    const float dd = (v0 * cd[3 * i] + v1 * cd[3 * i + 1]) + v2 * cd[3 * i + 2];
    cw[i] = w0 * (dd > 0.f ? 1.f : 0.f);
  }
  __syncthreads();
  if (tid == 0) {
    float sum = 0.f;
    for (int i = 0; i < m; ++i) sum += cw[i];
    s_sum = sum;
  }
  __syncthreads();
  if (tid < m) {
    const int i = tid;
    const float w1 = cw[i] / s_sum;
    cw[i] = w1;
    term[12 * i] = cd[3 * i] * w1;
    term[12 * i + 1] = cd[3 * i + 1] * w1;
    term[12 * i + 2] = cd[3 * i + 2] * w1;
  }
  __syncthreads();
  if (A.w_final)
    for (int i = tid; i < k; i += kPoseThreads) A.w_final[(int64_t)b * k + i] = (i < kv && keep[i]) ? cw[cpos[i]] : 0.f;
  if (tid != 0) return;
  float wd[3] = {0.f, 0.f, 0.f};
  for (int i = 0; i < m; ++i) {
    wd[0] += term[12 * i];
    wd[1] += term[12 * i + 1];
    wd[2] += term[12 * i + 2];
  }
  POSE_T(6)
  const float ctr[3] = {s_ctr[0], s_ctr[1], s_ctr[2]};
  const bool ok = s_ok != 0;
  const float wn = sqrtf((wd[0] * wd[0] + wd[1] * wd[1]) + wd[2] * wd[2]);
  const V3 neg = v3(-(wd[0] / wn), -(wd[1] / wn), -(wd[2] / wn));
  const V3 upv = v3(A.up[3 * b], A.up[3 * b + 1], A.up[3 * b + 2]);
  float Rw[9];
  make_rotation_mat(neg, upv, Rw);
  int st = ok ? 0 : 4;
  if (det3(Rw) < 1.0e-7f) {  // test.py:194-196
    st |= 1;
    for (int i = 0; i < 9; ++i) Rw[i] = (i % 4 == 0) ? 1.f : 0.f;
  }
  float Ri[9];
  if (!inv3(Rw, Ri))
    for (int i = 0; i < 9; ++i) Ri[i] = NAN;
  float out[16];
  for (int i = 0; i < 16; ++i) out[i] = (i % 5 == 0) ? 1.f : 0.f;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) out[4 * r + c] = Ri[3 * r + c];
    out[4 * r + 3] = ctr[r];
  }
  bool nan = false;
  for (int i = 0; i < 16; ++i) nan = nan || !(out[i] == out[i]);
  if (nan) {  // test.py:216-218
    st |= 2;
    for (int i = 0; i < 16; ++i) out[i] = (i % 5 == 0) ? 1.f : 0.f;
  }
  for (int i = 0; i < 16; ++i) A.c2w[16 * b + i] = out[i];
  A.status[b] = st;
  if (A.centre)
    for (int c = 0; c < 3; ++c) A.centre[3 * b + c] = ctr[c];
  if (A.n_kept) A.n_kept[b] = m;
  if (A.errors) {
    float te = NAN, ae = NAN;
    if (A.gt) pose_errors(A.gt + 16 * b, out, &te, &ae);
    A.errors[2 * b] = te;
    A.errors[2 * b + 1] = ae;
  }
  POSE_T(7)
}

// ------------------------------------------------------------------------------------------------
// DistanceBasedScoreLoss target scores (distance_based_loss.py:147-283): raw target per ray, sum, rescale so that the
// targets sum to the token count.  HBM-bound: 24 B read + 4 B write per ray in the first sweep, 4 + 4 in the second.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_distance_target(const float* __restrict__ ori, const float* __restrict__ dir, int64_t r,
                                                         const float* __restrict__ pose, float* __restrict__ target,
                                                         double* __restrict__ partial) {
  __shared__ double sm[4];
  float P[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) P[i] = pose[i];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < r; i += (int64_t)gridDim.x * blockDim.x) {
    const float t = distance_target(P, v3(ori[3 * i], ori[3 * i + 1], ori[3 * i + 2]), v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
    target[i] = t;
    acc += (double)t;
  }
  acc = sdg_wave_sum(acc);
  if (sdg_lane() == 0) sm[sdg_wave()] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
__global__ void __launch_bounds__(256) k_distance_scale(float* __restrict__ target, int64_t r, const double* __restrict__ partial, int n_part,
                                                        float n_tokens, float* __restrict__ d_sum) {
  __shared__ float mult;
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < n_part; ++i) s += partial[i];              // fixed order: deterministic
    const float sum = (float)s;
    mult = (1.f / sum) * n_tokens;                                  // python scalar / tensor = reciprocal * scalar
    if (blockIdx.x == 0 && d_sum) *d_sum = sum;
  }
  __syncthreads();
  const float m = mult;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < r; i += (int64_t)gridDim.x * blockDim.x) target[i] = target[i] * m;
}

}  // namespace

#ifdef SDG_POSE_PROF
extern "C" int sixdgs_debug_pose_prof(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pose_prof), sizeof(long long) * 16); }
#endif

extern "C" size_t sixdgs_distance_target_workspace_bytes(int64_t r) {
  (void)r;
  return 1024 * sizeof(double);
}

extern "C" int sixdgs_distance_target(const float* rays_ori, const float* rays_dir, int64_t r, const float* d_pose, int n_tokens,
                                      float* target, float* d_sum, void* ws, size_t ws_bytes, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(r >= 0 && n_tokens >= 0);
  if (r == 0) return 0;
  SDG_CHECK_ARG(rays_ori && rays_dir && d_pose && target && ws && ((uintptr_t)ws % 8) == 0);
  if (ws_bytes < 1024 * sizeof(double)) return SIXDGS_E_WORKSPACE;
  const int nb = (int)(sdg_cdiv(r, 256) < 1024 ? sdg_cdiv(r, 256) : 1024);
  hipStream_t s = sdg_stream(stream);
  hipLaunchKernelGGL(k_distance_target, dim3((unsigned)nb), dim3(256), 0, s, rays_ori, rays_dir, r, d_pose, target, (double*)ws);
  hipLaunchKernelGGL(k_distance_scale, dim3((unsigned)nb), dim3(256), 0, s, target, r, (const double*)ws, nb, (float)n_tokens, d_sum);
  SDG_LAUNCH_OK();
  return 0;
}

extern "C" int sixdgs_solve_pose(const float* rays_ori, const float* rays_dir, int64_t r, const int64_t* idx, const float* val,
                                 int k, const float* up, const float* gt_c2w, int batch, float* c2w, int32_t* status,
                                 float* w_final, int32_t* n_kept, float* centre, float* errors, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(batch >= 0 && k >= 1 && k <= kMaxK && r >= 0);
  if (batch == 0) return 0;
  SDG_CHECK_ARG(rays_ori && rays_dir && idx && val && up && c2w && status);
  PoseArgs A = {rays_ori, rays_dir, idx, val, up, gt_c2w, c2w, status, w_final, n_kept, centre, errors, r, k};
  hipLaunchKernelGGL(k_solve_pose, dim3((unsigned)batch), dim3(kPoseThreads), 0, sdg_stream(stream), A);
  SDG_LAUNCH_OK();
  return 0;
}
