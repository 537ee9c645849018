// geometry.hip -- scene-side kernels of the 6DGS pose path (once per scene): validity mask, kNN
// normals, quadricell / iso-cell ray emission with fused SH colour.
//
// All of these are HBM- or latency-bound streaming kernels over the Gaussian arrays (232 B read
// per Gaussian, 36 B written per ray): coalesced reads of the per-attribute
// arrays, one workgroup per ellipsoid for the ragged quadricell emitter (arc-length table in LDS,
// order-preserving compaction with wave ballots), one thread per ray for the iso-cell emitter.
#include "common.h"
#include "device_math.h"

using namespace sdg;

namespace {

constexpr int kEmitThreads = 256;
constexpr int kEmitWaves = kEmitThreads / 64;
constexpr int kMaxTable = 1024;  // increments per ring table (reference: 999)

// ------------------------------------------------------------------------------------------------
// a2 mask_degraded_ellipsoids
// ------------------------------------------------------------------------------------------------
__global__ void k_mask_degraded(const float* __restrict__ log_scale, int64_t n, float target, uint8_t* __restrict__ mask) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = expf(log_scale[3 * i]), b = expf(log_scale[3 * i + 1]), c = expf(log_scale[3 * i + 2]);
  float side;
  long long rings = total_rings(a, b, c, target, &side);
  mask[i] = rings < (long long)target ? 1 : 0;
}

__global__ void k_sym_eig(const float* __restrict__ mats, int64_t n, float* __restrict__ vals, float* __restrict__ vecs) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float A[9], va[3], ve[9];
  for (int k = 0; k < 9; ++k) A[k] = mats[9 * i + k];
  sym_eig_3x3(A, va, vecs ? ve : nullptr);
  for (int k = 0; k < 3; ++k) vals[3 * i + k] = va[k];
  if (vecs)
    for (int k = 0; k < 9; ++k) vecs[9 * i + k] = ve[k];
}

// ------------------------------------------------------------------------------------------------
// a4 compute_normals: exact brute-force kNN (k <= 32), cloud tiles staged through LDS
// ------------------------------------------------------------------------------------------------
constexpr int kKnnTile = 1024;
constexpr int kKnnMaxK = 32;
__global__ void __launch_bounds__(256) k_normals_knn(const float* __restrict__ query, int64_t nq,
                                                      const float* __restrict__ cloud, int64_t E, int k,
                                                      float* __restrict__ normals, int64_t* __restrict__ knn) {
  __shared__ float tile[kKnnTile * 3];
  int64_t qi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool active = qi < nq;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (active) { qx = query[3 * qi]; qy = query[3 * qi + 1]; qz = query[3 * qi + 2]; }
  __shared__ float s_bd[kKnnMaxK][256];      // best list [slot][thread] in LDS (as per-thread arrays with run-time indices: 400 B of scratch)
  __shared__ int s_bi[kKnnMaxK][256];
  const int tl = threadIdx.x;
#define bd(J) s_bd[J][tl]
#define bi(J) s_bi[J][tl]
  int cnt = 0;
  float worst = INFINITY;
  for (int64_t t0 = 0; t0 < E; t0 += kKnnTile) {
    int tn = (int)((E - t0) < kKnnTile ? (E - t0) : kKnnTile);
    __syncthreads();
    for (int i = threadIdx.x; i < tn * 3; i += blockDim.x) tile[i] = cloud[3 * t0 + i];
    __syncthreads();
    if (!active) continue;
    for (int j = 0; j < tn; ++j) {
      float dx = qx - tile[3 * j], dy = qy - tile[3 * j + 1], dz = qz - tile[3 * j + 2];
      float d = (dx * dx + dy * dy) + dz * dz;
      if (cnt < k || d < worst) {
        int pos = cnt < k ? cnt : k - 1;
        while (pos > 0 && bd(pos - 1) > d) {
          bd(pos) = bd(pos - 1);
          bi(pos) = bi(pos - 1);
          --pos;
        }
        bd(pos) = d;
        bi(pos) = (int)(t0 + j);
        if (cnt < k) ++cnt;
        if (cnt == k) worst = bd(k - 1);
      }
    }
  }
  if (!active) return;
  V3 n = normal_from_neighbours_at([&](int j) {
    const float* c = cloud + 3 * (int64_t)bi(j);
    return v3(c[0], c[1], c[2]);
  }, cnt);
  normals[3 * qi] = n.x;
  normals[3 * qi + 1] = n.y;
  normals[3 * qi + 2] = n.z;
  if (knn)
    for (int j = 0; j < k; ++j) knn[qi * k + j] = j < cnt ? bi(j) : -1;
#undef bd
#undef bi
}

// ------------------------------------------------------------------------------------------------
// a4 on a uniform grid (the reference's cdist is O(E^2) and only ever sees E = 1000; full-scene emission
// needs E = 10^5..10^6).  EXACT and identical to the brute-force kernel above, neighbour order included: candidates are kept
// sorted by (squared distance, index); the cells around the query are visited in Chebyshev shells and the search stops once
// the k-th candidate is strictly closer than anything outside the visited cube can be.
//   build: bounding box (ordered-uint atomics) -> cell size -> counting sort of the points by cell (histogram, 3-kernel
//   exclusive scan, scatter of (x, y, z, index) as float4); query: one thread per query, in cell order when the queries are
//   the cloud itself, so a wavefront walks the same cells.
// ------------------------------------------------------------------------------------------------
struct GridParams {
  float ox, oy, oz, h, inv_h;
  int g;
};
__device__ __forceinline__ unsigned ordered_bits(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_float(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ void __launch_bounds__(256) k_grid_bbox(const float* __restrict__ cloud, int64_t e, unsigned* __restrict__ bbox) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = cloud[3 * i + a];
      lo[a] = fminf(lo[a], v);
      hi[a] = fmaxf(hi[a], v);
    }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float l = lo[a], h = hi[a];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      l = fminf(l, __shfl_xor(l, o, 64));
      h = fmaxf(h, __shfl_xor(h, o, 64));
    }
    if (sdg_lane() == 0) {
      atomicMin(bbox + a, ordered_bits(l));
      atomicMax(bbox + 3 + a, ordered_bits(h));
    }
  }
}
__global__ void k_grid_params(const unsigned* __restrict__ bbox, int g, GridParams* __restrict__ out) {
  float ext = 0.f;
  for (int a = 0; a < 3; ++a) ext = fmaxf(ext, ordered_float(bbox[3 + a]) - ordered_float(bbox[a]));
  if (!(ext > 0.f) || !(ext < INFINITY)) ext = 1.f;          // a single point / non-finite input: any grid works
  GridParams p;
  p.ox = ordered_float(bbox[0]);
  p.oy = ordered_float(bbox[1]);
  p.oz = ordered_float(bbox[2]);
  p.h = ext / (float)g * 1.0001f;
  p.inv_h = 1.f / p.h;
  p.g = g;
  *out = p;
}
__device__ __forceinline__ int grid_coord(float v, float o, float inv_h, int g) {
  const int c = (int)((v - o) * inv_h);
  return c < 0 ? 0 : (c >= g ? g - 1 : c);
}
__global__ void __launch_bounds__(256) k_grid_count(const float* __restrict__ cloud, int64_t e, const GridParams* __restrict__ gp,
                                                    int* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= e) return;
  const GridParams p = *gp;
  const int cx = grid_coord(cloud[3 * i], p.ox, p.inv_h, p.g), cy = grid_coord(cloud[3 * i + 1], p.oy, p.inv_h, p.g),
            cz = grid_coord(cloud[3 * i + 2], p.oz, p.inv_h, p.g);
  atomicAdd(count + ((int64_t)cz * p.g + cy) * p.g + cx, 1);
}
// exclusive scan of int32 counts in three launches: per-1024 block sums, scan of the sums (single block), local scans + offset
__global__ void __launch_bounds__(1024) k_scan_block_sums(const int* __restrict__ v, int64_t n, int* __restrict__ sums) {
  __shared__ int sm[17];
  const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  int tot;
  (void)sdg_block_exclusive_scan<int, 16>(i < n ? v[i] : 0, sm, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(1024) k_scan_sums(int* __restrict__ sums, int64_t n) {
  __shared__ int sm[17];
  int base = 0;
  for (int64_t c0 = 0; c0 < n; c0 += 1024) {
    const int64_t i = c0 + threadIdx.x;
    int tot;
    const int ex = sdg_block_exclusive_scan<int, 16>(i < n ? sums[i] : 0, sm, &tot);
    if (i < n) sums[i] = base + ex;
    base += tot;
  }
}
__global__ void __launch_bounds__(1024) k_scan_apply(const int* __restrict__ v, int64_t n, const int* __restrict__ sums,
                                                     int* __restrict__ start /*[n+1]*/) {
  __shared__ int sm[17];
  const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  int tot;
  const int ex = sdg_block_exclusive_scan<int, 16>(i < n ? v[i] : 0, sm, &tot);
  if (i < n) start[i] = sums[blockIdx.x] + ex;
  if (i == n - 1) start[n] = sums[blockIdx.x] + ex + v[i];
}
__global__ void __launch_bounds__(256) k_grid_fill(const float* __restrict__ cloud, int64_t e, const GridParams* __restrict__ gp,
                                                   const int* __restrict__ start, int* __restrict__ cursor, float4* __restrict__ sorted) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= e) return;
  const GridParams p = *gp;
  const float x = cloud[3 * i], y = cloud[3 * i + 1], z = cloud[3 * i + 2];
  const int64_t c = ((int64_t)grid_coord(z, p.oz, p.inv_h, p.g) * p.g + grid_coord(y, p.oy, p.inv_h, p.g)) * p.g +
                    grid_coord(x, p.ox, p.inv_h, p.g);
  const int pos = start[c] + atomicAdd(cursor + c, 1);
  sorted[pos] = make_float4(x, y, z, __int_as_float((int)i));
}

__global__ void __launch_bounds__(128) k_normals_knn_grid(const float* __restrict__ query, int64_t nq, const float* __restrict__ cloud,
                                                          const GridParams* __restrict__ gp, const int* __restrict__ start,
                                                          const float4* __restrict__ sorted, int by_cell, int k,
                                                          float* __restrict__ normals, int64_t* __restrict__ knn) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nq) return;
  const GridParams p = *gp;
  // queries == cloud: thread t takes the t-th point in cell order (neighbouring threads, neighbouring cells)
  int64_t qi = t;
  float qx, qy, qz;
  if (by_cell) {
    const float4 s = sorted[t];
    qx = s.x; qy = s.y; qz = s.z;
    qi = __float_as_int(s.w);
  } else {
    qx = query[3 * t]; qy = query[3 * t + 1]; qz = query[3 * t + 2];
  }
  const int g = p.g;
  const int cx = grid_coord(qx, p.ox, p.inv_h, g), cy = grid_coord(qy, p.oy, p.inv_h, g), cz = grid_coord(qz, p.oz, p.inv_h, g);
  // the best list lives in LDS, [slot][thread]: as per-thread arrays with run-time indices it was 400 B of scratch per thread (round 2)
  __shared__ float s_bd[kKnnMaxK][128];
  __shared__ int s_bi[kKnnMaxK][128];
  const int tl = threadIdx.x;
#define bd(J) s_bd[J][tl]
#define bi(J) s_bi[J][tl]
  int cnt = 0;
  float worst = INFINITY;
  int worst_i = 0x7fffffff;
  auto visit = [&](int64_t row, int x0, int x1) {       // cells [x0, x1] of grid row `row` are one contiguous point range
    const int b = start[row * g + x0], e2 = start[row * g + x1 + 1];
    for (int j = b; j < e2; ++j) {
      const float4 s = sorted[j];
      const float dx = qx - s.x, dy = qy - s.y, dz = qz - s.z;
      const float d = (dx * dx + dy * dy) + dz * dz;
      const int id = __float_as_int(s.w);
      if (cnt < k || d < worst || (d == worst && id < worst_i)) {
        int pos = cnt < k ? cnt : k - 1;
        while (pos > 0 && (bd(pos - 1) > d || (bd(pos - 1) == d && bi(pos - 1) > id))) {
          bd(pos) = bd(pos - 1);
          bi(pos) = bi(pos - 1);
          --pos;
        }
        bd(pos) = d;
        bi(pos) = id;
        if (cnt < k) ++cnt;
        if (cnt == k) { worst = bd(k - 1); worst_i = bi(k - 1); }
      }
    }
  };
  for (int s = 0; s < g; ++s) {
    const int z0 = max(cz - s, 0), z1 = min(cz + s, g - 1), y0 = max(cy - s, 0), y1 = min(cy + s, g - 1);
    const int x0 = max(cx - s, 0), x1 = min(cx + s, g - 1);
    for (int z = z0; z <= z1; ++z)
      for (int y = y0; y <= y1; ++y) {
        const int64_t row = (int64_t)z * g + y;
        if (z == cz - s || z == cz + s || y == cy - s || y == cy + s) visit(row, x0, x1);          // a face row of the shell
        else {
          if (cx - s >= 0) visit(row, cx - s, cx - s);
          if (s > 0 && cx + s < g) visit(row, cx + s, cx + s);
        }
      }
    if (cnt == k) {
      // everything not yet visited lies beyond a face of the visited cube that is still inside the grid
      float bound = INFINITY;
      if (cx - s > 0) bound = fminf(bound, qx - (p.ox + (float)(cx - s) * p.h));
      if (cx + s < g - 1) bound = fminf(bound, (p.ox + (float)(cx + s + 1) * p.h) - qx);
      if (cy - s > 0) bound = fminf(bound, qy - (p.oy + (float)(cy - s) * p.h));
      if (cy + s < g - 1) bound = fminf(bound, (p.oy + (float)(cy + s + 1) * p.h) - qy);
      if (cz - s > 0) bound = fminf(bound, qz - (p.oz + (float)(cz - s) * p.h));
      if (cz + s < g - 1) bound = fminf(bound, (p.oz + (float)(cz + s + 1) * p.h) - qz);
      if (bound == INFINITY) break;                         // the cube covers the grid
      bound -= 1e-3f * p.h;                                 // cell assignment and face positions are rounded
      if (bound > 0.f && worst < bound * bound) break;
    }
    if (cx - s <= 0 && cx + s >= g - 1 && cy - s <= 0 && cy + s >= g - 1 && cz - s <= 0 && cz + s >= g - 1) break;
  }
  // the neighbours' coordinates are read where they are used (three sweeps over <= 32 cached points) instead of staged in a 96-float array
  const V3 n = normal_from_neighbours_at([&](int j) {
    const float* c = cloud + 3 * (int64_t)bi(j);
    return v3(c[0], c[1], c[2]);
  }, cnt);
  normals[3 * qi] = n.x;
  normals[3 * qi + 1] = n.y;
  normals[3 * qi + 2] = n.z;
  if (knn)
    for (int j = 0; j < k; ++j) knn[qi * k + j] = j < cnt ? bi(j) : -1;
#undef bd
#undef bi
}

// ------------------------------------------------------------------------------------------------
// exclusive scan of int64 counts (single workgroup, chunked) + total
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_exclusive_scan(const int64_t* __restrict__ counts, int64_t n,
                                                          int64_t* __restrict__ offsets, int64_t* __restrict__ total) {
  __shared__ long long sm[17];
  long long base = 0;
  for (int64_t c0 = 0; c0 < n; c0 += 1024) {
    int64_t i = c0 + threadIdx.x;
    long long v = i < n ? (long long)counts[i] : 0;
    long long tot;
    long long ex = sdg_block_exclusive_scan<long long, 16>(v, sm, &tot);
    if (i < n) offsets[i] = base + ex;
    base += tot;
  }
  if (threadIdx.x == 0) total[0] = base;
}

// ------------------------------------------------------------------------------------------------
// a6/a7/a10 quadricell emitter: one workgroup (256 threads) per ellipsoid
// ------------------------------------------------------------------------------------------------
enum QcMode { QC_COUNT_CELLS = 0, QC_COUNT_RAYS = 1, QC_WRITE_CELLS = 2, QC_WRITE_RAYS = 3 };

struct QcArgs {
  const float* xyz;      // [N,3]
  const float* scale;    // [N,3]
  const float* rot;      // [N,4]
  const float* f_dc;     // [N,1,3]
  const float* f_rest;   // [N,ncoef-1,3]
  const int64_t* sel;    // [E] or null
  const float* normals;  // [E,3]
  const int64_t* offsets;  // [E] (write modes)
  int64_t* counts;       // [E] (count modes)
  int64_t* cell_counts;  // [E] (QC_COUNT_RAYS: also cells) or null
  float* out_a;          // points (cells) / ori (rays)
  float* out_dir;
  float* out_rgb;
  int64_t* out_id;       // ellipsoid id (cells) / Gaussian id (rays)
  int64_t E;
  int scale_is_log;
  int sh_degree, n_coef;
  float target;
  int table_res;
};

template <int MODE>
__global__ void __launch_bounds__(kEmitThreads) k_quadricell(QcArgs A) {
  __shared__ float table[kMaxTable + 1];
  __shared__ double sm_d[kEmitWaves + 1];
  __shared__ int sm_i[kEmitWaves + 1];
  __shared__ float sh[48];

  const int64_t e = blockIdx.x;
  const int tid = threadIdx.x;
  const int64_t g = A.sel ? A.sel[e] : e;
  float a = A.scale[3 * g], b = A.scale[3 * g + 1], c = A.scale[3 * g + 2];
  if (A.scale_is_log) { a = expf(a); b = expf(b); c = expf(c); }
  float side;
  long long rings = total_rings(a, b, c, A.target, &side);
  // Emission sets are expected to pass mask_degraded_ellipsoids (rings < 50 at 50 target cells, so
  // <= ~115 rings at 256).  A needle that does not would ask for up to 2^63 rings: emit nothing for it.
  if (rings > 4096) rings = 0;

  float R[9];
  float nx = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
  if (MODE == QC_COUNT_RAYS || MODE == QC_WRITE_RAYS) {
    float q[4] = {A.rot[4 * g], A.rot[4 * g + 1], A.rot[4 * g + 2], A.rot[4 * g + 3]};
    quat_to_rotmat(q, R);
    nx = A.normals[3 * e];
  }
  if (MODE == QC_WRITE_RAYS) {
    cx = A.xyz[3 * g]; cy = A.xyz[3 * g + 1]; cz = A.xyz[3 * g + 2];
    if (A.out_rgb != nullptr && tid < 3 * A.n_coef && tid < 48) {
      int k = tid / 3, ch = tid - 3 * k;
      sh[tid] = (k == 0) ? A.f_dc[3 * g + ch] : A.f_rest[(g * (A.n_coef - 1) + (k - 1)) * 3 + ch];
    }
  }
  long long n_cells = 0;  // cells before the mask (uniform)
  long long n_kept = 0;   // rays/cells emitted so far for this ellipsoid (uniform)
  const long long out_base = (MODE == QC_WRITE_CELLS || MODE == QC_WRITE_RAYS) ? (long long)A.offsets[e] : 0;
  const int res = A.table_res;
  const float rings_f = (float)rings;

  for (long long ring = 0; ring < rings; ++ring) {
    Ring rg = ring_params(a, b, c, side, rings_f, (float)ring);
    const int npts = ring_cells(rg);
    if (npts == 0) continue;
    n_cells += npts;
    if (MODE == QC_COUNT_CELLS) continue;

    // ---- arc-length table: table[0] = 0, table[j+1] = float(sum_{i<=j} inc_i) (double accumulate,
    //      as torch.cumsum does on CPU), then normalised to end at 2*pi --------------------------------
    const int j0 = tid * 4;
    float inc[4];
    double loc[4];
    double run = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int j = j0 + u;
      inc[u] = (j < res - 1) ? ring_table_increment(rg, j) : 0.f;
      run += (double)inc[u];
      loc[u] = run;
    }
    double tot;
    double ex = sdg_block_exclusive_scan<double, kEmitWaves>(run, sm_d, &tot);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int j = j0 + u;
      if (j < res - 1) table[j + 1] = (float)(ex + loc[u]);
    }
    if (tid == 0) table[0] = 0.f;
    __syncthreads();
    const float last = table[res - 1];
    __syncthreads();
    for (int j = tid; j < res; j += kEmitThreads) table[j] = kTwoPi * (table[j] / last);
    __syncthreads();

    // ---- cells of this ring ---------------------------------------------------------------------
    for (int c0 = 0; c0 < npts; c0 += kEmitThreads) {
      const int j = c0 + tid;
      const bool in = j < npts;
      V3 p = v3(0.f, 0.f, 0.f), pw = v3(0.f, 0.f, 0.f);
      bool keep = false;
      if (in) {
        float theta = (float)j * rg.dtheta;
        int pick = ring_table_pick(table, res, theta);
        p = ring_point(rg, table[pick]);
        if (MODE == QC_WRITE_CELLS) keep = true;
        else {
          pw = rotate(R, p);
          keep = hemisphere_keep(nx, pw);
        }
      }
      if (MODE == QC_COUNT_RAYS) {
        unsigned long long bal = __ballot(keep);
        if (sdg_lane() == 0) sm_i[sdg_wave()] = __popcll(bal);
        __syncthreads();
        int tot_i = 0;
        for (int w = 0; w < kEmitWaves; ++w) tot_i += sm_i[w];
        __syncthreads();
        n_kept += tot_i;
      } else {
        int tot_i;
        int pos = sdg_block_exclusive_scan<int, kEmitWaves>(keep ? 1 : 0, sm_i, &tot_i);
        if (keep) {
          long long o = out_base + n_kept + pos;
          if (MODE == QC_WRITE_CELLS) {
            A.out_a[3 * o] = p.x; A.out_a[3 * o + 1] = p.y; A.out_a[3 * o + 2] = p.z;
            A.out_id[o] = e;
          } else {
            V3 d = normalize_eps(pw);
            A.out_a[3 * o] = pw.x + cx; A.out_a[3 * o + 1] = pw.y + cy; A.out_a[3 * o + 2] = pw.z + cz;
            A.out_dir[3 * o] = d.x; A.out_dir[3 * o + 1] = d.y; A.out_dir[3 * o + 2] = d.z;
            A.out_id[o] = g;
            if (A.out_rgb) {
#pragma unroll
              for (int ch = 0; ch < 3; ++ch) A.out_rgb[3 * o + ch] = sh_channel(sh + ch, 3, A.sh_degree, -d.x, -d.y, -d.z);
            }
          }
        }
        n_kept += tot_i;
      }
    }
    __syncthreads();  // table reused by the next ring
  }
  if (tid == 0) {
    if (MODE == QC_COUNT_CELLS) A.counts[e] = n_cells;
    if (MODE == QC_COUNT_RAYS) {
      A.counts[e] = n_kept;
      if (A.cell_counts) A.cell_counts[e] = n_cells;
    }
  }
}

__global__ void k_sum_i64(const int64_t* __restrict__ v, int64_t n, int64_t* __restrict__ out) {
  __shared__ long long sm[5];
  long long acc = 0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) acc += v[i];
  acc = sdg_wave_sum(acc);
  if (sdg_lane() == 0) sm[sdg_wave()] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sm[w];
    out[0] = t;
  }
}

// ------------------------------------------------------------------------------------------------
// a8/a9 iso-cell directions, rotation, emitter
// ------------------------------------------------------------------------------------------------
__global__ void k_isocell_dirs(int n, int n0, int64_t total, float* __restrict__ dirs) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  // ring r (1-based) starts at n0*(r-1)^2
  int r = (int)floorf(sqrtf((float)(i / n0))) + 1;
  while ((int64_t)n0 * (r - 1) * (r - 1) > i) --r;
  while ((int64_t)n0 * r * r <= i) ++r;
  int j = (int)(i - (int64_t)n0 * (r - 1) * (r - 1));
  V3 d = isocell_dir(n, n0, r, j);
  dirs[3 * i] = d.x; dirs[3 * i + 1] = d.y; dirs[3 * i + 2] = d.z;
}

__global__ void k_rotate_isocell(const float* __restrict__ dirs, int64_t K, const float* __restrict__ normals, int64_t E,
                                 float* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E * K) return;
  int64_t e = i / K, k = i - e * K;
  float Rm[9];
  isocell_rotation(v3(normals[3 * e], normals[3 * e + 1], normals[3 * e + 2]), Rm);
  V3 d = isocell_apply(Rm, v3(dirs[3 * k], dirs[3 * k + 1], dirs[3 * k + 2]));
  out[3 * i] = d.x; out[3 * i + 1] = d.y; out[3 * i + 2] = d.z;
}

struct IsoArgs {
  const float *xyz, *scale, *rot, *f_dc, *f_rest, *normals, *dirs;
  const int64_t* sel;
  float *ori, *dir, *rgb;
  int64_t* src;
  int64_t E, K;
  int scale_is_log, sh_degree, n_coef;
};
// One workgroup per kIsoEPB ellipsoids.  Phase 1: one thread per ellipsoid builds what all of its K rays share -- rotation
// matrix of the quaternion, Rodrigues matrix of the normal, reciprocal... (the divisions stay per ray, as in the reference
// order of operations), centre -- and all threads gather the 48 SH coefficients, everything into LDS (the previous version
// recomputed all of it per ray from L2 and kept the SH block in scratch: 0.48 TB/s).  Phase 2: the block walks the E*K rays
// of its ellipsoids 256 at a time: per-ray arithmetic from LDS, results staged in LDS, then written as three contiguous
// runs (ori / dir / rgb: every wave instruction stores 256 consecutive bytes).  The kernel is bound by the 36 B per ray it writes.
constexpr int kIsoEPB = 16;
__global__ void __launch_bounds__(256) k_emit_isocell(IsoArgs A) {
  __shared__ float eR[kIsoEPB][9], eRm[kIsoEPB][9], eS[kIsoEPB][3], eC[kIsoEPB][3], eSh[kIsoEPB][48];
  __shared__ long long eG[kIsoEPB];
  __shared__ float stage[3][768];
  const int t = threadIdx.x;
  const int64_t e0 = (int64_t)blockIdx.x * kIsoEPB;
  const int ne = (int)min((int64_t)kIsoEPB, A.E - e0);
  if (t < ne) {
    const int64_t e = e0 + t, g = A.sel ? A.sel[e] : e;
    eG[t] = g;
    float s0 = A.scale[3 * g], s1 = A.scale[3 * g + 1], s2 = A.scale[3 * g + 2];
    if (A.scale_is_log) { s0 = expf(s0); s1 = expf(s1); s2 = expf(s2); }
    eS[t][0] = s0; eS[t][1] = s1; eS[t][2] = s2;
    const float q[4] = {A.rot[4 * g], A.rot[4 * g + 1], A.rot[4 * g + 2], A.rot[4 * g + 3]};
    float R[9], Rm[9];
    quat_to_rotmat(q, R);
    isocell_rotation(v3(A.normals[3 * e], A.normals[3 * e + 1], A.normals[3 * e + 2]), Rm);
#pragma unroll
    for (int c = 0; c < 9; ++c) { eR[t][c] = R[c]; eRm[t][c] = Rm[c]; }
    eC[t][0] = A.xyz[3 * g]; eC[t][1] = A.xyz[3 * g + 1]; eC[t][2] = A.xyz[3 * g + 2];
  }
  __syncthreads();
  if (A.rgb)
    for (int idx = t; idx < ne * 48; idx += 256) {
      const int el = idx / 48, c = idx - el * 48, kk = c / 3, ch = c - kk * 3;
      const int64_t g = eG[el];
      eSh[el][c] = kk == 0 ? A.f_dc[3 * g + ch] : (kk < A.n_coef ? A.f_rest[(g * (A.n_coef - 1) + (kk - 1)) * 3 + ch] : 0.f);
    }
  __syncthreads();
  const int64_t total = (int64_t)ne * A.K, base = e0 * A.K;
  for (int64_t p = 0; p < total; p += 256) {
    const int64_t r = p + t;
    if (r < total) {
      const int el = (int)(r / A.K);
      const int64_t k = r - (int64_t)el * A.K;
      const V3 d = isocell_apply(eRm[el], v3(A.dirs[3 * k], A.dirs[3 * k + 1], A.dirs[3 * k + 2]));
      // surface point of the ellipsoid along d: local dl = R^T d, t = 1/sqrt(sum (dl_i/s_i)^2)
      const float* R = eR[el];
      const float l0 = (R[0] * d.x + R[3] * d.y) + R[6] * d.z;
      const float l1 = (R[1] * d.x + R[4] * d.y) + R[7] * d.z;
      const float l2 = (R[2] * d.x + R[5] * d.y) + R[8] * d.z;
      const float u0 = l0 / eS[el][0], u1 = l1 / eS[el][1], u2 = l2 / eS[el][2];
      const float tt = 1.f / sqrtf((u0 * u0 + u1 * u1) + u2 * u2);
      stage[0][3 * t] = eC[el][0] + tt * d.x;
      stage[0][3 * t + 1] = eC[el][1] + tt * d.y;
      stage[0][3 * t + 2] = eC[el][2] + tt * d.z;
      stage[1][3 * t] = d.x; stage[1][3 * t + 1] = d.y; stage[1][3 * t + 2] = d.z;
      if (A.rgb)
#pragma unroll
        for (int c = 0; c < 3; ++c) stage[2][3 * t + c] = sh_channel(eSh[el] + c, 3, A.sh_degree, -d.x, -d.y, -d.z);
      if (A.src) A.src[base + r] = eG[el];
    }
    __syncthreads();
    const int n3 = 3 * (int)min((int64_t)256, total - p);
    const int64_t o = 3 * (base + p);
    for (int idx = t; idx < n3; idx += 256) {
      A.ori[o + idx] = stage[0][idx];
      A.dir[o + idx] = stage[1][idx];
      if (A.rgb) A.rgb[o + idx] = stage[2][idx];
    }
    __syncthreads();
  }
}

__global__ void k_eval_sh_color(const float* __restrict__ sh, int n_coef, const float* __restrict__ dirs, int64_t R, int deg,
                                float* __restrict__ rgb) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  float x = -dirs[3 * i], y = -dirs[3 * i + 1], z = -dirs[3 * i + 2];
  for (int ch = 0; ch < 3; ++ch) rgb[3 * i + ch] = sh_channel(sh + (3 * i + ch) * (int64_t)n_coef, 1, deg, x, y, z);
}

inline dim3 grid1d(int64_t n, int block) { return dim3((unsigned)sdg_cdiv(n, block)); }

}  // namespace

extern "C" {

int sixdgs_mask_degraded(const float* log_scale, int64_t n, int target_points, uint8_t* mask, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(n >= 0 && target_points > 0);
  if (n == 0) return 0;
  SDG_CHECK_ARG(log_scale && mask);
  hipLaunchKernelGGL(k_mask_degraded, grid1d(n, 256), dim3(256), 0, sdg_stream(stream), log_scale, n, (float)target_points, mask);
  SDG_LAUNCH_OK();
  return 0;
}

int sixdgs_sym_eig_3x3(const float* mats, int64_t n, float* vals, float* vecs, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(n >= 0);
  if (n == 0) return 0;
  SDG_CHECK_ARG(mats && vals);
  hipLaunchKernelGGL(k_sym_eig, grid1d(n, 128), dim3(128), 0, sdg_stream(stream), mats, n, vals, vecs);
  SDG_LAUNCH_OK();
  return 0;
}

int sixdgs_normals_knn(const float* query, int64_t nq, const float* cloud, int64_t e, int k, float* normals, int64_t* knn,
                       sixdgs_stream_t stream) {
  SDG_CHECK_ARG(nq >= 0 && e >= 1 && k >= 1 && k <= kKnnMaxK);
  if (nq == 0) return 0;
  SDG_CHECK_ARG(query && cloud && normals);
  hipLaunchKernelGGL(k_normals_knn, grid1d(nq, 256), dim3(256), 0, sdg_stream(stream), query, nq, cloud, e, k, normals, knn);
  SDG_LAUNCH_OK();
  return 0;
}

namespace {
struct GridPlan {
  int g;
  int64_t ncell;
  size_t off_bbox, off_params, off_count, off_start, off_sums, off_sorted, bytes;
};
GridPlan grid_plan(int64_t e) {
  GridPlan p;
  p.g = e < 100000 ? 64 : (e < 1500000 ? 128 : 256);       // ~2..30 points per occupied cell for the scenes of SURVEY 8
  p.ncell = (int64_t)p.g * p.g * p.g;
  size_t o = 0;
  p.off_bbox = o; o += 256;
  p.off_params = o; o += 256;
  p.off_count = o; o += sdg_align((size_t)p.ncell * 4);
  p.off_start = o; o += sdg_align((size_t)(p.ncell + 1) * 4);
  p.off_sums = o; o += sdg_align((size_t)(p.ncell / 1024 + 1) * 4);
  p.off_sorted = o; o += sdg_align((size_t)(e > 0 ? e : 1) * 16);
  p.bytes = o;
  return p;
}
}  // namespace

size_t sixdgs_normals_knn_grid_workspace_bytes(int64_t e) { return grid_plan(e).bytes; }

int sixdgs_normals_knn_grid(const float* query, int64_t nq, const float* cloud, int64_t e, int k, float* normals, int64_t* knn,
                            void* ws, size_t ws_bytes, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(nq >= 0 && e >= 1 && e < 2147483647LL && k >= 1 && k <= kKnnMaxK);
  if (nq == 0) return 0;
  SDG_CHECK_ARG(query && cloud && normals && ws && ((uintptr_t)ws % 256) == 0);
  const GridPlan p = grid_plan(e);
  if (ws_bytes < p.bytes) return SIXDGS_E_WORKSPACE;
  hipStream_t s = sdg_stream(stream);
  char* base = (char*)ws;
  unsigned* bbox = (unsigned*)(base + p.off_bbox);
  GridParams* gp = (GridParams*)(base + p.off_params);
  int* count = (int*)(base + p.off_count);
  int* start = (int*)(base + p.off_start);
  int* sums = (int*)(base + p.off_sums);
  float4* sorted = (float4*)(base + p.off_sorted);
  hipError_t er = hipMemsetAsync(bbox, 0xff, 3 * sizeof(unsigned), s);          // ordered encoding: min <- largest key
  if (er == hipSuccess) er = hipMemsetAsync(bbox + 3, 0, 3 * sizeof(unsigned), s);
  if (er == hipSuccess) er = hipMemsetAsync(count, 0, (size_t)p.ncell * 4, s);
  if (er != hipSuccess) return (int)er;
  const int64_t nb = sdg_cdiv(p.ncell, 1024);
  hipLaunchKernelGGL(k_grid_bbox, dim3((unsigned)(sdg_cdiv(e, 256) < 1024 ? sdg_cdiv(e, 256) : 1024)), dim3(256), 0, s, cloud, e, bbox);
  hipLaunchKernelGGL(k_grid_params, dim3(1), dim3(1), 0, s, bbox, p.g, gp);
  hipLaunchKernelGGL(k_grid_count, grid1d(e, 256), dim3(256), 0, s, cloud, e, gp, count);
  hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nb), dim3(1024), 0, s, count, p.ncell, sums);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, sums, nb);
  hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(1024), 0, s, count, p.ncell, sums, start);
  er = hipMemsetAsync(count, 0, (size_t)p.ncell * 4, s);                        // reused as the scatter cursor
  if (er != hipSuccess) return (int)er;
  hipLaunchKernelGGL(k_grid_fill, grid1d(e, 256), dim3(256), 0, s, cloud, e, gp, start, count, sorted);
  const int by_cell = (query == cloud && nq == e) ? 1 : 0;
  hipLaunchKernelGGL(k_normals_knn_grid, grid1d(nq, 128), dim3(128), 0, s, query, nq, cloud, gp, start, sorted, by_cell, k, normals, knn);
  SDG_LAUNCH_OK();
  return 0;
}

int sixdgs_quadricell_cell_counts(const float* scale, int64_t e, int target_points, int64_t* d_counts, int64_t* d_offsets,
                                  int64_t* d_total, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(e >= 0 && target_points > 0 && d_total);
  hipStream_t s = sdg_stream(stream);
  if (e > 0) {
    SDG_CHECK_ARG(scale && d_counts && d_offsets);
    QcArgs A = {};
    A.scale = scale; A.counts = d_counts; A.E = e; A.target = (float)target_points; A.table_res = 1000;
    hipLaunchKernelGGL(k_quadricell<QC_COUNT_CELLS>, dim3((unsigned)e), dim3(kEmitThreads), 0, s, A);
    SDG_LAUNCH_OK();
  }
  hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, s, d_counts, e, d_offsets, d_total);
  SDG_LAUNCH_OK();
  return 0;
}

int sixdgs_quadricell_centers(const float* scale, int64_t e, int target_points, int table_res, const int64_t* d_cell_offsets,
                              float* points, int64_t* ellipsoid_id, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(e >= 0 && target_points > 0 && table_res >= 2 && table_res <= kMaxTable + 1);
  if (e == 0) return 0;
  SDG_CHECK_ARG(scale && d_cell_offsets && points && ellipsoid_id);
  QcArgs A = {};
  A.scale = scale; A.offsets = d_cell_offsets; A.out_a = points; A.out_id = ellipsoid_id; A.E = e;
  A.target = (float)target_points; A.table_res = table_res;
  hipLaunchKernelGGL(k_quadricell<QC_WRITE_CELLS>, dim3((unsigned)e), dim3(kEmitThreads), 0, sdg_stream(stream), A);
  SDG_LAUNCH_OK();
  return 0;
}

int sixdgs_emit_quadricell_count(const float* xyz, const float* scale, int scale_is_log, const float* rot, const int64_t* sel,
                                 int64_t e, const float* normals, int target_points, int table_res, int64_t* d_counts,
                                 int64_t* d_offsets, int64_t* d_total, sixdgs_stream_t stream) {
  (void)xyz;
  SDG_CHECK_ARG(e >= 0 && target_points > 0 && table_res >= 2 && table_res <= kMaxTable + 1 && d_total);
  hipStream_t s = sdg_stream(stream);
  if (e > 0) {
    SDG_CHECK_ARG(scale && rot && normals && d_counts && d_offsets);
    QcArgs A = {};
    A.scale = scale; A.rot = rot; A.sel = sel; A.normals = normals; A.counts = d_counts;
    A.cell_counts = d_offsets;  // borrowed: cell counts land here first, summed into d_total[1] below
    A.E = e; A.scale_is_log = scale_is_log; A.target = (float)target_points; A.table_res = table_res;
    hipLaunchKernelGGL(k_quadricell<QC_COUNT_RAYS>, dim3((unsigned)e), dim3(kEmitThreads), 0, s, A);
    SDG_LAUNCH_OK();
    hipLaunchKernelGGL(k_sum_i64, dim3(1), dim3(256), 0, s, d_offsets, e, d_total + 1);
    SDG_LAUNCH_OK();
  } else {
    hipError_t er = hipMemsetAsync(d_total + 1, 0, sizeof(int64_t), s);
    if (er != hipSuccess) return (int)er;
  }
  hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, s, d_counts, e, d_offsets, d_total);
  SDG_LAUNCH_OK();
  return 0;
}

int sixdgs_emit_quadricell_write(const float* xyz, const float* scale, int scale_is_log, const float* rot, const float* f_dc,
                                 const float* f_rest, int sh_degree, int n_coef, const int64_t* sel, int64_t e,
                                 const float* normals, int target_points, int table_res, const int64_t* d_offsets, float* ori,
                                 float* dir, float* rgb, int64_t* src, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(e >= 0 && target_points > 0 && table_res >= 2 && table_res <= kMaxTable + 1);
  SDG_CHECK_ARG(sh_degree >= 0 && sh_degree <= 3 && n_coef >= (sh_degree + 1) * (sh_degree + 1) && n_coef <= 16);
  if (e == 0) return 0;
  SDG_CHECK_ARG(xyz && scale && rot && normals && d_offsets && ori && dir && src);
  SDG_CHECK_ARG(!rgb || (f_dc && (n_coef == 1 || f_rest)));
  QcArgs A = {};
  A.xyz = xyz; A.scale = scale; A.rot = rot; A.f_dc = f_dc; A.f_rest = f_rest; A.sel = sel; A.normals = normals;
  A.offsets = d_offsets; A.out_a = ori; A.out_dir = dir; A.out_rgb = rgb; A.out_id = src; A.E = e;
  A.scale_is_log = scale_is_log; A.sh_degree = sh_degree; A.n_coef = n_coef; A.target = (float)target_points;
  A.table_res = table_res;
  hipLaunchKernelGGL(k_quadricell<QC_WRITE_RAYS>, dim3((unsigned)e), dim3(kEmitThreads), 0, sdg_stream(stream), A);
  SDG_LAUNCH_OK();
  return 0;
}

int sixdgs_isocell_distribution(int ray_target, int n0, float* dirs, int64_t* h_count, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(ray_target > 0 && n0 > 0);
  int n = isocell_rings(ray_target, n0);
  int64_t total = (int64_t)n0 * n * n;
  if (h_count) *h_count = total;
  if (!dirs) return 0;
  hipLaunchKernelGGL(k_isocell_dirs, grid1d(total, 256), dim3(256), 0, sdg_stream(stream), n, n0, total, dirs);
  SDG_LAUNCH_OK();
  return 0;
}

int sixdgs_rotate_isocell(const float* dirs, int64_t k, const float* normals, int64_t e, float* out, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(k >= 0 && e >= 0);
  if (k * e == 0) return 0;
  SDG_CHECK_ARG(dirs && normals && out);
  hipLaunchKernelGGL(k_rotate_isocell, grid1d(e * k, 256), dim3(256), 0, sdg_stream(stream), dirs, k, normals, e, out);
  SDG_LAUNCH_OK();
  return 0;
}

int sixdgs_emit_isocell(const float* xyz, const float* scale, int scale_is_log, const float* rot, const float* f_dc,
                        const float* f_rest, int sh_degree, int n_coef, const int64_t* sel, int64_t e, const float* normals,
                        const float* dirs, int64_t k, float* ori, float* dir, float* rgb, int64_t* src, sixdgs_stream_t stream) {
  SDG_CHECK_ARG(e >= 0 && k >= 0);
  SDG_CHECK_ARG(sh_degree >= 0 && sh_degree <= 3 && n_coef >= (sh_degree + 1) * (sh_degree + 1) && n_coef <= 16);
  if (e * k == 0) return 0;
  SDG_CHECK_ARG(xyz && scale && rot && normals && dirs && ori && dir);
  SDG_CHECK_ARG(!rgb || (f_dc && (n_coef == 1 || f_rest)));
  IsoArgs A = {xyz, scale, rot, f_dc, f_rest, normals, dirs, sel, ori, dir, rgb, src, e, k, scale_is_log, sh_degree, n_coef};
  hipLaunchKernelGGL(k_emit_isocell, grid1d(e, kIsoEPB), dim3(256), 0, sdg_stream(stream), A);
  SDG_LAUNCH_OK();
  return 0;
}

int sixdgs_eval_sh_color(const float* sh, int n_coef, const float* dirs, int64_t r, int sh_degree, float* rgb,
                         sixdgs_stream_t stream) {
  SDG_CHECK_ARG(r >= 0 && sh_degree >= 0 && sh_degree <= 3 && n_coef >= (sh_degree + 1) * (sh_degree + 1));
  if (r == 0) return 0;
  SDG_CHECK_ARG(sh && dirs && rgb);
  hipLaunchKernelGGL(k_eval_sh_color, grid1d(r, 256), dim3(256), 0, sdg_stream(stream), sh, n_coef, dirs, r, sh_degree, rgb);
  SDG_LAUNCH_OK();
  return 0;
}

}  // extern "C"
