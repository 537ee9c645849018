// hostcheck.cpp -- HOST instantiation of device_math.h (the per-thread math of the HIP kernels) and of dense_layout.h (the index
// arithmetic of the ray-MLP chain's operand planes) behind
// a tiny C ABI, so the CPU test-suite can compare the product's arithmetic with the oracle and the
// golden vectors without a GPU.  Contains no kernels; never used by the product path.
#include <stdlib.h>

#include "device_math.h"
#include "sweep_plan.h"
#include "dense_layout.h"
using namespace sdg;

extern "C" {

void hc_mask_degraded(const float* scale, long long n, int P, unsigned char* mask) {
  for (long long i = 0; i < n; ++i) {
    float side;
    mask[i] = total_rings(scale[3 * i], scale[3 * i + 1], scale[3 * i + 2], (float)P, &side) < (long long)P;
  }
}

long long hc_quadricell_centers(const float* scale, long long E, int P, int res, float* points, long long* eid) {
  long long C = 0;
  float* table = (float*)malloc(sizeof(float) * res);
  for (long long e = 0; e < E; ++e) {
    float a = scale[3 * e], b = scale[3 * e + 1], c = scale[3 * e + 2], side;
    long long rings = total_rings(a, b, c, (float)P, &side);
    for (long long ring = 0; ring < rings; ++ring) {
      Ring rg = ring_params(a, b, c, side, (float)rings, (float)ring);
      int npts = ring_cells(rg);
      if (!npts) continue;
      if (points) {
        double acc = 0.0;
        table[0] = 0.f;
        for (int j = 0; j < res - 1; ++j) { acc += (double)ring_table_increment(rg, j); table[j + 1] = (float)acc; }
        float last = table[res - 1];
        for (int j = 0; j < res; ++j) table[j] = kTwoPi * (table[j] / last);
        for (int j = 0; j < npts; ++j) {
          int pick = ring_table_pick(table, res, (float)j * rg.dtheta);
          V3 p = ring_point(rg, table[pick]);
          points[3 * (C + j)] = p.x; points[3 * (C + j) + 1] = p.y; points[3 * (C + j) + 2] = p.z;
          eid[C + j] = e;
        }
      }
      C += npts;
    }
  }
  free(table);
  return C;
}

long long hc_emit_rays(const float* points, const long long* eid, long long C, const float* normals, const float* centers,
                       const float* rot4, float* ori, float* dir, long long* mid) {
  long long r = 0;
  for (long long i = 0; i < C; ++i) {
    long long e = eid[i];
    float R[9];
    quat_to_rotmat(rot4 + 4 * e, R);
    V3 pw = rotate(R, v3(points[3 * i], points[3 * i + 1], points[3 * i + 2]));
    if (!hemisphere_keep(normals[3 * e], pw)) continue;
    V3 d = normalize_eps(pw);
    ori[3 * r] = pw.x + centers[3 * e]; ori[3 * r + 1] = pw.y + centers[3 * e + 1]; ori[3 * r + 2] = pw.z + centers[3 * e + 2];
    dir[3 * r] = d.x; dir[3 * r + 1] = d.y; dir[3 * r + 2] = d.z;
    mid[r++] = e;
  }
  return r;
}

void hc_sym_eig(const float* mats, long long n, float* vals, float* vecs) {
  for (long long i = 0; i < n; ++i) sym_eig_3x3(mats + 9 * i, vals + 3 * i, vecs ? vecs + 9 * i : nullptr);
}

void hc_normals_from_knn(const float* cloud, const long long* knn, long long nq, int k, float* normals) {
  float* nb = (float*)malloc(sizeof(float) * 3 * k);
  for (long long q = 0; q < nq; ++q) {
    for (int j = 0; j < k; ++j)
      for (int c = 0; c < 3; ++c) nb[3 * j + c] = cloud[3 * knn[q * k + j] + c];
    V3 n = normal_from_neighbours(nb, k);
    normals[3 * q] = n.x; normals[3 * q + 1] = n.y; normals[3 * q + 2] = n.z;
  }
  free(nb);
}

long long hc_isocell_dirs(int target, int n0, float* dirs) {
  int n = isocell_rings(target, n0);
  long long o = 0;
  for (int ring = 1; ring <= n; ++ring)
    for (int j = 0; j < n0 * (2 * ring - 1); ++j, ++o)
      if (dirs) { V3 d = isocell_dir(n, n0, ring, j); dirs[3 * o] = d.x; dirs[3 * o + 1] = d.y; dirs[3 * o + 2] = d.z; }
  return o;
}

void hc_rotate_isocell(const float* dirs, long long K, const float* normals, long long E, float* out) {
  for (long long e = 0; e < E; ++e) {
    float Rm[9];
    isocell_rotation(v3(normals[3 * e], normals[3 * e + 1], normals[3 * e + 2]), Rm);
    for (long long k = 0; k < K; ++k) {
      V3 d = isocell_apply(Rm, v3(dirs[3 * k], dirs[3 * k + 1], dirs[3 * k + 2]));
      float* o = out + 3 * (e * K + k);
      o[0] = d.x; o[1] = d.y; o[2] = d.z;
    }
  }
}

void hc_sh_color(const float* sh, int ncoef, const float* dirs, long long R, int deg, float* rgb) {
  for (long long i = 0; i < R; ++i)
    for (int ch = 0; ch < 3; ++ch)
      rgb[3 * i + ch] = sh_channel(sh + (3 * i + ch) * ncoef, 1, deg, -dirs[3 * i], -dirs[3 * i + 1], -dirs[3 * i + 2]);
}

void hc_ray_input(const float* ori, const float* dir, const float* rgb, long long R, float* x) {
  for (long long i = 0; i < R; ++i)
    for (int c = 0; c < 144; ++c) x[144 * i + c] = ray_input_element(ori + 3 * i, dir + 3 * i, rgb + 3 * i, c);
}

void hc_make_rotation_mat(const float* d, const float* up, float* Rm) { make_rotation_mat(v3(d[0], d[1], d[2]), v3(up[0], up[1], up[2]), Rm); }

int hc_solve_centre(const float* Rm, const float* q, float* c) { return solve_centre(Rm, q, c) ? 1 : 0; }

void hc_pose_errors(const float* gt, const float* pr, float* out2) { pose_errors(gt, pr, out2, out2 + 1); }

void hc_distance_target(const float* pose, const float* ori, const float* dir, long long r, float* out) {
  for (long long i = 0; i < r; ++i)
    out[i] = distance_target(pose, v3(ori[3 * i], ori[3 * i + 1], ori[3 * i + 2]), v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
}

// ---- dense_layout.h: the ray-MLP chain's operand planes (tests/test_dense_layout.py walks a tag through them) ----
int hc_dl_row_perm(int m) { return dl::row_perm(m); }
int hc_dl_acc_row(int lane, int r) { return dl::acc_row(lane, r); }
int hc_dl_acc_feature(int permuted, int lane, int r) { return dl::acc_feature(permuted != 0, lane, r); }
long long hc_dl_plane_offset(int chunk_major, long long ray, int nslab, int slab, int plane, int c) {
  return chunk_major ? dl::cm_offset(ray, nslab, slab, plane, c) : dl::rm_offset(ray, nslab, slab, plane, c);
}
unsigned hc_dl_lds_offset(unsigned row, unsigned plane, unsigned c) { return dl::lds_offset(row, plane, c); }
unsigned hc_dl_frag_offset(unsigned row0, unsigned lane, unsigned ks, unsigned plane) { return dl::frag_offset(row0, lane, ks, plane); }
unsigned hc_dl_load_ray(int chunk_major, unsigned tid, unsigned jp) { return dl::load_ray(chunk_major != 0, tid, jp); }
unsigned hc_dl_load_chunk8(int chunk_major, unsigned tid) { return dl::load_chunk8(chunk_major != 0, tid); }
unsigned hc_dl_cm_src_offset(unsigned ray_in_tile, unsigned chunk8, unsigned granule_stride) {
  return dl::cm_src_offset(ray_in_tile, chunk8 * (unsigned)dl::kChunkRun, granule_stride);
}
int hc_sweep_launch_images(int left, int cap) { return sdg::sweep_launch_images(left, cap); }
// sweep_pack flattened: per launch hc_sweep_pack_ints() ints = n_slots, n_images, q_img[slots][4], q_lq[slots][4], img[], img_nq[], img_q[][4]
int hc_sweep_pack_ints() { return 2 + 2 * sdg::kSweepMaxSlots * sdg::kSlotQuarters + (2 + sdg::kSlotQuarters) * sdg::kSweepMaxLaunchImages; }
int hc_sweep_max_slots() { return sdg::kSweepMaxSlots; }
int hc_sweep_pack(const int* h_n_tok, int batch, int cap, int* out, int max_launches) {
  const std::vector<sdg::SweepSlots> plan = sdg::sweep_pack(h_n_tok, batch, cap);
  const int per = hc_sweep_pack_ints();
  for (size_t l = 0; l < plan.size() && (int)l < max_launches; ++l) {
    const sdg::SweepSlots& t = plan[l];
    int* o = out + (size_t)l * per;
    *o++ = t.n_slots;
    *o++ = t.n_images;
    for (int s = 0; s < sdg::kSweepMaxSlots; ++s)
      for (int w = 0; w < sdg::kSlotQuarters; ++w) *o++ = t.q_img[s][w];
    for (int s = 0; s < sdg::kSweepMaxSlots; ++s)
      for (int w = 0; w < sdg::kSlotQuarters; ++w) *o++ = t.q_lq[s][w];
    for (int i = 0; i < sdg::kSweepMaxLaunchImages; ++i) *o++ = t.img[i];
    for (int i = 0; i < sdg::kSweepMaxLaunchImages; ++i) *o++ = t.img_nq[i];
    for (int i = 0; i < sdg::kSweepMaxLaunchImages; ++i)
      for (int w = 0; w < sdg::kSlotQuarters; ++w) *o++ = t.img_q[i][w];
  }
  return (int)plan.size();
}
int hc_dl_const(int which) {
  const int v[] = {dl::kSlabB, dl::kPRow, dl::kGran, dl::kGranSlab, dl::kChunkRun};
  return which >= 0 && which < 5 ? v[which] : -1;
}

}  // extern "C"
