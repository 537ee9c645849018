// device_math.h -- per-thread fp32 math of the 6DGS pose path, usable from device AND host code
// (the host instantiation is what the CPU test-suite exercises through libsixdgs_hostcheck.so).
// Every routine cites the reference code whose arithmetic (operation order included) it follows.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define SDG_HD __host__ __device__ __forceinline__

namespace sdg {

constexpr float kTwoPi = 6.283185307179586f;
constexpr float kFourPi = 12.566370614359172f;
constexpr float kPi = 3.141592653589793f;
constexpr float kEps = 1.1920928955078125e-07f;  // torch.finfo(float32).eps

struct V3 {
  float x, y, z;
};
SDG_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
SDG_HD V3 cross(const V3& a, const V3& b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
SDG_HD float dot(const V3& a, const V3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
SDG_HD float norm(const V3& a) { return sqrtf(dot(a, a)); }

// ---- a1: scene/gaussian_model.py:129-134 + utils/general_utils.py:103-126 -------------------
// get_rotation = F.normalize(q) (eps 1e-12), build_rotation normalises again; q = (w,x,y,z).
SDG_HD void quat_to_rotmat(const float* q, float* R) {
  float n0 = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  float d0 = fmaxf(n0, 1e-12f);
  float a = q[0] / d0, b = q[1] / d0, c = q[2] / d0, d = q[3] / d0;
  float n1 = sqrtf(a * a + b * b + c * c + d * d);
  float r = a / n1, x = b / n1, y = c / n1, z = d / n1;
  R[0] = 1.f - 2.f * (y * y + z * z);
  R[1] = 2.f * (x * y - r * z);
  R[2] = 2.f * (x * z + r * y);
  R[3] = 2.f * (x * y + r * z);
  R[4] = 1.f - 2.f * (x * x + z * z);
  R[5] = 2.f * (y * z - r * x);
  R[6] = 2.f * (x * z - r * y);
  R[7] = 2.f * (y * z + r * x);
  R[8] = 1.f - 2.f * (x * x + y * y);
}

// ---- a2: pose_estimation/quadricell.py:86-97, 163-188 ------------------------------------------
SDG_HD float ellipse_perimeter(float b, float c) {
  float s = b + c;
  float dm = b - c;
  return kPi * (s + (3.f * (dm * dm)) / (10.f * s + sqrtf(b * b + (14.f * b) * c + c * c)));
}
SDG_HD float ellipsoid_surface(float a, float b, float c) {
  const float p = 1.6075f;
  float t = (powf(a * b, p) + powf(a * c, p) + powf(b * c, p)) / 3.f;
  return kFourPi * powf(t, (float)(1.0 / 1.6075));
}
// total_rings (int64 semantics of `.to(torch.long)`; NaN -> INT64_MIN) and the cell side.
SDG_HD long long total_rings(float a, float b, float c, float target_points, float* side_out) {
  float side = sqrtf(ellipsoid_surface(a, b, c) / target_points);
  float rb = floorf(ellipse_perimeter(a, b) / (2.f * side));
  float rc = floorf(ellipse_perimeter(a, c) / (2.f * side));
  *side_out = side;
  float h = (rb + rc) * 0.5f;
  if (!(h == h)) return (long long)0x8000000000000000ull;
  if (h >= 9.0e18f) return (long long)0x7fffffffffffffffll;
  return (long long)h;
}

// ---- a6: pose_estimation/quadricell.py:100-105, 191-319 ----------------------------------------
struct Ring {
  float bs, cs;   // scaled minor semi axes of the ring's ellipse
  float npts_f;   // floor(perimeter / side) as float (may be 0 / NaN)
  float dtheta;   // 2*pi / npts  (computed as reciprocal * 2*pi, Tensor.__rtruediv__)
  float z;        // ring centre along the local slicing axis (stored as local z)
};
SDG_HD Ring ring_params(float a, float b, float c, float side, float rings_f, float ring_f) {
  Ring r;
  float delta_ring = (2.f * a) / rings_f;
  float x = 0.5f * delta_ring + delta_ring * ring_f;
  float xa = x - a;
  float f = 1.f - (xa * xa) / (a * a);
  r.bs = sqrtf(f * (b * b));
  r.cs = sqrtf(f * (c * c));
  r.npts_f = floorf(ellipse_perimeter(r.bs, r.cs) / side);
  r.dtheta = (1.f / r.npts_f) * kTwoPi;
  r.z = x - a;
  return r;
}
// (capped: an ellipsoid that fails mask_degraded_ellipsoids could otherwise ask for ~1e9 cells per ring)
SDG_HD int ring_cells(const Ring& r) { return (r.npts_f >= 1.f) ? (int)fminf(r.npts_f, 1048576.f) : 0; }
// integrand sample j of the arc-length table: sqrt(bs sin^2 + cs cos^2) * dtheta, theta = j*dtheta
// (quirks kept: un-squared semi axes; the table step is the ring's own cell step).
SDG_HD float ring_table_increment(const Ring& r, int j) {
  float th = (float)j * r.dtheta;
  float sn = sinf(th), cn = cosf(th);
  return sqrtf(r.bs * (sn * sn) + r.cs * (cn * cn)) * r.dtheta;
}
// largest column c in [0, res-2] with table[1+c] < theta, else 0 (nonzero+coalesce(max) of
// quadricell.py:283-297).  The table is non-decreasing, so this is a binary search.
template <typename TableT>
SDG_HD int ring_table_pick(const TableT& table, int res, float theta) {
  int lo = 0, hi = res - 1;  // count of entries in table[1..res-1] that are < theta
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (table[1 + mid] < theta) lo = mid + 1; else hi = mid;
  }
  return lo > 0 ? lo - 1 : 0;
}
SDG_HD V3 ring_point(const Ring& r, float theta_prime) {
  return V3{r.bs * cosf(theta_prime), r.cs * sinf(theta_prime), r.z};
}

// ---- a7: pose_estimation/quadricell.py:322-386 (direction_mode="isocell") ------------------------
SDG_HD V3 rotate(const float* R, const V3& p) {
  return V3{(R[0] * p.x + R[1] * p.y) + R[2] * p.z, (R[3] * p.x + R[4] * p.y) + R[5] * p.z,
            (R[6] * p.x + R[7] * p.y) + R[8] * p.z};
}
// literal mask of mask_quadricell: (n[:, :, None] @ p[:, None, :])[..., 0, 0] = n.x * p.x
SDG_HD bool hemisphere_keep(float normal_x, const V3& p_world) { return normal_x * p_world.x > 0.f; }
SDG_HD V3 normalize_eps(const V3& v) {  // F.normalize(eps=1e-12)
  float d = fmaxf(norm(v), 1e-12f);
  return V3{v.x / d, v.y / d, v.z / d};
}

// ---- a10: utils/sh_utils.py:55-118 + sampling.py:116-124 ----------------------------------------
// one colour channel; s = the channel's coefficients with stride `st`; view dir = -ray dir.
SDG_HD float sh_channel(const float* s, int st, int deg, float x, float y, float z) {
  const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
  float r = C0 * s[0];
  if (deg > 0) {
    r = ((r - (C1 * y) * s[1 * st]) + (C1 * z) * s[2 * st]) - (C1 * x) * s[3 * st];
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      r = ((((r + (1.0925484305920792f * xy) * s[4 * st]) + (-1.0925484305920792f * yz) * s[5 * st]) +
            (0.31539156525252005f * ((2.f * zz - xx) - yy)) * s[6 * st]) +
           (-1.0925484305920792f * xz) * s[7 * st]) +
          (0.5462742152960396f * (xx - yy)) * s[8 * st];
      if (deg > 2) {
        r = ((((((r + ((-0.5900435899266435f * y) * (3.f * xx - yy)) * s[9 * st]) +
                 ((2.890611442640554f * xy) * z) * s[10 * st]) +
                ((-0.4570457994644658f * y) * ((4.f * zz - xx) - yy)) * s[11 * st]) +
               ((0.3731763325901154f * z) * ((2.f * zz - 3.f * xx) - 3.f * yy)) * s[12 * st]) +
              ((-0.4570457994644658f * x) * ((4.f * zz - xx) - yy)) * s[13 * st]) +
             ((1.445305721320277f * z) * (xx - yy)) * s[14 * st]) +
            ((-0.5900435899266435f * x) * (xx - 3.f * yy)) * s[15 * st];
      }
    }
  }
  return fmaxf(r + 0.5f, 0.f);
}

// ---- a5: pose_estimation/sym_eig_3x3.py ---------------------------------------------------------
SDG_HD float sign_nz(float x) { return x > 0.f ? 1.f : -1.f; }
// LU with partial pivoting, in place; returns the permutation sign (0 when a pivot is exactly 0)
SDG_HD float lu3(float* m, int* piv) {
  float sg = 1.f;
  for (int c = 0; c < 3; ++c) {
    int p = c;
    for (int r = c + 1; r < 3; ++r)
      if (fabsf(m[3 * r + c]) > fabsf(m[3 * p + c])) p = r;
    piv[c] = p;
    if (p != c) {
      for (int k = 0; k < 3; ++k) {
        float t = m[3 * c + k];
        m[3 * c + k] = m[3 * p + k];
        m[3 * p + k] = t;
      }
      sg = -sg;
    }
    float d = m[3 * c + c];
    if (d == 0.f) return 0.f;
    for (int r = c + 1; r < 3; ++r) {
      float f = m[3 * r + c] / d;
      m[3 * r + c] = f;
      for (int k = c + 1; k < 3; ++k) m[3 * r + k] -= f * m[3 * c + k];
    }
  }
  return sg;
}
SDG_HD float det3(const float* A) {
  float m[9];
  int piv[3];
  for (int i = 0; i < 9; ++i) m[i] = A[i];
  float sg = lu3(m, piv);
  return sg * m[0] * m[4] * m[8];
}
SDG_HD void lu3_solve(const float* lu, const int* piv, float* b) {
  for (int c = 0; c < 3; ++c)
    if (piv[c] != c) {
      float t = b[c];
      b[c] = b[piv[c]];
      b[piv[c]] = t;
    }
  b[1] -= lu[3] * b[0];
  b[2] -= lu[6] * b[0];
  b[2] -= lu[7] * b[1];
  b[2] /= lu[8];
  b[1] -= lu[5] * b[2];
  b[1] /= lu[4];
  b[0] -= lu[1] * b[1];
  b[0] -= lu[2] * b[2];
  b[0] /= lu[0];
}
SDG_HD bool inv3(const float* A, float* inv) {
  float m[9];
  int piv[3];
  for (int i = 0; i < 9; ++i) m[i] = A[i];
  if (lu3(m, piv) == 0.f) return false;
  for (int c = 0; c < 3; ++c) {
    float e[3] = {0.f, 0.f, 0.f};
    e[c] = 1.f;
    lu3_solve(m, piv, e);
    inv[c] = e[0];
    inv[3 + c] = e[1];
    inv[6 + c] = e[2];
  }
  return true;
}

SDG_HD V3 eig_ev0(const float* M) {  // _get_ev0, sym_eig_3x3.py:112-143
  V3 r0 = v3(M[0], M[1], M[2]), r1 = v3(M[3], M[4], M[5]), r2 = v3(M[6], M[7], M[8]);
  V3 cp[3] = {cross(r0, r1), cross(r1, r2), cross(r0, r2)};
  // regulariser uses the sign pattern of the FIRST cross product for all three
  V3 sg = v3(sign_nz(cp[0].x), sign_nz(cp[0].y), sign_nz(cp[0].z));
  float best_n = -1.f;
  V3 best = cp[0];
  for (int i = 0; i < 3; ++i) {
    V3 c = v3(cp[i].x + kEps * sg.x, cp[i].y + kEps * sg.y, cp[i].z + kEps * sg.z);
    float n = (c.x * c.x + c.y * c.y) + c.z * c.z;
    if (n > best_n) {
      best_n = n;
      best = c;
    }
  }
  float d = sqrtf(best_n);
  return v3(best.x / d, best.y / d, best.z / d);
}
SDG_HD void eig_uv(const V3& w, V3* u, V3* v) {  // _get_uv, :168-188
  float aw[3] = {fabsf(w.x), fabsf(w.y), fabsf(w.z)};
  int mi = 0;
  if (aw[1] < aw[mi]) mi = 1;
  if (aw[2] < aw[mi]) mi = 2;
  V3 t;
  if (mi == 0) t = v3(0.f, -w.z, w.y);
  else if (mi == 1) t = v3(-w.z, 0.f, w.x);
  else t = v3(-w.y, w.x, 0.f);
  *u = normalize_eps(t);
  *v = cross(w, *u);
}
SDG_HD V3 eig_ev1(const float* M, const V3& u, const V3& v) {  // _get_ev1, :191-231
  V3 Mu = rotate(M, u), Mv = rotate(M, v);
  float m00 = dot(u, Mu), m01 = dot(u, Mv), m10 = dot(v, Mu), m11 = dot(v, Mv);
  float acute = sign_nz(m00 * m10 + m01 * m11);
  float rs0 = m00 + acute * m10, rs1 = m01 + acute * m11;
  float sg = sign_nz(rs0);
  rs0 += kEps * sg;
  rs1 += kEps * sg;
  float a0 = rs1, a1 = -rs0;  // rowspace @ [[0,-1],[1,0]]
  float d = fmaxf(sqrtf(a0 * a0 + a1 * a1), 1e-12f);
  a0 /= d;
  a1 /= d;
  return v3(u.x * a0 + v.x * a1, u.y * a0 + v.y * a1, u.z * a0 + v.z * a1);
}
SDG_HD void eig_triple(const float* A, float alpha0, float alpha1, V3* e0, V3* e1, V3* e2) {  // :74-109
  float M[9];
  for (int i = 0; i < 9; ++i) M[i] = A[i];
  M[0] = A[0] - alpha0; M[4] = A[4] - alpha0; M[8] = A[8] - alpha0;
  *e0 = eig_ev0(M);
  V3 u, v;
  eig_uv(*e0, &u, &v);
  M[0] = A[0] - alpha1; M[4] = A[4] - alpha1; M[8] = A[8] - alpha1;
  *e1 = eig_ev1(M, u, v);
  *e2 = cross(*e0, *e1);
}
// vals[3] (alpha0 <= alpha1 <= alpha2), vecs (row-major, eigenvectors in COLUMNS) optional.
SDG_HD void sym_eig_3x3(const float* A, float* vals, float* vecs) {  // :246-307
  float q = ((A[0] + A[4]) + A[8]) / 3.f;
  float sq = 0.f;
  for (int i = 0; i < 9; ++i) sq += A[i] * A[i];
  float dsq = (A[0] * A[0] + A[4] * A[4]) + A[8] * A[8];
  float p1 = (sq - dsq) / 2.f;
  float d0 = A[0] - q, d1 = A[4] - q, d2 = A[8] - q;
  float p2 = ((d0 * d0 + d1 * d1) + d2 * d2) + 2.f * fmaxf(p1, kEps);
  float p = sqrtf(p2 / 6.f);
  float B[9];
  for (int i = 0; i < 9; ++i) B[i] = A[i];
  B[0] -= q; B[4] -= q; B[8] -= q;
  for (int i = 0; i < 9; ++i) B[i] = B[i] / p;
  float r = det3(B) / 2.f;
  r = fminf(fmaxf(r, -1.f + kEps), 1.f - kEps);
  float phi = acosf(r) / 3.f;
  float e1 = q + (2.f * p) * cosf(phi);
  float e2 = q + (2.f * p) * cosf(phi + 2.0943951023931953f);
  float e3 = (3.f * q - e1) - e2;
  float t = p1 / (6.f * kEps);
  float soft = expf(-(t * t));
  float g0 = A[0], g1 = A[4], g2 = A[8], s;
  if (g0 > g1) { s = g0; g0 = g1; g1 = s; }
  if (g1 > g2) { s = g1; g1 = g2; g2 = s; }
  if (g0 > g1) { s = g0; g0 = g1; g1 = s; }
  vals[0] = soft * g0 + (1.f - soft) * e2;
  vals[1] = soft * g1 + (1.f - soft) * e3;
  vals[2] = soft * g2 + (1.f - soft) * e1;
  if (!vecs) return;
  V3 a0, a1, a2;
  if ((vals[1] - vals[0]) > (vals[2] - vals[1])) {
    eig_triple(A, vals[0], vals[1], &a0, &a1, &a2);
  } else {  // (ev2, ev1, ev0) of the (alpha2, alpha1) construction
    V3 b0, b1, b2;
    eig_triple(A, vals[2], vals[1], &b0, &b1, &b2);
    a0 = b2; a1 = b1; a2 = b0;
  }
  vecs[0] = a0.x; vecs[1] = a1.x; vecs[2] = a2.x;
  vecs[3] = a0.y; vecs[4] = a1.y; vecs[5] = a2.y;
  vecs[6] = a0.z; vecs[7] = a1.z; vecs[8] = a2.z;
}

// ---- a4 tail: sampling.py:37-59,85-113: normal from k centred neighbours ---------------------------
// nb: [k][3] neighbour coordinates (any order); returns the unit normal.
// `at(j)` returns neighbour j as V3 (three sweeps over the neighbours: mean, scatter matrix, sign vote)
template <typename At>
SDG_HD V3 normal_from_neighbours_at(At at, int k) {
  float mx = 0.f, my = 0.f, mz = 0.f;
  for (int j = 0; j < k; ++j) { const V3 p = at(j); mx += p.x; my += p.y; mz += p.z; }
  mx /= (float)k; my /= (float)k; mz /= (float)k;
  float cov[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < k; ++j) {
    const V3 p = at(j);
    float dx = p.x - mx, dy = p.y - my, dz = p.z - mz;
    cov[0] += dx * dx; cov[1] += dx * dy; cov[2] += dx * dz;
    cov[3] += dy * dx; cov[4] += dy * dy; cov[5] += dy * dz;
    cov[6] += dz * dx; cov[7] += dz * dy; cov[8] += dz * dz;
  }
  float vals[3], vecs[9];
  sym_eig_3x3(cov, vals, vecs);
  V3 n = v3(vecs[0], vecs[3], vecs[6]);
  int npos = 0;
  for (int j = 0; j < k; ++j) {
    const V3 p = at(j);
    float pr = (n.x * (p.x - mx) + n.y * (p.y - my)) + n.z * (p.z - mz);
    npos += (pr > 0.f) ? 1 : 0;
  }
  float sgn = ((float)npos < 0.5f * (float)k) ? -1.f : 1.f;
  n = v3(sgn * n.x, sgn * n.y, sgn * n.z);
  float d = norm(n);
  return v3(n.x / d, n.y / d, n.z / d);
}
SDG_HD V3 normal_from_neighbours(const float* nb, int k) {
  return normal_from_neighbours_at([nb](int j) { return v3(nb[3 * j], nb[3 * j + 1], nb[3 * j + 2]); }, k);
}

// ---- a8: pose_estimation/isocell.py:6-84 (isrand=-1) --------------------------------------------------
SDG_HD int isocell_rings(int ray_target, int n0) { return (int)ceil(sqrt((double)ray_target / (double)n0)); }
// direction of cell j (0-based) of ring `ring` (1-based) among n rings
SDG_HD V3 isocell_dir(int n, int n0, int ring, int j) {
  double dR = 1.0 / (double)n;
  float R = (float)ring * (float)dR - (float)(dR / 2.0);
  float nc = (float)(n0 * (2 * ring - 1));
  float dth = (1.f / nc) * kTwoPi;
  float th = (float)j * dth + dth / 2.f;
  float X = R * cosf(th), Y = R * sinf(th);
  float zz = (1.f - X * X) - Y * Y;
  return v3(X, Y, zz > 0.f ? sqrtf(zz) : 0.f);
}

// ---- a9: pose_estimation/isocell.py:157-222 --------------------------------------------------------------
// Rodrigues matrix aligning z to normalize(n); NaN entries when n is (anti)parallel to z.
SDG_HD void isocell_rotation(const V3& n, float* Rm) {
  float nl = norm(n);
  V3 b = v3(n.x / nl, n.y / nl, n.z / nl);
  V3 a = v3(0.f, 0.f, 1.f);
  V3 v = cross(a, b);
  float c = dot(a, b);
  float s = norm(v);
  float km[9] = {0.f, -v.z, v.y, v.z, 0.f, -v.x, -v.y, v.x, 0.f};
  float f = (1.f - c) / (s * s);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float kk = (km[3 * i] * km[j] + km[3 * i + 1] * km[3 + j]) + km[3 * i + 2] * km[6 + j];
      Rm[3 * i + j] = ((i == j ? 1.f : 0.f) + km[3 * i + j]) + kk * f;
    }
}
SDG_HD V3 isocell_apply(const float* Rm, const V3& d) {  // d . R^T
  return rotate(Rm, d);
}

// ---- a12: pose_estimation/ray_preprocessor.py:3-9,36-44 ---------------------------------------------------
// element `col` (0..143) of the padded MLP input of one ray; p,d,c = origin, direction, colour.
SDG_HD float ray_input_element(const float* p, const float* d, const float* c, int col) {
  if (col < 9) return col < 3 ? p[col] : (col < 6 ? d[col - 3] : c[col - 6]);
  if (col >= 141) return 0.f;
  int o = col - 9;
  const float* src;
  int F;
  if (o < 48) { src = p; F = 8; }
  else if (o < 96) { src = d; F = 8; o -= 48; }
  else { src = c; F = 6; o -= 96; }
  int half = 3 * F;
  bool is_cos = o >= half;
  if (is_cos) o -= half;
  int comp = o / F, f = o - comp * F;
  float v = src[comp] * (float)(1 << f);
  return is_cos ? cosf(v) : sinf(v);
}

// ---- a18-a20: pose_estimation/line_intersection.py:5-34,75-154 ---------------------------------------------
SDG_HD void make_rotation_mat(const V3& direction, const V3& up, float* Rm) {
  V3 xa = cross(up, direction);
  float nx = norm(xa);
  xa = v3(xa.x / nx, xa.y / nx, xa.z / nx);
  V3 ya = cross(direction, xa);
  float ny = norm(ya);
  ya = v3(ya.x / ny, ya.y / ny, ya.z / ny);
  Rm[0] = xa.x; Rm[1] = xa.y; Rm[2] = xa.z;
  Rm[3] = ya.x; Rm[4] = ya.y; Rm[5] = ya.z;
  Rm[6] = direction.x; Rm[7] = direction.y; Rm[8] = direction.z;
}
// Solve (sum_i I - d d^T) c = sum_i (I - d d^T) o given the accumulated 3x3 / 3-vector.
// Returns false (centre = NaN) when det < 1e-7 (line_intersection.py:139-142).
SDG_HD bool solve_centre(const float* Rm, const float* q, float* centre) {
  float m[9];
  int piv[3];
  for (int i = 0; i < 9; ++i) m[i] = Rm[i];
  float sg = lu3(m, piv);
  float det = sg * m[0] * m[4] * m[8];
  if (sg == 0.f || det < 1.0e-7f || !(det == det)) {
    centre[0] = centre[1] = centre[2] = NAN;
    return false;
  }
  float b[3] = {q[0], q[1], q[2]};
  lu3_solve(m, piv, b);
  centre[0] = b[0]; centre[1] = b[1]; centre[2] = b[2];
  return true;
}
// error_computation.py:3-8
// distance_based_loss.py:5-71 (best_one_to_one_rays_selector, the part DistanceBasedScoreLoss.forward consumes): the raw target
// score of one ray for the ground-truth c2w `pose` (row-major 4x4): 1 - tanh(distance of the camera centre to the ray, clamped
// to the ray origin behind it), zeroed for origins behind the camera plane ((p/|p| + 1)/2 with p = (o - c) . z_cam; NaN at p = 0).
SDG_HD float distance_target(const float* pose, const V3& o, const V3& d) {
  const V3 c = v3(pose[3], pose[7], pose[11]);
  const V3 v = v3(c.x - o.x, c.y - o.y, c.z - o.z);
  const float t = dot(v, d);
  const V3 cl = t < 0.f ? o : v3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);
  const float dist = norm(v3(cl.x - c.x, cl.y - c.y, cl.z - c.z));
  const float target = 1.f - tanhf(dist);
  const V3 z = v3(pose[2], pose[6], pose[10]);
  const float p = dot(v3(o.x - c.x, o.y - c.y, o.z - c.z), z);
  const float sgn = ((p / fabsf(p)) + 1.f) / 2.f;
  return target * sgn;
}

SDG_HD void pose_errors(const float* gt, const float* pr, float* terr, float* aerr) {
  float d0 = gt[3] - pr[3], d1 = gt[7] - pr[7], d2 = gt[11] - pr[11];
  *terr = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
  float Re[9], Ri[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Re[3 * r + c] = pr[4 * r + c];
  if (!inv3(Re, Ri)) { *aerr = NAN; return; }
  float tr = 0.f;
  for (int r = 0; r < 3; ++r) tr += (gt[4 * r] * Ri[r] + gt[4 * r + 1] * Ri[3 + r]) + gt[4 * r + 2] * Ri[6 + r];
  float ca = fminf(fmaxf((tr - 1.f) / 2.f, -1.f), 1.f);
  *aerr = acosf(ca) * 57.29577951308232f;
}

}  // namespace sdg
