/*
 * sixdgs_oracle.c -- CPU restatement of the 6DGS pose-estimation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (6dgs_amd/) never does.
 *
 * Parity status: PINNED against golden vectors produced by importing the reference in the
 * build container (oracle/gen_golden.py -> tests/golden/g1..g7; checked by tests/test_oracle_*.py).
 * The reference ships no tests/golden vectors of its own.
 *
 * Each function restates, in plain C with the reference's fp32 operation order, the reference
 * function named in its comment (paths relative to the reference root).  PyTorch CPU semantics
 * that matter and were verified in the build container:
 *   - torch.cumsum(float32) on CPU accumulates in double and rounds each prefix to float;
 *   - python_scalar / tensor  ==  tensor.reciprocal() * (float)scalar   (Tensor.__rtruediv__);
 *   - tensor / python_scalar is a true fp32 division on CPU;
 *   - torch.isin(..., assume_unique=True) uses the sort-based path when the test set has at least
 *     (int64)(10 * numel(elements)^0.145) entries: element p is "in" iff an equal value exists
 *     later in cat(elements, test) -- i.e. the FIRST of two duplicated origins survives.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <immintrin.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI_D 3.141592653589793
static const float TWO_PI_F = (float)(2.0 * PI_D);
static const float FOUR_PI_F = (float)(4.0 * PI_D);
static const float PI_F = (float)PI_D;

int o_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void o_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------
 * a1  quaternion -> rotation matrix.  scene/gaussian_model.py:129-134 (get_rotation =
 * F.normalize(_rotation), eps 1e-12) followed by utils/general_utils.py:103-126 build_rotation
 * (which normalises again).  rot is (w,x,y,z).  R9 row-major.
 * ------------------------------------------------------------------------------------------ */
static void quat_to_rot(const float* q4, float* R) {
  float n0 = sqrtf(q4[0] * q4[0] + q4[1] * q4[1] + q4[2] * q4[2] + q4[3] * q4[3]);
  float d = n0 > 1e-12f ? n0 : 1e-12f;
  float a0 = q4[0] / d, a1 = q4[1] / d, a2 = q4[2] / d, a3 = q4[3] / d;
  float n1 = sqrtf(a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3);
  float r = a0 / n1, x = a1 / n1, y = a2 / n1, z = a3 / n1;
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
void o_build_rotation(const float* rot4, int64_t n, float* R9) {
  for (int64_t i = 0; i < n; ++i) quat_to_rot(rot4 + 4 * i, R9 + 9 * i);
}

/* ------------------------------------------------------------------------------------------
 * a2  pose_estimation/quadricell.py:86-97 ellipse_perimeter, :163-168 ellipsoid_surface,
 *     :171-188 mask_degraded_ellipsoids
 * ------------------------------------------------------------------------------------------ */
static float ellipse_perimeter(float b, float c) {
  float s = b + c;
  float dm = b - c;
  float num = 3.f * (dm * dm);
  float den = 10.f * s + sqrtf(b * b + (14.f * b) * c + c * c);
  return PI_F * (s + num / den);
}
static float ellipsoid_surface(float a, float b, float c) {
  float t = (powf(a * b, 1.6075f) + powf(a * c, 1.6075f) + powf(b * c, 1.6075f)) / 3.f;
  return FOUR_PI_F * powf(t, (float)(1.0 / 1.6075));
}
static int64_t total_rings_of(float a, float b, float c, float target_points, float* side_out) {
  float side = sqrtf(ellipsoid_surface(a, b, c) / target_points);
  float rb = floorf(ellipse_perimeter(a, b) / (2.f * side));
  float rc = floorf(ellipse_perimeter(a, c) / (2.f * side));
  if (side_out) *side_out = side;
  float h = (rb + rc) * 0.5f;
  if (!(h == h)) return INT64_MIN; /* NaN -> torch's .to(long) gives INT64_MIN */
  return (int64_t)h;
}
void o_total_rings(const float* scale3, int64_t n, int target_points, int64_t* rings) {
  for (int64_t i = 0; i < n; ++i)
    rings[i] = total_rings_of(scale3[3 * i], scale3[3 * i + 1], scale3[3 * i + 2], (float)target_points, 0);
}
void o_mask_degraded(const float* scale3, int64_t n, int target_points, uint8_t* mask) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i)
    mask[i] = total_rings_of(scale3[3 * i], scale3[3 * i + 1], scale3[3 * i + 2], (float)target_points, 0) <
              (int64_t)target_points;
}

/* ------------------------------------------------------------------------------------------
 * a6  pose_estimation/quadricell.py:191-319 compute_quadricell_centers (+ :100-160).
 * Two-call protocol: points == NULL -> returns the number of cells C.
 * Output order: ellipsoid -> ring -> cell (the order of the reference's repeat_interleave chain).
 * Literal quirks kept: the arc-length table of `res` entries is sampled at theta = j*dtheta_ring
 * (the ring's own cell step), its integrand uses the UN-squared scaled semi axes, and theta' is the
 * table entry at the largest column c with table[1+c] < theta_cell (0 if none).
 * ------------------------------------------------------------------------------------------ */
int64_t o_quadricell_centers(const float* scale3, int64_t E, int target_points, int res, float* points,
                             int64_t* ellipsoid_id) {
  int64_t C = 0;
  float* table = (float*)malloc(sizeof(float) * (size_t)res);
  for (int64_t e = 0; e < E; ++e) {
    float a = scale3[3 * e], b = scale3[3 * e + 1], c = scale3[3 * e + 2];
    float side;
    int64_t rings = total_rings_of(a, b, c, (float)target_points, &side);
    if (rings <= 0) continue;
    float rings_f = (float)rings;
    float delta_ring = (2.f * a) / rings_f;
    for (int64_t ring = 0; ring < rings; ++ring) {
      float ring_f = (float)ring;
      float x = 0.5f * delta_ring + delta_ring * ring_f;
      float xa = x - a;
      float f = 1.f - (xa * xa) / (a * a);
      float bs = sqrtf(f * (b * b));
      float cs = sqrtf(f * (c * c));
      float per = ellipse_perimeter(bs, cs);
      float npts_f = floorf(per / side);
      if (!(npts_f >= 1.f)) continue; /* 0 cells, or NaN */
      int64_t npts = (int64_t)npts_f;
      float dtheta = (1.f / npts_f) * TWO_PI_F; /* 2*pi / tensor == reciprocal * scalar */
      if (points) {
        /* arc-length table: [0, cumsum(ds * dtheta)] in double, each prefix rounded to float */
        double acc = 0.0;
        table[0] = 0.f;
        for (int j = 0; j < res - 1; ++j) {
          float th = (float)j * dtheta;
          float sn = sinf(th), cn = cosf(th);
          float ds = sqrtf(bs * (sn * sn) + cs * (cn * cn));
          acc += (double)(ds * dtheta);
          table[j + 1] = (float)acc;
        }
        float last = table[res - 1];
        for (int j = 0; j < res; ++j) table[j] = TWO_PI_F * (table[j] / last);
        float z = 0.5f * delta_ring + delta_ring * ring_f - a;
        for (int64_t j = 0; j < npts; ++j) {
          float cell_theta = (float)j * dtheta;
          int pick = 0;
          for (int col = 0; col < res - 1; ++col)
            if (table[1 + col] < cell_theta) pick = col;
          float tp = table[pick];
          points[3 * (C + j) + 0] = bs * cosf(tp);
          points[3 * (C + j) + 1] = cs * sinf(tp);
          points[3 * (C + j) + 2] = z;
          ellipsoid_id[C + j] = e;
        }
      }
      C += npts;
    }
  }
  free(table);
  return C;
}

/* ------------------------------------------------------------------------------------------
 * a7  pose_estimation/quadricell.py:348-386 mask_and_compute_rays(direction_mode="isocell")
 *     with :322-341.  NOTE the literal mask: `(n[:, :, None] @ p[:, None, :])[..., 0, 0]` is the
 *     outer product's (0,0) entry, i.e. n.x * p_world.x > 0 -- not the dot product.
 * Returns the number of rays kept; outputs sized for C.
 * ------------------------------------------------------------------------------------------ */
int64_t o_mask_and_compute_rays(const float* points, const int64_t* eid, int64_t C, const float* normals,
                                const float* centers, const float* R9, float* ori, float* dir, int64_t* mid) {
  int64_t r = 0;
  for (int64_t i = 0; i < C; ++i) {
    int64_t e = eid[i];
    const float* R = R9 + 9 * e;
    const float* p = points + 3 * i;
    float w[3];
    for (int k = 0; k < 3; ++k) w[k] = R[3 * k] * p[0] + R[3 * k + 1] * p[1] + R[3 * k + 2] * p[2];
    float proj = normals[3 * e] * w[0];
    if (!(proj > 0.f)) continue;
    float nrm = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    float d = nrm > 1e-12f ? nrm : 1e-12f;
    for (int k = 0; k < 3; ++k) {
      dir[3 * r + k] = w[k] / d;
      ori[3 * r + k] = w[k] + centers[3 * e + k];
    }
    mid[r] = e;
    ++r;
  }
  return r;
}

/* ------------------------------------------------------------------------------------------
 * a5  pose_estimation/sym_eig_3x3.py:246-307 sym_eig_3x3 (+ :38-243 helpers), eps = FLT_EPSILON.
 * A: row-major 3x3.  vals ascending-ish (eig2, eig3, eig1); vecs row-major with COLUMNS = eigvecs.
 * ------------------------------------------------------------------------------------------ */
static float sign_nz(float x) { return x > 0.f ? 1.f : -1.f; }
static void cross3(const float* a, const float* b, float* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
static float det3_lu(const float* A) {
  /* torch.det: LU with partial pivoting, product of the diagonal */
  float m[9];
  memcpy(m, A, sizeof(m));
  float det = 1.f;
  for (int c = 0; c < 3; ++c) {
    int p = c;
    for (int r = c + 1; r < 3; ++r)
      if (fabsf(m[3 * r + c]) > fabsf(m[3 * p + c])) p = r;
    if (p != c) {
      for (int k = 0; k < 3; ++k) {
        float t = m[3 * c + k];
        m[3 * c + k] = m[3 * p + k];
        m[3 * p + k] = t;
      }
      det = -det;
    }
    float piv = m[3 * c + c];
    det *= piv;
    if (piv == 0.f) return 0.f;
    for (int r = c + 1; r < 3; ++r) {
      float f = m[3 * r + c] / piv;
      for (int k = c; k < 3; ++k) m[3 * r + k] -= f * m[3 * c + k];
    }
  }
  return det;
}
static void get_ev0(const float* M, float eps, float* ev) { /* :112-143 */
  float cp[3][3];
  cross3(M + 0, M + 3, cp[0]);
  cross3(M + 3, M + 6, cp[1]);
  cross3(M + 0, M + 6, cp[2]);
  /* cross_products += eps * sign(cross_products[..., :1, :]) : the FIRST row's signs, broadcast */
  float sg[3] = {sign_nz(cp[0][0]), sign_nz(cp[0][1]), sign_nz(cp[0][2])};
  float nsq[3];
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 3; ++k) cp[r][k] += eps * sg[k];
    nsq[r] = cp[r][0] * cp[r][0] + cp[r][1] * cp[r][1] + cp[r][2] * cp[r][2];
  }
  int best = 0;
  for (int r = 1; r < 3; ++r)
    if (nsq[r] > nsq[best]) best = r;
  float n = sqrtf(nsq[best]);
  for (int k = 0; k < 3; ++k) ev[k] = cp[best][k] / n;
}
static void get_uv(const float* w, float* u, float* v) { /* :168-188 */
  int mi = 0;
  for (int k = 1; k < 3; ++k)
    if (fabsf(w[k]) < fabsf(w[mi])) mi = k;
  /* rotation by pi/2 about axis mi acting on the two remaining coordinates (rest ascending):
     out[rest0] = -w[rest1]; out[rest1] = w[rest0]; out[mi] = 0 */
  int r0 = (mi == 0) ? 1 : 0;
  int r1 = (mi == 2) ? 1 : 2;
  float t[3] = {0.f, 0.f, 0.f};
  t[r0] = 0.f * w[r0] + -1.f * w[r1];
  t[r1] = 1.f * w[r0] + 0.f * w[r1];
  float n = sqrtf(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
  float d = n > 1e-12f ? n : 1e-12f;
  for (int k = 0; k < 3; ++k) u[k] = t[k] / d;
  cross3(w, u, v);
}
static void get_ev1(const float* M, const float* u, const float* v, float eps, float* ev) { /* :191-231 */
  /* m = J^T M J with J = [u v] (3x2) */
  float Mu[3], Mv[3];
  for (int r = 0; r < 3; ++r) {
    Mu[r] = M[3 * r] * u[0] + M[3 * r + 1] * u[1] + M[3 * r + 2] * u[2];
    Mv[r] = M[3 * r] * v[0] + M[3 * r + 1] * v[1] + M[3 * r + 2] * v[2];
  }
  float m00 = u[0] * Mu[0] + u[1] * Mu[1] + u[2] * Mu[2];
  float m01 = u[0] * Mv[0] + u[1] * Mv[1] + u[2] * Mv[2];
  float m10 = v[0] * Mu[0] + v[1] * Mu[1] + v[2] * Mu[2];
  float m11 = v[0] * Mv[0] + v[1] * Mv[1] + v[2] * Mv[2];
  float acute = sign_nz(m00 * m10 + m01 * m11);
  float rs0 = m00 + acute * m10;
  float rs1 = m01 + acute * m11;
  float sg = sign_nz(rs0);
  rs0 += eps * sg;
  rs1 += eps * sg;
  /* rowspace @ [[0,-1],[1,0]] = (rs1, -rs0) ; normalize */
  float a0 = rs0 * 0.f + rs1 * 1.f;
  float a1 = rs0 * -1.f + rs1 * 0.f;
  float n = sqrtf(a0 * a0 + a1 * a1);
  float d = n > 1e-12f ? n : 1e-12f;
  a0 /= d;
  a1 /= d;
  for (int k = 0; k < 3; ++k) ev[k] = u[k] * a0 + v[k] * a1;
}
static void construct_eigenvecs(const float* A, float alpha0, float alpha1, float eps, float* e0, float* e1,
                                float* e2) { /* :74-109 */
  float M[9];
  memcpy(M, A, sizeof(M));
  M[0] = A[0] - alpha0 * 1.f; M[4] = A[4] - alpha0 * 1.f; M[8] = A[8] - alpha0 * 1.f;
  get_ev0(M, eps, e0);
  float u[3], v[3];
  get_uv(e0, u, v);
  memcpy(M, A, sizeof(M));
  M[0] = A[0] - alpha1 * 1.f; M[4] = A[4] - alpha1 * 1.f; M[8] = A[8] - alpha1 * 1.f;
  get_ev1(M, u, v, eps, e1);
  cross3(e0, e1, e2);
}
void o_sym_eig_3x3_one(const float* A, float* vals, float* vecs) {
  const float eps = 1.1920928955078125e-07f;
  float tr = (A[0] + A[4]) + A[8];
  float q = tr / 3.f;
  float sq = 0.f;
  for (int i = 0; i < 9; ++i) sq += A[i] * A[i];
  float dsq = (A[0] * A[0] + A[4] * A[4]) + A[8] * A[8];
  float p1 = (sq - dsq) / 2.f;
  float p1c = p1 > eps ? p1 : eps;
  float d0 = A[0] - q, d1 = A[4] - q, d2 = A[8] - q;
  float p2 = ((d0 * d0 + d1 * d1) + d2 * d2) + 2.f * p1c;
  float p = sqrtf(p2 / 6.f);
  float B[9];
  for (int i = 0; i < 9; ++i) B[i] = A[i];
  B[0] -= q; B[4] -= q; B[8] -= q;
  for (int i = 0; i < 9; ++i) B[i] = B[i] / p;
  float r = det3_lu(B) / 2.f;
  float lo = -1.f + eps, hi = 1.f - eps;
  r = r < lo ? lo : (r > hi ? hi : r);
  float phi = acosf(r) / 3.f;
  float eig1 = q + (2.f * p) * cosf(phi);
  float eig2 = q + (2.f * p) * cosf(phi + (float)(2.0 * PI_D / 3.0));
  float eig3 = (3.f * q - eig1) - eig2;
  float ev[3] = {eig2, eig3, eig1};
  float t = p1 / (6.f * eps);
  float soft = expf(-(t * t));
  float dg[3] = {A[0], A[4], A[8]};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2 - i; ++j)
      if (dg[j] > dg[j + 1]) { float s = dg[j]; dg[j] = dg[j + 1]; dg[j + 1] = s; }
  for (int k = 0; k < 3; ++k) vals[k] = soft * dg[k] + (1.f - soft) * ev[k];
  if (!vecs) return;
  /* _construct_eigenvecs_set :38-71 */
  float a0[3], a1[3], a2[3], b0[3], b1[3], b2[3];
  construct_eigenvecs(A, vals[0], vals[1], eps, a0, a1, a2);
  construct_eigenvecs(A, vals[2], vals[1], eps, b0, b1, b2);
  int use01 = (vals[1] - vals[0]) > (vals[2] - vals[1]);
  for (int k = 0; k < 3; ++k) {
    if (use01) {
      vecs[3 * k + 0] = a0[k]; vecs[3 * k + 1] = a1[k]; vecs[3 * k + 2] = a2[k];
    } else { /* stacked reversed: (ev2, ev1, ev0) of the (alpha2, alpha1) construction */
      vecs[3 * k + 0] = b2[k]; vecs[3 * k + 1] = b1[k]; vecs[3 * k + 2] = b0[k];
    }
  }
}
void o_sym_eig_3x3(const float* A, int64_t n, float* vals, float* vecs) {
  for (int64_t i = 0; i < n; ++i) o_sym_eig_3x3_one(A + 9 * i, vals + 3 * i, vecs ? vecs + 9 * i : 0);
}

/* ------------------------------------------------------------------------------------------
 * a4  pose_estimation/sampling.py:62-113 compute_normals (k nearest incl. self, centred scatter
 *     X^T X, smallest-eigenvalue eigenvector, sign by majority vote :37-59, normalised).
 * Neighbour selection: squared distances computed directly; ties -> lowest index.  (The
 * reference's cdist uses the |x|^2+|y|^2-2xy expansion; the neighbour SET is identical except at
 * ulp-level ties of the k-th distance -- knn_out lets tests compare the sets.)
 * ------------------------------------------------------------------------------------------ */
static float disambiguate_sign(const float* df, int k, const float* vec) { /* returns +1 or -1 */
  int npos = 0;
  for (int j = 0; j < k; ++j) {
    float proj = (vec[0] * df[3 * j] + vec[1] * df[3 * j + 1]) + vec[2] * df[3 * j + 2];
    if (proj > 0.f) ++npos;
  }
  float flip = ((float)npos < 0.5f * (float)k) ? 1.f : 0.f;
  return 1.f - 2.f * flip;
}
void o_compute_normals(const float* chunk, int64_t nq, const float* cloud, int64_t E, int k, float* normals,
                       int64_t* knn_out) {
#pragma omp parallel
  {
    float* bd = (float*)malloc(sizeof(float) * (size_t)k);
    int64_t* bi = (int64_t*)malloc(sizeof(int64_t) * (size_t)k);
    float* df = (float*)malloc(sizeof(float) * 3 * (size_t)k);
#pragma omp for schedule(static)
    for (int64_t qi = 0; qi < nq; ++qi) {
      const float* q = chunk + 3 * qi;
      int cnt = 0;
      for (int64_t j = 0; j < E; ++j) {
        float dx = q[0] - cloud[3 * j], dy = q[1] - cloud[3 * j + 1], dz = q[2] - cloud[3 * j + 2];
        float d = (dx * dx + dy * dy) + dz * dz;
        if (cnt < k || d < bd[cnt - 1]) {
          int pos = cnt < k ? cnt : k - 1;
          while (pos > 0 && bd[pos - 1] > d) {
            bd[pos] = bd[pos - 1];
            bi[pos] = bi[pos - 1];
            --pos;
          }
          bd[pos] = d;
          bi[pos] = j;
          if (cnt < k) ++cnt;
        }
      }
      float mean[3] = {0.f, 0.f, 0.f};
      for (int j = 0; j < cnt; ++j)
        for (int c = 0; c < 3; ++c) mean[c] += cloud[3 * bi[j] + c];
      for (int c = 0; c < 3; ++c) mean[c] /= (float)cnt;
      float cov[9] = {0};
      for (int j = 0; j < cnt; ++j) {
        for (int c = 0; c < 3; ++c) df[3 * j + c] = cloud[3 * bi[j] + c] - mean[c];
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) cov[3 * a + b] += df[3 * j + a] * df[3 * j + b];
      }
      float vals[3], vecs[9];
      o_sym_eig_3x3_one(cov, vals, vecs);
      float n[3] = {vecs[0], vecs[3], vecs[6]};
      float s = disambiguate_sign(df, cnt, n);
      for (int c = 0; c < 3; ++c) n[c] = s * n[c];
      float nn = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
      for (int c = 0; c < 3; ++c) normals[3 * qi + c] = n[c] / nn;
      if (knn_out)
        for (int j = 0; j < k; ++j) knn_out[qi * k + j] = j < cnt ? bi[j] : -1;
    }
    free(bd);
    free(bi);
    free(df);
  }
}

/* ------------------------------------------------------------------------------------------
 * a8  pose_estimation/isocell.py:6-84 isocell_distribution (isrand = -1 branch :62-64)
 * returns Ntot = N0*n^2; dirs may be NULL for the size query.
 * ------------------------------------------------------------------------------------------ */
int64_t o_isocell_distribution(int ray_target, int N0, float* dirs) {
  int n = (int)ceil(sqrt((double)ray_target / (double)N0));
  int64_t Ntot = (int64_t)N0 * n * n;
  if (!dirs) return Ntot;
  double dR = 1.0 / n;
  float dRf = (float)dR, halfdR = (float)(dR / 2.0);
  int64_t o = 0;
  for (int ring = 1; ring <= n; ++ring) {
    int64_t nc = (int64_t)N0 * (2 * ring - 1);
    float R = (float)ring * dRf - halfdR;
    float dth = (1.f / (float)nc) * TWO_PI_F;
    for (int64_t j = 0; j < nc; ++j, ++o) {
      float th0 = (float)j * dth; /* 0 + ring_cell_ids * dth */
      float th = th0 + dth / 2.f;
      float X = R * cosf(th), Y = R * sinf(th);
      float zz = (1.f - X * X) - Y * Y; /* real part of the complex64 expression */
      float Z = zz > 0.f ? sqrtf(zz) : 0.f;
      dirs[3 * o] = X;
      dirs[3 * o + 1] = Y;
      dirs[3 * o + 2] = Z;
    }
  }
  return Ntot;
}

/* ------------------------------------------------------------------------------------------
 * a9  pose_estimation/isocell.py:171-222 rotate_isocell (Rodrigues alignment of z to the unit
 *     normal; NaN when the normal is parallel to z, as the reference).  out [E][K][3]
 * ------------------------------------------------------------------------------------------ */
void o_rotate_isocell(const float* dirs, int64_t K, const float* normals, int64_t E, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t e = 0; e < E; ++e) {
    const float* nn = normals + 3 * e;
    float nl = sqrtf((nn[0] * nn[0] + nn[1] * nn[1]) + nn[2] * nn[2]);
    float b[3] = {nn[0] / nl, nn[1] / nl, nn[2] / nl};
    float a[3] = {0.f, 0.f, 1.f};
    float v[3];
    cross3(a, b, v);
    float c = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
    float s = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    float km[9] = {0.f, -v[2], v[1], v[2], 0.f, -v[0], -v[1], v[0], 0.f};
    float kk[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) kk[3 * i + j] = (km[3 * i] * km[j] + km[3 * i + 1] * km[3 + j]) + km[3 * i + 2] * km[6 + j];
    float f = (1.f - c) / (s * s);
    float Rm[9];
    for (int i = 0; i < 9; ++i) Rm[i] = ((i % 4 == 0 ? 1.f : 0.f) + km[i]) + kk[i] * f;
    for (int64_t k = 0; k < K; ++k) {
      const float* d = dirs + 3 * k;
      float* o = out + 3 * (e * K + k);
      for (int j = 0; j < 3; ++j) o[j] = (d[0] * Rm[3 * j] + d[1] * Rm[3 * j + 1]) + d[2] * Rm[3 * j + 2];
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * a10 pose_estimation/sampling.py:116-124 evaluate_viewdirs_color -> utils/sh_utils.py:55-118
 *     eval_sh(deg, sh[R,3,16], -dir), clamp_min(+0.5, 0).   sh layout [R][3][ncoef].
 * ------------------------------------------------------------------------------------------ */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
void o_eval_sh_color(const float* sh, int ncoef, const float* dirs, int64_t R, int deg, float* rgb) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < R; ++i) {
    float x = -dirs[3 * i], y = -dirs[3 * i + 1], z = -dirs[3 * i + 2];
    for (int ch = 0; ch < 3; ++ch) {
      const float* s = sh + ((size_t)i * 3 + ch) * ncoef;
      float r = SH_C0 * s[0];
      if (deg > 0) {
        r = ((r - (SH_C1 * y) * s[1]) + (SH_C1 * z) * s[2]) - (SH_C1 * x) * s[3];
        if (deg > 1) {
          float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
          r = ((((r + (SH_C2[0] * xy) * s[4]) + (SH_C2[1] * yz) * s[5]) +
                (SH_C2[2] * ((2.f * zz - xx) - yy)) * s[6]) + (SH_C2[3] * xz) * s[7]) +
              (SH_C2[4] * (xx - yy)) * s[8];
          if (deg > 2) {
            r = ((((((r + ((SH_C3[0] * y) * (3.f * xx - yy)) * s[9]) + ((SH_C3[1] * xy) * z) * s[10]) +
                    ((SH_C3[2] * y) * ((4.f * zz - xx) - yy)) * s[11]) +
                   ((SH_C3[3] * z) * ((2.f * zz - 3.f * xx) - 3.f * yy)) * s[12]) +
                  ((SH_C3[4] * x) * ((4.f * zz - xx) - yy)) * s[13]) + ((SH_C3[5] * z) * (xx - yy)) * s[14]) +
                ((SH_C3[6] * x) * (xx - 3.f * yy)) * s[15];
          }
        }
      }
      float o = r + 0.5f;
      rgb[3 * i + ch] = o > 0.f ? o : 0.f;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * a12 pose_estimation/ray_preprocessor.py:3-9 positional_encoding + :36-44 input assembly
 *     x[141] = [pts, dir, rgb, PE(pts,8), PE(dir,8), PE(rgb,6)], PE = [sin(c-major,f-minor), cos(...)]
 * ------------------------------------------------------------------------------------------ */
#define RAY_IN 141
static void ray_input(const float* p, const float* d, const float* c, float* x) {
  for (int k = 0; k < 3; ++k) { x[k] = p[k]; x[3 + k] = d[k]; x[6 + k] = c[k]; }
  int o = 9;
  const float* src[3] = {p, d, c};
  const int fr[3] = {8, 8, 6};
  for (int s = 0; s < 3; ++s) {
    int F = fr[s];
    for (int k = 0; k < 3; ++k)
      for (int f = 0; f < F; ++f) {
        float v = src[s][k] * (float)(1 << f);
        x[o + k * F + f] = sinf(v);
        x[o + 3 * F + k * F + f] = cosf(v);
      }
    o += 6 * F;
  }
}
void o_ray_input(const float* ori, const float* dir, const float* rgb, int64_t R, float* x) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < R; ++i) ray_input(ori + 3 * i, dir + 3 * i, rgb + 3 * i, x + (size_t)RAY_IN * i);
}

/* y[M][N] = act(x[M][K] @ W[N][K]^T + b).  Every output is ONE chain of fp32 FMAs over k = 0 .. K-1 in order, then + b -- the
 * same value whatever the blocking.  Register-tiled (6 rows x 16 columns of accumulators in 12 ymm registers) over weights packed
 * into 16-column panels, so that the CPU baseline is not a strawman: round 6 replaced an 8 x 64 stack tile (190 GFLOP/s on the GPU
 * box's 128 threads) by this kernel, bit-identical results (tests/test_oracle_golden.py::test_linear_is_one_fma_chain_per_output). */
#define LB_MR 6
#define LB_NR 16
#define LB_ROWS 48 /* rows per work item: 8 register tiles share a weight panel from L1/L2 */
typedef struct { float* p; int K, N, NP; } packed_w; /* p[NP][K][16], columns beyond N are zero */
static packed_w pack_w(const float* W, int N, int K) {
  packed_w w;
  w.K = K; w.N = N; w.NP = (N + LB_NR - 1) / LB_NR;
  w.p = (float*)aligned_alloc(64, sizeof(float) * (size_t)w.NP * K * LB_NR);
  for (int pn = 0; pn < w.NP; ++pn)
    for (int k = 0; k < K; ++k)
      for (int j = 0; j < LB_NR; ++j) {
        int n = pn * LB_NR + j;
        w.p[((size_t)pn * K + k) * LB_NR + j] = n < N ? W[(size_t)n * K + k] : 0.f;
      }
  return w;
}
static void linear_block(const float* x, int64_t M, const packed_w* w, const float* b, int relu, float* y) {
  const int K = w->K, N = w->N;
  const int64_t nblk = (M + LB_ROWS - 1) / LB_ROWS;
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t blk = 0; blk < nblk; ++blk) {
    const int64_t m0 = blk * LB_ROWS, m1 = (m0 + LB_ROWS) < M ? (m0 + LB_ROWS) : M;
    for (int pn = 0; pn < w->NP; ++pn) {
      const float* wp = w->p + (size_t)pn * K * LB_NR;
      const int n0 = pn * LB_NR, nb = (N - n0) < LB_NR ? (N - n0) : LB_NR;
      for (int64_t r0 = m0; r0 < m1; r0 += LB_MR) {
        const int mb = (int)((m1 - r0) < LB_MR ? (m1 - r0) : LB_MR);
        __m256 a[LB_MR][2];
        for (int i = 0; i < LB_MR; ++i) a[i][0] = a[i][1] = _mm256_setzero_ps();
        const float* xr[LB_MR];
        for (int i = 0; i < LB_MR; ++i) xr[i] = x + (size_t)(r0 + (i < mb ? i : 0)) * K; /* tail rows repeat row 0, not stored */
        for (int k = 0; k < K; ++k) {
          const __m256 w0 = _mm256_load_ps(wp + (size_t)k * LB_NR), w1 = _mm256_load_ps(wp + (size_t)k * LB_NR + 8);
          for (int i = 0; i < LB_MR; ++i) {
            const __m256 xv = _mm256_broadcast_ss(xr[i] + k);
            a[i][0] = _mm256_fmadd_ps(xv, w0, a[i][0]);
            a[i][1] = _mm256_fmadd_ps(xv, w1, a[i][1]);
          }
        }
        for (int i = 0; i < mb; ++i) {
          float t[LB_NR];
          _mm256_storeu_ps(t, a[i][0]);
          _mm256_storeu_ps(t + 8, a[i][1]);
          for (int j = 0; j < nb; ++j) {
            float v = t[j] + b[n0 + j];
            y[(size_t)(r0 + i) * N + n0 + j] = (relu && v < 0.f) ? 0.f : v;
          }
        }
      }
    }
  }
}
void o_linear(const float* x, int64_t M, int K, const float* W, const float* b, int N, int relu, float* y) {
  packed_w w = pack_w(W, N, K);
  linear_block(x, M, &w, b, relu, y);
  free(w.p);
}

/* ------------------------------------------------------------------------------------------
 * a13 pose_estimation/ray_preprocessor.py:36-46 RayPreprocessor.forward (featureC=512, out 384)
 *     + a14's k_proj (our_multihead_attention.py:74).  feat/key may be NULL.
 * weights: W1[512][141] b1, W2[512][512] b2, W3[512][653] b3, W4[384][512] b4, Wk[384][384] bk
 * ------------------------------------------------------------------------------------------ */
void o_ray_features(const float* ori, const float* dir, const float* rgb, int64_t R, const float* W1, const float* b1,
                    const float* W2, const float* b2, const float* W3, const float* b3, const float* W4,
                    const float* b4, const float* Wk, const float* bk, float* feat, float* key) {
  const int64_t CH = 32768; /* rays per pass: 683 work items of 48 rows for the host's threads (128 on the GPU box) */
  packed_w w1 = pack_w(W1, 512, 141), w2 = pack_w(W2, 512, 512), w3 = pack_w(W3, 512, 653), w4 = pack_w(W4, 384, 512);
  packed_w wk = {0, 0, 0, 0};
  if (Wk) wk = pack_w(Wk, 384, 384);
  const int64_t ch = R < CH ? (R > 0 ? R : 1) : CH;
  float* x = (float*)malloc(sizeof(float) * ch * 141);
  float* h1 = (float*)malloc(sizeof(float) * ch * 512);
  float* cat = (float*)malloc(sizeof(float) * ch * 653);
  float* h3 = (float*)malloc(sizeof(float) * ch * 512);
  float* f = (float*)malloc(sizeof(float) * ch * 384);
  for (int64_t r0 = 0; r0 < R; r0 += CH) {
    int64_t m = (R - r0) < CH ? (R - r0) : CH;
    o_ray_input(ori + 3 * r0, dir + 3 * r0, rgb + 3 * r0, m, x);
    linear_block(x, m, &w1, b1, 1, h1);
    linear_block(h1, m, &w2, b2, 1, h3); /* h3 reused as h2 scratch */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < m; ++i) {
      memcpy(cat + (size_t)i * 653, h3 + (size_t)i * 512, sizeof(float) * 512);
      memcpy(cat + (size_t)i * 653 + 512, x + (size_t)i * 141, sizeof(float) * 141);
    }
    linear_block(cat, m, &w3, b3, 1, h3);
    float* fo = feat ? feat + (size_t)r0 * 384 : f;
    linear_block(h3, m, &w4, b4, 0, fo);
    if (key) linear_block(fo, m, &wk, bk, 0, key + (size_t)r0 * 384);
  }
  free(w1.p); free(w2.p); free(w3.p); free(w4.p); free(wk.p);
  free(x); free(h1); free(cat); free(h3); free(f);
}

/* ------------------------------------------------------------------------------------------
 * a14/a15 our_multihead_attention.py:4-12,70-79 + identification_module.py:80-82
 *   q[T][384] (already projected), key[R][384] -> scores[R] = sum_t softmax_r(q k^T / sqrt(384))
 * Two passes over the rays per token; logits in fp32, row sums in fp32 like torch's softmax
 * (torch accumulates the softmax denominator in fp32 vector lanes; tests use a tolerance).
 * ------------------------------------------------------------------------------------------ */
void o_attention_scores(const float* q, int T, const float* key, int64_t R, float* scores, float* rowmax_out,
                        float* rowsum_out) {
  const int D = 384;
  const float inv = 1.0f / sqrtf((float)D);
  (void)inv;
  const float sq = sqrtf((float)D);
  float* mx = (float*)malloc(sizeof(float) * (size_t)(T > 0 ? T : 1));
  double* sm = (double*)malloc(sizeof(double) * (size_t)(T > 0 ? T : 1));
  memset(scores, 0, sizeof(float) * (size_t)R);
  const int64_t RB = 512;
  int64_t nblk = (R + RB - 1) / RB;
  int nth = o_num_threads();
  float* tmx = (float*)malloc(sizeof(float) * (size_t)nth * (T > 0 ? T : 1));
  double* tsm = (double*)malloc(sizeof(double) * (size_t)nth * (T > 0 ? T : 1));
  /* pass 1: row max */
  for (int i = 0; i < nth * T; ++i) tmx[i] = -INFINITY;
#pragma omp parallel
  {
#ifdef _OPENMP
    int tid = omp_get_thread_num();
#else
    int tid = 0;
#endif
    float* lm = tmx + (size_t)tid * T;
#pragma omp for schedule(static)
    for (int64_t blk = 0; blk < nblk; ++blk) {
      int64_t r1 = (blk + 1) * RB < R ? (blk + 1) * RB : R;
      for (int64_t r = blk * RB; r < r1; ++r) {
        const float* kr = key + (size_t)r * D;
        for (int t = 0; t < T; ++t) {
          const float* qt = q + (size_t)t * D;
          float acc = 0.f;
#pragma omp simd reduction(+ : acc)
          for (int d = 0; d < D; ++d) acc += qt[d] * kr[d];
          float l = acc / sq;
          if (l > lm[t]) lm[t] = l;
        }
      }
    }
  }
  for (int t = 0; t < T; ++t) {
    mx[t] = -INFINITY;
    for (int i = 0; i < nth; ++i)
      if (tmx[(size_t)i * T + t] > mx[t]) mx[t] = tmx[(size_t)i * T + t];
  }
  /* pass 2: row sum of exp */
  for (int i = 0; i < nth * T; ++i) tsm[i] = 0.0;
#pragma omp parallel
  {
#ifdef _OPENMP
    int tid = omp_get_thread_num();
#else
    int tid = 0;
#endif
    double* ls = tsm + (size_t)tid * T;
#pragma omp for schedule(static)
    for (int64_t blk = 0; blk < nblk; ++blk) {
      int64_t r1 = (blk + 1) * RB < R ? (blk + 1) * RB : R;
      for (int64_t r = blk * RB; r < r1; ++r) {
        const float* kr = key + (size_t)r * D;
        for (int t = 0; t < T; ++t) {
          const float* qt = q + (size_t)t * D;
          float acc = 0.f;
#pragma omp simd reduction(+ : acc)
          for (int d = 0; d < D; ++d) acc += qt[d] * kr[d];
          ls[t] += (double)expf(acc / sq - mx[t]);
        }
      }
    }
  }
  for (int t = 0; t < T; ++t) {
    sm[t] = 0.0;
    for (int i = 0; i < nth; ++i) sm[t] += tsm[(size_t)i * T + t];
  }
  /* pass 3: column sums */
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < R; ++r) {
    const float* kr = key + (size_t)r * D;
    float s = 0.f;
    for (int t = 0; t < T; ++t) {
      const float* qt = q + (size_t)t * D;
      float acc = 0.f;
#pragma omp simd reduction(+ : acc)
      for (int d = 0; d < D; ++d) acc += qt[d] * kr[d];
      s += expf(acc / sq - mx[t]) / (float)sm[t];
    }
    scores[r] = s;
  }
  if (rowmax_out) memcpy(rowmax_out, mx, sizeof(float) * (size_t)T);
  if (rowsum_out)
    for (int t = 0; t < T; ++t) rowsum_out[t] = (float)sm[t];
  free(mx); free(sm); free(tmx); free(tsm);
}

/* torch.topk(scores, k) sorted descending; tie rule of this build: lowest index first. */
void o_topk(const float* scores, int64_t R, int k, int64_t* idx, float* val) {
  int cnt = 0;
  for (int64_t r = 0; r < R; ++r) {
    float s = scores[r];
    if (cnt < k || s > val[cnt - 1]) {
      int pos = cnt < k ? cnt : k - 1;
      while (pos > 0 && val[pos - 1] < s) {
        val[pos] = val[pos - 1];
        idx[pos] = idx[pos - 1];
        --pos;
      }
      val[pos] = s;
      idx[pos] = r;
      if (cnt < k) ++cnt;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * a17 pose_estimation/test.py:157-162 duplicate-origin filter.
 * sel_ori [k][3] = rays_ori[idx].  keep[k] out.  Returns number kept.
 * ------------------------------------------------------------------------------------------ */
int o_unique_origin_filter(const float* sel_ori, int k, uint8_t* keep) {
  int* cnt = (int*)calloc((size_t)k, sizeof(int));
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j)
      if (sel_ori[3 * i] == sel_ori[3 * j] && sel_ori[3 * i + 1] == sel_ori[3 * j + 1] &&
          sel_ori[3 * i + 2] == sel_ori[3 * j + 2])
        ++cnt[i];
  int ne = 3 * k, nt = 0;
  float* test = (float*)malloc(sizeof(float) * (size_t)(ne > 0 ? ne : 1));
  for (int i = 0; i < k; ++i)
    if (cnt[i] == 1)
      for (int c = 0; c < 3; ++c) test[nt++] = sel_ori[3 * i + c];
  int64_t thr = (int64_t)(10.0f * pow((double)ne, 0.145));
  int sorting = !(nt < thr);
  int kept = 0;
  for (int i = 0; i < k; ++i) {
    int any = 0;
    for (int c = 0; c < 3 && !any; ++c) {
      int p = 3 * i + c;
      float v = sel_ori[p];
      for (int t = 0; t < nt && !any; ++t)
        if (test[t] == v) any = 1;
      if (sorting)
        for (int q2 = p + 1; q2 < ne && !any; ++q2)
          if (sel_ori[q2] == v) any = 1;
    }
    keep[i] = (uint8_t)any;
    kept += any;
  }
  free(cnt);
  free(test);
  return kept;
}

/* 3x3 LU with partial pivoting (what torch.linalg.solve / det / inv do through LAPACK) */
static int lu3(float* m, int* piv, float* sign) {
  *sign = 1.f;
  for (int c = 0; c < 3; ++c) {
    int p = c;
    for (int r = c + 1; r < 3; ++r)
      if (fabsf(m[3 * r + c]) > fabsf(m[3 * p + c])) p = r;
    piv[c] = p;
    if (p != c) {
      for (int k = 0; k < 3; ++k) { float t = m[3 * c + k]; m[3 * c + k] = m[3 * p + k]; m[3 * p + k] = t; }
      *sign = -*sign;
    }
    float d = m[3 * c + c];
    if (d == 0.f) return 0;
    for (int r = c + 1; r < 3; ++r) {
      m[3 * r + c] /= d;
      for (int k = c + 1; k < 3; ++k) m[3 * r + k] -= m[3 * r + c] * m[3 * c + k];
    }
  }
  return 1;
}
static void lu3_solve(const float* lu, const int* piv, float* b) {
  for (int c = 0; c < 3; ++c) {
    if (piv[c] != c) { float t = b[c]; b[c] = b[piv[c]]; b[piv[c]] = t; }
  }
  for (int r = 1; r < 3; ++r)
    for (int k = 0; k < r; ++k) b[r] -= lu[3 * r + k] * b[k];
  for (int r = 2; r >= 0; --r) {
    for (int k = r + 1; k < 3; ++k) b[r] -= lu[3 * r + k] * b[k];
    b[r] /= lu[3 * r + r];
  }
}

/* ------------------------------------------------------------------------------------------
 * a18 pose_estimation/line_intersection.py:75-154 compute_line_intersection_impl2 (unweighted,
 *     as called at test.py:169-171,177-179).  NaN centre when det(R) < 1e-7.
 * ------------------------------------------------------------------------------------------ */
void o_line_intersection(const float* pts, const float* dirs, int k, float* centre) {
  float Rm[9] = {0}, q[3] = {0};
  for (int i = 0; i < k; ++i) {
    const float* d = dirs + 3 * i;
    const float* p = pts + 3 * i;
    float P[9];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) P[3 * a + b] = (a == b ? 1.f : 0.f) - d[a] * d[b];
    for (int a = 0; a < 9; ++a) Rm[a] += P[a];
    for (int a = 0; a < 3; ++a) q[a] += (P[3 * a] * p[0] + P[3 * a + 1] * p[1]) + P[3 * a + 2] * p[2];
  }
  float lu[9];
  int piv[3];
  float sg;
  memcpy(lu, Rm, sizeof(lu));
  int ok = lu3(lu, piv, &sg);
  float det = ok ? sg * lu[0] * lu[4] * lu[8] : 0.f;
  if (det < 1.0e-7f || !ok) {
    centre[0] = centre[1] = centre[2] = NAN;
    return;
  }
  lu3_solve(lu, piv, q);
  centre[0] = q[0]; centre[1] = q[1]; centre[2] = q[2];
}

/* ------------------------------------------------------------------------------------------
 * a17-a21 the tail of the per-image loop, pose_estimation/test.py:157-198,216-218,268-288:
 * filter -> normalise weights -> centre -> exclude_negatives (line_intersection.py:29-34) ->
 * renormalise -> centre again (identical: unweighted) -> watch dir -> make_rotation_mat
 * (line_intersection.py:5-26) -> det<1e-7 => I -> inv -> c2w; NaN => I4.
 * out: c2w[16], centre[3] (pre-fallback), w_final[k] (0 for dropped), keep[k],
 *      flags[2] = {singular_rotation, nan_pose}, n_kept.
 * ------------------------------------------------------------------------------------------ */
void o_make_rotation_mat(const float* direction, const float* up, float* Rm) {
  float xa[3], ya[3];
  cross3(up, direction, xa);
  float nx = sqrtf((xa[0] * xa[0] + xa[1] * xa[1]) + xa[2] * xa[2]);
  for (int k = 0; k < 3; ++k) xa[k] /= nx;
  cross3(direction, xa, ya);
  float ny = sqrtf((ya[0] * ya[0] + ya[1] * ya[1]) + ya[2] * ya[2]);
  for (int k = 0; k < 3; ++k) ya[k] /= ny;
  for (int k = 0; k < 3; ++k) { Rm[k] = xa[k]; Rm[3 + k] = ya[k]; Rm[6 + k] = direction[k]; }
}
static int inv3_lu(const float* A, float* inv) {
  float lu[9];
  int piv[3];
  float sg;
  memcpy(lu, A, sizeof(lu));
  if (!lu3(lu, piv, &sg)) return 0;
  for (int c = 0; c < 3; ++c) {
    float e[3] = {0.f, 0.f, 0.f};
    e[c] = 1.f;
    lu3_solve(lu, piv, e);
    for (int r = 0; r < 3; ++r) inv[3 * r + c] = e[r];
  }
  return 1;
}
void o_pose_from_topk(const float* rays_ori, const float* rays_dir, const int64_t* idx, const float* weights, int k,
                      const float* up, float* c2w, float* centre_out, float* w_final, uint8_t* keep, int* flags,
                      int* n_kept) {
  float* so = (float*)calloc(3 * (size_t)(k > 0 ? k : 1), sizeof(float));
  float* sd = (float*)calloc(3 * (size_t)(k > 0 ? k : 1), sizeof(float));
  float* w = (float*)calloc((size_t)(k > 0 ? k : 1), sizeof(float));
  for (int i = 0; i < k; ++i)
    for (int c = 0; c < 3; ++c) so[3 * i + c] = rays_ori[3 * idx[i] + c];
  int kept = o_unique_origin_filter(so, k, keep);
  int m = 0;
  for (int i = 0; i < k; ++i)
    if (keep[i]) {
      for (int c = 0; c < 3; ++c) {
        so[3 * m + c] = rays_ori[3 * idx[i] + c];
        sd[3 * m + c] = rays_dir[3 * idx[i] + c];
      }
      w[m++] = weights[i];
    }
  float sum = 0.f;
  for (int i = 0; i < m; ++i) sum += w[i];
  for (int i = 0; i < m; ++i) w[i] = w[i] / sum;
  float centre[3];
  o_line_intersection(so, sd, m, centre);
  for (int i = 0; i < m; ++i) {
    float v0 = centre[0] - so[3 * i], v1 = centre[1] - so[3 * i + 1], v2 = centre[2] - so[3 * i + 2];
    float d = (v0 * sd[3 * i] + v1 * sd[3 * i + 1]) + v2 * sd[3 * i + 2];
    w[i] = w[i] * (d > 0.f ? 1.f : 0.f);
  }
  sum = 0.f;
  for (int i = 0; i < m; ++i) sum += w[i];
  for (int i = 0; i < m; ++i) w[i] = w[i] / sum;
  o_line_intersection(so, sd, m, centre);
  float watch[3] = {0.f, 0.f, 0.f};
  for (int i = 0; i < m; ++i)
    for (int c = 0; c < 3; ++c) watch[c] += sd[3 * i + c] * w[i];
  float wn = sqrtf((watch[0] * watch[0] + watch[1] * watch[1]) + watch[2] * watch[2]);
  for (int c = 0; c < 3; ++c) watch[c] /= wn;
  float neg[3] = {-watch[0], -watch[1], -watch[2]};
  float Rw[9];
  o_make_rotation_mat(neg, up, Rw);
  float lu[9];
  int piv[3];
  float sg;
  memcpy(lu, Rw, sizeof(lu));
  int ok = lu3(lu, piv, &sg);
  float det = ok ? sg * lu[0] * lu[4] * lu[8] : 0.f;
  flags[0] = det < 1.0e-7f;
  if (flags[0]) {
    for (int i = 0; i < 9; ++i) Rw[i] = (i % 4 == 0) ? 1.f : 0.f;
  }
  float Ri[9];
  if (!inv3_lu(Rw, Ri))
    for (int i = 0; i < 9; ++i) Ri[i] = NAN;
  for (int i = 0; i < 16; ++i) c2w[i] = (i % 5 == 0) ? 1.f : 0.f;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) c2w[4 * r + c] = Ri[3 * r + c];
    c2w[4 * r + 3] = centre[r];
  }
  int nan = 0;
  for (int i = 0; i < 16; ++i)
    if (c2w[i] != c2w[i]) nan = 1;
  flags[1] = nan;
  if (nan)
    for (int i = 0; i < 16; ++i) c2w[i] = (i % 5 == 0) ? 1.f : 0.f;
  for (int c = 0; c < 3; ++c) centre_out[c] = centre[c];
  m = 0;
  for (int i = 0; i < k; ++i) w_final[i] = keep[i] ? w[m++] : 0.f;
  *n_kept = kept;
  free(so); free(sd); free(w);
}

/* a21 pose_estimation/error_computation.py:3-8 */
void o_pose_errors(const float* gt_c2w, const float* pred_c2w, float* terr, float* aerr_deg) {
  float d0 = gt_c2w[3] - pred_c2w[3], d1 = gt_c2w[7] - pred_c2w[7], d2 = gt_c2w[11] - pred_c2w[11];
  *terr = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
  float Re[9], Ri[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Re[3 * r + c] = pred_c2w[4 * r + c];
  if (!inv3_lu(Re, Ri)) {
    *aerr_deg = NAN;
    return;
  }
  float tr = 0.f;
  for (int r = 0; r < 3; ++r)
    tr += (gt_c2w[4 * r] * Ri[r] + gt_c2w[4 * r + 1] * Ri[3 + r]) + gt_c2w[4 * r + 2] * Ri[6 + r];
  float ca = (tr - 1.f) / 2.f;
  ca = ca < -1.f ? -1.f : (ca > 1.f ? 1.f : ca);
  *aerr_deg = acosf(ca) * (float)(180.0 / PI_D);
}

/* SURVEY 8(f)#1 forward: pose_estimation/distance_based_loss.py:5-71 (target_score of best_one_to_one_rays_selector) and
 * :204-222 (rescale to sum = total_number_of_features; python scalar / tensor = reciprocal * scalar).  The sum is taken in
 * double (torch's fp32 pairwise sum agrees with it to ~1e-7). */
void o_distance_target(const float* ori, const float* dir, int64_t r, const float* pose, int n_tokens, float* combined, float* sum_out) {
  const float cx = pose[3], cy = pose[7], cz = pose[11];
  const float zx = pose[2], zy = pose[6], zz = pose[10];
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int64_t i = 0; i < r; ++i) {
    const float ox = ori[3 * i], oy = ori[3 * i + 1], oz = ori[3 * i + 2];
    const float dx = dir[3 * i], dy = dir[3 * i + 1], dz = dir[3 * i + 2];
    const float vx = cx - ox, vy = cy - oy, vz = cz - oz;
    const float t = (vx * dx + vy * dy) + vz * dz;
    float px = ox, py = oy, pz = oz;
    if (!(t < 0.f)) { px = ox + t * dx; py = oy + t * dy; pz = oz + t * dz; }
    const float ex = px - cx, ey = py - cy, ez = pz - cz;
    const float dist = sqrtf((ex * ex + ey * ey) + ez * ez);
    float target = 1.f - tanhf(dist);
    const float p = ((ox - cx) * zx + (oy - cy) * zy) + (oz - cz) * zz;
    target = target * (((p / fabsf(p)) + 1.f) / 2.f);
    combined[i] = target;
    s += (double)target;
  }
  const float sum = (float)s;
  const float mult = (1.f / sum) * (float)n_tokens;
  for (int64_t i = 0; i < r; ++i) combined[i] = combined[i] * mult;
  if (sum_out) *sum_out = sum;
}
