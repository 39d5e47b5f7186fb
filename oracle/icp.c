/*
 * icp.c — CPU ORACLE (test infrastructure, not product code; parity unpinned,
 * see ope_oracle.h).
 *
 * Restates, step for step:
 *   IterativeClosestPoint::computeTransformation   impl/icp_mod.hpp:119-272
 *   IterativeClosestPoint::transformCloud          impl/icp_mod.hpp:48-115
 *   Registration::align / getFitnessScore          impl/registration_mod.hpp:131-219
 *   CorrespondenceEstimation::determine*           impl/correspondence_estimation_mod.hpp:127-303
 *   normal shooting                                impl/correspondence_estimation_normal_shooting_weighted.hpp:107-145
 *   rejector scores                                correspondence_rejection_mod.h:368-391
 *   self-occluded rejector                         impl/correspondence_rejection_self_occluded_normal.cpp:43-64
 *   DefaultConvergenceCriteria                     default_convergence_criteria_mod.h:94-121,226-234
 *   getAlignStrength                               icp_mod.h:249-260
 * and, from the un-vendored PCL 1.7.x / Eigen 3 (published algorithms):
 *   DefaultConvergenceCriteria::hasConverged, TransformationEstimationSVD
 *   (use_umeyama_) -> Eigen::umeyama(src, dst, with_scaling=false).
 */
#include "ope_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* 3x3 SVD by cyclic Jacobi on A^T A, then U = A V / s with            */
/* Gram-Schmidt completion for (near-)zero singular values.            */
/* ------------------------------------------------------------------ */
static void jacobi_eig3(double S[9], double V[9]) {
  /* S symmetric (row-major), on exit diagonal holds eigenvalues, V columns eigenvectors */
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = fabs(S[1]) + fabs(S[2]) + fabs(S[5]);
    double diag = fabs(S[0]) + fabs(S[4]) + fabs(S[8]);
    if (off <= 1e-300 || off <= 1e-17 * diag) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double apq = S[3 * p + q];
        if (apq == 0.0) continue;
        double app = S[3 * p + p], aqq = S[3 * q + q];
        double theta = (aqq - app) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) { /* columns p,q of S */
          double skp = S[3 * k + p], skq = S[3 * k + q];
          S[3 * k + p] = c * skp - s * skq;
          S[3 * k + q] = s * skp + c * skq;
        }
        for (int k = 0; k < 3; ++k) { /* rows p,q of S */
          double spk = S[3 * p + k], sqk = S[3 * q + k];
          S[3 * p + k] = c * spk - s * sqk;
          S[3 * q + k] = s * spk + c * sqk;
        }
        for (int k = 0; k < 3; ++k) {
          double vkp = V[3 * k + p], vkq = V[3 * k + q];
          V[3 * k + p] = c * vkp - s * vkq;
          V[3 * k + q] = s * vkp + c * vkq;
        }
      }
  }
}

static double det3(const double M[9]) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

void orc_svd3(const double A[9], double U[9], double s[3], double V[9]) {
  double AtA[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += A[3 * k + i] * A[3 * k + j];
      AtA[3 * i + j] = a;
    }
  double Vt[9];
  jacobi_eig3(AtA, Vt);
  double ev[3] = {AtA[0], AtA[4], AtA[8]};
  int ord[3] = {0, 1, 2};
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (ev[ord[j]] > ev[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
  for (int c = 0; c < 3; ++c) {
    for (int r = 0; r < 3; ++r) V[3 * r + c] = Vt[3 * r + ord[c]];
  }
  /* singular values = |A v_c| (more accurate than sqrt(eig) for small ones) */
  double Ucol[3][3];
  for (int c = 0; c < 3; ++c) {
    for (int r = 0; r < 3; ++r) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += A[3 * r + k] * V[3 * k + c];
      Ucol[c][r] = a;
    }
    s[c] = sqrt(Ucol[c][0] * Ucol[c][0] + Ucol[c][1] * Ucol[c][1] + Ucol[c][2] * Ucol[c][2]);
  }
  /* orthonormalise U columns (modified Gram-Schmidt with completion) */
  double tiny = 1e-14 * (s[0] > 0 ? s[0] : 1.0);
  for (int c = 0; c < 3; ++c) {
    double *u = Ucol[c];
    for (int p = 0; p < c; ++p) {
      double d = u[0] * Ucol[p][0] + u[1] * Ucol[p][1] + u[2] * Ucol[p][2];
      for (int r = 0; r < 3; ++r) u[r] -= d * Ucol[p][r];
    }
    double nrm = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    if (s[c] <= tiny || nrm <= 1e-8 * (s[c] > 0 ? s[c] : 1.0) + 1e-300) {
      /* pick any unit vector orthogonal to previous columns */
      if (c == 0) { u[0] = 1; u[1] = 0; u[2] = 0; }
      else if (c == 1) {
        const double *a = Ucol[0];
        int m = fabs(a[0]) < fabs(a[1]) ? (fabs(a[0]) < fabs(a[2]) ? 0 : 2) : (fabs(a[1]) < fabs(a[2]) ? 1 : 2);
        double e[3] = {0, 0, 0};
        e[m] = 1;
        double d = a[m];
        for (int r = 0; r < 3; ++r) u[r] = e[r] - d * a[r];
      } else {
        const double *a = Ucol[0], *b = Ucol[1];
        u[0] = a[1] * b[2] - a[2] * b[1];
        u[1] = a[2] * b[0] - a[0] * b[2];
        u[2] = a[0] * b[1] - a[1] * b[0];
      }
      nrm = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    }
    for (int r = 0; r < 3; ++r) u[r] /= nrm;
  }
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) U[3 * r + c] = Ucol[c][r];
}

/* Eigen::umeyama tail: sigma (row-major, = 1/n * dst_demean * src_demean^T),
 * means -> 4x4 column-major float. */
static void umeyama_finish(const double sigma[9], const double src_mean[3], const double dst_mean[3], float T[16]) {
  double U[9], s[3], V[9];
  orc_svd3(sigma, U, s, V);
  double S[3] = {1, 1, 1};
  if (det3(sigma) < 0) S[2] = -1;
  int rank = 0;
  for (int i = 0; i < 3; ++i)
    if (!(fabs(s[i]) <= fabs(s[0]) * 1e-5)) ++rank; /* Eigen isMuchSmallerThan(d_i, d_0), float dummy_precision 1e-5 */
  double R[9];
  if (rank == 2) {
    if (det3(U) * det3(V) > 0) { S[0] = S[1] = S[2] = 1; }
    else { S[0] = S[1] = 1; S[2] = -1; }
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += U[3 * i + k] * S[k] * V[3 * j + k];
      R[3 * i + j] = a;
    }
  double t[3];
  for (int i = 0; i < 3; ++i)
    t[i] = dst_mean[i] - (R[3 * i] * src_mean[0] + R[3 * i + 1] * src_mean[1] + R[3 * i + 2] * src_mean[2]);
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) T[4 * c + r] = (float)R[3 * r + c];
  T[3] = T[7] = T[11] = 0.f;
  T[12] = (float)t[0]; T[13] = (float)t[1]; T[14] = (float)t[2]; T[15] = 1.f;
}

int orc_umeyama(const float *src, const float *tgt, int n, int acc_mode, float T[16]) {
  if (n < 1) return -1;
  double sm[3], dm[3], sigma[9];
  if (acc_mode == 0) {
    /* Scalar=float throughout, as pcl::umeyama instantiated by
     * TransformationEstimationSVD<..., float> */
    float smf[3] = {0, 0, 0}, dmf[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
      for (int d = 0; d < 3; ++d) { smf[d] += src[3 * i + d]; dmf[d] += tgt[3 * i + d]; }
    float inv = 1.0f / (float)n;
    for (int d = 0; d < 3; ++d) { smf[d] *= inv; dmf[d] *= inv; }
    float sg[9] = {0};
    for (int i = 0; i < n; ++i) {
      float a[3], b[3];
      for (int d = 0; d < 3; ++d) { a[d] = src[3 * i + d] - smf[d]; b[d] = tgt[3 * i + d] - dmf[d]; }
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) sg[3 * r + c] += b[r] * a[c];
    }
    for (int k = 0; k < 9; ++k) sigma[k] = (double)(sg[k] * inv);
    for (int d = 0; d < 3; ++d) { sm[d] = smf[d]; dm[d] = dmf[d]; }
  } else {
    double s0[3] = {0, 0, 0}, d0[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
      for (int d = 0; d < 3; ++d) { s0[d] += src[3 * i + d]; d0[d] += tgt[3 * i + d]; }
    for (int d = 0; d < 3; ++d) { sm[d] = s0[d] / n; dm[d] = d0[d] / n; }
    for (int k = 0; k < 9; ++k) sigma[k] = 0;
    for (int i = 0; i < n; ++i) {
      double a[3], b[3];
      for (int d = 0; d < 3; ++d) { a[d] = src[3 * i + d] - sm[d]; b[d] = tgt[3 * i + d] - dm[d]; }
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) sigma[3 * r + c] += b[r] * a[c];
    }
    for (int k = 0; k < 9; ++k) sigma[k] /= n;
  }
  umeyama_finish(sigma, sm, dm, T);
  return 0;
}

int orc_umeyama_from_sums(const double S[17], const double pivot[3], float T[16]) {
  double n = S[0];
  if (n < 1) return -1;
  double sm[3], dm[3], sigma[9];
  for (int d = 0; d < 3; ++d) { sm[d] = S[1 + d] / n; dm[d] = S[4 + d] / n; }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) sigma[3 * r + c] = S[7 + 3 * r + c] / n - dm[r] * sm[c];
  for (int d = 0; d < 3; ++d) { sm[d] += pivot[d]; dm[d] += pivot[d]; }
  umeyama_finish(sigma, sm, dm, T);
  return 0;
}

/* ------------------------------------------------------------------ */
/* uPCL transformation_estimation_point_to_plane_lls.hpp (published algorithm, Low 2004):
 * products a,b,c,d are formed in float (the operands are `const float&`), accumulated in double;
 * x = ATA^-1 ATb; rotation rebuilt from the three angles with full sin/cos. */
static int solve6(double A[36], double b[6], double x[6]) {
  int perm[6] = {0, 1, 2, 3, 4, 5};
  for (int k = 0; k < 6; ++k) {
    int piv = k;
    for (int r = k + 1; r < 6; ++r)
      if (fabs(A[6 * perm[r] + k]) > fabs(A[6 * perm[piv] + k])) piv = r;
    int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
    double d = A[6 * perm[k] + k];
    if (fabs(d) < 1e-300) return -1;
    for (int r = k + 1; r < 6; ++r) {
      double f = A[6 * perm[r] + k] / d;
      for (int c = k; c < 6; ++c) A[6 * perm[r] + c] -= f * A[6 * perm[k] + c];
      b[perm[r]] -= f * b[perm[k]];
    }
  }
  for (int k = 5; k >= 0; --k) {
    double v = b[perm[k]];
    for (int c = k + 1; c < 6; ++c) v -= A[6 * perm[k] + c] * x[c];
    x[k] = v / A[6 * perm[k] + k];
  }
  return 0;
}

int orc_point_to_plane_lls(const float *src, const float *tgt, const float *tgt_nrm, int n, float T[16]) {
  static const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  memcpy(T, I4, sizeof I4);
  double ATA[36] = {0}, ATb[6] = {0};
  for (int i = 0; i < n; ++i) {
    const float sx = src[3 * i], sy = src[3 * i + 1], sz = src[3 * i + 2];
    const float dx = tgt[3 * i], dy = tgt[3 * i + 1], dz = tgt[3 * i + 2];
    const float nx = tgt_nrm[3 * i], ny = tgt_nrm[3 * i + 1], nz = tgt_nrm[3 * i + 2];
    if (!isfinite(sx) || !isfinite(sy) || !isfinite(sz) || !isfinite(dx) || !isfinite(dy) || !isfinite(dz) ||
        !isfinite(nx) || !isfinite(ny) || !isfinite(nz)) continue;
    const double v[6] = {(double)(nz * sy - ny * sz), (double)(nx * sz - nz * sx), (double)(ny * sx - nx * sy),
                         (double)nx, (double)ny, (double)nz};
    const double d = (double)(nx * dx + ny * dy + nz * dz - nx * sx - ny * sy - nz * sz);
    for (int r = 0; r < 6; ++r) {
      for (int c = r; c < 6; ++c) ATA[6 * r + c] += v[r] * v[c];
      ATb[r] += v[r] * d;
    }
  }
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < r; ++c) ATA[6 * r + c] = ATA[6 * c + r];
  double x[6];
  if (solve6(ATA, ATb, x) != 0) return -1;
  const double al = x[0], be = x[1], ga = x[2];
  double R[9];
  R[0] = cos(ga) * cos(be);
  R[1] = -sin(ga) * cos(al) + cos(ga) * sin(be) * sin(al);
  R[2] = sin(ga) * sin(al) + cos(ga) * sin(be) * cos(al);
  R[3] = sin(ga) * cos(be);
  R[4] = cos(ga) * cos(al) + sin(ga) * sin(be) * sin(al);
  R[5] = -cos(ga) * sin(al) + sin(ga) * sin(be) * cos(al);
  R[6] = -sin(be);
  R[7] = cos(be) * sin(al);
  R[8] = cos(be) * cos(al);
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) T[4 * c + r] = (float)R[3 * r + c];
  T[12] = (float)x[3]; T[13] = (float)x[4]; T[14] = (float)x[5];
  return 0;
}

/* ------------------------------------------------------------------ */
void orc_convergence_init(orc_convergence *c) {
  c->max_iterations = 100;
  c->failure_after_max_iter = 0;
  c->rotation_threshold = 0.99999;
  c->translation_threshold = 3e-4 * 3e-4;
  c->mse_threshold_relative = 0.00001;
  c->mse_threshold_absolute = 1e-12;
  c->max_iterations_similar_transforms = 0;
  c->iterations_similar_transforms = 0;
  c->prev_mse = DBL_MAX;
  c->cur_mse = DBL_MAX;
  c->state = ORC_CONV_NOT_CONVERGED;
}

int orc_convergence_step(orc_convergence *c, int iterations, const float T[16], double mse) {
  c->state = ORC_CONV_NOT_CONVERGED;
  /* 1. iteration cap */
  if (iterations >= c->max_iterations) {
    if (c->failure_after_max_iter) return 0;
    c->state = ORC_CONV_ITERATIONS;
    return 1;
  }
  /* 2. incremental-transform epsilon */
  double cos_angle = 0.5 * ((double)T[0] + (double)T[5] + (double)T[10] - 1.0);
  double tr2 = (double)T[12] * T[12] + (double)T[13] * T[13] + (double)T[14] * T[14];
  if (cos_angle >= c->rotation_threshold && tr2 <= c->translation_threshold) {
    if (c->iterations_similar_transforms < c->max_iterations_similar_transforms) {
      ++c->iterations_similar_transforms;
      return 0;
    }
    c->iterations_similar_transforms = 0;
    c->state = ORC_CONV_TRANSFORM;
    return 1;
  }
  /* 3. MSE, absolute then relative */
  c->cur_mse = mse;
  if (fabs(c->cur_mse - c->prev_mse) < c->mse_threshold_absolute) {
    if (c->iterations_similar_transforms < c->max_iterations_similar_transforms) {
      ++c->iterations_similar_transforms;
      return 0;
    }
    c->iterations_similar_transforms = 0;
    c->state = ORC_CONV_ABS_MSE;
    return 1;
  }
  if (fabs(c->cur_mse - c->prev_mse) / c->prev_mse < c->mse_threshold_relative) {
    if (c->iterations_similar_transforms < c->max_iterations_similar_transforms) {
      ++c->iterations_similar_transforms;
      return 0;
    }
    c->iterations_similar_transforms = 0;
    c->state = ORC_CONV_REL_MSE;
    return 1;
  }
  c->prev_mse = c->cur_mse;
  return 0;
}

/* ------------------------------------------------------------------ */
static inline int finite3(const float *p) { return isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]); }

void orc_transform_points(const float *in, int n, const float T[16], float *out) {
  for (int i = 0; i < n; ++i) {
    const float *p = in + 3 * i;
    float *o = out + 3 * i;
    if (!finite3(p)) { if (o != p) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; } continue; }
    float x = p[0], y = p[1], z = p[2];
    /* Eigen 4x4 * (x,y,z,1): column-major accumulation */
    o[0] = T[0] * x + T[4] * y + T[8] * z + T[12];
    o[1] = T[1] * x + T[5] * y + T[9] * z + T[13];
    o[2] = T[2] * x + T[6] * y + T[10] * z + T[14];
  }
}

void orc_transform_normals(const float *in, int n, const float T[16], float *out) {
  for (int i = 0; i < n; ++i) {
    const float *p = in + 3 * i;
    float *o = out + 3 * i;
    if (!finite3(p)) { if (o != p) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; } continue; }
    float x = p[0], y = p[1], z = p[2];
    o[0] = T[0] * x + T[4] * y + T[8] * z;
    o[1] = T[1] * x + T[5] * y + T[9] * z;
    o[2] = T[2] * x + T[6] * y + T[10] * z;
  }
}

static void mat4_mul_f(const float A[16], const float B[16], float C[16]) {
  float R[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      float a = 0;
      for (int k = 0; k < 4; ++k) a += A[4 * k + r] * B[4 * c + k];
      R[4 * c + r] = a;
    }
  memcpy(C, R, sizeof R);
}

static void mat4_mul_d(const double A[16], const double B[16], double C[16]) {
  double R[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      double a = 0;
      for (int k = 0; k < 4; ++k) a += A[4 * k + r] * B[4 * c + k];
      R[4 * c + r] = a;
    }
  memcpy(C, R, sizeof R);
}

static int is_identity(const float T[16]) {
  for (int i = 0; i < 16; ++i)
    if (T[i] != ((i % 5 == 0) ? 1.f : 0.f)) return 0;
  return 1;
}

void orc_icp_default_params(orc_icp_params *p) {
  memset(p, 0, sizeof *p);
  p->max_iterations = 10;
  p->transformation_epsilon = 0.0;
  p->euclidean_fitness_epsilon = -DBL_MAX;
  p->max_corr_dist = sqrt(DBL_MAX);
  p->use_reciprocal = 0;
  p->min_correspondences = 3;
  p->corr_mode = 0;
  p->k_normal_shooting = 20;
  p->surface_normal_thr = 0.7;
  p->self_occluded_thr = 0.6;
  p->mse_threshold_absolute = 1e-12;
  p->acc_mode = 0;
  p->transform_mode = 0;
}

double orc_fitness(const float *src_xyz, int ns, const float *tgt_xyz, int nt, const float T[16],
                   double max_range, int *n_used) {
  orc_kdtree *tree = orc_kdtree_build(tgt_xyz, nt, 15);
  double score = 0;
  int nr = 0;
  for (int i = 0; i < ns; ++i) {
    float p[3];
    orc_transform_points(src_xyz + 3 * i, 1, T, p);
    int32_t id; float d; int32_t f;
    orc_kdtree_knn(tree, p, 1, 1, &id, &d, &f);
    if (f && (double)d <= max_range) { score += d; nr++; }
  }
  orc_kdtree_free(tree);
  if (n_used) *n_used = nr;
  return nr > 0 ? score / nr : DBL_MAX;
}

void orc_icp_partial_sums(const float *src_xyz, int ns, const orc_kdtree *tgt_tree, const float *tgt_xyz,
                          const float T[16], double max_corr_dist, const double pivot[3], double S[17]) {
  for (int k = 0; k < 17; ++k) S[k] = 0;
  double md2 = max_corr_dist * max_corr_dist;
  for (int i = 0; i < ns; ++i) {
    if (!finite3(src_xyz + 3 * i)) continue;
    float p[3];
    orc_transform_points(src_xyz + 3 * i, 1, T, p);
    int32_t id; float d; int32_t f;
    orc_kdtree_knn(tgt_tree, p, 1, 1, &id, &d, &f);
    if (!f || (double)d > md2) continue;
    double s[3], t[3];
    for (int k = 0; k < 3; ++k) { s[k] = (double)p[k] - pivot[k]; t[k] = (double)tgt_xyz[3 * id + k] - pivot[k]; }
    S[0] += 1;
    for (int k = 0; k < 3; ++k) { S[1 + k] += s[k]; S[4 + k] += t[k]; }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) S[7 + 3 * r + c] += t[r] * s[c];
    S[16] += d;
  }
}

/* The same pass over `n_threads` contiguous slices of the source, one POSIX thread each, the slices' sums added in slice
 * order ("also reported with all cores").  The kd-tree is read-only during a search. */
typedef struct {
  const float *src; int ns; const orc_kdtree *tree; const float *tgt; const float *T; double mcd; const double *pivot; double S[17];
} ps_job;
static void *ps_worker(void *arg) {
  ps_job *j = (ps_job *)arg;
  orc_icp_partial_sums(j->src, j->ns, j->tree, j->tgt, j->T, j->mcd, j->pivot, j->S);
  return NULL;
}
void orc_icp_partial_sums_mt(const float *src_xyz, int ns, const orc_kdtree *tgt_tree, const float *tgt_xyz, const float T[16],
                             double max_corr_dist, const double pivot[3], int n_threads, double S[17]) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 1024) n_threads = 1024;
  ps_job *jobs = (ps_job *)calloc((size_t)n_threads, sizeof(ps_job));
  pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  char *started = (char *)calloc((size_t)n_threads, 1);
  for (int t = 0; t < n_threads; ++t) {
    const long long a = (long long)ns * t / n_threads, b = (long long)ns * (t + 1) / n_threads;
    jobs[t] = (ps_job){src_xyz + 3 * a, (int)(b - a), tgt_tree, tgt_xyz, T, max_corr_dist, pivot, {0}};
    started[t] = pthread_create(&th[t], NULL, ps_worker, &jobs[t]) == 0;
    if (!started[t]) ps_worker(&jobs[t]);   /* could not start a thread: do the slice here */
  }
  for (int k = 0; k < 17; ++k) S[k] = 0;
  for (int t = 0; t < n_threads; ++t) {
    if (started[t]) pthread_join(th[t], NULL);
    for (int k = 0; k < 17; ++k) S[k] += jobs[t].S[k];
  }
  free(jobs); free(th); free(started);
}

/* Search threads of orc_icp / orc_icp_fixed (tests that follow a whole 1 M-point run).  Only the per-query searches run in
 * parallel — each query's result lands in its own slot —; the list is then put together, rejected, summed and transformed by one
 * thread in query order, so the loop returns exactly what it returns with one thread. */
static int g_icp_threads = 1;
void orc_icp_set_threads(int n) { g_icp_threads = n < 1 ? 1 : (n > 256 ? 256 : n); }

typedef struct {
  const orc_kdtree *tree; const float *work; const float *wnrm; const float *tgt; int lo, hi; int corr_mode, kk;
  int32_t *id; float *d; uint8_t *ok; double *line;
} nn_job;
static void *nn_worker(void *arg) {
  nn_job *j = (nn_job *)arg;
  int32_t *nn_i = (int32_t *)malloc(sizeof(int32_t) * (size_t)j->kk);
  float *nn_d = (float *)malloc(sizeof(float) * (size_t)j->kk);
  for (int i = j->lo; i < j->hi; ++i) {
    const float *q = j->work + 3 * i;
    j->ok[i] = 0;
    if (!finite3(q)) continue;
    if (j->corr_mode == 0) {
      int32_t f;
      orc_kdtree_knn(j->tree, q, 1, 1, &j->id[i], &j->d[i], &f);
      j->ok[i] = f ? 1 : 0;
    } else {
      int32_t f;
      orc_kdtree_knn(j->tree, q, 1, j->kk, nn_i, nn_d, &f);
      if (f <= 0) continue;
      double min_dist = DBL_MAX;
      int min_index = 0;
      const float *nq = j->wnrm + 3 * i;
      for (int m = 0; m < f; ++m) {
        float vx = j->tgt[3 * nn_i[m]] - q[0], vy = j->tgt[3 * nn_i[m] + 1] - q[1], vz = j->tgt[3 * nn_i[m] + 2] - q[2];
        double N[3] = {nq[0], nq[1], nq[2]}, V[3] = {vx, vy, vz};
        double C[3] = {N[1] * V[2] - N[2] * V[1], N[2] * V[0] - N[0] * V[2], N[0] * V[1] - N[1] * V[0]};
        double dist = C[0] * C[0] + C[1] * C[1] + C[2] * C[2];
        if (dist < min_dist) { min_dist = dist; min_index = m; }
      }
      j->id[i] = nn_i[min_index]; j->d[i] = nn_d[min_index]; j->line[i] = min_dist; j->ok[i] = 1;
    }
  }
  free(nn_i); free(nn_d);
  return NULL;
}

int orc_icp(const float *src_xyz, const float *src_nrm, int ns, const float *tgt_xyz, const float *tgt_nrm, int nt,
            const float guess[16], const orc_icp_params *p, float out_T[16], orc_icp_result *res, float *T_hist,
            int32_t *corr_q_out, int32_t *corr_m_out, float *corr_d2_out) {
  return orc_icp_fixed(src_xyz, src_nrm, ns, tgt_xyz, tgt_nrm, nt, guess, p, NULL, NULL, 0, out_T, res, T_hist, corr_q_out, corr_m_out, corr_d2_out);
}

/* The loop with the reference's "fixed correspondences" (setFixedCorrespondences, icp_mod.h:268; unused by its programs):
 *  - 1-NN estimation puts the given pairs in FRONT of the searched ones, whatever their distance, with
 *    distance = (squared distance, float) * 1e10 (correspondence_estimation_mod.hpp:134-162) — so they also enter the MSE the
 *    convergence test looks at, at that scale;
 *  - the normal-shooting estimation only refreshes their distance field (squared distance to the source normal's line) and
 *    does NOT list them (…normal_shooting_weighted.hpp:81-101);
 *  - after the rejectors have run over the list, the FIRST rejector alone is applied to the given pairs once more and the
 *    survivors are appended — a second time for those the list already holds (icp_mod.hpp:210-224; only with a rejector).
 * corr_*_out then need room for ns + 2 * n_fixed entries. */
int orc_icp_fixed(const float *src_xyz, const float *src_nrm, int ns, const float *tgt_xyz, const float *tgt_nrm, int nt,
                  const float guess[16], const orc_icp_params *p, const int32_t *fixed_q, const int32_t *fixed_m, int n_fixed,
                  float out_T[16], orc_icp_result *res, float *T_hist, int32_t *corr_q_out, int32_t *corr_m_out, float *corr_d2_out) {
  static const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  memset(res, 0, sizeof *res);
  memcpy(out_T, I4, sizeof I4);
  /* Registration::initCompute: no target -> error, silent return (registration_mod.hpp:73-77);
   * setInputTarget rejects empty clouds (:60-64). */
  if (nt <= 0 || !tgt_xyz) return -1;
  if (ns <= 0 || !src_xyz) return -2;
  if (!guess) guess = I4;
  int need_src_nrm = (p->corr_mode == 1) || p->use_surface_normal_rej || p->use_self_occluded_rej;
  int need_tgt_nrm = p->use_surface_normal_rej || p->estimator == 1 || p->estimator == 2;
  if ((need_src_nrm && !src_nrm) || (need_tgt_nrm && !tgt_nrm)) return -3;
  if (n_fixed < 0 || (n_fixed > 0 && (!fixed_q || !fixed_m))) return -4;
  for (int f = 0; f < n_fixed; ++f)
    if (fixed_q[f] < 0 || fixed_q[f] >= ns || fixed_m[f] < 0 || fixed_m[f] >= nt) return -4;
  const size_t cap = (size_t)ns + 2 * (size_t)n_fixed;
  float *fixed_d = (float *)malloc(sizeof(float) * (size_t)(n_fixed > 0 ? n_fixed : 1));

  orc_kdtree *tree = orc_kdtree_build(tgt_xyz, nt, 15);
  orc_kdtree *rtree = NULL;

  float *work = (float *)malloc(sizeof(float) * 3 * (size_t)ns);
  float *wnrm = src_nrm ? (float *)malloc(sizeof(float) * 3 * (size_t)ns) : NULL;
  int32_t *cq = (int32_t *)malloc(sizeof(int32_t) * cap);
  int32_t *cm = (int32_t *)malloc(sizeof(int32_t) * cap);
  float *cd = (float *)malloc(sizeof(float) * cap);
  float *ps = (float *)malloc(sizeof(float) * 3 * cap);
  float *pt = (float *)malloc(sizeof(float) * 3 * cap);
  float *pn = (p->estimator == 1 || p->estimator == 2) ? (float *)malloc(sizeof(float) * 3 * cap) : NULL;
  int kk = p->k_normal_shooting > 0 ? p->k_normal_shooting : 1;
  int32_t *nn_i = (int32_t *)malloc(sizeof(int32_t) * (size_t)kk);
  float *nn_d = (float *)malloc(sizeof(float) * (size_t)kk);

  /* threaded search (orc_icp_set_threads): per-query slots; reciprocal search stays on one thread (its tree is rebuilt per iteration) */
  const int n_thr = (g_icp_threads > 1 && !p->use_reciprocal) ? g_icp_threads : 1;
  int32_t *sl_id = NULL; float *sl_d = NULL; uint8_t *sl_ok = NULL; double *sl_line = NULL;
  if (n_thr > 1) {
    sl_id = (int32_t *)malloc(sizeof(int32_t) * (size_t)ns);
    sl_d = (float *)malloc(sizeof(float) * (size_t)ns);
    sl_ok = (uint8_t *)malloc((size_t)ns);
    sl_line = (double *)malloc(sizeof(double) * (size_t)ns);
  }

  float final_T[16], Tk[16];
  double final_Td[16];
  memcpy(final_T, guess, sizeof final_T);
  for (int i = 0; i < 16; ++i) final_Td[i] = guess[i];
  if (!is_identity(guess)) {
    orc_transform_points(src_xyz, ns, guess, work);
    if (wnrm) orc_transform_normals(src_nrm, ns, guess, wnrm);
  } else {
    memcpy(work, src_xyz, sizeof(float) * 3 * (size_t)ns);
    if (wnrm) memcpy(wnrm, src_nrm, sizeof(float) * 3 * (size_t)ns);
  }
  memcpy(Tk, I4, sizeof Tk);

  orc_convergence cc;
  orc_convergence_init(&cc);
  cc.max_iterations = p->max_iterations;
  cc.mse_threshold_relative = p->euclidean_fitness_epsilon;
  cc.translation_threshold = p->transformation_epsilon;
  cc.rotation_threshold = 1.0 - p->transformation_epsilon;
  cc.mse_threshold_absolute = p->mse_threshold_absolute;
  cc.failure_after_max_iter = p->failure_after_max_iter;

  int iterations = 0, converged = 0, ncorr = 0;
  double max_d2 = p->max_corr_dist * p->max_corr_dist;
  do {
    ncorr = 0;
    if (p->use_reciprocal && !rtree) { /* source tree rebuilt every iteration: the dirty flag is set by setInputSource */ }
    if (p->use_reciprocal) {
      if (rtree) orc_kdtree_free(rtree);
      rtree = orc_kdtree_build(work, ns, 15);
    }
    for (int f = 0; f < n_fixed; ++f) {
      const float *q = work + 3 * fixed_q[f], *t = tgt_xyz + 3 * fixed_m[f];
      const float vx = t[0] - q[0], vy = t[1] - q[1], vz = t[2] - q[2];
      if (p->corr_mode == 0) {
        const float d2 = vx * vx + vy * vy + vz * vz;
        fixed_d[f] = (float)((double)d2 * 1e10);
        cq[ncorr] = fixed_q[f]; cm[ncorr] = fixed_m[f]; cd[ncorr] = fixed_d[f]; ++ncorr;
      } else {
        const float *nq = wnrm + 3 * fixed_q[f];
        const double N[3] = {nq[0], nq[1], nq[2]}, V[3] = {vx, vy, vz};
        const double Cx = N[1] * V[2] - N[2] * V[1], Cy = N[2] * V[0] - N[0] * V[2], Cz = N[0] * V[1] - N[1] * V[0];
        fixed_d[f] = (float)(Cx * Cx + Cy * Cy + Cz * Cz);
      }
    }
    if (n_thr > 1) {
      nn_job jobs[256];
      pthread_t th[256];
      char started[256];
      for (int t = 0; t < n_thr; ++t) {
        const long long a = (long long)ns * t / n_thr, b = (long long)ns * (t + 1) / n_thr;
        jobs[t] = (nn_job){tree, work, wnrm, tgt_xyz, (int)a, (int)b, p->corr_mode, kk, sl_id, sl_d, sl_ok, sl_line};
        started[t] = pthread_create(&th[t], NULL, nn_worker, &jobs[t]) == 0;
        if (!started[t]) nn_worker(&jobs[t]);
      }
      for (int t = 0; t < n_thr; ++t)
        if (started[t]) pthread_join(th[t], NULL);
      /* the list in query order, with the tests of the one-thread loop below */
      for (int i = 0; i < ns; ++i) {
        if (!sl_ok[i]) continue;
        if (p->corr_mode == 0) { if ((double)sl_d[i] > max_d2) continue; }
        else if (sl_line[i] > p->max_corr_dist) continue;      /* quirk Q2, as below */
        cq[ncorr] = i; cm[ncorr] = sl_id[i]; cd[ncorr] = sl_d[i]; ++ncorr;
      }
    } else
    for (int i = 0; i < ns; ++i) {
      const float *q = work + 3 * i;
      if (!finite3(q)) continue;
      if (p->corr_mode == 0) {
        int32_t id; float d; int32_t f;
        orc_kdtree_knn(tree, q, 1, 1, &id, &d, &f);
        if (!f || (double)d > max_d2) continue;
        if (p->use_reciprocal) {
          int32_t rid; float rd; int32_t rf;
          orc_kdtree_knn(rtree, tgt_xyz + 3 * id, 1, 1, &rid, &rd, &rf);
          if (!rf || (double)rd > max_d2 || rid != i) continue;
        }
        cq[ncorr] = i; cm[ncorr] = id; cd[ncorr] = d; ++ncorr;
      } else {
        int32_t f;
        orc_kdtree_knn(tree, q, 1, kk, nn_i, nn_d, &f);
        if (f <= 0) continue;
        double min_dist = DBL_MAX;
        int min_index = 0;
        const float *nq = wnrm + 3 * i;
        for (int j = 0; j < f; ++j) {
          /* pt = target - source in float, then double cross product */
          float vx = tgt_xyz[3 * nn_i[j]] - q[0], vy = tgt_xyz[3 * nn_i[j] + 1] - q[1], vz = tgt_xyz[3 * nn_i[j] + 2] - q[2];
          double N[3] = {nq[0], nq[1], nq[2]}, V[3] = {vx, vy, vz};
          double C[3] = {N[1] * V[2] - N[2] * V[1], N[2] * V[0] - N[0] * V[2], N[0] * V[1] - N[1] * V[0]};
          double dist = C[0] * C[0] + C[1] * C[1] + C[2] * C[2];
          if (dist < min_dist) { min_dist = dist; min_index = j; }
        }
        /* quirk Q2: squared line distance compared against the UNSQUARED max distance */
        if (min_dist > p->max_corr_dist) continue;
        cq[ncorr] = i; cm[ncorr] = nn_i[min_index]; cd[ncorr] = nn_d[min_index]; ++ncorr;
      }
    }
    /* rejectors, in the order they were added (poseestimator.cpp:334-337) */
    if (p->use_surface_normal_rej) {
      int m = 0;
      for (int c = 0; c < ncorr; ++c) {
        const float *a = wnrm + 3 * cq[c], *b = tgt_nrm + 3 * cm[c];
        double score = (double)((a[0] * b[0]) + (a[1] * b[1]) + (a[2] * b[2]));
        if (score > p->surface_normal_thr) { cq[m] = cq[c]; cm[m] = cm[c]; cd[m] = cd[c]; ++m; }
      }
      ncorr = m;
    }
    if (p->use_self_occluded_rej) {
      int m = 0;
      for (int c = 0; c < ncorr; ++c) {
        const float *a = wnrm + 3 * cq[c], *pp = work + 3 * cq[c];
        double s = sqrt((double)(pp[0] * pp[0] + pp[1] * pp[1] + pp[2] * pp[2]));
        double score = (double)((a[0] * (-pp[0] / s)) + (a[1] * (-pp[1] / s)) + (a[2] * (-pp[2] / s)));
        if (score > p->self_occluded_thr) { cq[m] = cq[c]; cm[m] = cm[c]; cd[m] = cd[c]; ++m; }
      }
      ncorr = m;
    }
    if (n_fixed > 0 && (p->use_surface_normal_rej || p->use_self_occluded_rej)) {
      /* "Apply the first rejector on the fixed correspondances" (icp_mod.hpp:210-224) */
      for (int f = 0; f < n_fixed; ++f) {
        const float *a = wnrm + 3 * fixed_q[f];
        double score;
        int keep;
        if (p->use_surface_normal_rej) {
          const float *b = tgt_nrm + 3 * fixed_m[f];
          score = (double)((a[0] * b[0]) + (a[1] * b[1]) + (a[2] * b[2]));
          keep = score > p->surface_normal_thr;
        } else {
          const float *pp = work + 3 * fixed_q[f];
          const double sl = sqrt((double)(pp[0] * pp[0] + pp[1] * pp[1] + pp[2] * pp[2]));
          score = (double)((a[0] * (-pp[0] / sl)) + (a[1] * (-pp[1] / sl)) + (a[2] * (-pp[2] / sl)));
          keep = score > p->self_occluded_thr;
        }
        if (keep) { cq[ncorr] = fixed_q[f]; cm[ncorr] = fixed_m[f]; cd[ncorr] = fixed_d[f]; ++ncorr; }
      }
    }
    if (ncorr < p->min_correspondences) {
      cc.state = ORC_CONV_NO_CORRESPONDENCES;
      converged = 0;
      break;
    }
    for (int c = 0; c < ncorr; ++c) {
      memcpy(ps + 3 * c, work + 3 * cq[c], 3 * sizeof(float));
      memcpy(pt + 3 * c, tgt_xyz + 3 * cm[c], 3 * sizeof(float));
      if (pn) memcpy(pn + 3 * c, tgt_nrm + 3 * cm[c], 3 * sizeof(float));
    }
    if (p->estimator == 1) orc_point_to_plane_lls(ps, pt, pn, ncorr, Tk);
    else if (p->estimator == 2) orc_point_to_plane_lm(ps, pt, pn, ncorr, p->lm_precision, Tk, NULL, NULL, NULL);   /* regmeshpcd.cpp:162,193 */
    else orc_umeyama(ps, pt, ncorr, p->acc_mode, Tk);
    if (p->transform_mode == 0) {
      orc_transform_points(work, ns, Tk, work);
      if (wnrm) orc_transform_normals(wnrm, ns, Tk, wnrm);
      mat4_mul_f(Tk, final_T, final_T);
    } else {
      double Tkd[16];
      for (int i = 0; i < 16; ++i) Tkd[i] = Tk[i];
      mat4_mul_d(Tkd, final_Td, final_Td);
      for (int i = 0; i < 16; ++i) final_T[i] = (float)final_Td[i];
      orc_transform_points(src_xyz, ns, final_T, work);
      if (wnrm) orc_transform_normals(src_nrm, ns, final_T, wnrm);
    }
    if (T_hist) memcpy(T_hist + 16 * (size_t)iterations, final_T, sizeof final_T);
    ++iterations;
    double mse = 0;
    for (int c = 0; c < ncorr; ++c) mse += cd[c];
    mse /= (double)ncorr;
    converged = orc_convergence_step(&cc, iterations, Tk, mse);
  } while (!converged);

  memcpy(out_T, final_T, sizeof final_T);
  res->iterations = iterations;
  res->converged = converged;
  res->state = cc.state;
  res->last_mse = cc.cur_mse;
  res->n_corr = ncorr;
  res->align_strength = (double)ncorr / (double)(ns + nt);
  res->fitness = orc_fitness(src_xyz, ns, tgt_xyz, nt, final_T, DBL_MAX, NULL);
  if (corr_q_out) memcpy(corr_q_out, cq, sizeof(int32_t) * (size_t)ncorr);
  if (corr_m_out) memcpy(corr_m_out, cm, sizeof(int32_t) * (size_t)ncorr);
  if (corr_d2_out) memcpy(corr_d2_out, cd, sizeof(float) * (size_t)ncorr);

  free(fixed_d);
  free(sl_id); free(sl_d); free(sl_ok); free(sl_line);
  free(work); free(wnrm); free(cq); free(cm); free(cd); free(ps); free(pt); free(pn); free(nn_i); free(nn_d);
  orc_kdtree_free(tree);
  if (rtree) orc_kdtree_free(rtree);
  return 0;
}
