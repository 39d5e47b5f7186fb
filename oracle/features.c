/*
 * features.c — CPU ORACLE (test infrastructure, not product code; parity unpinned,
 * see ope_oracle.h).
 *
 * Restates the un-vendored PCL 1.7.x primitives the reference's coarse stage
 * calls (published algorithms; call sites in DetectAndLocalize/src/poseestimator.cpp):
 *   pcl::NormalEstimation::compute              :151-156  (normal_3d.hpp, centroid.hpp
 *       computeMeanAndCovarianceMatrix single-pass float, eigen.hpp eigen33,
 *       flipNormalTowardsViewpoint)
 *   pcl::FPFHEstimation::compute                :121-125  (fpfh.hpp, pfh.cpp computePairFeatures)
 *   pcl::UniformSampling::compute               :141-145  (uniform_sampling.hpp, PCL<=1.7 keypoints API)
 *   pcl::SampleConsensusInitialAlignment::align :50-64    (ia_ransac.hpp)
 */
#include "libm_f32.h"
#include "ope_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* eigen33: smallest eigenvalue/eigenvector of a symmetric 3x3 (float). */
/* ------------------------------------------------------------------ */
static void compute_roots2(float b, float c, float roots[3]) {
  roots[0] = 0.f;
  float d = (float)(b * b - 4.0 * c);
  if (d < 0.0f) d = 0.0f;
  float sd = sqrtf(d);
  roots[2] = 0.5f * (b + sd);
  roots[1] = 0.5f * (b - sd);
}

static void compute_roots(const float m[9], float roots[3]) {
  float c0 = m[0] * m[4] * m[8] + 2.f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
  float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
  float c2 = m[0] + m[4] + m[8];
  if (fabsf(c0) < FLT_EPSILON) {
    compute_roots2(c2, c1, roots);
    return;
  }
  const float s_inv3 = (float)(1.0 / 3.0);
  const float s_sqrt3 = sqrtf(3.0f);
  float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.f) a_over_3 = 0.f;
  float half_b = 0.5f * (c0 + c2_over_3 * (2.f * c2_over_3 * c2_over_3 - c1));
  float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.f) q = 0.f;
  float rho = sqrtf(-a_over_3);
  float theta = lmf_atan2f(sqrtf(-q), half_b) * s_inv3;   /* libm_f32.h: the same bits on the CPU and on the GPU */
  float cos_theta, sin_theta;
  lmf_cos_sin_small(theta, &cos_theta, &sin_theta);
  roots[0] = c2_over_3 + 2.f * rho * cos_theta;
  roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  float t;
  if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
  if (roots[1] >= roots[2]) {
    t = roots[1]; roots[1] = roots[2]; roots[2] = t;
    if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
  }
  if (roots[0] <= 0.f) compute_roots2(c2, c1, roots);
}

static void cross3f(const float a[3], const float b[3], float c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* mat: symmetric row-major 3x3 */
static void eigen33(const float mat[9], float *eigenvalue, float evec[3]) {
  float scale = 0.f;
  for (int i = 0; i < 9; ++i)
    if (fabsf(mat[i]) > scale) scale = fabsf(mat[i]);
  if (scale <= FLT_MIN) scale = 1.0f;
  float sm[9];
  for (int i = 0; i < 9; ++i) sm[i] = mat[i] / scale;
  float roots[3];
  compute_roots(sm, roots);
  *eigenvalue = roots[0] * scale;
  sm[0] -= roots[0]; sm[4] -= roots[0]; sm[8] -= roots[0];
  float v1[3], v2[3], v3[3];
  cross3f(sm + 0, sm + 3, v1);
  cross3f(sm + 0, sm + 6, v2);
  cross3f(sm + 3, sm + 6, v3);
  float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
  float l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
  float l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
  const float *v; float l;
  if (l1 >= l2 && l1 >= l3) { v = v1; l = l1; }
  else if (l2 >= l1 && l2 >= l3) { v = v2; l = l2; }
  else { v = v3; l = l3; }
  float s = sqrtf(l);
  evec[0] = v[0] / s; evec[1] = v[1] / s; evec[2] = v[2] / s;
}

void orc_normals_knn(const float *xyz, int n, int k, const float vp[3], float *out_nrm, float *out_curv) {
  orc_kdtree *tree = orc_kdtree_build(xyz, n, 15);
  int32_t *nn = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
  float *nd = (float *)malloc(sizeof(float) * (size_t)k);
  const float qnan = NAN;
  for (int i = 0; i < n; ++i) {
    int32_t found = 0;
    orc_kdtree_knn(tree, xyz + 3 * i, 1, k, nn, nd, &found);
    if (found < 3) {
      out_nrm[3 * i] = out_nrm[3 * i + 1] = out_nrm[3 * i + 2] = qnan;
      if (out_curv) out_curv[i] = qnan;
      continue;
    }
    /* computeMeanAndCovarianceMatrix: single pass, float, neighbours in k-NN order */
    float accu[9] = {0};
    for (int j = 0; j < found; ++j) {
      const float *p = xyz + 3 * nn[j];
      accu[0] += p[0] * p[0]; accu[1] += p[0] * p[1]; accu[2] += p[0] * p[2];
      accu[3] += p[1] * p[1]; accu[4] += p[1] * p[2]; accu[5] += p[2] * p[2];
      accu[6] += p[0]; accu[7] += p[1]; accu[8] += p[2];
    }
    float fc = (float)found;
    for (int a = 0; a < 9; ++a) accu[a] /= fc;
    float cov[9];
    cov[0] = accu[0] - accu[6] * accu[6];
    cov[1] = accu[1] - accu[6] * accu[7];
    cov[2] = accu[2] - accu[6] * accu[8];
    cov[4] = accu[3] - accu[7] * accu[7];
    cov[5] = accu[4] - accu[7] * accu[8];
    cov[8] = accu[5] - accu[8] * accu[8];
    cov[3] = cov[1]; cov[6] = cov[2]; cov[7] = cov[5];
    float ev, nv[3];
    eigen33(cov, &ev, nv);
    float eig_sum = cov[0] + cov[4] + cov[8];
    float curv = (eig_sum != 0.f) ? fabsf(ev / eig_sum) : 0.f;
    /* flipNormalTowardsViewpoint */
    float vx = vp[0] - xyz[3 * i], vy = vp[1] - xyz[3 * i + 1], vz = vp[2] - xyz[3 * i + 2];
    float cos_theta = vx * nv[0] + vy * nv[1] + vz * nv[2];
    if (cos_theta < 0) { nv[0] *= -1; nv[1] *= -1; nv[2] *= -1; }
    out_nrm[3 * i] = nv[0]; out_nrm[3 * i + 1] = nv[1]; out_nrm[3 * i + 2] = nv[2];
    if (out_curv) out_curv[i] = curv;
  }
  free(nn); free(nd);
  orc_kdtree_free(tree);
}

/* ------------------------------------------------------------------ */
int orc_pair_features(const float p1[3], const float n1[3], const float p2[3], const float n2[3],
                      float *f1, float *f2, float *f3, float *f4) {
  float dp[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  *f4 = sqrtf(dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2]);
  if (*f4 == 0.0f) { *f1 = *f2 = *f3 = *f4 = 0.f; return 0; }
  float a[3] = {n1[0], n1[1], n1[2]}, b[3] = {n2[0], n2[1], n2[2]};
  float angle1 = (a[0] * dp[0] + a[1] * dp[1] + a[2] * dp[2]) / *f4;
  float angle2 = (b[0] * dp[0] + b[1] * dp[1] + b[2] * dp[2]) / *f4;
  if (lmf_acosf(fabsf(angle1)) > lmf_acosf(fabsf(angle2))) {   /* (libm_f32.h: glibc's bits, on any machine) */
    for (int d = 0; d < 3; ++d) { a[d] = n2[d]; b[d] = n1[d]; dp[d] *= -1.f; }
    *f3 = -angle2;
  } else
    *f3 = angle1;
  float v[3];
  cross3f(dp, a, v);
  float vn = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (vn == 0.0f) { *f1 = *f2 = *f3 = *f4 = 0.f; return 0; }
  v[0] /= vn; v[1] /= vn; v[2] /= vn;
  float w[3];
  cross3f(a, v, w);
  *f2 = v[0] * b[0] + v[1] * b[1] + v[2] * b[2];
  *f1 = lmf_atan2f(w[0] * b[0] + w[1] * b[1] + w[2] * b[2], a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
  return 1;
}

void orc_fpfh(const float *xyz, const float *nrm, int n, float radius, float *out33, float *spfh33_opt,
              double *mean_neighbours_opt) {
  const int NB = 11;
  orc_kdtree *tree = orc_kdtree_build(xyz, n, 15);
  int64_t *offs = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
  int64_t total = orc_kdtree_radius(tree, xyz, n, radius, 0, offs, NULL, NULL, 0);
  int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total > 0 ? total : 1));
  float *d2 = (float *)malloc(sizeof(float) * (size_t)(total > 0 ? total : 1));
  /* UNSORTED neighbour lists: the reference sets no search method on its FPFHEstimation (poseestimator.cpp:121-125), so
   * Feature::initCompute makes a pcl::search::KdTree(sorted = false) [uPCL-recall] and pass 2 adds the weighted rows in the
   * kd-tree's own order */
  orc_kdtree_radius(tree, xyz, n, radius, 0, offs, idx, d2, total);
  float *spfh = (float *)calloc((size_t)n * 33, sizeof(float));
  const float d_pi = 1.0f / (2.0f * (float)M_PI);
  /* pass 1: SPFH */
  for (int i = 0; i < n; ++i) {
    int64_t b = offs[i], e = offs[i + 1];
    int64_t m = e - b;
    if (m == 0) continue;
    float hist_incr = 100.0f / (float)(m - 1);
    float *h = spfh + (size_t)i * 33;
    for (int64_t j = b; j < e; ++j) {
      if (idx[j] == i) continue;
      float f1, f2, f3, f4;
      if (!orc_pair_features(xyz + 3 * i, nrm + 3 * i, xyz + 3 * idx[j], nrm + 3 * idx[j], &f1, &f2, &f3, &f4)) continue;
      int hi = (int)floor(NB * ((f1 + M_PI) * d_pi));
      if (hi < 0) hi = 0;
      if (hi >= NB) hi = NB - 1;
      h[hi] += hist_incr;
      hi = (int)floor(NB * ((f2 + 1.0) * 0.5));
      if (hi < 0) hi = 0;
      if (hi >= NB) hi = NB - 1;
      h[NB + hi] += hist_incr;
      hi = (int)floor(NB * ((f3 + 1.0) * 0.5));
      if (hi < 0) hi = 0;
      if (hi >= NB) hi = NB - 1;
      h[2 * NB + hi] += hist_incr;
    }
  }
  /* pass 2: weighting */
  for (int i = 0; i < n; ++i) {
    int64_t b = offs[i], e = offs[i + 1];
    float *o = out33 + (size_t)i * 33;
    if (e == b) {
      for (int d = 0; d < 33; ++d) o[d] = NAN;
      continue;
    }
    for (int d = 0; d < 33; ++d) o[d] = 0.f;
    double sum[3] = {0, 0, 0};
    for (int64_t j = b; j < e; ++j) {
      if (d2[j] == 0) continue;
      float w = 1.0f / d2[j];
      const float *h = spfh + (size_t)idx[j] * 33;
      for (int g = 0; g < 3; ++g)
        for (int f = 0; f < NB; ++f) {
          float val = h[g * NB + f] * w;
          sum[g] += val;
          o[g * NB + f] += val;
        }
    }
    for (int g = 0; g < 3; ++g) {
      if (sum[g] != 0) sum[g] = 100.0 / sum[g];
      for (int f = 0; f < NB; ++f) o[g * NB + f] *= (float)sum[g];
    }
  }
  if (spfh33_opt) memcpy(spfh33_opt, spfh, sizeof(float) * 33 * (size_t)n);
  if (mean_neighbours_opt) *mean_neighbours_opt = n > 0 ? (double)total / n : 0;
  free(offs); free(idx); free(d2); free(spfh);
  orc_kdtree_free(tree);
}

/* ------------------------------------------------------------------ */
typedef struct { int64_t key; int32_t idx; } us_leaf;
static int us_cmp(const void *a, const void *b) {
  int64_t ka = ((const us_leaf *)a)->key, kb = ((const us_leaf *)b)->key;
  return ka < kb ? -1 : (ka > kb ? 1 : 0);
}

int orc_uniform_sampling(const float *xyz, int n, float leaf, int32_t *out_idx) {
  if (n <= 0) return 0;
  float inv = 1.0f / leaf;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  int any = 0;
  for (int i = 0; i < n; ++i) {
    const float *p = xyz + 3 * i;
    if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
    any = 1;
    for (int d = 0; d < 3; ++d) { if (p[d] < mn[d]) mn[d] = p[d]; if (p[d] > mx[d]) mx[d] = p[d]; }
  }
  if (!any) return 0;
  int64_t min_b[3], div_b[3];
  for (int d = 0; d < 3; ++d) {
    min_b[d] = (int64_t)floorf(mn[d] * inv);
    div_b[d] = (int64_t)floorf(mx[d] * inv) - min_b[d] + 1;
  }
  int64_t mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
  /* sort (key, idx) pairs by key, stable in idx via composite compare, then scan */
  us_leaf *lv = (us_leaf *)malloc(sizeof(us_leaf) * (size_t)n);
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const float *p = xyz + 3 * i;
    if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
    int64_t ijk[3];
    for (int d = 0; d < 3; ++d) ijk[d] = (int64_t)floorf(p[d] * inv);
    lv[m].key = (ijk[0] - min_b[0]) * mul[0] + (ijk[1] - min_b[1]) * mul[1] + (ijk[2] - min_b[2]) * mul[2];
    lv[m].idx = i;
    ++m;
  }
  /* the per-leaf winner depends on input order (strict <, first wins ties): keep
   * input order inside each key by sorting on (key, idx) */
  for (int i = 0; i < m; ++i) lv[i].key = lv[i].key * (int64_t)n + lv[i].idx; /* composite */
  qsort(lv, (size_t)m, sizeof(us_leaf), us_cmp);
  int cnt = 0;
  int i = 0;
  while (i < m) {
    int64_t key = lv[i].key / n;
    int32_t best = lv[i].idx;
    const float *pb = xyz + 3 * best;
    float ijk[3];
    for (int d = 0; d < 3; ++d) ijk[d] = floorf(pb[d] * inv);
    int j = i + 1;
    while (j < m && lv[j].key / n == key) {
      const float *pc = xyz + 3 * lv[j].idx;
      pb = xyz + 3 * best;
      /* quirk Q7: metric coordinates minus integer voxel coordinates; w: (1-0)^2 */
      float dc = (pc[0] - ijk[0]) * (pc[0] - ijk[0]) + (pc[1] - ijk[1]) * (pc[1] - ijk[1]) +
                 (pc[2] - ijk[2]) * (pc[2] - ijk[2]) + 1.0f;
      float dp = (pb[0] - ijk[0]) * (pb[0] - ijk[0]) + (pb[1] - ijk[1]) * (pb[1] - ijk[1]) +
                 (pb[2] - ijk[2]) * (pb[2] - ijk[2]) + 1.0f;
      if (dc < dp) best = lv[j].idx;
      ++j;
    }
    out_idx[cnt++] = best;
    i = j;
  }
  free(lv);
  return cnt;
}

/* ------------------------------------------------------------------ */
void orc_feature_knn(const float *feat33, int n, const float *q33, int nq, int k, int32_t *idx, float *d2) {
  for (int qi = 0; qi < nq; ++qi) {
    int32_t *oi = idx + (size_t)qi * k;
    float *od = d2 + (size_t)qi * k;
    int cnt = 0;
    const float *q = q33 + (size_t)qi * 33;
    for (int j = 0; j < n; ++j) {
      const float *f = feat33 + (size_t)j * 33;
      float d = 0.f;
      for (int c = 0; c < 33; ++c) { float t = q[c] - f[c]; d += t * t; }
      if (!(d == d)) continue; /* NaN descriptors never match */
      int pos;
      if (cnt < k) pos = cnt++;
      else if (d < od[k - 1]) pos = k - 1;
      else continue;
      while (pos > 0 && od[pos - 1] > d) { od[pos] = od[pos - 1]; oi[pos] = oi[pos - 1]; --pos; }
      od[pos] = d; oi[pos] = j;
    }
    for (int j = cnt; j < k; ++j) { oi[j] = -1; od[j] = INFINITY; }
  }
}

double orc_sacia_error(const float *src_xyz, int ns, const orc_kdtree *tgt_tree, const float T[16],
                       double corr_dist_threshold) {
  /* computeErrorMetric with TruncatedError(threshold): float accumulation */
  float thr = (float)corr_dist_threshold;
  float error = 0.f;
  for (int i = 0; i < ns; ++i) {
    float p[3];
    orc_transform_points(src_xyz + 3 * i, 1, T, p);
    int32_t id; float d; int32_t f;
    orc_kdtree_knn(tgt_tree, p, 1, 1, &id, &d, &f);
    float e = (f && d <= thr) ? d / thr : 1.0f;
    error += e;
  }
  return (double)error;
}

/* injectable RNG: 64-bit LCG (Knuth MMIX), top 53 bits -> [0,1) */
static double lcg_next(uint64_t *s) {
  *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
  return (double)(*s >> 11) * (1.0 / 9007199254740992.0);
}
static int rnd_index(uint64_t *s, int n) { return (int)(n * lcg_next(s)); }

int orc_sacia(const float *src_xyz, const float *src_feat33, int ns, const float *tgt_xyz, const float *tgt_feat33,
              int nt, int n_iter, int nr_samples, int k_corr, double max_corr_dist, float min_sample_dist,
              uint64_t seed, const int32_t *forced_samples, float out_T[16], double *best_err, int32_t *best_iter) {
  static const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  memcpy(out_T, I4, sizeof I4);
  if (best_err) *best_err = 0;
  if (best_iter) *best_iter = -1;
  if (ns < nr_samples || nt < 1 || nr_samples < 1) return -1;
  orc_kdtree *tree = orc_kdtree_build(tgt_xyz, nt, 15);
  int32_t *samp = (int32_t *)malloc(sizeof(int32_t) * (size_t)nr_samples);
  int32_t *corr = (int32_t *)malloc(sizeof(int32_t) * (size_t)nr_samples);
  int32_t *nn = (int32_t *)malloc(sizeof(int32_t) * (size_t)k_corr);
  float *nd = (float *)malloc(sizeof(float) * (size_t)k_corr);
  float *ps = (float *)malloc(sizeof(float) * 3 * (size_t)nr_samples);
  float *pt = (float *)malloc(sizeof(float) * 3 * (size_t)nr_samples);
  uint64_t rng = seed;
  float lowest = 0.f;
  for (int it = 0; it < n_iter; ++it) {
    /* selectSamples receives min_sample_distance by value: a halving lasts for one hypothesis only */
    float msd = min_sample_dist;
    if (forced_samples) {
      memcpy(samp, forced_samples + (size_t)it * nr_samples, sizeof(int32_t) * (size_t)nr_samples);
      memcpy(corr, forced_samples + (size_t)n_iter * nr_samples + (size_t)it * nr_samples,
             sizeof(int32_t) * (size_t)nr_samples);
    } else {
      /* selectSamples */
      int cnt = 0, without = 0;
      int max_without = 3 * ns;
      while (cnt < nr_samples) {
        int si = rnd_index(&rng, ns);
        int valid = 1;
        for (int i = 0; i < cnt; ++i) {
          float dx = src_xyz[3 * si] - src_xyz[3 * samp[i]], dy = src_xyz[3 * si + 1] - src_xyz[3 * samp[i] + 1],
                dz = src_xyz[3 * si + 2] - src_xyz[3 * samp[i] + 2];
          float dist = sqrtf(dx * dx + dy * dy + dz * dz);
          if (si == samp[i] || dist < msd) { valid = 0; break; }
        }
        if (valid) { samp[cnt++] = si; without = 0; }
        else ++without;
        if (without >= max_without) { msd *= 0.5f; without = 0; }
      }
      /* findSimilarFeatures */
      for (int i = 0; i < nr_samples; ++i) {
        orc_feature_knn(tgt_feat33, nt, src_feat33 + (size_t)samp[i] * 33, 1, k_corr, nn, nd);
        int r = rnd_index(&rng, k_corr);
        corr[i] = nn[r] >= 0 ? nn[r] : nn[0];
      }
    }
    for (int i = 0; i < nr_samples; ++i) {
      memcpy(ps + 3 * i, src_xyz + 3 * samp[i], 3 * sizeof(float));
      memcpy(pt + 3 * i, tgt_xyz + 3 * corr[i], 3 * sizeof(float));
    }
    float T[16];
    orc_umeyama(ps, pt, nr_samples, 0, T);
    float err = (float)orc_sacia_error(src_xyz, ns, tree, T, max_corr_dist);
    if (it == 0 || err < lowest) {
      lowest = err;
      memcpy(out_T, T, sizeof T);
      if (best_iter) *best_iter = it;
    }
  }
  if (best_err) *best_err = lowest;
  free(samp); free(corr); free(nn); free(nd); free(ps); free(pt);
  orc_kdtree_free(tree);
  return 0;
}
