/*
 * kdtree.c — CPU ORACLE (test infrastructure, not product code; parity unpinned,
 * see ope_oracle.h).
 *
 * Exact k-NN / radius search on 3-D float points.  Restates the published
 * algorithm PCL delegates to (pcl::KdTreeFLANN -> flann::KDTreeSingleIndex,
 * Arya & Mount style): points reordered into leaf buckets (leaf_max_size 15 in
 * PCL), bounding-box tracking, sliding-midpoint split on the widest dimension,
 * search with incremental per-dimension box distances, eps = 0 (exact).
 * FLANN itself is an un-vendored dependency of the reference; call sites:
 * impl/correspondence_estimation_mod.hpp:170, impl/registration_mod.hpp:149.
 */
#include "ope_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  /* leaf: left = first point, right = one past last (indices into reordered
   * arrays), dim = -1.  internal: child indices, split plane. */
  int32_t left, right;
  int32_t dim;
  float div_low, div_high;
} kd_node;

struct orc_kdtree {
  int n;
  int leaf_max;
  float *pts;    /* reordered xyz, 3*n */
  int32_t *perm; /* reordered position -> original index */
  kd_node *nodes;
  int n_nodes, cap_nodes;
  float bb_lo[3], bb_hi[3];
};

static int new_node(orc_kdtree *t) {
  if (t->n_nodes == t->cap_nodes) {
    t->cap_nodes = t->cap_nodes ? 2 * t->cap_nodes : 1024;
    t->nodes = (kd_node *)realloc(t->nodes, sizeof(kd_node) * (size_t)t->cap_nodes);
  }
  return t->n_nodes++;
}

/* work arrays during build: idx[] holds original indices, src the original xyz */
typedef struct {
  const float *src;
  int32_t *idx;
} build_ctx;

static void plane_split(build_ctx *b, int lo, int count, int dim, float val, int *lim1, int *lim2) {
  int32_t *ind = b->idx + lo;
  int left = 0, right = count - 1;
  for (;;) {
    while (left <= right && b->src[3 * ind[left] + dim] < val) ++left;
    while (left <= right && b->src[3 * ind[right] + dim] >= val) --right;
    if (left > right) break;
    int32_t tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp;
    ++left; --right;
  }
  *lim1 = left;
  right = count - 1;
  for (;;) {
    while (left <= right && b->src[3 * ind[left] + dim] <= val) ++left;
    while (left <= right && b->src[3 * ind[right] + dim] > val) --right;
    if (left > right) break;
    int32_t tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp;
    ++left; --right;
  }
  *lim2 = left;
}

static int divide(orc_kdtree *t, build_ctx *b, int lo, int hi, float bb_lo[3], float bb_hi[3]) {
  int me = new_node(t);
  int count = hi - lo;
  if (count <= t->leaf_max) {
    t->nodes[me].left = lo;
    t->nodes[me].right = hi;
    t->nodes[me].dim = -1;
    /* tighten bbox to the leaf's points */
    for (int d = 0; d < 3; ++d) { bb_lo[d] = FLT_MAX; bb_hi[d] = -FLT_MAX; }
    for (int i = lo; i < hi; ++i)
      for (int d = 0; d < 3; ++d) {
        float v = b->src[3 * b->idx[i] + d];
        if (v < bb_lo[d]) bb_lo[d] = v;
        if (v > bb_hi[d]) bb_hi[d] = v;
      }
    return me;
  }
  /* sliding midpoint on the widest bbox dimension (ties: widest data spread) */
  const float EPS = 0.00001f;
  float max_span = bb_hi[0] - bb_lo[0];
  for (int d = 1; d < 3; ++d)
    if (bb_hi[d] - bb_lo[d] > max_span) max_span = bb_hi[d] - bb_lo[d];
  float max_spread = -1.f;
  int cut = 0;
  float cmin = 0, cmax = 0;
  for (int d = 0; d < 3; ++d) {
    if (bb_hi[d] - bb_lo[d] > (1.f - EPS) * max_span) {
      float mn = FLT_MAX, mx = -FLT_MAX;
      for (int i = lo; i < hi; ++i) {
        float v = b->src[3 * b->idx[i] + d];
        if (v < mn) mn = v;
        if (v > mx) mx = v;
      }
      if (mx - mn > max_spread) { max_spread = mx - mn; cut = d; cmin = mn; cmax = mx; }
    }
  }
  float split = (bb_lo[cut] + bb_hi[cut]) * 0.5f;
  if (split < cmin) split = cmin;
  else if (split > cmax) split = cmax;
  int lim1, lim2, index;
  plane_split(b, lo, count, cut, split, &lim1, &lim2);
  if (lim1 > count / 2) index = lim1;
  else if (lim2 < count / 2) index = lim2;
  else index = count / 2;
  if (index == 0 || index == count) index = count / 2; /* all-equal coordinates */

  float lbb_lo[3], lbb_hi[3], rbb_lo[3], rbb_hi[3];
  memcpy(lbb_lo, bb_lo, sizeof lbb_lo); memcpy(lbb_hi, bb_hi, sizeof lbb_hi);
  memcpy(rbb_lo, bb_lo, sizeof rbb_lo); memcpy(rbb_hi, bb_hi, sizeof rbb_hi);
  lbb_hi[cut] = split;
  rbb_lo[cut] = split;
  int l = divide(t, b, lo, lo + index, lbb_lo, lbb_hi);
  int r = divide(t, b, lo + index, hi, rbb_lo, rbb_hi);
  kd_node *nd = &t->nodes[me];
  nd->left = l;
  nd->right = r;
  nd->dim = cut;
  nd->div_low = lbb_hi[cut];
  nd->div_high = rbb_lo[cut];
  for (int d = 0; d < 3; ++d) {
    bb_lo[d] = lbb_lo[d] < rbb_lo[d] ? lbb_lo[d] : rbb_lo[d];
    bb_hi[d] = lbb_hi[d] > rbb_hi[d] ? lbb_hi[d] : rbb_hi[d];
  }
  return me;
}

orc_kdtree *orc_kdtree_build(const float *xyz, int n, int leaf_max) {
  orc_kdtree *t = (orc_kdtree *)calloc(1, sizeof *t);
  t->n = n;
  t->leaf_max = leaf_max > 0 ? leaf_max : 15;
  t->pts = (float *)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
  t->perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  if (n <= 0) return t;
  /* non-finite points are left out of the index (PCL builds the FLANN index
   * over finite points only when !is_dense, kdtree_flann.hpp convertCloudToArray) */
  int m = 0;
  for (int i = 0; i < n; ++i)
    if (isfinite(xyz[3 * i]) && isfinite(xyz[3 * i + 1]) && isfinite(xyz[3 * i + 2])) t->perm[m++] = i;
  t->n = m;
  if (m == 0) return t;
  for (int d = 0; d < 3; ++d) { t->bb_lo[d] = FLT_MAX; t->bb_hi[d] = -FLT_MAX; }
  for (int i = 0; i < m; ++i)
    for (int d = 0; d < 3; ++d) {
      float v = xyz[3 * t->perm[i] + d];
      if (v < t->bb_lo[d]) t->bb_lo[d] = v;
      if (v > t->bb_hi[d]) t->bb_hi[d] = v;
    }
  build_ctx b = {xyz, t->perm};
  float lo[3], hi[3];
  memcpy(lo, t->bb_lo, sizeof lo);
  memcpy(hi, t->bb_hi, sizeof hi);
  divide(t, &b, 0, m, lo, hi);
  for (int i = 0; i < m; ++i) memcpy(t->pts + 3 * i, xyz + 3 * t->perm[i], 3 * sizeof(float));
  return t;
}

void orc_kdtree_free(orc_kdtree *t) {
  if (!t) return;
  free(t->pts); free(t->perm); free(t->nodes); free(t);
}

/* ---------------- k-NN ---------------- */
typedef struct {
  int k, count;
  int32_t *idx;
  float *d2;
  float worst;
} knn_set;

static inline void knn_add(knn_set *s, float d, int32_t id) {
  int i;
  if (s->count < s->k) i = s->count++;
  else if (d < s->d2[s->k - 1]) i = s->k - 1;
  else return;
  /* insertion; equal distances keep first-found first */
  while (i > 0 && s->d2[i - 1] > d) { s->d2[i] = s->d2[i - 1]; s->idx[i] = s->idx[i - 1]; --i; }
  s->d2[i] = d;
  s->idx[i] = id;
  if (s->count == s->k) s->worst = s->d2[s->k - 1];
}

static void knn_search(const orc_kdtree *t, int ni, const float q[3], float mindist, float dists[3], knn_set *s) {
  const kd_node *nd = &t->nodes[ni];
  if (nd->dim < 0) {
    for (int i = nd->left; i < nd->right; ++i) {
      const float *p = t->pts + 3 * i;
      float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
      float d = dx * dx + dy * dy + dz * dz;
      if (d < s->worst) knn_add(s, d, t->perm[i]);
    }
    return;
  }
  int dim = nd->dim;
  float val = q[dim];
  float diff1 = val - nd->div_low, diff2 = val - nd->div_high;
  int best, other;
  float cut;
  if (diff1 + diff2 < 0) { best = nd->left; other = nd->right; cut = diff2 * diff2; }
  else { best = nd->right; other = nd->left; cut = diff1 * diff1; }
  knn_search(t, best, q, mindist, dists, s);
  float dst = dists[dim];
  mindist = mindist + cut - dst;
  dists[dim] = cut;
  if (mindist <= s->worst) knn_search(t, other, q, mindist, dists, s);
  dists[dim] = dst;
}

static float init_dists(const orc_kdtree *t, const float q[3], float dists[3]) {
  float sum = 0;
  for (int d = 0; d < 3; ++d) {
    dists[d] = 0;
    if (q[d] < t->bb_lo[d]) { float v = q[d] - t->bb_lo[d]; dists[d] = v * v; }
    if (q[d] > t->bb_hi[d]) { float v = q[d] - t->bb_hi[d]; dists[d] = v * v; }
    sum += dists[d];
  }
  return sum;
}

void orc_kdtree_knn(const orc_kdtree *t, const float *q, int nq, int k, int32_t *idx, float *d2, int32_t *found) {
  for (int i = 0; i < nq; ++i) {
    knn_set s = {k, 0, idx + (size_t)i * k, d2 + (size_t)i * k, INFINITY};
    const float *qi = q + 3 * i;
    if (t->n > 0 && isfinite(qi[0]) && isfinite(qi[1]) && isfinite(qi[2])) {
      float dists[3];
      float md = init_dists(t, qi, dists);
      knn_search(t, 0, qi, md, dists, &s);
    }
    for (int j = s.count; j < k; ++j) { s.idx[j] = -1; s.d2[j] = INFINITY; }
    if (found) found[i] = s.count;
  }
}

/* ---------------- radius ---------------- */
typedef struct {
  float r2;
  int64_t count;
  int32_t *idx;
  float *d2;
  int64_t cap;
} rad_set;

static void rad_search(const orc_kdtree *t, int ni, const float q[3], float mindist, float dists[3], rad_set *s) {
  const kd_node *nd = &t->nodes[ni];
  if (nd->dim < 0) {
    for (int i = nd->left; i < nd->right; ++i) {
      const float *p = t->pts + 3 * i;
      float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
      float d = dx * dx + dy * dy + dz * dz;
      if (d <= s->r2) {
        if (s->idx && s->count < s->cap) { s->idx[s->count] = t->perm[i]; s->d2[s->count] = d; }
        s->count++;
      }
    }
    return;
  }
  int dim = nd->dim;
  float val = q[dim];
  float diff1 = val - nd->div_low, diff2 = val - nd->div_high;
  int best, other;
  float cut;
  if (diff1 + diff2 < 0) { best = nd->left; other = nd->right; cut = diff2 * diff2; }
  else { best = nd->right; other = nd->left; cut = diff1 * diff1; }
  rad_search(t, best, q, mindist, dists, s);
  float dst = dists[dim];
  mindist = mindist + cut - dst;
  dists[dim] = cut;
  if (mindist <= s->r2) rad_search(t, other, q, mindist, dists, s);
  dists[dim] = dst;
}

static void sort_pairs(int32_t *idx, float *d2, int64_t n) {
  /* insertion/shell sort: neighbour lists are short */
  for (int64_t gap = n / 2; gap > 0; gap /= 2)
    for (int64_t i = gap; i < n; ++i) {
      float d = d2[i]; int32_t id = idx[i];
      int64_t j = i;
      while (j >= gap && d2[j - gap] > d) { d2[j] = d2[j - gap]; idx[j] = idx[j - gap]; j -= gap; }
      d2[j] = d; idx[j] = id;
    }
}

int64_t orc_kdtree_radius(const orc_kdtree *t, const float *q, int nq, float radius, int sorted,
                          int64_t *offsets, int32_t *idx, float *d2, int64_t cap) {
  int64_t total = 0;
  for (int i = 0; i < nq; ++i) {
    offsets[i] = total;
    const float *qi = q + 3 * i;
    if (t->n == 0 || !(isfinite(qi[0]) && isfinite(qi[1]) && isfinite(qi[2]))) continue;
    rad_set s = {radius * radius, 0, idx ? idx + total : NULL, d2 ? d2 + total : NULL,
                 idx ? (cap > total ? cap - total : 0) : 0};
    float dists[3];
    float md = init_dists(t, qi, dists);
    rad_search(t, 0, qi, md, dists, &s);
    if (idx && sorted) sort_pairs(idx + total, d2 + total, s.count < s.cap ? s.count : s.cap);
    total += s.count;
  }
  offsets[nq] = total;
  return total;
}

void orc_bruteforce_nn(const float *tgt, int nt, const float *q, int nq, int32_t *idx, float *d2) {
  for (int i = 0; i < nq; ++i) {
    float best = INFINITY;
    int32_t bi = -1;
    for (int j = 0; j < nt; ++j) {
      float dx = q[3 * i] - tgt[3 * j], dy = q[3 * i + 1] - tgt[3 * j + 1], dz = q[3 * i + 2] - tgt[3 * j + 2];
      float d = dx * dx + dy * dy + dz * dz;
      if (d < best) { best = d; bi = j; }
    }
    idx[i] = bi;
    d2[i] = best;
  }
}
