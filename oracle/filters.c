/*
 * filters.c — CPU ORACLE (test infrastructure, NOT product code): the point-cloud filters either side of
 * the registration path.  PARITY UNPINNED, see ope_oracle.h.
 *
 *   pcl::removeNaNFromPointCloud   DetectAndLocalize/src/poseestimator.cpp:192-194
 *   pcl::PassThrough::filter       BuildModel/src/processingpcd.cpp:8-36 (z, then y, then x)
 *   pcl::VoxelGrid::filter         BuildModel/src/processingpcd.cpp:39-52
 *   pcl::StatisticalOutlierRemoval DetectAndLocalize/src/processingpcd.cpp:62-77 (getOutlierRemove: meanK 30)
 *
 * The three classes live in the un-vendored PCL (nominally 1.7.2); their published algorithms are restated:
 * filters/impl/passthrough.hpp (applyFilterIndices), filters/impl/voxel_grid.hpp (applyFilter),
 * common/impl/io / filter.hpp (removeNaNFromPointCloud), filters/impl/statistical_outlier_removal.hpp
 * (applyFilterIndices).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "ope_oracle.h"

static int finite3(const float *p) { return isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]); }

/* removeNaNFromPointCloud: input order kept, index = position in the input */
int orc_remove_nan(const float *xyz, int n, int32_t *out_idx) {
  int m = 0;
  for (int i = 0; i < n; ++i)
    if (finite3(xyz + 3 * i)) out_idx[m++] = i;
  return m;
}

/* PassThrough with filter_limit_negative = false, keep_organized = false, applied to the three fields one
 * after the other as getPassThrough does: a point survives iff it is finite and lo[d] <= p[d] <= hi[d]
 * for every d (a field's own non-finite value removes the point; limits are inclusive:
 * "if (value > max || value < min) -> removed"). */
int orc_pass_through(const float *xyz, int n, const float lo[3], const float hi[3], int32_t *out_idx) {
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const float *p = xyz + 3 * i;
    if (!finite3(p)) continue;
    int keep = 1;
    for (int d = 0; d < 3; ++d)
      if (p[d] > hi[d] || p[d] < lo[d]) keep = 0;
    if (keep) out_idx[m++] = i;
  }
  return m;
}

typedef struct { int64_t key; int32_t idx; } vg_cell;
static int vg_cmp(const void *a, const void *b) {
  const vg_cell *x = (const vg_cell *)a, *y = (const vg_cell *)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* VoxelGrid::applyFilter, xyz fields, min_points_per_voxel = 0.
 * Returns the number of voxels (centroids written to out_xyz in ascending voxel index, PCL's own order),
 * or -1 when PCL would warn "Leaf size is too small for the input dataset" and hand back the input.
 * PCL sorts (voxel, point) pairs on the voxel index alone with std::sort, which leaves the order of the
 * float additions inside a voxel unspecified; it is fixed here to ascending input index.  The centroid is
 * sum * (1 / count): Eigen 3.2's operator/= on a float vector multiplies by the reciprocal. */
static int voxel_grid_impl(const float *xyz, const uint32_t *rgb, int n, const float leaf[3], float *out_xyz, uint32_t *out_rgb) {
  float inv[3], mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int d = 0; d < 3; ++d) inv[d] = 1.0f / leaf[d];
  int any = 0;
  for (int i = 0; i < n; ++i) {
    const float *p = xyz + 3 * i;
    if (!finite3(p)) continue;
    any = 1;
    for (int d = 0; d < 3; ++d) { if (p[d] < mn[d]) mn[d] = p[d]; if (p[d] > mx[d]) mx[d] = p[d]; }
  }
  if (!any) return 0;
  int64_t dxyz[3];
  for (int d = 0; d < 3; ++d) dxyz[d] = (int64_t)((mx[d] - mn[d]) * inv[d]) + 1;
  if (dxyz[0] * dxyz[1] * dxyz[2] > (int64_t)2147483647) return -1;
  int min_b[3], div_b[3];
  for (int d = 0; d < 3; ++d) {
    min_b[d] = (int)floorf(mn[d] * inv[d]);
    div_b[d] = (int)floorf(mx[d] * inv[d]) - min_b[d] + 1;
  }
  const int64_t mul[3] = {1, div_b[0], (int64_t)div_b[0] * div_b[1]};
  vg_cell *cv = (vg_cell *)malloc(sizeof(vg_cell) * (size_t)(n > 0 ? n : 1));
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const float *p = xyz + 3 * i;
    if (!finite3(p)) continue;
    int64_t key = 0;
    for (int d = 0; d < 3; ++d) key += (int64_t)(int)(floorf(p[d] * inv[d]) - (float)min_b[d]) * mul[d];
    cv[m].key = key;
    cv[m].idx = i;
    ++m;
  }
  qsort(cv, (size_t)m, sizeof(vg_cell), vg_cmp);
  int cnt = 0;
  for (int i = 0; i < m;) {
    int j = i;
    float c[3] = {0.f, 0.f, 0.f}, col[3] = {0.f, 0.f, 0.f};
    while (j < m && cv[j].key == cv[i].key) {
      const float *p = xyz + 3 * cv[j].idx;
      c[0] += p[0]; c[1] += p[1]; c[2] += p[2];
      if (rgb) {
        /* voxel_grid.hpp, "RGB special case": the three channels of pcl::RGB (memory order b, g, r, a) enter the centroid
         * vector as floats of their own and are averaged like every other field */
        const uint32_t v = rgb[cv[j].idx];
        col[0] += (float)((v >> 16) & 255u); col[1] += (float)((v >> 8) & 255u); col[2] += (float)(v & 255u);
      }
      ++j;
    }
    const float r = 1.0f / (float)(j - i);
    out_xyz[3 * cnt] = c[0] * r; out_xyz[3 * cnt + 1] = c[1] * r; out_xyz[3 * cnt + 2] = c[2] * r;
    if (rgb) {
      /* "pack r/g/b into rgb": int rgb = (static_cast<int>(r) << 16) | (static_cast<int>(g) << 8) | static_cast<int>(b) */
      out_rgb[cnt] = ((uint32_t)(int)(col[0] * r) << 16) | ((uint32_t)(int)(col[1] * r) << 8) | (uint32_t)(int)(col[2] * r);
    }
    ++cnt;
    i = j;
  }
  free(cv);
  return cnt;
}

/* StatisticalOutlierRemoval::applyFilterIndices (negative_ = false), restated [uPCL-recall, PCL 1.7.2]:
 *   pass 1, per point: non-finite -> distance 0 and NOT counted; else nearestKSearch(mean_k + 1) and
 *           distance = (float)( sum_{k=1..mean_k} sqrt(d2[k]) / mean_k )   (k = 0 is the point itself; the sum and the
 *           square roots are double: an unqualified sqrt() on a float resolves to ::sqrt(double) there);
 *   mean / variance over the WHOLE distance vector (zeros of non-finite points included in the sums, excluded from
 *           the count): mean = sum / valid, variance = (sq_sum - sum*sum/valid) / (valid - 1), all in double;
 *   pass 2: a point is removed iff distance > mean + stddev_mul * stddev.  Non-finite points carry distance 0 and are
 *           therefore KEPT (PCL quirk; the reference removes NaNs beforehand, poseestimator.cpp:192-194).
 * Fewer than mean_k + 1 finite points: the neighbours that exist are summed, still divided by mean_k.
 * out_idx: kept indices in input order (capacity n); out_dist (optional, n floats): the distance vector. */
int orc_statistical_outlier_removal(const float *xyz, int n, int mean_k, double stddev_mul, int32_t *out_idx, float *out_dist) {
  if (n <= 0 || mean_k < 1) return 0;
  float *dist = (float *)malloc(sizeof(float) * (size_t)n);
  const int k = mean_k + 1;
  int32_t *nn = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
  float *nd = (float *)malloc(sizeof(float) * (size_t)k);
  orc_kdtree *tree = orc_kdtree_build(xyz, n, 15);
  int valid = 0;
  for (int i = 0; i < n; ++i) {
    const float *p = xyz + 3 * i;
    dist[i] = 0.0f;
    if (!finite3(p)) continue;
    int32_t found = 0;
    orc_kdtree_knn(tree, p, 1, k, nn, nd, &found);
    if (found == 0) continue;
    double sum = 0.0;
    for (int j = 1; j < found; ++j) sum += sqrt((double)nd[j]);
    dist[i] = (float)(sum / (double)mean_k);
    ++valid;
  }
  orc_kdtree_free(tree);
  free(nn);
  free(nd);
  double sum = 0.0, sq_sum = 0.0;
  for (int i = 0; i < n; ++i) { sum += dist[i]; sq_sum += (double)dist[i] * (double)dist[i]; }
  const double mean = sum / (double)valid;
  const double variance = (sq_sum - sum * sum / (double)valid) / ((double)valid - 1.0);
  const double thr = mean + stddev_mul * sqrt(variance);
  int m = 0;
  for (int i = 0; i < n; ++i) {
    if ((double)dist[i] > thr) continue;
    out_idx[m++] = i;
  }
  if (out_dist)
    for (int i = 0; i < n; ++i) out_dist[i] = dist[i];
  free(dist);
  return m;
}

int orc_voxel_grid(const float *xyz, int n, const float leaf[3], float *out_xyz) {
  return voxel_grid_impl(xyz, NULL, n, leaf, out_xyz, NULL);
}

/* VoxelGrid<PointXYZRGB>::applyFilter with downsample_all_data_ = true, the default (ProcessingPcd::getDownSampled,
 * BuildModel/src/processingpcd.cpp:44-59): besides xyz the packed colour is averaged channel by channel in float and
 * re-packed by truncation, alpha byte 0.  rgb / out_rgb: the 32 bits of PointXYZRGB::rgb. */
int orc_voxel_grid_rgb(const float *xyz, const uint32_t *rgb, int n, const float leaf[3], float *out_xyz, uint32_t *out_rgb) {
  return voxel_grid_impl(xyz, rgb, n, leaf, out_xyz, out_rgb);
}
