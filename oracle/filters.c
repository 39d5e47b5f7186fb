/*
 * filters.c — CPU ORACLE (test infrastructure, NOT product code): the point-cloud filters either side of
 * the registration path.  PARITY UNPINNED, see ope_oracle.h.
 *
 *   pcl::removeNaNFromPointCloud   DetectAndLocalize/src/poseestimator.cpp:192-194
 *   pcl::PassThrough::filter       BuildModel/src/processingpcd.cpp:8-36 (z, then y, then x)
 *   pcl::VoxelGrid::filter         BuildModel/src/processingpcd.cpp:39-52
 *
 * The three classes live in the un-vendored PCL (nominally 1.7.2); their published algorithms are restated:
 * filters/impl/passthrough.hpp (applyFilterIndices), filters/impl/voxel_grid.hpp (applyFilter),
 * common/impl/io / filter.hpp (removeNaNFromPointCloud).
 */
#include <float.h>
#include <math.h>
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
int orc_voxel_grid(const float *xyz, int n, const float leaf[3], float *out_xyz) {
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
    float c[3] = {0.f, 0.f, 0.f};
    while (j < m && cv[j].key == cv[i].key) {
      const float *p = xyz + 3 * cv[j].idx;
      c[0] += p[0]; c[1] += p[1]; c[2] += p[2];
      ++j;
    }
    const float r = 1.0f / (float)(j - i);
    out_xyz[3 * cnt] = c[0] * r; out_xyz[3 * cnt + 1] = c[1] * r; out_xyz[3 * cnt + 2] = c[2] * r;
    ++cnt;
    i = j;
  }
  free(cv);
  return cnt;
}
