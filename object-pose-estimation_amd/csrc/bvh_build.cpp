// bvh_build.cpp — host-side construction of the implicit balanced OBB tree (see ope_internal.hpp).
//
// Replaces the kd-tree build inside Registration::initCompute
// (vPCL impl/registration_mod.hpp:80-84: tree_->setInputCloud(target_)).  The structure is chosen for
// the GPU traversal, not to mimic FLANN:
//   * a perfect binary tree (no child pointers, no per-leaf metadata), split at the rank that keeps
//     every leaf bucket at floor/ceil(n / 2^D) points, so leaf ranges are computed arithmetically;
//   * every node carries an ORIENTED bounding box (PCA axes of its points).  Registration targets are
//     surface samples: an axis-aligned box around a tilted 3 mm patch is ~1-2 mm thick, while the
//     oriented box is as thin as the patch's sagitta (tens of µm).  For a query far from the surface
//     (clutter, early iterations) the candidate set {boxes closer than the current best} shrinks by
//     more than an order of magnitude, which is what bounds the longest wave of the search kernel.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <numeric>

#include "ope_internal.hpp"

namespace ope {

namespace {

// symmetric 3x3 eigen-decomposition (cyclic Jacobi, fp64); columns of V are eigenvectors
void jacobi_eig3(double S[9], double V[9]) {
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 50; ++sweep) {
    const double off = std::fabs(S[1]) + std::fabs(S[2]) + std::fabs(S[5]);
    const double diag = std::fabs(S[0]) + std::fabs(S[4]) + std::fabs(S[8]);
    if (off <= 1e-300 || off <= 1e-16 * diag) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = S[3 * p + q];
        if (apq == 0.0) continue;
        const double theta = (S[3 * q + q] - S[3 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          const double a = S[3 * k + p], b = S[3 * k + q];
          S[3 * k + p] = c * a - s * b;
          S[3 * k + q] = s * a + c * b;
        }
        for (int k = 0; k < 3; ++k) {
          const double a = S[3 * p + k], b = S[3 * q + k];
          S[3 * p + k] = c * a - s * b;
          S[3 * q + k] = s * a + c * b;
        }
        for (int k = 0; k < 3; ++k) {
          const double a = V[3 * k + p], b = V[3 * k + q];
          V[3 * k + p] = c * a - s * b;
          V[3 * k + q] = s * a + c * b;
        }
      }
  }
}

struct Builder {
  const float *xyz;
  std::vector<uint32_t> order;  // permutation being partitioned
  size_t n;
  int D;
  float *nodes;     // kNodeFloats per node
  double scale;     // root extent, sets the absolute rounding margin

  size_t leaf_start(size_t j) const { return (size_t)(((unsigned long long)j * n) >> D); }

  // Oriented box of points order[b..e): PCA axes, then mid-range centre and half extents along them.
  void fit_obb(size_t node, size_t b, size_t e) {
    float *o = nodes + kNodeFloats * node;
    if (e <= b) {  // empty leaf (n < 2^D): a box nothing can be close to
      for (int i = 0; i < 12; ++i) o[i] = 0.f;
      o[0] = o[1] = o[2] = 1e30f;
      o[4] = 1.f; o[9] = 1.f;
      return;
    }
    double mean[3] = {0, 0, 0};
    for (size_t i = b; i < e; ++i)
      for (int d = 0; d < 3; ++d) mean[d] += xyz[3 * (size_t)order[i] + d];
    for (int d = 0; d < 3; ++d) mean[d] /= (double)(e - b);
    double C[9] = {0};
    for (size_t i = b; i < e; ++i) {
      const float *p = xyz + 3 * (size_t)order[i];
      const double v[3] = {p[0] - mean[0], p[1] - mean[1], p[2] - mean[2]};
      for (int r = 0; r < 3; ++r)
        for (int c = r; c < 3; ++c) C[3 * r + c] += v[r] * v[c];
    }
    C[3] = C[1]; C[6] = C[2]; C[7] = C[5];
    double V[9];
    jacobi_eig3(C, V);
    // two axes as floats, re-orthonormalised in fp64 after rounding; the third is their cross product,
    // exactly as the kernel recomputes it
    float a0[3], a1[3];
    for (int d = 0; d < 3; ++d) { a0[d] = (float)V[3 * d + 0]; a1[d] = (float)V[3 * d + 1]; }
    double A[3][3];
    for (int d = 0; d < 3; ++d) { A[0][d] = a0[d]; A[1][d] = a1[d]; }
    A[2][0] = A[0][1] * A[1][2] - A[0][2] * A[1][1];
    A[2][1] = A[0][2] * A[1][0] - A[0][0] * A[1][2];
    A[2][2] = A[0][0] * A[1][1] - A[0][1] * A[1][0];
    double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (size_t i = b; i < e; ++i) {
      const float *p = xyz + 3 * (size_t)order[i];
      for (int k = 0; k < 3; ++k) {
        const double t = A[k][0] * p[0] + A[k][1] * p[1] + A[k][2] * p[2];
        lo[k] = std::min(lo[k], t);
        hi[k] = std::max(hi[k], t);
      }
    }
    // centre = mid-range point expressed back in world coordinates (axes are orthonormal to ~1e-7)
    double mid[3], c[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) mid[k] = 0.5 * (lo[k] + hi[k]);
    for (int d = 0; d < 3; ++d) c[d] = mid[0] * A[0][d] + mid[1] * A[1][d] + mid[2] * A[2][d];
    const float cf[3] = {(float)c[0], (float)c[1], (float)c[2]};
    // half extents about the ROUNDED centre, measured with the rounded axes, plus a margin that
    // covers the kernel's fp32 evaluation of the projections (relative 4e-6 of the offset, i.e.
    // >10 fp32 ulps, and an absolute floor)
    double h[3] = {0, 0, 0}, far = 0;
    for (size_t i = b; i < e; ++i) {
      const float *p = xyz + 3 * (size_t)order[i];
      const double v[3] = {(double)p[0] - cf[0], (double)p[1] - cf[1], (double)p[2] - cf[2]};
      far = std::max(far, std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]));
      for (int k = 0; k < 3; ++k) h[k] = std::max(h[k], std::fabs(A[k][0] * v[0] + A[k][1] * v[1] + A[k][2] * v[2]));
    }
    const double margin = 4e-6 * far + 1e-7 * scale;
    o[0] = cf[0]; o[1] = cf[1]; o[2] = cf[2];
    o[4] = a0[0]; o[5] = a0[1]; o[6] = a0[2];
    o[8] = a1[0]; o[9] = a1[1]; o[10] = a1[2];
    for (int k = 0; k < 3; ++k) o[4 * k + 3] = std::nextafter((float)(h[k] + margin), FLT_MAX);
  }

  // node covers leaves [la, lb)
  void rec(size_t node, size_t la, size_t lb) {
    const size_t b = leaf_start(la), e = leaf_start(lb);
    if (lb - la > 1) {
      // split dimension: widest extent of the points of this node
      float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
      for (size_t i = b; i < e; ++i) {
        const float *p = xyz + 3 * (size_t)order[i];
        for (int d = 0; d < 3; ++d) { mn[d] = std::min(mn[d], p[d]); mx[d] = std::max(mx[d], p[d]); }
      }
      int dim = 0;
      if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1;
      if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
      const size_t lm = (la + lb) / 2;
      const size_t mid = leaf_start(lm);
      const float *base = xyz;
      std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e,
                       [base, dim](uint32_t a, uint32_t c) {
                         const float va = base[3 * (size_t)a + dim], vc = base[3 * (size_t)c + dim];
                         return va < vc || (va == vc && a < c);
                       });
      rec(2 * node, la, lm);
      rec(2 * node + 1, lm, lb);
    }
    fit_obb(node, b, e);
  }
};

}  // namespace

void build_bvh_host(const float *xyz, const int32_t *ids, const float *nrm, size_t n, int leaf_size, HostBvh &out) {
  if (leaf_size < 1) leaf_size = 16;
  int D = 0;
  while (((n + ((size_t)1 << D) - 1) >> D) > (size_t)leaf_size) ++D;
  // the traversal keeps one pending-bound slot per level in LDS: cap the depth, grow the leaves instead
  if (D > kMaxDepth) D = kMaxDepth;
  out.depth = D;
  const size_t n_nodes = (size_t)2 << D;
  out.nodes.assign(n_nodes * kNodeFloats, 0.f);
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (size_t i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], xyz[3 * i + d]); hi[d] = std::max(hi[d], xyz[3 * i + d]); }
  Builder b;
  b.xyz = xyz;
  b.n = n;
  b.D = D;
  b.nodes = out.nodes.data();
  b.scale = std::max({(double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2], 1e-3}) +
            std::max({std::fabs((double)lo[0]), std::fabs((double)hi[0]), std::fabs((double)lo[1]),
                      std::fabs((double)hi[1]), std::fabs((double)lo[2]), std::fabs((double)hi[2])});
  b.order.resize(n);
  std::iota(b.order.begin(), b.order.end(), 0u);
  b.rec(1, 0, (size_t)1 << D);
  // node 0 is unused: give it the root box so stray reads are harmless
  std::memcpy(out.nodes.data(), out.nodes.data() + kNodeFloats, kNodeFloats * sizeof(float));
  out.pts4.resize(n * 4);
  if (nrm) out.nrm4.resize(n * 4);
  for (size_t i = 0; i < n; ++i) {
    const size_t s = b.order[i];
    out.pts4[4 * i + 0] = xyz[3 * s + 0];
    out.pts4[4 * i + 1] = xyz[3 * s + 1];
    out.pts4[4 * i + 2] = xyz[3 * s + 2];
    const int32_t id = ids ? ids[s] : (int32_t)s;
    std::memcpy(&out.pts4[4 * i + 3], &id, 4);
    if (nrm) {
      out.nrm4[4 * i + 0] = nrm[3 * s + 0];
      out.nrm4[4 * i + 1] = nrm[3 * s + 1];
      out.nrm4[4 * i + 2] = nrm[3 * s + 2];
      out.nrm4[4 * i + 3] = 0.f;
    }
  }
}

}  // namespace ope
