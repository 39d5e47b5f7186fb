// bvh_build.cpp — host-side construction of the implicit balanced BVH (see ope_internal.hpp).
//
// Replaces the kd-tree build inside Registration::initCompute
// (vPCL impl/registration_mod.hpp:80-84: tree_->setInputCloud(target_)).  The structure is
// chosen for the GPU traversal, not to mimic FLANN: a perfect binary tree (no child pointers,
// no per-leaf metadata) whose nodes carry tight AABBs, split at the rank that keeps every leaf
// bucket at floor/ceil(n / 2^D) points so leaf ranges are computed arithmetically.
#include <algorithm>
#include <cfloat>
#include <cstring>
#include <numeric>

#include "ope_internal.hpp"

namespace ope {

namespace {

struct Builder {
  const float *xyz;
  std::vector<uint32_t> order;  // permutation being partitioned
  size_t n;
  int D;
  std::vector<float> *boxes;

  size_t leaf_start(size_t j) const { return (size_t)(((unsigned long long)j * n) >> D); }

  // node covers leaves [la, lb); returns its box in lo/hi
  void rec(size_t node, size_t la, size_t lb, float lo[3], float hi[3]) {
    size_t b = leaf_start(la), e = leaf_start(lb);
    if (lb - la == 1) {
      for (int d = 0; d < 3; ++d) { lo[d] = FLT_MAX; hi[d] = -FLT_MAX; }
      for (size_t i = b; i < e; ++i) {
        const float *p = xyz + 3 * (size_t)order[i];
        for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], p[d]); hi[d] = std::max(hi[d], p[d]); }
      }
    } else {
      // split dimension: widest extent of the points of this node
      float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
      for (size_t i = b; i < e; ++i) {
        const float *p = xyz + 3 * (size_t)order[i];
        for (int d = 0; d < 3; ++d) { mn[d] = std::min(mn[d], p[d]); mx[d] = std::max(mx[d], p[d]); }
      }
      int dim = 0;
      if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1;
      if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
      size_t lm = (la + lb) / 2;
      size_t mid = leaf_start(lm);
      const float *base = xyz;
      std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e,
                       [base, dim](uint32_t a, uint32_t c) {
                         float va = base[3 * (size_t)a + dim], vc = base[3 * (size_t)c + dim];
                         return va < vc || (va == vc && a < c);
                       });
      float llo[3], lhi[3], rlo[3], rhi[3];
      rec(2 * node, la, lm, llo, lhi);
      rec(2 * node + 1, lm, lb, rlo, rhi);
      for (int d = 0; d < 3; ++d) { lo[d] = std::min(llo[d], rlo[d]); hi[d] = std::max(lhi[d], rhi[d]); }
    }
    float *bx = boxes->data() + 6 * node;
    for (int d = 0; d < 3; ++d) { bx[d] = lo[d]; bx[3 + d] = hi[d]; }
  }
};

}  // namespace

void build_bvh_host(const float *xyz, const int32_t *ids, const float *nrm, size_t n, int leaf_size, HostBvh &out) {
  if (leaf_size < 1) leaf_size = 16;
  int D = 0;
  while (((n + ((size_t)1 << D) - 1) >> D) > (size_t)leaf_size) ++D;
  out.depth = D;
  size_t n_nodes = (size_t)2 << D;
  out.boxes.assign(n_nodes * 6, 0.f);
  Builder b;
  b.xyz = xyz;
  b.n = n;
  b.D = D;
  b.boxes = &out.boxes;
  b.order.resize(n);
  std::iota(b.order.begin(), b.order.end(), 0u);
  float lo[3], hi[3];
  b.rec(1, 0, (size_t)1 << D, lo, hi);
  // node 0 is unused: give it the root box so stray reads are harmless
  std::memcpy(out.boxes.data(), out.boxes.data() + 6, 6 * sizeof(float));
  out.pts4.resize(n * 4);
  if (nrm) out.nrm4.resize(n * 4);
  for (size_t i = 0; i < n; ++i) {
    size_t s = b.order[i];
    out.pts4[4 * i + 0] = xyz[3 * s + 0];
    out.pts4[4 * i + 1] = xyz[3 * s + 1];
    out.pts4[4 * i + 2] = xyz[3 * s + 2];
    int32_t id = ids ? ids[s] : (int32_t)s;
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
