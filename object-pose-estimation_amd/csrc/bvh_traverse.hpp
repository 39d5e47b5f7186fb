// bvh_traverse.hpp — stackless per-lane traversal of the implicit balanced BVH (gfx950).
//
// One lane = one query.  The "stack" is a 32-bit trail register: as the lane descends one level
// it shifts the trail left and sets bit 0 when the far child still has to be visited; popping is
// ctz(trail) levels up and across to the sibling.  No LDS, no scratch, and the register footprint
// stays small enough for 8 waves per SIMD.  Queries arrive Morton-sorted, so neighbouring lanes
// walk the same nodes and their 48-byte child-box loads coalesce into the same cache lines.
//
// Exactness: all distances use the same unfused fp32 operation order as the CPU oracle
// ((dx*dx + dy*dy) + dz*dz); the box lower bound uses that order too, so by monotonicity of
// rounding bound(box) <= dist(q, p) for every p in the box and pruning never drops the true
// nearest neighbour.
#pragma once

#include "ope_internal.hpp"

namespace ope {

__device__ __forceinline__ float sq_dist3(float dx, float dy, float dz) {
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

__device__ __forceinline__ float box_dist2(float lx, float ly, float lz, float hx, float hy, float hz, float qx,
                                           float qy, float qz) {
  float dx = fmaxf(fmaxf(__fsub_rn(lx, qx), __fsub_rn(qx, hx)), 0.f);
  float dy = fmaxf(fmaxf(__fsub_rn(ly, qy), __fsub_rn(qy, hy)), 0.f);
  float dz = fmaxf(fmaxf(__fsub_rn(lz, qz), __fsub_rn(qz, hz)), 0.f);
  return sq_dist3(dx, dy, dz);
}

// Visitor concept:
//   bool prune(float bound) const   -> true if a subtree whose lower bound is `bound` can be skipped
//   void point(float d2, const float4& p, uint32_t pos)   -> candidate at reordered position pos
template <class Visitor>
__device__ __forceinline__ void bvh_traverse(const BvhView &t, float qx, float qy, float qz, Visitor &v) {
  const uint32_t leaf0 = 1u << t.depth;
  uint32_t node = 1;
  uint32_t trail = 0;
  {
    const float *rb = t.boxes + 6;
    if (v.prune(box_dist2(rb[0], rb[1], rb[2], rb[3], rb[4], rb[5], qx, qy, qz))) return;
  }
  for (;;) {
    bool dead = false;
    while (node < leaf0) {
      const float4 *cb = reinterpret_cast<const float4 *>(t.boxes + 12 * (size_t)node);
      const float4 a = cb[0], b = cb[1], c = cb[2];
      const float d0 = box_dist2(a.x, a.y, a.z, a.w, b.x, b.y, qx, qy, qz);
      const float d1 = box_dist2(b.z, b.w, c.x, c.y, c.z, c.w, qx, qy, qz);
      const bool right = d1 < d0;
      const float dn = right ? d1 : d0;
      const float df = right ? d0 : d1;
      if (v.prune(dn)) { dead = true; break; }
      trail = (trail << 1) | (v.prune(df) ? 0u : 1u);
      node = 2 * node + (right ? 1u : 0u);
    }
    if (!dead) {
      const uint32_t j = node - leaf0;
      const uint32_t s = (uint32_t)(((unsigned long long)j * t.n) >> t.depth);
      const uint32_t e = (uint32_t)(((unsigned long long)(j + 1) * t.n) >> t.depth);
      for (uint32_t i = s; i < e; ++i) {
        const float4 p = t.pts[i];
        const float d = sq_dist3(__fsub_rn(qx, p.x), __fsub_rn(qy, p.y), __fsub_rn(qz, p.z));
        v.point(d, p, i);
      }
    }
    // pop: nearest pending sibling on the way up, re-checked against the (possibly tighter) bound
    for (;;) {
      if (trail == 0) return;
      const int k = __builtin_ctz(trail);
      node = (node >> k) ^ 1u;
      trail = (trail >> k) & ~1u;
      const float2 *bx = reinterpret_cast<const float2 *>(t.boxes + 6 * (size_t)node);
      const float2 u = bx[0], w = bx[1], z = bx[2];
      if (!v.prune(box_dist2(u.x, u.y, w.x, w.y, z.x, z.y, qx, qy, qz))) break;
    }
  }
}

struct NearestVisitor {
  float best;
  int idx;  // ORIGINAL target index
  uint32_t pos;
  __device__ __forceinline__ bool prune(float bound) const { return !(bound < best); }
  __device__ __forceinline__ void point(float d, const float4 &p, uint32_t i) {
    if (d < best) { best = d; idx = __float_as_int(p.w); pos = i; }
  }
};

// Apply the 3x4 fp32 transform rows (r00 r01 r02 tx | r10 … | r20 …) with the oracle's operation
// order: ((r0*x + r1*y) + r2*z) + t, unfused.
__device__ __forceinline__ float xform_row(const float *r, float x, float y, float z) {
  return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(r[0], x), __fmul_rn(r[1], y)), __fmul_rn(r[2], z)), r[3]);
}
__device__ __forceinline__ float rot_row(const float *r, float x, float y, float z) {
  return __fadd_rn(__fadd_rn(__fmul_rn(r[0], x), __fmul_rn(r[1], y)), __fmul_rn(r[2], z));
}

}  // namespace ope
