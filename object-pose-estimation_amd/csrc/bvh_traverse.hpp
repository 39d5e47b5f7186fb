// bvh_traverse.hpp — per-lane traversal of the implicit balanced OBB tree (gfx950, wave64).
//
// One lane = one query.  Control state is two registers: the heap id of the current node and a
// "trail" bit per level (bit set = the sibling on that level is still pending).  The lower bound of
// every pending sibling is parked in LDS (one float per level and lane, lane-strided so a wave's
// accesses hit 64 different banks), so backing up the tree costs LDS reads only: every visited node
// costs exactly one global fetch (its two children, 96 contiguous bytes).  Queries arrive
// Morton-sorted, so neighbouring lanes fetch the same lines.
//
// Measured alternatives (C3, 1 M x 100 k, same box): a wave-cooperative "packet" walk (uniform
// loads, ballots) was 2x slower than private walks because clutter waves are incoherent and the
// union of 64 long walks is serial; axis-aligned boxes needed 10-40x more node visits than oriented
// boxes for queries far from the surface, and those few waves set the kernel's duration.
//
// Exactness: point distances use the same unfused fp32 operation order as the CPU oracle
// ((dx*dx + dy*dy) + dz*dz).  The OBB bound is evaluated in fp32 from a box that the build inflated
// by a margin covering that evaluation, and is scaled down by 4e-6 (> 30 ulps): it never exceeds the
// computed distance of a point inside the box, so pruning cannot drop the true nearest neighbour.
#pragma once

#include "ope_internal.hpp"

namespace ope {

// Native 16-byte vector: a load through this type is ONE global_load_dwordx4 (loads through HIP's
// float4 struct were split into odd 4/8/12-byte pieces by the load vectoriser).
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f ld16(const float4 *p) { return *reinterpret_cast<const v4f *>(p); }

__device__ __forceinline__ float sq_dist3(float dx, float dy, float dz) {
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// squared distance lower bound from q to the oriented box {n0, n1, n2} (see BvhView)
__device__ __forceinline__ float obb_dist2(const v4f n0, const v4f n1, const v4f n2, float qx, float qy, float qz) {
  // Bounds need not match the oracle bit for bit (only point distances do): FMAs here, the build's margin
  // and the final 0.999996 cover the rounding.
  const float dx = qx - n0.x, dy = qy - n0.y, dz = qz - n0.z;
  const float a2x = __fmaf_rn(n1.y, n2.z, -(n1.z * n2.y));
  const float a2y = __fmaf_rn(n1.z, n2.x, -(n1.x * n2.z));
  const float a2z = __fmaf_rn(n1.x, n2.y, -(n1.y * n2.x));
  const float t0 = fmaxf(fabsf(__fmaf_rn(dz, n1.z, __fmaf_rn(dy, n1.y, dx * n1.x))) - n0.w, 0.f);
  const float t1 = fmaxf(fabsf(__fmaf_rn(dz, n2.z, __fmaf_rn(dy, n2.y, dx * n2.x))) - n1.w, 0.f);
  const float t2 = fmaxf(fabsf(__fmaf_rn(dz, a2z, __fmaf_rn(dy, a2y, dx * a2x))) - n2.w, 0.f);
  return __fmaf_rn(t2, t2, __fmaf_rn(t1, t1, t0 * t0)) * 0.999996f;
}

// Visitor concept:
//   bool prune(float bound) const                                    -> subtree with this lower bound can be skipped
//   void point(float d2, const v4f& p, uint32_t pos, uint32_t leaf)   -> candidate p (w = original index bits) at reordered position pos
//   void on_node()                                                    -> instrumentation hook (empty in product visitors)
// `stk` points at this lane's slot of an LDS array float[kMaxDepth + 1][stk_stride].
//
// start_leaf == 0: classic top-down walk from the root.
// start_leaf != 0 (heap id of a leaf, e.g. the leaf that held this query's nearest neighbour in the
// previous ICP iteration): the walk starts by scanning that leaf, which usually yields the final
// best at once.  Every ancestor's sibling is then pending; their D box bounds are evaluated UP FRONT
// in batches of four independent fetches (48 bytes each) and parked in LDS, so the back-up phase is
// LDS-only and the D dependent round trips of a level-by-level check collapse into D/4.  Only
// siblings that beat the current best are expanded.  Same exact result as the top-down walk.
//
// (An LDS copy of the first tree levels was tried: selecting between an LDS and a global pointer per
// lane turns every node fetch into a flat_load, which waits on both counters and was slower.)
__device__ __forceinline__ void load_node(const BvhView &t, uint32_t node, v4f &a, v4f &b, v4f &c) {
  const float4 *o = t.nodes + 3 * (size_t)node;
  a = ld16(o); b = ld16(o + 1); c = ld16(o + 2);
}

// Visitors for which presenting a point a second time changes nothing (1-NN with a strict comparison) take a leaf scan
// without per-point range checks; see the leaf branch of bvh_walk.
template <class Visitor> struct leaf_rescan_is_harmless { static constexpr bool value = false; };

template <class Visitor>
__device__ __forceinline__ void bvh_walk(const BvhView &t, float qx, float qy, float qz, Visitor &v, float *stk, int stk_stride,
                                         uint32_t node, uint32_t trail, float minb, bool node_done);

template <class Visitor>
__device__ __forceinline__ void bvh_traverse(const BvhView &t, float qx, float qy, float qz, Visitor &v, float *stk,
                                             int stk_stride, uint32_t start_leaf = 0) {
  const uint32_t leaf0 = 1u << t.depth;
  uint32_t node = 1;
  uint32_t trail = 0;  // bit k: the sibling of the k-th ancestor (bit 0: of `node` itself) is pending
  float minb = INFINITY;  // never above the smallest parked bound: if it cannot beat the best, nothing pending can
  if (start_leaf != 0) {
    node = start_leaf;
    trail = leaf0 - 1u;
    // bounds of all D ancestor siblings, four fetches in flight at a time
    const int D = t.depth;
    for (int k = 0; k < D; k += 4) {
      v4f a0, b0, c0, a1, b1, c1, a2, b2, c2, a3, b3, c3;
      const uint32_t s0 = (start_leaf >> k) ^ 1u;
      const uint32_t s1 = (k + 1 < D) ? ((start_leaf >> (k + 1)) ^ 1u) : s0;
      const uint32_t s2 = (k + 2 < D) ? ((start_leaf >> (k + 2)) ^ 1u) : s0;
      const uint32_t s3 = (k + 3 < D) ? ((start_leaf >> (k + 3)) ^ 1u) : s0;
      load_node(t, s0, a0, b0, c0);
      load_node(t, s1, a1, b1, c1);
      load_node(t, s2, a2, b2, c2);
      load_node(t, s3, a3, b3, c3);
      const float e0 = obb_dist2(a0, b0, c0, qx, qy, qz), e1 = obb_dist2(a1, b1, c1, qx, qy, qz),
                  e2 = obb_dist2(a2, b2, c2, qx, qy, qz), e3 = obb_dist2(a3, b3, c3, qx, qy, qz);
      stk[(D - k) * stk_stride] = e0;
      if (k + 1 < D) stk[(D - k - 1) * stk_stride] = e1;
      if (k + 2 < D) stk[(D - k - 2) * stk_stride] = e2;
      if (k + 3 < D) stk[(D - k - 3) * stk_stride] = e3;
      minb = fminf(minb, fminf(fminf(e0, e1), fminf(e2, e3)));  // duplicates of e0 in the tail are harmless
    }
  } else {
    v4f a, b, c;
    load_node(t, 1, a, b, c);
    if (v.prune(obb_dist2(a, b, c, qx, qy, qz))) return;
  }
  bvh_walk(t, qx, qy, qz, v, stk, stk_stride, node, trail, minb, false);
}

// The walk proper, from a state {node, trail, minb, parked bounds}.  Flat loop: each trip advances every lane by
// one unit of work (an inner-node step OR a whole leaf scan), then backs up through LDS-parked bounds.  Measured
// alternatives on C3 (same box, same build otherwise): "while-while" (all lanes walk to a leaf, then scan
// together) 442 us vs 283 us; one unified 6-load trip per lane state 467 us; round 2: back up -> step -> leaf scan within
// one trip for a lane that can flow through (fewer trips, the same phases per trip): tree kernel 155-163 us vs 153-156 on
// the C3 frame, 97-98 vs 95-97 without clutter (the costly chunks are served by packet and group walks, not by this loop).
// Round 3: two levels per trip (the four grandchildren, 192 contiguous bytes, instead of the two children): 149.5 vs 143 us on a
// clutter-only launch, 162-165 vs 154 on C3 — extra box tests and eight spilled VGPRs cost more than the halved trips gain.
// node_done: `node` has been dealt with already (start by backing up).
template <class Visitor>
__device__ __forceinline__ void bvh_walk(const BvhView &t, float qx, float qy, float qz, Visitor &v, float *stk, int stk_stride,
                                         uint32_t node, uint32_t trail, float minb, bool node_done) {
  const uint32_t leaf0 = 1u << t.depth;
  for (;;) {
    if (node_done) {
      node_done = false;
    } else if (node < leaf0) {
      v.on_node();
      v4f c0, c1, c2, c3, c4, c5;
      load_node(t, 2 * node, c0, c1, c2);
      load_node(t, 2 * node + 1, c3, c4, c5);
      const float d0 = obb_dist2(c0, c1, c2, qx, qy, qz);
      const float d1 = obb_dist2(c3, c4, c5, qx, qy, qz);
      const bool right = d1 < d0;
      const float dn = right ? d1 : d0;
      const float df = right ? d0 : d1;
      if (!v.prune(dn)) {
        node = 2 * node + (right ? 1u : 0u);
        const bool pend = !v.prune(df);
        trail = (trail << 1) | (pend ? 1u : 0u);
        if (pend) { stk[(31 - __clz(node)) * stk_stride] = df; minb = fminf(minb, df); }
        continue;
      }
    } else {
      const uint32_t j = node - leaf0;
      const uint32_t s = (uint32_t)(((unsigned long long)j * t.n) >> t.depth);
      const uint32_t e = (uint32_t)(((unsigned long long)(j + 1) * t.n) >> t.depth);
      bool scanned = false;
      if constexpr (leaf_rescan_is_harmless<Visitor>::value) {
        // The kernel is bound by instruction issue, so the scan is written for few instructions: eight 16-byte loads
        // from ONE per-lane base address (immediate offsets), the last batch moved back so that it ends at the leaf's
        // end instead of clamping indices and guarding every point (a point seen twice cannot win twice: d < best is
        // strict, and first presentations keep their order), compare + two selects per point, `leaf` set once.
        if (e - s >= 8u) {
          scanned = true;
          const uint32_t pos0 = v.pos;
#define OPE_LEAF_POINT_SEL(P, IDX)                                                                       \
  {                                                                                                      \
    const float d_ = sq_dist3(__fsub_rn(qx, P.x), __fsub_rn(qy, P.y), __fsub_rn(qz, P.z));               \
    const bool c_ = d_ < v.best;                                                                         \
    v.best = c_ ? d_ : v.best;                                                                           \
    v.pos = c_ ? (IDX) : v.pos;                                                                          \
  }
          for (uint32_t i = s;; i += 8) {
            const bool last = i + 8u >= e;
            const uint32_t b = last ? e - 8u : i;
            const float4 *p = t.pts + b;
            const v4f p0 = ld16(p), p1 = ld16(p + 1), p2 = ld16(p + 2), p3 = ld16(p + 3), p4 = ld16(p + 4), p5 = ld16(p + 5),
                      p6 = ld16(p + 6), p7 = ld16(p + 7);
            OPE_LEAF_POINT_SEL(p0, b);
            OPE_LEAF_POINT_SEL(p1, b + 1u);
            OPE_LEAF_POINT_SEL(p2, b + 2u);
            OPE_LEAF_POINT_SEL(p3, b + 3u);
            OPE_LEAF_POINT_SEL(p4, b + 4u);
            OPE_LEAF_POINT_SEL(p5, b + 5u);
            OPE_LEAF_POINT_SEL(p6, b + 6u);
            OPE_LEAF_POINT_SEL(p7, b + 7u);
            if (last) break;
          }
#undef OPE_LEAF_POINT_SEL
          v.leaf = (v.pos != pos0) ? node : v.leaf;
        }
      }
      // eight independent 16-byte loads in flight per batch (visitor calls guarded, so every point is presented
      // exactly once): a default 8-point bucket is ONE trip
#define OPE_LEAF_POINT(P, IDX) \
  v.point(sq_dist3(__fsub_rn(qx, P.x), __fsub_rn(qy, P.y), __fsub_rn(qz, P.z)), P, IDX, node)
      for (uint32_t i = s; !scanned && i < e; i += 8) {
        // one base address, immediate offsets; a batch may run past the leaf (the visitor calls are guarded) and, at
        // the last leaf, up to seven entries past the points: the index allocates kPtsPad zeroed entries there
        const float4 *pb = t.pts + i;
        const v4f p0 = ld16(pb), p1 = ld16(pb + 1), p2 = ld16(pb + 2), p3 = ld16(pb + 3), p4 = ld16(pb + 4), p5 = ld16(pb + 5),
                  p6 = ld16(pb + 6), p7 = ld16(pb + 7);
        OPE_LEAF_POINT(p0, i);
        if (i + 1 < e) OPE_LEAF_POINT(p1, i + 1);
        if (i + 2 < e) OPE_LEAF_POINT(p2, i + 2);
        if (i + 3 < e) OPE_LEAF_POINT(p3, i + 3);
        if (i + 4 < e) OPE_LEAF_POINT(p4, i + 4);
        if (i + 5 < e) OPE_LEAF_POINT(p5, i + 5);
        if (i + 6 < e) OPE_LEAF_POINT(p6, i + 6);
        if (i + 7 < e) OPE_LEAF_POINT(p7, i + 7);
      }
#undef OPE_LEAF_POINT
    }
    // back up to the deepest pending sibling whose parked bound still beats the current best (LDS only);
    // most queries end here without touching LDS: no parked bound is below `minb`
    if (v.prune(minb)) return;
    for (;;) {
      if (trail == 0) return;
      const int k = __builtin_ctz(trail);
      node = (node >> k) ^ 1u;
      trail = (trail >> k) & ~1u;
      if (!v.prune(stk[(31 - __clz(node)) * stk_stride])) break;
    }
  }
}

constexpr uint32_t kNoPos = 0xffffffffu;

struct NearestVisitor {
  float best;
  uint32_t pos;   // reordered target position of the best point, kNoPos if none
  uint32_t leaf;  // heap id of the leaf that holds it (next iteration's start hint)
  __device__ __forceinline__ bool prune(float bound) const { return !(bound < best); }
  __device__ __forceinline__ float bound() const { return best; }   // a subtree is worth entering iff its lower bound is below this
  __device__ __forceinline__ void point(float d, const v4f &, uint32_t i, uint32_t lf) {
    if (d < best) { best = d; pos = i; leaf = lf; }
  }
  __device__ __forceinline__ void on_node() {}
};

template <> struct leaf_rescan_is_harmless<NearestVisitor> { static constexpr bool value = true; };

// ------------------------------------------------------------------------------------------
// Per-lane 1-NN walk with DEFERRED leaf scans (round 3).  In the flat loop of bvh_walk a trip costs the wave a node step AND a
// leaf scan AND a back-up as soon as its lanes are at different phases of their walks, which on the chunks that end a launch
// (clutter far from the surface: ~50 node steps and ~9 leaf scans per lane) is every trip: in-kernel stamps put the leaf
// scan at half of such a trip although a lane needs it on one trip in seven.  Here a lane that reaches a leaf only notes it
// (a queue of `qcap` entries in the unused rows of its parked-bound column) and walks on; the wave scans the noted leaves
// together, position by position, when no lane can walk any further (queues full, or walks finished), and then resumes.
// Exact: a deferred leaf is scanned before the walk returns, and a bound that is tested against an older `best` only
// lets more through.  The leaf of the start hint is scanned up front (it gives the bound everything else is pruned with)
// and so is the first leaf of a walk that has no candidate yet.
#ifndef OPE_SCAN_BATCH
#define OPE_SCAN_BATCH 8u
#endif
__device__ __forceinline__ void scan_leaf_nearest(const BvhView &t, uint32_t node, float qx, float qy, float qz, NearestVisitor &v) {
  const uint32_t j = node - (1u << t.depth);
  const uint32_t s = (uint32_t)(((unsigned long long)j * t.n) >> t.depth);
  const uint32_t e = (uint32_t)(((unsigned long long)(j + 1) * t.n) >> t.depth);
  const uint32_t pos0 = v.pos;
#define OPE_LEAF_POINT_SEL(P, IDX)                                                                       \
  {                                                                                                      \
    const float d_ = sq_dist3(__fsub_rn(qx, P.x), __fsub_rn(qy, P.y), __fsub_rn(qz, P.z));               \
    const bool c_ = d_ < v.best;                                                                         \
    v.best = c_ ? d_ : v.best;                                                                           \
    v.pos = c_ ? (IDX) : v.pos;                                                                          \
  }
  if (e - s >= OPE_SCAN_BATCH) {
    for (uint32_t i = s;; i += OPE_SCAN_BATCH) {
      const bool last = i + OPE_SCAN_BATCH >= e;
      const uint32_t b = last ? e - OPE_SCAN_BATCH : i;
      const float4 *p = t.pts + b;
#if OPE_SCAN_BATCH == 8
      const v4f p0 = ld16(p), p1 = ld16(p + 1), p2 = ld16(p + 2), p3 = ld16(p + 3), p4 = ld16(p + 4), p5 = ld16(p + 5),
                p6 = ld16(p + 6), p7 = ld16(p + 7);
#else
      const v4f p0 = ld16(p), p1 = ld16(p + 1), p2 = ld16(p + 2), p3 = ld16(p + 3);
#endif
      OPE_LEAF_POINT_SEL(p0, b);
      OPE_LEAF_POINT_SEL(p1, b + 1u);
      OPE_LEAF_POINT_SEL(p2, b + 2u);
      OPE_LEAF_POINT_SEL(p3, b + 3u);
#if OPE_SCAN_BATCH == 8
      OPE_LEAF_POINT_SEL(p4, b + 4u);
      OPE_LEAF_POINT_SEL(p5, b + 5u);
      OPE_LEAF_POINT_SEL(p6, b + 6u);
      OPE_LEAF_POINT_SEL(p7, b + 7u);
#endif
      if (last) break;
    }
  } else {
    for (uint32_t i = s; i < e; ++i) {
      const v4f p0 = ld16(t.pts + i);
      OPE_LEAF_POINT_SEL(p0, i);
    }
  }
#undef OPE_LEAF_POINT_SEL
  v.leaf = (v.pos != pos0) ? node : v.leaf;
}

__device__ __forceinline__ void bvh_traverse_deferred(const BvhView &t, float qx, float qy, float qz, NearestVisitor &v, float *stk,
                                                      int stk_stride, uint32_t start_leaf, int qcap) {
  const uint32_t leaf0 = 1u << t.depth;
  const int D = t.depth;
  uint32_t node = 1;
  uint32_t trail = 0;
  float minb = INFINITY;
  bool node_done = false;
  if (start_leaf != 0) {
    node = start_leaf;
    trail = leaf0 - 1u;
    for (int k = 0; k < D; k += 4) {
      v4f a0, b0, c0, a1, b1, c1, a2, b2, c2, a3, b3, c3;
      const uint32_t s0 = (start_leaf >> k) ^ 1u;
      const uint32_t s1 = (k + 1 < D) ? ((start_leaf >> (k + 1)) ^ 1u) : s0;
      const uint32_t s2 = (k + 2 < D) ? ((start_leaf >> (k + 2)) ^ 1u) : s0;
      const uint32_t s3 = (k + 3 < D) ? ((start_leaf >> (k + 3)) ^ 1u) : s0;
      load_node(t, s0, a0, b0, c0);
      load_node(t, s1, a1, b1, c1);
      load_node(t, s2, a2, b2, c2);
      load_node(t, s3, a3, b3, c3);
      const float e0 = obb_dist2(a0, b0, c0, qx, qy, qz), e1 = obb_dist2(a1, b1, c1, qx, qy, qz),
                  e2 = obb_dist2(a2, b2, c2, qx, qy, qz), e3 = obb_dist2(a3, b3, c3, qx, qy, qz);
      stk[(D - k) * stk_stride] = e0;
      if (k + 1 < D) stk[(D - k - 1) * stk_stride] = e1;
      if (k + 2 < D) stk[(D - k - 2) * stk_stride] = e2;
      if (k + 3 < D) stk[(D - k - 3) * stk_stride] = e3;
      minb = fminf(minb, fminf(fminf(e0, e1), fminf(e2, e3)));
    }
    scan_leaf_nearest(t, start_leaf, qx, qy, qz, v);   // the bound everything else is pruned with
    node_done = true;
  } else {
    v4f a, b, c;
    load_node(t, 1, a, b, c);
    if (v.prune(obb_dist2(a, b, c, qx, qy, qz))) return;
  }
  uint32_t *lq = reinterpret_cast<uint32_t *>(stk) + (D + 1) * stk_stride;   // rows D+1 .. D+qcap of this lane's column
  int nq = 0;
  bool walking = true, stall = false;
  for (;;) {
    // ---- walk: node steps and back-ups only
    while (__ballot(walking && !stall && nq < qcap) != 0ull) {
      if (walking && !stall && nq < qcap) {
        bool descend = false;
        if (!node_done) {
          if (node < leaf0) {
            v4f c0, c1, c2, c3, c4, c5;
            load_node(t, 2 * node, c0, c1, c2);
            load_node(t, 2 * node + 1, c3, c4, c5);
            const float d0 = obb_dist2(c0, c1, c2, qx, qy, qz);
            const float d1 = obb_dist2(c3, c4, c5, qx, qy, qz);
            const bool right = d1 < d0;
            const float dn = right ? d1 : d0;
            const float df = right ? d0 : d1;
            if (!v.prune(dn)) {
              node = 2 * node + (right ? 1u : 0u);
              const bool pend = !v.prune(df);
              trail = (trail << 1) | (pend ? 1u : 0u);
              if (pend) { stk[(31 - __clz(node)) * stk_stride] = df; minb = fminf(minb, df); }
              descend = true;
            }
          } else {
            lq[nq * stk_stride] = node;
            ++nq;
            stall = v.pos == kNoPos;   // no candidate yet: this leaf is scanned before the walk goes on
          }
        }
        node_done = false;
        if (!descend) {
          // back up to the deepest pending sibling whose parked bound still beats the current best
          if (v.prune(minb)) walking = false;
          else {
            for (;;) {
              if (trail == 0) { walking = false; break; }
              const int k = __builtin_ctz(trail);
              node = (node >> k) ^ 1u;
              trail = (trail >> k) & ~1u;
              if (!v.prune(stk[(31 - __clz(node)) * stk_stride])) break;
            }
          }
        }
      }
    }
    // ---- scan what the lanes have noted, position by position
    for (int r = 0; __ballot(r < nq) != 0ull; ++r)
      if (r < nq) scan_leaf_nearest(t, lq[r * stk_stride], qx, qy, qz, v);
    nq = 0;
    stall = false;
    if (__ballot(walking) == 0ull) return;
  }
}

// ------------------------------------------------------------------------------------------
// Packet traversal: ONE walk for the 64 queries of a coherent chunk.  Morton-adjacent surface points were matched to
// a handful of leaves in the previous iteration and walk nearly the same nodes; here the walk's control state (node,
// pending bits) is wave-uniform, so nodes and leaf points are fetched through the SCALAR cache (one s_load per record
// for the whole wave instead of 64 divergent 16-byte gathers) and only the box / distance arithmetic is per lane.
// A sibling is entered if ANY lane can still improve there; per-lane bounds of pending siblings are parked in the
// same LDS slots as in the per-lane walk.  Exact for every lane.  Returns false (nothing lost: v keeps what was
// found) if the chunk is not coherent enough (more than kPacketMaxLeaves distinct start leaves, or a lane without
// one); the caller then runs the per-lane walk.  Measured on C3: 6 leaves 184 us, 3 leaves 197 us (= no packets),
// 12 leaves 306 us (the union of less coherent walks is long); handing over to per-lane walks after the shared
// start (leaf scans + ancestor siblings through the scalar cache only) 198 us.  Round 2: leaf points fetched eight at a
// time (two dependent scalar trips per default bucket instead of three, 32 SGPRs): C3 frame 160-162 us vs 155-159, clutter
// alone unchanged, i.e. not the trips of the leaf scans; fetching the children (or first points) of the sibling just parked
// ahead of the back-up that reaches it keeps 32 more SGPRs alive, which the 80-VGPR build pays with 83 spilled VGPRs.
constexpr int kPacketMaxLeaves = 6;

// Developer builds count what a packet walk did (tools/chain_probe.py); the product build compiles the counters away.
struct PacketStats { uint32_t steps, leaves, backups; };
#ifdef OPE_DEVELOPER
#define OPE_PKT_COUNT(ps, field) do { if (ps) ++(ps)->field; } while (0)
#else
#define OPE_PKT_COUNT(ps, field) do { } while (0)
#endif

// A 16-byte load through the CONSTANT address space: with a wave-uniform address the compiler selects s_load_dwordx4
// (scalar cache, result in SGPRs).  Legal because the index is never written while a search kernel runs.
typedef const __attribute__((address_space(4))) v4f *scalar_ptr_v4f;
__device__ __forceinline__ v4f ld16_scalar(const float4 *p) {
  return *reinterpret_cast<scalar_ptr_v4f>(reinterpret_cast<unsigned long long>(p));
}
__device__ __forceinline__ void load_node_scalar(const BvhView &t, uint32_t node, v4f &a, v4f &b, v4f &c) {
  const float4 *o = t.nodes + 3 * (size_t)node;
  a = ld16_scalar(o); b = ld16_scalar(o + 1); c = ld16_scalar(o + 2);
}
// box bound with the third axis given (packet walk: a fourth scalar fetch instead of nine per-lane multiplies)
__device__ __forceinline__ float obb_dist2_axes(const v4f n0, const v4f n1, const v4f n2, const v4f n3, float qx, float qy, float qz) {
  const float dx = qx - n0.x, dy = qy - n0.y, dz = qz - n0.z;
  const float t0 = fmaxf(fabsf(__fmaf_rn(dz, n1.z, __fmaf_rn(dy, n1.y, dx * n1.x))) - n0.w, 0.f);
  const float t1 = fmaxf(fabsf(__fmaf_rn(dz, n2.z, __fmaf_rn(dy, n2.y, dx * n2.x))) - n1.w, 0.f);
  const float t2 = fmaxf(fabsf(__fmaf_rn(dz, n3.z, __fmaf_rn(dy, n3.y, dx * n3.x))) - n2.w, 0.f);
  return __fmaf_rn(t2, t2, __fmaf_rn(t1, t1, t0 * t0)) * 0.999996f;
}
// One node of the packet walk in SGPRs: the 48-byte record plus its stored third axis (the packet walk requires the
// axis2 side array).  Byte offsets are formed in 32 bits (the arrays are far below 4 GB): one s_mul / s_lshl per node
// instead of a 64-bit multiply-add chain -- scalar instructions take issue slots like vector ones.
struct PacketNode { v4f a, b, c, x; };
__device__ __forceinline__ void load_packet_node(const BvhView &t, uint32_t node, PacketNode &n) {
  const float4 *o = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(t.nodes) + (size_t)(node * 48u));
  const float4 *x = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(t.axis2) + (size_t)(node * 16u));
  n.a = ld16_scalar(o); n.b = ld16_scalar(o + 1); n.c = ld16_scalar(o + 2);
  n.x = ld16_scalar(x);
}
// the two children of `node`: 96 + 32 contiguous bytes
__device__ __forceinline__ void load_packet_children(const BvhView &t, uint32_t node, PacketNode &l, PacketNode &r) {
  const float4 *o = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(t.nodes) + (size_t)(node * 96u));
  const float4 *x = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(t.axis2) + (size_t)(node * 32u));
  l.a = ld16_scalar(o); l.b = ld16_scalar(o + 1); l.c = ld16_scalar(o + 2);
  r.a = ld16_scalar(o + 3); r.b = ld16_scalar(o + 4); r.c = ld16_scalar(o + 5);
  l.x = ld16_scalar(x); r.x = ld16_scalar(x + 1);
}
__device__ __forceinline__ float packet_node_bound(const PacketNode &n, float qx, float qy, float qz) {
  return obb_dist2_axes(n.a, n.b, n.c, n.x, qx, qy, qz);
}

// Leaf scan of the packet walk: the leaf's points come through the scalar cache, four 16-byte fetches from ONE base
// address in flight (immediate offsets, a single wait).  The last batch is moved back so that it ends at the leaf's
// end instead of being range-checked per point: a point presented twice cannot change the result (d < best is
// strict), and the order of first presentations is the same as in the per-lane scan.  The kernel is bound by
// instruction issue (one slot per SIMD and 4 cycles, scalar instructions included), so per point this is 8 VALU for
// the distance + compare + two selects, no scalar address arithmetic and no branch; `leaf` is set once per leaf.
__device__ __forceinline__ void scan_leaf_uniform(const BvhView &t, uint32_t node, float qx, float qy, float qz, NearestVisitor &v) {
  const uint32_t j = node - (1u << t.depth);
  const uint32_t s = (uint32_t)(((unsigned long long)j * t.n) >> t.depth);
  const uint32_t e = (uint32_t)(((unsigned long long)(j + 1) * t.n) >> t.depth);
  const uint32_t pos0 = v.pos;
#define OPE_PKT_POINT(P, IDX)                                                                                  \
  {                                                                                                            \
    const float d_ = sq_dist3(__fsub_rn(qx, P.x), __fsub_rn(qy, P.y), __fsub_rn(qz, P.z));                     \
    const bool c_ = d_ < v.best;                                                                               \
    v.best = c_ ? d_ : v.best;                                                                                 \
    v.pos = c_ ? (IDX) : v.pos;                                                                                \
  }
  if (e - s >= 4u) {
    for (uint32_t i = s;; i += 4) {
      const bool last = i + 4u >= e;
      const uint32_t b = last ? e - 4u : i;
      const float4 *p = t.pts + b;
      const v4f p0 = ld16_scalar(p), p1 = ld16_scalar(p + 1), p2 = ld16_scalar(p + 2), p3 = ld16_scalar(p + 3);
      OPE_PKT_POINT(p0, b);
      OPE_PKT_POINT(p1, b + 1u);
      OPE_PKT_POINT(p2, b + 2u);
      OPE_PKT_POINT(p3, b + 3u);
      if (last) break;
    }
  } else {
    for (uint32_t i = s; i < e; ++i) {
      const v4f p0 = ld16_scalar(t.pts + i);
      OPE_PKT_POINT(p0, i);
    }
  }
#undef OPE_PKT_POINT
  v.leaf = (v.pos != pos0) ? node : v.leaf;
}

// The same scan for visitors that must be shown every point exactly once (k-nearest lists): guarded batches of four.
template <class Visitor>
__device__ __forceinline__ void scan_leaf_uniform_once(const BvhView &t, uint32_t node, float qx, float qy, float qz, Visitor &v) {
  const uint32_t j = node - (1u << t.depth);
  const uint32_t s = (uint32_t)(((unsigned long long)j * t.n) >> t.depth);
  const uint32_t e = (uint32_t)(((unsigned long long)(j + 1) * t.n) >> t.depth);
  for (uint32_t i = s; i < e; i += 4) {
    // a batch may run past the leaf (guarded) and, at the last leaf, into the kPtsPad zeroed entries
    const float4 *p = t.pts + i;
    const v4f p0 = ld16_scalar(p), p1 = ld16_scalar(p + 1), p2 = ld16_scalar(p + 2), p3 = ld16_scalar(p + 3);
    v.point(sq_dist3(__fsub_rn(qx, p0.x), __fsub_rn(qy, p0.y), __fsub_rn(qz, p0.z)), p0, i, node);
    if (i + 1 < e) v.point(sq_dist3(__fsub_rn(qx, p1.x), __fsub_rn(qy, p1.y), __fsub_rn(qz, p1.z)), p1, i + 1, node);
    if (i + 2 < e) v.point(sq_dist3(__fsub_rn(qx, p2.x), __fsub_rn(qy, p2.y), __fsub_rn(qz, p2.z)), p2, i + 2, node);
    if (i + 3 < e) v.point(sq_dist3(__fsub_rn(qx, p3.x), __fsub_rn(qy, p3.y), __fsub_rn(qz, p3.z)), p3, i + 3, node);
  }
}
template <class Visitor>
__device__ __forceinline__ void packet_scan_leaf(const BvhView &t, uint32_t node, float qx, float qy, float qz, Visitor &v) {
  if constexpr (leaf_rescan_is_harmless<Visitor>::value) scan_leaf_uniform(t, node, qx, qy, qz, v);
  else scan_leaf_uniform_once(t, node, qx, qy, qz, v);
}

template <class Visitor>
__device__ __forceinline__ bool bvh_traverse_packet(const BvhView &t, float qx, float qy, float qz, bool active, Visitor &v,
                                                    uint32_t hint, float *stk, int stk_stride, PacketStats *ps = nullptr) {
  const uint32_t leaf0 = 1u << t.depth;
  const int D = t.depth;
  const unsigned long long act = __ballot(active);
  if (t.axis2 == nullptr || act == 0ull || __ballot(active && hint == 0u) != 0ull) return false;
  // the distinct start leaves, scanned by every lane
  uint32_t seen[kPacketMaxLeaves];
  int nd = 0;
  unsigned long long todo = act;
  while (todo != 0ull) {
    if (nd == kPacketMaxLeaves) return false;
    const uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)hint, (int)__builtin_ctzll(todo));
    packet_scan_leaf(t, L, qx, qy, qz, v);
    todo &= ~__ballot(hint == L);
    seen[nd++] = L;
  }
  // per-lane bounds of the first leaf's ancestor siblings (that leaf plus those D subtrees cover the tree), two
  // scalar fetches (24 SGPRs) in flight, parked like the per-lane walk parks them
  const uint32_t anchor = seen[0];
  uint32_t node = anchor, trail = leaf0 - 1u;   // wave-uniform
  float minb = INFINITY;
  for (int k = 0; k < D; k += 2) {
    const uint32_t s0 = (anchor >> k) ^ 1u;
    const uint32_t s1 = (k + 1 < D) ? ((anchor >> (k + 1)) ^ 1u) : s0;
    PacketNode n0, n1;
    load_packet_node(t, s0, n0);
    load_packet_node(t, s1, n1);
    const float e0 = packet_node_bound(n0, qx, qy, qz), e1 = packet_node_bound(n1, qx, qy, qz);
    stk[(D - k) * stk_stride] = e0;
    if (k + 1 < D) stk[(D - k - 1) * stk_stride] = e1;
    minb = fminf(minb, fminf(e0, e1));
  }
  for (;;) {
    // back up to the deepest pending sibling that ANY lane can still improve in (inactive lanes carry best = -inf)
    if (__ballot(minb < v.bound()) == 0ull) return true;
    bool more = false;
    while (trail != 0u) {
      const int k = __builtin_ctz(trail);
      node = (node >> k) ^ 1u;
      trail = (trail >> k) & ~1u;
      if (__ballot(stk[(31 - __clz(node)) * stk_stride] < v.bound()) != 0ull) { more = true; break; }
    }
    if (!more) return true;
    OPE_PKT_COUNT(ps, backups);
    // walk down from there
    for (;;) {
      if (node >= leaf0) {
        bool dup = false;
        for (int q = 0; q < nd; ++q) dup = dup || (seen[q] == node);
        if (!dup) { packet_scan_leaf(t, node, qx, qy, qz, v); OPE_PKT_COUNT(ps, leaves); }
        break;
      }
      PacketNode cl, cr;
      load_packet_children(t, node, cl, cr);
      OPE_PKT_COUNT(ps, steps);
      const float d0 = packet_node_bound(cl, qx, qy, qz), d1 = packet_node_bound(cr, qx, qy, qz);
      const unsigned long long n0 = __ballot(d0 < v.bound()), n1 = __ballot(d1 < v.bound());
      if ((n0 | n1) == 0ull) break;
      // nearer child first by majority; the other one is parked if any lane wants it
      const bool right = (n0 == 0ull) || (n1 != 0ull && 2 * __popcll(__ballot(d1 < d0) & act) > __popcll(act));
      const bool pend = right ? (n0 != 0ull) : (n1 != 0ull);
      const float df = right ? d0 : d1;
      node = 2 * node + (right ? 1u : 0u);
      trail = (trail << 1) | (pend ? 1u : 0u);
      if (pend) { stk[(31 - __clz(node)) * stk_stride] = df; minb = fminf(minb, df); }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Group ("oct") traversal: EIGHT lanes serve ONE query.  Used for the chunks whose queries are far from
// the model surface (clutter): their private walks are 50-100 dependent trips long and set the
// duration of the whole search kernel.  Here every trip looks three levels ahead — lane g evaluates
// the box of descendant (node << 3) + g — the group jumps straight to the nearest of the eight,
// parks the (exact) bounds of the three sibling subtrees it skipped, and a leaf bucket is scanned
// with one 16-byte load per lane.  Same exact nearest neighbour; about a third of the trips.
// All lanes of a group carry identical control state; g = lane & 7.
__device__ __forceinline__ float group8_min(float v) {
  v = fminf(v, __shfl_xor(v, 1, 64));
  v = fminf(v, __shfl_xor(v, 2, 64));
  v = fminf(v, __shfl_xor(v, 4, 64));
  return v;
}

__device__ __forceinline__ void bvh_traverse_oct(const BvhView &t, float qx, float qy, float qz, NearestVisitor &v, float *stk,
                                                 int stk_stride, uint32_t start_leaf) {
  const uint32_t g = threadIdx.x & 7u;
  const uint32_t leaf0 = 1u << t.depth;
  const int D = t.depth;
  uint32_t node = 1, trail = 0;
  float minb = INFINITY;
  if (start_leaf != 0) {
    node = start_leaf;
    trail = leaf0 - 1u;
    // ancestor-sibling bounds: lane g takes levels g, g + 8 and g + 16
    float e0 = INFINITY, e1 = INFINITY;
    {
      v4f a, b, c;
      const int k0 = (int)g, k1 = (int)g + 8, k2 = (int)g + 16;
      const uint32_t s0 = (k0 < D) ? ((start_leaf >> k0) ^ 1u) : 1u;
      const uint32_t s1 = (k1 < D) ? ((start_leaf >> k1) ^ 1u) : 1u;
      load_node(t, s0, a, b, c);
      v4f a1, b1, c1;
      load_node(t, s1, a1, b1, c1);
      if (k0 < D) { e0 = obb_dist2(a, b, c, qx, qy, qz); stk[(D - k0) * stk_stride] = e0; }
      if (k1 < D) { e1 = obb_dist2(a1, b1, c1, qx, qy, qz); stk[(D - k1) * stk_stride] = e1; }
      if (D > 16) {   // deep trees only (more than a million points)
        const uint32_t s2 = (k2 < D) ? ((start_leaf >> k2) ^ 1u) : 1u;
        load_node(t, s2, a, b, c);
        if (k2 < D) { const float e2 = obb_dist2(a, b, c, qx, qy, qz); stk[(D - k2) * stk_stride] = e2; e1 = fminf(e1, e2); }
      }
    }
    minb = group8_min(fminf(e0, e1));
  } else {
    v4f a, b, c;
    load_node(t, 1, a, b, c);
    if (v.prune(obb_dist2(a, b, c, qx, qy, qz))) return;
  }
  for (;;) {
    if (node < leaf0) {
      const int d = 31 - __clz(node);
      const int L = min(3, D - d);               // levels to jump (group-uniform)
      const uint32_t cnt = 1u << L;
      const uint32_t mine = (node << L) + (g & (cnt - 1u));
      v4f a, b, c;
      load_node(t, mine, a, b, c);
      const float dg = (g < cnt) ? obb_dist2(a, b, c, qx, qy, qz) : INFINITY;
      const float dmin = group8_min(dg);
      if (!v.prune(dmin)) {
        // nearest descendant (lowest lane on ties), then the bounds of the skipped sibling subtrees
        const unsigned long long m = __ballot(dg == dmin) >> ((threadIdx.x & 63u) & ~7u);
        const uint32_t bsel = (uint32_t)__builtin_ctz((uint32_t)(m & 0xffu));
        for (int l = 1; l <= L; ++l) {
          const int sh = L - l;
          const bool in_sib = (g < cnt) && ((g >> sh) == ((bsel >> sh) ^ 1u));
          const float bl = group8_min(in_sib ? dg : INFINITY);
          const bool pend = !v.prune(bl);
          trail = (trail << 1) | (pend ? 1u : 0u);
          if (pend) { stk[(d + l) * stk_stride] = bl; minb = fminf(minb, bl); }
        }
        node = (node << L) + bsel;
        continue;
      }
    } else {
      const uint32_t j = node - leaf0;
      const uint32_t s = (uint32_t)(((unsigned long long)j * t.n) >> t.depth);
      const uint32_t e = (uint32_t)(((unsigned long long)(j + 1) * t.n) >> t.depth);
      for (uint32_t i = s; i < e; i += 16) {
        const uint32_t i0 = i + g, i1 = i + 8 + g;
        const v4f p0 = ld16(t.pts + min(i0, e - 1)), p1 = ld16(t.pts + min(i1, e - 1));
        float d0 = (i0 < e) ? sq_dist3(__fsub_rn(qx, p0.x), __fsub_rn(qy, p0.y), __fsub_rn(qz, p0.z)) : INFINITY;
        const float d1 = (i1 < e) ? sq_dist3(__fsub_rn(qx, p1.x), __fsub_rn(qy, p1.y), __fsub_rn(qz, p1.z)) : INFINITY;
        uint32_t ip = i0;
        if (d1 < d0) { d0 = d1; ip = i1; }
        const float dm = group8_min(d0);
        if (dm < v.best) {
          const unsigned long long m = __ballot(d0 == dm) >> ((threadIdx.x & 63u) & ~7u);
          const uint32_t w = (uint32_t)__builtin_ctz((uint32_t)(m & 0xffu));
          v.best = dm;
          v.pos = (uint32_t)__shfl((int)ip, (int)((threadIdx.x & 63u & ~7u) + w), 64);
          v.leaf = node;
        }
      }
    }
    if (v.prune(minb)) return;
    for (;;) {
      if (trail == 0) return;
      const int k = __builtin_ctz(trail);
      node = (node >> k) ^ 1u;
      trail = (trail >> k) & ~1u;
      if (!v.prune(stk[(31 - __clz(node)) * stk_stride])) break;
    }
  }
}

// k-nearest list of one lane, kept in LDS with a per-thread stride (bank-conflict free), ascending.
// `worst` starts at +inf for plain k-NN, or just above r^2 for "the k nearest within radius r".
struct KnnVisitor {
  float *d;       // &lds_d[threadIdx.x], element j at d[j*stride]
  uint32_t *pos;  // reordered target position
  int stride, k, count;
  float worst;
  __device__ __forceinline__ bool prune(float bound) const { return !(bound < worst); }
  __device__ __forceinline__ void point(float dist, const v4f &, uint32_t i, uint32_t) {
    if (!(dist < worst)) return;
    int j = (count < k) ? count++ : k - 1;
    while (j > 0 && d[(j - 1) * stride] > dist) {
      d[j * stride] = d[(j - 1) * stride];
      pos[j * stride] = pos[(j - 1) * stride];
      --j;
    }
    d[j * stride] = dist;
    pos[j * stride] = i;
    if (count == k) worst = d[(k - 1) * stride];
  }
  __device__ __forceinline__ void on_node() {}
};

// k-nearest list of one lane in REGISTERS, ascending, for a compile-time K (normal shooting's k = 20).  The LDS list
// above costs a data-dependent shifting loop per candidate, which under SIMT runs as long as the unluckiest lane
// needs (and its 64 KB per block hold the kernel at two waves per SIMD); here an insertion is a fixed, branch-free
// sequence of K compare/select steps, skipped for the whole wave when no lane has a candidate.  (Round 2 put a
// four-entry per-lane LDS queue in front of it, so that the wave inserts from all queues together instead of at nearly
// every presented point: normal shooting k = 20 on C3 0.612 ms per iteration against 0.619 without, k = 10 0.391 / 0.381,
// normals unchanged, eight entries 0.84 — the insertions are not what the k-NN walks wait for; not kept.  Round 3 repeated
// it with the queue drained only when a lane's is full and at the walk's end, exact to the last correspondence: 8 entries
// at four waves per SIMD 0.574 ms against 0.537, 14 entries at three waves 0.711; and with a 32-entry buffer selected from
// at the end, under the previous launch's bound: 2.05 ms.  Same conclusion.)
#ifdef OPE_KNN_STATS   // (developer counters: executions at wave level are counted by the first active lane)
#define OPE_KNN_STAT(J) do { if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__ballot(true))) ++stat[J]; } while (0)
#else
#define OPE_KNN_STAT(J) do { } while (0)
#endif
template <int K>
struct KnnRegVisitor {
  float d[K];
  uint32_t p[K];
  int count;
#ifdef OPE_KNN_STATS
  uint32_t stat[5] = {0, 0, 0, 0, 0};   // presentations, insertion sequences, node steps, - (wave level); passing candidates (per lane)
#endif
  uint32_t leaf;   // leaf of the current nearest entry (next iteration's start hint)
  // bound: only points strictly closer than this can enter the list (+inf: plain k-NN; a finite value that is known to
  // lie above the K-th neighbour's distance gives the same list and prunes from the first box on)
  __device__ __forceinline__ void init(bool active, float bound = INFINITY) {
#pragma unroll
    for (int j = 0; j < K; ++j) { d[j] = active ? bound : -INFINITY; p[j] = 0; }
    count = 0;
    leaf = 0;
  }
  __device__ __forceinline__ bool prune(float bound) const { return !(bound < d[K - 1]); }
  __device__ __forceinline__ float bound() const { return d[K - 1]; }
  __device__ __forceinline__ void point(float dist, const v4f &, uint32_t i, uint32_t lf) {
    const bool ins = dist < d[K - 1];
#ifdef OPE_KNN_STATS
    OPE_KNN_STAT(0);             // point presentations (wave level)
    if (ins) ++stat[4];          // candidates that pass (per lane)
#endif
    if (__ballot(ins) == 0ull) return;
    OPE_KNN_STAT(1);             // insertion sequences executed (wave level)
    if (ins && dist < d[0]) leaf = lf;
    count += (ins && count < K) ? 1 : 0;
    // Sorted insert of `dist` into ascending d[]: new d[j] = median(d[j-1], d[j], dist) — one v_med3_f32 per slot —
    // and the index follows the same move: p[j-1] where dist sorts before d[j-1], i where it lands, p[j] otherwise
    // (strict comparisons: an equal earlier entry stays in front, like the LDS list).  Lanes without a candidate
    // (dist >= d[K-1]) come out unchanged.
    bool lt_hi = dist < d[K - 1];   // dist < d[j]
#pragma unroll
    for (int j = K - 1; j > 0; --j) {
      const bool lt_lo = dist < d[j - 1];
      p[j] = lt_lo ? p[j - 1] : (lt_hi ? i : p[j]);
      d[j] = __builtin_amdgcn_fmed3f(d[j - 1], d[j], dist);
      lt_hi = lt_lo;
    }
    p[0] = lt_hi ? i : p[0];
    d[0] = fminf(d[0], dist);
  }
  __device__ __forceinline__ void on_node() { OPE_KNN_STAT(2); }
};

constexpr int kKnnBlock = 256;
constexpr int kKnnMaxK = 32;
constexpr size_t kKnnLdsBytes = (sizeof(float) + sizeof(uint32_t)) * kKnnBlock * kKnnMaxK;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sum over each row of 16 lanes with four DPP steps (pure VALU: quad_perm xor 1, xor 2, row_half_mirror,
// row_mirror); every lane of the row ends up with the row total.
template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_move_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_move_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_move_f64<0x141>(v);  // row_half_mirror
  v += dpp_move_f64<0x140>(v);  // row_mirror
  return v;
}

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Apply the 3x4 fp32 transform rows (r00 r01 r02 tx | r10 … | r20 …) with the oracle's operation
// order: ((r0*x + r1*y) + r2*z) + t, unfused.
__device__ __forceinline__ float xform_row(const float *r, float x, float y, float z) {
  return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(r[0], x), __fmul_rn(r[1], y)), __fmul_rn(r[2], z)), r[3]);
}
__device__ __forceinline__ float rot_row(const float *r, float x, float y, float z) {
  return __fadd_rn(__fadd_rn(__fmul_rn(r[0], x), __fmul_rn(r[1], y)), __fmul_rn(r[2], z));
}

}  // namespace ope
