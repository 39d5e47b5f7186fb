// bvh_build_device.hip — construction of the implicit balanced OBB tree on the device.
//
// Replaces the kd-tree build inside Registration::initCompute (vPCL impl/registration_mod.hpp:80-84:
// tree_->setInputCloud(target_)).  Same structure and the same conservative box margins as the host builder
// (bvh_build.cpp, kept as the A/B reference behind OPE_HOST_BUILD): a perfect binary tree whose node (l, j)
// owns the contiguous point range [start(j << (D-l)), start((j+1) << (D-l))), start(k) = k*n >> D.
//
//   for every level l < D:   per-node bounding box (wave-reduced atomics on order-preserving integer images of
//                            the floats) -> split axis = widest extent -> ONE rocPRIM radix sort of
//                            (node << 16 | 16-bit position along the node's axis): every node's range is sorted in
//                            place, so its lower half is its left child.  D sorts of n keys in total.
//   then, level by level:    one block per node (first ten levels of large clouds: see TopWork) fits the oriented box: mean and covariance in fp64, cyclic
//                            Jacobi, mid-range centre along the rounded axes, half extents about the ROUNDED
//                            centre plus the margin that covers the traversal's fp32 evaluation.
//
// The tree differs from the host builder's only where equal coordinates straddle a median; the search is exact
// for any tree whose boxes contain their points.
#include <cstring>
#include <string>

#include <rocprim/rocprim.hpp>

#include <cfloat>
#include <cmath>

#include "ope_internal.hpp"

namespace ope {

namespace {

constexpr int kFitBlock = 256;

__device__ __forceinline__ uint32_t enc_f32(float f) {
  const uint32_t b = (uint32_t)__float_as_int(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(uint32_t u) {
  const uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __int_as_float((int)b);
}

// leaf bucket that owns sorted position p: the largest j with (j*n >> D) <= p
__device__ __forceinline__ uint32_t leaf_of(uint32_t p, uint32_t n, int D) {
  return (uint32_t)(((((unsigned long long)p + 1ull) << D) - 1ull) / n);
}
__device__ __forceinline__ uint32_t leaf_start(unsigned long long j, uint32_t n, int D) { return (uint32_t)((j * n) >> D); }

constexpr int kBboxMaxRows = 16;   // consecutive 256-point rows one block walks (large clouds; 1 for small ones)

// Per-node bounding boxes of `level`.  A wave walks `rows` rows of the sorted order; as long as its 64 points
// stay inside one node it only keeps per-lane minima / maxima, and it issues ONE set of six atomics when the node
// changes or the tile ends (at the top levels every wave of a 4 M-point cloud used to hit the same six words).
__global__ __launch_bounds__(256) void level_bbox_kernel(const float4 *__restrict__ pts, const uint32_t *__restrict__ order, uint32_t n,
                                                          int D, int level, int rows, uint32_t *__restrict__ mn, uint32_t *__restrict__ mx) {
  constexpr uint32_t kNone = 0xffffffffu;
  uint32_t cur = kNone;   // wave-uniform: node of the run being accumulated
  float lx = INFINITY, ly = INFINITY, lz = INFINITY, hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
  auto flush = [&]() {
    if (cur == kNone) return;
    for (int off = 32; off >= 1; off >>= 1) {
      lx = fminf(lx, __shfl_xor(lx, off, 64)); ly = fminf(ly, __shfl_xor(ly, off, 64)); lz = fminf(lz, __shfl_xor(lz, off, 64));
      hx = fmaxf(hx, __shfl_xor(hx, off, 64)); hy = fmaxf(hy, __shfl_xor(hy, off, 64)); hz = fmaxf(hz, __shfl_xor(hz, off, 64));
    }
    if ((threadIdx.x & 63u) == 0u) {
      atomicMin(mn + 3 * cur + 0, enc_f32(lx)); atomicMin(mn + 3 * cur + 1, enc_f32(ly)); atomicMin(mn + 3 * cur + 2, enc_f32(lz));
      atomicMax(mx + 3 * cur + 0, enc_f32(hx)); atomicMax(mx + 3 * cur + 1, enc_f32(hy)); atomicMax(mx + 3 * cur + 2, enc_f32(hz));
    }
    cur = kNone;
    lx = ly = lz = INFINITY; hx = hy = hz = -INFINITY;
  };
  const unsigned long long base = (unsigned long long)blockIdx.x * (256ull * (unsigned)rows);
  for (int r = 0; r < rows; ++r) {
    // each wave walks its own contiguous 64*rows points, so consecutive rows mostly stay inside one node
    const unsigned long long pp = base + (unsigned long long)(threadIdx.x >> 6) * (64ull * (unsigned)rows) + (unsigned long long)r * 64ull + (threadIdx.x & 63u);
    const bool active = pp < n;
    const uint32_t p = (uint32_t)pp;
    const uint32_t node = active ? (leaf_of(p, n, D) >> (D - level)) : kNone;
    float x = 0.f, y = 0.f, z = 0.f;
    if (active) { const float4 q = pts[order[p]]; x = q.x; y = q.y; z = q.z; }
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)node);
    if (__ballot(node != first) == 0ull) {   // the whole wave in one node (or past the end)
      if (first == kNone) continue;
      if (first != cur) { flush(); cur = first; }
      lx = fminf(lx, x); ly = fminf(ly, y); lz = fminf(lz, z);
      hx = fmaxf(hx, x); hy = fmaxf(hy, y); hz = fmaxf(hz, z);
    } else {
      // several nodes in the wave (deep levels).  The lanes are sorted by node: segmented min / max scan over the
      // runs of equal node, and only the last lane of a run issues atomics
      flush();
      const uint32_t ln = threadIdx.x & 63u;
      float l0 = x, l1 = y, l2 = z, h0 = x, h1 = y, h2 = z;
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t on = (uint32_t)__shfl_up((int)node, off, 64);
        const float a0 = __shfl_up(l0, off, 64), a1 = __shfl_up(l1, off, 64), a2 = __shfl_up(l2, off, 64);
        const float b0 = __shfl_up(h0, off, 64), b1 = __shfl_up(h1, off, 64), b2 = __shfl_up(h2, off, 64);
        if (ln >= (uint32_t)off && on == node) {
          l0 = fminf(l0, a0); l1 = fminf(l1, a1); l2 = fminf(l2, a2);
          h0 = fmaxf(h0, b0); h1 = fmaxf(h1, b1); h2 = fmaxf(h2, b2);
        }
      }
      const uint32_t nxt = (uint32_t)__shfl_down((int)node, 1, 64);
      if (active && (ln == 63u || nxt != node)) {
        atomicMin(mn + 3 * node + 0, enc_f32(l0)); atomicMin(mn + 3 * node + 1, enc_f32(l1)); atomicMin(mn + 3 * node + 2, enc_f32(l2));
        atomicMax(mx + 3 * node + 0, enc_f32(h0)); atomicMax(mx + 3 * node + 1, enc_f32(h1)); atomicMax(mx + 3 * node + 2, enc_f32(h2));
      }
    }
  }
  // the runs still open at the end of the tile: merge the four waves' runs in LDS, one atomic set per distinct node
  // (at the top levels that is one set per block instead of one per wave)
  __shared__ uint32_t s_node[4];
  __shared__ float s_box[4][6];
  if (cur != kNone) {
    for (int off = 32; off >= 1; off >>= 1) {
      lx = fminf(lx, __shfl_xor(lx, off, 64)); ly = fminf(ly, __shfl_xor(ly, off, 64)); lz = fminf(lz, __shfl_xor(lz, off, 64));
      hx = fmaxf(hx, __shfl_xor(hx, off, 64)); hy = fmaxf(hy, __shfl_xor(hy, off, 64)); hz = fmaxf(hz, __shfl_xor(hz, off, 64));
    }
  }
  if ((threadIdx.x & 63u) == 0u) {
    const uint32_t wv = threadIdx.x >> 6;
    s_node[wv] = cur;
    s_box[wv][0] = lx; s_box[wv][1] = ly; s_box[wv][2] = lz; s_box[wv][3] = hx; s_box[wv][4] = hy; s_box[wv][5] = hz;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int a = 0; a < 4; ++a) {
      const uint32_t nd = s_node[a];
      if (nd == kNone) continue;
      float b0 = s_box[a][0], b1 = s_box[a][1], b2 = s_box[a][2], b3 = s_box[a][3], b4 = s_box[a][4], b5 = s_box[a][5];
      for (int c = a + 1; c < 4; ++c)
        if (s_node[c] == nd) {
          b0 = fminf(b0, s_box[c][0]); b1 = fminf(b1, s_box[c][1]); b2 = fminf(b2, s_box[c][2]);
          b3 = fmaxf(b3, s_box[c][3]); b4 = fmaxf(b4, s_box[c][4]); b5 = fmaxf(b5, s_box[c][5]);
          s_node[c] = kNone;
        }
      atomicMin(mn + 3 * nd + 0, enc_f32(b0)); atomicMin(mn + 3 * nd + 1, enc_f32(b1)); atomicMin(mn + 3 * nd + 2, enc_f32(b2));
      atomicMax(mx + 3 * nd + 0, enc_f32(b3)); atomicMax(mx + 3 * nd + 1, enc_f32(b4)); atomicMax(mx + 3 * nd + 2, enc_f32(b5));
    }
  }
}

__global__ __launch_bounds__(256) void level_key_kernel(const float4 *__restrict__ pts, const uint32_t *__restrict__ order, uint32_t n, int D,
                                                         int level, const uint32_t *__restrict__ mn, const uint32_t *__restrict__ mx,
                                                         unsigned long long *__restrict__ keys) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const uint32_t node = leaf_of(p, n, D) >> (D - level);
  const float ex = dec_f32(mx[3 * node + 0]) - dec_f32(mn[3 * node + 0]), ey = dec_f32(mx[3 * node + 1]) - dec_f32(mn[3 * node + 1]),
              ez = dec_f32(mx[3 * node + 2]) - dec_f32(mn[3 * node + 2]);
  int dim = 0;
  float e = ex;
  if (ey > e) { dim = 1; e = ey; }
  if (ez > e) dim = 2;
  const float4 q = pts[order[p]];
  const float c = dim == 0 ? q.x : (dim == 1 ? q.y : q.z);
  // 16-bit position inside the node's extent along its split axis instead of the full 32-bit coordinate: a third fewer
  // radix passes per level.  Points closer than extent / 65536 keep their previous order (the sort is stable); the
  // split is by rank either way, and the search is exact for any tree whose boxes contain their points.
  const float lo_d = dec_f32(mn[3 * node + dim]);
  const float ext = dim == 0 ? ex : (dim == 1 ? ey : ez);
  const float tq = ext > 0.f ? (c - lo_d) * (65535.0f / ext) : 0.f;
  const uint32_t qk = (uint32_t)fminf(fmaxf(tq, 0.f), 65535.0f);
  keys[p] = ((unsigned long long)node << 16) | (unsigned long long)qk;
}

// The same key in 32 bits for the levels where it fits (node < 2^level, level + 16 <= 32): a third less traffic per radix pass.
__global__ __launch_bounds__(256) void level_key32_kernel(const float4 *__restrict__ pts, const uint32_t *__restrict__ order, uint32_t n, int D,
                                                           int level, const uint32_t *__restrict__ mn, const uint32_t *__restrict__ mx,
                                                           uint32_t *__restrict__ keys) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const uint32_t node = leaf_of(p, n, D) >> (D - level);
  const float ex = dec_f32(mx[3 * node + 0]) - dec_f32(mn[3 * node + 0]), ey = dec_f32(mx[3 * node + 1]) - dec_f32(mn[3 * node + 1]),
              ez = dec_f32(mx[3 * node + 2]) - dec_f32(mn[3 * node + 2]);
  int dim = 0;
  float e = ex;
  if (ey > e) { dim = 1; e = ey; }
  if (ez > e) dim = 2;
  const float4 q = pts[order[p]];
  const float c = dim == 0 ? q.x : (dim == 1 ? q.y : q.z);
  const float lo_d = dec_f32(mn[3 * node + dim]);
  const float ext = dim == 0 ? ex : (dim == 1 ? ey : ez);
  const float tq = ext > 0.f ? (c - lo_d) * (65535.0f / ext) : 0.f;
  keys[p] = (node << 16) | (uint32_t)fminf(fmaxf(tq, 0.f), 65535.0f);
}

// ---- the bottom of the tree in ONE launch.  From the level on at which a node holds at most kBotPoints points, a block takes one
// node and carries its range through ALL remaining levels: per level the sub-nodes' bounding boxes (LDS atomics on the
// order-preserving images), the widest axis, the same 16-bit position key, and a stable block-wide radix sort of
// (sub-node, key) with the point's source index as payload — registers and LDS only, the points themselves are gathered from
// global memory (L2) once per level.  A level of the global loop above is a dozen launches (two fills, boxes, keys and the
// passes of a device-wide sort); for a cloud of 500 k points that loop now ends after eight levels instead of fifteen, and a
// cloud of up to kBotPoints points (key points, descriptor clouds) is ordered by a single block.
// 16-bit position of a point along the widest axis of its node's bounding box (level_key_kernel's key).  Not inlined: with
// this selection chain inlined into the eight-item loop below, instruction selection of ROCm 7.2's compiler crashes.
__device__ __noinline__ uint32_t split_key16(float x, float y, float z, float lx, float ly, float lz, float hx, float hy, float hz) {
  const float ex = hx - lx, ey = hy - ly, ez = hz - lz;
  const bool by = ey > ex;
  const float e1 = by ? ey : ex;
  const bool bz = ez > e1;
  const float cc = bz ? z : (by ? y : x), lo = bz ? lz : (by ? ly : lx), ext = bz ? ez : e1;
  const float tq = ext > 0.f ? (cc - lo) * (65535.0f / ext) : 0.f;
  return (uint32_t)fminf(fmaxf(tq, 0.f), 65535.0f);
}
constexpr int kBotItems = 8;
constexpr uint32_t kBotPoints = 256 * kBotItems;
constexpr int kBotMaxLevels = 11;   // sub-nodes of a block's node: at most 2^10 (rows of the box table)
__global__ __launch_bounds__(256) void bottom_levels_kernel(const float4 *__restrict__ pts, const uint32_t *__restrict__ order_in, uint32_t n, int D,
                                                             int L0, uint32_t *__restrict__ order_out) {
  using Sort = rocprim::block_radix_sort<uint32_t, 256, kBotItems, uint32_t>;
  __shared__ typename Sort::storage_type s_sort;
  __shared__ uint32_t s_mn[3u << (kBotMaxLevels - 1)], s_mx[3u << (kBotMaxLevels - 1)];
  const uint32_t j0 = blockIdx.x;   // node (L0, j0)
  const uint32_t b = leaf_start((unsigned long long)j0 << (D - L0), n, D), e = leaf_start((unsigned long long)(j0 + 1) << (D - L0), n, D);
  const uint32_t m = e - b;
  uint32_t idx[kBotItems], leaf[kBotItems];
#pragma unroll
  for (int i = 0; i < kBotItems; ++i) {   // blocked arrangement: item i of thread t is position t * kBotItems + i of the range
    const uint32_t p = threadIdx.x * kBotItems + i;
    idx[i] = p < m ? order_in[b + p] : 0xffffffffu;
    leaf[i] = p < m ? leaf_of(b + p, n, D) : 0u;
  }
  for (int level = L0; level < D; ++level) {
    const uint32_t nsub = 1u << (level - L0), sub0 = j0 << (level - L0);
    for (uint32_t k = threadIdx.x; k < nsub * 3u; k += 256u) { s_mn[k] = 0xffffffffu; s_mx[k] = 0u; }
    __syncthreads();
    float c[kBotItems][3];
    uint32_t sub[kBotItems];
#pragma unroll
    for (int i = 0; i < kBotItems; ++i) {
      const bool live = idx[i] != 0xffffffffu;
      sub[i] = live ? (leaf[i] >> (D - level)) - sub0 : 0u;
      const float4 q = pts[live ? idx[i] : 0u];
      c[i][0] = q.x; c[i][1] = q.y; c[i][2] = q.z;
      if (live) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { atomicMin(&s_mn[3u * sub[i] + d], enc_f32(c[i][d])); atomicMax(&s_mx[3u * sub[i] + d], enc_f32(c[i][d])); }
      }
    }
    __syncthreads();
    uint32_t keys[kBotItems];
#pragma unroll
    for (int i = 0; i < kBotItems; ++i) {
      const bool live = idx[i] != 0xffffffffu;
      const uint32_t sn = sub[i];
      const uint32_t q16 = split_key16(c[i][0], c[i][1], c[i][2], dec_f32(s_mn[3u * sn]), dec_f32(s_mn[3u * sn + 1u]), dec_f32(s_mn[3u * sn + 2u]),
                                       dec_f32(s_mx[3u * sn]), dec_f32(s_mx[3u * sn + 1u]), dec_f32(s_mx[3u * sn + 2u]));
      keys[i] = live ? ((sn << 16) | q16) : 0xffffffffu;   // (padding sorts behind every point)
    }
    __syncthreads();
    Sort().sort(keys, idx, s_sort, 0, 32);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < kBotItems; ++i) {
    const uint32_t p = threadIdx.x * kBotItems + i;
    if (p < m) order_out[b + p] = idx[i];
  }
}

__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t *v, uint32_t n, uint32_t value) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) v[i] = value;
}
__global__ __launch_bounds__(256) void fill_boxes_kernel(uint32_t *mn, uint32_t *mx, uint32_t n) {   // empty boxes: min above, max below every image
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) { mn[i] = 0xffffffffu; mx[i] = 0u; }
}
__global__ __launch_bounds__(256) void iota_u32_kernel(uint32_t *v, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) v[i] = i;
}

__global__ __launch_bounds__(256) void gather_points_kernel(const float4 *__restrict__ src, const float4 *__restrict__ src_nrm,
                                                             const uint32_t *__restrict__ order, uint32_t n, float4 *__restrict__ dst,
                                                             float4 *__restrict__ dst_nrm) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const uint32_t o = order[p];
  dst[p] = src[o];
  if (dst_nrm) { float4 m = src_nrm[o]; m.w = 0.f; dst_nrm[p] = m; }
}

// ---- block-wide reductions of a few doubles
template <class Op, int BLOCK = kFitBlock>
__device__ __forceinline__ double block_reduce(double v, double *s_tmp /*[BLOCK/64]*/, Op op) {
  for (int off = 32; off >= 1; off >>= 1) v = op(v, __shfl_xor(v, off, 64));
  if (BLOCK == 64) return v;   // one wave: the butterfly left the result in every lane
  __syncthreads();   // s_tmp may still be read from the previous call
  if ((threadIdx.x & 63u) == 0u) s_tmp[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = s_tmp[0];
  for (int w = 1; w < BLOCK / 64; ++w) r = op(r, s_tmp[w]);
  return r;
}
struct OpAdd { __device__ double operator()(double a, double b) const { return a + b; } };
struct OpMin { __device__ double operator()(double a, double b) const { return a < b ? a : b; } };
struct OpMax { __device__ double operator()(double a, double b) const { return a > b ? a : b; } };

// symmetric 3x3 eigen-decomposition (cyclic Jacobi, fp64); columns of V are eigenvectors
__device__ void jacobi_eig3_dev(double S[9], double V[9]) {
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 50; ++sweep) {
    const double off = fabs(S[1]) + fabs(S[2]) + fabs(S[5]);
    const double diag = fabs(S[0]) + fabs(S[4]) + fabs(S[8]);
    if (off <= 1e-300 || off <= 1e-16 * diag) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = S[3 * p + q];
        if (apq == 0.0) continue;
        const double theta = (S[3 * q + q] - S[3 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
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

// One block per node of `level`: oriented box of the node's points (final order), written in the 48-byte layout.
// BLOCK = 256 for the populous levels, 64 (one wave, no barriers in the reductions) where a node holds few points.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void fit_obb_kernel(const float4 *__restrict__ pts, uint32_t n, int D, int level, double scale,
                                                            float *__restrict__ nodes, float4 *__restrict__ axis2) {
  __shared__ double s_tmp[BLOCK / 64];
  __shared__ double s_A[9];
  __shared__ float s_cf[3];
  const uint32_t j = blockIdx.x;
  const uint32_t node = (1u << level) + j;
  const uint32_t b = leaf_start((unsigned long long)j << (D - level), n, D), e = leaf_start((unsigned long long)(j + 1) << (D - level), n, D);
  float *o = nodes + (size_t)kNodeFloats * node;
  if (e <= b) {  // empty leaf (n < 2^D): a box nothing can be close to
    if (threadIdx.x < kNodeFloats) {
      float v = 0.f;
      if (threadIdx.x < 3) v = 1e30f;
      if (threadIdx.x == 4 || threadIdx.x == 9) v = 1.f;
      o[threadIdx.x] = v;
    }
    if (threadIdx.x == 0) axis2[node] = make_float4(0.f, 0.f, 1.f, 0.f);
    return;
  }
  const double cnt = (double)(e - b);
  double sx = 0, sy = 0, sz = 0;
  for (uint32_t i = b + threadIdx.x; i < e; i += BLOCK) { const float4 p = pts[i]; sx += p.x; sy += p.y; sz += p.z; }
  const double mx = block_reduce<OpAdd, BLOCK>(sx, s_tmp, OpAdd()) / cnt, my = block_reduce<OpAdd, BLOCK>(sy, s_tmp, OpAdd()) / cnt, mz = block_reduce<OpAdd, BLOCK>(sz, s_tmp, OpAdd()) / cnt;
  double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
  for (uint32_t i = b + threadIdx.x; i < e; i += BLOCK) {
    const float4 p = pts[i];
    const double vx = p.x - mx, vy = p.y - my, vz = p.z - mz;
    c00 += vx * vx; c01 += vx * vy; c02 += vx * vz; c11 += vy * vy; c12 += vy * vz; c22 += vz * vz;
  }
  c00 = block_reduce<OpAdd, BLOCK>(c00, s_tmp, OpAdd()); c01 = block_reduce<OpAdd, BLOCK>(c01, s_tmp, OpAdd()); c02 = block_reduce<OpAdd, BLOCK>(c02, s_tmp, OpAdd());
  c11 = block_reduce<OpAdd, BLOCK>(c11, s_tmp, OpAdd()); c12 = block_reduce<OpAdd, BLOCK>(c12, s_tmp, OpAdd()); c22 = block_reduce<OpAdd, BLOCK>(c22, s_tmp, OpAdd());
  if (threadIdx.x == 0) {
    double C[9] = {c00, c01, c02, c01, c11, c12, c02, c12, c22}, V[9];
    jacobi_eig3_dev(C, V);
    // two axes as floats; the third is their cross product, as the traversal recomputes it
    float a0[3], a1[3];
    for (int d = 0; d < 3; ++d) { a0[d] = (float)V[3 * d + 0]; a1[d] = (float)V[3 * d + 1]; }
    for (int d = 0; d < 3; ++d) { s_A[d] = a0[d]; s_A[3 + d] = a1[d]; }
    s_A[6] = s_A[1] * s_A[5] - s_A[2] * s_A[4];
    s_A[7] = s_A[2] * s_A[3] - s_A[0] * s_A[5];
    s_A[8] = s_A[0] * s_A[4] - s_A[1] * s_A[3];
    o[4] = a0[0]; o[5] = a0[1]; o[6] = a0[2];
    o[8] = a1[0]; o[9] = a1[1]; o[10] = a1[2];
    axis2[node] = make_float4((float)s_A[6], (float)s_A[7], (float)s_A[8], 0.f);
  }
  __syncthreads();
  double A[9];
  for (int k = 0; k < 9; ++k) A[k] = s_A[k];
  double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
  for (uint32_t i = b + threadIdx.x; i < e; i += BLOCK) {
    const float4 p = pts[i];
    for (int k = 0; k < 3; ++k) {
      const double t = A[3 * k] * p.x + A[3 * k + 1] * p.y + A[3 * k + 2] * p.z;
      lo[k] = t < lo[k] ? t : lo[k];
      hi[k] = t > hi[k] ? t : hi[k];
    }
  }
  double mid[3];
  for (int k = 0; k < 3; ++k) mid[k] = 0.5 * (block_reduce<OpMin, BLOCK>(lo[k], s_tmp, OpMin()) + block_reduce<OpMax, BLOCK>(hi[k], s_tmp, OpMax()));
  if (threadIdx.x == 0)
    for (int d = 0; d < 3; ++d) s_cf[d] = (float)(mid[0] * A[d] + mid[1] * A[3 + d] + mid[2] * A[6 + d]);
  __syncthreads();
  const float cf0 = s_cf[0], cf1 = s_cf[1], cf2 = s_cf[2];
  // half extents about the ROUNDED centre, measured with the rounded axes
  double h[3] = {0, 0, 0}, far = 0;
  for (uint32_t i = b + threadIdx.x; i < e; i += BLOCK) {
    const float4 p = pts[i];
    const double vx = (double)p.x - cf0, vy = (double)p.y - cf1, vz = (double)p.z - cf2;
    const double r = sqrt(vx * vx + vy * vy + vz * vz);
    far = r > far ? r : far;
    for (int k = 0; k < 3; ++k) {
      const double t = fabs(A[3 * k] * vx + A[3 * k + 1] * vy + A[3 * k + 2] * vz);
      h[k] = t > h[k] ? t : h[k];
    }
  }
  far = block_reduce<OpMax, BLOCK>(far, s_tmp, OpMax());
  for (int k = 0; k < 3; ++k) h[k] = block_reduce<OpMax, BLOCK>(h[k], s_tmp, OpMax());
  if (threadIdx.x == 0) {
    // margin covering the traversal's fp32 evaluation of the projections (relative 4e-6 of the offset, i.e.
    // > 10 fp32 ulps, and an absolute floor), exactly as bvh_build.cpp
    const double margin = 4e-6 * far + 1e-7 * scale;
    o[0] = cf0; o[1] = cf1; o[2] = cf2;
    for (int k = 0; k < 3; ++k) o[4 * k + 3] = nextafterf((float)(h[k] + margin), FLT_MAX);
  }
}


// The levels whose nodes hold at most 64 points (the leaves and the two or three levels above them: most of the nodes): one LANE
// per node instead of one wave — a block of 64 threads per 14-point leaf kept 50 lanes idle through four passes and a Jacobi
// solve on lane 0 (leaf level of a 914 k-point index: 254 us).  Same formulas, margins and pass structure as fit_obb_kernel; the
// sums run in point order instead of a tree (boxes differ in their last bits, never in what they hold).
__global__ __launch_bounds__(64) void fit_obb_small_kernel(const float4 *__restrict__ pts, uint32_t n, int D, int level, double scale,
                                                           float *__restrict__ nodes, float4 *__restrict__ axis2) {
  const uint32_t j = blockIdx.x * 64u + threadIdx.x;
  if (j >= (1u << level)) return;
  const uint32_t node = (1u << level) + j;
  const uint32_t b = leaf_start((unsigned long long)j << (D - level), n, D), e = leaf_start((unsigned long long)(j + 1) << (D - level), n, D);
  float *o = nodes + (size_t)kNodeFloats * node;
  if (e <= b) {  // empty leaf (n < 2^D): a box nothing can be close to
    for (int k = 0; k < kNodeFloats; ++k) o[k] = k < 3 ? 1e30f : (k == 4 || k == 9) ? 1.f : 0.f;
    axis2[node] = make_float4(0.f, 0.f, 1.f, 0.f);
    return;
  }
  const double cnt = (double)(e - b);
  double sx = 0, sy = 0, sz = 0;
  for (uint32_t i = b; i < e; ++i) { const float4 p = pts[i]; sx += p.x; sy += p.y; sz += p.z; }
  const double mx = sx / cnt, my = sy / cnt, mz = sz / cnt;
  double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
  for (uint32_t i = b; i < e; ++i) {
    const float4 p = pts[i];
    const double vx = p.x - mx, vy = p.y - my, vz = p.z - mz;
    c00 += vx * vx; c01 += vx * vy; c02 += vx * vz; c11 += vy * vy; c12 += vy * vz; c22 += vz * vz;
  }
  double C[9] = {c00, c01, c02, c01, c11, c12, c02, c12, c22}, V[9];
  jacobi_eig3_dev(C, V);
  float a0[3], a1[3];
  for (int d = 0; d < 3; ++d) { a0[d] = (float)V[3 * d + 0]; a1[d] = (float)V[3 * d + 1]; }
  double A[9];
  for (int d = 0; d < 3; ++d) { A[d] = a0[d]; A[3 + d] = a1[d]; }
  A[6] = A[1] * A[5] - A[2] * A[4];
  A[7] = A[2] * A[3] - A[0] * A[5];
  A[8] = A[0] * A[4] - A[1] * A[3];
  double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
  for (uint32_t i = b; i < e; ++i) {
    const float4 p = pts[i];
    for (int k = 0; k < 3; ++k) {
      const double t = A[3 * k] * p.x + A[3 * k + 1] * p.y + A[3 * k + 2] * p.z;
      lo[k] = t < lo[k] ? t : lo[k];
      hi[k] = t > hi[k] ? t : hi[k];
    }
  }
  const double mid[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
  float cf[3];
  for (int d = 0; d < 3; ++d) cf[d] = (float)(mid[0] * A[d] + mid[1] * A[3 + d] + mid[2] * A[6 + d]);
  double h[3] = {0, 0, 0}, far = 0;
  for (uint32_t i = b; i < e; ++i) {
    const float4 p = pts[i];
    const double vx = (double)p.x - cf[0], vy = (double)p.y - cf[1], vz = (double)p.z - cf[2];
    const double r = sqrt(vx * vx + vy * vy + vz * vz);
    far = r > far ? r : far;
    for (int k = 0; k < 3; ++k) {
      const double t = fabs(A[3 * k] * vx + A[3 * k + 1] * vy + A[3 * k + 2] * vz);
      h[k] = t > h[k] ? t : h[k];
    }
  }
  const double margin = 4e-6 * far + 1e-7 * scale;   // as fit_obb_kernel
  o[0] = cf[0]; o[1] = cf[1]; o[2] = cf[2];
  o[4] = a0[0]; o[5] = a0[1]; o[6] = a0[2];
  o[8] = a1[0]; o[9] = a1[1]; o[10] = a1[2];
  for (int k = 0; k < 3; ++k) o[4 * k + 3] = nextafterf((float)(h[k] + margin), FLT_MAX);
  axis2[node] = make_float4((float)A[6], (float)A[7], (float)A[8], 0.f);
}

// ------------------------------------------------------------------------------------------------------------------
// Top of the tree for large clouds.  One block per node leaves the first levels to a handful of blocks (the root of a
// 4 M-point cloud: one block, four passes over every point, 17 ms).  For levels < kTopLevels the fit is therefore
// done from the 2^kTopLevels node ranges of level kTopLevels ("slices", one block each):
//   moments   per slice: sum of d and d dT about a fixed pivot, fp64                       (top_moments_kernel)
//   combine   pairwise up the tree in a fixed order: deterministic                          (top_combine_kernel)
//   axes      per node: mean, covariance, Jacobi, rounded axes                              (top_axes_kernel)
//   range     per slice and ancestor: min / max of the projections -> atomicMin/Max on
//             order-preserving integer images (exact, order-independent)                    (top_range_kernel)
//   centre    per node: mid-range point, rounded to float                                   (top_centre_kernel)
//   extent    per slice and ancestor: half extents about the ROUNDED centre, farthest point (top_extent_kernel)
//   write     per node: margins, 48-byte record                                             (top_write_kernel)
// Same formulas and margins as fit_obb_kernel; only the mean / covariance come from raw moments instead of two
// passes (axes may differ in the last bits, the boxes contain their points either way).
constexpr int kTopLevels = 10;
constexpr uint32_t kTopSlices = 1u << kTopLevels;
constexpr size_t kTopMinPoints = 32768;

__device__ __forceinline__ unsigned long long enc_f64(double d) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(d);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double dec_f64(unsigned long long u) {
  const unsigned long long b = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
  return __longlong_as_double((long long)b);
}

struct TopWork {
  double mom[2 * kTopSlices][9];            // heap-indexed raw moments: sum d (3), sum d dT (xx xy xz yy yz zz)
  double A[kTopSlices][9];                  // rows: rounded axis 0, axis 1, their cross product (nodes 1..kTopSlices-1)
  unsigned long long lo[kTopSlices][3];     // order-preserving images (memset 0xff = above every image)
  unsigned long long hi[kTopSlices][3];     // (memset 0 = below every image)
  unsigned long long ext[kTopSlices][4];    // bit patterns of non-negative doubles h0 h1 h2 far (memset 0)
  float cf[kTopSlices][4];
};

__device__ __forceinline__ void slice_range(uint32_t s, uint32_t n, int D, uint32_t &b, uint32_t &e) {
  const int sh = D - kTopLevels;
  b = leaf_start((unsigned long long)s << sh, n, D);
  e = leaf_start((unsigned long long)(s + 1) << sh, n, D);
}

__global__ __launch_bounds__(kFitBlock) void top_moments_kernel(const float4 *__restrict__ pts, uint32_t n, int D, double px, double py,
                                                                double pz, TopWork *__restrict__ w) {
  __shared__ double s_tmp[kFitBlock / 64];
  uint32_t b, e;
  slice_range(blockIdx.x, n, D, b, e);
  double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t i = b + threadIdx.x; i < e; i += kFitBlock) {
    const float4 p = pts[i];
    const double dx = p.x - px, dy = p.y - py, dz = p.z - pz;
    m[0] += dx; m[1] += dy; m[2] += dz;
    m[3] += dx * dx; m[4] += dx * dy; m[5] += dx * dz; m[6] += dy * dy; m[7] += dy * dz; m[8] += dz * dz;
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const double r = block_reduce(m[k], s_tmp, OpAdd());
    if (threadIdx.x == 0) w->mom[kTopSlices + blockIdx.x][k] = r;
  }
}

__global__ __launch_bounds__(1024) void top_combine_kernel(TopWork *__restrict__ w) {
  for (int l = kTopLevels - 1; l >= 0; --l) {
    if (threadIdx.x < (1u << l)) {
      const uint32_t node = (1u << l) + threadIdx.x;
      for (int k = 0; k < 9; ++k) w->mom[node][k] = w->mom[2 * node][k] + w->mom[2 * node + 1][k];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kFitBlock) void top_axes_kernel(uint32_t n, int D, TopWork *__restrict__ w, float *__restrict__ nodes,
                                                             float4 *__restrict__ axis2) {
  const uint32_t node = blockIdx.x * kFitBlock + threadIdx.x;
  if (node == 0 || node >= kTopSlices) return;
  const int l = 31 - __clz(node);
  const uint32_t j = node - (1u << l);
  const uint32_t b = leaf_start((unsigned long long)j << (D - l), n, D), e = leaf_start((unsigned long long)(j + 1) << (D - l), n, D);
  const double cnt = (double)(e - b);
  const double *m = w->mom[node];
  const double mx = m[0] / cnt, my = m[1] / cnt, mz = m[2] / cnt;   // mean, relative to the pivot
  double C[9], V[9];
  C[0] = m[3] - cnt * mx * mx; C[1] = m[4] - cnt * mx * my; C[2] = m[5] - cnt * mx * mz;
  C[4] = m[6] - cnt * my * my; C[5] = m[7] - cnt * my * mz; C[8] = m[8] - cnt * mz * mz;
  C[3] = C[1]; C[6] = C[2]; C[7] = C[5];
  jacobi_eig3_dev(C, V);
  float a0[3], a1[3];
  for (int d = 0; d < 3; ++d) { a0[d] = (float)V[3 * d + 0]; a1[d] = (float)V[3 * d + 1]; }
  double A[9];
  for (int d = 0; d < 3; ++d) { A[d] = a0[d]; A[3 + d] = a1[d]; }
  A[6] = A[1] * A[5] - A[2] * A[4];
  A[7] = A[2] * A[3] - A[0] * A[5];
  A[8] = A[0] * A[4] - A[1] * A[3];
  for (int k = 0; k < 9; ++k) w->A[node][k] = A[k];
  float *o = nodes + (size_t)kNodeFloats * node;
  o[4] = a0[0]; o[5] = a0[1]; o[6] = a0[2];
  o[8] = a1[0]; o[9] = a1[1]; o[10] = a1[2];
  axis2[node] = make_float4((float)A[6], (float)A[7], (float)A[8], 0.f);
}

__global__ __launch_bounds__(kFitBlock) void top_range_kernel(const float4 *__restrict__ pts, uint32_t n, int D, TopWork *__restrict__ w) {
  __shared__ double s_tmp[kFitBlock / 64];
  uint32_t b, e;
  slice_range(blockIdx.x, n, D, b, e);
  for (int l = 0; l < kTopLevels; ++l) {
    const uint32_t node = (1u << l) + (blockIdx.x >> (kTopLevels - l));
    double A[9];
    for (int k = 0; k < 9; ++k) A[k] = w->A[node][k];
    double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (uint32_t i = b + threadIdx.x; i < e; i += kFitBlock) {
      const float4 p = pts[i];
      for (int k = 0; k < 3; ++k) {
        const double t = A[3 * k] * p.x + A[3 * k + 1] * p.y + A[3 * k + 2] * p.z;
        lo[k] = t < lo[k] ? t : lo[k];
        hi[k] = t > hi[k] ? t : hi[k];
      }
    }
    for (int k = 0; k < 3; ++k) {
      const double rl = block_reduce(lo[k], s_tmp, OpMin()), rh = block_reduce(hi[k], s_tmp, OpMax());
      if (threadIdx.x == 0 && e > b) {
        atomicMin(&w->lo[node][k], enc_f64(rl));
        atomicMax(&w->hi[node][k], enc_f64(rh));
      }
    }
  }
}

__global__ __launch_bounds__(kFitBlock) void top_centre_kernel(TopWork *__restrict__ w) {
  const uint32_t node = blockIdx.x * kFitBlock + threadIdx.x;
  if (node == 0 || node >= kTopSlices) return;
  const double *A = w->A[node];
  double mid[3];
  for (int k = 0; k < 3; ++k) mid[k] = 0.5 * (dec_f64(w->lo[node][k]) + dec_f64(w->hi[node][k]));
  for (int d = 0; d < 3; ++d) w->cf[node][d] = (float)(mid[0] * A[d] + mid[1] * A[3 + d] + mid[2] * A[6 + d]);
}

__global__ __launch_bounds__(kFitBlock) void top_extent_kernel(const float4 *__restrict__ pts, uint32_t n, int D, TopWork *__restrict__ w) {
  __shared__ double s_tmp[kFitBlock / 64];
  uint32_t b, e;
  slice_range(blockIdx.x, n, D, b, e);
  for (int l = 0; l < kTopLevels; ++l) {
    const uint32_t node = (1u << l) + (blockIdx.x >> (kTopLevels - l));
    double A[9];
    for (int k = 0; k < 9; ++k) A[k] = w->A[node][k];
    const float cf0 = w->cf[node][0], cf1 = w->cf[node][1], cf2 = w->cf[node][2];
    double h[3] = {0, 0, 0}, far = 0;
    for (uint32_t i = b + threadIdx.x; i < e; i += kFitBlock) {
      const float4 p = pts[i];
      const double vx = (double)p.x - cf0, vy = (double)p.y - cf1, vz = (double)p.z - cf2;
      const double r = sqrt(vx * vx + vy * vy + vz * vz);
      far = r > far ? r : far;
      for (int k = 0; k < 3; ++k) {
        const double t = fabs(A[3 * k] * vx + A[3 * k + 1] * vy + A[3 * k + 2] * vz);
        h[k] = t > h[k] ? t : h[k];
      }
    }
    far = block_reduce(far, s_tmp, OpMax());
    for (int k = 0; k < 3; ++k) h[k] = block_reduce(h[k], s_tmp, OpMax());
    if (threadIdx.x == 0) {   // non-negative doubles order like their bit patterns
      for (int k = 0; k < 3; ++k) atomicMax(&w->ext[node][k], (unsigned long long)__double_as_longlong(h[k]));
      atomicMax(&w->ext[node][3], (unsigned long long)__double_as_longlong(far));
    }
  }
}

__global__ __launch_bounds__(kFitBlock) void top_write_kernel(const TopWork *__restrict__ w, double scale, float *__restrict__ nodes) {
  const uint32_t node = blockIdx.x * kFitBlock + threadIdx.x;
  if (node == 0 || node >= kTopSlices) return;
  float *o = nodes + (size_t)kNodeFloats * node;
  const double far = __longlong_as_double((long long)w->ext[node][3]);
  const double margin = 4e-6 * far + 1e-7 * scale;   // as fit_obb_kernel / bvh_build.cpp
  o[0] = w->cf[node][0]; o[1] = w->cf[node][1]; o[2] = w->cf[node][2];
  for (int k = 0; k < 3; ++k) o[4 * k + 3] = nextafterf((float)(__longlong_as_double((long long)w->ext[node][k]) + margin), FLT_MAX);
}

}  // namespace

// The level sorts: rocPRIM's default takes its merge sort up to 2^20 items — two dozen launches of 5-8 us for a 17..25-bit key over
// 913 k points (165 us per level, nine levels: half of that index's build) — where three or four Onesweep passes do.
// Measured (ope_index_build, ms, default / Onesweep above 64 k items): 100 k points 0.92 / 1.15, 500 k 2.1 / 1.98, 1 M 2.75 / 2.5,
// the outlier filter of the C3 frame (914 k points) 5.45 / 4.85: Onesweep from 256 k items on.
#ifndef OPE_LEVEL_MERGE_SORT_LIMIT
#define OPE_LEVEL_MERGE_SORT_LIMIT 262144
#endif
using LevelSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, OPE_LEVEL_MERGE_SORT_LIMIT>;

// d_src: n finite points (float4, w = original index bits) in any order, d_src_nrm optional (same order).
// Allocates *d_nodes ((2 << D) * 48 B), *d_pts and (if normals) *d_nrm.
// tmp_alloc: the index is a temporary of one entry point (the outlier filter's, a feature stage's): its buffers come from the
// stream's cache of temporaries (tmp_malloc) instead of hipMalloc — three allocations and, later, three device-synchronising frees of
// 2-15 MB each, 0.2-0.3 ms apiece (ope_index::tmp_alloc, ope_index_free).
hipError_t build_bvh_device(hipStream_t stream, const float4 *d_src, const float4 *d_src_nrm, size_t n, int leaf_size,
                            const float bb_lo[3], const float bb_hi[3], int *out_depth, float4 **d_nodes, float4 **d_pts,
                            float4 **d_nrm, float4 **d_axis2, bool tmp_alloc) {
  auto alloc = [&](void **p, size_t bytes) { return tmp_alloc ? tmp_malloc(stream, p, bytes) : hipMalloc(p, bytes); };
  if (leaf_size < 1) leaf_size = 16;
  int D = 0;
  while (((n + ((size_t)1 << D) - 1) >> D) > (size_t)leaf_size) ++D;
  if (D > kMaxDepth) D = kMaxDepth;
  *out_depth = D;
  const size_t n_nodes = (size_t)2 << D;
  const double scale = std::max({(double)bb_hi[0] - bb_lo[0], (double)bb_hi[1] - bb_lo[1], (double)bb_hi[2] - bb_lo[2], 1e-3}) +
                       std::max({std::fabs((double)bb_lo[0]), std::fabs((double)bb_hi[0]), std::fabs((double)bb_lo[1]),
                                 std::fabs((double)bb_hi[1]), std::fabs((double)bb_lo[2]), std::fabs((double)bb_hi[2])});
  unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
  uint32_t *d_order = nullptr, *d_order2 = nullptr, *d_mn = nullptr, *d_mx = nullptr;
  void *d_tmp = nullptr;
  TopWork *d_top = nullptr;
  const uint32_t nn = (uint32_t)n;
  const unsigned nb = (unsigned)((n + 255) / 256);
  const int bbox_rows = (int)std::max<size_t>(1, std::min<size_t>(kBboxMaxRows, n / (256 * 1024)));   // keep >= ~1024 blocks
  const unsigned nb_bbox = (unsigned)((n + 256 * (size_t)bbox_rows - 1) / (256 * (size_t)bbox_rows));
  hipError_t e = alloc((void **)d_nodes, n_nodes * kNodeFloats * sizeof(float));
  if (e == hipSuccess) e = alloc((void **)d_axis2, n_nodes * sizeof(float4));
  if (e == hipSuccess) e = alloc((void **)d_pts, sizeof(float4) * (n + kPtsPad));
  if (e == hipSuccess) e = hipMemsetAsync(*d_pts + n, 0, sizeof(float4) * kPtsPad, stream);
  if (e == hipSuccess && d_src_nrm) e = alloc((void **)d_nrm, sizeof(float4) * n);
  if (e == hipSuccess) e = tmp_malloc(stream, (void **)&d_keys, 8 * n);
  if (e == hipSuccess) e = tmp_malloc(stream, (void **)&d_keys2, 8 * n);
  if (e == hipSuccess) e = tmp_malloc(stream, (void **)&d_order, 4 * n);
  if (e == hipSuccess) e = tmp_malloc(stream, (void **)&d_order2, 4 * n);
  if (e == hipSuccess) e = tmp_malloc(stream, (void **)&d_mn, 12 * ((size_t)1 << D));
  if (e == hipSuccess) e = tmp_malloc(stream, (void **)&d_mx, 12 * ((size_t)1 << D));
  // the levels whose nodes hold more than kBotPoints points: one device-wide sort each; everything below: one launch
  int L0 = 0;
  while (L0 < D && ((n + ((size_t)1 << L0) - 1) >> L0) > (size_t)kBotPoints) ++L0;
  if (D - L0 > kBotMaxLevels) L0 = D - kBotMaxLevels;   // (leaf size 1 on a small cloud: the kernel's box table has 2^10 rows)
  uint32_t *d_keys32 = reinterpret_cast<uint32_t *>(d_keys), *d_keys32b = reinterpret_cast<uint32_t *>(d_keys2);
  size_t tmp_bytes = 0, tmp_bytes32 = 0;
  if (e == hipSuccess) e = rocprim::radix_sort_pairs<LevelSortConfig>(nullptr, tmp_bytes, d_keys, d_keys2, d_order, d_order2, n, 0, 64, stream);
  if (e == hipSuccess) e = rocprim::radix_sort_pairs<LevelSortConfig>(nullptr, tmp_bytes32, d_keys32, d_keys32b, d_order, d_order2, n, 0, 32, stream);
  tmp_bytes = std::max(tmp_bytes, tmp_bytes32);
  if (e == hipSuccess) e = tmp_malloc(stream, &d_tmp, tmp_bytes);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(iota_u32_kernel, dim3(nb), dim3(256), 0, stream, d_order, nn);
    for (int level = 0; level < L0 && e == hipSuccess; ++level) {
      const uint32_t cnt = 3u << level;
      hipLaunchKernelGGL(fill_boxes_kernel, dim3((cnt + 255) / 256), dim3(256), 0, stream, d_mn, d_mx, cnt);
      hipLaunchKernelGGL(level_bbox_kernel, dim3(nb_bbox), dim3(256), 0, stream, d_src, d_order, nn, D, level, bbox_rows, d_mn, d_mx);
      size_t tb = tmp_bytes;
      if (level + 16 <= 32) {
        hipLaunchKernelGGL(level_key32_kernel, dim3(nb), dim3(256), 0, stream, d_src, d_order, nn, D, level, d_mn, d_mx, d_keys32);
        e = rocprim::radix_sort_pairs<LevelSortConfig>(d_tmp, tb, d_keys32, d_keys32b, d_order, d_order2, n, 0, 16 + std::max(level, 1), stream);
      } else {
        hipLaunchKernelGGL(level_key_kernel, dim3(nb), dim3(256), 0, stream, d_src, d_order, nn, D, level, d_mn, d_mx, d_keys);
        e = rocprim::radix_sort_pairs<LevelSortConfig>(d_tmp, tb, d_keys, d_keys2, d_order, d_order2, n, 0, 16 + std::max(level, 1), stream);
      }
      std::swap(d_order, d_order2);
    }
    if (e == hipSuccess && L0 < D) {
      hipLaunchKernelGGL(bottom_levels_kernel, dim3(1u << L0), dim3(256), 0, stream, d_src, d_order, nn, D, L0, d_order2);
      std::swap(d_order, d_order2);
    }
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(gather_points_kernel, dim3(nb), dim3(256), 0, stream, d_src, d_src_nrm, d_order, nn, *d_pts,
                       d_src_nrm ? *d_nrm : nullptr);
    int first_level = 0;
    if (n >= kTopMinPoints && D > kTopLevels) {   // first levels from 2^kTopLevels slices (see TopWork)
      e = tmp_malloc(stream, (void **)&d_top, sizeof(TopWork));
      if (e == hipSuccess) e = hipMemsetAsync(d_top, 0, sizeof(TopWork), stream);
      if (e == hipSuccess) e = hipMemsetAsync(d_top->lo, 0xff, sizeof d_top->lo, stream);
      if (e == hipSuccess) {
        const double px = 0.5 * ((double)bb_lo[0] + bb_hi[0]), py = 0.5 * ((double)bb_lo[1] + bb_hi[1]), pz = 0.5 * ((double)bb_lo[2] + bb_hi[2]);
        float *nodes_f = reinterpret_cast<float *>(*d_nodes);
        const unsigned per_node_blocks = kTopSlices / kFitBlock;
        hipLaunchKernelGGL(top_moments_kernel, dim3(kTopSlices), dim3(kFitBlock), 0, stream, *d_pts, nn, D, px, py, pz, d_top);
        hipLaunchKernelGGL(top_combine_kernel, dim3(1), dim3(1024), 0, stream, d_top);
        hipLaunchKernelGGL(top_axes_kernel, dim3(per_node_blocks), dim3(kFitBlock), 0, stream, nn, D, d_top, nodes_f, *d_axis2);
        hipLaunchKernelGGL(top_range_kernel, dim3(kTopSlices), dim3(kFitBlock), 0, stream, *d_pts, nn, D, d_top);
        hipLaunchKernelGGL(top_centre_kernel, dim3(per_node_blocks), dim3(kFitBlock), 0, stream, d_top);
        hipLaunchKernelGGL(top_extent_kernel, dim3(kTopSlices), dim3(kFitBlock), 0, stream, *d_pts, nn, D, d_top);
        hipLaunchKernelGGL(top_write_kernel, dim3(per_node_blocks), dim3(kFitBlock), 0, stream, d_top, scale, nodes_f);
        first_level = kTopLevels;
      }
    }
    for (int level = first_level; level <= D && e == hipSuccess; ++level) {
      // (one lane per node pays once a level has lanes for the GPU: 800-point index 0.195 -> 0.232 ms with it, 4 M points 6.0 -> 4.5)
      if (((n + ((size_t)1 << level) - 1) >> level) <= 64 && level >= 12)
        hipLaunchKernelGGL(fit_obb_small_kernel, dim3(((1u << level) + 63u) / 64u), dim3(64), 0, stream, *d_pts, nn, D, level, scale,
                           reinterpret_cast<float *>(*d_nodes), *d_axis2);
      else if ((n >> level) > 512)
        hipLaunchKernelGGL(fit_obb_kernel<kFitBlock>, dim3(1u << level), dim3(kFitBlock), 0, stream, *d_pts, nn, D, level, scale,
                           reinterpret_cast<float *>(*d_nodes), *d_axis2);
      else
        hipLaunchKernelGGL(fit_obb_kernel<64>, dim3(1u << level), dim3(64), 0, stream, *d_pts, nn, D, level, scale,
                           reinterpret_cast<float *>(*d_nodes), *d_axis2);
    }
    // node 0 is unused: give it the root box so stray reads are harmless
    e = hipMemcpyAsync(*d_nodes, reinterpret_cast<float *>(*d_nodes) + kNodeFloats, kNodeFloats * sizeof(float), hipMemcpyDeviceToDevice,
                       stream);
    if (e == hipSuccess) e = hipMemcpyAsync(*d_axis2, *d_axis2 + 1, sizeof(float4), hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
  }
  for (void *p : {(void *)d_keys, (void *)d_keys2, (void *)d_order, (void *)d_order2, (void *)d_mn, (void *)d_mx, d_tmp, (void *)d_top})
    tmp_free(stream, p);
  return e;
}

}  // namespace ope
