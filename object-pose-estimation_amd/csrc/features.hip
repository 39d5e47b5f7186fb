// features.hip — coarse-stage kernels: radius search, surface normals, SPFH/FPFH descriptors and the
// SAC-IA hypothesis scoring (gfx950, wave64).  All neighbourhood queries walk the same OBB tree as ICP.
//
// Reference call sites (DetectAndLocalize/src/poseestimator.cpp):
//   pcl::NormalEstimation(k=30)::compute                  :151-156
//   pcl::FPFHEstimation(r=0.03)::compute                  :121-125
//   pcl::SampleConsensusInitialAlignment::align           :50-64
// The arithmetic restated here is PCL 1.7.x's (normal_3d.hpp, centroid.hpp, eigen.hpp, fpfh.hpp,
// pfh.cpp, ia_ransac.hpp), including its quirks.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <vector>

#include "bvh_traverse.hpp"
#include "libm_f32.hpp"

namespace ope {

constexpr int kFeatBlock = 256;
constexpr int kSpfhStride = 36;  // 33 bins padded to 9 x 16 bytes

// ------------------------------------------------------------------------------------------ radius
struct RadiusVisitor {
  float r2;
  int count;
  KnnVisitor knn;  // keeps the knn.k nearest of those within r (k may be 0)
  __device__ __forceinline__ bool prune(float bound) const { return bound > r2; }
  __device__ __forceinline__ void point(float d, const v4f &p, uint32_t i, uint32_t lf) {
    if (d <= r2) {
      ++count;
      if (knn.k > 0) knn.point(d, p, i, lf);
    }
  }
  __device__ __forceinline__ void on_node() {}
};

__global__ __launch_bounds__(kKnnBlock) void radius_search_kernel(CloudView q, BvhView tgt, float r2, int max_nn,
                                                                   int32_t *__restrict__ counts,
                                                                   int32_t *__restrict__ out_idx,
                                                                   float *__restrict__ out_d2) {
  extern __shared__ unsigned char s_dyn[];
  float *ld = reinterpret_cast<float *>(s_dyn) + threadIdx.x;
  uint32_t *lp = reinterpret_cast<uint32_t *>(s_dyn + sizeof(float) * kKnnBlock * kKnnMaxK) + threadIdx.x;
  __shared__ float s_stk[kMaxDepth + 1][kKnnBlock];
  float *stk = &s_stk[0][threadIdx.x];
  for (uint32_t i = blockIdx.x * kKnnBlock + threadIdx.x; i < q.n; i += gridDim.x * kKnnBlock) {
    RadiusVisitor v{r2, 0, KnnVisitor{ld, lp, kKnnBlock, max_nn, 0, INFINITY}};
    if (i < q.n_valid) {
      const float4 s = q.xyzw[i];
      bvh_traverse(tgt, s.x, s.y, s.z, v, stk, kKnnBlock);
    }
    counts[i] = v.count;
    for (int j = 0; j < max_nn; ++j) {
      const bool have = j < v.knn.count;
      out_idx[(size_t)i * max_nn + j] = have ? __float_as_int(tgt.pts[lp[j * kKnnBlock]].w) : -1;
      out_d2[(size_t)i * max_nn + j] = have ? ld[j * kKnnBlock] : INFINITY;
    }
  }
}

// ------------------------------------------------------------------------------------------ self-search start leaves
// Normals, outlier removal and the like search a cloud against an index over THE SAME cloud: every query is itself a point of
// some leaf, and a walk that starts there (scan the leaf, then only the ancestor siblings that can still matter, as the ICP
// walks start from last iteration's leaf) skips the descent from the root.  One thread per leaf writes its heap id under the
// original index of each of its points.
__global__ __launch_bounds__(256) void self_leaf_kernel(BvhView t, uint32_t *__restrict__ leaf_of_orig) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= (1u << t.depth)) return;
  const uint32_t s = (uint32_t)(((unsigned long long)j * t.n) >> t.depth), e = (uint32_t)(((unsigned long long)(j + 1) * t.n) >> t.depth);
  for (uint32_t p = s; p < e; ++p) leaf_of_orig[(uint32_t)__float_as_int(t.pts[p].w)] = (1u << t.depth) + j;
}
// leaf_of_orig: n_total words, zeroed here (0 = not in the index: the walk starts at the root)
hipError_t self_leaves(hipStream_t stream, const BvhView &t, size_t n_total, uint32_t *leaf_of_orig) {
  hipError_t e = hipMemsetAsync(leaf_of_orig, 0, 4 * n_total, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(self_leaf_kernel, dim3(((1u << t.depth) + 255u) / 256u), dim3(256), 0, stream, t, leaf_of_orig);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ normals
// pcl::eigen33 / computeRoots (common/impl/eigen.hpp), Scalar = float
__device__ void compute_roots2(float b, float c, float roots[3]) {
  roots[0] = 0.f;
  float d = (float)(b * b - 4.0 * c);
  if (d < 0.0f) d = 0.0f;
  const float sd = sqrtf(d);
  roots[2] = 0.5f * (b + sd);
  roots[1] = 0.5f * (b - sd);
}

__device__ void compute_roots(const float m[9], float roots[3]) {
  const float c0 = m[0] * m[4] * m[8] + 2.f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
  const float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
  const float c2 = m[0] + m[4] + m[8];
  if (fabsf(c0) < 1.1920929e-07f) {
    compute_roots2(c2, c1, roots);
    return;
  }
  const float s_inv3 = (float)(1.0 / 3.0);
  const float s_sqrt3 = sqrtf(3.0f);
  const float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > 0.f) a_over_3 = 0.f;
  const float half_b = 0.5f * (c0 + c2_over_3 * (2.f * c2_over_3 * c2_over_3 - c1));
  float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > 0.f) q = 0.f;
  const float rho = sqrtf(-a_over_3);
  const float theta = lmf_atan2f(sqrtf(-q), half_b) * s_inv3;   // libm_f32.hpp: the same bits as the CPU path
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

__device__ void eigen33_smallest(const float mat[9], float *eigenvalue, float evec[3]) {
  float scale = 0.f;
#pragma unroll
  for (int i = 0; i < 9; ++i) scale = fmaxf(scale, fabsf(mat[i]));
  if (scale <= 1.17549435e-38f) scale = 1.0f;
  float sm[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) sm[i] = mat[i] / scale;
  float roots[3];
  compute_roots(sm, roots);
  *eigenvalue = roots[0] * scale;
  sm[0] -= roots[0]; sm[4] -= roots[0]; sm[8] -= roots[0];
  const float v1[3] = {sm[1] * sm[5] - sm[2] * sm[4], sm[2] * sm[3] - sm[0] * sm[5], sm[0] * sm[4] - sm[1] * sm[3]};
  const float v2[3] = {sm[1] * sm[8] - sm[2] * sm[7], sm[2] * sm[6] - sm[0] * sm[8], sm[0] * sm[7] - sm[1] * sm[6]};
  const float v3[3] = {sm[4] * sm[8] - sm[5] * sm[7], sm[5] * sm[6] - sm[3] * sm[8], sm[3] * sm[7] - sm[4] * sm[6]};
  const float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
  const float l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
  const float l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
  float vx, vy, vz, l;
  if (l1 >= l2 && l1 >= l3) { vx = v1[0]; vy = v1[1]; vz = v1[2]; l = l1; }
  else if (l2 >= l1 && l2 >= l3) { vx = v2[0]; vy = v2[1]; vz = v2[2]; l = l2; }
  else { vx = v3[0]; vy = v3[1]; vz = v3[2]; l = l3; }
  const float s = sqrtf(l);
  evec[0] = vx / s; evec[1] = vy / s; evec[2] = vz / s;
}

// NormalEstimation::computeFeature with a k-NN neighbourhood.  The k nearest (self included) come out
// of the traversal ascending by distance, exactly the order in which PCL's single-pass fp32
// computeMeanAndCovarianceMatrix accumulates them.
// KREG = 0: list in LDS (any k <= 32); KREG = 12 / 30: list in registers for the two values the reference uses
// (regmeshpcd.cpp:80, poseestimator.cpp:154), see KnnRegVisitor.
template <int KREG>
__global__ __launch_bounds__(kKnnBlock) void normals_kernel(CloudView q, BvhView tgt, int k, float vpx, float vpy,
                                                             float vpz, float4 *__restrict__ out_nrm, const uint32_t *__restrict__ self_leaf) {
  extern __shared__ unsigned char s_dyn[];
  float *ld = reinterpret_cast<float *>(s_dyn) + threadIdx.x;
  uint32_t *lp = reinterpret_cast<uint32_t *>(s_dyn + sizeof(float) * kKnnBlock * kKnnMaxK) + threadIdx.x;
  __shared__ float s_stk[kMaxDepth + 1][kKnnBlock];
  float *stk = &s_stk[0][threadIdx.x];
  const float qnan = __int_as_float(0x7fc00000);
  for (uint32_t i = blockIdx.x * kKnnBlock + threadIdx.x; i < q.n; i += gridDim.x * kKnnBlock) {
    if (i >= q.n_valid) { out_nrm[i] = make_float4(qnan, qnan, qnan, qnan); continue; }
    const float4 s = q.xyzw[i];
    const uint32_t h = self_leaf ? self_leaf[(uint32_t)__float_as_int(s.w)] : 0u;   // the query's own leaf, if the index is over the same cloud
    float accu[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int count;
#define OPE_ACCUMULATE_NEIGHBOUR(P)                                       \
  accu[0] += P.x * P.x; accu[1] += P.x * P.y; accu[2] += P.x * P.z;       \
  accu[3] += P.y * P.y; accu[4] += P.y * P.z; accu[5] += P.z * P.z;       \
  accu[6] += P.x; accu[7] += P.y; accu[8] += P.z
    if constexpr (KREG > 0) {
      KnnRegVisitor<KREG> v;
      v.init(true);
      bvh_traverse(tgt, s.x, s.y, s.z, v, stk, kKnnBlock, h);
      count = v.count;
      if (count >= 3) {
#pragma unroll
        for (int j = 0; j < KREG; ++j)
          if (j < count) { const float4 p = tgt.pts[v.p[j]]; OPE_ACCUMULATE_NEIGHBOUR(p); }
      }
    } else {
      KnnVisitor v{ld, lp, kKnnBlock, k, 0, INFINITY};
      bvh_traverse(tgt, s.x, s.y, s.z, v, stk, kKnnBlock, h);
      count = v.count;
      if (count >= 3)
        for (int j = 0; j < count; ++j) { const float4 p = tgt.pts[lp[j * kKnnBlock]]; OPE_ACCUMULATE_NEIGHBOUR(p); }
    }
#undef OPE_ACCUMULATE_NEIGHBOUR
    if (count < 3) { out_nrm[i] = make_float4(qnan, qnan, qnan, qnan); continue; }
    const float fc = (float)count;
#pragma unroll
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
    eigen33_smallest(cov, &ev, nv);
    const float eig_sum = cov[0] + cov[4] + cov[8];
    const float curv = (eig_sum != 0.f) ? fabsf(ev / eig_sum) : 0.f;
    // flipNormalTowardsViewpoint
    const float cos_theta = (vpx - s.x) * nv[0] + (vpy - s.y) * nv[1] + (vpz - s.z) * nv[2];
    if (cos_theta < 0) { nv[0] *= -1; nv[1] *= -1; nv[2] *= -1; }
    out_nrm[i] = make_float4(nv[0], nv[1], nv[2], curv);
  }
}

// ------------------------------------------------------------------------------------------ FPFH
// pcl::computePairFeatures (features/src/pfh.cpp); returns false if rejected
__device__ __forceinline__ bool pair_features(float p1x, float p1y, float p1z, float n1x, float n1y, float n1z,
                                              float p2x, float p2y, float p2z, float n2x, float n2y, float n2z,
                                              float &f1, float &f2, float &f3) {
  float dx = p2x - p1x, dy = p2y - p1y, dz = p2z - p1z;
  const float f4 = sqrtf(dx * dx + dy * dy + dz * dz);
  if (f4 == 0.0f) return false;
  float ax = n1x, ay = n1y, az = n1z, bx = n2x, by = n2y, bz = n2z;
  const float angle1 = (ax * dx + ay * dy + az * dz) / f4;
  const float angle2 = (bx * dx + by * dy + bz * dz) / f4;
  if (lmf_acosf(fabsf(angle1)) > lmf_acosf(fabsf(angle2))) {   // (libm_f32.hpp: the same bits as the C library of the CPU path)
    ax = n2x; ay = n2y; az = n2z;
    bx = n1x; by = n1y; bz = n1z;
    dx *= -1.f; dy *= -1.f; dz *= -1.f;
    f3 = -angle2;
  } else {
    f3 = angle1;
  }
  float vx = dy * az - dz * ay, vy = dz * ax - dx * az, vz = dx * ay - dy * ax;
  const float vn = sqrtf(vx * vx + vy * vy + vz * vz);
  if (vn == 0.0f) return false;
  vx /= vn; vy /= vn; vz /= vn;
  const float wx = ay * vz - az * vy, wy = az * vx - ax * vz, wz = ax * vy - ay * vx;
  f2 = vx * bx + vy * by + vz * bz;
  f1 = lmf_atan2f(wx * bx + wy * by + wz * bz, ax * bx + ay * by + az * bz);
  return true;
}

// Pass 1: SPFH of every point (FPFHEstimation::computePointSPFHSignature).  Bin counts live in LDS
// (33 counters per lane, lane-strided); the row is scaled by 100/(m-1) at the end and stored at the
// point's position in the INDEX order, so that pass 2 gathers neighbouring rows from nearby memory.
struct SpfhVisitor {
  const BvhView *t;
  float r2;
  float px, py, pz, nx, ny, nz;
  int self_idx;
  int m;
  uint32_t self_pos;
  uint32_t *hist;  // &lds[threadIdx.x], bin b at hist[b * kFeatBlock]
  __device__ __forceinline__ bool prune(float bound) const { return bound > r2; }
  __device__ __forceinline__ void point(float d, const v4f &p, uint32_t i, uint32_t) {
    if (!(d <= r2)) return;
    ++m;
    if (__float_as_int(p.w) == self_idx) { self_pos = i; return; }
    const float4 nj = t->nrm[i];
    float f1, f2, f3;
    if (!pair_features(px, py, pz, nx, ny, nz, p.x, p.y, p.z, nj.x, nj.y, nj.z, f1, f2, f3)) return;
    const double d_pi = (double)(1.0f / (2.0f * 3.14159274f));
    int h = (int)floor(11 * (((double)f1 + 3.14159265358979323846) * d_pi));
    h = min(max(h, 0), 10);
    hist[h * kFeatBlock] += 1u;
    h = (int)floor(11 * (((double)f2 + 1.0) * 0.5));
    h = min(max(h, 0), 10);
    hist[(11 + h) * kFeatBlock] += 1u;
    h = (int)floor(11 * (((double)f3 + 1.0) * 0.5));
    h = min(max(h, 0), 10);
    hist[(22 + h) * kFeatBlock] += 1u;
  }
  __device__ __forceinline__ void on_node() {}
};

__global__ __launch_bounds__(kFeatBlock) void spfh_kernel(CloudView q, BvhView tgt, float r2, float *__restrict__ spfh,
                                                           uint32_t *__restrict__ self_pos_out,
                                                           unsigned long long *__restrict__ neighbour_total) {
  __shared__ float s_stk[kMaxDepth + 1][kFeatBlock];
  __shared__ uint32_t s_hist[33][kFeatBlock];
  float *stk = &s_stk[0][threadIdx.x];
  uint32_t *hist = &s_hist[0][threadIdx.x];
  unsigned long long local_m = 0;
  for (uint32_t i = blockIdx.x * kFeatBlock + threadIdx.x; i < q.n_valid; i += gridDim.x * kFeatBlock) {
#pragma unroll
    for (int b = 0; b < 33; ++b) hist[b * kFeatBlock] = 0u;
    const float4 s = q.xyzw[i];
    const float4 n = q.nrm[i];
    SpfhVisitor v{&tgt, r2, s.x, s.y, s.z, n.x, n.y, n.z, __float_as_int(s.w), 0, kNoPos, hist};
    bvh_traverse(tgt, s.x, s.y, s.z, v, stk, kFeatBlock);
    self_pos_out[i] = v.self_pos;
    local_m += (unsigned long long)v.m;
    if (v.self_pos != kNoPos) {
      const float hist_incr = 100.0f / (float)(v.m - 1);
      float *row = spfh + (size_t)v.self_pos * kSpfhStride;
#pragma unroll
      for (int b = 0; b < 33; ++b) {
        // PCL adds hist_incr to the bin once per pair, in float (fpfh.hpp computePointSPFHSignature): c additions of the same
        // value, whatever their order — not c * hist_incr rounded once
        const uint32_t c = hist[b * kFeatBlock];
        float h = 0.f;
        for (uint32_t a = 0; a < c; ++a) h = __fadd_rn(h, hist_incr);
        row[b] = h;
      }
      row[33] = row[34] = row[35] = 0.f;
    }
  }
  atomicAdd(neighbour_total, local_m);
}

// Pass 2: FPFHEstimation::weightPointSPFHSignature.  Self (d2 == 0) is excluded and the weight is
// 1/d2 with d2 the SQUARED distance (PCL quirk Q6); each 11-bin group is rescaled to sum to 100.
struct FpfhVisitor {
  const float *spfh;
  float r2;
  float acc[33];
  double sum0, sum1, sum2;
  int m;
  __device__ __forceinline__ bool prune(float bound) const { return bound > r2; }
  __device__ __forceinline__ void point(float d, const v4f &, uint32_t i, uint32_t) {
    if (!(d <= r2)) return;
    ++m;
    if (d == 0.f) return;
    const float w = 1.0f / d;
    const v4f *row = reinterpret_cast<const v4f *>(spfh + (size_t)i * kSpfhStride);
    float h[36];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const v4f r = row[c];
      h[4 * c] = r.x; h[4 * c + 1] = r.y; h[4 * c + 2] = r.z; h[4 * c + 3] = r.w;
    }
#pragma unroll
    for (int b = 0; b < 11; ++b) { const float val = h[b] * w; sum0 += (double)val; acc[b] += val; }
#pragma unroll
    for (int b = 11; b < 22; ++b) { const float val = h[b] * w; sum1 += (double)val; acc[b] += val; }
#pragma unroll
    for (int b = 22; b < 33; ++b) { const float val = h[b] * w; sum2 += (double)val; acc[b] += val; }
  }
  __device__ __forceinline__ void on_node() {}
};

__global__ __launch_bounds__(kFeatBlock) void fpfh_kernel(CloudView q, BvhView tgt, float r2, const float *__restrict__ spfh,
                                                           float *__restrict__ out33) {
  __shared__ float s_stk[kMaxDepth + 1][kFeatBlock];
  float *stk = &s_stk[0][threadIdx.x];
  const float qnan = __int_as_float(0x7fc00000);
  for (uint32_t i = blockIdx.x * kFeatBlock + threadIdx.x; i < q.n; i += gridDim.x * kFeatBlock) {
    const float4 s = q.xyzw[i];
    float *o = out33 + (size_t)__float_as_int(s.w) * 33;
    if (i >= q.n_valid) {
      for (int b = 0; b < 33; ++b) o[b] = qnan;
      continue;
    }
    FpfhVisitor v;
    v.spfh = spfh; v.r2 = r2; v.sum0 = v.sum1 = v.sum2 = 0.0; v.m = 0;
#pragma unroll
    for (int b = 0; b < 33; ++b) v.acc[b] = 0.f;
    bvh_traverse(tgt, s.x, s.y, s.z, v, stk, kFeatBlock);
    if (v.m == 0) {
      for (int b = 0; b < 33; ++b) o[b] = qnan;
      continue;
    }
    const double k0 = v.sum0 != 0 ? 100.0 / v.sum0 : 0.0, k1 = v.sum1 != 0 ? 100.0 / v.sum1 : 0.0,
                 k2 = v.sum2 != 0 ? 100.0 / v.sum2 : 0.0;
#pragma unroll
    for (int b = 0; b < 11; ++b) o[b] = v.acc[b] * (float)k0;
#pragma unroll
    for (int b = 11; b < 22; ++b) o[b] = v.acc[b] * (float)k1;
#pragma unroll
    for (int b = 22; b < 33; ++b) o[b] = v.acc[b] * (float)k2;
  }
}

// ------------------------------------------------------------------------------------------ SAC-IA
// findSimilarFeatures: k nearest target descriptors (33-D, squared L2, fp32 sequential sum) of each
// query descriptor.  One block per query; every thread keeps its k best, thread 0 merges by (d, index).
constexpr int kFeatK = 8;
__global__ __launch_bounds__(256) void feature_knn_kernel(const float *__restrict__ tgt_feat, int nt,
                                                           const float *__restrict__ q_feat, int k,
                                                           int32_t *__restrict__ out_idx) {
  __shared__ float s_d[256][kFeatK];
  __shared__ int s_i[256][kFeatK];
  __shared__ float s_q[33];
  if (threadIdx.x < 33) s_q[threadIdx.x] = q_feat[(size_t)blockIdx.x * 33 + threadIdx.x];
  __syncthreads();
  float bd[kFeatK];
  int bi[kFeatK];
#pragma unroll
  for (int j = 0; j < kFeatK; ++j) { bd[j] = INFINITY; bi[j] = -1; }
  for (int t = threadIdx.x; t < nt; t += 256) {
    const float *f = tgt_feat + (size_t)t * 33;
    float d = 0.f;
    for (int c = 0; c < 33; ++c) { const float u = s_q[c] - f[c]; d += u * u; }
    if (!(d == d)) continue;  // NaN descriptors never match
    if (d < bd[kFeatK - 1]) {
      bd[kFeatK - 1] = d; bi[kFeatK - 1] = t;
#pragma unroll
      for (int j = kFeatK - 1; j > 0; --j)
        if (bd[j - 1] > bd[j]) {
          const float td = bd[j]; bd[j] = bd[j - 1]; bd[j - 1] = td;
          const int ti = bi[j]; bi[j] = bi[j - 1]; bi[j - 1] = ti;
        }
    }
  }
#pragma unroll
  for (int j = 0; j < kFeatK; ++j) { s_d[threadIdx.x][j] = bd[j]; s_i[threadIdx.x][j] = bi[j]; }
  __syncthreads();
  __shared__ unsigned char cur[256];
  cur[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    // k-way selection by (distance, index): k <= 8 passes over the 256 list heads
    for (int r = 0; r < k; ++r) {
      float best = INFINITY;
      int best_i = -1, best_t = -1;
      for (int t = 0; t < 256; ++t) {
        const int c = cur[t];
        if (c >= kFeatK) continue;
        const float d = s_d[t][c];
        const int id = s_i[t][c];
        if (id < 0) continue;
        if (d < best || (d == best && id < best_i)) { best = d; best_i = id; best_t = t; }
      }
      out_idx[(size_t)blockIdx.x * k + r] = best_i;
      if (best_t >= 0) cur[best_t]++;
    }
  }
}

// computeErrorMetric for many hypotheses at once: grid.y = hypothesis, each lane one source point.
// TruncatedError(e) = e <= thr ? e / thr : 1 on the SQUARED 1-NN distance.
__global__ __launch_bounds__(256) void sacia_error_kernel(CloudView src, BvhView tgt, const float *__restrict__ T_rows,
                                                           float thr, double *__restrict__ partials) {
  __shared__ float s_stk[kMaxDepth + 1][256];
  __shared__ double s_red[4];
  float *stk = &s_stk[0][threadIdx.x];
  const float *F = T_rows + 12 * (size_t)blockIdx.y;
  float err = 0.f;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < src.n_valid; i += gridDim.x * 256) {
    const float4 s = src.xyzw[i];
    const float x = xform_row(F + 0, s.x, s.y, s.z);
    const float y = xform_row(F + 4, s.x, s.y, s.z);
    const float z = xform_row(F + 8, s.x, s.y, s.z);
    NearestVisitor v{INFINITY, kNoPos, 0};
    bvh_traverse(tgt, x, y, z, v, stk, 256);
    err += (v.pos != kNoPos && v.best <= thr) ? v.best / thr : 1.0f;
  }
  const double w = wave_sum((double)err);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// ------------------------------------------------------------------------------------------ host side
static int self_index(ope_ctx *ctx, const ope_cloud *cloud, ope_index **out) {
  ope_index_params p;
  ope_index_default_params(&p);
  return index_build_tmp(ctx, cloud, &p, out);   // (freed by the entry point that asked for it: a temporary)
}

static void colmajor_to_rows12(const float *T, float rows[12]) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) rows[4 * r + c] = T[4 * c + r];
}

// Eigen::umeyama on a handful of pairs, fp64 on the host (hypothesis generation is O(400 * 5))
static void jacobi3_host(double S[9], double V[9]) {
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
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
        for (int k = 0; k < 3; ++k) { const double a = S[3 * k + p], b = S[3 * k + q]; S[3 * k + p] = c * a - s * b; S[3 * k + q] = s * a + c * b; }
        for (int k = 0; k < 3; ++k) { const double a = S[3 * p + k], b = S[3 * q + k]; S[3 * p + k] = c * a - s * b; S[3 * q + k] = s * a + c * b; }
        for (int k = 0; k < 3; ++k) { const double a = V[3 * k + p], b = V[3 * k + q]; V[3 * k + p] = c * a - s * b; V[3 * k + q] = s * a + c * b; }
      }
  }
}

static double det3h(const double M[9]) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

static void umeyama_host(const float *src, const float *dst, int n, float T[16]) {
  double sm[3] = {0, 0, 0}, dm[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) { sm[d] += src[3 * i + d]; dm[d] += dst[3 * i + d]; }
  for (int d = 0; d < 3; ++d) { sm[d] /= n; dm[d] /= n; }
  double sigma[9] = {0};
  for (int i = 0; i < n; ++i)
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) sigma[3 * r + c] += (dst[3 * i + r] - dm[r]) * (src[3 * i + c] - sm[c]);
  for (int k = 0; k < 9; ++k) sigma[k] /= n;
  // SVD via eigen-decomposition of sigma^T sigma
  double AtA[9], Vt[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) AtA[3 * i + j] = sigma[i] * sigma[j] + sigma[3 + i] * sigma[3 + j] + sigma[6 + i] * sigma[6 + j];
  jacobi3_host(AtA, Vt);
  int ord[3] = {0, 1, 2};
  const double ev[3] = {AtA[0], AtA[4], AtA[8]};
  std::sort(ord, ord + 3, [&](int a, int b) { return ev[a] > ev[b]; });
  double V[9], U[9], sv[3], Uc[3][3];
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) V[3 * r + c] = Vt[3 * r + ord[c]];
  for (int c = 0; c < 3; ++c) {
    for (int r = 0; r < 3; ++r) Uc[c][r] = sigma[3 * r] * V[c] + sigma[3 * r + 1] * V[3 + c] + sigma[3 * r + 2] * V[6 + c];
    sv[c] = std::sqrt(Uc[c][0] * Uc[c][0] + Uc[c][1] * Uc[c][1] + Uc[c][2] * Uc[c][2]);
  }
  const double tiny = 1e-14 * (sv[0] > 0 ? sv[0] : 1.0);
  for (int c = 0; c < 3; ++c) {
    double *u = Uc[c];
    for (int p = 0; p < c; ++p) {
      const double d = u[0] * Uc[p][0] + u[1] * Uc[p][1] + u[2] * Uc[p][2];
      for (int r = 0; r < 3; ++r) u[r] -= d * Uc[p][r];
    }
    double nrm = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    if (sv[c] <= tiny || nrm <= 1e-8 * sv[c] + 1e-300) {
      if (c == 0) { u[0] = 1; u[1] = 0; u[2] = 0; }
      else if (c == 1) {
        const double *a = Uc[0];
        const int m = std::fabs(a[0]) < std::fabs(a[1]) ? (std::fabs(a[0]) < std::fabs(a[2]) ? 0 : 2) : (std::fabs(a[1]) < std::fabs(a[2]) ? 1 : 2);
        const double d = a[m];
        for (int r = 0; r < 3; ++r) u[r] = (r == m ? 1.0 : 0.0) - d * a[r];
      } else {
        const double *a = Uc[0], *b = Uc[1];
        u[0] = a[1] * b[2] - a[2] * b[1]; u[1] = a[2] * b[0] - a[0] * b[2]; u[2] = a[0] * b[1] - a[1] * b[0];
      }
      nrm = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    }
    for (int r = 0; r < 3; ++r) u[r] /= nrm;
  }
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) U[3 * r + c] = Uc[c][r];
  double Sg[3] = {1, 1, 1};
  if (det3h(sigma) < 0) Sg[2] = -1;
  int rank = 0;
  for (int i = 0; i < 3; ++i)
    if (!(std::fabs(sv[i]) <= std::fabs(sv[0]) * 1e-5)) ++rank;
  if (rank == 2) { Sg[0] = Sg[1] = 1; Sg[2] = (det3h(U) * det3h(V) > 0) ? 1 : -1; }
  double R[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += U[3 * i + k] * Sg[k] * V[3 * j + k];
      R[3 * i + j] = a;
    }
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) T[4 * c + r] = (float)R[3 * r + c];
  T[3] = T[7] = T[11] = 0.f;
  for (int i = 0; i < 3; ++i) T[12 + i] = (float)(dm[i] - (R[3 * i] * sm[0] + R[3 * i + 1] * sm[1] + R[3 * i + 2] * sm[2]));
  T[15] = 1.f;
}

}  // namespace ope

using namespace ope;

extern "C" {

int ope_radius_search(ope_ctx *ctx, const ope_cloud *queries, const ope_index *index, float radius, int max_nn,
                      int32_t *counts, int32_t *out_idx, float *out_d2) {
  if (!ctx || !queries || !index || !counts || max_nn < 0 || max_nn > kKnnMaxK || (max_nn > 0 && (!out_idx || !out_d2)))
    return set_err(ctx, OPE_EINVAL, "ope_radius_search: bad argument (0 <= max_nn <= 32)");
  { const int rch = queries->ensure_host(); if (rch != OPE_OK) return rch; }
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n = queries->n;
  if (n == 0) return OPE_OK;
  int32_t *d_cnt = nullptr, *d_idx = nullptr;
  float *d_d2 = nullptr;
  const size_t m = std::max<size_t>((size_t)max_nn * n, 1);
  hipError_t e = tmp_malloc(ctx->stream, (void **)&d_cnt, sizeof(int32_t) * n);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_idx, sizeof(int32_t) * m);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_d2, sizeof(float) * m);
  std::vector<int32_t> hc(n), hi(m);
  std::vector<float> hd(m);
  if (e == hipSuccess) {
    const int nblocks = (int)std::min<size_t>((n + kKnnBlock - 1) / kKnnBlock, 4096);
    hipLaunchKernelGGL(radius_search_kernel, dim3(nblocks), dim3(kKnnBlock), kKnnLdsBytes, ctx->stream, queries->view(),
                       index->view(), radius * radius, max_nn, d_cnt, d_idx, d_d2);
    e = hipMemcpyAsync(hc.data(), d_cnt, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream);
  }
  if (e == hipSuccess && max_nn) e = hipMemcpyAsync(hi.data(), d_idx, sizeof(int32_t) * m, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && max_nn) e = hipMemcpyAsync(hd.data(), d_d2, sizeof(float) * m, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  tmp_free(ctx->stream, d_cnt);
  tmp_free(ctx->stream, d_idx);
  tmp_free(ctx->stream, d_d2);
  if (e != hipSuccess) return set_err(ctx, OPE_EHIP, std::string("ope_radius_search: ") + hipGetErrorString(e));
  for (size_t i = 0; i < n; ++i) {
    const size_t o = (size_t)queries->perm[i];
    counts[o] = hc[i];
    if (max_nn) {
      std::memcpy(out_idx + o * max_nn, hi.data() + i * max_nn, sizeof(int32_t) * max_nn);
      std::memcpy(out_d2 + o * max_nn, hd.data() + i * max_nn, sizeof(float) * max_nn);
    }
  }
  return OPE_OK;
}

// normals of `cloud`'s points from their k nearest neighbours in `index` (null: an index over the cloud itself)
static int normals_impl(ope_ctx *ctx, ope_cloud *cloud, const ope_index *index, int k, const float vp[3], float *out_normals,
                        float *out_curvature, const char *who) {
  if (!ctx || !cloud || k < 1 || k > kKnnMaxK) return set_err(ctx, OPE_EINVAL, std::string(who) + ": bad argument (1 <= k <= 32)");
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n = cloud->n;
  if (n == 0) return OPE_OK;
  static const float origin[3] = {0.f, 0.f, 0.f};
  const float *v = vp ? vp : origin;
  if (!cloud->d_nrm) OPE_HIP(ctx, hipMalloc((void **)&cloud->d_nrm, sizeof(float4) * n));
  const bool want_host = out_normals || out_curvature;   // both null: the normals only stay attached to the cloud
  std::vector<float> packed(want_host ? n * 4 : 0);
  if (cloud->n_valid > 0) {
    ope_index *own = nullptr;
    if (!index) {
      int rc = self_index(ctx, cloud, &own);
      if (rc != OPE_OK) return rc;
      index = own;
    }
    const int nblocks = (int)std::min<size_t>((n + kKnnBlock - 1) / kKnnBlock, 4096);
    // own index: every query starts its walk at the leaf that holds it (self_leaves above).  A k-NN walk from the root has no
    // bound until its list is full and opens far more of the tree than it needs: 1 M-point frame, k = 12 and 30 together,
    // 16.6 -> 3.2 ms; results bit-equal (a start leaf never changes what a walk finds)
    uint32_t *d_self_leaf = nullptr;
    if (own) {
      OPE_HIP(ctx, tmp_malloc(ctx->stream, (void **)&d_self_leaf, 4 * n));
      OPE_HIP(ctx, self_leaves(ctx->stream, index->view(), n, d_self_leaf));
    }
    struct FreeSelf { hipStream_t s; uint32_t *p; ~FreeSelf() { if (p) tmp_free(s, p); } } free_self{ctx->stream, d_self_leaf};
    TraceRange r_n(ctx, "normals");
    // B_nrm = N (12 + 12 k + 16)
    KernelTimer kt(ctx, "normals_kernel", (double)cloud->n_valid * (12.0 + 12.0 * k + 16.0));
    if (k == 12)
      hipLaunchKernelGGL(normals_kernel<12>, dim3(nblocks), dim3(kKnnBlock), 0, ctx->stream, cloud->view(), index->view(), k, v[0], v[1],
                         v[2], cloud->d_nrm, d_self_leaf);
    else if (k == 30)
      hipLaunchKernelGGL(normals_kernel<30>, dim3(nblocks), dim3(kKnnBlock), 0, ctx->stream, cloud->view(), index->view(), k, v[0], v[1],
                         v[2], cloud->d_nrm, d_self_leaf);
    else
      hipLaunchKernelGGL(normals_kernel<0>, dim3(nblocks), dim3(kKnnBlock), kKnnLdsBytes, ctx->stream, cloud->view(), index->view(), k,
                         v[0], v[1], v[2], cloud->d_nrm, d_self_leaf);
    kt.stop();
    hipError_t e = hipSuccess;
    if (want_host) e = hipMemcpyAsync(packed.data(), cloud->d_nrm, sizeof(float4) * n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (own) ope_index_free(own);
    if (e != hipSuccess) return set_err(ctx, OPE_EHIP, std::string(who) + ": " + hipGetErrorString(e));
  } else {
    const float qn = std::numeric_limits<float>::quiet_NaN();
    std::vector<float> nan4(n * 4, qn);
    OPE_HIP(ctx, h2d_copy(ctx->stream, cloud->d_nrm, nan4.data(), sizeof(float4) * n));
    if (want_host) packed = nan4;
  }
  OPE_DUMP_HASH("normals d_nrm", cloud->d_nrm, 16 * n, true);
  if (!want_host) return OPE_OK;
  { const int rch = cloud->ensure_host(); if (rch != OPE_OK) return rch; }
  for (size_t i = 0; i < n; ++i) {
    const size_t o = (size_t)cloud->perm[i];
    if (out_normals) { out_normals[3 * o] = packed[4 * i]; out_normals[3 * o + 1] = packed[4 * i + 1]; out_normals[3 * o + 2] = packed[4 * i + 2]; }
    if (out_curvature) out_curvature[o] = packed[4 * i + 3];
  }
  return OPE_OK;
}

int ope_normals(ope_ctx *ctx, ope_cloud *cloud, int k, const float vp[3], float *out_normals, float *out_curvature) {
  return normals_impl(ctx, cloud, nullptr, k, vp, out_normals, out_curvature, "ope_normals");
}

int ope_normals_from(ope_ctx *ctx, ope_cloud *queries, const ope_index *index, int k, const float vp[3], float *out_normals,
                     float *out_curvature) {
  if (!index) return set_err(ctx, OPE_EINVAL, "ope_normals_from: bad argument");
  return normals_impl(ctx, queries, index, k, vp, out_normals, out_curvature, "ope_normals_from");
}

int ope_fpfh(ope_ctx *ctx, const ope_cloud *cloud, float radius, float *out33) {
  if (!ctx || !cloud || !out33 || !(radius > 0)) return set_err(ctx, OPE_EINVAL, "ope_fpfh: bad argument");
  if (!cloud->d_nrm) return set_err(ctx, OPE_EINVAL, "ope_fpfh: the cloud carries no normals (call ope_normals / ope_cloud_set_normals)");
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n = cloud->n;
  if (n == 0) return OPE_OK;
  if (cloud->n_valid == 0) {
    std::fill(out33, out33 + n * 33, std::numeric_limits<float>::quiet_NaN());
    return OPE_OK;
  }
  ope_index *ix = nullptr;
  int rc = self_index(ctx, cloud, &ix);
  if (rc != OPE_OK) return rc;
  float *d_spfh = nullptr, *d_out = nullptr;
  uint32_t *d_self = nullptr;
  unsigned long long *d_total = nullptr;
  hipError_t e = tmp_malloc(ctx->stream, (void **)&d_spfh, sizeof(float) * kSpfhStride * ix->n);
  if (e == hipSuccess) e = hipMemsetAsync(d_spfh, 0, sizeof(float) * kSpfhStride * ix->n, ctx->stream);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_out, sizeof(float) * 33 * n);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_self, sizeof(uint32_t) * n);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_total, sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemsetAsync(d_total, 0, sizeof(unsigned long long), ctx->stream);
  if (e == hipSuccess) {
    const int nblocks = (int)std::min<size_t>((n + kFeatBlock - 1) / kFeatBlock, 4096);
    const float r2 = radius * radius;
    KernelTimer kt_s(ctx, "spfh_kernel", 0.0), kt_w(ctx, "fpfh_kernel", 0.0, /*start_now=*/false);
    {
      TraceRange r_s(ctx, "fpfh_spfh");
      hipLaunchKernelGGL(spfh_kernel, dim3(nblocks), dim3(kFeatBlock), 0, ctx->stream, cloud->view(), ix->view(), r2, d_spfh,
                         d_self, d_total);
    }
    kt_s.stop();
    kt_w.start();
    {
      TraceRange r_w(ctx, "fpfh_weight");
      hipLaunchKernelGGL(fpfh_kernel, dim3(nblocks), dim3(kFeatBlock), 0, ctx->stream, cloud->view(), ix->view(), r2, d_spfh,
                         d_out);
    }
    kt_w.stop();
    unsigned long long total_nb = 0;
    e = hipMemcpyAsync(out33, d_out, sizeof(float) * 33 * n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&total_nb, d_total, sizeof total_nb, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    {
      // pass 1 N (24 + 24 m + 132), pass 2 N (136 m + 132), m = measured mean neighbours in the radius
      const double N = (double)cloud->n_valid, m = N > 0 ? (double)total_nb / N : 0.0;
      kt_s.set_bytes(N * (24.0 + 24.0 * m + 132.0));
      kt_w.set_bytes(N * (136.0 * m + 132.0));
      ctx->last_fpfh_mean_neighbours = m;
    }
  }
  tmp_free(ctx->stream, d_spfh);
  tmp_free(ctx->stream, d_out);
  tmp_free(ctx->stream, d_self);
  tmp_free(ctx->stream, d_total);
  ope_index_free(ix);
  if (e != hipSuccess) return set_err(ctx, OPE_EHIP, std::string("ope_fpfh: ") + hipGetErrorString(e));
  return OPE_OK;
}

void ope_sacia_default_params(ope_sacia_params *p) {
  if (!p) return;
  p->max_iterations = 400;
  p->nr_samples = 5;
  p->k_correspondences = 5;
  p->max_corr_dist = 0.05;
  p->min_sample_dist = 0.01f;
  p->seed = 1;
}

int ope_sacia(ope_ctx *ctx, const ope_cloud *src, const float *src_feat33, const ope_cloud *tgt, const ope_index *tgt_index,
              const float *tgt_feat33, const ope_sacia_params *params, const int32_t *forced_samples, float out_T[16],
              double *best_error, int32_t *best_iteration) {
  static const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  if (out_T) std::memcpy(out_T, I4, sizeof I4);
  if (best_error) *best_error = 0;
  if (best_iteration) *best_iteration = -1;
  if (!ctx || !src || !tgt || !tgt_index || !out_T || (!forced_samples && (!src_feat33 || !tgt_feat33)))
    return set_err(ctx, OPE_EINVAL, "ope_sacia: bad argument");
  ope_sacia_params p;
  ope_sacia_default_params(&p);
  if (params) p = *params;
  const int ns = (int)src->n, nt = (int)tgt->n, S = p.nr_samples, K = p.k_correspondences, H = p.max_iterations;
  if (S < 1 || ns < S || nt < 1 || H < 1 || K < 1 || K > kFeatK)
    return set_err(ctx, OPE_EINVAL, "ope_sacia: need nr_samples <= |source|, 1 <= k_correspondences <= 8");
  { const int rch = src->ensure_host(); if (rch != OPE_OK) return rch; }
  { const int rch = tgt->ensure_host(); if (rch != OPE_OK) return rch; }
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  TraceRange r_sac(ctx, "sacia");

  // ---- selectSamples / findSimilarFeatures draws (host; the RNG stream does not depend on device results)
  std::vector<int32_t> samp((size_t)H * S), pick((size_t)H * S), corr((size_t)H * S);
  if (forced_samples) {
    std::memcpy(samp.data(), forced_samples, sizeof(int32_t) * samp.size());
    std::memcpy(corr.data(), forced_samples + samp.size(), sizeof(int32_t) * corr.size());
  } else {
    uint64_t rng = p.seed;
    auto next = [&rng]() {
      rng = rng * 6364136223846793005ULL + 1442695040888963407ULL;
      return (double)(rng >> 11) * (1.0 / 9007199254740992.0);
    };
    const float *xyz = src->h_xyz.data();
    for (int it = 0; it < H; ++it) {
      // selectSamples takes min_sample_distance BY VALUE (ia_ransac.hpp): a halving after 3*ns failed draws lasts for
      // this hypothesis only, the next one starts from the configured distance again
      float msd = p.min_sample_dist;
      int cnt = 0, without = 0;
      const int max_without = 3 * ns;
      int32_t *sm = &samp[(size_t)it * S];
      while (cnt < S) {
        const int si = (int)(ns * next());
        bool valid = true;
        for (int i = 0; i < cnt; ++i) {
          const float dx = xyz[3 * si] - xyz[3 * sm[i]], dy = xyz[3 * si + 1] - xyz[3 * sm[i] + 1], dz = xyz[3 * si + 2] - xyz[3 * sm[i] + 2];
          const float dist = std::sqrt(dx * dx + dy * dy + dz * dz);
          if (si == sm[i] || dist < msd) { valid = false; break; }
        }
        if (valid) { sm[cnt++] = si; without = 0; }
        else ++without;
        if (without >= max_without) { msd *= 0.5f; without = 0; }
      }
      for (int i = 0; i < S; ++i) pick[(size_t)it * S + i] = (int)(K * next());
    }
    // ---- k nearest target descriptors of every distinct sampled source descriptor (device)
    std::map<int32_t, int> slot;
    std::vector<int32_t> uniq;
    for (int32_t s : samp)
      if (slot.emplace(s, (int)uniq.size()).second) uniq.push_back(s);
    std::vector<float> qf(uniq.size() * 33);
    for (size_t u = 0; u < uniq.size(); ++u) std::memcpy(&qf[u * 33], src_feat33 + (size_t)uniq[u] * 33, 33 * sizeof(float));
    float *d_tf = nullptr, *d_qf = nullptr;
    int32_t *d_nn = nullptr;
    std::vector<int32_t> nn(uniq.size() * K);
    hipError_t e = tmp_malloc(ctx->stream, (void **)&d_tf, sizeof(float) * 33 * (size_t)nt);
    if (e == hipSuccess) e = h2d_copy(ctx->stream, d_tf, tgt_feat33, sizeof(float) * 33 * (size_t)nt);
    if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_qf, sizeof(float) * qf.size());
    if (e == hipSuccess) e = h2d_copy(ctx->stream, d_qf, qf.data(), sizeof(float) * qf.size());
    if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_nn, sizeof(int32_t) * nn.size());
    if (e == hipSuccess) {
      {
        // every query descriptor is compared with every target descriptor: 132 B of each, the target set once per query
        KernelTimer kt(ctx, "feature_knn_kernel", 132.0 * ((double)uniq.size() * (double)nt + (double)uniq.size()));
        hipLaunchKernelGGL(feature_knn_kernel, dim3((unsigned)uniq.size()), dim3(256), 0, ctx->stream, d_tf, nt, d_qf, K, d_nn);
      }
      e = hipMemcpyAsync(nn.data(), d_nn, sizeof(int32_t) * nn.size(), hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    tmp_free(ctx->stream, d_tf);
    tmp_free(ctx->stream, d_qf);
    tmp_free(ctx->stream, d_nn);
    if (e != hipSuccess) return set_err(ctx, OPE_EHIP, std::string("ope_sacia(feature knn): ") + hipGetErrorString(e));
    for (size_t j = 0; j < samp.size(); ++j) {
      const int32_t *row = &nn[(size_t)slot[samp[j]] * K];
      const int32_t c = row[pick[j]];
      corr[j] = c >= 0 ? c : row[0];
    }
  }

  // ---- one rigid transform per hypothesis from its nr_samples pairs (TransformationEstimationSVD)
  std::vector<float> T((size_t)H * 16), rows((size_t)H * 12), ps((size_t)S * 3), pt((size_t)S * 3);
  for (int it = 0; it < H; ++it) {
    for (int i = 0; i < S; ++i) {
      const int32_t a = samp[(size_t)it * S + i], b = corr[(size_t)it * S + i];
      if (a < 0 || a >= ns || b < 0 || b >= nt) return set_err(ctx, OPE_EINVAL, "ope_sacia: sample index out of range");
      std::memcpy(&ps[3 * i], &src->h_xyz[3 * (size_t)a], 12);
      std::memcpy(&pt[3 * i], &tgt->h_xyz[3 * (size_t)b], 12);
    }
    umeyama_host(ps.data(), pt.data(), S, &T[(size_t)it * 16]);
    colmajor_to_rows12(&T[(size_t)it * 16], &rows[(size_t)it * 12]);
  }

  // ---- error metric of all hypotheses (device), then the reference's "lowest error wins" scan
  const int bx = (int)std::min<size_t>(std::max<size_t>((src->n_valid + 255) / 256, 1), 1024);
  float *d_rows = nullptr;
  double *d_part = nullptr;
  std::vector<double> part((size_t)H * bx);
  hipError_t e = tmp_malloc(ctx->stream, (void **)&d_rows, sizeof(float) * rows.size());
  if (e == hipSuccess) e = h2d_copy(ctx->stream, d_rows, rows.data(), sizeof(float) * rows.size());
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_part, sizeof(double) * part.size());
  if (e == hipSuccess) {
    {
      // 24 N_s per hypothesis (read the source point, gather its nearest target point)
      KernelTimer kt(ctx, "sacia_error_kernel", 24.0 * (double)src->n_valid * (double)H);
      hipLaunchKernelGGL(sacia_error_kernel, dim3(bx, H), dim3(256), 0, ctx->stream, src->view(), tgt_index->view(), d_rows,
                         (float)p.max_corr_dist, d_part);
    }
    e = hipMemcpyAsync(part.data(), d_part, sizeof(double) * part.size(), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  tmp_free(ctx->stream, d_rows);
  tmp_free(ctx->stream, d_part);
  if (e != hipSuccess) return set_err(ctx, OPE_EHIP, std::string("ope_sacia(error metric): ") + hipGetErrorString(e));
  // non-finite source points score 1.0 each, as a failed search does in computeErrorMetric
  const double invalid = (double)(src->n - src->n_valid);
  double lowest = 0;
  int best = -1;
  for (int it = 0; it < H; ++it) {
    double err = invalid;
    for (int b = 0; b < bx; ++b) err += part[(size_t)it * bx + b];
    if (it == 0 || (float)err < (float)lowest) { lowest = err; best = it; }
  }
  std::memcpy(out_T, &T[(size_t)best * 16], sizeof(float) * 16);
  if (best_error) *best_error = lowest;
  if (best_iteration) *best_iteration = best;
  return OPE_OK;
}

}  // extern "C"
