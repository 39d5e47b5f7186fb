// filters.hip — the point-cloud filters either side of the registration path, on the device
// (SURVEY.md §8f row 3).  HBM-bound byte work: one pass over the cloud, a rocPRIM sort/scan/select.
//
//   pcl::removeNaNFromPointCloud   DetectAndLocalize/src/poseestimator.cpp:192-194
//   pcl::PassThrough::filter       BuildModel/src/processingpcd.cpp:8-36 (z, then y, then x: an axis-aligned box)
//   pcl::VoxelGrid::filter         BuildModel/src/processingpcd.cpp:39-52
//   pcl::StatisticalOutlierRemoval DetectAndLocalize/src/processingpcd.cpp:62-77 (getOutlierRemove, meanK 30)
//
// Index filters return ORIGINAL indices in ascending order (PCL keeps the input order).  The cloud on the
// device is Morton-sorted, so the keep flag of each point is scattered to its original position and a
// rocPRIM select over a counting iterator gathers the survivors.
// VoxelGrid: key = (voxel index << 32 | original index) -> radix sort -> one lane per voxel sums its points
// in ascending original index (PCL's std::sort leaves that order unspecified) and multiplies by 1/count
// (Eigen 3.2's operator/= on float vectors) -> centroids in ascending voxel index, PCL's own output order.
#include <cstring>
#include <string>

#include <rocprim/rocprim.hpp>

#include <cfloat>
#include <cmath>
#include <vector>

#include "bvh_traverse.hpp"

namespace ope {

hipError_t self_leaves(hipStream_t, const BvhView &, size_t, uint32_t *);   // features.hip

__global__ __launch_bounds__(256) void box_flags_kernel(CloudView c, float lox, float loy, float loz, float hix, float hiy,
                                                         float hiz, unsigned char *__restrict__ flags_orig) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= c.n) return;
  const float4 p = c.xyzw[i];
  // passthrough.hpp: "if (value > max || value < min) -> removed"; non-finite points (sorted last) never pass
  const bool keep = i < c.n_valid && !(p.x > hix || p.x < lox) && !(p.y > hiy || p.y < loy) && !(p.z > hiz || p.z < loz);
  flags_orig[(uint32_t)__float_as_int(p.w)] = keep ? 1 : 0;
}

__global__ __launch_bounds__(256) void grid_key_kernel(CloudView c, float invx, float invy, float invz, int min_bx, int min_by,
                                                        int min_bz, unsigned div_x, unsigned long long div_xy,
                                                        unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= c.n) return;
  vals[i] = i;
  if (i >= c.n_valid) { keys[i] = ~0ull; return; }
  const float4 p = c.xyzw[i];
  // voxel_grid.hpp: ijk = static_cast<int>(floor(x * inverse_leaf) - static_cast<float>(min_b))
  const int ix = (int)(floorf(p.x * invx) - (float)min_bx), iy = (int)(floorf(p.y * invy) - (float)min_by),
            iz = (int)(floorf(p.z * invz) - (float)min_bz);
  const unsigned long long voxel = (unsigned long long)ix + (unsigned long long)iy * div_x + (unsigned long long)iz * div_xy;
  keys[i] = (voxel << 32) | (unsigned long long)(uint32_t)__float_as_int(p.w);
}

__global__ __launch_bounds__(256) void run_start_kernel(const unsigned long long *__restrict__ keys, uint32_t n_valid, uint32_t n,
                                                         uint32_t *__restrict__ flags /* n + 1 */) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p > n) return;
  flags[p] = (p < n_valid && (p == 0 || (keys[p - 1] >> 32) != (keys[p] >> 32))) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void centroid_kernel(CloudView c, const unsigned long long *__restrict__ keys,
                                                        const uint32_t *__restrict__ vals, const uint32_t *__restrict__ flags,
                                                        const uint32_t *__restrict__ slot, uint32_t n_valid,
                                                        float *__restrict__ out_xyz, const uint32_t *__restrict__ rgb_orig,
                                                        uint32_t *__restrict__ out_rgb) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n_valid || flags[p] == 0u) return;
  const unsigned long long vox = keys[p] >> 32;
  float sx = 0.f, sy = 0.f, sz = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
  uint32_t j = p;
  for (; j < n_valid && (keys[j] >> 32) == vox; ++j) {
    const float4 q = c.xyzw[vals[j]];
    sx = __fadd_rn(sx, q.x); sy = __fadd_rn(sy, q.y); sz = __fadd_rn(sz, q.z);
    if (rgb_orig) {
      // voxel_grid.hpp "RGB special case": the channels of pcl::RGB join the centroid vector as three floats
      const uint32_t v = rgb_orig[(uint32_t)__float_as_int(q.w)];
      cr = __fadd_rn(cr, (float)((v >> 16) & 255u)); cg = __fadd_rn(cg, (float)((v >> 8) & 255u)); cb = __fadd_rn(cb, (float)(v & 255u));
    }
  }
  const float r = __fdiv_rn(1.0f, (float)(j - p));
  float *o = out_xyz + 3 * (size_t)slot[p];
  o[0] = __fmul_rn(sx, r); o[1] = __fmul_rn(sy, r); o[2] = __fmul_rn(sz, r);
  if (rgb_orig)   // (static_cast<int>(r) << 16) | (static_cast<int>(g) << 8) | static_cast<int>(b)
    out_rgb[slot[p]] = ((uint32_t)(int)__fmul_rn(cr, r) << 16) | ((uint32_t)(int)__fmul_rn(cg, r) << 8) | (uint32_t)(int)__fmul_rn(cb, r);
}

// ---------------------------------------------------------------------------------------------------------------------
// StatisticalOutlierRemoval, pass 1: per point the mean distance to its mean_k nearest neighbours (the point itself is
// the first entry of the k = mean_k + 1 nearest and is skipped).  The k-NN distances come out of the tree walk in
// ascending order; their square roots are summed in double in that order, as statistical_outlier_removal.hpp does.
// KREG = mean_k + 1 for the reference's value (meanK 30, processingpcd.cpp:70): distances only, in registers.
template <int K>
struct KnnDistVisitor {
  float d[K];
  int count;
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < K; ++j) d[j] = INFINITY;
    count = 0;
  }
  __device__ __forceinline__ bool prune(float bound) const { return !(bound < d[K - 1]); }
  __device__ __forceinline__ void point(float dist, const v4f &, uint32_t, uint32_t) {
    const bool ins = dist < d[K - 1];
    if (__ballot(ins) == 0ull) return;
    count += (ins && count < K) ? 1 : 0;
    // sorted insert: new d[j] = median(d[j-1], d[j], dist), one v_med3_f32 per slot; lanes without a candidate unchanged
#pragma unroll
    for (int j = K - 1; j > 0; --j) d[j] = __builtin_amdgcn_fmed3f(d[j - 1], d[j], dist);
    d[0] = fminf(d[0], dist);
  }
  __device__ __forceinline__ void on_node() {}
};

template <int KREG>
__global__ __launch_bounds__(kKnnBlock) void sor_mean_distance_kernel(CloudView q, BvhView tgt, int mean_k,
                                                                       float *__restrict__ dist_orig, const uint32_t *__restrict__ self_leaf) {
  extern __shared__ unsigned char s_dyn[];
  float *ld = reinterpret_cast<float *>(s_dyn) + threadIdx.x;
  uint32_t *lp = reinterpret_cast<uint32_t *>(s_dyn + sizeof(float) * kKnnBlock * kKnnMaxK) + threadIdx.x;
  __shared__ float s_stk[kMaxDepth + 1][kKnnBlock];
  float *stk = &s_stk[0][threadIdx.x];
  for (uint32_t i = blockIdx.x * kKnnBlock + threadIdx.x; i < q.n; i += gridDim.x * kKnnBlock) {
    const float4 s = q.xyzw[i];
    const uint32_t orig = (uint32_t)__float_as_int(s.w);
    if (i >= q.n_valid) { dist_orig[orig] = 0.0f; continue; }   // non-finite: distance 0, not counted as valid
    double sum = 0.0;
    if constexpr (KREG > 0) {
      KnnDistVisitor<KREG> v;
      v.init();
      bvh_traverse(tgt, s.x, s.y, s.z, v, stk, kKnnBlock, self_leaf ? self_leaf[orig] : 0u);   // from the query's own leaf (features.hip: self_leaves)
#pragma unroll
      for (int j = 1; j < KREG; ++j)
        if (j < v.count) sum += sqrt((double)v.d[j]);
    } else {
      KnnVisitor v{ld, lp, kKnnBlock, mean_k + 1, 0, INFINITY};
      bvh_traverse(tgt, s.x, s.y, s.z, v, stk, kKnnBlock, self_leaf ? self_leaf[orig] : 0u);
      for (int j = 1; j < v.count; ++j) sum += sqrt((double)ld[j * kKnnBlock]);
    }
    dist_orig[orig] = (float)(sum / (double)mean_k);
  }
}

// CorrespondenceRejector predicates on given pairs, same arithmetic as the fused versions in icp_accumulate_kernel
__global__ __launch_bounds__(256) void reject_pairs_kernel(int kind, const float *__restrict__ a, const float *__restrict__ b, uint32_t n,
                                                           double thr, unsigned char *__restrict__ keep) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float ax = a[3 * i], ay = a[3 * i + 1], az = a[3 * i + 2], bx = b[3 * i], by = b[3 * i + 1], bz = b[3 * i + 2];
  bool ok;
  if (kind == OPE_REJ_SURFACE_NORMAL) {
    const float score = __fadd_rn(__fadd_rn(__fmul_rn(ax, bx), __fmul_rn(ay, by)), __fmul_rn(az, bz));
    ok = (double)score > thr;
  } else {
    const double sl = sqrt((double)__fadd_rn(__fadd_rn(__fmul_rn(bx, bx), __fmul_rn(by, by)), __fmul_rn(bz, bz)));
    const double score = (double)ax * (-(double)bx / sl) + (double)ay * (-(double)by / sl) + (double)az * (-(double)bz / sl);
    ok = score > thr;
  }
  keep[i] = ok ? 1 : 0;
}

// survivors' ORIGINAL indices in input order, left on the device (*d_out_ret, caller frees); count on the host
static int box_filter_dev(ope_ctx *ctx, const ope_cloud *cloud, const float lo[3], const float hi[3], int32_t **d_out_ret, unsigned int *count_ret,
                          const char *who) {
  *d_out_ret = nullptr;
  *count_ret = 0;
  const size_t n = cloud->n;
  if (n == 0) return OPE_OK;
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  unsigned char *d_flags = nullptr;
  int32_t *d_out = nullptr;
  unsigned int *d_count = nullptr;
  void *d_tmp = nullptr;
  unsigned int count = 0;
  hipError_t e = tmp_malloc(ctx->stream, (void **)&d_flags, n);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_out, 4 * n);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_count, 4);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(box_flags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, cloud->view(), lo[0], lo[1],
                       lo[2], hi[0], hi[1], hi[2], d_flags);
    rocprim::counting_iterator<int32_t> iota(0);
    size_t tb = 0;
    e = rocprim::select(nullptr, tb, iota, d_flags, d_out, d_count, n, ctx->stream);
    if (e == hipSuccess) e = tmp_malloc(ctx->stream, &d_tmp, tb);
    if (e == hipSuccess) e = rocprim::select(d_tmp, tb, iota, d_flags, d_out, d_count, n, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&count, d_count, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  for (void *p : {(void *)d_flags, (void *)d_count, d_tmp}) tmp_free(ctx->stream, p);
  if (e != hipSuccess) {
    tmp_free(ctx->stream, d_out);
    return set_err(ctx, OPE_EHIP, std::string(who) + ": " + hipGetErrorString(e));
  }
  *d_out_ret = d_out;
  *count_ret = count;
  return OPE_OK;
}

static int box_filter(ope_ctx *ctx, const ope_cloud *cloud, const float lo[3], const float hi[3], int32_t *out_idx,
                      size_t *n_out, const char *who) {
  *n_out = 0;
  int32_t *d_out = nullptr;
  unsigned int count = 0;
  const int rc = box_filter_dev(ctx, cloud, lo, hi, &d_out, &count, who);
  if (rc != OPE_OK) return rc;
  hipError_t e = hipSuccess;
  if (count) e = hipMemcpy(out_idx, d_out, 4 * (size_t)count, hipMemcpyDeviceToHost);
  tmp_free(ctx->stream, d_out);
  if (e != hipSuccess) return set_err(ctx, OPE_EHIP, std::string(who) + ": " + hipGetErrorString(e));
  OPE_DUMP_HASH(who, out_idx, 4 * (size_t)count, false);
  *n_out = count;
  return OPE_OK;
}

// the same with the survivors handed on as a device-resident cloud (and, optionally, their indices to the host): the flags
// by original index go straight into the order-preserving compaction (sampling.hip), nothing is sorted again
static int box_filter_cloud(ope_ctx *ctx, const ope_cloud *cloud, const float lo[3], const float hi[3], ope_cloud **out, int32_t *out_idx,
                            size_t *n_out, const char *who) {
  *out = nullptr;
  if (n_out) *n_out = 0;
  const size_t n = cloud->n;
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  unsigned char *d_flags = nullptr;
  int32_t *d_idx = nullptr;
  hipError_t e = tmp_malloc(ctx->stream, (void **)&d_flags, std::max<size_t>(n, 1));
  if (e == hipSuccess && out_idx) e = tmp_malloc(ctx->stream, (void **)&d_idx, 4 * std::max<size_t>(n, 1));
  int rc = OPE_OK;
  size_t count = 0;
  if (e == hipSuccess) {
    if (n)
      hipLaunchKernelGGL(box_flags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, cloud->view(), lo[0], lo[1], lo[2], hi[0], hi[1],
                         hi[2], d_flags);
    rc = compact_cloud_device(ctx, cloud, d_flags, out, d_idx, &count);
    if (rc == OPE_OK && out_idx && count) e = hipMemcpy(out_idx, d_idx, 4 * count, hipMemcpyDeviceToHost);
  }
  tmp_free(ctx->stream, d_flags);
  tmp_free(ctx->stream, d_idx);
  if (rc != OPE_OK) return rc;
  if (e != hipSuccess) { ope_cloud_free(*out); *out = nullptr; return set_err(ctx, OPE_EHIP, std::string(who) + ": " + hipGetErrorString(e)); }
  if (n_out) *n_out = count;
  return OPE_OK;
}

// keep[i] = !(dist[i] > thr), by ORIGINAL index (non-finite points carry distance 0 and pass: PCL quirk)
__global__ __launch_bounds__(256) void sor_flags_kernel(const float *__restrict__ dist, uint32_t n, double thr, unsigned char *__restrict__ keep) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) keep[i] = !((double)dist[i] > thr) ? 1 : 0;
}

// The filter's threshold without the trip through the host.  The reference sums the distance vector and its squares
// sequentially in double (filters/impl/statistical_outlier_removal.hpp as restated in oracle/filters.c:155-159); a parallel
// sum rounds differently in the last bits, so what the device computes is an INTERVAL that holds the reference's threshold
// whatever order it adds in — both orders lie within gamma_n = n u / (1 - n u) of the exact sums, u = 2^-53, and every later
// operation adds u — and then counts the distances inside it.  None there (the interval is ~1e-9 of the threshold wide): every
// comparison `distance > threshold` comes out as the reference's, bit for bit the same selection.  One there: the caller
// takes the sequential sums on the host, as before.
// out[0] = sum, out[1] = sum of squares (fp64 atomics over block partials)
__global__ __launch_bounds__(256) void sor_sums_kernel(const float *__restrict__ dist, uint32_t n, double *__restrict__ out) {
  __shared__ double s_a[256], s_b[256];
  double a = 0.0, b = 0.0;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const double d = (double)dist[i];
    a += d;
    b += d * d;
  }
  s_a[threadIdx.x] = a; s_b[threadIdx.x] = b;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) { s_a[threadIdx.x] += s_a[threadIdx.x + off]; s_b[threadIdx.x] += s_b[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { unsafeAtomicAdd(out, s_a[0]); unsafeAtomicAdd(out + 1, s_b[0]); }
}
// keep[i] as sor_flags_kernel with the interval's midpoint; *in_band += distances inside [thr_lo, thr_hi]
__global__ __launch_bounds__(256) void sor_flags_band_kernel(const float *__restrict__ dist, uint32_t n, double thr, double thr_lo, double thr_hi,
                                                             unsigned char *__restrict__ keep, uint32_t *__restrict__ in_band) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  bool band = false;
  if (i < n) {
    const double d = (double)dist[i];
    keep[i] = !(d > thr) ? 1 : 0;
    band = d >= thr_lo && d <= thr_hi;
  }
  if (__ballot(band) != 0ull && band) atomicAdd(in_band, 1u);
}

// the interval of the reference's threshold from sums of any order; false: no usable interval (take the sequential sums)
static bool sor_threshold_interval(double S, double Q, double n, double valid, double mul, double &thr, double &lo, double &hi) {
  if (!(valid > 1.0) || !(S >= 0.0) || !(Q >= 0.0) || !std::isfinite(S) || !std::isfinite(Q) || !std::isfinite(mul)) return false;
  const double u = 1.1102230246251565e-16;            // 2^-53
  const double eps = 4.0 * (n + 2.0) * u, w = 8.0 * u;  // (2 gamma_n with room; a few roundings of single operations)
  const double S_lo = S * (1.0 - eps), S_hi = S * (1.0 + eps), Q_lo = Q * (1.0 - eps), Q_hi = Q * (1.0 + eps);
  const double mean_lo = S_lo / valid * (1.0 - w), mean_hi = S_hi / valid * (1.0 + w);
  const double t_lo = S_lo * S_lo / valid * (1.0 - w), t_hi = S_hi * S_hi / valid * (1.0 + w);
  const double num_lo = Q_lo - t_hi, num_hi = Q_hi - t_lo;
  if (!(num_lo > 0.0)) return false;                   // (a variance that may round to zero or below: sqrt decides, sequentially)
  const double var_lo = num_lo * (1.0 - w) / (valid - 1.0) * (1.0 - w), var_hi = num_hi * (1.0 + w) / (valid - 1.0) * (1.0 + w);
  const double sd_lo = std::sqrt(var_lo) * (1.0 - w), sd_hi = std::sqrt(var_hi) * (1.0 + w);
  const double a = mul * sd_lo, b = mul * sd_hi;
  const double x = mean_lo + std::min(a, b), y = mean_hi + std::max(a, b);
  lo = x - std::fabs(x) * w - std::fabs(std::min(a, b)) * w;
  hi = y + std::fabs(y) * w + std::fabs(std::max(a, b)) * w;
  const double mean = S / valid, variance = (Q - S * S / valid) / (valid - 1.0);
  thr = mean + mul * std::sqrt(variance);
  if (!(thr >= lo && thr <= hi)) thr = 0.5 * (lo + hi);
  return std::isfinite(lo) && std::isfinite(hi) && lo <= hi;
}

}  // namespace ope

using namespace ope;

extern "C" int ope_remove_nan(ope_ctx *ctx, const ope_cloud *cloud, int32_t *out_idx, size_t *n_out) {
  if (!ctx || !cloud || !out_idx || !n_out) return set_err(ctx, OPE_EINVAL, "ope_remove_nan: bad argument");
  const float lo[3] = {-INFINITY, -INFINITY, -INFINITY}, hi[3] = {INFINITY, INFINITY, INFINITY};
  return box_filter(ctx, cloud, lo, hi, out_idx, n_out, "ope_remove_nan");
}

extern "C" int ope_pass_through(ope_ctx *ctx, const ope_cloud *cloud, const float lo[3], const float hi[3], int32_t *out_idx,
                                size_t *n_out) {
  if (!ctx || !cloud || !lo || !hi || !out_idx || !n_out) return set_err(ctx, OPE_EINVAL, "ope_pass_through: bad argument");
  return box_filter(ctx, cloud, lo, hi, out_idx, n_out, "ope_pass_through");
}

extern "C" int ope_voxel_grid(ope_ctx *ctx, const ope_cloud *cloud, const float leaf[3], float *out_xyz, size_t *n_out) {
  return ope_voxel_grid_rgb(ctx, cloud, leaf, nullptr, out_xyz, nullptr, n_out);
}

extern "C" int ope_voxel_grid_rgb(ope_ctx *ctx, const ope_cloud *cloud, const float leaf[3], const uint32_t *rgb, float *out_xyz,
                                  uint32_t *out_rgb, size_t *n_out) {
  if (!ctx || !cloud || !leaf || !out_xyz || !n_out || !(leaf[0] > 0) || !(leaf[1] > 0) || !(leaf[2] > 0) || (rgb && !out_rgb))
    return set_err(ctx, OPE_EINVAL, "ope_voxel_grid: bad argument");
  *n_out = 0;
  const size_t n = cloud->n, nv = cloud->n_valid;
  if (n == 0 || nv == 0) return OPE_OK;
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  float inv[3];
  long long min_b[3], div_b[3], dxyz[3];
  for (int d = 0; d < 3; ++d) {
    inv[d] = 1.0f / leaf[d];
    dxyz[d] = (long long)((cloud->bb_hi[d] - cloud->bb_lo[d]) * inv[d]) + 1;
    min_b[d] = (long long)(int)std::floor(cloud->bb_lo[d] * inv[d]);
    div_b[d] = (long long)(int)std::floor(cloud->bb_hi[d] * inv[d]) - min_b[d] + 1;
  }
  // voxel_grid.hpp: "Leaf size is too small for the input dataset. Integer indices would overflow." -> output = input
  if (dxyz[0] * dxyz[1] * dxyz[2] > 2147483647LL)
    return set_err(ctx, OPE_ERANGE, "ope_voxel_grid: leaf size too small for the input dataset (voxel index overflows)");
  unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
  uint32_t *d_vals = nullptr, *d_vals2 = nullptr, *d_flags = nullptr, *d_slot = nullptr;
  float *d_out = nullptr;
  uint32_t *d_rgb = nullptr, *d_out_rgb = nullptr;
  void *d_tmp = nullptr;
  uint32_t count = 0;
  hipError_t e = tmp_malloc(ctx->stream, (void **)&d_keys, 8 * n);
  if (e == hipSuccess && rgb) e = tmp_malloc(ctx->stream, (void **)&d_rgb, 4 * n);
  if (e == hipSuccess && rgb) e = tmp_malloc(ctx->stream, (void **)&d_out_rgb, 4 * nv);
  if (e == hipSuccess && rgb) e = h2d_copy(ctx->stream, d_rgb, rgb, 4 * n);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_keys2, 8 * n);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_vals, 4 * n);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_vals2, 4 * n);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_flags, 4 * (n + 1));
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_slot, 4 * (n + 1));
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_out, 12 * nv);
  if (e == hipSuccess) {
    const unsigned nb = (unsigned)((n + 256) / 256);
    hipLaunchKernelGGL(grid_key_kernel, dim3(nb), dim3(256), 0, ctx->stream, cloud->view(), inv[0], inv[1], inv[2], (int)min_b[0],
                       (int)min_b[1], (int)min_b[2], (unsigned)div_b[0], (unsigned long long)(div_b[0] * div_b[1]), d_keys, d_vals);
    size_t tmp_sort = 0, tmp_scan = 0;
    e = rocprim::radix_sort_pairs(nullptr, tmp_sort, d_keys, d_keys2, d_vals, d_vals2, n, 0, 64, ctx->stream);
    if (e == hipSuccess)
      e = rocprim::exclusive_scan(nullptr, tmp_scan, d_flags, d_slot, 0u, n + 1, rocprim::plus<uint32_t>(), ctx->stream);
    const size_t tmp_bytes = std::max(tmp_sort, tmp_scan);
    if (e == hipSuccess) e = tmp_malloc(ctx->stream, &d_tmp, std::max<size_t>(tmp_bytes, 16));
    size_t tb = tmp_bytes;
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(d_tmp, tb, d_keys, d_keys2, d_vals, d_vals2, n, 0, 64, ctx->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(run_start_kernel, dim3(nb), dim3(256), 0, ctx->stream, d_keys2, (uint32_t)nv, (uint32_t)n, d_flags);
      tb = tmp_bytes;
      e = rocprim::exclusive_scan(d_tmp, tb, d_flags, d_slot, 0u, n + 1, rocprim::plus<uint32_t>(), ctx->stream);
    }
    if (e == hipSuccess) {
      hipLaunchKernelGGL(centroid_kernel, dim3(nb), dim3(256), 0, ctx->stream, cloud->view(), d_keys2, d_vals2, d_flags, d_slot,
                         (uint32_t)nv, d_out, d_rgb, d_out_rgb);
      e = hipMemcpyAsync(&count, d_slot + n, 4, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && count) e = hipMemcpy(out_xyz, d_out, 12 * (size_t)count, hipMemcpyDeviceToHost);
    if (e == hipSuccess && count && rgb) e = hipMemcpy(out_rgb, d_out_rgb, 4 * (size_t)count, hipMemcpyDeviceToHost);
  }
  for (void *p : {(void *)d_keys, (void *)d_keys2, (void *)d_vals, (void *)d_vals2, (void *)d_flags, (void *)d_slot, (void *)d_out,
                  (void *)d_rgb, (void *)d_out_rgb, d_tmp})
    tmp_free(ctx->stream, p);
  if (e != hipSuccess) return set_err(ctx, OPE_EHIP, std::string("ope_voxel_grid: ") + hipGetErrorString(e));
  *n_out = count;
  return OPE_OK;
}

// Core of StatisticalOutlierRemoval.  want_cloud: the inliers as a device-resident cloud (*out_cloud); out_idx (optional):
// their ORIGINAL indices on the host.
static int sor_core(ope_ctx *ctx, const ope_cloud *cloud, int mean_k, double stddev_mul, int32_t *out_idx, size_t *n_out, float *out_mean_dist,
                    bool want_cloud, ope_cloud **out_cloud) {
  if (n_out) *n_out = 0;
  if (out_cloud) *out_cloud = nullptr;
  const size_t n = cloud->n;
  if (n == 0) return want_cloud ? select_cloud_device(ctx, cloud, nullptr, 0, out_cloud) : OPE_OK;
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  // the device-resident form keeps the distances on the device unless its threshold interval leaves a comparison open
  const bool device_threshold = want_cloud && out_mean_dist == nullptr && cloud->n_valid > 1;
  std::vector<float> dist(n, 0.0f);
  float *d_dist = nullptr;
  OPE_HIP(ctx, tmp_malloc(ctx->stream, (void **)&d_dist, 4 * n));
  struct FreeDist { hipStream_t s; float *p; ~FreeDist() { tmp_free(s, p); } } free_dist{ctx->stream, d_dist};
  if (cloud->n_valid > 0) {
    TraceRange r(ctx, "sor");
    ope_index *ix = nullptr;
    ope_index_params ip;
    ope_index_default_params(&ip);
    ip.grid = 0;   // this index serves one k-NN pass
    int rc = index_build_tmp(ctx, cloud, &ip, &ix);   // (a temporary of this call: no hipMalloc / hipFree)
    if (rc != OPE_OK) return rc;
    hipError_t e = hipSuccess;
    {
      const int nblocks = (int)std::min<size_t>((n + kKnnBlock - 1) / kKnnBlock, 8192);
      // algorithmic bytes, by analogy with the normals: read the point, gather k neighbours, write one float
      KernelTimer kt(ctx, "sor_mean_distance_kernel", (double)cloud->n_valid * (12.0 + 12.0 * (mean_k + 1) + 4.0));
      uint32_t *d_self_leaf = nullptr;
      // every query starts at the leaf that holds it: 914 k points, meanK 30: 7.2 -> 1.75 ms for the 1 M-point frame
      if (tmp_malloc(ctx->stream, (void **)&d_self_leaf, 4 * n) == hipSuccess &&
          self_leaves(ctx->stream, ix->view(), n, d_self_leaf) != hipSuccess) { tmp_free(ctx->stream, d_self_leaf); d_self_leaf = nullptr; }
      if (mean_k == 30)
        hipLaunchKernelGGL(sor_mean_distance_kernel<31>, dim3(nblocks), dim3(kKnnBlock), 0, ctx->stream, cloud->view(), ix->view(), mean_k, d_dist, d_self_leaf);
      else
        hipLaunchKernelGGL(sor_mean_distance_kernel<0>, dim3(nblocks), dim3(kKnnBlock), kKnnLdsBytes, ctx->stream, cloud->view(), ix->view(),
                           mean_k, d_dist, d_self_leaf);
      if (d_self_leaf) tmp_free(ctx->stream, d_self_leaf);
      kt.stop();
      if (!device_threshold) {
        e = hipMemcpyAsync(dist.data(), d_dist, 4 * n, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      }
    }
    ope_index_free(ix);
    if (e != hipSuccess) return set_err(ctx, OPE_EHIP, std::string("ope_statistical_outlier_removal: ") + hipGetErrorString(e));
  } else {
    OPE_HIP(ctx, hipMemsetAsync(d_dist, 0, 4 * n, ctx->stream));
  }
  if (device_threshold) {
    // ---- the device-resident form: threshold interval from parallel sums, selection on the device (sor_sums_kernel)
    double *d_sums = nullptr;
    uint32_t *d_band = nullptr;
    unsigned char *d_flags = nullptr;
    int32_t *d_sel = nullptr;
    hipError_t e = tmp_malloc(ctx->stream, (void **)&d_sums, 32);
    d_band = reinterpret_cast<uint32_t *>(d_sums + 2);
    if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_flags, n);
    if (e == hipSuccess && out_idx) e = tmp_malloc(ctx->stream, (void **)&d_sel, 4 * n);
    struct FreeAll { hipStream_t s; void *a, *b, *c; ~FreeAll() { tmp_free(s, a); tmp_free(s, b); tmp_free(s, c); } } free_all{ctx->stream, d_sums, d_flags, d_sel};
    double hs[2] = {0.0, 0.0};
    if (e == hipSuccess) e = hipMemsetAsync(d_sums, 0, 32, ctx->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(sor_sums_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0, ctx->stream, d_dist, (uint32_t)n, d_sums);
      e = hipMemcpyAsync(hs, d_sums, 16, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return set_err(ctx, OPE_EHIP, std::string("ope_statistical_outlier_removal_cloud: ") + hipGetErrorString(e));
    double thr = 0.0, thr_lo = 0.0, thr_hi = 0.0;
    bool settled = sor_threshold_interval(hs[0], hs[1], (double)n, (double)cloud->n_valid, stddev_mul, thr, thr_lo, thr_hi);
    int rc = OPE_OK;
    size_t count = 0;
    if (settled) {
      uint32_t band = 0;
      hipLaunchKernelGGL(sor_flags_band_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_dist, (uint32_t)n, thr, thr_lo, thr_hi, d_flags, d_band);
      e = hipMemcpyAsync(&band, d_band, 4, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) rc = compact_cloud_device(ctx, cloud, d_flags, out_cloud, d_sel, &count);   // (synchronises: `band` has arrived)
      if (e != hipSuccess || rc != OPE_OK) (void)hipStreamSynchronize(ctx->stream);   // (no copy into `band` may outlive this frame)
      if (e != hipSuccess) return set_err(ctx, OPE_EHIP, std::string("ope_statistical_outlier_removal_cloud: ") + hipGetErrorString(e));
      if (rc != OPE_OK) return rc;
      settled = band == 0u;
      if (!settled) { ope_cloud_free(*out_cloud); *out_cloud = nullptr; }
    }
    if (settled) {
      if (out_idx && count) e = hipMemcpy(out_idx, d_sel, 4 * count, hipMemcpyDeviceToHost);
      if (e != hipSuccess) { ope_cloud_free(*out_cloud); *out_cloud = nullptr; return set_err(ctx, OPE_EHIP, std::string("ope_statistical_outlier_removal_cloud: ") + hipGetErrorString(e)); }
      if (n_out) *n_out = count;
      return OPE_OK;
    }
    // a distance inside the interval (or no interval): the reference's own sequential sums decide
    OPE_HIP(ctx, hipMemcpy(dist.data(), d_dist, 4 * n, hipMemcpyDeviceToHost));
  }
  // mean and standard deviation of the distance vector: the reference's own sequential double sums over the points in
  // input order (4 bytes per point through the host)
  double sum = 0.0, sq_sum = 0.0;
  for (size_t i = 0; i < n; ++i) { sum += dist[i]; sq_sum += (double)dist[i] * (double)dist[i]; }
  const double valid = (double)cloud->n_valid;
  const double mean = sum / valid;
  const double variance = (sq_sum - sum * sum / valid) / (valid - 1.0);
  const double thr = mean + stddev_mul * std::sqrt(variance);
  if (out_mean_dist) std::memcpy(out_mean_dist, dist.data(), 4 * n);
  if (!want_cloud) {
    size_t m = 0;
    for (size_t i = 0; i < n; ++i)
      if (!((double)dist[i] > thr)) out_idx[m++] = (int32_t)i;   // non-finite points carry distance 0 and pass (PCL quirk)
    *n_out = m;
    return OPE_OK;
  }
  // the inliers stay on the device: flags from the same comparison, compacted in input order (no re-sort)
  unsigned char *d_flags = nullptr;
  int32_t *d_sel = nullptr;
  hipError_t e = tmp_malloc(ctx->stream, (void **)&d_flags, n);
  if (e == hipSuccess && out_idx) e = tmp_malloc(ctx->stream, (void **)&d_sel, 4 * n);
  int rc = OPE_OK;
  size_t count = 0;
  if (e == hipSuccess) {
    hipLaunchKernelGGL(sor_flags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_dist, (uint32_t)n, thr, d_flags);
    rc = compact_cloud_device(ctx, cloud, d_flags, out_cloud, d_sel, &count);
    if (rc == OPE_OK && out_idx && count) e = hipMemcpy(out_idx, d_sel, 4 * count, hipMemcpyDeviceToHost);
  }
  tmp_free(ctx->stream, d_flags);
  tmp_free(ctx->stream, d_sel);
  if (rc != OPE_OK) return rc;
  if (e != hipSuccess) { ope_cloud_free(*out_cloud); *out_cloud = nullptr; return set_err(ctx, OPE_EHIP, std::string("ope_statistical_outlier_removal_cloud: ") + hipGetErrorString(e)); }
  if (n_out) *n_out = count;
  return OPE_OK;
}

extern "C" int ope_statistical_outlier_removal(ope_ctx *ctx, const ope_cloud *cloud, int mean_k, double stddev_mul,
                                               int32_t *out_idx, size_t *n_out, float *out_mean_dist) {
  if (!ctx || !cloud || !out_idx || !n_out || mean_k < 1 || mean_k + 1 > kKnnMaxK)
    return set_err(ctx, OPE_EINVAL, "ope_statistical_outlier_removal: bad argument (1 <= mean_k <= 31)");
  return sor_core(ctx, cloud, mean_k, stddev_mul, out_idx, n_out, out_mean_dist, false, nullptr);
}

extern "C" int ope_statistical_outlier_removal_cloud(ope_ctx *ctx, const ope_cloud *cloud, int mean_k, double stddev_mul, ope_cloud **out,
                                                     int32_t *out_idx, size_t *n_out) {
  if (!ctx || !cloud || !out || mean_k < 1 || mean_k + 1 > kKnnMaxK)
    return set_err(ctx, OPE_EINVAL, "ope_statistical_outlier_removal_cloud: bad argument (1 <= mean_k <= 31)");
  return sor_core(ctx, cloud, mean_k, stddev_mul, out_idx, n_out, nullptr, true, out);
}

extern "C" int ope_pass_through_cloud(ope_ctx *ctx, const ope_cloud *cloud, const float lo[3], const float hi[3], ope_cloud **out, int32_t *out_idx,
                                      size_t *n_out) {
  if (!ctx || !cloud || !lo || !hi || !out) return set_err(ctx, OPE_EINVAL, "ope_pass_through_cloud: bad argument");
  return box_filter_cloud(ctx, cloud, lo, hi, out, out_idx, n_out, "ope_pass_through_cloud");
}

extern "C" int ope_remove_nan_cloud(ope_ctx *ctx, const ope_cloud *cloud, ope_cloud **out, int32_t *out_idx, size_t *n_out) {
  if (!ctx || !cloud || !out) return set_err(ctx, OPE_EINVAL, "ope_remove_nan_cloud: bad argument");
  const float lo[3] = {-INFINITY, -INFINITY, -INFINITY}, hi[3] = {INFINITY, INFINITY, INFINITY};
  return box_filter_cloud(ctx, cloud, lo, hi, out, out_idx, n_out, "ope_remove_nan_cloud");
}

extern "C" int ope_reject_pairs(ope_ctx *ctx, int kind, const float *a, const float *b, size_t n, double threshold, unsigned char *keep) {
  if (!ctx || (n && (!a || !b || !keep)) || (kind != OPE_REJ_SURFACE_NORMAL && kind != OPE_REJ_SELF_OCCLUDED) || n > (size_t)0x7fffffff)
    return set_err(ctx, OPE_EINVAL, "ope_reject_pairs: bad argument");
  if (n == 0) return OPE_OK;
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  float *d_a = nullptr, *d_b = nullptr;
  unsigned char *d_k = nullptr;
  hipError_t e = tmp_malloc(ctx->stream, (void **)&d_a, 12 * n);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_b, 12 * n);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_k, n);
  if (e == hipSuccess) e = h2d_copy(ctx->stream, d_a, a, 12 * n);
  if (e == hipSuccess) e = h2d_copy(ctx->stream, d_b, b, 12 * n);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(reject_pairs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, kind, d_a, d_b, (uint32_t)n, threshold, d_k);
    e = hipMemcpyAsync(keep, d_k, n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  for (void *p : {(void *)d_a, (void *)d_b, (void *)d_k})
    tmp_free(ctx->stream, p);
  if (e != hipSuccess) return set_err(ctx, OPE_EHIP, std::string("ope_reject_pairs: ") + hipGetErrorString(e));
  return OPE_OK;
}
