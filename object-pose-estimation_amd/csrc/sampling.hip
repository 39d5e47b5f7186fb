// sampling.hip — pcl::UniformSampling::compute(PointCloud<int>&) on the device
// (reference call: DetectAndLocalize/src/poseestimator.cpp:141-145, PCL <= 1.7 keypoints API).
//
// PCL hashes points into voxels of edge `leaf` and keeps, per voxel, the input index whose point is
// "closest to the voxel's integer coordinates" — it subtracts INTEGER voxel indices from METRIC
// coordinates (and carries the w = 1 lane into the norm), first index winning ties (quirk Q7) — and
// emits survivors in boost::unordered_map order.  Here: one 64-bit key (voxel << 32 | input index)
// per point, a rocPRIM radix sort, one lane per voxel segment replaying PCL's comparison in input
// order, and a rocPRIM select; survivors therefore come out in ascending voxel-key order, which is
// deterministic (the reference's order is unspecified).
#include <cstring>
#include <string>

#include <algorithm>
#include <rocprim/rocprim.hpp>

#include <cfloat>
#include <cmath>
#include <vector>

#include "ope_internal.hpp"

namespace ope {

__global__ __launch_bounds__(256) void voxel_key_kernel(CloudView c, float inv_leaf, int min_bx, int min_by, int min_bz,
                                                         unsigned div_x, unsigned div_xy, int idx_bits, unsigned long long *keys,
                                                         uint32_t *vals) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= c.n) return;
  vals[i] = i;
  if (i >= c.n_valid) { keys[i] = ~0ull; return; }
  const float4 p = c.xyzw[i];
  const int ix = (int)floorf(p.x * inv_leaf) - min_bx, iy = (int)floorf(p.y * inv_leaf) - min_by,
            iz = (int)floorf(p.z * inv_leaf) - min_bz;
  const unsigned long long voxel = (unsigned long long)ix + (unsigned long long)iy * div_x + (unsigned long long)iz * div_xy;
  // (voxel ‖ original index in as few bits as they need: the sort below runs over those bits only)
  keys[i] = (voxel << idx_bits) | (unsigned long long)(uint32_t)__float_as_int(p.w);
}

// The voxel's survivor (uniform_sampling.hpp).  (Until round 4 one lane per voxel walked its members in order; a voxel of the C3
// cluster holds a thousand points: 0.8 ms of the 1.3 the call took.)  PCL's rule — the first member, replaced by every later one that is STRICTLY nearer to the voxel's
// integer corner — is the minimum of (distance, position in the sorted order): a wave takes 64 consecutive positions, reduces
// runs of equal voxels with shuffles, and each run's head adds its minimum to the voxel's slot, best[position of the voxel's first
// member], with one 64-bit atomicMin (distances are >= 1: their float bits order like the floats).  A run that continues a
// voxel begun in an earlier wave finds that voxel's first member by bisection in the sorted keys.
__global__ __launch_bounds__(256) void voxel_min_kernel(CloudView c, float inv_leaf, const unsigned long long *__restrict__ keys,
                                                         const uint32_t *__restrict__ vals, uint32_t n_valid, int idx_bits,
                                                         unsigned long long *__restrict__ best) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  const bool on = p < n_valid;
  unsigned long long vox = ~0ull, m = ~0ull;
  if (on) {
    vox = keys[p] >> idx_bits;
    const float4 q = c.xyzw[vals[p]];
    const float ix = floorf(q.x * inv_leaf), iy = floorf(q.y * inv_leaf), iz = floorf(q.z * inv_leaf);
    const float d = (q.x - ix) * (q.x - ix) + (q.y - iy) * (q.y - iy) + (q.z - iz) * (q.z - iz) + 1.0f;
    m = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)p;
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long ov = __shfl_down(vox, off, 64), om = __shfl_down(m, off, 64);
    if (lane + off < 64u && ov == vox && om < m) m = om;
  }
  const unsigned long long pv = __shfl_up(vox, 1, 64);
  if (!on || (lane != 0u && pv == vox)) return;   // not the head of a run
  uint32_t first = p;
  if (lane == 0u && p > 0u && (keys[p - 1] >> idx_bits) == vox) {
    uint32_t lo = 0u, hi = p - 1u;                 // first position whose voxel is not below this one
    while (lo < hi) {
      const uint32_t mid = (lo + hi) / 2u;
      if ((keys[mid] >> idx_bits) < vox) lo = mid + 1u; else hi = mid;
    }
    first = lo;
  }
  atomicMin(best + first, m);
}

__global__ __launch_bounds__(256) void voxel_pick_kernel(CloudView c, const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                          uint32_t n_valid, int idx_bits, const unsigned long long *__restrict__ best,
                                                          int32_t *__restrict__ winner, unsigned char *__restrict__ flags) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= c.n) return;
  const bool start = p < n_valid && (p == 0u || (keys[p - 1] >> idx_bits) != (keys[p] >> idx_bits));
  winner[p] = start ? __float_as_int(c.xyzw[vals[(uint32_t)(best[p] & 0xffffffffull)]].w) : -1;
  flags[p] = start ? 1 : 0;
}

}  // namespace ope

namespace ope {

__global__ void iota_kernel(uint32_t *v, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) v[i] = i;
}

// Chunk plan of the ICP accumulate kernel: chunk ids sorted by DESCENDING measured cost.
// tmp/tmp_bytes: caller-provided scratch (call once with tmp == nullptr to size it).
int chunk_plan(hipStream_t stream, const uint32_t *cost, uint32_t *cost_sorted, const uint32_t *ids, uint32_t *order,
               uint32_t n, void *tmp, size_t &tmp_bytes) {
  hipError_t e = rocprim::radix_sort_pairs_desc(tmp, tmp_bytes, cost, cost_sorted, ids, order, n, 0, 32, stream);
  return e == hipSuccess ? 0 : -1;
}

// first index of a descending cost list whose entry is <= thr, searched in [0, cap)
__device__ __forceinline__ uint32_t count_costlier(const uint32_t *cost_sorted_desc, uint32_t cap, float thr) {
  uint32_t lo = 0, hi = cap;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) / 2;
    if ((float)cost_sorted_desc[mid] > thr) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Plan of a launch from the chunk costs in descending order (one block of 256 threads).
// plan_info[0] = n_heavy, the chunks walked by 8-lane groups (capped at n/4):
//   factor > 0:      those costlier than factor x the median chunk (launches with few chunks per wave: latency-bound)
//   load_factor > 0: those costlier than load_factor x (sum of all costs / waves), i.e. chunks that on their own far
//                    outlast the share of work a wave has in a balanced launch (launches that fill the GPU)
// plan_info[5] = n_alone, the next-costliest chunks that get a wave to themselves: those at or above the fair share L of
//   the waves that are left, L = (work not yet given away) / (waves not yet given away), iterated to its fixed point.  A
//   group-walked chunk counts as kOctWork x its per-lane cost (eight slots of about a third of the duration each).
constexpr double kOctWork = 8.0 * kOctSlotShare;  // wave-time of a group-walked chunk relative to its per-lane walk
bool g_plan_no_alone = false;   // developer A/B switch (OPE_NO_ALONE, set by api.hip in DEVELOPER builds)
__device__ __forceinline__ double block_sum_range(const uint32_t *v, uint32_t a, uint32_t b, double *s_sum) {
  double acc = 0.0;
  for (uint32_t i = a + threadIdx.x; i < b; i += 256) acc += (double)v[i];
  __syncthreads();   // s_sum may still be read from the previous call
  s_sum[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) s_sum[threadIdx.x] += s_sum[threadIdx.x + off];
    __syncthreads();
  }
  return s_sum[0];
}
__global__ __launch_bounds__(1024) void plan_heavy_kernel(const uint32_t *cost_sorted_desc, uint32_t n, float factor, float load_factor,
                                                          uint32_t n_waves, uint32_t *plan_info) {
  // prefix sums of the sorted costs at a stride of `per` entries (one pass, one block scan); thread 0 then answers every
  // "cost of the m costliest chunks" with one LDS read and fewer than `per` loads
  using scan_t = rocprim::block_scan<double, 1024>;
  __shared__ typename scan_t::storage_type scan_storage;
  __shared__ double s_pre[1024];
  const uint32_t per = (n + 1023u) / 1024u;
  double acc = 0.0;
  for (uint32_t i = threadIdx.x * per; i < min(n, (threadIdx.x + 1u) * per); ++i) acc += (double)cost_sorted_desc[i];
  double excl = 0.0;
  scan_t().exclusive_scan(acc, excl, 0.0, scan_storage);
  s_pre[threadIdx.x] = excl;
  __syncthreads();
  if (threadIdx.x != 0) return;
  auto sum_first = [&](uint32_t m) {
    m = min(m, n);
    const uint32_t blk = per ? min(m / per, 1023u) : 0u;
    double v = s_pre[blk];
    for (uint32_t i = blk * per; i < m; ++i) v += (double)cost_sorted_desc[i];
    return v;
  };
  const double total = sum_first(n);
  uint32_t nh = 0, k = 0;
  double rest = total;
  if (n >= 8) {
    if (factor > 0.f) nh = count_costlier(cost_sorted_desc, n / 4, factor * (float)cost_sorted_desc[n / 2]);
    else if (load_factor > 0.f) nh = count_costlier(cost_sorted_desc, n / 4, load_factor * (float)(total / (double)max(n_waves, 1u)));
  }
  const double heavy = sum_first(nh);
  // the eight slots of a group-walked chunk are dealt like ordinary chunks; everything but the alone chunks is "the rest"
  rest = kOctWork * heavy + (total - heavy);
  for (int pass = 0; pass < 4; ++pass) {
    const uint32_t w_left = n_waves > k ? n_waves - k : 1u;
    const float fair = (float)(rest / (double)w_left);
    // chunks nh .. n-1 at or above the fair share (descending list: a prefix), at most half the waves
    uint32_t lo = nh, hi = min(n, nh + n_waves / 2u);
    while (lo < hi) {
      const uint32_t mid = (lo + hi) / 2;
      if ((float)cost_sorted_desc[mid] >= fair && fair > 0.f) lo = mid + 1; else hi = mid;
    }
    const uint32_t k_new = lo - nh;
    if (k_new == k) break;
    k = k_new;
    rest = kOctWork * heavy + (total - heavy) - (sum_first(nh + k) - heavy);
  }
  plan_info[0] = nh;
  plan_info[5] = k;
  // for the merged slot list (plan_slots_kernel): how many of its leading entries get a wave to themselves — the k
  // per-lane chunks above, plus the slots of group-walked chunks that on their own reach the fair share (rare)
  const uint32_t w_left = n_waves > k ? n_waves - k : 1u;
  const float fair = (float)(rest / (double)w_left);
  uint32_t lo = 0, hi = nh;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) / 2;
    if (kOctSlotShare * (float)cost_sorted_desc[mid] >= fair && fair > 0.f) lo = mid + 1; else hi = mid;
  }
  plan_info[7] = min(k + 8u * lo, n_waves / 2u);
}

// The launch's slots in descending order of expected duration: a per-lane chunk of rank q counts its cost, each of the
// eight slots of a group-walked chunk of rank i < n_heavy counts kOctSlotShare x the chunk's cost.  Both sequences are
// descending already, so every entry finds its place with one binary search in the other (ties: per-lane first).
// Entry: rank in bits 0-27, slot in bits 28-30, bit 31 = walked by 8-lane groups.
__global__ __launch_bounds__(256) void plan_slots_kernel(const uint32_t *__restrict__ cost_sorted_desc, uint32_t n, const uint32_t *__restrict__ plan_info,
                                                         uint32_t *__restrict__ slot_list) {
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const uint32_t nh = min(plan_info[0], n);
  const float c = (float)cost_sorted_desc[r];
  if (r < nh) {
    const float key = kOctSlotShare * c;
    uint32_t lo = nh, hi = n;   // first per-lane rank whose cost is below the key
    while (lo < hi) {
      const uint32_t mid = (lo + hi) / 2;
      if ((float)cost_sorted_desc[mid] >= key) lo = mid + 1; else hi = mid;
    }
    const uint32_t base = 8u * r + (lo - nh);
#pragma unroll
    for (uint32_t j = 0; j < 8u; ++j) slot_list[base + j] = r | (j << 28) | 0x80000000u;
  } else {
    uint32_t lo = 0, hi = nh;   // first group-walked rank whose key is not above this cost
    while (lo < hi) {
      const uint32_t mid = (lo + hi) / 2;
      if (kOctSlotShare * (float)cost_sorted_desc[mid] > c) lo = mid + 1; else hi = mid;
    }
    slot_list[(r - nh) + 8u * lo] = r;
  }
}

void plan_slots(hipStream_t stream, const uint32_t *cost_sorted_desc, uint32_t n, const uint32_t *plan_info, uint32_t *slot_list) {
  hipLaunchKernelGGL(plan_slots_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, cost_sorted_desc, n, plan_info, slot_list);
}

void plan_heavy(hipStream_t stream, const uint32_t *cost_sorted_desc, uint32_t n, float factor, float load_factor, uint32_t n_waves,
                uint32_t *plan_info) {
  hipLaunchKernelGGL(plan_heavy_kernel, dim3(1), dim3(1024), 0, stream, cost_sorted_desc, n, factor, load_factor, n_waves, plan_info);
  if (g_plan_no_alone) (void)hipMemsetAsync(plan_info + 5, 0, 4, stream);
}

// ---- plan step of the GRID accumulate kernel (icp_accumulate_grid_kernel): everything stays on the device
// plan_info[0] = n_heavy, plan_info[1] = n_grid_q (written by the partition below)
//
// 1. query order: grid-class queries first (flag 1), tree-class after them (rocPRIM partition of 0..n-1 by the class
//    flags; the rejected part comes out in reverse order, which keeps it spatially coherent all the same)
// 2. sort keys of the tree chunks: chunk ids below the number of tree chunks carry max(cost, 1), all others 0, so that
//    the first n_tc entries of the descending sort are exactly a permutation of the tree chunks
// 3. the number of heavy chunks among them
__global__ void grid_plan_keys_kernel(const uint32_t *__restrict__ cost, uint32_t nch, uint32_t n_valid, const uint32_t *__restrict__ plan_info,
                                      uint32_t *__restrict__ keys) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= nch) return;
  const uint32_t n_tc = (n_valid - min(plan_info[1], n_valid) + 63u) / 64u;
  keys[c] = c < n_tc ? max(cost[c], 1u) : 0u;
}

__global__ __launch_bounds__(256) void grid_plan_heavy_kernel(const uint32_t *cost_sorted_desc, const uint32_t *cost, uint32_t n_valid, uint32_t n_waves,
                                                              float factor_override, float load_factor, uint32_t *plan_info) {
  __shared__ double s_sum[256];
  __shared__ uint32_t s_nh, s_k;
  const uint32_t n = (n_valid - min(plan_info[1], n_valid) + 63u) / 64u;       // tree chunks: the first n entries of the sorted list
  const uint32_t n_gc = (min(plan_info[1], n_valid) + 63u) / 64u;              // grid chunks: their costs sit behind the tree chunks' in `cost`
  const double tree_total = block_sum_range(cost_sorted_desc, 0, n, s_sum);
  const double grid_total = block_sum_range(cost, n, n + n_gc, s_sum);
  const double total = tree_total + grid_total;
  if (threadIdx.x == 0) {
    // the fewer tree chunks there are per wave, the more the slowest one decides the launch (see enqueue_accumulate)
    const float cpw = (float)n / (float)max(n_waves, 1u);
    const float factor = factor_override >= 0.f ? factor_override : fminf(7.0f, fmaxf(2.0f, 1.2f + 1.5f * cpw));
    // 8-lane group walks trade lane-cycles for latency.  A launch that leaves waves idle (its tree chunks x8 plus its grid
    // chunks at about a third each do not fill them) takes the median rule; a full one only hands over the chunks that
    // far outlast a wave's fair share of the work.
    const float load = (8.0f * (float)n + 0.33f * (float)n_gc) / (float)max(n_waves, 1u);
    uint32_t nh = 0;
    if (n >= 8) {
      if (factor_override >= 0.f || load <= 1.8f) { if (factor > 0.f) nh = count_costlier(cost_sorted_desc, n / 4, factor * (float)cost_sorted_desc[n / 2]); }
      else if (load_factor > 0.f) nh = count_costlier(cost_sorted_desc, n / 4, load_factor * (float)(total / (double)max(n_waves, 1u)));
    }
    s_nh = nh;
    s_k = 0;
  }
  __syncthreads();
  const uint32_t nh = s_nh;
  const double heavy = block_sum_range(cost_sorted_desc, 0, nh, s_sum);
  double rest = kOctWork * heavy + (total - heavy);
  uint32_t k = 0;
  for (int pass = 0; pass < 4; ++pass) {
    if (threadIdx.x == 0) {
      const uint32_t w_left = n_waves > k ? n_waves - k : 1u;
      const float fair = (float)(rest / (double)w_left);
      uint32_t lo = nh, hi = min(n, nh + n_waves / 2u);
      while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if ((float)cost_sorted_desc[mid] >= fair && fair > 0.f) lo = mid + 1; else hi = mid;
      }
      s_k = lo - nh;
    }
    __syncthreads();
    const uint32_t k_new = s_k;
    if (k_new == k) break;
    const double alone = block_sum_range(cost_sorted_desc, nh, nh + k_new, s_sum);
    rest = kOctWork * heavy + (total - heavy) - alone;
    k = k_new;
  }
  if (threadIdx.x == 0) {
    plan_info[0] = nh;
    plan_info[5] = k;
  }
}

// Share of queries the grid can never answer, whatever the pose: their nearest model point is further away than one
// cell, so the ball around them spans more than the 27 cells the scan covers (clutter; a query that merely MOVED a lot
// since the last iteration has a small d2 and comes back to the grid once the loop settles).  plan_info[2] = count.
__global__ __launch_bounds__(256) void grid_count_far_kernel(const float *__restrict__ corr_d2, uint32_t n_valid, float thr2, uint32_t *__restrict__ plan_info) {
  uint32_t c = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_valid; i += gridDim.x * 256) c += corr_d2[i] > thr2 ? 1u : 0u;
  for (int off = 32; off >= 1; off >>= 1) c += (uint32_t)__shfl_xor((int)c, off, 64);
  if ((threadIdx.x & 63u) == 0 && c) atomicAdd(plan_info + 2, c);
}
void grid_count_far(hipStream_t stream, const float *corr_d2, uint32_t n_valid, float thr2, uint32_t *plan_info) {
  (void)hipMemsetAsync(plan_info + 2, 0, 4, stream);
  hipLaunchKernelGGL(grid_count_far_kernel, dim3(std::min<uint32_t>((n_valid + 255) / 256, 512)), dim3(256), 0, stream, corr_d2, n_valid, thr2, plan_info);
}

// d_keys: scratch of nch entries.  tmp sized by grid_plan_tmp_bytes.
int grid_plan(hipStream_t stream, bool repartition, const unsigned char *qclass, uint32_t n_valid, uint32_t *qorder, uint32_t *plan_info,
              const uint32_t *cost, uint32_t *keys, uint32_t *cost_sorted, const uint32_t *ids, uint32_t *order, uint32_t nch, uint32_t n_waves,
              float factor_override, float load_factor, void *tmp, size_t tmp_bytes) {
  hipError_t e = hipSuccess;
  if (repartition) {
    size_t tb = tmp_bytes;
    e = rocprim::partition(tmp, tb, rocprim::counting_iterator<uint32_t>(0), qclass, qorder, plan_info + 1, (size_t)n_valid, stream);
    if (e != hipSuccess) return -1;
  }
  hipLaunchKernelGGL(grid_plan_keys_kernel, dim3((nch + 255) / 256), dim3(256), 0, stream, cost, nch, n_valid, plan_info, keys);
  size_t tb = tmp_bytes;
  e = rocprim::radix_sort_pairs_desc(tmp, tb, keys, cost_sorted, ids, order, nch, 0, 32, stream);
  if (e != hipSuccess) return -1;
  hipLaunchKernelGGL(grid_plan_heavy_kernel, dim3(1), dim3(256), 0, stream, cost_sorted, cost, n_valid, n_waves, factor_override, load_factor, plan_info);
  if (g_plan_no_alone) (void)hipMemsetAsync(plan_info + 5, 0, 4, stream);
  return 0;
}

size_t grid_plan_tmp_bytes(uint32_t n_valid, uint32_t nch) {
  size_t a = 0, b = 0;
  (void)rocprim::partition(nullptr, a, rocprim::counting_iterator<uint32_t>(0), (const unsigned char *)nullptr, (uint32_t *)nullptr,
                           (uint32_t *)nullptr, (size_t)std::max<uint32_t>(n_valid, 1), (hipStream_t) nullptr);
  (void)rocprim::radix_sort_pairs_desc(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                       (size_t)std::max<uint32_t>(nch, 1), 0, 32, (hipStream_t) nullptr);
  return std::max<size_t>(std::max(a, b), 16);
}

void fill_iota(hipStream_t stream, uint32_t *v, uint32_t n) {
  hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, v, n);
}

}  // namespace ope

// ------------------------------------------------------------------------------------------
// Morton ordering of an uploaded cloud (ope_cloud_upload): key = (30-bit Morton code over the cloud's own
// bounding box, non-finite points above every code) << 32 | input index, rocPRIM radix sort, gather into
// float4 {x, y, z, input index}.  Keys are unique, so the order equals a host std::sort of the same keys.
namespace ope {

__device__ __forceinline__ uint32_t expand_bits10_dev(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

__global__ __launch_bounds__(256) void morton_key_kernel(const float *__restrict__ raw, uint32_t n, float lox, float loy, float loz,
                                                          float ivx, float ivy, float ivz, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = raw[3 * (size_t)i], y = raw[3 * (size_t)i + 1], z = raw[3 * (size_t)i + 2];
  uint32_t code;
  if (!(isfinite(x) && isfinite(y) && isfinite(z))) code = 1u << 30;
  else {
    const uint32_t qx = min(1023u, (uint32_t)fmaxf(0.f, __fmul_rn(__fsub_rn(x, lox), ivx)));
    const uint32_t qy = min(1023u, (uint32_t)fmaxf(0.f, __fmul_rn(__fsub_rn(y, loy), ivy)));
    const uint32_t qz = min(1023u, (uint32_t)fmaxf(0.f, __fmul_rn(__fsub_rn(z, loz), ivz)));
    code = expand_bits10_dev(qx) | (expand_bits10_dev(qy) << 1) | (expand_bits10_dev(qz) << 2);
  }
  keys[i] = code;
  vals[i] = i;
}

__global__ __launch_bounds__(256) void morton_gather_kernel(const float *__restrict__ raw, const uint32_t *__restrict__ order,
                                                             uint32_t n, float4 *__restrict__ xyzw, int32_t *__restrict__ perm) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t o = order[i];
  perm[i] = (int32_t)o;
  xyzw[i] = make_float4(raw[3 * (size_t)o], raw[3 * (size_t)o + 1], raw[3 * (size_t)o + 2], __int_as_float((int)o));
}

// d_raw: n*3 floats in input order (device).  Outputs: d_xyzw (n float4, sorted), d_perm (n, sorted position -> input index).
// The order is that of the keys (Morton code, then input index): a STABLE sort of the 31-bit codes with the indices as
// values (four radix passes over 8 bytes per point; round 2 sorted 64-bit code-and-index keys in eight).
hipError_t morton_order_device(hipStream_t stream, const float *d_raw, size_t n, const float lo[3], const float inv[3],
                               float4 *d_xyzw, int32_t *d_perm) {
  if (n == 0) return hipSuccess;
  uint32_t *d_keys = nullptr, *d_keys2 = nullptr, *d_vals = nullptr, *d_vals2 = nullptr;
  void *d_tmp = nullptr;
  hipError_t e = tmp_malloc(stream, (void **)&d_keys, 4 * n);
  if (e == hipSuccess) e = tmp_malloc(stream, (void **)&d_keys2, 4 * n);
  if (e == hipSuccess) e = tmp_malloc(stream, (void **)&d_vals, 4 * n);
  if (e == hipSuccess) e = tmp_malloc(stream, (void **)&d_vals2, 4 * n);
  if (e == hipSuccess) {
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(morton_key_kernel, dim3(nb), dim3(256), 0, stream, d_raw, (uint32_t)n, lo[0], lo[1], lo[2], inv[0], inv[1],
                       inv[2], d_keys, d_vals);
    size_t tb = 0;
    e = rocprim::radix_sort_pairs(nullptr, tb, d_keys, d_keys2, d_vals, d_vals2, n, 0, 31, stream);
    if (e == hipSuccess) e = tmp_malloc(stream, &d_tmp, tb);
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(d_tmp, tb, d_keys, d_keys2, d_vals, d_vals2, n, 0, 31, stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(morton_gather_kernel, dim3(nb), dim3(256), 0, stream, d_raw, d_vals2, (uint32_t)n, d_xyzw, d_perm);
      e = hipStreamSynchronize(stream);
    }
  }
  for (void *p : {(void *)d_keys, (void *)d_keys2, (void *)d_vals, (void *)d_vals2, d_tmp}) tmp_free(stream, p);
  return e;
}

// Bounding box + finite count of the points a block has seen: wave reduction, then the block's waves through LDS, then ONE atomic
// per block and word (15 625 waves each firing seven same-address atomics made these kernels take 1-3 ms; round 3).
__device__ __forceinline__ void block_bbox_commit(const float v[3], bool fin, uint32_t *__restrict__ mn, uint32_t *__restrict__ mx, uint32_t *__restrict__ n_finite) {
  __shared__ uint32_t s_lo[3][4], s_hi[3][4], s_cnt[4];
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const uint32_t u = (uint32_t)__float_as_int(v[d]);
    const uint32_t key = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    uint32_t lo = fin ? key : 0xffffffffu, hi = fin ? key : 0u;
    for (int off = 32; off >= 1; off >>= 1) {
      lo = min(lo, (uint32_t)__shfl_xor((int)lo, off, 64));
      hi = max(hi, (uint32_t)__shfl_xor((int)hi, off, 64));
    }
    if ((threadIdx.x & 63u) == 0) { s_lo[d][wave] = lo; s_hi[d][wave] = hi; }
  }
  const unsigned long long m = __ballot(fin);
  if ((threadIdx.x & 63u) == 0) s_cnt[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x < 3) {
    const int d = threadIdx.x;
    atomicMin(mn + d, min(min(s_lo[d][0], s_lo[d][1]), min(s_lo[d][2], s_lo[d][3])));
    atomicMax(mx + d, max(max(s_hi[d][0], s_hi[d][1]), max(s_hi[d][2], s_hi[d][3])));
  } else if (threadIdx.x == 3 && n_finite) {
    const uint32_t c = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (c) atomicAdd(n_finite, c);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// ope_cloud_concat: raw (original-order) xyz of [T * a ; b] and its bounding box, on the device.
// pcl::transformPointCloud (float, ((r0 x + r1 y) + r2 z) + t, non-finite points passed through) followed by
// operator+= (BuildModel regmeshpcd.cpp:203,254).
__global__ __launch_bounds__(256) void concat_kernel(CloudView a, const float *__restrict__ T_rows, CloudView b, float *__restrict__ raw,
                                                     uint32_t *__restrict__ mn, uint32_t *__restrict__ mx) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const uint32_t n = a.n + b.n;
  float x = 0.f, y = 0.f, z = 0.f;
  bool fin = false;
  if (i < n) {
    const bool from_a = i < a.n;
    const float4 p = from_a ? a.xyzw[i] : b.xyzw[i - a.n];
    const uint32_t o = (uint32_t)__float_as_int(p.w) + (from_a ? 0u : a.n);
    fin = from_a ? (i < a.n_valid) : (i - a.n < b.n_valid);
    x = p.x; y = p.y; z = p.z;
    if (from_a && fin && T_rows != nullptr) {
      x = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T_rows[0], p.x), __fmul_rn(T_rows[1], p.y)), __fmul_rn(T_rows[2], p.z)), T_rows[3]);
      y = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T_rows[4], p.x), __fmul_rn(T_rows[5], p.y)), __fmul_rn(T_rows[6], p.z)), T_rows[7]);
      z = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T_rows[8], p.x), __fmul_rn(T_rows[9], p.y)), __fmul_rn(T_rows[10], p.z)), T_rows[11]);
      fin = isfinite(x) && isfinite(y) && isfinite(z);
    }
    raw[3 * (size_t)o] = x; raw[3 * (size_t)o + 1] = y; raw[3 * (size_t)o + 2] = z;
  }
  // bounding box of the finite points: order-preserving integer keys, one atomic per block and word
  const float v[3] = {x, y, z};
  block_bbox_commit(v, fin, mn, mx, nullptr);
}

hipError_t concat_device(hipStream_t stream, const CloudView &a, const float *d_T_rows, const CloudView &b, float *d_raw, float lo[3],
                         float hi[3]) {
  const uint32_t n = a.n + b.n;
  uint32_t *d_mm = nullptr;
  hipError_t e = tmp_malloc(stream, (void **)&d_mm, 32);
  uint32_t init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0u, 0u, 0u, 0};
  if (e == hipSuccess) e = h2d_copy(stream, d_mm, init, sizeof init);
  uint32_t res[8];
  if (e == hipSuccess) {
    hipLaunchKernelGGL(concat_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a, d_T_rows, b, d_raw, d_mm, d_mm + 4);
    e = hipMemcpyAsync(res, d_mm, sizeof res, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
  }
  tmp_free(stream, d_mm);
  if (e != hipSuccess) return e;
  for (int d = 0; d < 3; ++d) {
    auto unkey = [](uint32_t k) { const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; float f; std::memcpy(&f, &u, 4); return f; };
    lo[d] = unkey(res[d]);
    hi[d] = unkey(res[4 + d]);
  }
  return hipSuccess;
}

}  // namespace ope

// ---------------------------------------------------------------------------------------------------------------------
// A new cloud from ORIGINAL indices of a device-resident one, without a trip through the host (what the filters either side
// of the path hand on: PassThrough -> StatisticalOutlierRemoval -> UniformSampling, rosinterface.cpp:212-213,
// poseestimator.cpp:141-145).  The new cloud's original order is the order of `idx`; its points are re-ordered along the
// Morton curve of ITS bounding box like an uploaded cloud's.
namespace ope {

__global__ __launch_bounds__(256) void inverse_perm_kernel(CloudView c, uint32_t *__restrict__ inv) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p < c.n) inv[(uint32_t)__float_as_int(c.xyzw[p].w)] = p;
}

__global__ __launch_bounds__(256) void select_gather_kernel(CloudView c, const uint32_t *__restrict__ inv, const int32_t *__restrict__ idx, uint32_t n_sel,
                                                            float *__restrict__ raw, uint32_t *__restrict__ mn, uint32_t *__restrict__ mx,
                                                            uint32_t *__restrict__ n_finite) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  float v[3] = {0.f, 0.f, 0.f};
  bool fin = false;
  if (j < n_sel) {
    const uint32_t p = inv[(uint32_t)idx[j]];
    const float4 q = c.xyzw[p];
    fin = p < c.n_valid;
    v[0] = q.x; v[1] = q.y; v[2] = q.z;
    raw[3 * (size_t)j] = q.x; raw[3 * (size_t)j + 1] = q.y; raw[3 * (size_t)j + 2] = q.z;
  }
  block_bbox_commit(v, fin, mn, mx, n_finite);
}

// normals of the selected points, in the new cloud's sorted order
__global__ __launch_bounds__(256) void select_normals_kernel(const float4 *__restrict__ nrm_old, const uint32_t *__restrict__ inv, const int32_t *__restrict__ idx,
                                                             const int32_t *__restrict__ perm_new, uint32_t n_sel, float4 *__restrict__ nrm_new) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p < n_sel) nrm_new[p] = nrm_old[inv[(uint32_t)idx[perm_new[p]]]];
}

// d_idx: n_sel ORIGINAL indices of `cloud`, on the device.  Synchronises the stream (the bounding box comes back to the host).
int select_cloud_device(ope_ctx *ctx, const ope_cloud *cloud, const int32_t *d_idx, size_t n_sel, ope_cloud **out) {
  *out = nullptr;
  ope_cloud *c = new ope_cloud();
  c->ctx = ctx;
  c->n = n_sel;
  c->host_valid = false;
  uint32_t *d_inv = nullptr, *d_mm = nullptr;
  float *d_raw = nullptr;
  int32_t *d_perm = nullptr;
  const CloudView cv = cloud->view();
  hipError_t e = hipMalloc((void **)&c->d_xyzw, sizeof(float4) * std::max<size_t>(n_sel, 1));
  if (e == hipSuccess && n_sel) e = tmp_malloc(ctx->stream, (void **)&d_inv, 4 * std::max<size_t>(cloud->n, 1));
  if (e == hipSuccess && n_sel) e = tmp_malloc(ctx->stream, (void **)&d_raw, 12 * n_sel);
  if (e == hipSuccess && n_sel) e = tmp_malloc(ctx->stream, (void **)&d_perm, 4 * n_sel);
  if (e == hipSuccess && n_sel) e = tmp_malloc(ctx->stream, (void **)&d_mm, 48);
  uint32_t res[12] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0u, 0u, 0u, 0, 0u, 0, 0, 0};
  if (e == hipSuccess && n_sel) {
    e = h2d_copy(ctx->stream, d_mm, res, sizeof res);
    hipLaunchKernelGGL(inverse_perm_kernel, dim3((unsigned)((cloud->n + 255) / 256)), dim3(256), 0, ctx->stream, cv, d_inv);
    hipLaunchKernelGGL(select_gather_kernel, dim3((unsigned)((n_sel + 255) / 256)), dim3(256), 0, ctx->stream, cv, d_inv, d_idx, (uint32_t)n_sel, d_raw,
                       d_mm, d_mm + 4, d_mm + 8);
    if (e == hipSuccess) e = hipMemcpyAsync(res, d_mm, sizeof res, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  if (e == hipSuccess && n_sel) {
    auto unkey = [](uint32_t k) { const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; float f; std::memcpy(&f, &u, 4); return f; };
    c->n_valid = res[8];
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, inv3[3];
    if (c->n_valid > 0)
      for (int d = 0; d < 3; ++d) { lo[d] = unkey(res[d]); hi[d] = unkey(res[4 + d]); }
    std::memcpy(c->bb_lo, lo, sizeof lo);
    std::memcpy(c->bb_hi, hi, sizeof hi);
    for (int d = 0; d < 3; ++d) inv3[d] = (hi[d] > lo[d]) ? 1023.999f / (hi[d] - lo[d]) : 0.f;
    e = morton_order_device(ctx->stream, d_raw, n_sel, lo, inv3, c->d_xyzw, d_perm);
    if (e == hipSuccess && cloud->d_nrm) {
      e = hipMalloc((void **)&c->d_nrm, sizeof(float4) * n_sel);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(select_normals_kernel, dim3((unsigned)((n_sel + 255) / 256)), dim3(256), 0, ctx->stream, cloud->d_nrm, d_inv, d_idx, d_perm,
                           (uint32_t)n_sel, c->d_nrm);
        e = hipStreamSynchronize(ctx->stream);
      }
    }
  }
  for (void *p : {(void *)d_inv, (void *)d_raw, (void *)d_perm, (void *)d_mm}) tmp_free(ctx->stream, p);
  if (e != hipSuccess) {
    ope_cloud_free(c);
    return set_err(ctx, OPE_EHIP, std::string("cloud selection: ") + hipGetErrorString(e));
  }
  *out = c;
  return OPE_OK;
}

// ---- the same for ORDER-PRESERVING filters (NaN removal, pass-through, outlier removal: PCL emits the survivors in input
// order): the survivors keep their relative order along the parent's Morton curve, so nothing is sorted again — two prefix
// sums (the survivors' rank in input order = their new original index; their rank in sorted order = their new position)
// and one scatter.  keep: one byte per ORIGINAL index.  d_idx_out (optional, device, room for every point): the survivors'
// original indices in input order.
__global__ __launch_bounds__(256) void keep_flags_kernel(CloudView c, const unsigned char *__restrict__ keep, uint32_t *__restrict__ f_orig,
                                                         uint32_t *__restrict__ f_sorted) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i > c.n) return;
  if (i == c.n) { f_orig[i] = 0u; f_sorted[i] = 0u; return; }
  f_orig[i] = keep[i] ? 1u : 0u;
  f_sorted[i] = keep[(uint32_t)__float_as_int(c.xyzw[i].w)] ? 1u : 0u;
}
__global__ __launch_bounds__(256) void compact_kernel(CloudView c, const uint32_t *__restrict__ f_sorted, const uint32_t *__restrict__ r_orig,
                                                      const uint32_t *__restrict__ r_sorted, float4 *__restrict__ xyzw_out, float4 *__restrict__ nrm_out,
                                                      int32_t *__restrict__ idx_out, uint32_t *__restrict__ mn, uint32_t *__restrict__ mx,
                                                      uint32_t *__restrict__ n_finite) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  float v[3] = {0.f, 0.f, 0.f};
  bool fin = false;
  if (p < c.n && f_sorted[p]) {
    const float4 q = c.xyzw[p];
    const uint32_t o = (uint32_t)__float_as_int(q.w), o_new = r_orig[o], dst = r_sorted[p];
    xyzw_out[dst] = make_float4(q.x, q.y, q.z, __int_as_float((int)o_new));
    if (nrm_out) nrm_out[dst] = c.nrm[p];
    if (idx_out) idx_out[o_new] = (int32_t)o;
    fin = p < c.n_valid;
    v[0] = q.x; v[1] = q.y; v[2] = q.z;
  }
  block_bbox_commit(v, fin, mn, mx, n_finite);
}

int compact_cloud_device(ope_ctx *ctx, const ope_cloud *cloud, const unsigned char *d_keep, ope_cloud **out, int32_t *d_idx_out, size_t *n_out) {
  *out = nullptr;
  if (n_out) *n_out = 0;
  const size_t n = cloud->n;
  hipStream_t st = ctx->stream;
  uint32_t *d_fo = nullptr, *d_fs = nullptr, *d_ro = nullptr, *d_rs = nullptr, *d_mm = nullptr;
  float4 *d_x = nullptr, *d_n = nullptr;
  void *d_tmp = nullptr;
  size_t tb = 0;
  uint32_t res[12] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0u, 0u, 0u, 0, 0u, 0, 0, 0};
  uint32_t count = 0;
  hipError_t e = tmp_malloc(st, (void **)&d_fo, 4 * (n + 1));
  if (e == hipSuccess) e = tmp_malloc(st, (void **)&d_fs, 4 * (n + 1));
  if (e == hipSuccess) e = tmp_malloc(st, (void **)&d_ro, 4 * (n + 1));
  if (e == hipSuccess) e = tmp_malloc(st, (void **)&d_rs, 4 * (n + 1));
  if (e == hipSuccess) e = tmp_malloc(st, (void **)&d_mm, 48);
  if (e == hipSuccess) e = rocprim::exclusive_scan(nullptr, tb, d_fo, d_ro, 0u, n + 1, rocprim::plus<uint32_t>(), st);
  if (e == hipSuccess) e = tmp_malloc(st, &d_tmp, tb);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(keep_flags_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, st, cloud->view(), d_keep, d_fo, d_fs);
    size_t t1 = tb;
    e = rocprim::exclusive_scan(d_tmp, t1, d_fo, d_ro, 0u, n + 1, rocprim::plus<uint32_t>(), st);
    size_t t2 = tb;
    if (e == hipSuccess) e = rocprim::exclusive_scan(d_tmp, t2, d_fs, d_rs, 0u, n + 1, rocprim::plus<uint32_t>(), st);
    if (e == hipSuccess) e = hipMemcpyAsync(&count, d_ro + n, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
  }
  ope_cloud *c = nullptr;
  if (e == hipSuccess) {
    // (the new cloud's own buffers outlive the call: plain allocations)
    c = new ope_cloud();
    c->ctx = ctx;
    c->n = count;
    c->host_valid = false;
    e = hipMalloc((void **)&d_x, sizeof(float4) * std::max<size_t>(count, 1));
    if (e == hipSuccess && cloud->d_nrm) e = hipMalloc((void **)&d_n, sizeof(float4) * std::max<size_t>(count, 1));
    c->d_xyzw = d_x;
    c->d_nrm = d_n;
  }
  if (e == hipSuccess && n) {
    e = h2d_copy(st, d_mm, res, sizeof res);
    hipLaunchKernelGGL(compact_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, cloud->view(), d_fs, d_ro, d_rs, d_x, d_n, d_idx_out, d_mm, d_mm + 4,
                       d_mm + 8);
    if (e == hipSuccess) e = hipMemcpyAsync(res, d_mm, sizeof res, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
  }
  for (void *p : {(void *)d_fo, (void *)d_fs, (void *)d_ro, (void *)d_rs, (void *)d_mm, d_tmp}) tmp_free(st, p);
  if (e != hipSuccess) {
    if (c) ope_cloud_free(c);
    return set_err(ctx, OPE_EHIP, std::string("cloud compaction: ") + hipGetErrorString(e));
  }
  auto unkey = [](uint32_t k) { const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; float f; std::memcpy(&f, &u, 4); return f; };
  c->n_valid = res[8];
  if (c->n_valid > 0)
    for (int d = 0; d < 3; ++d) { c->bb_lo[d] = unkey(res[d]); c->bb_hi[d] = unkey(res[4 + d]); }
  *out = c;
  if (n_out) *n_out = count;
  return OPE_OK;
}

}  // namespace ope

using namespace ope;

// the survivors' ORIGINAL indices, ascending voxel key, left on the device (*d_out_ret, caller frees); count on the host
static int uniform_sampling_dev(ope_ctx *ctx, const ope_cloud *cloud, float leaf, int32_t **d_out_ret, unsigned int *count_ret) {
  *d_out_ret = nullptr;
  *count_ret = 0;
  const size_t n = cloud->n;
  if (n == 0 || cloud->n_valid == 0) return OPE_OK;
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  const float inv = 1.0f / leaf;
  long long min_b[3], div_b[3];
  for (int d = 0; d < 3; ++d) {
    min_b[d] = (long long)std::floor(cloud->bb_lo[d] * inv);
    div_b[d] = (long long)std::floor(cloud->bb_hi[d] * inv) - min_b[d] + 1;
  }
  // PCL's own leaf index is a 32-bit int: "Leaf size is too small for the input dataset"
  if ((double)div_b[0] * (double)div_b[1] * (double)div_b[2] >= 2147483648.0)
    return set_err(ctx, OPE_EINVAL, "ope_uniform_sampling: leaf size too small for the input dataset (voxel index overflows)");
  unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
  uint32_t *d_vals = nullptr, *d_vals2 = nullptr;
  int32_t *d_win = nullptr, *d_out = nullptr;
  unsigned char *d_flags = nullptr;
  unsigned int *d_count = nullptr;
  void *d_tmp = nullptr;
  hipError_t e = tmp_malloc(ctx->stream, (void **)&d_keys, 8 * n);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_keys2, 8 * n);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_vals, 4 * n);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_vals2, 4 * n);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_win, 4 * n);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_out, 4 * n);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_flags, n);
  if (e == hipSuccess) e = tmp_malloc(ctx->stream, (void **)&d_count, 4);
  unsigned int count = 0;
  if (e == hipSuccess) {
    const unsigned nb = (unsigned)((n + 255) / 256);
    // key = voxel ‖ original index, packed: index bits so that no index is all ones (the keys of non-finite points, all ones, stay
    // last), voxel bits for the table's size; the radix sort runs over idx_bits + vox_bits bits instead of 64 (C3's cluster: 39)
    int idx_bits = 1, vox_bits = 1;
    while (((size_t)1 << idx_bits) <= n) ++idx_bits;
    while (((unsigned long long)1 << vox_bits) <= (unsigned long long)(div_b[0] * div_b[1] * div_b[2])) ++vox_bits;
    const int key_bits = std::min(64, idx_bits + vox_bits);
    hipLaunchKernelGGL(voxel_key_kernel, dim3(nb), dim3(256), 0, ctx->stream, cloud->view(), inv, (int)min_b[0], (int)min_b[1],
                       (int)min_b[2], (unsigned)div_b[0], (unsigned)(div_b[0] * div_b[1]), idx_bits, d_keys, d_vals);
    size_t tmp_sort = 0, tmp_sel = 0;
    e = rocprim::radix_sort_pairs(nullptr, tmp_sort, d_keys, d_keys2, d_vals, d_vals2, n, 0, 64, ctx->stream);
    if (e == hipSuccess)
      e = rocprim::select(nullptr, tmp_sel, d_win, d_flags, d_out, d_count, n, ctx->stream);
    const size_t tmp_bytes = std::max(tmp_sort, tmp_sel);
    if (e == hipSuccess) e = tmp_malloc(ctx->stream, &d_tmp, tmp_bytes);
    size_t tb = tmp_bytes;
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(d_tmp, tb, d_keys, d_keys2, d_vals, d_vals2, n, 0, key_bits, ctx->stream);
    if (e == hipSuccess) {
      // (the unsorted keys are done with: their buffer holds the voxels' slots)
      e = hipMemsetAsync(d_keys, 0xff, 8 * n, ctx->stream);
      hipLaunchKernelGGL(voxel_min_kernel, dim3(nb), dim3(256), 0, ctx->stream, cloud->view(), inv, d_keys2, d_vals2, (uint32_t)cloud->n_valid, idx_bits, d_keys);
      hipLaunchKernelGGL(voxel_pick_kernel, dim3(nb), dim3(256), 0, ctx->stream, cloud->view(), d_keys2, d_vals2, (uint32_t)cloud->n_valid, idx_bits, d_keys,
                         d_win, d_flags);
      tb = tmp_bytes;
      e = rocprim::select(d_tmp, tb, d_win, d_flags, d_out, d_count, n, ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&count, d_count, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  for (void *p : {(void *)d_keys, (void *)d_keys2, (void *)d_vals, (void *)d_vals2, (void *)d_win, (void *)d_flags, (void *)d_count, d_tmp}) tmp_free(ctx->stream, p);
  if (e != hipSuccess) {
    tmp_free(ctx->stream, d_out);
    return set_err(ctx, OPE_EHIP, std::string("ope_uniform_sampling: ") + hipGetErrorString(e));
  }
  *d_out_ret = d_out;
  *count_ret = count;
  return OPE_OK;
}

extern "C" int ope_uniform_sampling(ope_ctx *ctx, const ope_cloud *cloud, float leaf, int32_t *out_idx, size_t *n_out) {
  if (!ctx || !cloud || !out_idx || !n_out || !(leaf > 0)) return set_err(ctx, OPE_EINVAL, "ope_uniform_sampling: bad argument");
  *n_out = 0;
  int32_t *d_out = nullptr;
  unsigned int count = 0;
  const int rc = uniform_sampling_dev(ctx, cloud, leaf, &d_out, &count);
  if (rc != OPE_OK) return rc;
  hipError_t e = hipSuccess;
  if (count) e = hipMemcpy(out_idx, d_out, 4 * (size_t)count, hipMemcpyDeviceToHost);
  tmp_free(ctx->stream, d_out);
  if (e != hipSuccess) return set_err(ctx, OPE_EHIP, std::string("ope_uniform_sampling: ") + hipGetErrorString(e));
  OPE_DUMP_HASH("uniform_sampling idx", out_idx, 4 * (size_t)count, false);
  *n_out = count;
  return OPE_OK;
}

extern "C" int ope_uniform_sampling_cloud(ope_ctx *ctx, const ope_cloud *cloud, float leaf, ope_cloud **out, int32_t *out_idx, size_t *n_out) {
  if (!ctx || !cloud || !out || !(leaf > 0)) return set_err(ctx, OPE_EINVAL, "ope_uniform_sampling_cloud: bad argument");
  *out = nullptr;
  if (n_out) *n_out = 0;
  int32_t *d_out = nullptr;
  unsigned int count = 0;
  int rc = uniform_sampling_dev(ctx, cloud, leaf, &d_out, &count);
  if (rc != OPE_OK) return rc;
  hipError_t e = hipSuccess;
  if (count && out_idx) e = hipMemcpyAsync(out_idx, d_out, 4 * (size_t)count, hipMemcpyDeviceToHost, ctx->stream);
  rc = e == hipSuccess ? select_cloud_device(ctx, cloud, d_out, count, out) : set_err(ctx, OPE_EHIP, std::string("ope_uniform_sampling_cloud: ") + hipGetErrorString(e));
  tmp_free(ctx->stream, d_out);
  if (rc == OPE_OK && n_out) *n_out = count;
  return rc;
}

extern "C" int ope_cloud_select(ope_ctx *ctx, const ope_cloud *cloud, const int32_t *idx, size_t n, ope_cloud **out) {
  if (!ctx || !cloud || !out || (n && !idx) || n > (size_t)0x7fffffff) return set_err(ctx, OPE_EINVAL, "ope_cloud_select: bad argument");
  *out = nullptr;
  for (size_t j = 0; j < n; ++j)
    if (idx[j] < 0 || (size_t)idx[j] >= cloud->n) return set_err(ctx, OPE_EINVAL, "ope_cloud_select: index out of range");
  OPE_HIP(ctx, hipSetDevice(ctx->device));
  int32_t *d_idx = nullptr;
  if (n) {
    OPE_HIP(ctx, tmp_malloc(ctx->stream, (void **)&d_idx, 4 * n));
    const hipError_t e = h2d_copy(ctx->stream, d_idx, idx, 4 * n);
    if (e != hipSuccess) { tmp_free(ctx->stream, d_idx); return set_err(ctx, OPE_EHIP, std::string("ope_cloud_select: ") + hipGetErrorString(e)); }
  }
  const int rc = select_cloud_device(ctx, cloud, d_idx, n, out);
  tmp_free(ctx->stream, d_idx);
  return rc;
}
