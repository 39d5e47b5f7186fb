// grid_build.hip — the "radix-bucketed" side of the target index (north_star: "brute-force/radix-bucketed nearest-
// neighbour ... with coalesced HBM loads of packed xyz"): a uniform grid over the target's bounding box.
//
// The OBB tree (bvh_build_device.hip) answers ANY nearest-neighbour query exactly in ~15 dependent fetches; this grid
// answers the common ICP query — a scene point whose previous match is a fraction of a millimetre away — in two: the
// cells the ball |x - q| <= |q - previous match| overlaps are a handful of contiguous runs of a cell-sorted copy of the
// points (icp_kernels.hip, GRID instantiation).  Layout:
//   cell id  = (iz * dim_y + iy) * dim_x + ix,  i = floor((p - lo) * inv_cell)   (x fastest: an x-run of cells is one run of points)
//   gpts     : the target points sorted by cell id (radix sort), float4 {x, y, z, ORIGINAL index}, + kPtsPad zeroed entries
//   gnrm     : their normals in the same order (if the target has any)
//   cell_start[n_cells + 1] : first sorted position of every cell (exclusive scan of the cell histogram)
//   gpos_of_bvhpos[n]       : sorted position of the point at BVH position p (a tree-walk result becomes a grid hint)
// Cell size: chosen so that an occupied cell holds ~kGridTargetFill points (two refinement passes over the measured
// occupancy), and grown until the table has at most kGridMaxCells cells (it must stay L2-resident: 4 B per cell).
#include <algorithm>
#include <cmath>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "ope_internal.hpp"

namespace ope {

constexpr double kGridDefaultFill = 6.0;
constexpr uint32_t kGridDefaultMaxCells = 1u << 20;   // 4 MB of cell starts at most

__device__ __forceinline__ uint32_t grid_cell_of(const GridView &g, float x, float y, float z) {
  // the same expression, in the same order, as the query side uses for its cell ranges (monotone in x, y, z)
  const int ix = min(max((int)floorf((x - g.lo[0]) * g.inv), 0), g.dim[0] - 1);
  const int iy = min(max((int)floorf((y - g.lo[1]) * g.inv), 0), g.dim[1] - 1);
  const int iz = min(max((int)floorf((z - g.lo[2]) * g.inv), 0), g.dim[2] - 1);
  return ((uint32_t)iz * (uint32_t)g.dim[1] + (uint32_t)iy) * (uint32_t)g.dim[0] + (uint32_t)ix;
}

__global__ __launch_bounds__(256) void grid_keys_kernel(GridView g, const float4 *__restrict__ pts, uint32_t n, uint32_t *__restrict__ keys,
                                                         uint32_t *__restrict__ vals) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  keys[i] = grid_cell_of(g, p.x, p.y, p.z);
  vals[i] = i;
}

__global__ __launch_bounds__(256) void grid_count_occupied_kernel(const uint32_t *__restrict__ sorted_keys, uint32_t n, uint32_t *__restrict__ count) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const bool head = i < n && (i == 0 || sorted_keys[i] != sorted_keys[i - 1]);
  const unsigned long long m = __ballot(head);
  if ((threadIdx.x & 63u) == 0 && m) atomicAdd(count, (uint32_t)__popcll(m));
}

__global__ __launch_bounds__(256) void grid_scatter_kernel(const uint32_t *__restrict__ sorted_keys, const uint32_t *__restrict__ sorted_vals,
                                                            const float4 *__restrict__ pts, const float4 *__restrict__ nrm, uint32_t n,
                                                            float4 *__restrict__ gpts, float4 *__restrict__ gnrm,
                                                            uint32_t *__restrict__ gpos_of_bvhpos, uint32_t *__restrict__ hist) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t b = sorted_vals[j];
  gpts[j] = pts[b];
  if (nrm) gnrm[j] = nrm[b];
  gpos_of_bvhpos[b] = j;
  atomicAdd(hist + sorted_keys[j], 1u);
}

// One pass at a given cell size: keys + radix sort; returns the number of occupied cells.
static hipError_t grid_sort(hipStream_t stream, const GridView &g, const float4 *d_pts, uint32_t n, uint32_t *d_keys, uint32_t *d_keys2,
                            uint32_t *d_vals, uint32_t *d_vals2, uint32_t *d_count, void *d_tmp, size_t tmp_bytes, uint32_t *occupied) {
  const unsigned nb = (n + 255) / 256;
  hipLaunchKernelGGL(grid_keys_kernel, dim3(nb), dim3(256), 0, stream, g, d_pts, n, d_keys, d_vals);
  const uint32_t n_cells = (uint32_t)g.dim[0] * (uint32_t)g.dim[1] * (uint32_t)g.dim[2];
  int bits = 1;
  while ((1ull << bits) < n_cells) ++bits;
  size_t tb = tmp_bytes;
  hipError_t e = rocprim::radix_sort_pairs(d_tmp, tb, d_keys, d_keys2, d_vals, d_vals2, n, 0, bits, stream);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(d_count, 0, 4, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(grid_count_occupied_kernel, dim3(nb), dim3(256), 0, stream, d_keys2, n, d_count);
  e = hipMemcpyAsync(occupied, d_count, 4, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  return e;
}

static void grid_dims(GridView &g, const float lo[3], const float hi[3], double cell, uint32_t kGridMaxCells) {
  for (;;) {
    g.inv = (float)(1.0 / cell);
    unsigned long long cells = 1;
    for (int d = 0; d < 3; ++d) {
      g.lo[d] = lo[d];
      g.dim[d] = std::max(1, (int)std::floor(((double)hi[d] - (double)lo[d]) / cell) + 1);
      cells *= (unsigned long long)g.dim[d];
    }
    if (cells <= kGridMaxCells) return;
    cell *= 1.26;   // x2 in volume
  }
}

// d_pts: the index's points in BVH order (n entries, w = original index), d_nrm optional.
hipError_t build_grid_device(hipStream_t stream, const float4 *d_pts, const float4 *d_nrm, size_t n_, const float bb_lo[3],
                             const float bb_hi[3], double fill_target, uint32_t max_cells, GridView *out, float4 **out_gpts, float4 **out_gnrm, uint32_t **out_cell_start,
                             uint32_t **out_gpos) {
  const uint32_t n = (uint32_t)n_;
  *out_gpts = nullptr; *out_gnrm = nullptr; *out_cell_start = nullptr; *out_gpos = nullptr;
  const double kGridTargetFill = fill_target > 0 ? fill_target : kGridDefaultFill;
  const uint32_t kGridMaxCells = max_cells > 0 ? max_cells : kGridDefaultMaxCells;
  GridView g{};
  // first guess: a surface sample — n points over an area of about (bbox diagonal)^2 / 3
  const double ex = (double)bb_hi[0] - bb_lo[0], ey = (double)bb_hi[1] - bb_lo[1], ez = (double)bb_hi[2] - bb_lo[2];
  const double diag2 = ex * ex + ey * ey + ez * ez;
  double cell = std::sqrt(std::max(diag2, 1e-12) / 3.0 * kGridTargetFill / std::max<double>(n, 1));
  cell = std::max(cell, 1e-7 * std::sqrt(std::max(diag2, 1e-12)) + 1e-12);
  uint32_t *d_keys = nullptr, *d_keys2 = nullptr, *d_vals = nullptr, *d_vals2 = nullptr, *d_count = nullptr;
  void *d_tmp = nullptr;
  size_t tmp_bytes = 0;
  hipError_t e = hipMalloc((void **)&d_keys, 4 * (size_t)n);
  if (e == hipSuccess) e = hipMalloc((void **)&d_keys2, 4 * (size_t)n);
  if (e == hipSuccess) e = hipMalloc((void **)&d_vals, 4 * (size_t)n);
  if (e == hipSuccess) e = hipMalloc((void **)&d_vals2, 4 * (size_t)n);
  if (e == hipSuccess) e = hipMalloc((void **)&d_count, 4);
  if (e == hipSuccess) e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys, d_keys2, d_vals, d_vals2, n, 0, 32, stream);
  size_t scan_bytes = 0;
  if (e == hipSuccess)
    e = rocprim::exclusive_scan(nullptr, scan_bytes, d_keys, d_keys2, 0u, (size_t)kGridMaxCells + 1, rocprim::plus<uint32_t>(), stream);
  tmp_bytes = std::max(tmp_bytes, scan_bytes);
  if (e == hipSuccess) e = hipMalloc(&d_tmp, std::max<size_t>(tmp_bytes, 16));
  uint32_t occupied = 0;
  for (int pass = 0; e == hipSuccess && pass < 3; ++pass) {
    grid_dims(g, bb_lo, bb_hi, cell, kGridMaxCells);
    e = grid_sort(stream, g, d_pts, n, d_keys, d_keys2, d_vals, d_vals2, d_count, d_tmp, tmp_bytes, &occupied);
    if (e != hipSuccess || pass == 2) break;
    const double fill = (double)n / std::max<uint32_t>(occupied, 1);
    if (fill > 0.6 * kGridTargetFill && fill < 1.7 * kGridTargetFill) break;
    // occupied cells of a surface scale with 1 / cell^2, of a volume with 1 / cell^3: take the milder exponent
    const double want = std::sqrt(kGridTargetFill / fill);
    const double next = (double)(1.0 / g.inv) * std::min(4.0, std::max(0.25, want));
    GridView t{};
    grid_dims(t, bb_lo, bb_hi, next, kGridMaxCells);
    if (t.inv == g.inv) break;   // the table-size cap decides
    cell = next;
  }
  uint32_t *d_cell_start = nullptr, *d_hist = nullptr, *d_gpos = nullptr;
  float4 *d_gpts = nullptr, *d_gnrm = nullptr;
  const size_t n_cells = (size_t)g.dim[0] * g.dim[1] * g.dim[2];
  if (e == hipSuccess) e = hipMalloc((void **)&d_cell_start, 4 * (n_cells + 1));
  if (e == hipSuccess) e = hipMalloc((void **)&d_hist, 4 * (n_cells + 1));
  if (e == hipSuccess) e = hipMemsetAsync(d_hist, 0, 4 * (n_cells + 1), stream);
  if (e == hipSuccess) e = hipMalloc((void **)&d_gpos, 4 * (size_t)n);
  if (e == hipSuccess) e = hipMalloc((void **)&d_gpts, sizeof(float4) * ((size_t)n + kPtsPad));
  if (e == hipSuccess) e = hipMemsetAsync(d_gpts + n, 0, sizeof(float4) * kPtsPad, stream);
  if (e == hipSuccess && d_nrm) e = hipMalloc((void **)&d_gnrm, sizeof(float4) * (size_t)n);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(grid_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_keys2, d_vals2, d_pts, d_nrm, n, d_gpts, d_gnrm,
                       d_gpos, d_hist);
    size_t tb = tmp_bytes;
    e = rocprim::exclusive_scan(d_tmp, tb, d_hist, d_cell_start, 0u, n_cells + 1, rocprim::plus<uint32_t>(), stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  for (void *p : {(void *)d_keys, (void *)d_keys2, (void *)d_vals, (void *)d_vals2, (void *)d_count, d_tmp, (void *)d_hist})
    if (p) (void)hipFree(p);
  if (e != hipSuccess) {
    for (void *p : {(void *)d_cell_start, (void *)d_gpos, (void *)d_gpts, (void *)d_gnrm})
      if (p) (void)hipFree(p);
    return e;
  }
  g.n_cells = (uint32_t)n_cells;
  g.occupied = occupied;
  *out = g;
  *out_gpts = d_gpts; *out_gnrm = d_gnrm; *out_cell_start = d_cell_start; *out_gpos = d_gpos;
  return hipSuccess;
}

// ---- when a run moves from the grid kernel to the tree kernel: the previous matches the grid kernel kept as positions in
// the cell-sorted points become the tree kernel's start leaves, so that its first launch starts from every query's previous
// match instead of the root (round 2 left the hints of grid-answered queries untouched: three launches of 250 us followed
// every switch, the driver's window began with them).
__global__ __launch_bounds__(256) void invert_gpos_kernel(const uint32_t *__restrict__ gpos_of_bvhpos, uint32_t n, uint32_t *__restrict__ bvhpos_of_gpos) {
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b < n) bvhpos_of_gpos[gpos_of_bvhpos[b]] = b;
}
__global__ __launch_bounds__(256) void ghint_to_leaf_kernel(const uint32_t *__restrict__ ghint, const uint32_t *__restrict__ bvhpos_of_gpos, uint32_t nq,
                                                            uint32_t n_t, int depth, uint32_t *__restrict__ hint) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nq) return;
  const uint32_t g = ghint[i];
  if (g == 0u || g > n_t) return;
  const unsigned long long b = bvhpos_of_gpos[g - 1u];
  // leaf j owns the positions [j n >> D, (j + 1) n >> D)
  unsigned long long j = (b << depth) / n_t;
  while ((((j + 1ull) * n_t) >> depth) <= b) ++j;
  while (j > 0ull && ((j * n_t) >> depth) > b) --j;
  hint[i] = (1u << depth) + (uint32_t)j;
}
// d_inv: scratch of n_t words
void grid_hints_to_leaves(hipStream_t stream, const uint32_t *ghint, const uint32_t *gpos_of_bvhpos, uint32_t nq, uint32_t n_t, int depth, uint32_t *hint,
                          uint32_t *d_inv) {
  if (nq == 0 || n_t == 0) return;
  hipLaunchKernelGGL(invert_gpos_kernel, dim3((n_t + 255) / 256), dim3(256), 0, stream, gpos_of_bvhpos, n_t, d_inv);
  hipLaunchKernelGGL(ghint_to_leaf_kernel, dim3((nq + 255) / 256), dim3(256), 0, stream, ghint, d_inv, nq, n_t, depth, hint);
}

// ---- before the first launch of a run: how many source points, under the initial guess, lie further than `margin` outside
// the target's bounding box?  A lower bound on the number of queries the 27-cell scan can never answer (clutter), taken
// before anything has been searched: a run over a cluttered frame starts on the tree kernel instead of moving there a few
// launches later (every move costs the kernel taken over a few launches with cold plans: round 3 measured ~30 us on each of
// the four launches that followed it).
struct Rows12 { float r[12]; };
__global__ __launch_bounds__(256) void count_outside_kernel(CloudView src, Rows12 F, float lox, float loy, float loz, float hix, float hiy, float hiz,
                                                            uint32_t *__restrict__ count) {
  uint32_t c = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < src.n_valid; i += gridDim.x * 256) {
    const float4 s = src.xyzw[i];
    const float x = F.r[0] * s.x + F.r[1] * s.y + F.r[2] * s.z + F.r[3];
    const float y = F.r[4] * s.x + F.r[5] * s.y + F.r[6] * s.z + F.r[7];
    const float z = F.r[8] * s.x + F.r[9] * s.y + F.r[10] * s.z + F.r[11];
    c += (x < lox || x > hix || y < loy || y > hiy || z < loz || z > hiz) ? 1u : 0u;
  }
  for (int off = 32; off >= 1; off >>= 1) c += (uint32_t)__shfl_xor((int)c, off, 64);
  if ((threadIdx.x & 63u) == 0 && c) atomicAdd(count, c);
}
void grid_count_outside(hipStream_t stream, const CloudView &src, const float rows[12], const float lo[3], const float hi[3], float margin, uint32_t *d_count) {
  Rows12 F;
  for (int k = 0; k < 12; ++k) F.r[k] = rows[k];
  (void)hipMemsetAsync(d_count, 0, 4, stream);
  hipLaunchKernelGGL(count_outside_kernel, dim3(std::min<uint32_t>((src.n_valid + 255) / 256, 1024)), dim3(256), 0, stream, src, F, lo[0] - margin, lo[1] - margin,
                     lo[2] - margin, hi[0] + margin, hi[1] + margin, hi[2] + margin, d_count);
}

}  // namespace ope
