// lm.hip — OPE_EST_POINT_TO_PLANE_LM: the estimator BuildModel installs,
// pcl::registration::TransformationEstimationPointToPlane (BuildModel/src/regmeshpcd.cpp:162,193): Levenberg-Marquardt
// over WarpPointRigid6D (tx, ty, tz, qx, qy, qz) on the point-to-plane residuals of the current correspondences, run by
// Eigen's (unsupported) LevenbergMarquardt on a forward-difference Jacobian, all in float inside ICP.
//
// The residual f_i(x) = (W(x) s_i - t_i) . n_i is LINEAR in the twelve entries of the warp matrix W(x) = [R | t]:
//   f_i = f_i(I) + phi_i . (W(x) - I),   phi_i = n_i (x) [s_i ; 1]   (12 numbers per correspondence),
// so every quantity the minimiser asks for is a quadratic form of D = W(x) - I over sums that do not depend on x:
//   A = sum phi phi^T (78),  g = sum f_i(I) phi_i (12),  c0 = sum f_i(I)^2 (1)      -- the 91 sums of lm_stats_kernel
//   f^T f = c0 + 2 g.D + D^T A D,   J_j = Phi u_j with u_j = (W(x + h_j e_j) - W(x)) / h_j,
//   J^T J = u^T A u,   J^T f = u^T (g + A D).
// One pass over the correspondences per ICP iteration (round 2: one pass per functor evaluation, 25-50 per iteration, each
// with a read-back that synchronised the host), the sums all-reduced once in sharded runs, and the whole minimisation —
// MINPACK's lmder logic as Eigen transcribes it (minimizeOneStep, lmpar2, qrsolv; factor 100, maxfev 400, ftol = xtol =
// sqrt(FLT_EPSILON), gtol 0; NumericalDiff: h = sqrt(FLT_EPSILON) |x_j|, or sqrt(FLT_EPSILON) at 0) in double on the 6 x 6
// quantities — runs in one lane of lm_solve_update_kernel, followed in the same launch by the ICP update step.  Nothing
// synchronises the host.
// What is kept of the reference's float arithmetic: the warp matrices themselves (quaternion normalisation and
// toRotationMatrix in float, lm_warp_matrix) and the forward-difference steps; the residuals are formed from those float
// matrices in exact arithmetic instead of float, i.e. without the reference's per-residual rounding (1e-7 relative).  R and
// Q^T f, which Eigen takes from a column-pivoted Householder QR of the m x 6 float Jacobian, come from the Cholesky factor of
// J^T J: R^T R = J^T J, Q^T f = R^-T J^T f — the same numbers in exact arithmetic.  What float rounding can change in the
// result is measured in tests/test_oracle_lm.py (oracle in float vs double); this path lands with the double one.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

#include "bvh_traverse.hpp"

namespace ope {

constexpr int kLmStats = 91;   // 78 (upper triangle of A, row by row) + 12 (g) + 1 (c0)

// WarpPointRigid6D::setParam in float (Eigen::Quaternionf::normalize / toRotationMatrix)
__host__ __device__ static inline void lm_warp_matrix(const float x[6], float M[12]) {
  float qx = x[3], qy = x[4], qz = x[5];
  float qw = (float)sqrt((double)(1.0f - (qx * qx + qy * qy + qz * qz)));
  const float nn = (float)sqrt((double)(qw * qw + qx * qx + qy * qy + qz * qz));
  qw /= nn; qx /= nn; qy /= nn; qz /= nn;
  const float tx = 2.0f * qx, ty = 2.0f * qy, tz = 2.0f * qz;
  const float twx = tx * qw, twy = ty * qw, twz = tz * qw;
  const float txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  M[0] = 1.0f - (tyy + tzz); M[1] = txy - twz; M[2] = txz + twy; M[3] = x[0];
  M[4] = txy + twz; M[5] = 1.0f - (txx + tzz); M[6] = tyz - twx; M[7] = x[1];
  M[8] = txz - twy; M[9] = tyz + twx; M[10] = 1.0f - (txx + tyy); M[11] = x[2];
}

// corr_pos: per sorted source position the BVH position of its match, or -1 (what the accumulate kernels store in
// corr_match while the LM estimator is selected).  The 91 sums are added into `stats` (zeroed by the solve kernel).
__global__ __launch_bounds__(256) void lm_stats_kernel(CloudView src, BvhView tgt, const IcpState *__restrict__ st,
                                                       const int32_t *__restrict__ corr_pos, double *__restrict__ stats) {
  if (st->done) return;
  __shared__ double s_red[4][kLmStats];
  float F[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) F[k] = st->Ff[k];
  double acc[kLmStats];
#pragma unroll
  for (int k = 0; k < kLmStats; ++k) acc[k] = 0.0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < src.n_valid; i += gridDim.x * 256) {
    const int32_t pos = corr_pos[i];
    if (pos < 0) continue;
    const float4 s = src.xyzw[i];
    // the source as the estimator sees it: input_transformed = final_transformation * source (icp_mod.hpp:246)
    const float sx = xform_row(F + 0, s.x, s.y, s.z), sy = xform_row(F + 4, s.x, s.y, s.z), sz = xform_row(F + 8, s.x, s.y, s.z);
    const float4 t = tgt.pts[pos], n = tgt.nrm[pos];
    double phi[12];
    const double sv[4] = {(double)sx, (double)sy, (double)sz, 1.0}, nv[3] = {(double)n.x, (double)n.y, (double)n.z};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) phi[4 * a + b] = nv[a] * sv[b];
    // the residual at the identity, (s - t) . n: differences of floats are exact in double
    const double f0 = ((double)sx - (double)t.x) * nv[0] + ((double)sy - (double)t.y) * nv[1] + ((double)sz - (double)t.z) * nv[2];
    int slot = 0;
#pragma unroll
    for (int r = 0; r < 12; ++r)
#pragma unroll
      for (int c = r; c < 12; ++c) acc[slot++] += phi[r] * phi[c];
#pragma unroll
    for (int r = 0; r < 12; ++r) acc[78 + r] += f0 * phi[r];
    acc[90] += f0 * f0;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kLmStats; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) s_red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kLmStats)
    unsafeAtomicAdd(stats + threadIdx.x, s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}

// pos -> ORIGINAL target index for ope_icp_correspondences of an LM run
__global__ __launch_bounds__(256) void lm_pos_to_orig_kernel(BvhView tgt, int32_t *__restrict__ corr, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t pos = corr[i];
  corr[i] = pos >= 0 ? __float_as_int(tgt.pts[pos].w) : -1;
}
void launch_lm_pos_to_orig(hipStream_t stream, const BvhView &tgt, int32_t *d_corr, uint32_t n) {
  if (n) hipLaunchKernelGGL(lm_pos_to_orig_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, tgt, d_corr, n);
}

// one pass over the run's correspondences: the 91 sums are added into d_stats (left at zero by icp_lm_update_kernel)
void launch_lm_stats(hipStream_t stream, int n_cu, const CloudView &src, const BvhView &tgt, const IcpState *st, const int32_t *d_corr_pos, double *d_stats) {
  const int nblocks = (int)std::min<size_t>(std::max<size_t>(((size_t)src.n_valid + 255) / 256, 1), (size_t)std::max(n_cu, 1) * 2);
  hipLaunchKernelGGL(lm_stats_kernel, dim3(nblocks), dim3(256), 0, stream, src, tgt, st, d_corr_pos, d_stats);
}

}  // namespace ope
