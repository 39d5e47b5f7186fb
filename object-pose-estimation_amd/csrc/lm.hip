// lm.hip — OPE_EST_POINT_TO_PLANE_LM: the estimator BuildModel installs,
// pcl::registration::TransformationEstimationPointToPlane (BuildModel/src/regmeshpcd.cpp:162,193): Levenberg-Marquardt
// over WarpPointRigid6D (tx, ty, tz, qx, qy, qz) on the point-to-plane residuals of the current correspondences, run by
// Eigen's (unsupported) LevenbergMarquardt on a forward-difference Jacobian, all in float inside ICP.
//
// Split between device and host:
//   device  one pass over the stored correspondences per functor evaluation.  The residual f_i = (W(x) s_i - t_i) . n_i
//           is evaluated in FLOAT exactly as the reference's functor does (warp matrix from the quaternion, left-to-right
//           products); the Jacobian pass evaluates f at x and at x + h_j e_j for the six parameters in the same pass and
//           reduces  J^T J (21), J^T f (6), f^T f (1)  in double (J_ij = (f_i(x + h_j e_j) - f_i(x)) / h_j, float).
//   host    MINPACK's lmder logic as Eigen transcribes it (minimizeOneStep, lmpar2, qrsolv; factor 100, maxfev 400,
//           ftol = xtol = sqrt(FLT_EPSILON), gtol 0; NumericalDiff: h = sqrt(FLT_EPSILON) |x_j|, or sqrt(FLT_EPSILON) at 0)
//           in double on the 6 x 6 quantities.  R and Q^T f, which Eigen takes from a column-pivoted Householder QR of the
//           m x 6 float Jacobian, come from the Cholesky factor of J^T J: R^T R = J^T J, Q^T f = R^-T J^T f — the same
//           numbers in exact arithmetic (pivoting only re-orders the elimination).
// What the float QR's rounding can change in the result is measured in tests/test_oracle_lm.py (oracle in float vs double);
// the device path lands between the two.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

#include "bvh_traverse.hpp"

namespace ope {

constexpr int kLmSums = 28;   // 21 (upper triangle of J^T J, row by row) + 6 (J^T f) + 1 (f^T f)

struct LmWarps {
  float M[7][12];   // rows of [R | t] at x, then at x + h_j e_j
  float h[6];
};

// WarpPointRigid6D::setParam in float (Eigen::Quaternionf::normalize / toRotationMatrix)
static void lm_warp_matrix(const float x[6], float M[12]) {
  float qx = x[3], qy = x[4], qz = x[5];
  float qw = (float)std::sqrt((double)(1.0f - (qx * qx + qy * qy + qz * qz)));
  const float nn = (float)std::sqrt((double)(qw * qw + qx * qx + qy * qy + qz * qz));
  qw /= nn; qx /= nn; qy /= nn; qz /= nn;
  const float tx = 2.0f * qx, ty = 2.0f * qy, tz = 2.0f * qz;
  const float twx = tx * qw, twy = ty * qw, twz = tz * qw;
  const float txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  M[0] = 1.0f - (tyy + tzz); M[1] = txy - twz; M[2] = txz + twy; M[3] = x[0];
  M[4] = txy + twz; M[5] = 1.0f - (txx + tzz); M[6] = tyz - twx; M[7] = x[1];
  M[8] = txz - twy; M[9] = tyz + twx; M[10] = 1.0f - (txx + tyy); M[11] = x[2];
}

__device__ __forceinline__ float lm_residual(const float *M, float sx, float sy, float sz, const float4 t, const float4 n) {
  const float wx = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(M[0], sx), __fmul_rn(M[1], sy)), __fmul_rn(M[2], sz)), M[3]);
  const float wy = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(M[4], sx), __fmul_rn(M[5], sy)), __fmul_rn(M[6], sz)), M[7]);
  const float wz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(M[8], sx), __fmul_rn(M[9], sy)), __fmul_rn(M[10], sz)), M[11]);
  return __fadd_rn(__fadd_rn(__fmul_rn(__fsub_rn(wx, t.x), n.x), __fmul_rn(__fsub_rn(wy, t.y), n.y)), __fmul_rn(__fsub_rn(wz, t.z), n.z));
}

// corr_pos: per sorted source position the BVH position of its match, or -1 (what the accumulate kernels store in
// corr_match while the LM estimator is selected).  JAC: full Jacobian pass (28 sums) or residual norm only (sum 27).
template <bool JAC>
__global__ __launch_bounds__(256) void lm_eval_kernel(CloudView src, BvhView tgt, const IcpState *__restrict__ st,
                                                      const int32_t *__restrict__ corr_pos, LmWarps W, double *__restrict__ sums) {
  __shared__ double s_red[4][kLmSums];
  float F[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) F[k] = st->Ff[k];
  double acc[kLmSums];
#pragma unroll
  for (int k = 0; k < kLmSums; ++k) acc[k] = 0.0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < src.n_valid; i += gridDim.x * 256) {
    const int32_t pos = corr_pos[i];
    if (pos < 0) continue;
    const float4 s = src.xyzw[i];
    // the source as the estimator sees it: input_transformed = final_transformation * source (icp_mod.hpp:246)
    const float sx = xform_row(F + 0, s.x, s.y, s.z), sy = xform_row(F + 4, s.x, s.y, s.z), sz = xform_row(F + 8, s.x, s.y, s.z);
    const float4 t = tgt.pts[pos], n = tgt.nrm[pos];
    const float f0 = lm_residual(W.M[0], sx, sy, sz, t, n);
    acc[27] += (double)f0 * (double)f0;
    if (JAC) {
      float J[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) J[j] = __fdiv_rn(__fsub_rn(lm_residual(W.M[j + 1], sx, sy, sz, t, n), f0), W.h[j]);
      int slot = 0;
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) acc[slot++] += (double)J[r] * (double)J[c];
#pragma unroll
      for (int r = 0; r < 6; ++r) acc[21 + r] += (double)J[r] * (double)f0;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kLmSums; ++k) {
    if (!JAC && k != 27) continue;
    const double v = wave_sum(acc[k]);
    if (lane == 0) s_red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kLmSums && (JAC || threadIdx.x == 27))
    unsafeAtomicAdd(sums + threadIdx.x, s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
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

int comm_allreduce_sums(ope_ctx *ctx, double *d_sums, int count);

namespace {

struct LmHost {
  ope_ctx *ctx;
  CloudView src;
  BvhView tgt;
  const int32_t *corr_pos;
  double *d_sums;   // kLmSums doubles
  int nfev = 0;
  int nblocks;

  // one functor evaluation (or 7 with the Jacobian) over all correspondences -> the 28 sums on the host
  int eval(const double x[6], bool jac, double out[kLmSums]) {
    LmWarps W;
    float xf[6];
    for (int j = 0; j < 6; ++j) xf[j] = (float)x[j];
    lm_warp_matrix(xf, W.M[0]);
    const float eps = std::sqrt(FLT_EPSILON);
    for (int j = 0; j < 6; ++j) {
      float h = eps * std::fabs(xf[j]);
      if (h == 0.f) h = eps;
      float xx[6];
      std::memcpy(xx, xf, sizeof xx);
      xx[j] += h;
      lm_warp_matrix(xx, W.M[j + 1]);
      W.h[j] = h;
    }
    OPE_HIP(ctx, hipMemsetAsync(d_sums, 0, sizeof(double) * kLmSums, ctx->stream));
    if (jac)
      hipLaunchKernelGGL(lm_eval_kernel<true>, dim3(nblocks), dim3(256), 0, ctx->stream, src, tgt, ctx->d_state, corr_pos, W, d_sums);
    else
      hipLaunchKernelGGL(lm_eval_kernel<false>, dim3(nblocks), dim3(256), 0, ctx->stream, src, tgt, ctx->d_state, corr_pos, W, d_sums);
    if (ctx->nccl_comm) {
      const int rc = comm_allreduce_sums(ctx, d_sums, kLmSums);
      if (rc != OPE_OK) return rc;
    }
    OPE_HIP(ctx, hipMemcpyAsync(out, d_sums, sizeof(double) * kLmSums, hipMemcpyDeviceToHost, ctx->stream));
    OPE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    nfev += jac ? 7 : 1;
    return OPE_OK;
  }
};

double norm6(const double v[6]) {
  double s = 0;
  for (int i = 0; i < 6; ++i) s += v[i] * v[i];
  return std::sqrt(s);
}

// internal::qrsolv (identity permutation)
void qrsolv6(double s[6][6], const double d[6], const double qtb[6], double x[6], double sdiag[6]) {
  const int n = 6;
  double wa[6];
  for (int j = 0; j < n; ++j) {
    x[j] = s[j][j];
    for (int i = j + 1; i < n; ++i) s[i][j] = s[j][i];
    wa[j] = qtb[j];
  }
  for (int j = 0; j < n; ++j) {
    if (d[j] != 0) {
      for (int k = j + 1; k < n; ++k) sdiag[k] = 0;
      sdiag[j] = d[j];
      double qtbpj = 0;
      for (int k = j; k < n; ++k) {
        if (sdiag[k] == 0) continue;
        double c, sn;
        const double a = s[k][k], b = sdiag[k];
        if (std::fabs(a) < std::fabs(b)) { const double ct = a / b; sn = 1.0 / std::sqrt(1.0 + ct * ct); c = sn * ct; }
        else { const double tn = b / a; c = 1.0 / std::sqrt(1.0 + tn * tn); sn = c * tn; }
        s[k][k] = c * s[k][k] + sn * sdiag[k];
        const double temp = c * wa[k] + sn * qtbpj;
        qtbpj = -sn * wa[k] + c * qtbpj;
        wa[k] = temp;
        for (int i = k + 1; i < n; ++i) {
          const double t2 = c * s[i][k] + sn * sdiag[i];
          sdiag[i] = -sn * s[i][k] + c * sdiag[i];
          s[i][k] = t2;
        }
      }
    }
    sdiag[j] = s[j][j];
    s[j][j] = x[j];
  }
  int nsing = n;
  for (int j = 0; j < n; ++j) { if (sdiag[j] == 0 && nsing == n) nsing = j; if (nsing < n) wa[j] = 0; }
  for (int j = nsing - 1; j >= 0; --j) {
    double sum = 0;
    for (int i = j + 1; i < nsing; ++i) sum += s[i][j] * wa[i];
    wa[j] = (wa[j] - sum) / sdiag[j];
  }
  for (int j = 0; j < n; ++j) x[j] = wa[j];
}

// internal::lmpar2 (identity permutation)
void lmpar6(const double r[6][6], const double diag[6], const double qtb[6], double delta, double &par, double x[6]) {
  const int n = 6;
  const double dwarf = DBL_MIN;
  double wa1[6], wa2[6];
  int rank = n;
  for (int j = 0; j < n; ++j) if (r[j][j] == 0 && rank == n) rank = j;
  for (int j = 0; j < n; ++j) wa1[j] = j < rank ? qtb[j] : 0;
  for (int j = rank - 1; j >= 0; --j) {
    double sum = 0;
    for (int k = j + 1; k < rank; ++k) sum += r[j][k] * wa1[k];
    wa1[j] = (wa1[j] - sum) / r[j][j];
  }
  for (int j = 0; j < n; ++j) x[j] = wa1[j];
  int iter = 0;
  for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
  double dxnorm = norm6(wa2);
  double fp = dxnorm - delta;
  if (fp <= 0.1 * delta) { par = 0; return; }
  double parl = 0;
  if (rank == n) {
    for (int j = 0; j < n; ++j) wa1[j] = diag[j] * (wa2[j] / dxnorm);
    for (int j = 0; j < n; ++j) {
      double sum = 0;
      for (int i = 0; i < j; ++i) sum += r[i][j] * wa1[i];
      wa1[j] = (wa1[j] - sum) / r[j][j];
    }
    const double temp = norm6(wa1);
    parl = fp / delta / temp / temp;
  }
  for (int j = 0; j < n; ++j) {
    double sum = 0;
    for (int i = 0; i <= j; ++i) sum += r[i][j] * qtb[i];
    wa1[j] = sum / diag[j];
  }
  const double gnorm = norm6(wa1);
  double paru = gnorm / delta;
  if (paru == 0) paru = dwarf / std::min(delta, 0.1);
  par = std::max(par, parl);
  par = std::min(par, paru);
  if (par == 0) par = gnorm / dxnorm;
  for (;;) {
    ++iter;
    if (par == 0) par = std::max(dwarf, 0.001 * paru);
    const double sp = std::sqrt(par);
    double s[6][6], sdiag[6];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) s[i][j] = r[i][j];
    for (int j = 0; j < n; ++j) wa1[j] = sp * diag[j];
    qrsolv6(s, wa1, qtb, x, sdiag);
    for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
    dxnorm = norm6(wa2);
    double temp = fp;
    fp = dxnorm - delta;
    if (std::fabs(fp) <= 0.1 * delta || (parl == 0 && fp <= temp && temp < 0) || iter == 10) break;
    for (int j = 0; j < n; ++j) wa1[j] = diag[j] * (wa2[j] / dxnorm);
    for (int j = 0; j < n; ++j) {
      wa1[j] /= sdiag[j];
      temp = wa1[j];
      for (int i = j + 1; i < n; ++i) wa1[i] -= s[i][j] * temp;
    }
    temp = norm6(wa1);
    const double parc = fp / delta / temp / temp;
    if (fp > 0) parl = std::max(parl, par);
    if (fp < 0) paru = std::min(paru, par);
    par = std::max(parl, par + parc);
  }
  if (iter == 0) par = 0;
}

}  // namespace

// LevenbergMarquardt::minimize from x = 0 on the correspondences of the last accumulate launch.
// out_T: column-major float 4x4 (the warp matrix of the minimiser).  n_corr: the run's correspondence count (global).
int lm_point_to_plane(ope_ctx *ctx, const CloudView &src, const BvhView &tgt, const int32_t *d_corr_pos, double *d_sums, long long n_corr,
                      float out_T[16], int *nfev_out) {
  static const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::memcpy(out_T, I4, sizeof I4);
  if (nfev_out) *nfev_out = 0;
  if (n_corr < 4) return OPE_OK;   // "Number or points in source (%d) differs than target" / "< 4": PCL returns without a transform
  LmHost L{ctx, src, tgt, d_corr_pos, d_sums};
  L.nblocks = (int)std::min<size_t>(std::max<size_t>(((size_t)src.n_valid + 255) / 256, 1), 2048);
  const int n = 6;
  const double epsmch = FLT_EPSILON;   // the reference optimises in float: its tolerances and difference steps are float's
  const double ftol = std::sqrt(epsmch), xtol = ftol, gtol = 0, factor = 100;
  const int maxfev = 400;
  double x[6] = {0, 0, 0, 0, 0, 0}, S[kLmSums], diag[6], qtf[6], wa1[6], wa2[6], wa3[6], r[6][6];
  int rc = L.eval(x, false, S);   // minimizeInit: f(x0)
  if (rc != OPE_OK) return rc;
  double fnorm = std::sqrt(S[27]), par = 0, delta = 0, xnorm = 0;
  int iter = 1;
  for (;;) {
    rc = L.eval(x, true, S);   // NumericalDiff<Forward>::df: f(x) once more and one evaluation per parameter
    if (rc != OPE_OK) return rc;
    // R^T R = J^T J (Cholesky, upper), Q^T f = R^-T (J^T f)
    double A[6][6];
    {
      int k = 0;
      for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) { A[i][j] = S[k]; A[j][i] = S[k]; ++k; }
    }
    for (int j = 0; j < n; ++j) wa2[j] = std::sqrt(std::max(A[j][j], 0.0));   // column norms
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) r[i][j] = 0;
    for (int j = 0; j < n; ++j) {
      double d = A[j][j];
      for (int k = 0; k < j; ++k) d -= r[k][j] * r[k][j];
      r[j][j] = d > 0 ? std::sqrt(d) : 0.0;
      for (int i = j + 1; i < n; ++i) {
        double v = A[j][i];
        for (int k = 0; k < j; ++k) v -= r[k][j] * r[k][i];
        r[j][i] = r[j][j] != 0 ? v / r[j][j] : 0.0;
      }
    }
    for (int j = 0; j < n; ++j) {
      double v = S[21 + j];
      for (int k = 0; k < j; ++k) v -= r[k][j] * qtf[k];
      qtf[j] = r[j][j] != 0 ? v / r[j][j] : 0.0;
    }
    if (iter == 1) {
      for (int j = 0; j < n; ++j) diag[j] = wa2[j] == 0 ? 1 : wa2[j];
      for (int j = 0; j < n; ++j) wa3[j] = diag[j] * x[j];
      xnorm = norm6(wa3);
      delta = factor * xnorm;
      if (delta == 0) delta = factor;
    }
    double gnorm = 0;
    if (fnorm != 0)
      for (int j = 0; j < n; ++j)
        if (wa2[j] != 0) {
          double sum = 0;
          for (int i = 0; i <= j; ++i) sum += r[i][j] * (qtf[i] / fnorm);
          gnorm = std::max(gnorm, std::fabs(sum / wa2[j]));
        }
    if (gnorm <= gtol) break;
    for (int j = 0; j < n; ++j) diag[j] = std::max(diag[j], wa2[j]);
    double ratio = 0;
    bool done = false;
    do {
      lmpar6(r, diag, qtf, delta, par, wa1);
      for (int j = 0; j < n; ++j) { wa1[j] = -wa1[j]; wa2[j] = x[j] + wa1[j]; wa3[j] = diag[j] * wa1[j]; }
      const double pnorm = norm6(wa3);
      if (iter == 1) delta = std::min(delta, pnorm);
      double S1[kLmSums];
      rc = L.eval(wa2, false, S1);
      if (rc != OPE_OK) return rc;
      const double fnorm1 = std::sqrt(S1[27]);
      double actred = -1;
      if (0.1 * fnorm1 < fnorm) actred = 1 - (fnorm1 / fnorm) * (fnorm1 / fnorm);
      for (int i = 0; i < n; ++i) {
        double sum = 0;
        for (int j = i; j < n; ++j) sum += r[i][j] * wa1[j];
        wa3[i] = sum;
      }
      const double t1 = norm6(wa3) / fnorm, t2 = std::sqrt(par) * pnorm / fnorm;
      const double temp1 = t1 * t1, temp2 = t2 * t2;
      const double prered = temp1 + temp2 / 0.5, dirder = -(temp1 + temp2);
      ratio = prered != 0 ? actred / prered : 0;
      if (ratio <= 0.25) {
        double temp = 0;
        if (actred >= 0) temp = 0.5;
        if (actred < 0) temp = 0.5 * dirder / (dirder + 0.5 * actred);
        if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
        delta = temp * std::min(delta, pnorm / 0.1);
        par /= temp;
      } else if (!(par != 0 && ratio < 0.75)) {
        delta = pnorm / 0.5;
        par = 0.5 * par;
      }
      if (ratio >= 1e-4) {
        for (int j = 0; j < n; ++j) { x[j] = wa2[j]; wa2[j] = diag[j] * x[j]; }
        xnorm = norm6(wa2);
        fnorm = fnorm1;
        ++iter;
      }
      const bool small_red = std::fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1;
      if (small_red || delta <= xtol * xnorm || L.nfev >= maxfev) { done = true; break; }
      if ((std::fabs(actred) <= epsmch && prered <= epsmch && 0.5 * ratio <= 1) || delta <= epsmch * xnorm || gnorm <= epsmch) { done = true; break; }
    } while (ratio < 1e-4);
    if (done) break;
  }
  float xf[6], M[12];
  for (int j = 0; j < 6; ++j) xf[j] = (float)x[j];
  lm_warp_matrix(xf, M);
  for (int rr = 0; rr < 3; ++rr)
    for (int c = 0; c < 4; ++c) out_T[4 * c + rr] = M[4 * rr + c];
  if (nfev_out) *nfev_out = L.nfev;
  return OPE_OK;
}

}  // namespace ope
