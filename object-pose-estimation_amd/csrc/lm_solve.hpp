// lm_solve.hpp — the Levenberg-Marquardt minimisation of OPE_EST_POINT_TO_PLANE_LM on the 91 sums of lm_stats_kernel
// (lm.hip), one lane.  Eigen's unsupported LevenbergMarquardt (the transcription of MINPACK's lmder that
// pcl::registration::TransformationEstimationPointToPlane runs, BuildModel/src/regmeshpcd.cpp:162,193): minimizeInit /
// minimizeOneStep, internal::lmpar2, internal::qrsolv, NumericalDiff<Forward>; factor 100, maxfev 400, ftol = xtol =
// sqrt(epsilon), gtol 0, epsilon = FLT_EPSILON (the reference optimises in float: its tolerances and difference steps are
// float's).  The 6 x 6 algebra is in double.
#pragma once

#include <cfloat>

namespace ope {

struct LmQuad {
  double A[12][12];   // sum phi phi^T
  double g[12];       // sum f(I) phi
  double c0;          // sum f(I)^2
};

// the 91 packed sums (upper triangle of A row by row, g, c0) -> q, by the threads of one block
__device__ __forceinline__ void lm_load_stats(const double *stats, LmQuad &q) {
  for (int e = threadIdx.x; e < 144; e += blockDim.x) {
    const int r = e / 12, c = e % 12, lo = r < c ? r : c, hi = r < c ? c : r;
    q.A[r][c] = stats[lo * 12 - lo * (lo - 1) / 2 + (hi - lo)];
  }
  for (int r = threadIdx.x; r < 12; r += blockDim.x) q.g[r] = stats[78 + r];
  if (threadIdx.x == 0) q.c0 = stats[90];
}

__device__ __forceinline__ void lm_warp_matrix_dev(const float x[6], float M[12]) {
  // WarpPointRigid6D::setParam in float (Eigen::Quaternionf::normalize / toRotationMatrix); every operation rounded once
  float qx = x[3], qy = x[4], qz = x[5];
  float qw = (float)sqrt((double)__fsub_rn(1.0f, __fadd_rn(__fadd_rn(__fmul_rn(qx, qx), __fmul_rn(qy, qy)), __fmul_rn(qz, qz))));
  const float nn = (float)sqrt((double)__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(qw, qw), __fmul_rn(qx, qx)), __fmul_rn(qy, qy)), __fmul_rn(qz, qz)));
  qw = __fdiv_rn(qw, nn); qx = __fdiv_rn(qx, nn); qy = __fdiv_rn(qy, nn); qz = __fdiv_rn(qz, nn);
  const float tx = __fmul_rn(2.0f, qx), ty = __fmul_rn(2.0f, qy), tz = __fmul_rn(2.0f, qz);
  const float twx = __fmul_rn(tx, qw), twy = __fmul_rn(ty, qw), twz = __fmul_rn(tz, qw);
  const float txx = __fmul_rn(tx, qx), txy = __fmul_rn(ty, qx), txz = __fmul_rn(tz, qx), tyy = __fmul_rn(ty, qy), tyz = __fmul_rn(tz, qy),
              tzz = __fmul_rn(tz, qz);
  M[0] = __fsub_rn(1.0f, __fadd_rn(tyy, tzz)); M[1] = __fsub_rn(txy, twz); M[2] = __fadd_rn(txz, twy); M[3] = x[0];
  M[4] = __fadd_rn(txy, twz); M[5] = __fsub_rn(1.0f, __fadd_rn(txx, tzz)); M[6] = __fsub_rn(tyz, twx); M[7] = x[1];
  M[8] = __fsub_rn(txz, twy); M[9] = __fadd_rn(tyz, twx); M[10] = __fsub_rn(1.0f, __fadd_rn(txx, tyy)); M[11] = x[2];
}

// D = W(x) - I for the float warp matrix of x
__device__ __forceinline__ void lm_delta_of(const double x[6], double D[12], float M[12]) {
  float xf[6];
  for (int j = 0; j < 6; ++j) xf[j] = (float)x[j];
  lm_warp_matrix_dev(xf, M);
  for (int k = 0; k < 12; ++k) D[k] = (double)M[k] - ((k == 0 || k == 5 || k == 10) ? 1.0 : 0.0);
}

// f^T f at D
__device__ __forceinline__ double lm_fsq(const LmQuad &q, const double D[12]) {
  double v = q.c0;
  for (int r = 0; r < 12; ++r) {
    double ar = 0;
    for (int c = 0; c < 12; ++c) ar += q.A[r][c] * D[c];
    v += D[r] * (2.0 * q.g[r] + ar);
  }
  return v > 0 ? v : 0;
}

__device__ __forceinline__ double lm_norm6(const double v[6]) {
  double s = 0;
  for (int i = 0; i < 6; ++i) s += v[i] * v[i];
  return sqrt(s);
}

// internal::qrsolv (identity permutation)
__device__ inline void lm_qrsolv6(double s[6][6], const double d[6], const double qtb[6], double x[6], double sdiag[6]) {
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
        if (fabs(a) < fabs(b)) { const double ct = a / b; sn = 1.0 / sqrt(1.0 + ct * ct); c = sn * ct; }
        else { const double tn = b / a; c = 1.0 / sqrt(1.0 + tn * tn); sn = c * tn; }
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
__device__ inline void lm_lmpar6(const double r[6][6], const double diag[6], const double qtb[6], double delta, double &par, double x[6]) {
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
  double dxnorm = lm_norm6(wa2);
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
    const double temp = lm_norm6(wa1);
    parl = fp / delta / temp / temp;
  }
  for (int j = 0; j < n; ++j) {
    double sum = 0;
    for (int i = 0; i <= j; ++i) sum += r[i][j] * qtb[i];
    wa1[j] = sum / diag[j];
  }
  const double gnorm = lm_norm6(wa1);
  double paru = gnorm / delta;
  if (paru == 0) paru = dwarf / fmin(delta, 0.1);
  par = fmax(par, parl);
  par = fmin(par, paru);
  if (par == 0) par = gnorm / dxnorm;
  for (;;) {
    ++iter;
    if (par == 0) par = fmax(dwarf, 0.001 * paru);
    const double sp = sqrt(par);
    double s[6][6], sdiag[6];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) s[i][j] = r[i][j];
    for (int j = 0; j < n; ++j) wa1[j] = sp * diag[j];
    lm_qrsolv6(s, wa1, qtb, x, sdiag);
    for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
    dxnorm = lm_norm6(wa2);
    double temp = fp;
    fp = dxnorm - delta;
    if (fabs(fp) <= 0.1 * delta || (parl == 0 && fp <= temp && temp < 0) || iter == 10) break;
    for (int j = 0; j < n; ++j) wa1[j] = diag[j] * (wa2[j] / dxnorm);
    for (int j = 0; j < n; ++j) {
      wa1[j] /= sdiag[j];
      temp = wa1[j];
      for (int i = j + 1; i < n; ++i) wa1[i] -= s[i][j] * temp;
    }
    temp = lm_norm6(wa1);
    const double parc = fp / delta / temp / temp;
    if (fp > 0) parl = fmax(parl, par);
    if (fp < 0) paru = fmin(paru, par);
    par = fmax(parl, par + parc);
  }
  if (iter == 0) par = 0;
}

// LevenbergMarquardt::minimize from x = 0 on the sums of this iteration's correspondences.  out_T: column-major float 4x4
// (the warp matrix of the minimiser).  n_corr: the run's correspondence count (global).  Returns the number of functor
// evaluations (nfev as Eigen counts them: 7 per Jacobian).
// q: the sums, unpacked (lm_load_stats) — in LDS, so that the lane's 12 x 12 loops read it with ds_read instead of spilling it
__device__ inline int lm_minimize_lane(const LmQuad &q, long long n_corr, float out_T[16]) {
  for (int i = 0; i < 16; ++i) out_T[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  if (n_corr < 4) return 0;   // "Number or points in source (%d) differs than target" / "< 4": PCL returns without a transform
  const int n = 6;
  const double epsmch = FLT_EPSILON;
  const double ftol = sqrt(epsmch), xtol = ftol, gtol = 0, factor = 100;
  const int maxfev = 400;
  int nfev = 0;
  double x[6] = {0, 0, 0, 0, 0, 0}, diag[6], qtf[6], wa1[6], wa2[6], wa3[6], r[6][6];
  double D[12];
  float M[12];
  lm_delta_of(x, D, M);
  double fnorm = sqrt(lm_fsq(q, D)), par = 0, delta = 0, xnorm = 0;   // minimizeInit: f(x0)
  ++nfev;
  int iter = 1;
  for (;;) {
    // NumericalDiff<Forward>::df: f(x) once more and one evaluation per parameter; J_j = Phi u_j, u_j = (W(x + h e_j) - W(x)) / h
    double u[6][12], AD[12];
    lm_delta_of(x, D, M);
    {
      const float eps = sqrtf(FLT_EPSILON);
      float xf[6];
      for (int j = 0; j < 6; ++j) xf[j] = (float)x[j];
      for (int j = 0; j < 6; ++j) {
        float h = __fmul_rn(eps, fabsf(xf[j]));
        if (h == 0.f) h = eps;
        float xx[6], Mj[12];
        for (int k = 0; k < 6; ++k) xx[k] = xf[k];
        xx[j] = __fadd_rn(xx[j], h);
        lm_warp_matrix_dev(xx, Mj);
        for (int k = 0; k < 12; ++k) u[j][k] = ((double)Mj[k] - (double)M[k]) / (double)h;
      }
    }
    nfev += 7;
    for (int rr = 0; rr < 12; ++rr) {
      double a = q.g[rr];
      for (int c = 0; c < 12; ++c) a += q.A[rr][c] * D[c];
      AD[rr] = a;   // Phi^T f = g + A D
    }
    double A6[6][6], Jtf[6];
    for (int i = 0; i < n; ++i) {
      double Au[12];
      for (int rr = 0; rr < 12; ++rr) {
        double a = 0;
        for (int c = 0; c < 12; ++c) a += q.A[rr][c] * u[i][c];
        Au[rr] = a;
      }
      for (int j = i; j < n; ++j) {
        double a = 0;
        for (int rr = 0; rr < 12; ++rr) a += u[j][rr] * Au[rr];
        A6[i][j] = a; A6[j][i] = a;
      }
      double b = 0;
      for (int rr = 0; rr < 12; ++rr) b += u[i][rr] * AD[rr];
      Jtf[i] = b;
    }
    // R^T R = J^T J (Cholesky, upper), Q^T f = R^-T (J^T f)
    for (int j = 0; j < n; ++j) wa2[j] = sqrt(fmax(A6[j][j], 0.0));   // column norms
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) r[i][j] = 0;
    for (int j = 0; j < n; ++j) {
      double d = A6[j][j];
      for (int k = 0; k < j; ++k) d -= r[k][j] * r[k][j];
      r[j][j] = d > 0 ? sqrt(d) : 0.0;
      for (int i = j + 1; i < n; ++i) {
        double v = A6[j][i];
        for (int k = 0; k < j; ++k) v -= r[k][j] * r[k][i];
        r[j][i] = r[j][j] != 0 ? v / r[j][j] : 0.0;
      }
    }
    for (int j = 0; j < n; ++j) {
      double v = Jtf[j];
      for (int k = 0; k < j; ++k) v -= r[k][j] * qtf[k];
      qtf[j] = r[j][j] != 0 ? v / r[j][j] : 0.0;
    }
    if (iter == 1) {
      for (int j = 0; j < n; ++j) diag[j] = wa2[j] == 0 ? 1 : wa2[j];
      for (int j = 0; j < n; ++j) wa3[j] = diag[j] * x[j];
      xnorm = lm_norm6(wa3);
      delta = factor * xnorm;
      if (delta == 0) delta = factor;
    }
    double gnorm = 0;
    if (fnorm != 0)
      for (int j = 0; j < n; ++j)
        if (wa2[j] != 0) {
          double sum = 0;
          for (int i = 0; i <= j; ++i) sum += r[i][j] * (qtf[i] / fnorm);
          gnorm = fmax(gnorm, fabs(sum / wa2[j]));
        }
    if (gnorm <= gtol) break;
    for (int j = 0; j < n; ++j) diag[j] = fmax(diag[j], wa2[j]);
    double ratio = 0;
    bool done = false;
    do {
      lm_lmpar6(r, diag, qtf, delta, par, wa1);
      for (int j = 0; j < n; ++j) { wa1[j] = -wa1[j]; wa2[j] = x[j] + wa1[j]; wa3[j] = diag[j] * wa1[j]; }
      const double pnorm = lm_norm6(wa3);
      if (iter == 1) delta = fmin(delta, pnorm);
      double D1[12];
      float M1[12];
      lm_delta_of(wa2, D1, M1);
      const double fnorm1 = sqrt(lm_fsq(q, D1));
      ++nfev;
      double actred = -1;
      if (0.1 * fnorm1 < fnorm) actred = 1 - (fnorm1 / fnorm) * (fnorm1 / fnorm);
      for (int i = 0; i < n; ++i) {
        double sum = 0;
        for (int j = i; j < n; ++j) sum += r[i][j] * wa1[j];
        wa3[i] = sum;
      }
      const double t1 = lm_norm6(wa3) / fnorm, t2 = sqrt(par) * pnorm / fnorm;
      const double temp1 = t1 * t1, temp2 = t2 * t2;
      const double prered = temp1 + temp2 / 0.5, dirder = -(temp1 + temp2);
      ratio = prered != 0 ? actred / prered : 0;
      if (ratio <= 0.25) {
        double temp = 0;
        if (actred >= 0) temp = 0.5;
        if (actred < 0) temp = 0.5 * dirder / (dirder + 0.5 * actred);
        if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
        delta = temp * fmin(delta, pnorm / 0.1);
        par /= temp;
      } else if (!(par != 0 && ratio < 0.75)) {
        delta = pnorm / 0.5;
        par = 0.5 * par;
      }
      if (ratio >= 1e-4) {
        for (int j = 0; j < n; ++j) { x[j] = wa2[j]; wa2[j] = diag[j] * x[j]; }
        xnorm = lm_norm6(wa2);
        fnorm = fnorm1;
        ++iter;
      }
      const bool small_red = fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1;
      if (small_red || delta <= xtol * xnorm || nfev >= maxfev) { done = true; break; }
      if ((fabs(actred) <= epsmch && prered <= epsmch && 0.5 * ratio <= 1) || delta <= epsmch * xnorm || gnorm <= epsmch) { done = true; break; }
    } while (ratio < 1e-4);
    if (done) break;
  }
  lm_delta_of(x, D, M);
  for (int rr = 0; rr < 3; ++rr)
    for (int c = 0; c < 4; ++c) out_T[4 * c + rr] = M[4 * rr + c];
  out_T[3] = 0.f; out_T[7] = 0.f; out_T[11] = 0.f; out_T[15] = 1.f;
  return nfev;
}

}  // namespace ope
