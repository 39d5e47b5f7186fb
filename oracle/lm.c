/*
 * lm.c — CPU ORACLE (test infrastructure, NOT product code): the estimator BuildModel installs,
 * pcl::registration::TransformationEstimationPointToPlane (Levenberg-Marquardt; BuildModel/src/regmeshpcd.cpp:162,193).
 * PARITY UNPINNED, see ope_oracle.h.  The algorithm text is in lm_impl.inc, instantiated here for float (the
 * reference's MatScalar inside ICP) and for double (to measure what float rounding alone does to the result).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ope_oracle.h"

#define REAL float
#define NAME(x) x##_f
#include "lm_impl.inc"
#undef REAL
#undef NAME
#define REAL double
#define NAME(x) x##_d
#include "lm_impl.inc"
#undef REAL
#undef NAME

/* estimateRigidTransformation on n already-paired points: src = the current (transformed) source points, tgt / tgt_nrm
 * = matched target points and normals.  precision 0: float (as PCL), 1: double.  T: column-major 4x4 = the warp matrix
 * of the minimiser.  Returns 0, or -1 for fewer than 4 pairs (PCL refuses those and leaves the transform untouched). */
int orc_point_to_plane_lm(const float *src, const float *tgt, const float *tgt_nrm, int n, int precision, float T[16],
                          double x_out[6], int *nfev_out, int *status_out) {
  static const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  memcpy(T, I4, sizeof I4);
  if (n < 4) return -1;
  int status = 0, nfev = 0;
  double M[12], xo[6];
  if (precision == 0) {
    lm_problem_f P = {src, tgt, tgt_nrm, n};
    float x[6] = {0, 0, 0, 0, 0, 0}, Mf[12];
    nfev = lm_minimize_f(&P, x, &status);
    warp_matrix_f(x, Mf);
    for (int i = 0; i < 12; ++i) M[i] = Mf[i];
    for (int i = 0; i < 6; ++i) xo[i] = x[i];
  } else {
    lm_problem_d P = {src, tgt, tgt_nrm, n};
    double x[6] = {0, 0, 0, 0, 0, 0};
    nfev = lm_minimize_d(&P, x, &status);
    warp_matrix_d(x, M);
    for (int i = 0; i < 6; ++i) xo[i] = x[i];
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) T[4 * c + r] = (float)M[4 * r + c];
  if (x_out) memcpy(x_out, xo, sizeof xo);
  if (nfev_out) *nfev_out = nfev;
  if (status_out) *status_out = status;
  return 0;
}
