/*
 * pose.c — CPU ORACLE (test infrastructure, NOT product code): the L2 composite of the hot path,
 * DetectAndLocalize/src/poseestimator.cpp, restated on top of the oracle's primitives.  PARITY UNPINNED, see
 * ope_oracle.h.
 *
 *   PoseEstimator::PoseEstimator                 poseestimator.cpp:3-13
 *   PoseEstimator::estimateCoarsePose            :16-73
 *   PoseEstimator::getFpfhFeatures               :110-128
 *   PoseEstimator::subSampleAndCalculateNormals  :131-158
 *   PoseEstimator::estimateFinePose              :161-379
 *   PoseEstimator::estimateFinalPose             :383-448
 *
 * What is deliberately kept from the reference:
 *   - the gate `fitnessScoreFine > 0.0001` on re-running the coarse stage (:399) and the state that crosses frames
 *     (alignedSource, cloudModel, firstTimePose, fitnessScoreFine, alignedStrength: poseestimator.h:50-53);
 *   - quirk Q4: pose = coarsePose * finePose (:421) and finalPose = rigidmodelPose * pose (:439), i.e. the products are
 *     taken in the opposite order to the one in which the transforms were applied to the cloud;
 *   - the re-anchoring fit of the stored model onto the INCOMING source by identity correspondences (:429-435);
 *   - the guards: < 10 target features -> identity and alignedSource = source (:40-45); < 100 target points after the
 *     fine stage's sub-sampling -> identity, source untouched (:218-223); empty target -> both stages skipped (:397-418);
 *   - the ICP configuration of :310-341: 100 iterations, transformation / fitness epsilon 1e-8, normal shooting k = 20
 *     with PCL's default (unbounded) correspondence distance (the 0.01 at :247 goes to a stand-alone call whose result
 *     is discarded), SurfaceNormal rejector 0.7, SelfOccluded rejector 0.6 only on request, SVD estimator.
 * What differs, and why:
 *   - PCL draws SAC-IA's samples from an unseeded rand() whose stream continues from frame to frame; here
 *     the k-th coarse call of an estimator uses the explicit stream `sacia_seed + k`;
 *   - UniformSampling's survivors come out in ascending voxel-key order instead of boost::unordered_map order (Q7).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ope_oracle.h"

static const float kI4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

void orc_pose_estimator_init(orc_pose_estimator *pe) {
  memset(pe, 0, sizeof *pe);
  pe->first_time_pose = 0;
  pe->fitness_score_fine = 10.0; /* "random high value", poseestimator.cpp:6 */
  pe->aligned_strength = 0.0;
  memcpy(pe->final_pose, kI4, sizeof kI4);
  pe->sacia_seed = 1;
  pe->acc_mode = 1;
  pe->transform_mode = 1;
}

void orc_pose_estimator_free(orc_pose_estimator *pe) {
  free(pe->aligned_source);
  free(pe->cloud_model);
  pe->aligned_source = pe->cloud_model = NULL;
  pe->n_aligned = pe->n_model = 0;
}

static void set_cloud(float **dst, int *n_dst, const float *src, int n) {
  free(*dst);
  *dst = (float *)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
  if (n > 0) memcpy(*dst, src, sizeof(float) * 3 * (size_t)n);
  *n_dst = n;
}

/* Eigen::Matrix4f product, column-major, float accumulation in Eigen's order (sum over k ascending) */
static void mul44f(const float *a, const float *b, float *o) {
  float r[16];
  for (int c = 0; c < 4; ++c)
    for (int i = 0; i < 4; ++i) {
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += a[4 * k + i] * b[4 * c + k];
      r[4 * c + i] = s;
    }
  memcpy(o, r, sizeof r);
}

/* subSampleAndCalculateNormals (:131-158): UniformSampling(radius = leaf) -> copy of the survivors -> NormalEstimation(k = 30).
 * Outputs are malloc'ed (caller frees). */
int orc_sub_sample_and_normals(const float *xyz, int n, float leaf, float **out_xyz, float **out_nrm, float **out_curv) {
  int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  const int m = n > 0 ? orc_uniform_sampling(xyz, n, leaf, idx) : 0;
  float *p = (float *)malloc(sizeof(float) * 3 * (size_t)(m > 0 ? m : 1));
  for (int i = 0; i < m; ++i) memcpy(p + 3 * i, xyz + 3 * (size_t)idx[i], 12);
  free(idx);
  float *nr = (float *)malloc(sizeof(float) * 3 * (size_t)(m > 0 ? m : 1));
  float *cv = (float *)malloc(sizeof(float) * (size_t)(m > 0 ? m : 1));
  const float vp[3] = {0.f, 0.f, 0.f};
  if (m > 0) orc_normals_knn(p, m, 30, vp, nr, cv);
  *out_xyz = p;
  *out_nrm = nr;
  if (out_curv) *out_curv = cv; else free(cv);
  return m;
}

/* getFpfhFeatures (:110-128): sub-sample at 0.01, FPFH r = 0.03.  The key points REPLACE the input cloud (:117-118). */
static int fpfh_features(const float *xyz, int n, float **key_xyz, float **feat33) {
  float *nrm = NULL;
  const int m = orc_sub_sample_and_normals(xyz, n, 0.01f, key_xyz, &nrm, NULL);
  *feat33 = (float *)malloc(sizeof(float) * 33 * (size_t)(m > 0 ? m : 1));
  if (m > 0) orc_fpfh(*key_xyz, nrm, m, 0.03f, *feat33, NULL, NULL);
  free(nrm);
  return m;
}

int orc_estimate_coarse_pose(orc_pose_estimator *pe, const float *src, int ns, const float *tgt, int nt, float out_T[16]) {
  memcpy(out_T, kI4, sizeof kI4);
  float *sk = NULL, *sf = NULL, *tk = NULL, *tf = NULL;
  const int nsk = fpfh_features(src, ns, &sk, &sf);
  const int ntk = fpfh_features(tgt, nt, &tk, &tf);
  int rc = 0;
  if (ntk < 10) { /* "NO target cloud in Initial Alignment" (:40-45) */
    set_cloud(&pe->aligned_source, &pe->n_aligned, src, ns);
  } else {
    double err = 0;
    int32_t best = -1;
    rc = orc_sacia(sk, sf, nsk, tk, tf, ntk, 400, 5, 5, 0.05, 0.01f, pe->sacia_seed + (uint64_t)pe->coarse_calls, NULL, out_T, &err,
                   &best);
    ++pe->coarse_calls;
    pe->last_sacia_error = err;
    pe->last_sacia_best = best;
    if (rc != 0) memcpy(out_T, kI4, sizeof kI4);
    /* alignedSource = pose * (full-resolution source) (:66-70) */
    float *al = (float *)malloc(sizeof(float) * 3 * (size_t)(ns > 0 ? ns : 1));
    orc_transform_points(src, ns, out_T, al);
    free(pe->aligned_source);
    pe->aligned_source = al;
    pe->n_aligned = ns;
  }
  pe->last_n_src_keys = nsk;
  pe->last_n_tgt_keys = ntk;
  free(sk); free(sf); free(tk); free(tf);
  return rc;
}

/* drop the points whose normal is not finite (removeNaNNormalsFromPointCloud, :213-216); returns the new count */
static int drop_nan_normals(float *xyz, float *nrm, int n) {
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const float *q = nrm + 3 * i;
    if (!(isfinite(q[0]) && isfinite(q[1]) && isfinite(q[2]))) continue;
    if (m != i) { memcpy(xyz + 3 * m, xyz + 3 * i, 12); memcpy(nrm + 3 * m, q, 12); }
    ++m;
  }
  return m;
}

/* estimateFinePose (:161-379).  `src` (n points, the caller's cloud: alignedSource in estimateFinalPose) is REPLACED by
 * pose * src on success (:357-360), exactly like the Ptr& of the reference. */
int orc_estimate_fine_pose(orc_pose_estimator *pe, float *src, int ns, const float *tgt, int nt, float out_T[16]) {
  memcpy(out_T, kI4, sizeof kI4);
  /* removeNaNFromPointCloud on copies of both clouds (:186-194) */
  int32_t *keep = (int32_t *)malloc(sizeof(int32_t) * (size_t)((ns > nt ? ns : nt) + 1));
  const int ms = orc_remove_nan(src, ns, keep);
  float *s0 = (float *)malloc(sizeof(float) * 3 * (size_t)(ms > 0 ? ms : 1));
  for (int i = 0; i < ms; ++i) memcpy(s0 + 3 * i, src + 3 * (size_t)keep[i], 12);
  const int mt = orc_remove_nan(tgt, nt, keep);
  float *t0 = (float *)malloc(sizeof(float) * 3 * (size_t)(mt > 0 ? mt : 1));
  for (int i = 0; i < mt; ++i) memcpy(t0 + 3 * i, tgt + 3 * (size_t)keep[i], 12);
  free(keep);
  float *sk = NULL, *sn = NULL, *tk = NULL, *tn = NULL;
  int nsk = orc_sub_sample_and_normals(s0, ms, 0.008f, &sk, &sn, NULL);
  int ntk = orc_sub_sample_and_normals(t0, mt, 0.008f, &tk, &tn, NULL);
  free(s0); free(t0);
  nsk = drop_nan_normals(sk, sn, nsk);
  ntk = drop_nan_normals(tk, tn, ntk);
  int rc = 0;
  if (ntk >= 100) { /* else "NO target cloud in Final Alignment" (:218-223) */
    orc_icp_params p;
    orc_icp_default_params(&p);
    p.max_iterations = 100;             /* :322 */
    p.transformation_epsilon = 1e-8;    /* :325 */
    p.euclidean_fitness_epsilon = 1e-8; /* :328 */
    p.corr_mode = 1;                    /* CorrespondenceEstimationNormalShooting (:331) */
    p.k_normal_shooting = 20;           /* :246 */
    p.use_surface_normal_rej = 1;       /* :334 */
    p.surface_normal_thr = 0.7;         /* :272 */
    p.use_self_occluded_rej = pe->use_self_occluded; /* :336, SURVEY Q3 */
    p.self_occluded_thr = 0.6;          /* :291 */
    p.estimator = 0;                    /* TransformationEstimationSVD (:341) */
    p.acc_mode = pe->acc_mode;
    p.transform_mode = pe->transform_mode;
    orc_icp_result res;
    rc = orc_icp(sk, sn, nsk, tk, tn, ntk, NULL, &p, out_T, &res, NULL, NULL, NULL, NULL);
    if (rc == 0) {
      pe->fitness_score_fine = res.fitness;       /* :354 */
      pe->aligned_strength = res.align_strength;  /* :363 */
      pe->last_icp_iterations = res.iterations;
      pe->last_icp_state = res.state;
      pe->last_icp_n_corr = res.n_corr;
      float *al = (float *)malloc(sizeof(float) * 3 * (size_t)(ns > 0 ? ns : 1));
      orc_transform_points(src, ns, out_T, al);   /* :358-360 */
      memcpy(src, al, sizeof(float) * 3 * (size_t)ns);
      free(al);
    } else {
      memcpy(out_T, kI4, sizeof kI4);
    }
  }
  pe->last_n_fine_src = nsk;
  pe->last_n_fine_tgt = ntk;
  free(sk); free(sn); free(tk); free(tn);
  return rc;
}

/* estimateFinalPose (:383-448).  `src` (ns points) is the caller's source cloud and is OVERWRITTEN with alignedSource
 * (:441; the two have the same size because alignedSource is always a transformed copy of a source of this size —
 * callers that change the source size between frames get min(ns, n_aligned) points copied). */
int orc_estimate_final_pose(orc_pose_estimator *pe, float *src, int ns, const float *tgt, int nt, float out_pose[16],
                            double *fitness_score, double *align_strength) {
  if (pe->first_time_pose == 0) set_cloud(&pe->cloud_model, &pe->n_model, src, ns); /* :386-388 */
  pe->first_time_pose++;
  float coarse[16], fine[16];
  memcpy(coarse, kI4, sizeof kI4);
  memcpy(fine, kI4, sizeof kI4);
  int rc = 0;
  if (nt > 0 && pe->fitness_score_fine > 0.0001) rc = orc_estimate_coarse_pose(pe, src, ns, tgt, nt, coarse); /* :397-404 */
  if (rc == 0 && nt > 0 && pe->aligned_source != NULL) rc = orc_estimate_fine_pose(pe, pe->aligned_source, pe->n_aligned, tgt, nt, fine); /* :412-418 */
  float pose[16];
  mul44f(coarse, fine, pose); /* :421, quirk Q4 */
  /* pose of the incoming source relative to the stored model, identity correspondences (:425-436) */
  float rigid[16];
  memcpy(rigid, kI4, sizeof kI4);
  {
    const int n = pe->n_model < ns ? pe->n_model : ns;
    if (n > 0) orc_umeyama(pe->cloud_model, src, n, pe->acc_mode, rigid);
  }
  mul44f(rigid, pose, pe->final_pose); /* :439 */
  memcpy(out_pose, pe->final_pose, sizeof pe->final_pose);
  memcpy(pe->last_coarse, coarse, sizeof coarse);
  memcpy(pe->last_fine, fine, sizeof fine);
  memcpy(pe->last_rigid, rigid, sizeof rigid);
  if (pe->aligned_source != NULL) { /* :441 */
    const int n = pe->n_aligned < ns ? pe->n_aligned : ns;
    memcpy(src, pe->aligned_source, sizeof(float) * 3 * (size_t)n);
  }
  if (fitness_score) *fitness_score = pe->fitness_score_fine; /* :444-445 */
  if (align_strength) *align_strength = pe->aligned_strength;
  return rc;
}
