/*
 * ope_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the registration hot path of
 * gopi-erabati/Object-Pose-Estimation (DetectAndLocalize): the vendored,
 * modified PCL ICP loop plus the un-vendored PCL 1.7.x primitives it calls.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or sample
 * scenes, and PCL/Eigen/FLANN are not installed here, so the reference itself
 * cannot be run.  This oracle is pinned only by analytic known-answer tests
 * and independent numpy/scipy cross-checks (tests/test_oracle_*.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call anything in this directory.  The product library
 * (object-pose-estimation_amd/libope_hip.so) never does.
 *
 * All 4x4 transforms are COLUMN-MAJOR float[16] (Eigen::Matrix4f layout;
 * T[12..14] is the translation), as in the reference
 * (DetectAndLocalize/src/rosinterface.cpp:435-437).
 * Point arrays are packed float xyz triples; normals likewise.
 */
#ifndef OPE_ORACLE_H
#define OPE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* kd-tree: exact k-NN / radius search.  Stands in for                 */
/* pcl::search::KdTree -> pcl::KdTreeFLANN -> flann::KDTreeSingleIndex */
/* (leaf_max_size 15, L2_Simple, eps 0), instantiated at               */
/* registration_mod.h:104-105, correspondence_estimation_mod.h:97-98,  */
/* poseestimator.cpp:151.  Distances returned are SQUARED L2, sorted   */
/* ascending for k-NN (PCL convention).                                */
/* ------------------------------------------------------------------ */
typedef struct orc_kdtree orc_kdtree;

orc_kdtree *orc_kdtree_build(const float *xyz, int n, int leaf_max);
void orc_kdtree_free(orc_kdtree *t);
/* For each of nq queries write k (idx,d2) pairs, ascending d2; slots past
 * the number found (n<k or non-finite query) get idx=-1, d2=INFINITY.
 * found[nq] (optional) receives the per-query count. */
void orc_kdtree_knn(const orc_kdtree *t, const float *q, int nq, int k,
                    int32_t *idx, float *d2, int32_t *found);
/* Radius search (d2 <= r*r, PCL/FLANN convention).  offsets[nq+1] is always
 * written.  If idx/d2 are non-NULL they receive up to cap entries in
 * per-query blocks, each block sorted ascending by d2 when `sorted` != 0
 * (FPFH uses an unsorted tree; order then follows tree traversal).
 * Returns the total number of neighbours. */
int64_t orc_kdtree_radius(const orc_kdtree *t, const float *q, int nq, float radius,
                          int sorted, int64_t *offsets, int32_t *idx, float *d2, int64_t cap);

/* Brute-force 1-NN, used to validate the tree in tests. */
void orc_bruteforce_nn(const float *tgt, int nt, const float *q, int nq, int32_t *idx, float *d2);

/* ------------------------------------------------------------------ */
/* Rigid transform estimation                                          */
/* pcl::registration::TransformationEstimationSVD (use_umeyama_=true)  */
/* -> pcl::umeyama / Eigen::umeyama(src,dst,false)  [uPCL, restated]   */
/* called at impl/icp_mod.hpp:243 and poseestimator.cpp:435.           */
/* acc_mode 0: float accumulation (Scalar=float, as PCL);              */
/* acc_mode 1: double accumulation, float result.                      */
/* Returns 0, or -1 if n < 1.                                          */
/* ------------------------------------------------------------------ */
int orc_umeyama(const float *src, const float *tgt, int n, int acc_mode, float T[16]);
/* Same from the 17 raw sums the GPU reduces:
 * S = { n, sum s (3), sum t (3), sum t_i s_j (9, row-major i,j), sum d2 }.
 * Sums are taken about `pivot` (subtracted from both s and t). */
int orc_umeyama_from_sums(const double S[17], const double pivot[3], float T[16]);

/* uPCL TransformationEstimationPointToPlaneLLS::estimateRigidTransformation on n paired points
 * (tgt_nrm = normals of the matched target points).  Returns 0, or -1 if the 6x6 system is singular. */
int orc_point_to_plane_lls(const float *src, const float *tgt, const float *tgt_nrm, int n, float T[16]);

/* uPCL TransformationEstimationPointToPlane (Levenberg-Marquardt over WarpPointRigid6D; the estimator BuildModel
 * installs, regmeshpcd.cpp:162,193) on n paired points: lm.c / lm_impl.inc.  precision 0 = float (the reference's
 * MatScalar), 1 = double.  x_out: (tx, ty, tz, qx, qy, qz); nfev_out: functor evaluations; status_out: Eigen's
 * LevenbergMarquardtSpace::Status.  Returns -1 (T = identity) for fewer than 4 pairs. */
int orc_point_to_plane_lm(const float *src, const float *tgt, const float *tgt_nrm, int n, int precision, float T[16],
                          double x_out[6], int *nfev_out, int *status_out);

/* 3x3 SVD (two-sided Jacobi via A^T A eigen-decomposition refinement),
 * A = U diag(s) V^T, s descending, row-major 3x3 arrays. */
void orc_svd3(const double A[9], double U[9], double s[3], double V[9]);

/* ------------------------------------------------------------------ */
/* Convergence criteria: DefaultConvergenceCriteria                    */
/* (default_convergence_criteria_mod.h:94-121,226-234 + uPCL           */
/* impl/default_convergence_criteria.hpp::hasConverged, restated).     */
/* ------------------------------------------------------------------ */
enum {
  ORC_CONV_NOT_CONVERGED = 0,
  ORC_CONV_ITERATIONS = 1,
  ORC_CONV_TRANSFORM = 2,
  ORC_CONV_ABS_MSE = 3,
  ORC_CONV_REL_MSE = 4,
  ORC_CONV_NO_CORRESPONDENCES = 5
};

typedef struct {
  int max_iterations;               /* 100  */
  int failure_after_max_iter;       /* 0    */
  double rotation_threshold;        /* 0.99999 */
  double translation_threshold;     /* 9e-8 */
  double mse_threshold_relative;    /* 1e-5 */
  double mse_threshold_absolute;    /* 1e-12 */
  int max_iterations_similar_transforms; /* 0 */
  /* state */
  int iterations_similar_transforms;
  double prev_mse, cur_mse;
  int state;
} orc_convergence;

void orc_convergence_init(orc_convergence *c);
/* One hasConverged() evaluation.  T = last incremental transform (col-major
 * float), mse = mean of correspondence distances of this iteration. */
int orc_convergence_step(orc_convergence *c, int iterations, const float T[16], double mse);

/* ------------------------------------------------------------------ */
/* ICP: IterativeClosestPoint::computeTransformation                   */
/* (impl/icp_mod.hpp:119-272), Registration::align                     */
/* (impl/registration_mod.hpp:176-219), getFitnessScore (:131-165),    */
/* getAlignStrength (icp_mod.h:249-260).                               */
/* ------------------------------------------------------------------ */
typedef struct {
  int max_iterations;              /* Registration default 10 (registration_mod.h:106) */
  double transformation_epsilon;   /* 0 */
  double euclidean_fitness_epsilon;/* -DBL_MAX */
  double max_corr_dist;            /* sqrt(DBL_MAX) */
  int use_reciprocal;              /* 0 */
  int min_correspondences;         /* 3 */
  int corr_mode;                   /* 0: 1-NN (correspondence_estimation_mod.hpp:127-213)
                                      1: normal shooting (…normal_shooting_weighted.hpp:107-145) */
  int k_normal_shooting;           /* 20 (poseestimator.cpp:246) */
  int use_surface_normal_rej;      /* CorrespondenceRejectorSurfaceNormal */
  double surface_normal_thr;       /* 0.7 (poseestimator.cpp:272) */
  int use_self_occluded_rej;       /* CorrespondenceRejectorSelfOccludedNormal */
  double self_occluded_thr;        /* 0.6 (poseestimator.cpp:291) */
  double mse_threshold_absolute;   /* 1e-12; <0 disables (throughput runs) */
  int failure_after_max_iter;      /* 0 */
  int acc_mode;                    /* umeyama accumulation, see orc_umeyama */
  int estimator;                   /* 0: TransformationEstimationSVD (poseestimator.cpp:306,341)
                                      2: TransformationEstimationPointToPlane (LM, lm.c): BuildModel's estimator
                                         (regmeshpcd.cpp:162,193); lm_precision 0 = float as PCL, 1 = double
                                      1: TransformationEstimationPointToPlaneLLS — the default estimator of
                                         IterativeClosestPointWithNormals (icp_mod.h:352-357), linearised
                                         point-to-plane; needs target normals.  (BuildModel selects the
                                         LM-based point-to-plane estimator, regmeshpcd.cpp:162,193: same cost
                                         function, non-linear solve — not restated.) */
  int lm_precision;                /* estimator 2: 0 = float (PCL), 1 = double */
  int transform_mode;              /* 0: incremental float transform of the working cloud
                                         each iteration (reference, icp_mod.hpp:246);
                                      1: final_T (composed in double) applied to the
                                         original source each iteration (device design) */
} orc_icp_params;

typedef struct {
  int iterations;
  int converged;
  int state;
  double last_mse;
  int n_corr;            /* post-rejection correspondences of the last iteration */
  double fitness;        /* getFitnessScore(DBL_MAX) after align */
  double align_strength; /* n_corr / (ns + nt) */
} orc_icp_result;

void orc_icp_default_params(orc_icp_params *p);

/* Returns 0 on success (including "not converged"), <0 on bad input.
 * T_hist (optional, max_iterations*16) receives final_T after each iteration.
 * corr_* (optional, capacity ns) receive the last iteration's correspondences. */
int orc_icp(const float *src_xyz, const float *src_nrm, int ns,
            const float *tgt_xyz, const float *tgt_nrm, int nt,
            const float guess[16], const orc_icp_params *p,
            float out_T[16], orc_icp_result *res,
            float *T_hist, int32_t *corr_q, int32_t *corr_m, float *corr_d2);

/* The same loop with the reference's injected "fixed correspondences" (icp_mod.h:268, icp_mod.hpp:150-151,210-224;
 * correspondence_estimation_mod.hpp:134-162): see icp.c.  corr_* need room for ns + 2 * n_fixed entries. */
int orc_icp_fixed(const float *src_xyz, const float *src_nrm, int ns, const float *tgt_xyz, const float *tgt_nrm, int nt,
                  const float guess[16], const orc_icp_params *p, const int32_t *fixed_q, const int32_t *fixed_m, int n_fixed,
                  float out_T[16], orc_icp_result *res, float *T_hist, int32_t *corr_q, int32_t *corr_m, float *corr_d2);

/* Number of threads the per-query searches of orc_icp / orc_icp_fixed run on (default 1).  Lists, rejectors, sums and
 * transforms stay on one thread in query order: results do not depend on the setting. */
void orc_icp_set_threads(int n);

/* Registration::getFitnessScore(max_range) for an arbitrary transform. */
double orc_fitness(const float *src_xyz, int ns, const float *tgt_xyz, int nt,
                   const float T[16], double max_range, int *n_used);

/* One correspondence pass + the 17 sums (for the sharded / multi-rank tests):
 * src is transformed by T (float math), 1-NN'd into tgt, thresholded by
 * max_corr_dist, and S (see orc_umeyama_from_sums) accumulated about pivot. */
void orc_icp_partial_sums(const float *src_xyz, int ns, const orc_kdtree *tgt_tree,
                          const float *tgt_xyz, const float T[16], double max_corr_dist,
                          const double pivot[3], double S[17]);
/* the same over n_threads contiguous slices of the source, one thread each (the all-core CPU figure of bench.py) */
void orc_icp_partial_sums_mt(const float *src_xyz, int ns, const orc_kdtree *tgt_tree, const float *tgt_xyz, const float T[16],
                             double max_corr_dist, const double pivot[3], int n_threads, double S[17]);

/* pcl::transformPointCloud (float math): out = R*p + t; non-finite points
 * are copied through unchanged (icp_mod.hpp:71-72,104-105). */
void orc_transform_points(const float *in, int n, const float T[16], float *out);
void orc_transform_normals(const float *in, int n, const float T[16], float *out);

/* ------------------------------------------------------------------ */
/* Features (uPCL, restated): NormalEstimation, FPFHEstimation,         */
/* UniformSampling, SampleConsensusInitialAlignment.                    */
/* ------------------------------------------------------------------ */
/* NormalEstimation::compute with k-NN (poseestimator.cpp:151-156).
 * out_nrm n*3, out_curv n.  Viewpoint vp (0,0,0 in the reference). */
void orc_normals_knn(const float *xyz, int n, int k, const float vp[3], float *out_nrm, float *out_curv);

/* pcl::computePairFeatures; returns 0 if rejected (f4==0 or |v|==0). */
int orc_pair_features(const float p1[3], const float n1[3], const float p2[3], const float n2[3],
                      float *f1, float *f2, float *f3, float *f4);

/* FPFHEstimation::compute, radius search, 11+11+11 bins (poseestimator.cpp:121-125).
 * out n*33.  Also optionally returns SPFH (n*33) and mean neighbour count. */
void orc_fpfh(const float *xyz, const float *nrm, int n, float radius, float *out33,
              float *spfh33_opt, double *mean_neighbours_opt);

/* UniformSampling::compute (PCL 1.7 keypoints API, poseestimator.cpp:141-145)
 * with DETERMINISTIC output order (ascending voxel key) instead of
 * boost::unordered_map order.  Returns count; out_idx capacity n. */
int orc_uniform_sampling(const float *xyz, int n, float leaf, int32_t *out_idx);

/* Filters either side of the path (filters.c): removeNaNFromPointCloud (poseestimator.cpp:192-194),
 * PassThrough on x/y/z with inclusive limits (processingpcd.cpp:8-36), VoxelGrid centroids
 * (processingpcd.cpp:39-52; -1 = "leaf size too small", PCL hands back the input).
 * Index outputs have capacity n, out_xyz capacity n*3; the count is returned. */
int orc_remove_nan(const float *xyz, int n, int32_t *out_idx);
int orc_pass_through(const float *xyz, int n, const float lo[3], const float hi[3], int32_t *out_idx);
int orc_voxel_grid(const float *xyz, int n, const float leaf[3], float *out_xyz);
/* the same with the packed colour of PointXYZRGB carried (channel-wise float mean, truncated; processingpcd.cpp:44-59) */
int orc_voxel_grid_rgb(const float *xyz, const uint32_t *rgb, int n, const float leaf[3], float *out_xyz, uint32_t *out_rgb);
/* pcl::StatisticalOutlierRemoval (ProcessingPcd::getOutlierRemove, processingpcd.cpp:62-77: setMeanK(30),
 * setStddevMulThresh(threshold)): indices of the inliers in input order; out_dist (optional) the per-point mean distance. */
int orc_statistical_outlier_removal(const float *xyz, int n, int mean_k, double stddev_mul, int32_t *out_idx, float *out_dist);

/* SAC-IA error metric for one hypothesis: sum of TruncatedError(1-NN d2). */
double orc_sacia_error(const float *src_xyz, int ns, const orc_kdtree *tgt_tree,
                       const float T[16], double corr_dist_threshold);

/* SampleConsensusInitialAlignment::computeTransformation (poseestimator.cpp:50-64).
 * RNG is an injectable 64-bit LCG stream (PCL uses unseeded rand()).
 * If forced_samples != NULL it holds n_iter*nr_samples source indices followed
 * by n_iter*nr_samples target indices and no RNG is used. */
int orc_sacia(const float *src_xyz, const float *src_feat33, int ns,
              const float *tgt_xyz, const float *tgt_feat33, int nt,
              int n_iter, int nr_samples, int k_corr, double max_corr_dist, float min_sample_dist,
              uint64_t seed, const int32_t *forced_samples,
              float out_T[16], double *best_err, int32_t *best_iter);

/* 33-D feature k-NN (brute force), used by SAC-IA findSimilarFeatures. */
void orc_feature_knn(const float *feat33, int n, const float *q33, int nq, int k, int32_t *idx, float *d2);

/* ------------------------------------------------------------------ */
/* The L2 composite: PoseEstimator (poseestimator.cpp:3-448), pose.c.  */
/* ------------------------------------------------------------------ */
typedef struct {
  /* state that crosses frames (poseestimator.h:50-53) */
  int first_time_pose;
  double fitness_score_fine, aligned_strength;
  float final_pose[16];
  float *aligned_source; int n_aligned;
  float *cloud_model;    int n_model;
  /* knobs of the restatement */
  uint64_t sacia_seed;   /* the k-th coarse call uses the stream sacia_seed + k (PCL: unseeded rand()) */
  int coarse_calls;
  int use_self_occluded; /* CorrespondenceRejectorSelfOccludedNormal 0.6 (poseestimator.cpp:290-291,336); default 0, SURVEY Q3 */
  int acc_mode;          /* umeyama accumulation (orc_umeyama): 1 = double, as the device; 0 = float, as PCL */
  int transform_mode;    /* orc_icp_params.transform_mode: 1 = as the device, 0 = as PCL */
  /* diagnostics of the last call */
  float last_coarse[16], last_fine[16], last_rigid[16];
  double last_sacia_error; int last_sacia_best;
  int last_n_src_keys, last_n_tgt_keys, last_n_fine_src, last_n_fine_tgt;
  int last_icp_iterations, last_icp_state, last_icp_n_corr;
} orc_pose_estimator;

void orc_pose_estimator_init(orc_pose_estimator *pe);
void orc_pose_estimator_free(orc_pose_estimator *pe);
/* subSampleAndCalculateNormals (:131-158); outputs malloc'ed, caller frees; returns the number of key points */
int orc_sub_sample_and_normals(const float *xyz, int n, float leaf, float **out_xyz, float **out_nrm, float **out_curv);
int orc_estimate_coarse_pose(orc_pose_estimator *pe, const float *src, int ns, const float *tgt, int nt, float out_T[16]);
int orc_estimate_fine_pose(orc_pose_estimator *pe, float *src_inout, int ns, const float *tgt, int nt, float out_T[16]);
int orc_estimate_final_pose(orc_pose_estimator *pe, float *src_inout, int ns, const float *tgt, int nt, float out_pose[16],
                            double *fitness_score, double *align_strength);

#ifdef __cplusplus
}
#endif
#endif /* OPE_ORACLE_H */
