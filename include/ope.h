/*
 * ope.h — C ABI of libope_hip.so, the MI355X-native (gfx950) replacement for the
 * registration hot path of gopi-erabati/Object-Pose-Estimation (DetectAndLocalize).
 *
 * The reference has no FFI of its own: its operator API is the PCL class-template
 * protocol.  Each entry point below names the reference call it replaces
 * (paths relative to the reference root; vPCL = the vendored include/pcl/registration/ headers).
 * The C++ façade include/ope/pcl_compat.hpp re-creates those PCL call shapes on
 * top of this ABI; INTEGRATION.md shows the re-pointing of poseestimator.cpp.
 *
 * Conventions
 *  - every function returns OPE_OK (0) or a negative OPE_E* code; no exceptions
 *    cross the ABI; ope_last_error(ctx) gives the message of the last failure.
 *  - 4x4 transforms are COLUMN-MAJOR float[16] (Eigen::Matrix4f memory layout,
 *    translation in [12..14]; reference: rosinterface.cpp:435-437).
 *  - host buffers are only read/written during the call; device copies are owned
 *    by the ope_cloud / ope_index handles.  Handles are not thread-safe; calls on
 *    one ope_ctx must be externally serialised (reference: one PoseEstimator per
 *    process, rosinterface.h:54).
 *  - this library is GPU-only.  There is no CPU fallback: ope_ctx_create fails
 *    with OPE_ENODEV when no HIP device is present.
 */
#ifndef OPE_H
#define OPE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OPE_ABI_VERSION 5

enum {
  OPE_OK = 0,
  OPE_EINVAL = -1,  /* bad argument (NULL handle, n == 0 where forbidden, …) */
  OPE_ENODEV = -2,  /* no HIP device / bad ordinal */
  OPE_EHIP = -3,    /* a HIP runtime call failed */
  OPE_ENOMEM = -4,
  OPE_ESTATE = -5,  /* call out of sequence (e.g. ope_icp_step without begin) */
  OPE_ECOMM = -6,   /* RCCL failure */
  OPE_EEMPTY = -7,  /* empty target cloud (registration_mod.hpp:60-64) */
  OPE_ERANGE = -8   /* voxel index would overflow 32 bits: PCL warns "leaf size is too small" and returns its input */
};

typedef struct ope_ctx ope_ctx;
typedef struct ope_cloud ope_cloud;
typedef struct ope_index ope_index;

/* ---------------- context ---------------- */
int ope_abi_version(void);
int ope_device_count(void);
/* One context per GPU (one process per GPU in multi-GPU runs). */
int ope_ctx_create(ope_ctx **out, int device_ordinal);
void ope_ctx_destroy(ope_ctx *ctx);
/* Run on an externally owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream);
 * NULL restores the context's own stream. */
int ope_ctx_set_stream(ope_ctx *ctx, void *hip_stream);
int ope_ctx_sync(ope_ctx *ctx);
/* Named roctx ranges (rocprofv3 --marker-trace) around the host side of the path: icp_iter {nn, reduce}, normals,
 * fpfh_spfh, fpfh_weight, sacia, index_build, uniform_sampling.  Off by default.  The reference's own instrumentation is
 * pcl::ScopeTime("Initial Alignment" / "Final Alignment") (poseestimator.cpp:61,349): the facade keeps those names. */
int ope_ctx_set_tracing(ope_ctx *ctx, int on);
/* Bound of every device-side wait of the overlapped update launches (ope_icp_params.update_launch), seconds, 0 .. 40; default 2.
 * A launch whose partner has not reported within the bound gives up and the run resumes in line at the next ope_icp_poll /
 * ope_icp_end (nothing is lost but the time waited; the context launches in line from then on).  Raise it for launches that
 * legitimately take longer than the bound (a source of hundreds of millions of points); a process that shares its GPU with other
 * work it cannot predict should rather ask for OPE_UPDATE_IN_LINE.  A bound of a microsecond makes every overlapped update give up at once
 * (how the test of the recovery path forces it). */
int ope_ctx_set_wait_limit(ope_ctx *ctx, double seconds);
const char *ope_last_error(const ope_ctx *ctx);

/* Measurement hook for the coarse-stage and filter kernels (normals_kernel, spfh_kernel, fpfh_kernel, feature_knn_kernel,
 * sacia_error_kernel, sor_mean_distance_kernel): while on, every launch is bracketed by HIP events on the launch stream and
 * recorded with its ALGORITHMIC bytes (normals N(12+12k+16); SPFH N(24+24m+132); FPFH N(136m+132); SAC-IA
 * 24 N_s per hypothesis; SOR N(12+12k+4)).  ope_profile_kernels_read synchronises and returns one record per kernel name,
 * summed over its launches since ope_profile_kernels(ctx, 1). */
typedef struct {
  char name[32];
  double ms;
  int launches;
  double algorithmic_bytes;
} ope_kernel_time;
int ope_profile_kernels(ope_ctx *ctx, int on);
int ope_profile_kernels_read(ope_ctx *ctx, ope_kernel_time *out, size_t cap, size_t *n_out);

/* ---------------- clouds ---------------- */
/* Upload n points from an array of structs: xyz floats at base + i*stride + xyz_off,
 * optional normal floats at normal_off (pass -1 for none).  Works directly on
 * pcl::PointXYZ (stride 16, xyz_off 0), pcl::PointXYZRGBNormal (stride 48,
 * normal_off 16) or packed float[3] (stride 12).  Replaces setInputSource /
 * setInputCloud (registration_mod.h:197-201; poseestimator.cpp:121,155,312).
 * Points are re-ordered on the device along a Morton curve (the permutation is
 * kept; all outputs are reported in ORIGINAL indices).  Non-finite points are
 * kept in the index space but never produce correspondences (icp_mod.hpp:71-72). */
int ope_cloud_upload(ope_ctx *ctx, const void *base, size_t n, size_t stride_bytes, size_t xyz_off,
                     ptrdiff_t normal_off, ope_cloud **out);
/* Attach / replace normals (n*3 packed floats, original order). */
int ope_cloud_set_normals(ope_ctx *ctx, ope_cloud *cloud, const float *normals_xyz);
/* pcl::transformPointCloud(a, ., T_a) followed by operator+= (BuildModel regmeshpcd.cpp:203,254: cloudTemp = aligned + target),
 * built on the device: out holds T_a * a (T_a may be NULL: identity) followed by b, ORIGINAL indices a's then b's.  Neither
 * input travels through the host; normals are not carried (re-estimated per pair in the reference, :72-90). */
int ope_cloud_concat(ope_ctx *ctx, const ope_cloud *a, const float T_a[16], const ope_cloud *b, ope_cloud **out);
/* xyz of a cloud in ORIGINAL order (n*3 floats): the way out for clouds made by ope_cloud_concat. */
int ope_cloud_download(ope_ctx *ctx, const ope_cloud *cloud, float *out_xyz);
/* A new cloud from n ORIGINAL indices of `cloud` (host array, any order, repeats allowed), gathered on the device: the new
 * cloud's original order is the order of idx; normals attached to `cloud` are carried.  What `cloud[idx]` would be after an
 * upload, without the trip through the host (the hand-over between the stages in front of the path:
 * rosinterface.cpp:212-213 -> poseestimator.cpp:141-156). */
int ope_cloud_select(ope_ctx *ctx, const ope_cloud *cloud, const int32_t *idx, size_t n, ope_cloud **out);
size_t ope_cloud_size(const ope_cloud *cloud);
void ope_cloud_free(ope_cloud *cloud);

/* ---------------- search index over a target cloud ---------------- */
typedef struct {
  int leaf_size; /* max points per leaf bucket of the OBB tree (default 16) */
  int grid;      /* 1 (default): 1-NN ICP runs also use a uniform grid over the points ("radix-bucketed" search): queries
                    whose previous match is close are answered from a few cell runs, all others by the tree.  A run whose
                    source turns out to hold more than 3 % of queries far from the target (measured on the device after
                    the first iterations) continues on the tree-only kernel, which is faster for that mix.
                    0: tree only.  2: grid kernel always. */
  float grid_fill;      /* target number of points per occupied grid cell (0 = default) */
  int grid_max_cells;   /* upper bound on the number of grid cells, 4 B each (0 = default: the table must stay L2-resident) */
} ope_index_params;
void ope_index_default_params(ope_index_params *p);
/* Replaces the kd-tree build of Registration::initCompute (registration_mod.hpp:80-84)
 * / pcl::search::KdTree::setInputCloud (poseestimator.cpp:151-152).
 * Returns OPE_EEMPTY for an empty (or all-non-finite) target. */
int ope_index_build(ope_ctx *ctx, const ope_cloud *target, const ope_index_params *params, ope_index **out);
void ope_index_free(ope_index *index);

/* Exact searches, results in ORIGINAL target indices, squared L2 distances
 * (pcl::search::KdTree::nearestKSearch / radiusSearch semantics).  `T` (optional)
 * is applied to the queries first (float math).  Host output buffers.
 *   nn:     out_idx[nq], out_d2[nq]; idx = -1, d2 = +inf for non-finite queries.
 *   knn:    out_idx[nq*k], out_d2[nq*k], ascending d2, -1/+inf padded.
 *   radius: counts[nq] always; if out_idx/out_d2 non-NULL, up to max_nn per query
 *           at stride max_nn (ascending d2). */
int ope_nn_search(ope_ctx *ctx, const ope_cloud *queries, const ope_index *index, const float *T,
                  int32_t *out_idx, float *out_d2);
int ope_knn_search(ope_ctx *ctx, const ope_cloud *queries, const ope_index *index, const float *T, int k,
                   int32_t *out_idx, float *out_d2);
int ope_radius_search(ope_ctx *ctx, const ope_cloud *queries, const ope_index *index, float radius, int max_nn,
                      int32_t *counts, int32_t *out_idx, float *out_d2);

/* ---------------- ICP ---------------- */
enum { /* DefaultConvergenceCriteria::ConvergenceState, default_convergence_criteria_mod.h:73-81 */
  OPE_CONV_NOT_CONVERGED = 0,
  OPE_CONV_ITERATIONS = 1,
  OPE_CONV_TRANSFORM = 2,
  OPE_CONV_ABS_MSE = 3,
  OPE_CONV_REL_MSE = 4,
  OPE_CONV_NO_CORRESPONDENCES = 5
};

enum { OPE_CORR_NEAREST = 0, OPE_CORR_NORMAL_SHOOTING = 1 };
/* transformation estimation: SVD/Umeyama (poseestimator.cpp:306,341), or the linearised point-to-plane
 * estimator that IterativeClosestPointWithNormals defaults to (icp_mod.h:352-357; needs TARGET normals).
 * OPE_EST_POINT_TO_PLANE_LM: pcl::registration::TransformationEstimationPointToPlane, the Levenberg-Marquardt estimator
 * BuildModel installs (regmeshpcd.cpp:162,193): same cost, minimised over (t, quaternion) with Eigen's LM logic on a
 * forward-difference Jacobian and float tolerances.  The residual is linear in the warp matrix, so one device pass per ICP
 * iteration reduces the 91 sums every functor evaluation is a quadratic form of, and the minimisation runs in the launch
 * that also updates the transform: nothing synchronises the host (ope_icp_run / ope_icp_iterate only — the step-wise
 * accumulate / update pair exchanges 17 or 44 sums, not these; needs TARGET normals). */
enum { OPE_EST_SVD = 0, OPE_EST_POINT_TO_PLANE_LLS = 1, OPE_EST_POINT_TO_PLANE_LM = 2 };
#define OPE_NUM_SUMS 17     /* {n, Σs, Σt, Σ t sᵀ, Σd²} */
#define OPE_NUM_SUMS_MAX 44 /* + upper triangle of AᵀA (21) and Aᵀb (6) for point-to-plane */

typedef struct {
  /* Registration defaults, registration_mod.h:106-118 */
  int max_iterations;               /* 10 */
  double transformation_epsilon;    /* 0 */
  double euclidean_fitness_epsilon; /* -DBL_MAX */
  double max_corr_dist;             /* sqrt(DBL_MAX) */
  int min_correspondences;          /* 3 */
  int use_reciprocal;               /* 0 */
  /* correspondence estimation: 1-NN (correspondence_estimation_mod.hpp:127-213) or
   * normal shooting over the k nearest (…normal_shooting_weighted.hpp:107-145) */
  int corr_mode;
  int k_normal_shooting; /* 20 (poseestimator.cpp:246) */
  /* rejectors, applied in this order (poseestimator.cpp:334-337) */
  int use_surface_normal_rej;  /* CorrespondenceRejectorSurfaceNormal, score correspondence_rejection_mod.h:368-376 */
  double surface_normal_thr;   /* 0.7 (poseestimator.cpp:272) */
  int use_self_occluded_rej;   /* correspondence_rejection_mod.h:382-391; opt-in, see SURVEY Q3 */
  double self_occluded_thr;    /* 0.6 (poseestimator.cpp:291) */
  /* DefaultConvergenceCriteria knobs reachable through getConvergeCriteria() */
  double mse_threshold_absolute; /* 1e-12; negative disables (fixed-length throughput runs) */
  int failure_after_max_iter;    /* 0 */
  /* host polling period for the on-device convergence flag (iterations); 0 = only at the end */
  int check_every;
  int estimator; /* OPE_EST_* */
  /* 0 (default): every block adds its partial sums into the run's sums with fp64 atomics: the addition order, and with
   * it the last bit of the sums (~1e-9 in the final transform after 100 iterations), varies from run to run.
   * 1: one row of partial sums per block and a fixed-tree reduction, the tree kernel with its chunks in natural order (no
   * cost-sorted schedule, no grid kernel: both follow measured times): bit-reproducible from run to run on one GPU model,
   * at the price of the schedule (launches that fill the GPU take longer). */
  int deterministic_sums;
  /* Which walk the OBB-tree kernel uses for the 64-query chunks of a 1-NN run.  OPE_WALK_AUTO (0, default): launches that
   * fill the GPU take one packet walk per coherent chunk and private per-lane walks for the rest, smaller launches
   * per-lane walks only.  OPE_WALK_LANE (1) / OPE_WALK_PACKET (2) force the instantiation (every walk is exact: the choice
   * moves time, and lets a test pin each kernel by name, see ope_icp_kernel_launches). */
  int tree_walk;
  /* How the per-iteration update step (Umeyama / Cholesky lane + convergence test) is launched in ope_icp_run /
   * ope_icp_iterate.  OPE_UPDATE_OVERLAPPED (0, default): on a stream of its own next to the accumulate launch it follows,
   * waiting on the device for that launch's blocks, while the next accumulate launch is already being dispatched and its
   * blocks wait for the update's word: one kernel boundary per iteration instead of two around a 64-thread launch.  Taken by
   * plain 1-NN runs of one rank (not: normal shooting, reciprocal, deterministic_sums, the LM estimator, fixed correspondences,
   * sharded runs), the others launch in line whatever this says.  Every device-side wait is bounded (2 s); a run that hits
   * the bound (the GPU's block slots held by other work, a tool that serialises dispatches) resumes in line at the next
   * ope_icp_poll / ope_icp_end with nothing lost but the time, and the context stays in line from then on; a process under a
   * counter-collecting profiler (rocprofv3 --pmc: ROCPROF_COUNTER_COLLECTION in the environment) launches in line from the start.
   * OPE_UPDATE_IN_LINE (1): accumulate -> update -> accumulate on the one stream, as in rounds 1-2 (a profiler that
   * serialises dispatches, e.g. rocprofv3 --pmc, wants this).  Same arithmetic either way. */
  int update_launch;
  /* Skip certificates of the plain 1-NN search (what CorrespondenceEstimation::determineCorrespondences,
   * impl/correspondence_estimation_mod.hpp:165-177, recomputes from scratch every iteration).  Late in a run a query's nearest
   * neighbour rarely changes, and that can be PROVEN without a search.  A certificate is the outcome of one 6-nearest walk from
   * where the query was then (q_ref): its five nearest target points and a lower bound L — the sixth's distance — on its distance
   * to every other one.  Wherever the query is later, every non-candidate is at least L - |q - q_ref| away; while the nearest of
   * the five candidates, re-measured from where the query is now, stays strictly below that (past every fp32 rounding) and
   * strictly below the other four, it is the unique nearest neighbour: same index, same d2, bit for bit, and no walk.  Every
   * launch still writes every correspondence and adds every term of the sums.
   * OPE_CERT_AUTO (0, default): launches keep certificates from the iteration on whose update moves no scene point by more than
   * 1/24 of the target's point spacing (1/512 for a run that starts on the tree kernel because more than 3 % of its queries lie
   * far outside the target: clutter, whose walks set the pace until it can hold certificates too); decided on the device, no host
   * round trip.  A query builds one when the slack it can expect — read off its previous distance — is worth 24 launches of the
   * scene's current displacement.
   * OPE_CERT_OFF (1): never.  OPE_CERT_ALWAYS (2): from the first launch (tests).  Plain 1-NN runs, tree and grid kernel
   * (not: reciprocal, normal shooting, deterministic_sums); exact in every mode — the choice moves time only. */
  int skip_certificates;
} ope_icp_params;
enum { OPE_WALK_AUTO = 0, OPE_WALK_LANE = 1, OPE_WALK_PACKET = 2 };
enum { OPE_UPDATE_OVERLAPPED = 0, OPE_UPDATE_IN_LINE = 1 };
enum { OPE_CERT_AUTO = 0, OPE_CERT_OFF = 1, OPE_CERT_ALWAYS = 2 };

typedef struct {
  int iterations;        /* nr_iterations_ */
  int converged;         /* hasConverged() */
  int state;             /* OPE_CONV_* */
  double last_mse;       /* correspondences_cur_mse_ */
  int64_t n_corr;        /* post-rejection correspondences of the last iteration */
  double align_strength; /* getAlignStrength(): n_corr / (N_src + N_tgt), icp_mod.h:249-260
                            (N_src = the full source size set with ope_icp_set_global_sizes in sharded runs) */
} ope_icp_result;

void ope_icp_default_params(ope_icp_params *p);

/* IterativeClosestPoint::setFixedCorrespondences (vPCL icp_mod.h:268; getFixedCorrespondences :276, clearCorrespondences
 * :281) — the reference's injection of given pairs into every iteration, unused by its own programs.  index_query /
 * index_match: ORIGINAL indices into `src` and into the cloud the target index was built from; n = 0 clears.  They stay set
 * for every later run over the same source cloud and a target of the same size, and take part as the reference has them:
 * 1-NN estimation lists every given pair in front of the searched ones whatever its distance, with the distance field
 * (squared distance) * 1e10 — which therefore also enters the MSE of the convergence test — (correspondence_estimation_mod.hpp:
 * 134-162); normal shooting lists none (…normal_shooting_weighted.hpp:81-101); listed pairs pass the rejectors like any
 * other; then the FIRST rejector alone is applied to the given pairs and the survivors are appended, a second time for those
 * already listed (icp_mod.hpp:210-224; only when a rejector is installed; with both rejectors of ope_icp_params on, "the first"
 * is the surface-normal one — the params fix the order the reference's programs add them in, poseestimator.cpp:334-336).  SVD estimator only; not with reciprocal
 * correspondences; in a sharded run set them on exactly ONE rank (indices into that rank's shard): every rank that holds
 * pairs adds them to the sums that are then summed over the ranks.  ope_icp_correspondences lists the searched pairs only.  Runs with fixed
 * correspondences launch their update in line. */
int ope_icp_set_fixed_correspondences(ope_ctx *ctx, const ope_cloud *src, const ope_cloud *tgt_cloud, const int32_t *index_query,
                                      const int32_t *index_match, size_t n);
/* The given pairs as the LAST iteration of the last run saw them, in the order they were given (any output may be NULL;
 * *n = number of pairs set): distance[f] = the value the reference writes back into the caller's list through the pointer,
 * every iteration (impl/correspondence_estimation_mod.hpp:150-161: squared distance * 1e10 as float; under normal shooting the
 * squared distance to the source normal's line, impl/correspondence_estimation_normal_shooting_weighted.hpp:81-101);
 * listed[f] = 1 if the pair stands in the final correspondence list in front of the searched pairs (1-NN estimation, through
 * every rejector); appended[f] = 1 if it stands behind them once more (first rejector alone, impl/icp_mod.hpp:210-224).
 * The reference's correspondences_ = listed pairs, ope_icp_correspondences' pairs, appended pairs. */
int ope_icp_fixed_correspondences(ope_ctx *ctx, float *distance, int32_t *listed, int32_t *appended, size_t cap, size_t *n);

/* How many accumulate launches of the current (or last) run each search kernel served: the bucketed grid kernel, the
 * OBB-tree kernel in its per-lane and in its packet instantiation, the k-NN (normal shooting) kernel.  A run may move
 * between kernels (ope_index_params.grid = 1); tests use this to assert which kernel their comparison exercised. */
enum { OPE_KERNEL_GRID = 0, OPE_KERNEL_TREE_LANE = 1, OPE_KERNEL_TREE_PACKET = 2, OPE_KERNEL_KNN = 3, OPE_KERNEL_KINDS = 4 };
int ope_icp_kernel_launches(const ope_ctx *ctx, int64_t counts[OPE_KERNEL_KINDS]);
/* How many update steps of the current (or last) run were launched overlapped (ope_icp_params.update_launch). */
int64_t ope_icp_overlapped_updates(const ope_ctx *ctx);
/* How many runs of this context had an overlapped update launch give up its bounded wait (ope_ctx_set_wait_limit) and were
 * resumed in line on the same pose (0 unless the GPU was held up by something else for that long).  Results are unaffected;
 * TIMINGS are not: the launches enqueued behind the launch that gave up return at once and their iterations are enqueued
 * again, in line, by the next ope_icp_poll / ope_icp_end — a benchmark that brackets ope_icp_iterate with stream
 * synchronisation alone must check this counter (bench.py does, and measures again). */
int ope_icp_update_fallbacks(const ope_ctx *ctx);
/* Skip certificates of the run in progress (ope_icp_params.skip_certificates; synchronises the stream): out[0] = queries answered
 * from their certificate, summed over the run's launches; out[1] = accumulate launches that kept certificates; out[2] = 1 if the
 * run has reached the stage where it keeps them; out[3] = the last update's largest scene displacement in nanometres (what the
 * automatic mode compares with 1/24 of the target's point spacing). */
int ope_icp_certificate_stats(ope_ctx *ctx, int64_t out[4]);

/* Registration::align(output, guess) -> IterativeClosestPoint::computeTransformation
 * (registration_mod.hpp:176-219, icp_mod.hpp:119-272).  guess may be NULL (identity).
 * out_T receives getFinalTransformation(). */
int ope_icp_run(ope_ctx *ctx, const ope_cloud *src, const ope_index *tgt, const float *guess,
                const ope_icp_params *params, float out_T[16], ope_icp_result *result);

/* Step-wise form of the same loop, for one-process-per-GPU drivers that put a
 * collective between the local reduction and the transform update:
 *   begin -> { accumulate -> [all-reduce 17 doubles at ope_icp_sums_device()] -> update } * -> end
 * All launches go to the context stream; nothing synchronises the host except
 * ope_icp_poll / ope_icp_end. */
int ope_icp_begin(ope_ctx *ctx, const ope_cloud *src, const ope_index *tgt, const float *guess,
                  const ope_icp_params *params);
int ope_icp_accumulate(ope_ctx *ctx);
/* Device pointer to the fp64 sums {n, Σs[3], Σt[3], Σ t sᵀ[9], Σd²} (about the index pivot), followed for the
 * point-to-plane estimator by 21 + 6 normal-equation sums: all-reduce OPE_NUM_SUMS doubles (OPE_NUM_SUMS_MAX with
 * OPE_EST_POINT_TO_PLANE_LLS); the library reads and writes exactly that many. */
void *ope_icp_sums_device(ope_ctx *ctx);
int ope_icp_update(ope_ctx *ctx);
/* Use a caller-owned device buffer for the sums (e.g. a torch tensor that torch.distributed all-reduces):
 * OPE_NUM_SUMS doubles, OPE_NUM_SUMS_MAX for runs with OPE_EST_POINT_TO_PLANE_LLS; NULL restores the internal buffer. */
int ope_icp_set_sums_buffer(ope_ctx *ctx, void *device_ptr);
/* Enqueue n whole iterations (accumulate -> [RCCL all-reduce if ope_comm_init_rank was called] -> update)
 * without synchronising the host.  The accumulate launches ADD into the sums and ope_icp_update leaves them at
 * zero: in the stepwise form every ope_icp_accumulate must be followed by one ope_icp_update. */
int ope_icp_iterate(ope_ctx *ctx, int n_iterations);
int ope_icp_poll(ope_ctx *ctx, ope_icp_result *result); /* syncs the stream */
/* getFinalTransformation() of a run in progress (polls the device state): the transform after the iterations
 * enqueued so far. */
int ope_icp_current_transform(ope_ctx *ctx, float out_T[16]);
int ope_icp_end(ope_ctx *ctx, float out_T[16], ope_icp_result *result);
/* In sharded runs: the sizes getAlignStrength divides by (defaults: local sizes). */
int ope_icp_set_global_sizes(ope_ctx *ctx, int64_t n_src_total, int64_t n_tgt_total);

/* Measurement hook: bracket up to max_launches launches of the accumulate kernel (the dominant
 * kernel) with HIP events on the launch stream; 0 disables.  ope_icp_profile_read synchronises and
 * returns the summed kernel time and the number of launches timed since ope_icp_profile. */
int ope_icp_profile(ope_ctx *ctx, int max_launches);
int ope_icp_profile_read(ope_ctx *ctx, double *total_ms, int *n_launches);
/* The same per launch: ms[i] = duration of the i-th timed launch (at most cap of them), *n_out their number. */
int ope_icp_profile_launches(ope_ctx *ctx, float *ms, size_t cap, size_t *n_out);

/* Last iteration's post-rejection correspondences, compacted in query order
 * (pcl::Correspondences: index_query, index_match, distance = squared L2).  OPE_ESTATE when there is no finished
 * run on this context, or its source cloud has been freed since. */
int ope_icp_correspondences(ope_ctx *ctx, int32_t *index_query, int32_t *index_match, float *distance, size_t cap,
                            size_t *n);

/* getLastIncrementalTransformation() (transformation_, registration_mod.h): the incremental transform of the last
 * iteration of the run that ope_icp_end / ope_icp_poll last read back. */
int ope_icp_last_incremental(ope_ctx *ctx, float out_T[16]);

/* CorrespondenceRejector::getRemainingCorrespondences for n GIVEN correspondences (stand-alone use; inside the ICP loop the
 * same predicates are fused into the search kernel).  a, b: n packed float triples.
 *   OPE_REJ_SURFACE_NORMAL: a = source normal, b = matched target normal; keep iff a . b > threshold
 *                           (correspondence_rejection_mod.h:368-376)
 *   OPE_REJ_SELF_OCCLUDED:  a = source normal, b = source point; keep iff a . (-b / |b|) > threshold (:382-391) */
enum { OPE_REJ_SURFACE_NORMAL = 0, OPE_REJ_SELF_OCCLUDED = 1 };
int ope_reject_pairs(ope_ctx *ctx, int kind, const float *a, const float *b, size_t n, double threshold, unsigned char *keep);

/* Registration::getFitnessScore(max_range), registration_mod.hpp:131-165.
 * sum_out/n_out (optional) expose the partial sums for sharded runs. */
int ope_fitness(ope_ctx *ctx, const ope_cloud *src, const ope_index *tgt, const float T[16], double max_range,
                double *score, double *sum_out, int64_t *n_out);

/* registration::TransformationEstimationSVD::estimateRigidTransformation(src, tgt, correspondences, T)
 * (called stand-alone at poseestimator.cpp:429-435 with identity correspondences over the model).
 * src_xyz / tgt_xyz: n already-paired points, packed float[3] each.  Needs n >= 1. */
int ope_rigid_transform_svd(ope_ctx *ctx, const float *src_xyz, const float *tgt_xyz, size_t n, float out_T[16]);

/* pcl::transformPointCloud on the host copy of a result (float math): out = T * in. */
int ope_transform_cloud(ope_ctx *ctx, const ope_cloud *cloud, const float T[16], float *out_xyz);

/* ---------------- native RCCL path (optional; torch.distributed drivers use the step-wise API) ---------------- */
#define OPE_COMM_ID_BYTES 128
int ope_comm_get_unique_id(char id[OPE_COMM_ID_BYTES]);
int ope_comm_init_rank(ope_ctx *ctx, const char id[OPE_COMM_ID_BYTES], int nranks, int rank);
int ope_comm_destroy(ope_ctx *ctx);
/* How the 17 (44) sums of an iteration travel between the ranks of one node.
 *   OPE_COMM_AUTO (default): peer-to-peer slots if every rank could set them up and exchange a test pattern, else RCCL
 *   OPE_COMM_RCCL: ncclAllReduce on the run's stream, then the update kernel
 *   OPE_COMM_P2P:  peer-to-peer slots or an error — every rank stores its sums into a slot it owns in each peer's
 *                  fine-grained buffer (opened through hipIpc handles; 8-byte words {32 data bits, 32-bit sequence number},
 *                  so a word is valid the moment its number matches: no fence between data and flag), reads its own slots
 *                  in rank order and runs the update step in the same launch.  At most 8 ranks.
 * To be called with the same value on every rank, after ope_comm_init_rank and outside a run.
 * ope_comm_transport returns what iterations will use: OPE_COMM_RCCL or OPE_COMM_P2P (0 without a communicator). */
enum { OPE_COMM_AUTO = 0, OPE_COMM_RCCL = 1, OPE_COMM_P2P = 2 };
int ope_comm_set_transport(ope_ctx *ctx, int transport);
int ope_comm_transport(const ope_ctx *ctx);
/* The peer-to-peer slots WITHOUT an RCCL communicator, for drivers that have their own way of passing 64 bytes around
 * (torch.distributed with any backend, MPI, a file): every rank calls ope_comm_p2p_open and publishes the handle it gets;
 * every rank then calls ope_comm_p2p_connect with all handles in rank order.  connect is collective: it maps the peers'
 * buffers and runs the test exchange (up to 10 s); OPE_ECOMM on any rank means no rank may use the communicator — agree
 * on the return codes before iterating.  Such a communicator carries every estimator (17 / 44 sums per iteration, the LM
 * estimator's 17 + 91).  After an exchange has timed out (OPE_ECOMM from ope_icp_poll / ope_icp_end) the ranks' sequence
 * numbers no longer agree: ope_icp_begin refuses further runs until the communicator has been re-created
 * (ope_comm_destroy, then ope_comm_p2p_open / connect or ope_comm_init_rank again, on every rank). */
#define OPE_P2P_HANDLE_BYTES 64
int ope_comm_p2p_open(ope_ctx *ctx, char handle[OPE_P2P_HANDLE_BYTES]);
int ope_comm_p2p_connect(ope_ctx *ctx, const char *handles /* nranks * OPE_P2P_HANDLE_BYTES */, int nranks, int rank);

/* ---------------- features (coarse stage) ---------------- */
/* pcl::NormalEstimation::compute with setKSearch(k) and viewpoint vp (poseestimator.cpp:151-156).
 * out_normals n*3, out_curvature n (optional), ORIGINAL order; NaN where fewer than 3 neighbours.
 * The normals are also attached to `cloud` on the device; with both outputs NULL nothing is copied back. */
int ope_normals(ope_ctx *ctx, ope_cloud *cloud, int k, const float vp[3], float *out_normals, float *out_curvature);
/* The same with the neighbourhoods taken from another cloud's index: NormalEstimation::setSearchSurface
 * (pcl/features/feature.h), used here to shard the normals of one big cloud over ranks — every rank searches the whole
 * cloud's index for its own slice of the points. */
int ope_normals_from(ope_ctx *ctx, ope_cloud *queries, const ope_index *index, int k, const float vp[3], float *out_normals,
                     float *out_curvature);
/* pcl::FPFHEstimation::compute with setRadiusSearch(radius) on a cloud that carries
 * normals (poseestimator.cpp:121-125).  out33 n*33 floats. */
int ope_fpfh(ope_ctx *ctx, const ope_cloud *cloud, float radius, float *out33);
/* pcl::UniformSampling::compute(PointCloud<int>&) with setRadiusSearch(leaf)
 * (poseestimator.cpp:141-145); survivors in ascending voxel-key order. */
int ope_uniform_sampling(ope_ctx *ctx, const ope_cloud *cloud, float leaf, int32_t *out_idx, size_t *n_out);


/* ---------------- filters either side of the path ---------------- */
/* pcl::removeNaNFromPointCloud (poseestimator.cpp:192-194): ORIGINAL indices of the finite points, ascending.
 * out_idx has room for every point of the cloud. */
int ope_remove_nan(ope_ctx *ctx, const ope_cloud *cloud, int32_t *out_idx, size_t *n_out);
/* pcl::PassThrough::filter on "z", then "y", then "x" with setFilterLimits(lo, hi) each
 * (BuildModel processingpcd.cpp:8-36): a finite point survives iff lo[d] <= p[d] <= hi[d] for d = x, y, z
 * (limits inclusive; use -/+FLT_MAX for an unfiltered field).  ORIGINAL indices, ascending. */
int ope_pass_through(ope_ctx *ctx, const ope_cloud *cloud, const float lo[3], const float hi[3], int32_t *out_idx,
                     size_t *n_out);
/* pcl::VoxelGrid::filter with setLeafSize(leaf[0], leaf[1], leaf[2]) (BuildModel processingpcd.cpp:39-52):
 * one centroid per occupied voxel, in ascending voxel index (PCL's output order); xyz only (ope_voxel_grid_rgb
 * carries the colours).  out_xyz has room for 3 floats per finite input point.  OPE_ERANGE where PCL would warn
 * "Leaf size is too small for the input dataset" and return the input unchanged. */
int ope_voxel_grid(ope_ctx *ctx, const ope_cloud *cloud, const float leaf[3], float *out_xyz, size_t *n_out);
/* pcl::VoxelGrid<PointXYZRGB>::filter as ProcessingPcd::getDownSampled runs it (BuildModel processingpcd.cpp:44-59,
 * downsample_all_data_ = true): besides the centroid, the packed colour is averaged channel by channel in float and
 * re-packed by truncation, alpha byte 0 (voxel_grid.hpp, "RGB special case").  rgb: the 32 bits of PointXYZRGB::rgb of
 * every input point, ORIGINAL order; out_rgb: one word per centroid.  rgb == NULL: ope_voxel_grid. */
int ope_voxel_grid_rgb(ope_ctx *ctx, const ope_cloud *cloud, const float leaf[3], const uint32_t *rgb, float *out_xyz,
                       uint32_t *out_rgb, size_t *n_out);
/* pcl::StatisticalOutlierRemoval::filter with setMeanK(mean_k) and setStddevMulThresh(stddev_mul)
 * (ProcessingPcd::getOutlierRemove, DetectAndLocalize processingpcd.cpp:62-77: meanK 30): a point is removed iff its mean
 * distance to its mean_k nearest neighbours exceeds mean + stddev_mul * stddev over the cloud.  ORIGINAL indices of the
 * inliers, ascending (out_idx has room for every point); non-finite points pass, as in PCL.  1 <= mean_k <= 31.
 * out_mean_dist (optional, n floats, ORIGINAL order): the per-point mean distances. */
int ope_statistical_outlier_removal(ope_ctx *ctx, const ope_cloud *cloud, int mean_k, double stddev_mul, int32_t *out_idx,
                                    size_t *n_out, float *out_mean_dist);

/* Device-resident forms of the filters above: the survivors are handed on as a new cloud (*out, as ope_cloud_select of the
 * indices the host form returns would build it), so that crop -> outlier removal -> key points -> normals -> FPFH -> SAC-IA
 * runs without a host round trip per stage.  out_idx (optional, room for every point) receives the same ORIGINAL indices
 * as the host form; n_out (optional) their number. */
int ope_remove_nan_cloud(ope_ctx *ctx, const ope_cloud *cloud, ope_cloud **out, int32_t *out_idx, size_t *n_out);
int ope_pass_through_cloud(ope_ctx *ctx, const ope_cloud *cloud, const float lo[3], const float hi[3], ope_cloud **out, int32_t *out_idx,
                           size_t *n_out);
int ope_statistical_outlier_removal_cloud(ope_ctx *ctx, const ope_cloud *cloud, int mean_k, double stddev_mul, ope_cloud **out,
                                          int32_t *out_idx, size_t *n_out);
int ope_uniform_sampling_cloud(ope_ctx *ctx, const ope_cloud *cloud, float leaf, ope_cloud **out, int32_t *out_idx, size_t *n_out);

typedef struct {
  int max_iterations;      /* 400  (poseestimator.cpp:55) */
  int nr_samples;          /* 5    (:56) */
  int k_correspondences;   /* 5    (:57) */
  double max_corr_dist;    /* 0.05 (:58) */
  float min_sample_dist;   /* 0.01 (:59) */
  uint64_t seed;           /* PCL uses unseeded rand(); here an explicit LCG stream */
} ope_sacia_params;
void ope_sacia_default_params(ope_sacia_params *p);
/* pcl::SampleConsensusInitialAlignment::align (poseestimator.cpp:50-64).
 * src_feat/tgt_feat: n*33 floats in ORIGINAL order.  forced_samples (optional):
 * max_iterations*nr_samples source indices then as many target indices. */
int ope_sacia(ope_ctx *ctx, const ope_cloud *src, const float *src_feat33, const ope_cloud *tgt,
              const ope_index *tgt_index, const float *tgt_feat33, const ope_sacia_params *params,
              const int32_t *forced_samples, float out_T[16], double *best_error, int32_t *best_iteration);

#ifdef __cplusplus
}
#endif
#endif /* OPE_H */
