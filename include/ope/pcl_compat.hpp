// pcl_compat.hpp — header-only C++ façade over the C ABI (include/ope.h) that re-creates the PCL call
// shapes the reference consumes, so that DetectAndLocalize/src/poseestimator.cpp and
// BuildModel/src/regmeshpcd.cpp can be re-pointed with a namespace alias:
//
//     namespace pcl = ope::compat;          // instead of #include <pcl/...>
//
// Mirrored interfaces (names, argument meaning and error behaviour follow the reference):
//   Registration / IterativeClosestPoint[WithNormals]   vPCL registration_mod.h:151-479, icp_mod.h:165-281
//   registration::CorrespondenceEstimationNormalShooting poseestimator.cpp:242-246
//   registration::CorrespondenceRejectorSurfaceNormal    poseestimator.cpp:264-272
//   registration::CorrespondenceRejectorSelfOccludedNormal vPCL correspondence_rejection_self_occluded_normal.h
//   registration::TransformationEstimationSVD            poseestimator.cpp:306,435
//   registration::TransformationEstimationPointToPlane[LLS] BuildModel regmeshpcd.cpp:162,193; vPCL icp_mod.h:355
//   NormalEstimation / FPFHEstimation / UniformSampling  poseestimator.cpp:121-125,141-156
//   removeNaNFromPointCloud / PassThrough / VoxelGrid    poseestimator.cpp:192-194; BuildModel processingpcd.cpp:8-52
//   SampleConsensusInitialAlignment                      poseestimator.cpp:50-64
// Point types are layout-compatible PODs (x@0,y@4,z@8; normal@16 in PointXYZRGBNormal; 16-byte aligned).
// Errors never throw on the hot path: like PCL, a failed call logs to stderr and leaves
// hasConverged() == false / the transform at identity (registration_mod.hpp:60-64,73-77).
#pragma once

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <memory>
#include <string>
#include <vector>

#include "../ope.h"

namespace ope {
namespace compat {

// ------------------------------------------------------------------------------------------ PODs
struct alignas(16) PointXYZ { float x, y, z, data3 = 1.f; };
struct alignas(16) Normal { float normal_x, normal_y, normal_z, data_n3 = 0.f; float curvature = 0.f, pad_[3] = {0, 0, 0}; };
struct alignas(16) PointXYZRGB { float x, y, z, data3 = 1.f; float rgb = 0.f, pad_[3] = {0, 0, 0}; };
struct alignas(16) PointXYZRGBNormal {
  float x, y, z, data3 = 1.f;
  float normal_x = 0.f, normal_y = 0.f, normal_z = 0.f, data_n3 = 0.f;
  float rgb = 0.f, curvature = 0.f, pad_[2] = {0, 0};
};
struct FPFHSignature33 { float histogram[33]; };
struct Correspondence { int index_query = 0; int index_match = -1; float distance = FLT_MAX; };
typedef std::vector<Correspondence> Correspondences;
static_assert(sizeof(PointXYZ) == 16 && sizeof(Normal) == 32 && sizeof(PointXYZRGB) == 32 &&
              sizeof(PointXYZRGBNormal) == 48 && sizeof(FPFHSignature33) == 132 && sizeof(Correspondence) == 12,
              "PCL layouts");

template <class T> struct point_traits { static constexpr ptrdiff_t normal_offset = -1; static constexpr ptrdiff_t rgb_offset = -1; };
template <> struct point_traits<PointXYZRGB> { static constexpr ptrdiff_t normal_offset = -1; static constexpr ptrdiff_t rgb_offset = 16; };
template <> struct point_traits<PointXYZRGBNormal> { static constexpr ptrdiff_t normal_offset = 16; static constexpr ptrdiff_t rgb_offset = 32; };
template <> struct point_traits<Normal> { static constexpr ptrdiff_t normal_offset = 0; static constexpr ptrdiff_t rgb_offset = -1; };

template <class PointT> struct PointCloud {
  typedef std::shared_ptr<PointCloud<PointT>> Ptr;
  typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
  std::vector<PointT> points;
  uint32_t width = 0, height = 1;
  bool is_dense = true;
  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void clear() { points.clear(); width = 0; height = 1; }
  void resize(size_t n) { points.resize(n); width = (uint32_t)n; height = 1; }
  void push_back(const PointT &p) { points.push_back(p); width = (uint32_t)points.size(); height = 1; }
  PointT &operator[](size_t i) { return points[i]; }
  const PointT &operator[](size_t i) const { return points[i]; }
};

// Column-major 4x4, the memory layout of Eigen::Matrix4f.
struct Matrix4f {
  float m[16];
  static Matrix4f Identity() { Matrix4f r; std::memset(r.m, 0, sizeof r.m); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f; return r; }
  float &operator()(int r, int c) { return m[4 * c + r]; }
  float operator()(int r, int c) const { return m[4 * c + r]; }
  Matrix4f operator*(const Matrix4f &b) const {
    Matrix4f o;
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 4; ++r) {
        float a = 0.f;
        for (int k = 0; k < 4; ++k) a += m[4 * k + r] * b.m[4 * c + k];
        o.m[4 * c + r] = a;
      }
    return o;
  }
  bool operator!=(const Matrix4f &b) const { return std::memcmp(m, b.m, sizeof m) != 0; }
  const float *data() const { return m; }
  float *data() { return m; }
};

// ------------------------------------------------------------------------------------------ plumbing
inline ope_ctx *default_context(int device = 0) {
  static ope_ctx *ctx = nullptr;
  if (!ctx && ope_ctx_create(&ctx, device) != OPE_OK) {
    std::fprintf(stderr, "[ope] cannot create a GPU context: %s\n", ope_last_error(nullptr));
    ctx = nullptr;
  }
  return ctx;
}
inline void log_error(const char *where, ope_ctx *ctx) { std::fprintf(stderr, "[ope::%s] %s\n", where, ope_last_error(ctx)); }

struct CloudHandle {
  ope_cloud *h = nullptr;
  ~CloudHandle() { if (h) ope_cloud_free(h); }
};
struct IndexHandle {
  ope_index *h = nullptr;
  ~IndexHandle() { if (h) ope_index_free(h); }
};

template <class PointT> inline std::shared_ptr<CloudHandle> upload(const PointCloud<PointT> &c, bool with_normals) {
  auto r = std::make_shared<CloudHandle>();
  ope_ctx *ctx = default_context();
  if (!ctx) return r;
  const ptrdiff_t noff = with_normals ? point_traits<PointT>::normal_offset : -1;
  if (ope_cloud_upload(ctx, c.points.data(), c.points.size(), sizeof(PointT), 0, noff, &r->h) != OPE_OK) log_error("upload", ctx);
  return r;
}

template <class PointT> inline void transformPointCloud(const PointCloud<PointT> &in, PointCloud<PointT> &out, const Matrix4f &T) {
  if (&in != &out) out = in;
  for (auto &p : out.points) {
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    const float x = p.x, y = p.y, z = p.z;
    p.x = T.m[0] * x + T.m[4] * y + T.m[8] * z + T.m[12];
    p.y = T.m[1] * x + T.m[5] * y + T.m[9] * z + T.m[13];
    p.z = T.m[2] * x + T.m[6] * y + T.m[10] * z + T.m[14];
  }
}

// pcl::transformPointCloudWithNormals: xyz by T, the normal by T's rotation block (what
// IterativeClosestPointWithNormals::transformCloud applies to its working cloud, vPCL impl/icp_mod.hpp:311-318).
template <class PointT> inline void transformPointCloudWithNormals(const PointCloud<PointT> &in, PointCloud<PointT> &out, const Matrix4f &T) {
  static_assert(point_traits<PointT>::normal_offset >= 0, "transformPointCloudWithNormals needs a point type with normals");
  transformPointCloud(in, out, T);
  for (auto &p : out.points) {
    float *n = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(&p) + point_traits<PointT>::normal_offset);
    const float x = n[0], y = n[1], z = n[2];
    n[0] = T.m[0] * x + T.m[4] * y + T.m[8] * z;
    n[1] = T.m[1] * x + T.m[5] * y + T.m[9] * z;
    n[2] = T.m[2] * x + T.m[6] * y + T.m[10] * z;
  }
}

// pcl::ScopeTime: wall-clock of a scope, printed to stderr as "<title> took <ms>ms." when the scope ends.  The reference
// wraps its two stages in ScopeTime t("Initial Alignment") / ("Final Alignment") (poseestimator.cpp:61,349).
class ScopeTime {
 public:
  explicit ScopeTime(const char *title) : title_(title) { clock_gettime(CLOCK_MONOTONIC, &t0_); }
  double getTime() const {
    timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (t1.tv_sec - t0_.tv_sec) * 1e3 + (t1.tv_nsec - t0_.tv_nsec) * 1e-6;
  }
  ~ScopeTime() { std::fprintf(stderr, "%s took %gms.\n", title_.c_str(), getTime()); }
 private:
  std::string title_;
  timespec t0_;
};

// ------------------------------------------------------------------------------------------ strategies
namespace registration {

// CorrespondenceEstimationBase (vPCL correspondence_estimation_mod.h:286-299).  Inside ICP the estimation is fused into
// the search kernel and the object only selects the mode; called stand-alone (poseestimator.cpp:242-247 does, once, and
// throws the result away) determineCorrespondences runs ONE search pass of the same kernel through the C ABI
// (ope_icp_begin -> ope_icp_accumulate -> ope_icp_correspondences) and returns pcl::Correspondences in query order with
// squared distances.
template <class PointSource, class PointTarget> struct CorrespondenceEstimationBase {
  typedef std::shared_ptr<CorrespondenceEstimationBase> Ptr;
  virtual ~CorrespondenceEstimationBase() {}
  virtual int mode() const { return OPE_CORR_NEAREST; }
  virtual int k() const { return 1; }
  void setInputSource(const typename PointCloud<PointSource>::ConstPtr &c) { input_ = c; }
  void setInputCloud(const typename PointCloud<PointSource>::ConstPtr &c) { input_ = c; }
  void setInputTarget(const typename PointCloud<PointTarget>::ConstPtr &c) { target_ = c; }
  // max_distance: 1-NN keeps d2 <= max_distance^2 (correspondence_estimation_mod.hpp:171); normal shooting compares the
  // SQUARED point-to-line distance with the unsquared max_distance (quirk Q2, ...normal_shooting_weighted.hpp:136)
  virtual void determineCorrespondences(Correspondences &correspondences, double max_distance = std::sqrt(DBL_MAX)) {
    run_pass(correspondences, max_distance, false);
  }
  virtual void determineReciprocalCorrespondences(Correspondences &correspondences, double max_distance = std::sqrt(DBL_MAX)) {
    run_pass(correspondences, max_distance, true);
  }
 protected:
  void run_pass(Correspondences &out, double max_distance, bool reciprocal) {
    out.clear();
    ope_ctx *ctx = default_context();
    if (!ctx || !input_ || !target_ || target_->empty()) return;
    const bool nrm = mode() == OPE_CORR_NORMAL_SHOOTING;
    auto src = upload(*input_, nrm), tgt = upload(*target_, false);
    IndexHandle ix;
    if (!src->h || !tgt->h || ope_index_build(ctx, tgt->h, nullptr, &ix.h) != OPE_OK) { log_error("determineCorrespondences", ctx); return; }
    ope_icp_params p;
    ope_icp_default_params(&p);
    p.max_iterations = 1;
    p.max_corr_dist = max_distance;
    p.corr_mode = mode();
    p.k_normal_shooting = k();
    p.use_reciprocal = reciprocal ? 1 : 0;
    std::vector<int32_t> q(input_->size()), m(input_->size());
    std::vector<float> d(input_->size());
    size_t n = 0;
    if (ope_icp_begin(ctx, src->h, ix.h, nullptr, &p) != OPE_OK || ope_icp_accumulate(ctx) != OPE_OK ||
        ope_icp_correspondences(ctx, q.data(), m.data(), d.data(), q.size(), &n) != OPE_OK) {
      log_error("determineCorrespondences", ctx);
      return;
    }
    ope_icp_end(ctx, nullptr, nullptr);
    out.resize(n);
    for (size_t i = 0; i < n; ++i) { out[i].index_query = q[i]; out[i].index_match = m[i]; out[i].distance = d[i]; }
  }
  typename PointCloud<PointSource>::ConstPtr input_;
  typename PointCloud<PointTarget>::ConstPtr target_;
};
template <class S, class T, class Scalar = float> struct CorrespondenceEstimation : CorrespondenceEstimationBase<S, T> {
  typedef std::shared_ptr<CorrespondenceEstimation> Ptr;
};
template <class S, class T, class N, class Scalar = float> struct CorrespondenceEstimationNormalShooting : CorrespondenceEstimationBase<S, T> {
  typedef std::shared_ptr<CorrespondenceEstimationNormalShooting> Ptr;
  int k_ = 10;   // the class default (vPCL correspondence_estimation_normal_shooting_weighted.h:117)
  void setKSearch(int k) { k_ = k; }
  int mode() const override { return OPE_CORR_NORMAL_SHOOTING; }
  int k() const override { return k_; }
  // the source normals travel inside the source point type (PointXYZRGBNormal at poseestimator.cpp:244)
  template <class P> void setSourceNormals(const P &) {}
};

// CorrespondenceRejector (vPCL correspondence_rejection_mod.h:108-110,138-183).  Inside ICP the predicates are fused into
// the search kernel (`apply`); stand-alone, getRemainingCorrespondences evaluates the same predicate for the given
// correspondences on the device (ope_reject_pairs).
struct CorrespondenceRejector {
  typedef std::shared_ptr<CorrespondenceRejector> Ptr;
  virtual ~CorrespondenceRejector() {}
  virtual void apply(ope_icp_params &p) const = 0;
  virtual void getRemainingCorrespondences(const Correspondences &original, Correspondences &remaining) = 0;
  void setInputCorrespondences(const std::shared_ptr<const Correspondences> &c) { input_correspondences_ = c; }
  void getCorrespondences(Correspondences &out) {
    out.clear();
    if (input_correspondences_) getRemainingCorrespondences(*input_correspondences_, out);
  }
 protected:
  std::shared_ptr<const Correspondences> input_correspondences_;
  // packed xyz / normal triples of a cloud, pulled out of whatever point type it has
  struct Triples { std::vector<float> v; bool set = false; };
  template <class P> static void take_xyz(const PointCloud<P> &c, Triples &t) {
    t.v.resize(3 * c.size());
    for (size_t i = 0; i < c.size(); ++i) { t.v[3 * i] = c[i].x; t.v[3 * i + 1] = c[i].y; t.v[3 * i + 2] = c[i].z; }
    t.set = true;
  }
  template <class N> static void take_normals(const PointCloud<N> &c, Triples &t) {
    static_assert(point_traits<N>::normal_offset >= 0, "point type carries no normal");
    t.v.resize(3 * c.size());
    for (size_t i = 0; i < c.size(); ++i)
      std::memcpy(&t.v[3 * i], reinterpret_cast<const unsigned char *>(&c[i]) + point_traits<N>::normal_offset, 12);
    t.set = true;
  }
  static void filter(int kind, double thr, const Triples &a, bool a_by_query, const Triples &b, bool b_by_query,
                     const Correspondences &in, Correspondences &out) {
    out.clear();
    ope_ctx *ctx = default_context();
    if (!ctx || in.empty()) return;
    if (!a.set || !b.set) { std::fprintf(stderr, "[ope::CorrespondenceRejector] input clouds / normals were not set\n"); return; }
    std::vector<float> pa(3 * in.size()), pb(3 * in.size());
    for (size_t i = 0; i < in.size(); ++i) {
      const size_t ia = (size_t)(a_by_query ? in[i].index_query : in[i].index_match), ib = (size_t)(b_by_query ? in[i].index_query : in[i].index_match);
      if (3 * ia + 2 >= a.v.size() || 3 * ib + 2 >= b.v.size()) return;
      std::memcpy(&pa[3 * i], &a.v[3 * ia], 12);
      std::memcpy(&pb[3 * i], &b.v[3 * ib], 12);
    }
    std::vector<unsigned char> keep(in.size());
    if (ope_reject_pairs(ctx, kind, pa.data(), pb.data(), in.size(), thr, keep.data()) != OPE_OK) { log_error("getRemainingCorrespondences", ctx); return; }
    for (size_t i = 0; i < in.size(); ++i)
      if (keep[i]) out.push_back(in[i]);
  }
};
// keep a correspondence if n_src . n_tgt > threshold (score: correspondence_rejection_mod.h:368-376; poseestimator.cpp:264-273)
struct CorrespondenceRejectorSurfaceNormal : CorrespondenceRejector {
  typedef std::shared_ptr<CorrespondenceRejectorSurfaceNormal> Ptr;
  double threshold_ = 1.0;
  void setThreshold(double t) { threshold_ = t; }
  double getThreshold() const { return threshold_; }
  template <class P, class N> void initializeDataContainer() {}
  template <class P> void setInputSource(const typename PointCloud<P>::ConstPtr &) {}
  template <class P> void setInputTarget(const typename PointCloud<P>::ConstPtr &) {}
  template <class P, class N> void setInputNormals(const typename PointCloud<N>::ConstPtr &n) { if (n) take_normals(*n, src_n_); }
  template <class P, class N> void setTargetNormals(const typename PointCloud<N>::ConstPtr &n) { if (n) take_normals(*n, tgt_n_); }
  void apply(ope_icp_params &p) const override { p.use_surface_normal_rej = 1; p.surface_normal_thr = threshold_; }
  void getRemainingCorrespondences(const Correspondences &original, Correspondences &remaining) override {
    filter(OPE_REJ_SURFACE_NORMAL, threshold_, src_n_, true, tgt_n_, false, original, remaining);
  }
 private:
  Triples src_n_, tgt_n_;
};
// keep a correspondence if n_src . (-p_src / |p_src|) > threshold (vPCL correspondence_rejection_mod.h:382-391,
// impl/correspondence_rejection_self_occluded_normal.cpp:43-64); opt-in, see SURVEY Q3
struct CorrespondenceRejectorSelfOccludedNormal : CorrespondenceRejector {
  typedef std::shared_ptr<CorrespondenceRejectorSelfOccludedNormal> Ptr;
  double threshold_ = 1.0;
  void setThreshold(double t) { threshold_ = t; }
  double getThreshold() const { return threshold_; }
  template <class P, class N> void initializeDataContainer() {}
  template <class P> void setInputSource(const typename PointCloud<P>::ConstPtr &c) { if (c) take_xyz(*c, src_p_); }
  template <class P, class N> void setInputNormals(const typename PointCloud<N>::ConstPtr &n) { if (n) take_normals(*n, src_n_); }
  void apply(ope_icp_params &p) const override { p.use_self_occluded_rej = 1; p.self_occluded_thr = threshold_; }
  void getRemainingCorrespondences(const Correspondences &original, Correspondences &remaining) override {
    filter(OPE_REJ_SELF_OCCLUDED, threshold_, src_n_, true, src_p_, true, original, remaining);
  }
 private:
  Triples src_n_, src_p_;
};

template <class S, class T, class Scalar = float> struct TransformationEstimationSVD {
  typedef std::shared_ptr<TransformationEstimationSVD> Ptr;
  static constexpr int ope_estimator = OPE_EST_SVD;
  void estimateRigidTransformation(const PointCloud<S> &src, const PointCloud<T> &tgt, const Correspondences &corrs,
                                   Matrix4f &out) const {
    out = Matrix4f::Identity();
    ope_ctx *ctx = default_context();
    if (!ctx || corrs.empty()) return;
    std::vector<float> a(3 * corrs.size()), b(3 * corrs.size());
    for (size_t i = 0; i < corrs.size(); ++i) {
      const S &s = src.points[corrs[i].index_query];
      const T &t = tgt.points[corrs[i].index_match];
      a[3 * i] = s.x; a[3 * i + 1] = s.y; a[3 * i + 2] = s.z;
      b[3 * i] = t.x; b[3 * i + 1] = t.y; b[3 * i + 2] = t.z;
    }
    if (ope_rigid_transform_svd(ctx, a.data(), b.data(), corrs.size(), out.m) != OPE_OK) log_error("TransformationEstimationSVD", ctx);
  }
};

// Selected by type in setTransformationEstimation: the 6x6 normal equations of the linearised point-to-plane
// error (the IterativeClosestPointWithNormals default, vPCL icp_mod.h:355).
template <class S, class T, class Scalar = float> struct TransformationEstimationPointToPlaneLLS {
  typedef std::shared_ptr<TransformationEstimationPointToPlaneLLS> Ptr;
  static constexpr int ope_estimator = OPE_EST_POINT_TO_PLANE_LLS;
};
// PCL's class of this name minimises the same error with Levenberg-Marquardt over (t, quaternion)
// (BuildModel regmeshpcd.cpp:162,193): OPE_EST_POINT_TO_PLANE_LM, csrc/lm.hip.
template <class S, class T, class Scalar = float> struct TransformationEstimationPointToPlane {
  typedef std::shared_ptr<TransformationEstimationPointToPlane> Ptr;
  static constexpr int ope_estimator = OPE_EST_POINT_TO_PLANE_LM;
};

// DefaultConvergenceCriteria knobs the reference can reach through getConvergeCriteria()
struct DefaultConvergenceCriteria {
  enum ConvergenceState {
    CONVERGENCE_CRITERIA_NOT_CONVERGED, CONVERGENCE_CRITERIA_ITERATIONS, CONVERGENCE_CRITERIA_TRANSFORM,
    CONVERGENCE_CRITERIA_ABS_MSE, CONVERGENCE_CRITERIA_REL_MSE, CONVERGENCE_CRITERIA_NO_CORRESPONDENCES
  };
  double mse_threshold_absolute_ = 1e-12;
  bool failure_after_max_iter_ = false;
  ConvergenceState state_ = CONVERGENCE_CRITERIA_NOT_CONVERGED;
  void setAbsoluteMSE(double v) { mse_threshold_absolute_ = v; }
  double getAbsoluteMSE() const { return mse_threshold_absolute_; }
  void setFailureAfterMaximumIterations(bool v) { failure_after_max_iter_ = v; }
  ConvergenceState getConvergenceState() const { return state_; }
};

}  // namespace registration

// ------------------------------------------------------------------------------------------ ICP
template <class PointSource, class PointTarget, class Scalar = float> class IterativeClosestPoint {
 public:
  typedef PointCloud<PointSource> PointCloudSource;
  typedef PointCloud<PointTarget> PointCloudTarget;
  typedef Matrix4f Matrix4;

  IterativeClosestPoint() { ope_icp_default_params(&params_); criteria_ = std::make_shared<registration::DefaultConvergenceCriteria>(); }
  virtual ~IterativeClosestPoint() {}

  void setInputSource(const typename PointCloudSource::ConstPtr &cloud) { input_ = cloud; src_dev_.reset(); correspondences_.clear(); }
  typename PointCloudSource::ConstPtr const getInputSource() const { return input_; }
  typename PointCloudTarget::ConstPtr const getInputTarget() const { return target_; }
  void setInputCloud(const typename PointCloudSource::ConstPtr &cloud) { setInputSource(cloud); }
  void setInputTarget(const typename PointCloudTarget::ConstPtr &cloud) {
    if (!cloud || cloud->points.empty()) {  // registration_mod.hpp:60-64
      std::fprintf(stderr, "[ope::IterativeClosestPoint::setInputTarget] Invalid or empty point cloud dataset given!\n");
      return;
    }
    target_ = cloud;
    tgt_dev_.reset(); tgt_index_.reset();
  }
  void setMaximumIterations(int n) { params_.max_iterations = n; }
  void setTransformationEpsilon(double e) { params_.transformation_epsilon = e; }
  void setEuclideanFitnessEpsilon(double e) { params_.euclidean_fitness_epsilon = e; }
  void setMaxCorrespondenceDistance(double d) { params_.max_corr_dist = d; }
  void setRANSACOutlierRejectionThreshold(double) {}  // accepted and unused, as in the reference (poseestimator.cpp:319)
  void setUseReciprocalCorrespondences(bool b) { params_.use_reciprocal = b ? 1 : 0; }
  typedef registration::CorrespondenceEstimationBase<PointSource, PointTarget> CorrespondenceEstimation;
  void setCorrespondenceEstimation(const std::shared_ptr<CorrespondenceEstimation> &ce) { corr_est_ = ce; }
  void addCorrespondenceRejector(const registration::CorrespondenceRejector::Ptr &r) { rejectors_.push_back(r); }
  template <class TE> void setTransformationEstimation(const std::shared_ptr<TE> &) { params_.estimator = TE::ope_estimator; }
  std::shared_ptr<registration::DefaultConvergenceCriteria> getConvergeCriteria() { return criteria_; }
  // vPCL icp_mod.h:268-281 — the reference's injection of given pairs into every iteration (unused by its own programs).
  // The pointer is kept, as in the reference; the pairs are read at align().  The reference's correspondence estimation also
  // writes each pair's `distance` field back through the pointer, every iteration (correspondence_estimation_mod.hpp:150-161):
  // align() leaves the last iteration's values there, which is what a caller of the reference finds after align().
  void setFixedCorrespondences(Correspondences *correspondences) { corres_fixed_ = correspondences; }
  Correspondences getFixedCorrespondences() { return *corres_fixed_; }
  void clearCorrespondences() { if (corres_fixed_) corres_fixed_->clear(); }

  void align(PointCloudSource &output) { align(output, Matrix4f::Identity()); }
  void align(PointCloudSource &output, const Matrix4f &guess) {
    converged_ = false;
    final_ = Matrix4f::Identity();
    last_incremental_ = Matrix4f::Identity();
    nr_iterations_ = 0;
    n_corr_ = 0;
    correspondences_.clear();
    ope_ctx *ctx = default_context();
    if (!ctx) return;
    if (!target_) {  // registration_mod.hpp:73-77
      std::fprintf(stderr, "[ope::IterativeClosestPoint::compute] No input target dataset was given!\n");
      return;
    }
    if (!input_) return;
    ope_icp_params p = params_;
    p.corr_mode = corr_est_ ? corr_est_->mode() : OPE_CORR_NEAREST;
    if (corr_est_) p.k_normal_shooting = corr_est_->k();
    p.use_surface_normal_rej = p.use_self_occluded_rej = 0;
    for (auto &r : rejectors_) r->apply(p);
    p.mse_threshold_absolute = criteria_->mse_threshold_absolute_;
    p.failure_after_max_iter = criteria_->failure_after_max_iter_ ? 1 : 0;
    const bool nrm = p.corr_mode == OPE_CORR_NORMAL_SHOOTING || p.use_surface_normal_rej || p.use_self_occluded_rej ||
                     p.estimator == OPE_EST_POINT_TO_PLANE_LLS || p.estimator == OPE_EST_POINT_TO_PLANE_LM;
    // uploads are cached between align() calls on the same clouds; a cache made without normals is of no use once a
    // rejector, normal shooting or the point-to-plane estimator has been added
    if (nrm && !uploaded_with_normals_) { src_dev_.reset(); tgt_dev_.reset(); tgt_index_.reset(); }
    if (!src_dev_) src_dev_ = upload(*input_, nrm);
    if (!tgt_index_) {
      uploaded_with_normals_ = nrm;
      tgt_dev_ = upload(*target_, nrm);
      tgt_index_ = std::make_shared<IndexHandle>();
      if (tgt_dev_->h && ope_index_build(ctx, tgt_dev_->h, nullptr, &tgt_index_->h) != OPE_OK) log_error("align", ctx);
    }
    if (!src_dev_->h || !tgt_index_->h) return;
    // icp_mod.hpp:150-151: the given pairs, if any, go to the correspondence estimation of THIS run (the context is shared
    // between objects: they are cleared again below)
    std::vector<int32_t> fq, fm;
    {
      if (corres_fixed_)
        for (const Correspondence &c : *corres_fixed_) { fq.push_back(c.index_query); fm.push_back(c.index_match); }
      if (ope_icp_set_fixed_correspondences(ctx, src_dev_->h, tgt_dev_->h, fq.data(), fm.data(), fq.size()) != OPE_OK) {
        log_error("align (fixed correspondences)", ctx);
        return;
      }
    }
    ope_icp_result res;
    const int rc_run = ope_icp_run(ctx, src_dev_->h, tgt_index_->h, guess.m, &p, final_.m, &res);
    std::vector<int32_t> fixed_listed, fixed_appended;
    if (corres_fixed_ && !corres_fixed_->empty()) {
      std::vector<float> fd(fq.size());
      fixed_listed.resize(fq.size());
      fixed_appended.resize(fq.size());
      size_t nf = 0;
      if (rc_run == OPE_OK && ope_icp_fixed_correspondences(ctx, fd.data(), fixed_listed.data(), fixed_appended.data(), fd.size(), &nf) == OPE_OK &&
          nf == corres_fixed_->size())
        for (size_t f = 0; f < nf; ++f) (*corres_fixed_)[f].distance = fd[f];
      else
        fixed_listed.clear(), fixed_appended.clear();
      (void)ope_icp_set_fixed_correspondences(ctx, nullptr, nullptr, nullptr, nullptr, 0);
    }
    if (rc_run != OPE_OK) {
      log_error("align", ctx);
      final_ = Matrix4f::Identity();
      return;
    }
    converged_ = res.converged != 0;
    nr_iterations_ = res.iterations;
    n_corr_ = res.n_corr;
    criteria_->state_ = (registration::DefaultConvergenceCriteria::ConvergenceState)res.state;
    if (ope_icp_last_incremental(ctx, last_incremental_.m) != OPE_OK) last_incremental_ = Matrix4f::Identity();
    // correspondences_ is a member of the reference's object too (icp_mod.h:249-260 reads it): fetched now, while the
    // context still holds THIS object's run
    {
      std::vector<int32_t> q(input_->size()), m(input_->size());
      std::vector<float> d(input_->size());
      size_t n = 0;
      if (ope_icp_correspondences(ctx, q.data(), m.data(), d.data(), q.size(), &n) == OPE_OK) {
        // the reference's list: given pairs that the estimation listed (through every rejector), the searched pairs, the given
        // pairs once more where the first rejector alone lets them through (icp_mod.hpp:210-224)
        for (size_t f = 0; f < fixed_listed.size(); ++f)
          if (fixed_listed[f]) correspondences_.push_back((*corres_fixed_)[f]);
        const size_t at = correspondences_.size();
        correspondences_.resize(at + n);
        for (size_t i = 0; i < n; ++i) { correspondences_[at + i].index_query = q[i]; correspondences_[at + i].index_match = m[i]; correspondences_[at + i].distance = d[i]; }
        for (size_t f = 0; f < fixed_appended.size(); ++f)
          if (fixed_appended[f]) correspondences_.push_back((*corres_fixed_)[f]);
      }
    }
    transformOutput(output);  // icp_mod.hpp:269-271
  }

  Matrix4f getFinalTransformation() const { return final_; }
  Matrix4f getLastIncrementalTransformation() const { return last_incremental_; }   // transformation_, registration_mod.h
  bool hasConverged() const { return converged_; }
  int getNumberOfIterations() const { return nr_iterations_; }
  double getFitnessScore(double max_range = DBL_MAX) {
    ope_ctx *ctx = default_context();
    double score = DBL_MAX;
    if (ctx && src_dev_ && tgt_index_ && src_dev_->h && tgt_index_->h &&
        ope_fitness(ctx, src_dev_->h, tgt_index_->h, final_.m, max_range, &score, nullptr, nullptr) != OPE_OK)
      log_error("getFitnessScore", ctx);
    return score;
  }
  // icp_mod.h:249-260
  double getAlignStrength() const {
    const double denom = (double)((input_ ? input_->size() : 0) + (target_ ? target_->size() : 0));
    return denom > 0 ? (double)n_corr_ / denom : 0.0;
  }
  // pcl::Correspondences of the last iteration (post-rejection), query order
  Correspondences getCorrespondences() const { return correspondences_; }

 protected:
  // Registration::transformCloud of the final result: xyz only here, xyz + normals in IterativeClosestPointWithNormals
  virtual void transformOutput(PointCloudSource &output) const { transformPointCloud(*input_, output, final_); }
  typename PointCloudSource::ConstPtr input_;
  typename PointCloudTarget::ConstPtr target_;
  bool uploaded_with_normals_ = false;
  Matrix4f last_incremental_ = Matrix4f::Identity();
  Correspondences correspondences_;
  Correspondences *corres_fixed_ = nullptr;   // icp_mod.h:326
  std::shared_ptr<CloudHandle> src_dev_, tgt_dev_;
  std::shared_ptr<IndexHandle> tgt_index_;
  ope_icp_params params_;
  std::shared_ptr<CorrespondenceEstimation> corr_est_;
  std::vector<registration::CorrespondenceRejector::Ptr> rejectors_;
  std::shared_ptr<registration::DefaultConvergenceCriteria> criteria_;
  Matrix4f final_ = Matrix4f::Identity();
  bool converged_ = false;
  int nr_iterations_ = 0;
  int64_t n_corr_ = 0;
};

// Transforms the normals too (vPCL impl/icp_mod.hpp:311-318; the kernels always do) and starts from the point-to-plane
// LLS estimator (icp_mod.h:355).
template <class S, class T, class Scalar = float> class IterativeClosestPointWithNormals : public IterativeClosestPoint<S, T, Scalar> {
 public:
  IterativeClosestPointWithNormals() { this->params_.estimator = OPE_EST_POINT_TO_PLANE_LLS; }
 protected:
  void transformOutput(PointCloud<S> &output) const override { transformPointCloudWithNormals(*this->input_, output, this->final_); }
};

// ------------------------------------------------------------------------------------------ features
namespace search { template <class P> struct KdTree { typedef std::shared_ptr<KdTree> Ptr; explicit KdTree(bool = true) {} }; }

template <class PointInT, class PointOutT = Normal> class NormalEstimation {
 public:
  void setInputCloud(const typename PointCloud<PointInT>::ConstPtr &c) { input_ = c; }
  template <class Tree> void setSearchMethod(const Tree &) {}  // the GPU index replaces the kd-tree
  void setKSearch(int k) { k_ = k; }
  void setViewPoint(float x, float y, float z) { vp_[0] = x; vp_[1] = y; vp_[2] = z; }
  // pcl::Feature::setSearchSurface: neighbourhoods are taken from `surface` instead of the input cloud
  void setSearchSurface(const typename PointCloud<PointInT>::ConstPtr &surface) { surface_ = surface; }
  void compute(PointCloud<PointOutT> &out) {
    out.clear();
    ope_ctx *ctx = default_context();
    if (!ctx || !input_) return;
    auto dev = upload(*input_, false);
    if (!dev->h) return;
    const size_t n = input_->size();
    std::vector<float> nrm(3 * n), curv(n);
    int rc;
    if (surface_ && surface_ != input_) {
      auto sdev = upload(*surface_, false);
      IndexHandle six;
      rc = sdev->h ? ope_index_build(ctx, sdev->h, nullptr, &six.h) : OPE_EINVAL;
      if (rc == OPE_OK) rc = ope_normals_from(ctx, dev->h, six.h, k_, vp_, nrm.data(), curv.data());
    } else {
      rc = ope_normals(ctx, dev->h, k_, vp_, nrm.data(), curv.data());
    }
    if (rc != OPE_OK) { log_error("NormalEstimation", ctx); return; }
    out.resize(n);
    out.is_dense = true;
    for (size_t i = 0; i < n; ++i) {
      out.points[i].normal_x = nrm[3 * i]; out.points[i].normal_y = nrm[3 * i + 1]; out.points[i].normal_z = nrm[3 * i + 2];
      out.points[i].curvature = curv[i];
      if (!std::isfinite(nrm[3 * i])) out.is_dense = false;
    }
  }
 private:
  typename PointCloud<PointInT>::ConstPtr input_, surface_;
  int k_ = 0;
  float vp_[3] = {0, 0, 0};
};

template <class PointInT, class PointNT = Normal, class PointOutT = FPFHSignature33> class FPFHEstimation {
 public:
  void setInputCloud(const typename PointCloud<PointInT>::ConstPtr &c) { input_ = c; }
  void setInputNormals(const typename PointCloud<PointNT>::ConstPtr &n) { normals_ = n; }
  void setRadiusSearch(double r) { radius_ = r; }
  template <class Tree> void setSearchMethod(const Tree &) {}
  void compute(PointCloud<PointOutT> &out) {
    out.clear();
    ope_ctx *ctx = default_context();
    if (!ctx || !input_ || !normals_ || normals_->size() != input_->size()) return;
    auto dev = upload(*input_, false);
    if (!dev->h) return;
    const size_t n = input_->size();
    std::vector<float> nrm(3 * n);
    for (size_t i = 0; i < n; ++i) {
      nrm[3 * i] = normals_->points[i].normal_x; nrm[3 * i + 1] = normals_->points[i].normal_y; nrm[3 * i + 2] = normals_->points[i].normal_z;
    }
    out.resize(n);
    if (ope_cloud_set_normals(ctx, dev->h, nrm.data()) != OPE_OK ||
        ope_fpfh(ctx, dev->h, (float)radius_, &out.points[0].histogram[0]) != OPE_OK) {
      log_error("FPFHEstimation", ctx);
      out.clear();
    }
  }
 private:
  typename PointCloud<PointInT>::ConstPtr input_;
  typename PointCloud<PointNT>::ConstPtr normals_;
  double radius_ = 0;
};

template <class PointInT> class UniformSampling {
 public:
  void setInputCloud(const typename PointCloud<PointInT>::ConstPtr &c) { input_ = c; }
  void setRadiusSearch(double leaf) { leaf_ = leaf; }
  void compute(PointCloud<int> &out) {
    out.clear();
    ope_ctx *ctx = default_context();
    if (!ctx || !input_) return;
    auto dev = upload(*input_, false);
    if (!dev->h) return;
    std::vector<int32_t> idx(input_->size());
    size_t n = 0;
    if (ope_uniform_sampling(ctx, dev->h, (float)leaf_, idx.data(), &n) != OPE_OK) { log_error("UniformSampling", ctx); return; }
    out.points.assign(idx.begin(), idx.begin() + n);
    out.width = (uint32_t)n;
  }
 private:
  typename PointCloud<PointInT>::ConstPtr input_;
  double leaf_ = 0.01;
};

template <class P> inline void copyPointCloud(const PointCloud<P> &in, const std::vector<int> &indices, PointCloud<P> &out) {
  out.clear();
  out.points.reserve(indices.size());
  for (int i : indices) out.points.push_back(in.points[i]);
  out.width = (uint32_t)out.points.size();
}

// ------------------------------------------------------------------------------------------ filters
// pcl::removeNaNFromPointCloud (poseestimator.cpp:192-194); `index` maps output positions to input positions.
template <class P> inline void removeNaNFromPointCloud(const PointCloud<P> &in, PointCloud<P> &out, std::vector<int> &index) {
  index.clear();
  ope_ctx *ctx = default_context();
  auto dev = ctx ? upload(in, false) : std::shared_ptr<CloudHandle>();
  std::vector<int32_t> idx(in.size() + 1);
  size_t n = 0;
  if (!dev || !dev->h || ope_remove_nan(ctx, dev->h, idx.data(), &n) != OPE_OK) {
    if (ctx && dev && dev->h) log_error("removeNaNFromPointCloud", ctx);
    n = 0;
  }
  index.assign(idx.begin(), idx.begin() + n);
  PointCloud<P> tmp;
  tmp.points.reserve(n);
  for (size_t i = 0; i < n; ++i) tmp.points.push_back(in.points[index[i]]);
  tmp.width = (uint32_t)n;
  out = std::move(tmp);  // `in` and `out` may be the same object, as at poseestimator.cpp:193
  out.is_dense = true;
}

// pcl::PassThrough (processingpcd.cpp:13-33): one field per object, inclusive limits.
template <class PointT> class PassThrough {
 public:
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr &c) { input_ = c; }
  void setFilterFieldName(const std::string &f) { field_ = f; }
  void setFilterLimits(float lo, float hi) { lo_ = lo; hi_ = hi; }
  void filter(PointCloud<PointT> &out) {
    PointCloud<PointT> tmp;
    ope_ctx *ctx = default_context();
    if (ctx && input_ && !input_->empty()) {
      float lo[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}, hi[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
      const int d = field_ == "x" ? 0 : field_ == "y" ? 1 : field_ == "z" ? 2 : -1;
      if (d >= 0) { lo[d] = lo_; hi[d] = hi_; }   // an unknown / empty field name only drops non-finite points
      auto dev = upload(*input_, false);
      std::vector<int32_t> idx(input_->size());
      size_t n = 0;
      if (dev->h && ope_pass_through(ctx, dev->h, lo, hi, idx.data(), &n) != OPE_OK) { log_error("PassThrough", ctx); n = 0; }
      tmp.points.reserve(n);
      for (size_t i = 0; i < n; ++i) tmp.points.push_back(input_->points[idx[i]]);
      tmp.width = (uint32_t)n;
    }
    out = std::move(tmp);
  }
 private:
  typename PointCloud<PointT>::ConstPtr input_;
  std::string field_;
  float lo_ = -FLT_MAX, hi_ = FLT_MAX;
};

// pcl::VoxelGrid (processingpcd.cpp:44-59): centroids in ascending voxel index; point types with a colour get the
// channel-wise mean of their voxel's colours (PCL's downsample_all_data_ default), other fields are default-initialised.
template <class PointT> class VoxelGrid {
 public:
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr &c) { input_ = c; }
  void setLeafSize(float lx, float ly, float lz) { leaf_[0] = lx; leaf_[1] = ly; leaf_[2] = lz; }
  void filter(PointCloud<PointT> &out) {
    PointCloud<PointT> tmp;
    ope_ctx *ctx = default_context();
    if (ctx && input_ && !input_->empty()) {
      auto dev = upload(*input_, false);
      std::vector<float> xyz(3 * input_->size());
      std::vector<uint32_t> rgb_in, rgb_out;
      constexpr ptrdiff_t rgb_off = point_traits<PointT>::rgb_offset;
      if (rgb_off >= 0) {
        rgb_in.resize(input_->size());
        rgb_out.resize(input_->size());
        for (size_t i = 0; i < input_->size(); ++i)
          std::memcpy(&rgb_in[i], reinterpret_cast<const unsigned char *>(&input_->points[i]) + rgb_off, 4);
      }
      size_t n = 0;
      const int rc = !dev->h ? OPE_EINVAL
                     : rgb_off >= 0 ? ope_voxel_grid_rgb(ctx, dev->h, leaf_, rgb_in.data(), xyz.data(), rgb_out.data(), &n)
                                    : ope_voxel_grid(ctx, dev->h, leaf_, xyz.data(), &n);
      if (rc == OPE_ERANGE) {  // voxel_grid.hpp: warn and hand the input back
        std::fprintf(stderr, "[ope::VoxelGrid::applyFilter] Leaf size is too small for the input dataset. Integer indices would overflow.\n");
        out = *input_;
        return;
      }
      if (rc != OPE_OK) { log_error("VoxelGrid", ctx); n = 0; }
      tmp.points.resize(n);
      for (size_t i = 0; i < n; ++i) {
        tmp.points[i].x = xyz[3 * i]; tmp.points[i].y = xyz[3 * i + 1]; tmp.points[i].z = xyz[3 * i + 2];
        if (rgb_off >= 0) std::memcpy(reinterpret_cast<unsigned char *>(&tmp.points[i]) + rgb_off, &rgb_out[i], 4);
      }
      tmp.width = (uint32_t)n;
    }
    out = std::move(tmp);
  }
 private:
  typename PointCloud<PointT>::ConstPtr input_;
  float leaf_[3] = {0.01f, 0.01f, 0.01f};
};

// pcl::StatisticalOutlierRemoval (ProcessingPcd::getOutlierRemove, DetectAndLocalize processingpcd.cpp:62-77)
template <class PointT> class StatisticalOutlierRemoval {
 public:
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr &c) { input_ = c; }
  void setMeanK(int k) { mean_k_ = k; }
  void setStddevMulThresh(double m) { mul_ = m; }
  void filter(PointCloud<PointT> &out) {
    PointCloud<PointT> tmp;
    ope_ctx *ctx = default_context();
    if (ctx && input_ && !input_->empty()) {
      auto dev = upload(*input_, false);
      std::vector<int32_t> idx(input_->size());
      size_t n = 0;
      if (dev->h && ope_statistical_outlier_removal(ctx, dev->h, mean_k_, mul_, idx.data(), &n, nullptr) != OPE_OK) { log_error("StatisticalOutlierRemoval", ctx); n = 0; }
      tmp.points.reserve(n);
      for (size_t i = 0; i < n; ++i) tmp.points.push_back(input_->points[idx[i]]);
      tmp.width = (uint32_t)n;
    }
    out = std::move(tmp);
  }
 private:
  typename PointCloud<PointT>::ConstPtr input_;
  int mean_k_ = 1;      // PCL's defaults
  double mul_ = 0.0;
};

template <class PointSource, class PointTarget, class FeatureT> class SampleConsensusInitialAlignment {
 public:
  SampleConsensusInitialAlignment() { ope_sacia_default_params(&p_); p_.max_iterations = 10; p_.nr_samples = 3; p_.k_correspondences = 10; p_.min_sample_dist = 0.f; p_.max_corr_dist = std::sqrt(DBL_MAX); }
  void setInputSource(const typename PointCloud<PointSource>::ConstPtr &c) { src_ = c; }
  void setInputTarget(const typename PointCloud<PointTarget>::ConstPtr &c) { tgt_ = c; }
  void setSourceFeatures(const typename PointCloud<FeatureT>::ConstPtr &f) { sf_ = f; }
  void setTargetFeatures(const typename PointCloud<FeatureT>::ConstPtr &f) { tf_ = f; }
  void setMaximumIterations(int n) { p_.max_iterations = n; }
  void setNumberOfSamples(int n) { p_.nr_samples = n; }
  void setCorrespondenceRandomness(int k) { p_.k_correspondences = k; }
  void setMaxCorrespondenceDistance(double d) { p_.max_corr_dist = d; }
  void setMinSampleDistance(float d) { p_.min_sample_dist = d; }
  void setSeed(uint64_t s) { p_.seed = s; }  // PCL draws from unseeded rand(): made explicit here
  void align(PointCloud<PointSource> &output) {
    final_ = Matrix4f::Identity();
    converged_ = false;
    ope_ctx *ctx = default_context();
    if (!ctx || !src_ || !tgt_ || !sf_ || !tf_ || sf_->size() != src_->size() || tf_->size() != tgt_->size()) return;
    auto ds = upload(*src_, false), dt = upload(*tgt_, false);
    IndexHandle ix;
    if (!ds->h || !dt->h || ope_index_build(ctx, dt->h, nullptr, &ix.h) != OPE_OK) return;
    double err = 0;
    int32_t best = -1;
    if (ope_sacia(ctx, ds->h, &sf_->points[0].histogram[0], dt->h, ix.h, &tf_->points[0].histogram[0], &p_, nullptr,
                  final_.m, &err, &best) != OPE_OK) { log_error("SampleConsensusInitialAlignment", ctx); return; }
    converged_ = best >= 0;
    error_ = err;
    transformPointCloud(*src_, output, final_);
  }
  Matrix4f getFinalTransformation() const { return final_; }
  bool hasConverged() const { return converged_; }
  double getLowestError() const { return error_; }
 private:
  typename PointCloud<PointSource>::ConstPtr src_;
  typename PointCloud<PointTarget>::ConstPtr tgt_;
  typename PointCloud<FeatureT>::ConstPtr sf_, tf_;
  ope_sacia_params p_;
  Matrix4f final_ = Matrix4f::Identity();
  bool converged_ = false;
  double error_ = 0;
};

}  // namespace compat
}  // namespace ope
