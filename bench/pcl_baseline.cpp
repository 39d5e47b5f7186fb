// pcl_baseline.cpp — LIVE A/B harness of the hot path against the Point Cloud Library itself.
//
// ONE source, two builds:
//   make -C bench pcl      real PCL (found through pkg-config; nominally 1.7.2, what the reference links, CMakeLists.txt:8).
//                          With REF=<reference checkout> the reference's own vendored icp_mod.h /
//                          correspondence_estimation_*.h (DetectAndLocalize/include/pcl/registration) are the ICP classes,
//                          otherwise PCL's stock ones.
//   make -C bench facade   the same calls against include/ope/pcl_compat.hpp -> libope_hip.so (needs one MI355X to run).
// Both read the same two .pcd files and the same initial guess (tools/make_ab_inputs.py writes them from the bench's
// synthetic generator), run the same fixed number of ICP iterations with the convergence tests disabled, and print one JSON
// line: implementation, milliseconds per iteration, the final transform, fitness score and correspondence count.
// tools/ab_compare.py puts two such lines side by side (|T_a - T_b|_F <= 1e-4 = BASELINE.json's tolerance).
//
// This image has no PCL, Eigen, FLANN or Boost, so only the facade build is made and run by the tests here
// (tests/test_gpu_facade.py::test_pcl_ab_harness_*); the PCL build is for a maintainer's box and is the one thing that can
// move the oracle's status off "parity unpinned" (DESIGN.md §2).  Modes:
//   nn   IterativeClosestPoint<PointXYZ>: 1-NN correspondences + TransformationEstimationSVD — the bench's metric
//        (icp_mod.hpp:119-272, correspondence_estimation_mod.hpp:127-213)
//   nnfix  `nn` with eight given pairs injected through setFixedCorrespondences (icp_mod.h:268: pair i = source point i * ns / 8,
//        model point i * nt / 8) — the reference's vendored classes and the facade have it, stock PCL does not
//   ns   estimateFinePose's configuration (poseestimator.cpp:161-379): normals k = 30, IterativeClosestPointWithNormals with
//        CorrespondenceEstimationNormalShooting(k = 20) + CorrespondenceRejectorSurfaceNormal(0.7) + SVD
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

#ifdef OPE_FACADE
#include "ope/pcl_compat.hpp"
#include "ope/pcd_io.hpp"
namespace pcl = ope::compat;
typedef pcl::Matrix4f Mat4;
static const char *kImpl = "ope_facade";
#else
#include <pcl/point_types.h>
#include <pcl/point_cloud.h>
#include <pcl/io/pcd_io.h>
#include <pcl/features/normal_3d.h>
#include <pcl/search/kdtree.h>
#ifdef OPE_REF_VENDORED   // the reference's modified ICP classes (same class names as PCL's, poseestimator.h:9) in its own include directory
#include <pcl/registration/icp_mod.h>
static const char *kImpl = "pcl_vendored";
#else
#include <pcl/registration/icp.h>
static const char *kImpl = "pcl";
#endif
#include <pcl/registration/correspondence_estimation_normal_shooting.h>   // poseestimator.h:25,30: PCL's own, also in the reference
#include <pcl/registration/correspondence_rejection_surface_normal.h>
#include <pcl/registration/transformation_estimation_svd.h>
typedef Eigen::Matrix4f Mat4;
#endif

static bool read_guess(const char *path, Mat4 &T) {
  std::ifstream f(path);
  if (!f) return false;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      float v;
      if (!(f >> v)) return false;
      T(r, c) = v;
    }
  return true;
}

// given: the caller's list of injected pairs after align() — its distance fields are written by the correspondence estimation
// of every iteration (impl/correspondence_estimation_mod.hpp:159), so they hold the last iteration's values
static void print_line(const char *mode, int iterations, double ms_per_iteration, const Mat4 &T, double fitness, bool converged, size_t ns, size_t nt,
                       const pcl::Correspondences *given = nullptr) {
  std::printf("{\"impl\": \"%s\", \"mode\": \"%s\", \"n_source\": %zu, \"n_target\": %zu, \"iterations\": %d, \"ms_per_iteration\": %.6f, \"converged\": %s, \"fitness\": %.9g, ",
              kImpl, mode, ns, nt, iterations, ms_per_iteration, converged ? "true" : "false", fitness);
  if (given) {
    std::printf("\"given_distance\": [");
    for (size_t i = 0; i < given->size(); ++i) std::printf("%s%.9g", i ? ", " : "", (double)(*given)[i].distance);
    std::printf("], ");
  }
  std::printf("\"T\": [");
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) std::printf("%s%.9g", (r || c) ? ", " : "", (double)T(r, c));
  std::printf("]}\n");
}

template <class Icp> static void disable_convergence_tests(Icp &icp, int iterations) {
  icp.setMaximumIterations(iterations);
  icp.setTransformationEpsilon(0.0);
  icp.setEuclideanFitnessEpsilon(0.0);
  icp.getConvergeCriteria()->setAbsoluteMSE(-1.0);   // PCL's default 1e-12 would end a run that has settled
}

static int run_nn(const std::string &scene_path, const std::string &model_path, const Mat4 &guess, int iterations, bool fixed_pairs) {
  typedef pcl::PointXYZ P;
  pcl::PointCloud<P>::Ptr scene(new pcl::PointCloud<P>), model(new pcl::PointCloud<P>);
  if (pcl::io::loadPCDFile(scene_path, *scene) < 0 || pcl::io::loadPCDFile(model_path, *model) < 0) return 2;
  pcl::IterativeClosestPoint<P, P> icp;
  icp.setInputSource(scene);   // the scene is registered onto the model (BASELINE.json: 1 M scene points vs 100 k model points)
  icp.setInputTarget(model);
  disable_convergence_tests(icp, iterations);
  pcl::Correspondences given;
  if (fixed_pairs) {
#if defined(OPE_FACADE) || defined(OPE_REF_VENDORED)
    for (int i = 0; i < 8; ++i) {
      pcl::Correspondence c;
      c.index_query = (int)((size_t)i * scene->size() / 8);
      c.index_match = (int)((size_t)i * model->size() / 8);
      given.push_back(c);
    }
    icp.setFixedCorrespondences(&given);
#else
    std::fprintf(stderr, "mode nnfix needs the reference's vendored icp_mod.h (make pcl REF=...): stock PCL has no setFixedCorrespondences\n");
    return 1;
#endif
  }
  pcl::PointCloud<P> out;
  const auto t0 = std::chrono::steady_clock::now();
  icp.align(out, guess);
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  print_line(fixed_pairs ? "nnfix" : "nn", iterations, ms / iterations, icp.getFinalTransformation(), icp.getFitnessScore(), icp.hasConverged(), scene->size(), model->size(),
             fixed_pairs ? &given : nullptr);
  return 0;
}

template <class P> static typename pcl::PointCloud<pcl::PointXYZRGBNormal>::Ptr with_normals(const typename pcl::PointCloud<P>::Ptr &c) {
  pcl::PointCloud<pcl::Normal>::Ptr n(new pcl::PointCloud<pcl::Normal>);
  pcl::NormalEstimation<P, pcl::Normal> ne;
  typename pcl::search::KdTree<P>::Ptr tree(new pcl::search::KdTree<P>);
  ne.setSearchMethod(tree);
  ne.setKSearch(30);   // subSampleAndCalculateNormals, poseestimator.cpp:153
  ne.setInputCloud(c);
  ne.compute(*n);
  pcl::PointCloud<pcl::PointXYZRGBNormal>::Ptr o(new pcl::PointCloud<pcl::PointXYZRGBNormal>);
  for (size_t i = 0; i < c->points.size(); ++i) {
    if (!std::isfinite(n->points[i].normal_x)) continue;   // removeNaNNormalsFromPointCloud
    pcl::PointXYZRGBNormal q;
    q.x = c->points[i].x; q.y = c->points[i].y; q.z = c->points[i].z;
    q.normal_x = n->points[i].normal_x; q.normal_y = n->points[i].normal_y; q.normal_z = n->points[i].normal_z;
    q.curvature = n->points[i].curvature;
    o->push_back(q);
  }
  return o;
}

static int run_ns(const std::string &scene_path, const std::string &model_path, const Mat4 &guess, int iterations) {
  typedef pcl::PointXYZRGB P;
  typedef pcl::PointXYZRGBNormal PN;
  pcl::PointCloud<P>::Ptr scene(new pcl::PointCloud<P>), model(new pcl::PointCloud<P>);
  if (pcl::io::loadPCDFile(scene_path, *scene) < 0 || pcl::io::loadPCDFile(model_path, *model) < 0) return 2;
  pcl::PointCloud<PN>::Ptr src = with_normals<P>(scene), tgt = with_normals<P>(model);
  // poseestimator.cpp:242-246 (normal shooting over the 20 nearest), :268-273 (surface-normal rejector at 0.7), :306 (SVD)
  typedef pcl::registration::CorrespondenceEstimationNormalShooting<PN, PN, PN> NS;
  NS::Ptr ce(new NS);
  ce->setKSearch(20);
  pcl::registration::CorrespondenceRejectorSurfaceNormal::Ptr rej(new pcl::registration::CorrespondenceRejectorSurfaceNormal);
  rej->setThreshold(0.7);
  pcl::registration::TransformationEstimationSVD<PN, PN>::Ptr est(new pcl::registration::TransformationEstimationSVD<PN, PN>);
  pcl::IterativeClosestPointWithNormals<PN, PN> icp;
  icp.setCorrespondenceEstimation(ce);
  icp.addCorrespondenceRejector(rej);
  icp.setTransformationEstimation(est);
  icp.setInputSource(src);
  icp.setInputTarget(tgt);
  disable_convergence_tests(icp, iterations);
  pcl::PointCloud<PN> out;
  const auto t0 = std::chrono::steady_clock::now();
  icp.align(out, guess);
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  print_line("ns", iterations, ms / iterations, icp.getFinalTransformation(), icp.getFitnessScore(), icp.hasConverged(), src->size(), tgt->size());
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s <scene.pcd> <model.pcd> <guess.txt: 16 numbers, row-major, or '-'> <iterations> [nn|nnfix|ns]\n", argv[0]);
    return 1;
  }
  Mat4 guess = Mat4::Identity();
  if (std::strcmp(argv[3], "-") != 0 && !read_guess(argv[3], guess)) { std::fprintf(stderr, "cannot read the guess '%s'\n", argv[3]); return 1; }
  const int iterations = std::max(1, std::atoi(argv[4]));
  const std::string mode = argc > 5 ? argv[5] : "nn";
  if (mode == "nn" || mode == "nnfix") return run_nn(argv[1], argv[2], guess, iterations, mode == "nnfix");
  if (mode == "ns") return run_ns(argv[1], argv[2], guess, iterations);
  std::fprintf(stderr, "unknown mode '%s'\n", mode.c_str());
  return 1;
}
