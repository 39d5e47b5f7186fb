// example_buildmodel.cpp — BuildModel's RegMeshPcd::getIcpNormal (BuildModel/src/regmeshpcd.cpp:63-206) and the
// accumulate step of registerPointClouds (:240-258) written against the façade, to show that the PCL call shapes are
// preserved for this program too:
//   NormalEstimation(k = 12) on source and target                       :72-90
//   CorrespondenceEstimationNormalShooting(k = 20)                      :139-145
//   CorrespondenceRejectorSurfaceNormal(threshold)                      :148-158
//   TransformationEstimationPointToPlane                                :162
//   IterativeClosestPointWithNormals, eps 1e-8 / 1e-8, max iterations   :166-196
//   transformPointCloud(source) + target                                :203, :254
// Build:  g++ -std=c++17 -Iinclude include/ope/example_buildmodel.cpp -Lobject-pose-estimation_amd -lope_hip
// Needs one MI355X at run time (there is no CPU fallback); exits 0 when the two views are registered.
#include <cmath>
#include <cstdio>
#include <random>

#include "pcl_compat.hpp"

namespace pcl = ope::compat;
typedef pcl::PointXYZRGB PointTReg;

// the part of an asymmetric star-shaped body (~0.15 m across) that faces the direction (vx, vy, vz)
static pcl::PointCloud<PointTReg>::Ptr make_view(int n, unsigned seed, double vx, double vy, double vz) {
  std::mt19937_64 rng(seed);
  std::normal_distribution<double> g(0.0, 1.0);
  pcl::PointCloud<PointTReg>::Ptr c(new pcl::PointCloud<PointTReg>);
  while ((int)c->size() < n) {
    double x = g(rng), y = g(rng), z = g(rng);
    const double l = std::sqrt(x * x + y * y + z * z);
    x /= l; y /= l; z /= l;
    if (x * vx + y * vy + z * vz < 0.1) continue;   // back-face culled (the body is star-shaped: normal ~ direction)
    const double r = 0.07 * (1.0 + 0.25 * x * y + 0.3 * z * z * x + 0.2 * std::sin(3 * y) * z);
    PointTReg p;
    p.x = (float)(r * x); p.y = (float)(r * y * 0.8); p.z = (float)(r * z * 0.6);
    c->push_back(p);
  }
  return c;
}

static pcl::PointCloud<PointTReg>::Ptr getIcpNormal(const pcl::PointCloud<PointTReg>::Ptr &p_cloudSource,
                                                    const pcl::PointCloud<PointTReg>::Ptr &p_cloudTarget, float corrRejThreshNormAngle,
                                                    int p_maxIterations, pcl::Matrix4f &transformIcpNormal, double &score) {
  typedef pcl::PointXYZRGBNormal PN;
  pcl::PointCloud<PN>::Ptr cloudSourceWithNormal(new pcl::PointCloud<PN>), cloudTargetWithNormal(new pcl::PointCloud<PN>);
  pcl::PointCloud<PN> cloudIcpNormal;
  pcl::NormalEstimation<PointTReg, PN> normEst;
  pcl::search::KdTree<PointTReg>::Ptr kdtree(new pcl::search::KdTree<PointTReg>);
  normEst.setSearchMethod(kdtree);
  normEst.setKSearch(12);
  normEst.setInputCloud(p_cloudSource); normEst.compute(*cloudSourceWithNormal);
  normEst.setInputCloud(p_cloudTarget); normEst.compute(*cloudTargetWithNormal);
  for (size_t i = 0; i < p_cloudSource->size(); ++i) {   // pcl::copyPointCloud(xyz) into the normal clouds (:84,:90)
    (*cloudSourceWithNormal)[i].x = (*p_cloudSource)[i].x; (*cloudSourceWithNormal)[i].y = (*p_cloudSource)[i].y; (*cloudSourceWithNormal)[i].z = (*p_cloudSource)[i].z;
  }
  for (size_t i = 0; i < p_cloudTarget->size(); ++i) {
    (*cloudTargetWithNormal)[i].x = (*p_cloudTarget)[i].x; (*cloudTargetWithNormal)[i].y = (*p_cloudTarget)[i].y; (*cloudTargetWithNormal)[i].z = (*p_cloudTarget)[i].z;
  }
  typedef pcl::registration::CorrespondenceEstimationNormalShooting<PN, PN, PN> NS;
  NS::Ptr corrEstNormShoot(new NS);
  corrEstNormShoot->setInputSource(cloudSourceWithNormal); corrEstNormShoot->setSourceNormals(cloudSourceWithNormal);
  corrEstNormShoot->setInputTarget(cloudTargetWithNormal);
  corrEstNormShoot->setKSearch(20);
  pcl::registration::CorrespondenceRejectorSurfaceNormal::Ptr corrRejSurNorm(new pcl::registration::CorrespondenceRejectorSurfaceNormal);
  corrRejSurNorm->initializeDataContainer<PN, PN>();
  corrRejSurNorm->setThreshold(corrRejThreshNormAngle);
  pcl::registration::TransformationEstimationPointToPlane<PN, PN>::Ptr transfEstpointToPlane(
      new pcl::registration::TransformationEstimationPointToPlane<PN, PN>);
  pcl::IterativeClosestPointWithNormals<PN, PN> icpNorm;
  icpNorm.setInputSource(cloudSourceWithNormal);
  icpNorm.setInputTarget(cloudTargetWithNormal);
  icpNorm.setMaximumIterations(p_maxIterations);
  icpNorm.setTransformationEpsilon(1e-8);
  icpNorm.setEuclideanFitnessEpsilon(1e-8);
  icpNorm.setCorrespondenceEstimation(corrEstNormShoot);
  icpNorm.addCorrespondenceRejector(corrRejSurNorm);
  icpNorm.setTransformationEstimation(transfEstpointToPlane);
  icpNorm.align(cloudIcpNormal);
  score = icpNorm.getFitnessScore();
  std::printf("ICP converged with score: %g after %d iterations\n", score, icpNorm.getNumberOfIterations());
  transformIcpNormal = icpNorm.getFinalTransformation();
  pcl::PointCloud<PointTReg>::Ptr cloudAligned(new pcl::PointCloud<PointTReg>);
  pcl::transformPointCloud(*p_cloudSource, *cloudAligned, transformIcpNormal);
  return cloudAligned;
}

int main() {
  // two overlapping views of the same body; the second one seen from 11 degrees further round and nudged
  pcl::PointCloud<PointTReg>::Ptr view0 = make_view(30000, 1, 1.0, 0.0, 0.3), view1raw = make_view(30000, 2, 0.98, 0.19, 0.3),
                                  view1(new pcl::PointCloud<PointTReg>);
  pcl::Matrix4f nudge = pcl::Matrix4f::Identity();
  const double a = 0.03;
  nudge(0, 0) = (float)std::cos(a); nudge(0, 1) = (float)-std::sin(a); nudge(1, 0) = (float)std::sin(a); nudge(1, 1) = (float)std::cos(a);
  nudge(0, 3) = 0.003f; nudge(1, 3) = -0.002f; nudge(2, 3) = 0.001f;
  pcl::transformPointCloud(*view1raw, *view1, nudge);

  pcl::Matrix4f T;
  double score = 0;
  pcl::PointCloud<PointTReg>::Ptr cloudAlignedIcp = getIcpNormal(view0, view1, 0.7f, 500, T, score);
  // *cloudAlignedIcp += *cloudTarget (regmeshpcd.cpp:254)
  for (const auto &p : view1->points) cloudAlignedIcp->push_back(p);
  double err = 0;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) err += (double)(T(r, c) - nudge(r, c)) * (T(r, c) - nudge(r, c));
  err = std::sqrt(err);
  std::printf("accumulated cloud: %zu points; |T - nudge|_F = %g\n", cloudAlignedIcp->size(), err);
  return (cloudAlignedIcp->size() == 60000 && score < 1e-5 && err < 2e-2) ? 0 : 1;
}
