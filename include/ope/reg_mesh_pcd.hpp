// reg_mesh_pcd.hpp — BuildModel's RegMeshPcd (BuildModel/include/regmeshpcd.h:27-49, src/regmeshpcd.cpp:8-271) on the
// façade: getIcp (stock point-to-point ICP), getIcpNormal (normals k = 12, normal shooting k = 20, surface-normal
// rejector, point-to-plane LM estimator, eps 1e-8 / 1e-8) and registerPointClouds (sequential accumulate-and-register).
// generateMesh (:273-343) is surface reconstruction, out of scope.
//
// Kept from the reference on purpose:
//   * p_maxCorrDist only reaches a stand-alone determineCorrespondences call whose result is discarded (:140-159); the ICP
//     object itself keeps PCL's default correspondence distance;
//   * cloudTemp aliases cloudVector[0] (:229): `*cloudTemp = *cloudAlignedIcp` overwrites the caller's first frame with the
//     accumulated cloud, pair after pair;
//   * the accumulated cloud is `aligned source + target` in that order (:254-258), colours carried along.
#pragma once

#include <cstdio>
#include <vector>

#include "pcl_compat.hpp"

namespace ope {

class RegMeshPcd {
 public:
  typedef compat::PointXYZRGB PointTReg;
  typedef compat::PointCloud<PointTReg> Cloud;
  typedef compat::PointXYZRGBNormal PN;
  typedef compat::PointCloud<PN> CloudN;

  // per-pair results, for callers that want more than the cloud
  struct Pair { compat::Matrix4f T; int iterations; bool converged; double fitness; };
  const std::vector<Pair> &pairs() const { return pairs_; }

  // :8-59  plain IterativeClosestPoint (stock icp.h in the reference)
  Cloud::Ptr getIcp(Cloud::Ptr p_cloudSource, Cloud::Ptr p_cloudTarget, float p_maxCorrDist, float p_ransacStatOutThresh, int p_maxIterations) {
    compat::IterativeClosestPoint<PointTReg, PointTReg> icp;
    icp.setInputSource(p_cloudSource);
    icp.setInputTarget(p_cloudTarget);
    icp.setMaxCorrespondenceDistance(p_maxCorrDist);                  // :23
    icp.setRANSACOutlierRejectionThreshold(p_ransacStatOutThresh);    // :26 (accepted; the reference's ICP never reads it)
    icp.setMaximumIterations(p_maxIterations);                        // :29
    icp.setTransformationEpsilon(1e-8);                               // :32
    icp.setEuclideanFitnessEpsilon(1e-8);                             // :35
    Cloud::Ptr cloudAligned(new Cloud);
    icp.align(*cloudAligned);
    std::printf("ICP converged with score: %g\n", icp.getFitnessScore());
    return cloudAligned;
  }

  // :63-206
  Cloud::Ptr getIcpNormal(Cloud::Ptr p_cloudSource, Cloud::Ptr p_cloudTarget, float p_maxCorrDist, float /*p_ransacStatOutThresh*/, int p_maxIterations) {
    CloudN::Ptr cloudSourceWithNormal = withNormals(p_cloudSource), cloudTargetWithNormal = withNormals(p_cloudTarget);   // :72-90
    typedef compat::registration::CorrespondenceEstimationNormalShooting<PN, PN, PN> NS;
    NS::Ptr corrEstNormShoot(new NS);
    corrEstNormShoot->setInputSource(cloudSourceWithNormal);
    corrEstNormShoot->setSourceNormals(cloudSourceWithNormal);
    corrEstNormShoot->setInputTarget(cloudTargetWithNormal);
    corrEstNormShoot->setKSearch(20);                                 // :144
    (void)p_maxCorrDist;   // :145 passes it to a stand-alone determineCorrespondences whose result is never used
    compat::registration::CorrespondenceRejectorSurfaceNormal::Ptr corrRejSurNorm(new compat::registration::CorrespondenceRejectorSurfaceNormal);
    corrRejSurNorm->initializeDataContainer<PN, PN>();
    corrRejSurNorm->setThreshold(corrRejThreshNormAngle);             // :158
    compat::registration::TransformationEstimationPointToPlane<PN, PN>::Ptr transfEstpointToPlane(
        new compat::registration::TransformationEstimationPointToPlane<PN, PN>);                                          // :162 (LM)
    compat::IterativeClosestPointWithNormals<PN, PN> icpNorm;
    icpNorm.setInputSource(cloudSourceWithNormal);
    icpNorm.setInputTarget(cloudTargetWithNormal);
    icpNorm.setMaximumIterations(p_maxIterations);                    // :179
    icpNorm.setTransformationEpsilon(1e-8);                           // :182
    icpNorm.setEuclideanFitnessEpsilon(1e-8);                         // :184
    icpNorm.setCorrespondenceEstimation(corrEstNormShoot);            // :187
    icpNorm.addCorrespondenceRejector(corrRejSurNorm);                // :190
    icpNorm.setTransformationEstimation(transfEstpointToPlane);       // :193
    CloudN cloudIcpNormal;
    icpNorm.align(cloudIcpNormal);                                    // :196
    const double score = icpNorm.getFitnessScore();
    std::printf("ICP converged with score: %g\n", score);            // :198
    const compat::Matrix4f transformIcpNormal = icpNorm.getFinalTransformation();
    pairs_.push_back(Pair{transformIcpNormal, icpNorm.getNumberOfIterations(), icpNorm.hasConverged(), score});
    Cloud::Ptr cloudAligned(new Cloud);
    compat::transformPointCloud(*p_cloudSource, *cloudAligned, transformIcpNormal);   // :203
    return cloudAligned;
  }

  // :210-271
  Cloud::Ptr registerPointClouds(std::vector<Cloud::Ptr> &cloudVector, float maxCorrDist, float corrRejThresh, int maxIter) {
    const float ransacStatOutThresh2 = 0.02f;   // :222
    corrRejThreshNormAngle = corrRejThresh;     // :227
    pairs_.clear();
    Cloud::Ptr out(new Cloud);
    if (cloudVector.empty()) return out;
    Cloud::Ptr cloudTemp = cloudVector[0];      // :229 (aliases the caller's first frame)
    for (size_t i = 0; i + 1 < cloudVector.size(); ++i) {
      std::printf("ICP between frame %zu and %zu\n", i, i + 1);
      Cloud::Ptr cloudSource = cloudTemp, cloudTarget = cloudVector[i + 1];
      Cloud::Ptr cloudAlignedIcp = getIcpNormal(cloudSource, cloudTarget, maxCorrDist, ransacStatOutThresh2, maxIter);   // :251
      for (const auto &p : cloudTarget->points) cloudAlignedIcp->push_back(p);                                           // :254  *aligned += *target
      *cloudTemp = *cloudAlignedIcp;                                                                                    // :258
    }
    *out = *cloudTemp;   // :266
    return out;
  }

 private:
  // NormalEstimation<PointXYZRGB, PointXYZRGBNormal>(k = 12), then copyPointCloud of xyz / rgb into it (:72-90)
  static CloudN::Ptr withNormals(const Cloud::Ptr &c) {
    CloudN::Ptr n(new CloudN);
    compat::NormalEstimation<PointTReg, PN> normEst;
    normEst.setSearchMethod(std::make_shared<compat::search::KdTree<PointTReg>>());
    normEst.setKSearch(12);
    normEst.setInputCloud(c);
    normEst.compute(*n);
    for (size_t i = 0; i < c->size() && i < n->size(); ++i) {
      (*n)[i].x = (*c)[i].x; (*n)[i].y = (*c)[i].y; (*n)[i].z = (*c)[i].z; (*n)[i].rgb = (*c)[i].rgb;
    }
    return n;
  }

  float corrRejThreshNormAngle = 0.7f;
  std::vector<Pair> pairs_;
};

}  // namespace ope
