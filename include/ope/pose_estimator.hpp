// pose_estimator.hpp — the reference's PoseEstimator (DetectAndLocalize/include/poseestimator.h:47-84,
// src/poseestimator.cpp:3-448) on the façade: same public interface, argument meaning, cross-frame state and guard
// behaviour, with every PCL object replaced by its ope::compat counterpart (GPU through the C ABI).
//
//     ope::PoseEstimator poseEstimator;                                   // rosinterface.h:54 holds one per process
//     pose = poseEstimator.estimateFinalPose(cloudSource, cloudTargetSeg, fitnessScore, alignedStrength);   // rosinterface.cpp:250
//
// What the class keeps from the reference, on purpose:
//   * coarse stage only while the previous fine fit scored worse than 1e-4 (poseestimator.cpp:399);
//   * pose = coarsePose * finePose and finalPose = rigidmodelPose * pose, products in the reference's order (quirk Q4);
//   * alignedSource / cloudModel / firstTimePose / fitnessScoreFine / alignedStrength survive between frames;
//   * "< 10 target features" and "< 100 target points" return identity exactly where the reference does;
//   * the caller's source cloud is overwritten with alignedSource (:441), estimateFinePose replaces its argument (:360);
//   * pcl::ScopeTime("Initial Alignment") / ("Final Alignment") around the two align() calls (:61,:349).
// Two knobs PCL does not have: the SAC-IA stream is seeded explicitly (PCL draws from an unseeded rand(); the
// k-th coarse call uses seed + k) and the self-occlusion rejector is opt-in (in the reference it depends on
// the PCL version and on an ODR accident).
#pragma once

#include "pcl_compat.hpp"

namespace ope {

class PoseEstimator {
 public:
  typedef compat::PointXYZRGB PointT;
  typedef compat::PointCloud<PointT> Cloud;
  typedef compat::PointCloud<compat::Normal> NormalCloud;
  typedef compat::PointCloud<compat::FPFHSignature33> FeatureCloud;
  typedef compat::PointCloud<compat::PointXYZRGBNormal> CloudN;
  typedef compat::Matrix4f Matrix4f;

  PoseEstimator() : alignedSource(new Cloud), cloudModel(new Cloud) {}

  void setSacIaSeed(uint64_t seed) { sacia_seed_ = seed; }
  void setUseSelfOccludedRejector(bool on) { use_self_occluded_ = on; }
  // diagnostics of the last estimateFinalPose call
  Matrix4f lastCoarsePose() const { return last_coarse_; }
  Matrix4f lastFinePose() const { return last_fine_; }
  Matrix4f lastRigidModelPose() const { return last_rigid_; }
  int lastIcpIterations() const { return last_icp_iterations_; }
  int coarseCalls() const { return coarse_calls_; }

  // :131-158  UniformSampling(radius = leafSize) -> the survivors -> NormalEstimation(k = 30) on them
  void subSampleAndCalculateNormals(Cloud::Ptr p_inputCloud, Cloud::Ptr &p_inputCloudSubSampled, NormalCloud::Ptr &p_inputCloudSubSampledNormal,
                                    double leafSize) {
    p_inputCloudSubSampled.reset(new Cloud);
    compat::PointCloud<int> keyPointIndices;
    uniSamp.setInputCloud(p_inputCloud);
    uniSamp.setRadiusSearch(leafSize);
    uniSamp.compute(keyPointIndices);
    compat::copyPointCloud(*p_inputCloud, keyPointIndices.points, *p_inputCloudSubSampled);
    p_inputCloudSubSampledNormal.reset(new NormalCloud);
    normEst.setSearchMethod(std::make_shared<compat::search::KdTree<PointT>>());
    normEst.setKSearch(30);
    normEst.setInputCloud(p_inputCloudSubSampled);
    normEst.compute(*p_inputCloudSubSampledNormal);
  }

  // :110-128  key points at 1 cm REPLACE the input cloud; FPFH with a 3 cm radius on them
  void getFpfhFeatures(Cloud::Ptr &p_inputCloud, FeatureCloud::Ptr &cloudFeatures, Cloud::Ptr &cloudKeyPoints) {
    Cloud::Ptr sub;
    NormalCloud::Ptr nrm;
    subSampleAndCalculateNormals(p_inputCloud, sub, nrm, 0.01);
    *p_inputCloud = *sub;
    cloudKeyPoints = sub;
    cloudFeatures.reset(new FeatureCloud);
    fpfhEstimation.setInputCloud(sub);
    fpfhEstimation.setRadiusSearch(0.03);
    fpfhEstimation.setInputNormals(nrm);
    fpfhEstimation.compute(*cloudFeatures);
  }

  // :16-73
  Matrix4f estimateCoarsePose(Cloud::Ptr p_sourceCloud, Cloud::Ptr p_targetCloud) {
    Cloud::Ptr src(new Cloud(*p_sourceCloud)), tgt(new Cloud(*p_targetCloud)), srcKeys, tgtKeys;
    FeatureCloud::Ptr srcFeat, tgtFeat;
    getFpfhFeatures(src, srcFeat, srcKeys);
    getFpfhFeatures(tgt, tgtFeat, tgtKeys);
    if (tgtFeat->points.size() < 10) {   // :40-45
      *alignedSource = *p_sourceCloud;
      std::printf("NO target cloud in Initial Alignment\n");
      return Matrix4f::Identity();
    }
    compat::SampleConsensusInitialAlignment<PointT, PointT, compat::FPFHSignature33> sacia;
    sacia.setInputSource(src);
    sacia.setInputTarget(tgt);
    sacia.setSourceFeatures(srcFeat);
    sacia.setTargetFeatures(tgtFeat);
    sacia.setMaximumIterations(400);          // :55
    sacia.setNumberOfSamples(5);              // :56
    sacia.setCorrespondenceRandomness(5);     // :57
    sacia.setMaxCorrespondenceDistance(0.05); // :58
    sacia.setMinSampleDistance(0.01f);        // :59
    sacia.setSeed(sacia_seed_ + (uint64_t)coarse_calls_++);
    Cloud result;
    {
      compat::ScopeTime t("Initial Alignment");
      sacia.align(result);
    }
    const Matrix4f pose = sacia.getFinalTransformation();
    compat::transformPointCloud(*p_sourceCloud, *alignedSource, pose);   // :66-70, full resolution
    return pose;
  }

  // :161-379
  Matrix4f estimateFinePose(Cloud::Ptr &p_sourceCloud, Cloud::Ptr p_targetCloud) {
    Cloud::Ptr src(new Cloud), tgt(new Cloud);
    std::vector<int> index;
    compat::removeNaNFromPointCloud(*p_sourceCloud, *src, index);   // :186-194
    compat::removeNaNFromPointCloud(*p_targetCloud, *tgt, index);
    CloudN::Ptr srcPN = withNormals(src), tgtPN = withNormals(tgt);  // :196-216, sub-sampling at 8 mm
    if (tgtPN->points.size() < 100) {   // :218-223
      std::printf("NO target cloud in Final Alignment\n");
      return Matrix4f::Identity();
    }
    typedef compat::registration::CorrespondenceEstimationNormalShooting<compat::PointXYZRGBNormal, compat::PointXYZRGBNormal, compat::PointXYZRGBNormal> NS;
    NS::Ptr corrEstNormShoot(new NS);
    corrEstNormShoot->setInputSource(srcPN);
    corrEstNormShoot->setSourceNormals(srcPN);
    corrEstNormShoot->setInputTarget(tgtPN);
    corrEstNormShoot->setKSearch(20);   // :246
    compat::registration::CorrespondenceRejectorSurfaceNormal::Ptr corrRejSurNorm(new compat::registration::CorrespondenceRejectorSurfaceNormal);
    corrRejSurNorm->initializeDataContainer<compat::PointXYZRGBNormal, compat::PointXYZRGBNormal>();
    corrRejSurNorm->setThreshold(0.7);  // :272
    compat::registration::CorrespondenceRejectorSelfOccludedNormal::Ptr corrRejSelfNorm(new compat::registration::CorrespondenceRejectorSelfOccludedNormal);
    corrRejSelfNorm->setThreshold(0.6); // :291
    compat::registration::TransformationEstimationSVD<compat::PointXYZRGBNormal, compat::PointXYZRGBNormal>::Ptr transfEstSvd(
        new compat::registration::TransformationEstimationSVD<compat::PointXYZRGBNormal, compat::PointXYZRGBNormal>);
    compat::IterativeClosestPointWithNormals<compat::PointXYZRGBNormal, compat::PointXYZRGBNormal> icp;
    icp.setInputSource(srcPN);
    icp.setInputTarget(tgtPN);
    icp.setMaximumIterations(100);         // :322
    icp.setTransformationEpsilon(1e-8);    // :325
    icp.setEuclideanFitnessEpsilon(1e-8);  // :328
    icp.setCorrespondenceEstimation(corrEstNormShoot);
    icp.addCorrespondenceRejector(corrRejSurNorm);
    if (use_self_occluded_) icp.addCorrespondenceRejector(corrRejSelfNorm);
    icp.setTransformationEstimation(transfEstSvd);
    CloudN cloudAligned;
    {
      compat::ScopeTime t("Final Alignment");
      icp.align(cloudAligned);
    }
    fitnessScoreFine = icp.getFitnessScore();   // :354
    const Matrix4f pose = icp.getFinalTransformation();
    Cloud::Ptr moved(new Cloud);
    compat::transformPointCloud(*p_sourceCloud, *moved, pose);   // :358-360
    *p_sourceCloud = *moved;
    alignedStrength = icp.getAlignStrength();   // :363
    last_icp_iterations_ = icp.getNumberOfIterations();
    std::printf("Aligned Strength : %g\n", alignedStrength);
    return pose;
  }

  // :383-448
  Matrix4f estimateFinalPose(Cloud::Ptr &p_sourceCloud, Cloud::Ptr p_targetCloud, double &fitnessScore, double &alignStrength) {
    if (firstTimePose == 0) *cloudModel = *p_sourceCloud;   // the original model, kept for the re-anchoring fit
    ++firstTimePose;
    const bool have_target = !p_targetCloud->empty();
    Matrix4f coarsePose = Matrix4f::Identity(), finePose = Matrix4f::Identity();
    if (have_target && fitnessScoreFine > 0.0001) coarsePose = estimateCoarsePose(p_sourceCloud, p_targetCloud);
    if (have_target) finePose = estimateFinePose(alignedSource, p_targetCloud);
    const Matrix4f pose = coarsePose * finePose;   // :421
    // where the incoming source sits relative to the stored model: SVD fit over identity correspondences (:425-436)
    Matrix4f rigidmodelPose = Matrix4f::Identity();
    {
      compat::Correspondences corres(cloudModel->points.size());
      for (size_t i = 0; i < corres.size(); ++i) corres[i].index_query = corres[i].index_match = (int)i;
      if (!corres.empty() && p_sourceCloud->size() >= corres.size()) svd.estimateRigidTransformation(*cloudModel, *p_sourceCloud, corres, rigidmodelPose);
    }
    finalPose = rigidmodelPose * pose;   // :439
    *p_sourceCloud = *alignedSource;     // :441
    fitnessScore = fitnessScoreFine;
    alignStrength = alignedStrength;
    last_coarse_ = coarsePose; last_fine_ = finePose; last_rigid_ = rigidmodelPose;
    return finalPose;
  }

 private:
  // sub-sample at 8 mm, normals (k = 30), xyz + normal packed into PointXYZRGBNormal, NaN normals dropped (:196-216)
  CloudN::Ptr withNormals(const Cloud::Ptr &cloud) {
    Cloud::Ptr sub;
    NormalCloud::Ptr nrm;
    subSampleAndCalculateNormals(cloud, sub, nrm, 0.008);
    CloudN::Ptr out(new CloudN);
    for (size_t i = 0; i < sub->size() && i < nrm->size(); ++i) {
      const compat::Normal &n = (*nrm)[i];
      if (!std::isfinite(n.normal_x) || !std::isfinite(n.normal_y) || !std::isfinite(n.normal_z)) continue;
      compat::PointXYZRGBNormal q;
      q.x = (*sub)[i].x; q.y = (*sub)[i].y; q.z = (*sub)[i].z; q.rgb = (*sub)[i].rgb;
      q.normal_x = n.normal_x; q.normal_y = n.normal_y; q.normal_z = n.normal_z; q.curvature = n.curvature;
      out->push_back(q);
    }
    return out;
  }

  // state that crosses frames (poseestimator.h:50-53)
  Cloud::Ptr alignedSource, cloudModel;
  double fitnessScoreFine = 10.0, alignedStrength = 0.0;   // "random high value" (:6)
  Matrix4f finalPose = Matrix4f::Identity();
  int firstTimePose = 0;
  // objects (poseestimator.h:56-60)
  compat::FPFHEstimation<PointT, compat::Normal, compat::FPFHSignature33> fpfhEstimation;
  compat::UniformSampling<PointT> uniSamp;
  compat::NormalEstimation<PointT, compat::Normal> normEst;
  compat::registration::TransformationEstimationSVD<PointT, PointT> svd;
  // knobs and diagnostics
  uint64_t sacia_seed_ = 1;
  int coarse_calls_ = 0;
  bool use_self_occluded_ = false;
  Matrix4f last_coarse_ = Matrix4f::Identity(), last_fine_ = Matrix4f::Identity(), last_rigid_ = Matrix4f::Identity();
  int last_icp_iterations_ = 0;
};

}  // namespace ope
