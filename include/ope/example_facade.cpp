// example_facade.cpp — the reference's PoseEstimator call sequence (DetectAndLocalize/src/poseestimator.cpp)
// written against the façade, to show that the PCL call shapes are preserved:
//   subSampleAndCalculateNormals (:131-158)  UniformSampling + NormalEstimation(k=30)
//   getFpfhFeatures            (:110-128)    FPFHEstimation(r=0.03)
//   estimateCoarsePose         (:16-73)      SampleConsensusInitialAlignment(400, 5, 5, 0.05, 0.01)
//   estimateFinePose           (:161-379)    IterativeClosestPointWithNormals + NormalShooting(k=20)
//                                            + SurfaceNormal rejector(0.7) + SVD, 100 it, eps 1e-8
//   estimateFinalPose          (:383-448)    TransformationEstimationSVD with identity correspondences
// Build:  g++ -std=c++17 -Iinclude include/ope/example_facade.cpp -Lobject-pose-estimation_amd -lope_hip
// Needs one MI355X at run time (there is no CPU fallback); exits 0 when the recovered pose is right.
#include <cmath>
#include <cstdio>
#include <random>

#include "pcl_compat.hpp"

namespace pcl = ope::compat;
typedef pcl::PointXYZRGB PointT;

static pcl::PointCloud<PointT>::Ptr make_surface(int n, unsigned seed) {
  std::mt19937_64 rng(seed);
  std::normal_distribution<double> g(0.0, 1.0);
  pcl::PointCloud<PointT>::Ptr c(new pcl::PointCloud<PointT>);
  for (int i = 0; i < n; ++i) {
    double x = g(rng), y = g(rng), z = g(rng);
    const double l = std::sqrt(x * x + y * y + z * z);
    x /= l; y /= l; z /= l;
    // asymmetric star-shaped body, ~0.2 m across
    const double r = 0.07 * (1.0 + 0.25 * x * y + 0.3 * z * z * x + 0.2 * std::sin(3 * y) * z) *
                     std::pow(std::pow(std::fabs(x), 2.5) + std::pow(std::fabs(y / 0.75), 2.5) + std::pow(std::fabs(z / 0.5), 2.5), -1 / 2.5);
    PointT p;
    p.x = (float)(r * x); p.y = (float)(r * y); p.z = (float)(r * z);
    c->push_back(p);
  }
  return c;
}

static void subSampleAndCalculateNormals(const pcl::PointCloud<PointT>::Ptr &in, pcl::PointCloud<PointT>::Ptr &sub,
                                         pcl::PointCloud<pcl::Normal>::Ptr &nrm, double leaf) {
  sub.reset(new pcl::PointCloud<PointT>);
  pcl::UniformSampling<PointT> uniSamp;
  uniSamp.setInputCloud(in);
  uniSamp.setRadiusSearch(leaf);
  pcl::PointCloud<int> keyPointIndices;
  uniSamp.compute(keyPointIndices);
  pcl::copyPointCloud(*in, keyPointIndices.points, *sub);
  nrm.reset(new pcl::PointCloud<pcl::Normal>);
  pcl::NormalEstimation<PointT, pcl::Normal> normEst;
  pcl::search::KdTree<PointT>::Ptr kdtree(new pcl::search::KdTree<PointT>);
  normEst.setSearchMethod(kdtree);
  normEst.setKSearch(30);
  normEst.setInputCloud(sub);
  normEst.compute(*nrm);
}

int main() {
  // model at the origin; scene = independent re-sampling of the model, posed in front of the sensor
  pcl::PointCloud<PointT>::Ptr model = make_surface(40000, 1), scene0 = make_surface(40000, 2), scene(new pcl::PointCloud<PointT>);
  pcl::Matrix4f gt = pcl::Matrix4f::Identity();
  const double a = 0.6, b = -0.35;
  gt(0, 0) = (float)std::cos(a); gt(0, 1) = (float)-std::sin(a); gt(1, 0) = (float)std::sin(a); gt(1, 1) = (float)std::cos(a);
  pcl::Matrix4f ry = pcl::Matrix4f::Identity();
  ry(0, 0) = (float)std::cos(b); ry(0, 2) = (float)std::sin(b); ry(2, 0) = (float)-std::sin(b); ry(2, 2) = (float)std::cos(b);
  gt = gt * ry;
  gt(0, 3) = 0.03f; gt(1, 3) = -0.02f; gt(2, 3) = 0.7f;
  pcl::transformPointCloud(*scene0, *scene, gt);

  // ---- BuildModel's ProcessingPcd::getPassThrough (z, y, x) and getDownSampled (processingpcd.cpp:8-52) on the scene
  {
    pcl::PointCloud<PointT>::Ptr fz(new pcl::PointCloud<PointT>), fzy(new pcl::PointCloud<PointT>), fzyx(new pcl::PointCloud<PointT>),
        ds(new pcl::PointCloud<PointT>);
    pcl::PassThrough<PointT> passThrough;
    passThrough.setInputCloud(scene); passThrough.setFilterFieldName("z"); passThrough.setFilterLimits(0.0f, 0.7f); passThrough.filter(*fz);
    passThrough.setInputCloud(fz); passThrough.setFilterFieldName("y"); passThrough.setFilterLimits(-1.0f, 1.0f); passThrough.filter(*fzy);
    passThrough.setInputCloud(fzy); passThrough.setFilterFieldName("x"); passThrough.setFilterLimits(-1.0f, 1.0f); passThrough.filter(*fzyx);
    pcl::VoxelGrid<PointT> voxGrid;
    voxGrid.setInputCloud(scene); voxGrid.setLeafSize(0.005f, 0.005f, 0.005f); voxGrid.filter(*ds);
    std::printf("PassThrough z<=0.7 kept %zu of %zu points; VoxelGrid(5 mm) %zu centroids\n", fzyx->size(), scene->size(), ds->size());
    if (fzyx->empty() || fzyx->size() >= scene->size() || fz->size() != fzyx->size() || ds->empty() || ds->size() >= scene->size()) return 4;
  }

  // ---- estimateCoarsePose
  pcl::PointCloud<PointT>::Ptr srcKey, tgtKey;
  pcl::PointCloud<pcl::Normal>::Ptr srcN, tgtN;
  subSampleAndCalculateNormals(model, srcKey, srcN, 0.01);
  subSampleAndCalculateNormals(scene, tgtKey, tgtN, 0.01);
  pcl::PointCloud<pcl::FPFHSignature33>::Ptr srcF(new pcl::PointCloud<pcl::FPFHSignature33>), tgtF(new pcl::PointCloud<pcl::FPFHSignature33>);
  pcl::FPFHEstimation<PointT, pcl::Normal, pcl::FPFHSignature33> fpfh;
  fpfh.setInputCloud(srcKey); fpfh.setRadiusSearch(0.03); fpfh.setInputNormals(srcN); fpfh.compute(*srcF);
  fpfh.setInputCloud(tgtKey); fpfh.setRadiusSearch(0.03); fpfh.setInputNormals(tgtN); fpfh.compute(*tgtF);
  if (tgtF->points.size() < 10) { std::printf("NO target cloud in Initial Alignment\n"); return 2; }
  pcl::SampleConsensusInitialAlignment<PointT, PointT, pcl::FPFHSignature33> sacia;
  sacia.setInputSource(srcKey); sacia.setInputTarget(tgtKey);
  sacia.setSourceFeatures(srcF); sacia.setTargetFeatures(tgtF);
  sacia.setMaximumIterations(400); sacia.setNumberOfSamples(5); sacia.setCorrespondenceRandomness(5);
  sacia.setMaxCorrespondenceDistance(0.05); sacia.setMinSampleDistance(0.01f);
  pcl::PointCloud<PointT> result;
  sacia.align(result);
  const pcl::Matrix4f coarse = sacia.getFinalTransformation();
  pcl::PointCloud<PointT>::Ptr aligned(new pcl::PointCloud<PointT>);
  pcl::transformPointCloud(*model, *aligned, coarse);

  // ---- estimateFinePose
  {  // remove NaN points (poseestimator.cpp:192-194)
    std::vector<int> index;
    pcl::removeNaNFromPointCloud(*aligned, *aligned, index);
    pcl::removeNaNFromPointCloud(*scene, *scene, index);
  }
  pcl::PointCloud<PointT>::Ptr s8, t8;
  pcl::PointCloud<pcl::Normal>::Ptr s8n, t8n;
  subSampleAndCalculateNormals(aligned, s8, s8n, 0.008);
  subSampleAndCalculateNormals(scene, t8, t8n, 0.008);
  pcl::PointCloud<pcl::PointXYZRGBNormal>::Ptr sPN(new pcl::PointCloud<pcl::PointXYZRGBNormal>), tPN(new pcl::PointCloud<pcl::PointXYZRGBNormal>);
  auto pack = [](const pcl::PointCloud<PointT> &p, const pcl::PointCloud<pcl::Normal> &n, pcl::PointCloud<pcl::PointXYZRGBNormal> &o) {
    for (size_t i = 0; i < p.size(); ++i) {
      if (!std::isfinite(n[i].normal_x)) continue;  // removeNaNNormalsFromPointCloud
      pcl::PointXYZRGBNormal q;
      q.x = p[i].x; q.y = p[i].y; q.z = p[i].z;
      q.normal_x = n[i].normal_x; q.normal_y = n[i].normal_y; q.normal_z = n[i].normal_z; q.curvature = n[i].curvature;
      o.push_back(q);
    }
  };
  pack(*s8, *s8n, *sPN); pack(*t8, *t8n, *tPN);
  if (tPN->points.size() < 100) { std::printf("NO target cloud in Final Alignment\n"); return 3; }
  typedef pcl::registration::CorrespondenceEstimationNormalShooting<pcl::PointXYZRGBNormal, pcl::PointXYZRGBNormal, pcl::PointXYZRGBNormal> NS;
  NS::Ptr corrEstNormShoot(new NS);
  corrEstNormShoot->setInputSource(sPN); corrEstNormShoot->setSourceNormals(sPN); corrEstNormShoot->setInputTarget(tPN);
  corrEstNormShoot->setKSearch(20);
  pcl::registration::CorrespondenceRejectorSurfaceNormal::Ptr corrRejSurNorm(new pcl::registration::CorrespondenceRejectorSurfaceNormal);
  corrRejSurNorm->initializeDataContainer<pcl::PointXYZRGBNormal, pcl::PointXYZRGBNormal>();
  corrRejSurNorm->setThreshold(0.7);
  pcl::registration::TransformationEstimationSVD<pcl::PointXYZRGBNormal, pcl::PointXYZRGBNormal>::Ptr transfEstSvd(
      new pcl::registration::TransformationEstimationSVD<pcl::PointXYZRGBNormal, pcl::PointXYZRGBNormal>);
  pcl::IterativeClosestPointWithNormals<pcl::PointXYZRGBNormal, pcl::PointXYZRGBNormal> icp;
  icp.setInputSource(sPN);
  icp.setInputTarget(tPN);
  icp.setMaximumIterations(100);
  icp.setTransformationEpsilon(1e-8);
  icp.setEuclideanFitnessEpsilon(1e-8);
  icp.setCorrespondenceEstimation(corrEstNormShoot);
  icp.addCorrespondenceRejector(corrRejSurNorm);
  icp.setTransformationEstimation(transfEstSvd);
  pcl::PointCloud<pcl::PointXYZRGBNormal> cloudAligned;
  icp.align(cloudAligned);
  const double fitnessScoreFine = icp.getFitnessScore();
  const pcl::Matrix4f fine = icp.getFinalTransformation();
  const double alignedStrength = icp.getAlignStrength();
  std::printf("Aligned Strength : %g  fitness %g  iterations %d converged %d\n", alignedStrength, fitnessScoreFine,
              icp.getNumberOfIterations(), (int)icp.hasConverged());

  // ---- estimateFinalPose: pose of the aligned model relative to the original, by identity correspondences
  pcl::PointCloud<PointT>::Ptr alignedFine(new pcl::PointCloud<PointT>);
  pcl::transformPointCloud(*aligned, *alignedFine, fine);
  pcl::Correspondences corres(model->size());
  for (size_t i = 0; i < model->size(); ++i) corres[i].index_query = corres[i].index_match = (int)i;
  pcl::registration::TransformationEstimationSVD<PointT, PointT> svd;
  pcl::Matrix4f finalPose;
  svd.estimateRigidTransformation(*model, *alignedFine, corres, finalPose);

  double err = 0;
  for (int i = 0; i < 16; ++i) err += (finalPose.m[i] - gt.m[i]) * (finalPose.m[i] - gt.m[i]);
  err = std::sqrt(err);
  std::printf("|finalPose - groundTruth|_F = %.3e\n", err);
  for (int r = 0; r < 4; ++r) std::printf("  % .5f % .5f % .5f % .5f\n", finalPose(r, 0), finalPose(r, 1), finalPose(r, 2), finalPose(r, 3));
  // the fine stage works on 8 mm keypoints with normal shooting: a few degrees is what the reference pipeline delivers
  return (err < 0.1 && icp.hasConverged() && (fitnessScoreFine < 1e-4 || alignedStrength > 0.4)) ? 0 : 1;
}
