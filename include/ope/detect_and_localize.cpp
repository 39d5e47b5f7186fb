// detect_and_localize.cpp — config C1 from files: what DetectAndLocalize does per frame once a table-top cluster has
// been cut out (rosinterface.cpp:80 loads the model .pcd, :250 calls PoseEstimator::estimateFinalPose(model, cluster)),
// with pcl:: replaced by the façade and the GPU library behind it.
//
//   detect_and_localize <model.pcd> <scene.pcd> [<scene2.pcd> ...] [--seed N] [--self-occluded]
//
// One line per frame on stdout, parsed by tests/test_gpu_detect_and_localize.py:
//   frame <k> fitness <f> strength <s> coarse_calls <n> icp_iterations <n> final <16 floats, column-major> coarse <16> fine <16> rigid <16>
// and `aligned <path>` after saving the aligned model of the last frame next to the first scene file.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pcd_io.hpp"
#include "pose_estimator.hpp"

namespace pcl = ope::compat;

static void print16(const char *tag, const pcl::Matrix4f &m) {
  std::printf(" %s", tag);
  for (int i = 0; i < 16; ++i) std::printf(" %.9g", (double)m.m[i]);
}

int main(int argc, char **argv) {
  std::vector<std::string> files;
  uint64_t seed = 1;
  bool self_occluded = false;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--seed") && i + 1 < argc) seed = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "--self-occluded")) self_occluded = true;
    else files.push_back(argv[i]);
  }
  if (files.size() < 2) { std::fprintf(stderr, "usage: %s <model.pcd> <scene.pcd> [more scenes] [--seed N] [--self-occluded]\n", argv[0]); return 2; }
  typedef ope::PoseEstimator::PointT PointT;
  pcl::PointCloud<PointT>::Ptr cloudSourceOriginal(new pcl::PointCloud<PointT>), cloudSource(new pcl::PointCloud<PointT>);
  if (pcl::io::loadPCDFile(files[0], *cloudSourceOriginal) != 0) return 3;   // rosinterface.cpp:80
  *cloudSource = *cloudSourceOriginal;
  ope::PoseEstimator poseEstimator;
  poseEstimator.setSacIaSeed(seed);
  poseEstimator.setUseSelfOccludedRejector(self_occluded);
  for (size_t k = 1; k < files.size(); ++k) {
    pcl::PointCloud<PointT>::Ptr cloudTargetSeg(new pcl::PointCloud<PointT>);
    if (pcl::io::loadPCDFile(files[k], *cloudTargetSeg) != 0) return 3;
    double fitnessScore = 10.0, alignedStrength = 0.0;
    // later frames hand over the source as the previous call left it (rosinterface.cpp:285: cloudSource is not reset)
    const pcl::Matrix4f pose = poseEstimator.estimateFinalPose(cloudSource, cloudTargetSeg, fitnessScore, alignedStrength);
    std::printf("frame %zu fitness %.12g strength %.12g coarse_calls %d icp_iterations %d", k, fitnessScore, alignedStrength,
                poseEstimator.coarseCalls(), poseEstimator.lastIcpIterations());
    print16("final", pose);
    print16("coarse", poseEstimator.lastCoarsePose());
    print16("fine", poseEstimator.lastFinePose());
    print16("rigid", poseEstimator.lastRigidModelPose());
    std::printf("\n");
  }
  const std::string out = files[1] + ".aligned.pcd";
  if (pcl::io::savePCDFile(out, *cloudSource, true) != 0) return 4;
  std::printf("aligned %s\n", out.c_str());
  return 0;
}
