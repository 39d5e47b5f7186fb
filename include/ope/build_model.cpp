// build_model.cpp — config C5 from files, BuildModel's own sequence (BuildModel/src/main.cpp:113-153 load the frames,
// :207 registerPointClouds, :221 savePCDFile of the aligned cloud) with pcl:: replaced by the façade:
//
//   build_model <out.pcd> <corrRejThresh> <maxIter> <frame0.pcd> <frame1.pcd> [...]
//
// Prints one `pair <k> iterations <n> converged <0|1> fitness <f> T <16 floats, column-major>` line per registration.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "pcd_io.hpp"
#include "reg_mesh_pcd.hpp"

namespace pcl = ope::compat;

int main(int argc, char **argv) {
  if (argc < 6) { std::fprintf(stderr, "usage: %s <out.pcd> <corrRejThresh> <maxIter> <frame0.pcd> <frame1.pcd> [...]\n", argv[0]); return 2; }
  const std::string out_path = argv[1];
  const float corrRejThresh = (float)std::atof(argv[2]);
  const int maxIter = std::atoi(argv[3]);
  typedef ope::RegMeshPcd::PointTReg PointTReg;
  std::vector<pcl::PointCloud<PointTReg>::Ptr> cloudVector;
  for (int i = 4; i < argc; ++i) {
    pcl::PointCloud<PointTReg>::Ptr c(new pcl::PointCloud<PointTReg>);
    if (pcl::io::loadPCDFile(argv[i], *c) != 0) return 3;
    cloudVector.push_back(c);
  }
  ope::RegMeshPcd regMeshPcd;
  pcl::PointCloud<PointTReg>::Ptr cloudAligned = regMeshPcd.registerPointClouds(cloudVector, 0.005f, corrRejThresh, maxIter);   // main.cpp:207
  for (size_t k = 0; k < regMeshPcd.pairs().size(); ++k) {
    const auto &p = regMeshPcd.pairs()[k];
    std::printf("pair %zu iterations %d converged %d fitness %.12g T", k, p.iterations, (int)p.converged, p.fitness);
    for (int i = 0; i < 16; ++i) std::printf(" %.9g", (double)p.T.m[i]);
    std::printf("\n");
  }
  if (pcl::io::savePCDFile(out_path, *cloudAligned, true) != 0) return 4;   // main.cpp:221
  std::printf("Saved %zu data points to %s.\n", cloudAligned->size(), out_path.c_str());
  return 0;
}
