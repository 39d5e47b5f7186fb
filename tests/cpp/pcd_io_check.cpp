// Host-only check of ope::io::loadPCDFile / savePCDFile on well-formed and malformed files (no GPU call is made; the
// program links libope_hip.so only because the facade header declares the C ABI).  argv[1]: a scratch directory.
#include <cstdio>
#include <fstream>
#include <string>

#include "ope/pcd_io.hpp"

using namespace ope::compat;

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #cond); ++failures; } } while (0)

static void write_file(const std::string &path, const std::string &header, const std::string &payload) {
  std::ofstream f(path, std::ios::binary);
  f << header;
  f.write(payload.data(), (std::streamsize)payload.size());
}

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : ".";
  // round trip, binary, with colours
  PointCloud<PointXYZRGB> c;
  for (int i = 0; i < 5; ++i) { PointXYZRGB p; p.x = 0.1f * i; p.y = -0.2f * i; p.z = 1.f + i; uint32_t rgb = 0x00102030u + i; std::memcpy(&p.rgb, &rgb, 4); c.points.push_back(p); }
  c.width = 5; c.height = 1;
  CHECK(io::savePCDFile(dir + "/ok.pcd", c, true) == 0);
  PointCloud<PointXYZRGB> d;
  CHECK(io::loadPCDFile(dir + "/ok.pcd", d) == 0);
  CHECK(d.points.size() == 5);
  for (int i = 0; i < 5 && d.points.size() == 5; ++i) CHECK(std::memcmp(&d.points[i].rgb, &c.points[i].rgb, 4) == 0 && d.points[i].z == c.points[i].z);

  const std::string body(64, '\0');
  const std::string head = "# .PCD v0.7\nVERSION 0.7\nFIELDS x y z rgb\n";
  // negative size: file_off would go negative
  write_file(dir + "/neg_size.pcd", head + "SIZE -4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA binary\n", body);
  CHECK(io::loadPCDFile(dir + "/neg_size.pcd", d) == -1);
  // zero count
  write_file(dir + "/zero_count.pcd", head + "SIZE 4 4 4 4\nTYPE F F F F\nCOUNT 0 1 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA binary\n", body);
  CHECK(io::loadPCDFile(dir + "/zero_count.pcd", d) == -1);
  // short SIZE line
  write_file(dir + "/short_size.pcd", head + "SIZE 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA binary\n", body);
  CHECK(io::loadPCDFile(dir + "/short_size.pcd", d) == -1);
  // absurd point count: must return -1, not throw bad_alloc or read past the file
  write_file(dir + "/huge.pcd", head + "SIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 4000000000000\nHEIGHT 1\nPOINTS 4000000000000\nDATA binary\n", body);
  CHECK(io::loadPCDFile(dir + "/huge.pcd", d) == -1);
  // header says 8 points, payload holds 4
  write_file(dir + "/short_body.pcd", head + "SIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 8\nHEIGHT 1\nPOINTS 8\nDATA binary\n", body);
  CHECK(io::loadPCDFile(dir + "/short_body.pcd", d) == -1);
  // exactly 4 points: fine
  write_file(dir + "/four.pcd", head + "SIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 4\nHEIGHT 1\nPOINTS 4\nDATA binary\n", body);
  CHECK(io::loadPCDFile(dir + "/four.pcd", d) == 0 && d.points.size() == 4);
  // ascii
  write_file(dir + "/ascii.pcd", "FIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 2\nHEIGHT 1\nPOINTS 2\nDATA ascii\n", "1 2 3\n4 5 nan\n");
  PointCloud<PointXYZ> a;
  CHECK(io::loadPCDFile(dir + "/ascii.pcd", a) == 0 && a.points.size() == 2 && a.points[1].y == 5.f && !a.is_dense);
  std::printf(failures ? "pcd_io_check: %d failures\n" : "pcd_io_check: ok\n", failures);
  return failures ? 1 : 0;
}
