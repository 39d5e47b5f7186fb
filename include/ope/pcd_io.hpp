// pcd_io.hpp — pcl::io::loadPCDFile / savePCDFile for the façade's point types (header-only, no PCL).
//
// The on-disk format either side of the path: PCD v0.7, the reference's data is
// `FIELDS x y z rgb`, `SIZE 4 4 4 4`, `TYPE F F F F`, `DATA binary` (DetectAndLocalize/3DModel/*.pcd, 16 bytes per
// point, rgb packed in a float).  Call sites replaced: rosinterface.cpp:80,94,105 (model load),
// BuildModel/src/main.cpp:113-153 (frame load), :221 (savePCDFile of the aligned cloud).
// Return convention is PCL's: 0 on success, -1 on failure (message on stderr).  `DATA ascii` and `DATA binary` are
// read and written; `binary_compressed` is refused (the reference never writes it: savePCDFile(..., true) is plain binary).
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "pcl_compat.hpp"

namespace ope {
namespace compat {
namespace io {

namespace detail {

// where a named PCD field lives inside a façade point (byte offset), or -1
template <class PointT> struct field_map {
  static int offset(const std::string &) { return -1; }
  static const char *const *names(int &n) { n = 0; return nullptr; }
};
template <> struct field_map<PointXYZ> {
  static int offset(const std::string &f) { return f == "x" ? 0 : f == "y" ? 4 : f == "z" ? 8 : -1; }
  static const char *const *names(int &n) { static const char *const k[] = {"x", "y", "z"}; n = 3; return k; }
};
template <> struct field_map<PointXYZRGB> {
  static int offset(const std::string &f) { return f == "x" ? 0 : f == "y" ? 4 : f == "z" ? 8 : (f == "rgb" || f == "rgba") ? 16 : -1; }
  static const char *const *names(int &n) { static const char *const k[] = {"x", "y", "z", "rgb"}; n = 4; return k; }
};
template <> struct field_map<PointXYZRGBNormal> {
  static int offset(const std::string &f) {
    return f == "x" ? 0 : f == "y" ? 4 : f == "z" ? 8 : f == "normal_x" ? 16 : f == "normal_y" ? 20 : f == "normal_z" ? 24
         : (f == "rgb" || f == "rgba") ? 32 : f == "curvature" ? 36 : -1;
  }
  static const char *const *names(int &n) {
    static const char *const k[] = {"x", "y", "z", "rgb", "normal_x", "normal_y", "normal_z", "curvature"};
    n = 8; return k;
  }
};

struct Field { std::string name; int size = 4; char type = 'F'; int count = 1; int file_off = 0; int point_off = -1; };

inline double read_scalar(const unsigned char *p, char type, int size) {
  switch (type) {
    case 'F': { if (size == 4) { float v; std::memcpy(&v, p, 4); return v; } double v; std::memcpy(&v, p, 8); return v; }
    case 'U': { if (size == 1) return *p; if (size == 2) { uint16_t v; std::memcpy(&v, p, 2); return v; } uint32_t v; std::memcpy(&v, p, 4); return v; }
    default:  { if (size == 1) return (int8_t)*p; if (size == 2) { int16_t v; std::memcpy(&v, p, 2); return v; } int32_t v; std::memcpy(&v, p, 4); return v; }
  }
}

}  // namespace detail

template <class PointT> int loadPCDFile(const std::string &file_name, PointCloud<PointT> &cloud) {
  using detail::Field;
  cloud.clear();
  std::ifstream f(file_name, std::ios::binary);
  if (!f) { std::fprintf(stderr, "[ope::io::loadPCDFile] Could not find file '%s'.\n", file_name.c_str()); return -1; }
  std::vector<Field> fields;
  size_t n_points = 0, width = 0, height = 1;
  bool have_points = false, short_line = false;
  std::string data_kind, line;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ss(line);
    std::string key;
    ss >> key;
    if (key == "FIELDS" || key == "COLUMNS") {
      std::string name;
      while (ss >> name) { Field fl; fl.name = name; fields.push_back(fl); }
    } else if (key == "SIZE") {
      for (auto &fl : fields) ss >> fl.size;
      short_line = short_line || !ss;
    } else if (key == "TYPE") {
      for (auto &fl : fields) ss >> fl.type;
      short_line = short_line || !ss;
    } else if (key == "COUNT") {
      for (auto &fl : fields) ss >> fl.count;
      short_line = short_line || !ss;
    } else if (key == "WIDTH") {
      ss >> width;
    } else if (key == "HEIGHT") {
      ss >> height;
    } else if (key == "POINTS") {
      ss >> n_points; have_points = true;
    } else if (key == "DATA") {
      ss >> data_kind;
      break;
    }
  }
  if (fields.empty() || data_kind.empty()) { std::fprintf(stderr, "[ope::io::loadPCDFile] '%s': no FIELDS / DATA line.\n", file_name.c_str()); return -1; }
  if (!have_points) n_points = width * height;
  // header values are untrusted input: every field must have a size of 1, 2, 4 or 8 and a count of at least 1 (a short or
  // garbled SIZE / TYPE / COUNT line leaves a stream error behind or a zero / negative value), and the point count must be
  // one the file can hold
  long long rec = 0;
  bool any = false, bad = short_line;
  for (auto &fl : fields) {
    if (!(fl.size == 1 || fl.size == 2 || fl.size == 4 || fl.size == 8) || fl.count < 1 || fl.count > 65536 ||
        !(fl.type == 'F' || fl.type == 'U' || fl.type == 'I')) { bad = true; break; }
    fl.file_off = (int)rec;
    rec += (long long)fl.size * fl.count;
    if (rec > (1 << 24)) { bad = true; break; }
    fl.point_off = (fl.count == 1 && fl.size == 4) ? detail::field_map<PointT>::offset(fl.name) : -1;
    any = any || fl.point_off >= 0;
  }
  if (bad || rec <= 0) { std::fprintf(stderr, "[ope::io::loadPCDFile] '%s': bad SIZE / TYPE / COUNT in the header.\n", file_name.c_str()); return -1; }
  if (!any) { std::fprintf(stderr, "[ope::io::loadPCDFile] '%s': none of its fields exists in the point type.\n", file_name.c_str()); return -1; }
  {
    // what is left of the file bounds the point count (binary: rec bytes per point; ascii: at least two bytes per value)
    const std::streampos here = f.tellg();
    f.seekg(0, std::ios::end);
    const std::streampos end = f.tellg();
    f.seekg(here);
    const unsigned long long left = (here >= 0 && end >= here) ? (unsigned long long)(end - here) : 0ull;
    const unsigned long long need = data_kind == "binary" ? (unsigned long long)rec : 2ull * fields.size();
    if (n_points > left / need + 1ull || (data_kind == "binary" && (unsigned long long)n_points * (unsigned long long)rec > left)) {
      std::fprintf(stderr, "[ope::io::loadPCDFile] '%s': file shorter than its header says.\n", file_name.c_str());
      return -1;
    }
  }
  try {
    cloud.points.assign(n_points, PointT());
  } catch (const std::bad_alloc &) {
    std::fprintf(stderr, "[ope::io::loadPCDFile] '%s': out of memory for %zu points.\n", file_name.c_str(), n_points);
    cloud.clear();
    return -1;
  }
  if (data_kind == "binary") {
    std::vector<unsigned char> buf((size_t)rec * n_points);
    f.read(reinterpret_cast<char *>(buf.data()), (std::streamsize)buf.size());
    if ((size_t)f.gcount() != buf.size()) { std::fprintf(stderr, "[ope::io::loadPCDFile] '%s': file shorter than its header says.\n", file_name.c_str()); cloud.clear(); return -1; }
    for (size_t i = 0; i < n_points; ++i) {
      const unsigned char *r = buf.data() + i * (size_t)rec;
      unsigned char *p = reinterpret_cast<unsigned char *>(&cloud.points[i]);
      for (const auto &fl : fields)
        if (fl.point_off >= 0) std::memcpy(p + fl.point_off, r + fl.file_off, 4);   // 4-byte fields are copied bit for bit (rgb is a packed float)
    }
  } else if (data_kind == "ascii") {
    for (size_t i = 0; i < n_points; ++i) {
      unsigned char *p = reinterpret_cast<unsigned char *>(&cloud.points[i]);
      for (const auto &fl : fields)
        for (int c = 0; c < fl.count; ++c) {
          std::string tok;
          if (!(f >> tok)) { std::fprintf(stderr, "[ope::io::loadPCDFile] '%s': fewer values than POINTS.\n", file_name.c_str()); cloud.clear(); return -1; }
          if (fl.point_off < 0) continue;
          if (fl.type == 'F') { const float v = tok == "nan" ? std::nanf("") : std::strtof(tok.c_str(), nullptr); std::memcpy(p + fl.point_off, &v, 4); }
          else if (fl.type == 'U') { const uint32_t v = (uint32_t)std::strtoul(tok.c_str(), nullptr, 10); std::memcpy(p + fl.point_off, &v, 4); }
          else { const int32_t v = (int32_t)std::strtol(tok.c_str(), nullptr, 10); std::memcpy(p + fl.point_off, &v, 4); }
        }
    }
  } else {
    std::fprintf(stderr, "[ope::io::loadPCDFile] '%s': DATA %s is not supported (ascii and binary are).\n", file_name.c_str(), data_kind.c_str());
    cloud.clear();
    return -1;
  }
  cloud.width = (uint32_t)(width ? width : n_points);
  cloud.height = (uint32_t)(height ? height : 1);
  if ((size_t)cloud.width * cloud.height != n_points) { cloud.width = (uint32_t)n_points; cloud.height = 1; }
  cloud.is_dense = true;
  for (const auto &pt : cloud.points)
    if (!std::isfinite(pt.x) || !std::isfinite(pt.y) || !std::isfinite(pt.z)) { cloud.is_dense = false; break; }
  return 0;
}

template <class PointT> int savePCDFile(const std::string &file_name, const PointCloud<PointT> &cloud, bool binary_mode = false) {
  int nf = 0;
  const char *const *names = detail::field_map<PointT>::names(nf);
  if (nf == 0) { std::fprintf(stderr, "[ope::io::savePCDFile] point type has no PCD field map.\n"); return -1; }
  std::ofstream f(file_name, std::ios::binary);
  if (!f) { std::fprintf(stderr, "[ope::io::savePCDFile] Could not open '%s' for writing.\n", file_name.c_str()); return -1; }
  const size_t n = cloud.points.size();
  std::ostringstream h;
  h << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS";
  for (int i = 0; i < nf; ++i) h << ' ' << names[i];
  h << "\nSIZE";
  for (int i = 0; i < nf; ++i) h << " 4";
  h << "\nTYPE";
  for (int i = 0; i < nf; ++i) h << " F";
  h << "\nCOUNT";
  for (int i = 0; i < nf; ++i) h << " 1";
  h << "\nWIDTH " << n << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n << "\nDATA " << (binary_mode ? "binary" : "ascii") << "\n";
  f << h.str();
  std::vector<int> off((size_t)nf);
  for (int i = 0; i < nf; ++i) off[(size_t)i] = detail::field_map<PointT>::offset(names[i]);
  if (binary_mode) {
    std::vector<unsigned char> buf((size_t)nf * 4 * n);
    for (size_t i = 0; i < n; ++i) {
      const unsigned char *p = reinterpret_cast<const unsigned char *>(&cloud.points[i]);
      for (int k = 0; k < nf; ++k) std::memcpy(buf.data() + (i * (size_t)nf + (size_t)k) * 4, p + off[(size_t)k], 4);
    }
    f.write(reinterpret_cast<const char *>(buf.data()), (std::streamsize)buf.size());
  } else {
    char tmp[64];
    for (size_t i = 0; i < n; ++i) {
      const unsigned char *p = reinterpret_cast<const unsigned char *>(&cloud.points[i]);
      for (int k = 0; k < nf; ++k) {
        float v;
        std::memcpy(&v, p + off[(size_t)k], 4);
        if (std::strcmp(names[k], "rgb") == 0) std::snprintf(tmp, sizeof tmp, "%.9g", (double)v);   // packed colours need all 9 digits
        else std::snprintf(tmp, sizeof tmp, "%.9g", (double)v);
        f << tmp << (k + 1 < nf ? ' ' : '\n');
      }
    }
  }
  return f.good() ? 0 : -1;
}

}  // namespace io
}  // namespace compat
}  // namespace ope
