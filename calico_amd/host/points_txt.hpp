// points_txt.hpp — observation ingest from the reference's per-camera corner dump
// (SURVEY.md 8(f)2).  CameraCali writes <camera dir>/points.txt after detection
// (/root/reference/src/camera-calibration.cpp:283-294):
//     image <i>
//     <j> <x> <y> <j> <x> <y> ...        every corner j with points_present[i][j], 9 significant digits
//     end <i>
// for i = 0 .. number_images-1, followed by an "Image-board truth table" that is not needed
// here.  j indexes the corners of all patterns: pattern = j / corners_per_pattern, the
// same index that selects P_class->three_d_points[j] (solving-structure.cpp:523-559).
// Host-only, header-only; no dependency on the reference's types.
#pragma once
#include <cstdint>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace calico_amd {

struct PointsTxtImage {
  std::vector<int32_t> corner;
  std::vector<double> x, y;
};

// Returns false (with *err filled) on a malformed file.
inline bool ReadPointsTxt(const std::string& path, std::vector<PointsTxtImage>& images, std::string* err = nullptr) {
  std::ifstream in(path);
  auto fail = [&](const std::string& m) {
    if (err) *err = path + ": " + m;
    return false;
  };
  if (!in) return fail("cannot open");
  images.clear();
  std::string line;
  while (std::getline(in, line)) {
    std::istringstream head(line);
    std::string word;
    int idx = -1;
    if (!(head >> word)) continue;  // blank line
    if (word != "image") break;     // "Image-board truth table" and what follows
    if (!(head >> idx) || idx != (int)images.size()) return fail("image index out of order");
    PointsTxtImage im;
    if (!std::getline(in, line)) return fail("truncated after 'image'");
    std::istringstream body(line);
    int j;
    double x, y;
    while (body >> j) {
      if (!(body >> x >> y)) return fail("corner without coordinates in image " + std::to_string(idx));
      if (j < 0) return fail("negative corner index");
      im.corner.push_back(j);
      // The values were float32 detections (cv::Point2f, camera-calibration.cpp:257-258) printed
      // with 9 significant digits, which identify a float32 uniquely: snapping back gives the
      // doubles two_d_point_coordinates_dense held when the file was written.
      im.x.push_back((double)(float)x);
      im.y.push_back((double)(float)y);
    }
    if (!std::getline(in, line)) return fail("missing 'end'");
    std::istringstream tail(line);
    int idx2 = -1;
    if (!(tail >> word >> idx2) || word != "end" || idx2 != idx) return fail("bad 'end' line of image " + std::to_string(idx));
    images.push_back(std::move(im));
  }
  return true;
}

// Appends every corner of one camera's images to the SoA arrays calico_rp_add_observations
// takes.  Image i of the camera is time index start_time_this_camera + i
// (solving-structure.cpp:521); pattern_a_vars / time_f_vars map pattern and time indices
// to variable indices (multicamera.cpp:261-278).  `keep(i, j)` selects corners (e.g.
// points_used_min); pass nullptr to take all.
template <typename Keep>
inline void AppendCameraObservations(const std::vector<PointsTxtImage>& images, int camera_index,
                                     int start_time_this_camera, int corners_per_pattern,
                                     const std::vector<int>& pattern_a_vars, const std::vector<int>& time_f_vars,
                                     Keep keep, std::vector<int32_t>& cam, std::vector<int32_t>& pat_var,
                                     std::vector<int32_t>& time_var, std::vector<int32_t>& corner,
                                     std::vector<double>& u, std::vector<double>& v) {
  for (size_t i = 0; i < images.size(); ++i) {
    const PointsTxtImage& im = images[i];
    const int t = start_time_this_camera + (int)i;
    if (t < 0 || t >= (int)time_f_vars.size()) continue;
    for (size_t k = 0; k < im.corner.size(); ++k) {
      const int j = im.corner[k], p = j / corners_per_pattern;
      if (p >= (int)pattern_a_vars.size() || !keep((int)i, j)) continue;
      cam.push_back(camera_index);
      pat_var.push_back(pattern_a_vars[p]);
      time_var.push_back(time_f_vars[t]);
      corner.push_back(j);
      u.push_back(im.x[k]);
      v.push_back(im.y[k]);
    }
  }
}

}  // namespace calico_amd
