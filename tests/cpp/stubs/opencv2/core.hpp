// Minimal stand-in for <opencv2/core.hpp>: just the members include/kvfe_kimera_shim.hpp touches (cv::Mat data /
// rows / cols / step / type() / empty() / create(), cv::Point2f, cv::KeyPoint), so that the shim can be compiled
// and run in a container without OpenCV.  TEST INFRASTRUCTURE, not part of the product.
#pragma once
#include <cstddef>
#include <memory>
#include <vector>
#define CV_8UC1 0
namespace cv {
struct Point2f {
  float x = 0, y = 0;
  Point2f() = default;
  Point2f(float x_, float y_) : x(x_), y(y_) {}
};
struct KeyPoint {
  Point2f pt;
  float size = 0;
  KeyPoint() = default;
  KeyPoint(Point2f p, float s) : pt(p), size(s) {}
};
struct Mat {
  unsigned char* data = nullptr;
  int rows = 0, cols = 0;
  size_t step = 0;
  std::shared_ptr<std::vector<unsigned char>> buf;
  Mat() = default;
  Mat(int r, int c, int /*type*/, unsigned char* d, size_t s) : data(d), rows(r), cols(c), step(s) {}
  int type() const { return CV_8UC1; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  void create(int r, int c, int /*type*/) {
    buf = std::make_shared<std::vector<unsigned char>>((size_t)r * c);
    data = buf->data();
    rows = r;
    cols = c;
    step = (size_t)c;
  }
};
}  // namespace cv
