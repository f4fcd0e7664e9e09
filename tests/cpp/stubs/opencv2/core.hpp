// Minimal stand-in for <opencv2/core.hpp>: just the members include/kvfe_kimera_shim.hpp touches (cv::Mat data /
// rows / cols / step / type() / empty() / create(), cv::Point2f, cv::KeyPoint), so that the shim can be compiled
// and run in a container without OpenCV.  TEST INFRASTRUCTURE, not part of the product.
#pragma once
#include <cstddef>
#include <memory>
#include <vector>
#define CV_8UC1 0
#define CV_16SC1 3
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
  int type_ = CV_8UC1;
  Mat(int r, int c, int t, unsigned char* d, size_t s) : data(d), rows(r), cols(c), step(s), type_(t) {}
  int type() const { return type_; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  void create(int r, int c, int t) {
    const size_t es = t == CV_16SC1 ? 2 : 1;
    buf = std::make_shared<std::vector<unsigned char>>((size_t)r * c * es);
    data = buf->data();
    rows = r;
    cols = c;
    step = (size_t)c * es;
    type_ = t;
  }
};
}  // namespace cv
