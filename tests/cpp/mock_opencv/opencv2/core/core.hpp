// MOCK of the handful of OpenCV core types include/orbx_shim.hpp touches under -DORBX_WITH_OPENCV.
// TEST INFRASTRUCTURE, COMPILE CHECK ONLY: OpenCV is not in this image, so nothing else could ever
// compile that branch of the shim.  This file pins NOTHING about OpenCV's arithmetic -- it only gives the reference's
// signatures (cv::InputArray, cv::OutputArray, std::vector<cv::KeyPoint>&, cv::Mat) something to resolve against, with the
// member names and layouts the shim relies on (cv::KeyPoint = 28 bytes: pt, size, angle, response, octave, class_id;
// cv::Mat::data / cols / rows / step / type() / create() / release() / empty()).
#pragma once
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_Assert(expr) do { if (!(expr)) throw std::runtime_error("CV_Assert failed: " #expr); } while (0)

typedef unsigned char uchar;

namespace cv {

struct Point2f {
  float x = 0, y = 0;
};

class KeyPoint {
 public:
  Point2f pt;
  float size = 0, angle = -1, response = 0;
  int octave = 0, class_id = -1;
};

class Mat {
 public:
  Mat() {}
  Mat(int r, int c, int t) { create(r, c, t); }
  Mat(int r, int c, int t, void* ext, size_t st = 0) : data((uchar*)ext), cols(c), rows(r), step(st ? st : (size_t)c * chan(t)), type_(t) {}
  void create(int r, int c, int t) {
    if (r == rows && c == cols && t == type_ && data) return;
    own_ = std::make_shared<std::vector<uchar>>((size_t)r * c * chan(t));
    data = own_->data(); rows = r; cols = c; type_ = t; step = (size_t)c * chan(t);
  }
  void release() { own_.reset(); data = nullptr; rows = cols = 0; step = 0; }
  bool empty() const { return !data || rows == 0 || cols == 0; }
  int type() const { return type_; }
  int channels() const { return chan(type_); }
  uchar* data = nullptr;
  int cols = 0, rows = 0;
  size_t step = 0;

 private:
  static int chan(int t) { return (t >> 3) + 1; }
  std::shared_ptr<std::vector<uchar>> own_;
  int type_ = 0;
};

class _InputArray {
 public:
  _InputArray(const Mat& m) : m_(&m) {}
  bool empty() const { return m_->empty(); }
  Mat getMat() const { return *m_; }

 protected:
  const Mat* m_;
};
class _OutputArray : public _InputArray {
 public:
  _OutputArray(Mat& m) : _InputArray(m), w_(&m) {}
  void create(int r, int c, int t) const { w_->create(r, c, t); }
  void release() const { w_->release(); }
  Mat getMat() const { return *w_; }

 private:
  Mat* w_;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

}  // namespace cv

// ---- additions for tools/pin_opencv/pin_opencv.cpp (compile check only; declarations, no definitions) ----
#define CV_32F 5
#define CV_64F 6
#define CV_32FC2 13
namespace cv {
struct Size {
  int width = 0, height = 0;
  Size() {}
  Size(int w, int h) : width(w), height(h) {}
};
enum { INTER_LINEAR = 1, BORDER_REFLECT_101 = 4, COLOR_BGR2GRAY = 6, COLOR_RGB2GRAY = 7 };
float fastAtan2(float y, float x);
const char* getVersionString_();
}  // namespace cv
int cvRound(double v);
int cvRound(float v);
#define CV_VERSION "mock"
