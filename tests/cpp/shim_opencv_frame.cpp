// shim_opencv_frame.cpp — the -DORBX_WITH_OPENCV branch of include/orbx_shim.hpp driven through a Frame class shaped like
// the reference's (SlamTypes/Frame.hpp:20-111: public mpORBextractor, mvKeys, mvKeysUn, mDescriptors, N, the static image
// bounds), with the reference's own call sites kept verbatim:
//   Frame.cpp:58-60              (*mpORBextractor)(im, cv::Mat(), mvKeys, mDescriptors, unused);
//   demo_initialization.cpp:105  ORBmatcher orbMatcher(0.9, true);
//   demo_initialization.cpp:108  int nmatches = orbMatcher.SearchForInitialization(frame1, frame2, mvMatches, 100);
// Built against tests/cpp/mock_opencv (a compile-check mock of six cv:: types; it pins nothing about OpenCV).
// usage: shim_opencv_frame W H frameA.raw frameB.raw nfeatures iniTh minTh [W2 H2 big.raw]
//   prints the same RESULT line as shim_demo; with the optional second size a third frame goes through the SAME extractor
//   object first and last (the context grows, cpp:1531-1545) and "GROW n hash hash" is printed for it.
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "orbx_shim.hpp"

namespace ORB_SLAM_Tracking {

class ORBVocabulary;

class Frame {  // the slice of SlamTypes/Frame.{hpp,cpp} on this path
 public:
  Frame(cv::Mat& im, const double& timestamp, ORBextractor* extractor, ORBVocabulary* voc, cv::Mat& K, cv::Mat& distCoef)
      : mpORBvocabulary(voc), mpORBextractor(extractor), mTimestamp(timestamp) {
    (void)K; (void)distCoef;
    if (mbInitialComputations) {  // Frame.cpp:44-55 with k1 == 0: the image itself (Frame.cpp:127-131)
      mnMinX = 0; mnMaxX = im.cols; mnMinY = 0; mnMaxY = im.rows;
      mbInitialComputations = false;
    }
    std::vector<int> unused = {0, 0};
    (*mpORBextractor)(im, cv::Mat(), mvKeys, mDescriptors, unused);  // Frame.cpp:58-60, verbatim
    N = (int)mvKeys.size();
    mvKeysUn = mvKeys;  // Frame.cpp:137-141 (no distortion)
  }
  ORBVocabulary* mpORBvocabulary;
  ORBextractor* mpORBextractor;
  double mTimestamp;
  int N = 0;
  std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
  cv::Mat mDescriptors;
  static int mnMinX, mnMaxX, mnMinY, mnMaxY;
  static bool mbInitialComputations;
};
int Frame::mnMinX = 0, Frame::mnMaxX = 0, Frame::mnMinY = 0, Frame::mnMaxY = 0;
bool Frame::mbInitialComputations = true;

}  // namespace ORB_SLAM_Tracking

static std::vector<uint8_t> readRaw(const char* path, size_t n) {
  std::vector<uint8_t> v(n);
  std::ifstream f(path, std::ios::binary);
  f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)n);
  if ((size_t)f.gcount() != n) { std::fprintf(stderr, "short read %s\n", path); std::exit(2); }
  return v;
}
static unsigned long long fnv(const void* p, size_t n) {
  unsigned long long h = 1469598103934665603ull;
  for (size_t i = 0; i < n; i++) { h ^= ((const uint8_t*)p)[i]; h *= 1099511628211ull; }
  return h;
}

int main(int argc, char** argv) {
  if (argc < 8) return 2;
  using namespace ORB_SLAM_Tracking;
  const int W = std::atoi(argv[1]), H = std::atoi(argv[2]);
  std::vector<uint8_t> a = readRaw(argv[3], (size_t)W * H), b = readRaw(argv[4], (size_t)W * H);
  orbx::verbose() = false;  // the quiet flag: none of the reference's stdout lines may appear below
  try {
    ORBextractor orbExtractor(std::atoi(argv[5]), 1.2, 8, std::atoi(argv[6]), std::atoi(argv[7]));  // demo_initialization.cpp:72
    cv::Mat K, distCoef;
    std::vector<uint8_t> big;
    int W2 = 0, H2 = 0;
    auto grow = [&]() {
      cv::Mat imBig(H2, W2, CV_8UC1, big.data());
      std::vector<cv::KeyPoint> kb;
      cv::Mat db;
      std::vector<int> unused = {0, 0};
      const int r = orbExtractor(imBig, cv::Mat(), kb, db, unused);
      std::printf("GROW %d %llu %llu %d\n", r, fnv(kb.data(), kb.size() * sizeof(cv::KeyPoint)),
                  fnv(db.data, (size_t)db.rows * 32), orbExtractor.mvImagePyramid[1].cols);
    };
    if (argc >= 11) {
      W2 = std::atoi(argv[8]); H2 = std::atoi(argv[9]);
      big = readRaw(argv[10], (size_t)W2 * H2);
      grow();
    }
    cv::Mat im1Gray(H, W, CV_8UC1, a.data()), im2Gray(H, W, CV_8UC1, b.data());
    Frame frame1(im1Gray, 0.0, &orbExtractor, nullptr, K, distCoef);  // demo_initialization.cpp:76
    Frame frame2(im2Gray, 1.0, &orbExtractor, nullptr, K, distCoef);  // :77
    ORBmatcher orbMatcher(0.9, true);                                 // :105, verbatim
    std::vector<int> mvMatches;
    int nmatches = orbMatcher.SearchForInitialization(frame1, frame2, mvMatches, 100);  // :108, verbatim
    std::printf("RESULT %d %d %d %llu %llu %llu\n", frame1.N, frame2.N, nmatches,
                fnv(frame1.mvKeys.data(), frame1.mvKeys.size() * sizeof(cv::KeyPoint)),
                fnv(frame1.mDescriptors.data, (size_t)frame1.mDescriptors.rows * 32), fnv(mvMatches.data(), mvMatches.size() * sizeof(int)));
    if (argc >= 11) grow();
  } catch (const orbx::Error& e) {
    std::fprintf(stderr, "orbx error %d: %s\n", e.code, e.what());
    return 4;
  }
  return 0;
}
