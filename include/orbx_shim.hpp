// orbx_shim.hpp — C++ drop-in classes over the C ABI of orbx.h.
//
// Re-creates, with identical names, argument order and error behaviour, the classes of the reference that sit on
// the tracking hot path, so that its Frame / Initializer / demos keep compiling against this header instead of
// Features/ORBextractor.hpp and Features/ORBmatcher.hpp:
//
//   ORB_SLAM_Tracking::ORBextractor   Features/ORBextractor.hpp:55-158   (ctor, operator(), 8 getters, mvImagePyramid)
//   ORB_SLAM_Tracking::ORBmatcher     Features/ORBmatcher.hpp:13-61      (ctor, SearchForInitialization, 3 constants)
//
// The reference's signatures use cv:: types.  OpenCV is not part of this repository, so the shim is written against a
// minimal traits layer:
//   * without OpenCV (default): orbx::KeyPoint (28-byte POD, same layout as cv::KeyPoint), orbx::Image8 (pointer view),
//     std::vector<uint8_t> descriptor rows;
//   * with -DORBX_WITH_OPENCV (and OpenCV headers on the include path): the real cv::InputArray / cv::KeyPoint /
//     cv::OutputArray / cv::Mat signatures of the reference, bit-for-bit the same call as Frame.cpp:58-60.
// Header-only; link with -lorbx (orb_slam_tracking_amd/liborbx.so).  See INTEGRATION.md.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "orbx.h"

#ifdef ORBX_WITH_OPENCV
#include <opencv2/core/core.hpp>
#endif

namespace orbx {

struct KeyPoint {  // cv::KeyPoint layout
  struct { float x, y; } pt;
  float size, angle, response;
  int octave, class_id;
};
static_assert(sizeof(KeyPoint) == sizeof(orbx_keypoint), "KeyPoint must mirror orbx_keypoint");

struct Image8 {  // non-owning view of a CV_8UC1 image
  const uint8_t* data = nullptr;
  int cols = 0, rows = 0, step = 0;
  bool empty() const { return !data || cols <= 0 || rows <= 0; }
};

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// The reference prints from inside the hot path ("Sum of features = N" in the constructor, cpp:549; four statistics lines
// on every SearchForInitialization call, ORBmatcher.cpp:144-147).  The shim reproduces them by default; a host that does
// not want them calls orbx::verbose() = false once (or sets ORBX_QUIET=1 in the environment).
inline bool& verbose() {
  static bool v = std::getenv("ORBX_QUIET") == nullptr;
  return v;
}

// With OpenCV types the shim mirrors the reference's public mvImagePyramid (hpp:111) after every operator() call: eight
// device-to-host copies nobody in the reference reads.  A latency-sensitive host switches that off once.
inline bool& mirrorPyramid() {
  static bool v = true;
  return v;
}

// descriptor rows of a Frame-like object: N x 32 contiguous bytes (std::vector<uint8_t> in the POD build, cv::Mat CV_8U
// created at cpp:1573 in the OpenCV build)
inline const uint8_t* descriptorBytes(const std::vector<uint8_t>& d) { return d.data(); }
#ifdef ORBX_WITH_OPENCV
inline const uint8_t* descriptorBytes(const cv::Mat& d) { return d.data; }
#endif

}  // namespace orbx

namespace ORB_SLAM_Tracking {

#ifdef ORBX_WITH_OPENCV
typedef cv::KeyPoint KeyPointT;
#else
typedef orbx::KeyPoint KeyPointT;
#endif

class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

  // Features/ORBextractor.hpp:68-69.  The extra, defaulted arguments are only the device context's initial reservation:
  // operator() takes any image (cpp:1531-1545) and the context grows when a larger one arrives.
  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int maxWidth = 752,
               int maxHeight = 480, int device = 0)
      : nfeatures_(nfeatures), nlevels_(nlevels) {
    orbx_params p{nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST};
    const int r = orbx_create(&p, device, maxWidth, maxHeight, 1, nullptr, &ctx_);
    if (r != ORBX_OK) {
      // the reference prints and calls exit(1) for scaleFactor == 1 with nlevels > 1 (cpp:502-505); a library must not
      std::cerr << "ORBextractor: orbx_create failed (" << r << ")" << std::endl;
      throw orbx::Error(r, "orbx_create failed");
    }
    mvScaleFactor.resize(nlevels);
    mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    orbx_get_tables(ctx_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(),
                    mnFeaturesPerLevel.data());
    capacity_ = 0;
    for (int q : mnFeaturesPerLevel) capacity_ += q;
    if (orbx::verbose()) std::cout << "Sum of features = " << capacity_ << std::endl;  // cpp:549
    mvImagePyramid.resize(nlevels);
  }
  ~ORBextractor() { orbx_destroy(ctx_); }
  ORBextractor(const ORBextractor&) = delete;
  ORBextractor& operator=(const ORBextractor&) = delete;

#ifdef ORBX_WITH_OPENCV
  // Features/ORBextractor.hpp:83-85
  int operator()(cv::InputArray _image, cv::InputArray _mask, std::vector<cv::KeyPoint>& _keypoints,
                 cv::OutputArray _descriptors, std::vector<int>& vLappingArea) {
    (void)_mask;
    if (_image.empty()) return -1;  // cpp:1536
    cv::Mat image = _image.getMat();
    CV_Assert(image.type() == CV_8UC1);
    std::vector<orbx_keypoint>& k = scratchK_;  // (kept across calls: no allocation per frame)
    std::vector<uint8_t>& d = scratchD_;
    if (k.size() < (size_t)std::max(capacity_, 1)) k.resize(std::max(capacity_, 1));
    if (d.size() < (size_t)std::max(capacity_, 1) * 32) d.resize((size_t)std::max(capacity_, 1) * 32);
    int n = 0;
    const int r = orbx_extract(ctx_, image.data, image.cols, image.rows, (int)image.step, vLappingArea[0], vLappingArea[1],
                               k.data(), d.data(), capacity_, &n);
    if (r < 0) throw orbx::Error(r, orbx_last_error(ctx_));
    _keypoints.resize(n);
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "cv::KeyPoint layout");
    if (n) std::memcpy((void*)_keypoints.data(), k.data(), sizeof(orbx_keypoint) * n);
    if (n == 0) {
      _descriptors.release();  // cpp:1567-1571
    } else {
      _descriptors.create(n, 32, CV_8U);
      std::memcpy(_descriptors.getMat().data, d.data(), (size_t)n * 32);
    }
    if (orbx::mirrorPyramid()) refreshPyramid();
    return r;
  }
  std::vector<cv::Mat> mvImagePyramid;  // hpp:111 (levels without the 19-px ring; use imagePyramid(level, 19) for it)
#else
  // same call with POD types: image view, ignored mask, keypoints, N x 32 descriptor bytes, lapping area
  int operator()(const orbx::Image8& image, const orbx::Image8& /*mask*/, std::vector<orbx::KeyPoint>& keypoints,
                 std::vector<uint8_t>& descriptors, std::vector<int>& vLappingArea) {
    if (image.empty()) return -1;  // cpp:1536
    // extraction into the extractor's own scratch (kept across calls: no allocation and no value-initialisation of capacity-sized
    // vectors per frame), then exactly n entries into the caller's vectors; on an error the caller's vectors are left empty, as the
    // reference leaves them.  An ORBextractor is used by one thread at a time, like the reference's.
    static_assert(sizeof(orbx::KeyPoint) == sizeof(orbx_keypoint), "orbx::KeyPoint layout");
    std::vector<orbx_keypoint>& k = scratchK_;
    std::vector<uint8_t>& d = scratchD_;
    if (k.size() < (size_t)std::max(capacity_, 1)) k.resize(std::max(capacity_, 1));
    if (d.size() < (size_t)std::max(capacity_, 1) * 32) d.resize((size_t)std::max(capacity_, 1) * 32);
    int n = 0;
    const int r = orbx_extract(ctx_, image.data, image.cols, image.rows, image.step, vLappingArea[0], vLappingArea[1], k.data(), d.data(),
                               capacity_, &n);
    if (r < 0) {
      keypoints.clear();
      descriptors.clear();
      throw orbx::Error(r, orbx_last_error(ctx_));
    }
    const orbx::KeyPoint* kp = reinterpret_cast<const orbx::KeyPoint*>(k.data());
    keypoints.assign(kp, kp + n);                              // _keypoints = vector(nkeypoints), cpp:1581
    descriptors.assign(d.begin(), d.begin() + (size_t)n * 32);  // rows == #keypoints, cols == 32, CV_8U
    return r;
  }
  std::vector<std::vector<uint8_t>> mvImagePyramid;  // filled on demand by imagePyramid()
#endif

  // mvImagePyramid[level] with an optional REFLECT_101 ring of `border` pixels (cpp:1689,1708)
  std::vector<uint8_t> imagePyramid(int level, int border, int* w = nullptr, int* h = nullptr) {
    int lw = 0, lh = 0;
    if (orbx_level_size(ctx_, level, &lw, &lh) != ORBX_OK) throw orbx::Error(ORBX_E_BADARG, "no pyramid yet");
    std::vector<uint8_t> out((size_t)(lw + 2 * border) * (lh + 2 * border));
    const int r = orbx_download_pyramid(ctx_, 0, level, border, out.data(), lw + 2 * border);
    if (r != ORBX_OK) throw orbx::Error(r, orbx_last_error(ctx_));
    if (w) *w = lw + 2 * border;
    if (h) *h = lh + 2 * border;
    return out;
  }

  int inline GetLevels() { return nlevels_; }
  float inline GetScaleFactor() { return orbx_get_scale_factor(ctx_); }
  std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
  std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
  std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
  std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }
  std::vector<int> inline GetNumFeaturesPerLevel() { return mnFeaturesPerLevel; }

  orbx_ctx* context() { return ctx_; }

  // Optional: page-lock a frame buffer the host reuses (orbx_host_register) -- operator() then takes the image with one DMA
  // instead of the runtime's staging copy.  Unpin before freeing the buffer.
  void PinHostBuffer(void* ptr, size_t bytes) {
    const int r = orbx_host_register(ctx_, ptr, bytes);
    if (r != ORBX_OK) throw orbx::Error(r, orbx_last_error(ctx_));
  }
  void UnpinHostBuffer(void* ptr) { (void)orbx_host_unregister(ctx_, ptr); }

  // Drop-ins for the bodies of Frame::UndistortKeyPoints / Frame::ComputeImageBounds (SlamTypes/Frame.cpp:101-161):
  //   void Frame::UndistortKeyPoints() { mpORBextractor->UndistortKeyPoints(mvKeys, cam, mvKeysUn); N = mvKeysUn.size(); }
  //   void Frame::ComputeImageBounds() { mpORBextractor->ComputeImageBounds(cam, im.cols, im.rows, mnMinX, mnMaxX, mnMinY, mnMaxY); }
  // with cam = {mK(0,0), mK(1,1), mK(0,2), mK(1,2), mDistCoef(0..3)} (all CV_32F, Settings.hpp:28-39).
  // Drop-in for the body of Converter::toGray (Utils/Converter.cpp:5-19): `in` has `channels` interleaved bytes per pixel
  // (stride in bytes); out becomes width x height, tightly packed.  false = "Wrong image format", like upstream.
  bool toGray(const orbx::Image8& in, int channels, std::vector<uint8_t>& outImGray, bool bRGB = false) {
    outImGray.assign(in.empty() ? 0 : (size_t)in.cols * in.rows, 0);
    const int r = orbx_to_gray(ctx_, in.data, in.cols, in.rows, in.step, channels, bRGB ? 1 : 0, outImGray.data(), in.cols);
    if (r == ORBX_E_BADARG && channels != 1 && channels != 3) {
      std::cerr << "ERROR: Wrong image format" << std::endl;  // Converter.cpp:17
      return false;
    }
    if (r != ORBX_OK) throw orbx::Error(r, orbx_last_error(ctx_));
    return true;
  }

  // Drop-ins for the RANSAC scoring of Initializer::FindHomography / FindFundamental (Initialization/Initializer.cpp:180-211,
  // 231-265): all hypotheses of the loop are scored in one call.  H21s / H12s / F21s hold nModels row-major 3x3 matrices,
  // vnMatches12 is the matcher's output the Initializer builds mvMatches12 from (:24-33).  Returns the index the loop would
  // keep (-1 = none), scores[m], and vbMatchesInliers of every model (inliers[m * N + i]).
  int CheckHomography(int nModels, const float* H21s, const float* H12s, const std::vector<KeyPointT>& mvKeys1,
                      const std::vector<KeyPointT>& mvKeys2, const std::vector<int>& vnMatches12, float sigma,
                      std::vector<float>& scores, std::vector<uint8_t>& inliers, int& N) {
    scores.assign((size_t)nModels, 0.f);
    inliers.assign((size_t)nModels * mvKeys1.size() + 1, 0);
    int best = -1;
    const int r = orbx_check_homography(ctx_, nModels, H21s, H12s, reinterpret_cast<const orbx_keypoint*>(mvKeys1.data()),
                                        (int)mvKeys1.size(), reinterpret_cast<const orbx_keypoint*>(mvKeys2.data()),
                                        (int)mvKeys2.size(), vnMatches12.data(), sigma, scores.data(), inliers.data(), &N, &best);
    if (r != ORBX_OK) throw orbx::Error(r, orbx_last_error(ctx_));
    return best;
  }
  int CheckFundamental(int nModels, const float* F21s, const std::vector<KeyPointT>& mvKeys1,
                       const std::vector<KeyPointT>& mvKeys2, const std::vector<int>& vnMatches12, float sigma,
                       std::vector<float>& scores, std::vector<uint8_t>& inliers, int& N) {
    scores.assign((size_t)nModels, 0.f);
    inliers.assign((size_t)nModels * mvKeys1.size() + 1, 0);
    int best = -1;
    const int r = orbx_check_fundamental(ctx_, nModels, F21s, reinterpret_cast<const orbx_keypoint*>(mvKeys1.data()),
                                         (int)mvKeys1.size(), reinterpret_cast<const orbx_keypoint*>(mvKeys2.data()),
                                         (int)mvKeys2.size(), vnMatches12.data(), sigma, scores.data(), inliers.data(), &N, &best);
    if (r != ORBX_OK) throw orbx::Error(r, orbx_last_error(ctx_));
    return best;
  }

  // Drop-in for the candidate loop of Initializer::ReconstructHF (Initialization/Initializer.cpp:497-515): every (R, t) of the
  // decomposition is checked in one call.  R21s: nModels row-major 3x3, t21s: nModels x 3, K: row-major 3x3 (all float);
  // vbMatchesInliers has one flag per match (the pairs with vnMatches12[i] >= 0, in order).  Per model m: the return value of
  // CheckRT in nGood[m], vbTriGood in triGood[m * N1 + k], vP3D in p3d[(m * N1 + k) * 3 ..], parallax[m].
  void CheckRT(int nModels, const float* R21s, const float* t21s, const float* K, const std::vector<KeyPointT>& vKeys1,
               const std::vector<KeyPointT>& vKeys2, const std::vector<int>& vnMatches12, const std::vector<uint8_t>& vbMatchesInliers,
               float th2, std::vector<int>& nGood, std::vector<uint8_t>& triGood, std::vector<float>& p3d, std::vector<float>& parallax) {
    nGood.assign((size_t)nModels, 0);
    parallax.assign((size_t)nModels, 0.f);
    triGood.assign((size_t)nModels * vKeys1.size() + 1, 0);
    p3d.assign((size_t)nModels * vKeys1.size() * 3 + 3, 0.f);
    const int r = orbx_check_rt(ctx_, nModels, R21s, t21s, K, reinterpret_cast<const orbx_keypoint*>(vKeys1.data()), (int)vKeys1.size(),
                                reinterpret_cast<const orbx_keypoint*>(vKeys2.data()), (int)vKeys2.size(), vnMatches12.data(),
                                vbMatchesInliers.data(), th2, nGood.data(), triGood.data(), p3d.data(), parallax.data());
    if (r != ORBX_OK) throw orbx::Error(r, orbx_last_error(ctx_));
  }

  void UndistortKeyPoints(const std::vector<KeyPointT>& mvKeys, const orbx_camera& cam, std::vector<KeyPointT>& mvKeysUn) {
    mvKeysUn.resize(mvKeys.size());
    const int r = orbx_undistort_keypoints(ctx_, reinterpret_cast<const orbx_keypoint*>(mvKeys.data()), (int)mvKeys.size(), &cam,
                                           reinterpret_cast<orbx_keypoint*>(mvKeysUn.data()));
    if (r != ORBX_OK) throw orbx::Error(r, orbx_last_error(ctx_));
  }
  void ComputeImageBounds(const orbx_camera& cam, int cols, int rows, int& mnMinX, int& mnMaxX, int& mnMinY, int& mnMaxY) {
    orbx_bounds b{0, 0, 0, 0};
    const int r = orbx_image_bounds(ctx_, &cam, cols, rows, &b);
    if (r != ORBX_OK) throw orbx::Error(r, orbx_last_error(ctx_));
    mnMinX = b.min_x; mnMaxX = b.max_x; mnMinY = b.min_y; mnMaxY = b.max_y;
  }

 protected:
#ifdef ORBX_WITH_OPENCV
  void refreshPyramid() {
    for (int l = 0; l < nlevels_; l++) {
      int w = 0, h = 0;
      if (orbx_level_size(ctx_, l, &w, &h) != ORBX_OK) return;
      mvImagePyramid[l].create(h, w, CV_8UC1);
      orbx_download_pyramid(ctx_, 0, l, 0, mvImagePyramid[l].data, (int)mvImagePyramid[l].step);
    }
  }
#endif
  orbx_ctx* ctx_ = nullptr;
  std::vector<orbx_keypoint> scratchK_;
  std::vector<uint8_t> scratchD_;
  int nfeatures_, nlevels_, capacity_ = 0;
  std::vector<int> mnFeaturesPerLevel;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

// What ORBmatcher reads from a Frame (Features/ORBmatcher.cpp:14,28,37,44,51,59,109; Frame.cpp:163-206): mvKeysUn,
// mDescriptors (N x 32 contiguous bytes), N and the static image bounds.
struct FrameView {
  const KeyPointT* mvKeysUn = nullptr;
  const uint8_t* mDescriptors = nullptr;
  int N = 0;
  int mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
};

class ORBmatcher {
 public:
  // Features/ORBmatcher.hpp:15.  (The optional third argument pins the device context; without it the matcher uses the
  // extractor the frames were built with, Frame::mpORBextractor, SlamTypes/Frame.hpp:60.)
  ORBmatcher(float nnratio = 0.6, bool checkOri = true, ORBextractor* extractor = nullptr)
      : mfNNratio(nnratio), mbCheckOrientation(checkOri), ext_(extractor) {}
  void setExtractor(ORBextractor* e) { ext_ = e; }

  // Features/ORBmatcher.hpp:36 -- `SearchForInitialization(F1, F2, vnMatches12, 100)` exactly as the reference's callers
  // write it (demo_initialization.cpp:105-108, tracking.cpp:101-102), for the reference's Frame or any Frame-like type:
  // public mvKeysUn (vector<KeyPoint>), mDescriptors (N x 32 bytes), N, mpORBextractor (the shim's ORBextractor*), and the
  // statics mnMinX / mnMaxX / mnMinY / mnMaxY.  The device context comes from F1.mpORBextractor (F2's if F1 has none).
  template <class FrameT, class = typename std::enable_if<!std::is_same<typename std::decay<FrameT>::type, FrameView>::value>::type>
  int SearchForInitialization(FrameT& F1, FrameT& F2, std::vector<int>& vnMatches12, int windowSize = 100) {
    ORBextractor* e = ext_ ? ext_ : (F1.mpORBextractor ? F1.mpORBextractor : F2.mpORBextractor);
    return search(e, frameView(F1), frameView(F2), vnMatches12, windowSize);
  }
  template <class FrameT>
  static FrameView frameView(const FrameT& F) {
    FrameView v;
    v.mvKeysUn = F.mvKeysUn.data();
    v.mDescriptors = orbx::descriptorBytes(F.mDescriptors);
    v.N = F.N;
    v.mnMinX = FrameT::mnMinX; v.mnMaxX = FrameT::mnMaxX; v.mnMinY = FrameT::mnMinY; v.mnMaxY = FrameT::mnMaxY;
    return v;
  }

  // the same on plain views (hosts that keep keypoints / descriptors in their own containers); needs the extractor
  // given to the constructor or setExtractor
  int SearchForInitialization(const FrameView& F1, const FrameView& F2, std::vector<int>& vnMatches12, int windowSize = 100) {
    return search(ext_, F1, F2, vnMatches12, windowSize);
  }

  // Features/ORBmatcher.cpp:5-7 defines these out of class; `inline` gives the header-only shim a definition too, so that an
  // odr-use (std::min(ORBmatcher::TH_LOW, d)) links
  inline static constexpr int HISTO_LENGTH = 30;
  inline static constexpr int TH_LOW = 50;
  inline static constexpr int TH_HIGH = 100;

 private:
  int search(ORBextractor* e, const FrameView& F1, const FrameView& F2, std::vector<int>& vnMatches12, int windowSize) {
    if (!e) throw orbx::Error(ORBX_E_BADARG, "ORBmatcher: no ORBextractor (device context): the frames carry none and none was set");
    vnMatches12.assign(F1.N, -1);  // cpp:14
    orbx_bounds b{F2.mnMinX, F2.mnMaxX, F2.mnMinY, F2.mnMaxY};
    orbx_match_stats st{0, 0, 0};
    int nmatches = 0;
    const int r = orbx_match_init(e->context(), reinterpret_cast<const orbx_keypoint*>(F1.mvKeysUn), F1.mDescriptors, F1.N,
                                  reinterpret_cast<const orbx_keypoint*>(F2.mvKeysUn), F2.mDescriptors, F2.N, &b, windowSize,
                                  mfNNratio, mbCheckOrientation ? 1 : 0, vnMatches12.data(), &nmatches, &st);
    if (r != ORBX_OK) throw orbx::Error(r, orbx_last_error(e->context()));
    if (orbx::verbose()) {  // the reference prints these four lines on every call (cpp:144-147)
      std::cout << "SearchForInitialization done ----------------------" << std::endl;
      std::cout << "invalidMatchByDistance: " << st.invalid_by_distance << std::endl;
      std::cout << "invalidMatchByRatio: " << st.invalid_by_ratio << std::endl;
      std::cout << "invalidMatchByOrientation: " << st.invalid_by_orientation << std::endl;
    }
    return nmatches;
  }

  float mfNNratio;
  bool mbCheckOrientation;
  ORBextractor* ext_;
};

}  // namespace ORB_SLAM_Tracking
