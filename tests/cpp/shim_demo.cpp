// shim_demo.cpp — the reference's demo_initialization.cpp:65-113 call sequence, written against include/orbx_shim.hpp
// (POD types, no OpenCV):  extractor(2 frames) -> Frame-equivalent views -> ORBmatcher::SearchForInitialization.
// usage: shim_demo W H frameA.raw frameB.raw nfeatures iniTh minTh [distort]
//   distort = 1: the Frame constructor's undistortion + image bounds with the camera of Settings.yaml (Frame.cpp:44,64)
// prints: N1 N2 nmatches fnv1a(keypoints1) fnv1a(desc1) fnv1a(matches12)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "orbx_shim.hpp"

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
  const int W = std::atoi(argv[1]), H = std::atoi(argv[2]);
  std::vector<uint8_t> a = readRaw(argv[3], (size_t)W * H), b = readRaw(argv[4], (size_t)W * H);
  using namespace ORB_SLAM_Tracking;
  try {
    ORBextractor extractor(std::atoi(argv[5]), 1.2f, 8, std::atoi(argv[6]), std::atoi(argv[7]), W, H);  // demo :72
    std::vector<int> unused{0, 0};                                                                   // Frame.cpp:58
    std::vector<orbx::KeyPoint> k1, k2;
    std::vector<uint8_t> d1, d2;
    orbx::Image8 imA{a.data(), W, H, W}, imB{b.data(), W, H, W}, noMask;
    const int r1 = extractor(imA, noMask, k1, d1, unused);
    const int r2 = extractor(imB, noMask, k2, d2, unused);
    if (r1 != (int)k1.size() || r2 != (int)k2.size()) return 3;  // monoIndex == N for a {0,0} lapping area
    int minX = 0, maxX = W, minY = 0, maxY = H;
    std::vector<orbx::KeyPoint> u1 = k1, u2 = k2;  // mvKeysUn
    if (argc > 8 && std::atoi(argv[8]) == 1) {
      const orbx_camera cam{609.2855f, 609.3422f, 351.4274f, 237.7324f, -0.3492f, 0.1363f, 0.0f, 0.0f};
      extractor.ComputeImageBounds(cam, W, H, minX, maxX, minY, maxY);  // Frame.cpp:44
      extractor.UndistortKeyPoints(k1, cam, u1);                         // Frame.cpp:64
      extractor.UndistortKeyPoints(k2, cam, u2);
    }
    FrameView f1{u1.data(), d1.data(), (int)u1.size(), minX, maxX, minY, maxY}, f2{u2.data(), d2.data(), (int)u2.size(), minX, maxX, minY, maxY};
    ORBmatcher orbMatcher(0.9f, true, &extractor);  // demo :105
    std::vector<int> mvMatches;
    const int nmatches = orbMatcher.SearchForInitialization(f1, f2, mvMatches, 100);  // demo :108
    // odr-uses of the matcher's constants (Features/ORBmatcher.cpp:5-7 defines them out of class; the shim's are inline)
    const int& thLow = std::min(ORBmatcher::TH_LOW, ORBmatcher::TH_HIGH);
    if (thLow != 50 || *&ORBmatcher::HISTO_LENGTH != 30) return 5;
    std::printf("RESULT %zu %zu %d %llu %llu %llu\n", k1.size(), k2.size(), nmatches, fnv(k1.data(), k1.size() * sizeof(orbx::KeyPoint)),
                fnv(d1.data(), d1.size()), fnv(mvMatches.data(), mvMatches.size() * sizeof(int)));
  } catch (const orbx::Error& e) {
    std::fprintf(stderr, "orbx error %d: %s\n", e.code, e.what());
    return 4;
  }
  return 0;
}
