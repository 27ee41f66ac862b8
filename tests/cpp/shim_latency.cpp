// shim_latency.cpp — what the reference's own call costs through the C++ drop-in classes (include/orbx_shim.hpp, POD types):
//   (*mpORBextractor)(im, cv::Mat(), mvKeys, mDescriptors, vLapping)     SlamTypes/Frame.cpp:58-60   (one frame per call)
//   orbMatcher.SearchForInitialization(F1, F2, mvMatches, 100)            demo/demo_initialization.cpp:105-108
// usage: shim_latency W H frameA.raw frameB.raw nfeatures iniTh minTh [reps]
// prints one JSON line: extract_ms_per_frame, match_ms_per_pair (medians of `reps` calls after 20 warm-ups), N, nmatches.
#include <algorithm>
#include <chrono>
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
static double nowMs() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static double median(std::vector<double> v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main(int argc, char** argv) {
  if (argc < 8) return 2;
  const int W = std::atoi(argv[1]), H = std::atoi(argv[2]);
  const int reps = argc > 8 ? std::atoi(argv[8]) : 300;
  std::vector<uint8_t> a = readRaw(argv[3], (size_t)W * H), b = readRaw(argv[4], (size_t)W * H);
  using namespace ORB_SLAM_Tracking;
  orbx::verbose() = false;
  if (std::getenv("ORBX_LAT_TRACE")) orbx_debug_set("lat_trace", 1);  // (tools/cpp_latency.sh: the library itself reads no environment)
  try {
    ORBextractor extractor(std::atoi(argv[5]), 1.2f, 8, std::atoi(argv[6]), std::atoi(argv[7]), W, H);
    std::vector<int> unused{0, 0};
    std::vector<orbx::KeyPoint> k1, k2;
    std::vector<uint8_t> d1, d2;
    orbx::Image8 imA{a.data(), W, H, W}, imB{b.data(), W, H, W}, noMask;
    for (int i = 0; i < 20; i++) { extractor(imA, noMask, k1, d1, unused); extractor(imB, noMask, k2, d2, unused); }
    std::vector<double> te, tm;
    for (int i = 0; i < reps; i++) {
      const double t0 = nowMs();
      extractor((i & 1) ? imB : imA, noMask, (i & 1) ? k2 : k1, (i & 1) ? d2 : d1, unused);
      te.push_back(nowMs() - t0);
    }
    // the same with the two frame buffers page-locked once (ORBextractor::PinHostBuffer: a host that reuses its frame buffers)
    extractor.PinHostBuffer(a.data(), a.size());
    extractor.PinHostBuffer(b.data(), b.size());
    std::vector<double> tp;
    for (int i = 0; i < 20; i++) extractor(imA, noMask, k1, d1, unused);
    for (int i = 0; i < reps; i++) {
      const double t0 = nowMs();
      extractor((i & 1) ? imB : imA, noMask, (i & 1) ? k2 : k1, (i & 1) ? d2 : d1, unused);
      tp.push_back(nowMs() - t0);
    }
    extractor.UnpinHostBuffer(a.data());
    extractor.UnpinHostBuffer(b.data());
    FrameView f1{k1.data(), d1.data(), (int)k1.size(), 0, W, 0, H}, f2{k2.data(), d2.data(), (int)k2.size(), 0, W, 0, H};
    ORBmatcher orbMatcher(0.9f, true, &extractor);
    std::vector<int> mvMatches;
    int nmatches = 0;
    for (int i = 0; i < 20; i++) nmatches = orbMatcher.SearchForInitialization(f1, f2, mvMatches, 100);
    for (int i = 0; i < reps; i++) {
      const double t0 = nowMs();
      nmatches = orbMatcher.SearchForInitialization(f1, f2, mvMatches, 100);
      tm.push_back(nowMs() - t0);
    }
    std::printf("{\"extract_ms_per_frame\": %.5f, \"extract_ms_per_frame_pinned_input\": %.5f, \"match_ms_per_pair\": %.5f, \"extract_ms_p10\": %.5f, \"extract_ms_p90\": %.5f, "
                "\"keypoints\": [%zu, %zu], \"nmatches\": %d, \"reps\": %d}\n",
                median(te), median(tp), median(tm), [&] { auto v = te; std::sort(v.begin(), v.end()); return v[v.size() / 10]; }(),
                [&] { auto v = te; std::sort(v.begin(), v.end()); return v[v.size() * 9 / 10]; }(), k1.size(), k2.size(), nmatches, reps);
  } catch (const orbx::Error& e) {
    std::fprintf(stderr, "orbx error %d: %s\n", e.code, e.what());
    return 4;
  }
  return 0;
}
