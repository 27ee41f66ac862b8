// orbx_oracle.cpp — CPU restatement of the ORB extract + init-match hot path.
//
// *** TEST INFRASTRUCTURE ONLY ***  Nothing in the product (orb_slam_tracking_amd/, include/) may
// include, link or call this file.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg load liborbx_oracle.so, and only as the checker / the timed CPU baseline.
//
// *** PARITY UNPINNED ***  The reference (zeal-up/ORB_SLAM_Tracking) cannot be built in this
// image (needs OpenCV, Eigen, GTest, an un-fetched submodule) and has no tests or golden
// vectors for this path.  The pixel arithmetic it delegates to OpenCV (FAST, resize,
// GaussianBlur, fastAtan2, cvRound) is restated here from OpenCV 4.x's published generic C++
// behaviour.  Every function cites the reference file:line it follows.
//
// Build: see oracle/Makefile  (g++ -O3 -ffp-contract=off, no -march, like the reference's
// Release build, CMakeLists.txt:8-10).
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <list>
#include <thread>
#ifdef __linux__
#include <pthread.h>
#include <sched.h>
#endif
#include <utility>
#include <vector>

namespace {

// cv::KeyPoint layout: pt.x pt.y size angle response octave class_id = 28 B
struct KP {
  float x, y, size, angle, response;
  int32_t octave, class_id;
};
static_assert(sizeof(KP) == 28, "KeyPoint must be 28 bytes");

// cvRound: round-half-to-even.  Default FP environment = FE_TONEAREST.
inline int cvRoundF(float v) { return (int)lrintf(v); }
inline int cvRoundD(double v) { return (int)lrint(v); }

const int PATCH_SIZE = 31;       // Features/ORBextractor.cpp:87
const int HALF_PATCH_SIZE = 15;  // :89
const int EDGE_THRESHOLD = 19;   // :90
const float factorPI = (float)(M_PI / 180.f);  // :92

const int8_t kPattern[256 * 4] = {
#include "orbx_pattern_data.inc"
};  // Features/ORBextractor.cpp:233-490 (data table, see tools/gen_pattern.py)

struct Image {
  int w = 0, h = 0;
  std::vector<uint8_t> px;  // tightly packed, stride == w
  uint8_t at(int y, int x) const { return px[(size_t)y * w + x]; }
};

// ---------------------------------------------------------------------------------------------
// cv::resize(INTER_LINEAR), 8UC1, generic path.  Called at ORBextractor.cpp:1676.
// ---------------------------------------------------------------------------------------------
void resizeLinear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                  int dstride) {
  const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
  const double scale_x = 1.0 / inv_x, scale_y = 1.0 / inv_y;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> alpha(2 * dw), beta(2 * dh);
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)std::floor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    alpha[2 * dx] = (short)cvRoundF((1.f - fx) * 2048);
    alpha[2 * dx + 1] = (short)cvRoundF(fx * 2048);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)std::floor(fy);
    fy -= sy;
    yofs[dy] = sy;
    beta[2 * dy] = (short)cvRoundF((1.f - fy) * 2048);
    beta[2 * dy + 1] = (short)cvRoundF(fy * 2048);
  }
  std::vector<int> row0(dw), row1(dw);
  auto hline = [&](int sy, std::vector<int>& out) {
    sy = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);  // clip(), rows only
    const uint8_t* S = src + (size_t)sy * sstride;
    for (int dx = 0; dx < dw; dx++) {
      int sx = xofs[dx];
      int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;  // weight is 0 whenever this clamps
      out[dx] = S[sx] * alpha[2 * dx] + S[sx1] * alpha[2 * dx + 1];
    }
  };
  for (int dy = 0; dy < dh; dy++) {
    hline(yofs[dy], row0);
    hline(yofs[dy] + 1, row1);
    const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
    uint8_t* D = dst + (size_t)dy * dstride;
    for (int dx = 0; dx < dw; dx++) {
      int v = (((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2;
      D[dx] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  }
}

inline int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) {
    if (p < 0) p = -p;
    else p = 2 * n - 2 - p;
  }
  return p;
}

// ---------------------------------------------------------------------------------------------
// cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on a non-ROI 8U image, bit-exact
// fixed-point path: Q8 taps from the error-diffusion rule => [18,34,48,56,48,34,18] (sum 256);
// horizontal sum kept in Q8 (u16), vertical in Q16, round to nearest.  Call site: cpp:1598-1606.
// ---------------------------------------------------------------------------------------------
// Both version-dependent constants of the path are selectable (orbo_set_opencv_variant, mirrored by the product's
// orbx_set_opencv_variant): the Gaussian Q8 taps below and the BGR2GRAY coefficients of toGray.
//   variant 0: error diffusion, sum 256 (getGaussianKernelFixedPoint_ED, OpenCV >= 4.1.1 / 3.4.7)  [from-knowledge]
//   variant 1: every tap rounded, sum 257 (bit-exact path of 3.4.1 .. 4.1.0, integer filter before) [from-knowledge]
const int kGaussTab[2][7] = {{18, 34, 48, 56, 48, 34, 18}, {18, 34, 49, 55, 49, 34, 18}};
int gGaussVariant = 0, gGrayVariant = 0;
#define kGauss (kGaussTab[gGaussVariant])

void gaussian7(const uint8_t* src, int w, int h, int stride, uint8_t* dst, int dstride) {
  std::vector<uint16_t> tmp((size_t)w * h);
  for (int y = 0; y < h; y++) {
    const uint8_t* S = src + (size_t)y * stride;
    uint16_t* T = &tmp[(size_t)y * w];
    for (int x = 0; x < w; x++) {
      int s = 0;
      if (x >= 3 && x + 3 < w) {
        for (int k = -3; k <= 3; k++) s += kGauss[k + 3] * S[x + k];
      } else {
        for (int k = -3; k <= 3; k++) s += kGauss[k + 3] * S[reflect101(x + k, w)];
      }
      T[x] = (uint16_t)s;
    }
  }
  for (int y = 0; y < h; y++) {
    const uint16_t* R[7];
    for (int k = -3; k <= 3; k++) R[k + 3] = &tmp[(size_t)reflect101(y + k, h) * w];
    uint8_t* D = dst + (size_t)y * dstride;
    for (int x = 0; x < w; x++) {
      uint32_t s = 0;
      for (int k = 0; k < 7; k++) s += (uint32_t)kGauss[k] * R[k][x];
      uint32_t v = (s + 32768u) >> 16;
      D[x] = (uint8_t)(v > 255 ? 255 : v);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// cv::FAST(img, kps, t, nonmaxSuppression=true), TYPE_9_16.  Call sites cpp:1109,1119.
// ---------------------------------------------------------------------------------------------
const int kRingDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
const int kRingDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// arc strength: max over the 16 arcs of 9 contiguous ring pixels of min(+-(v - p_k))
inline int fastStrength(const uint8_t* p, int stride) {
  int d[16];
  const int v = p[0];
  for (int k = 0; k < 16; k++) d[k] = v - p[kRingDy[k] * stride + kRingDx[k]];
  int best = INT_MIN;
  for (int s = 0; s < 16; s++) {
    int mn = INT_MAX, mx = INT_MIN;
    for (int j = 0; j < 9; j++) {
      int q = d[(s + j) & 15];
      mn = std::min(mn, q);
      mx = std::max(mx, q);
    }
    best = std::max(best, std::max(mn, -mx));
  }
  return best;
}

struct Cand {
  float x, y, response;
};

// 16-bit circular mask has a run of >= 9 set bits
inline bool hasArc9(uint32_t m) {
  m |= m << 16;
  uint32_t r = m & (m >> 1);
  r &= r >> 2;
  r &= r >> 4;
  r &= m >> 8;
  return (r & 0xFFFFu) != 0;
}

void fastDetect(const uint8_t* img, int w, int h, int stride, int t, bool nms, std::vector<Cand>& out) {
  out.clear();
  t = std::min(std::max(t, 0), 255);
  if (w < 7 || h < 7) return;
  std::vector<int> score((size_t)w * h, 0);
  std::vector<uint8_t> corner((size_t)w * h, 0);
  int off[16];
  for (int k = 0; k < 16; k++) off[k] = kRingDy[k] * stride + kRingDx[k];
  for (int y = 3; y < h - 3; y++)
    for (int x = 3; x < w - 3; x++) {
      const uint8_t* p = img + (size_t)y * stride + x;
      const int v = p[0], lo = v - t, hi = v + t;
      // high-speed rejection on the 4 compass pairs (an arc of 9 covers one pixel of every opposite pair)
      uint32_t br = 0, dk = 0;
      bool alive = true;
      for (int k = 0; k < 8 && alive; k += 2) {
        const int a = p[off[k]], b = p[off[k + 8]];
        br |= (uint32_t)(a > hi) << k | (uint32_t)(b > hi) << (k + 8);
        dk |= (uint32_t)(a < lo) << k | (uint32_t)(b < lo) << (k + 8);
        const uint32_t pairBit = (1u << k) | (1u << (k + 8));
        if (!(br & pairBit) && !(dk & pairBit)) alive = false;
      }
      if (!alive) continue;
      for (int k = 1; k < 16; k += 2) {
        const int a = p[off[k]];
        br |= (uint32_t)(a > hi) << k;
        dk |= (uint32_t)(a < lo) << k;
      }
      if (!hasArc9(br) && !hasArc9(dk)) continue;
      corner[(size_t)y * w + x] = 1;
      score[(size_t)y * w + x] = fastStrength(p, stride) - 1;  // cornerScore = max(t, strength) - 1
    }
  for (int y = 3; y < h - 3; y++)
    for (int x = 3; x < w - 3; x++) {
      if (!corner[(size_t)y * w + x]) continue;
      const int sc = score[(size_t)y * w + x];
      if (nms) {
        bool keep = true;
        for (int dy = -1; dy <= 1 && keep; dy++)
          for (int dx = -1; dx <= 1; dx++) {
            if (!dx && !dy) continue;
            if (!(sc > score[(size_t)(y + dy) * w + (x + dx)])) { keep = false; break; }
          }
        if (!keep) continue;
      }
      out.push_back({(float)x, (float)y, (float)sc});
    }
}

// ---------------------------------------------------------------------------------------------
// cv::fastAtan2, degrees in [0,360).  Call site cpp:158.
// ---------------------------------------------------------------------------------------------
float fastAtan2(float y, float x) {
  static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
  static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
  static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
  static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
  const float eps = (float)2.2204460492503131e-16;
  float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + eps);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + eps);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// ---------------------------------------------------------------------------------------------
// Quadtree node (Features/ORBextractor.hpp:32-53) and DivideNode (cpp:617-676)
// ---------------------------------------------------------------------------------------------
struct Node {
  std::vector<Cand> keys;
  int ulx = 0, uly = 0, urx = 0, ury = 0, blx = 0, bly = 0, brx = 0, bry = 0;
  std::list<Node>::iterator lit;
  bool noMore = false;

  void divide(Node out[4]) const {
    const int halfX = (int)std::ceil((float)(urx - ulx) / 2);
    const int halfY = (int)std::ceil((float)(bry - uly) / 2);
    Node &n1 = out[0], &n2 = out[1], &n3 = out[2], &n4 = out[3];
    n1.ulx = ulx; n1.uly = uly;
    n1.urx = ulx + halfX; n1.ury = uly;
    n1.blx = ulx; n1.bly = uly + halfY;
    n1.brx = ulx + halfX; n1.bry = uly + halfY;
    n2.ulx = n1.urx; n2.uly = n1.ury;
    n2.urx = urx; n2.ury = ury;
    n2.blx = n1.brx; n2.bly = n1.bry;
    n2.brx = urx; n2.bry = uly + halfY;
    n3.ulx = n1.blx; n3.uly = n1.bly;
    n3.urx = n1.brx; n3.ury = n1.bry;
    n3.blx = blx; n3.bly = bly;
    n3.brx = n1.brx; n3.bry = bly;
    n4.ulx = n3.urx; n4.uly = n3.ury;
    n4.urx = n2.brx; n4.ury = n2.bry;
    n4.blx = n3.brx; n4.bly = n3.bry;
    n4.brx = brx; n4.bry = bry;
    for (const Cand& kp : keys) {
      if (kp.x < n1.urx) {
        if (kp.y < n1.bry) n1.keys.push_back(kp);
        else n3.keys.push_back(kp);
      } else if (kp.y < n1.bry) {
        n2.keys.push_back(kp);
      } else {
        n4.keys.push_back(kp);
      }
    }
    for (int i = 0; i < 4; i++)
      if (out[i].keys.size() == 1) out[i].noMore = true;
  }
};

typedef std::pair<int, Node*> SizedNode;
// compareNodes, cpp:684-696 (used with the UNSTABLE std::sort, cpp:912)
bool nodeLess(const SizedNode& a, const SizedNode& b) {
  if (a.first < b.first) return true;
  if (a.first > b.first) return false;
  return a.second->ulx < b.second->ulx;
}

// ORBextractor::DistributeOctTree, cpp:698-1011.
std::vector<Cand> distributeOctTree(const std::vector<Cand>& cands, int minX, int maxX, int minY, int maxY,
                                    int N) {
  const int nIni = (int)std::round((float)(maxX - minX) / (maxY - minY));  // :706
  const float hX = (float)(maxX - minX) / nIni;                           // :709
  std::list<Node> nodes;
  std::vector<Node*> roots(nIni > 0 ? nIni : 0);
  for (int i = 0; i < nIni; i++) {
    Node ni;
    ni.ulx = (int)(hX * (float)i); ni.uly = 0;
    ni.urx = (int)(hX * (float)(i + 1)); ni.ury = 0;
    ni.blx = ni.ulx; ni.bly = maxY - minY;
    ni.brx = ni.urx; ni.bry = maxY - minY;
    nodes.push_back(ni);
    roots[i] = &nodes.back();
  }
  for (const Cand& kp : cands) roots[(int)(kp.x / hX)]->keys.push_back(kp);  // :747
  for (auto it = nodes.begin(); it != nodes.end();) {
    if (it->keys.size() == 1) { it->noMore = true; ++it; }
    else if (it->keys.empty()) it = nodes.erase(it);
    else ++it;
  }
  bool finish = false;
  std::vector<SizedNode> pending;
  // one DivideNode + push_front of the non-empty children, children with >1 keys are recorded
  auto splitInto = [&](const Node& parent, int* nToExpand) {
    Node ch[4];
    parent.divide(ch);
    for (int c = 0; c < 4; c++) {
      if (ch[c].keys.empty()) continue;
      nodes.push_front(ch[c]);
      if (ch[c].keys.size() > 1) {
        if (nToExpand) ++*nToExpand;
        pending.push_back(std::make_pair((int)ch[c].keys.size(), &nodes.front()));
        nodes.front().lit = nodes.begin();
      }
    }
  };
  while (!finish) {
    int prevSize = (int)nodes.size();
    int nToExpand = 0;
    pending.clear();
    for (auto it = nodes.begin(); it != nodes.end();) {  // :805-879
      if (it->noMore) { ++it; continue; }
      splitInto(*it, &nToExpand);
      it = nodes.erase(it);
    }
    if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) {  // :887-890
      finish = true;
    } else if ((int)nodes.size() + nToExpand * 3 > N) {  // :897
      while (!finish) {
        prevSize = (int)nodes.size();
        std::vector<SizedNode> prev = pending;
        pending.clear();
        std::sort(prev.begin(), prev.end(), nodeLess);  // :912
        for (int j = (int)prev.size() - 1; j >= 0; j--) {
          splitInto(*prev[j].second, nullptr);
          nodes.erase(prev[j].second->lit);
          if ((int)nodes.size() >= N) break;
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) finish = true;
      }
    }
  }
  std::vector<Cand> result;
  for (const Node& nd : nodes) {  // :984-1007, first maximum wins
    const Cand* best = &nd.keys[0];
    float maxResp = best->response;
    for (size_t k = 1; k < nd.keys.size(); k++)
      if (nd.keys[k].response > maxResp) { best = &nd.keys[k]; maxResp = nd.keys[k].response; }
    result.push_back(*best);
  }
  return result;
}

// ---------------------------------------------------------------------------------------------
// IC_Angle, cpp:103-159
// ---------------------------------------------------------------------------------------------
float icAngle(const Image& im, float px, float py, const int* umax, int* m10_out = nullptr, int* m01_out = nullptr) {
  int m_01 = 0, m_10 = 0;
  const int cx = cvRoundF(px), cy = cvRoundF(py);
  const uint8_t* center = &im.px[(size_t)cy * im.w + cx];
  for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
  const int step = im.w;
  for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
    int v_sum = 0;
    const int d = umax[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = center[v * step + u], val_minus = center[-v * step + u];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  if (m10_out) *m10_out = m_10;
  if (m01_out) *m01_out = m_01;
  return fastAtan2((float)m_01, (float)m_10);
}

// ---- the libm reading of cpp:174 and cpp:536 (orbo_set_libm_variant, mirrored by the product's orbx_set_libm_variant) ----
// cpp:174 calls UNQUALIFIED cos(angle) / sin(angle) on a float, and the file has no `using namespace std` (cpp:69-71 import
// list / pair / vector only).  Which function that is depends on what the OpenCV headers pull in:
//   * only <cmath>: the global namespace holds ::cos(double) alone -> the float is promoted, the double result converted back:
//     (float)cos((double)angle)                                                        = variant 0, ORBX_LIBM_DOUBLE
//   * libstdc++'s <math.h> wrapper anywhere in the include chain (it does `using std::cos;`): overload resolution picks
//     std::cos(float) = __builtin_cosf -> glibc's cosf                                 = variant 1, ORBX_LIBM_FLOAT
// glibc >= 2.28 evaluates sinf / cosf with an f64 polynomial that is NOT correctly rounded (sysdeps/ieee754/flt-32/
// s_sincosf.h, s_sinf.c, s_cosf.c, s_sincosf_data.c; the algorithm of ARM's optimized-routines): over the 1,135,869,953 f32
// angles in [0, 360] it differs from variant 0 for 483,807 (cos) / 1,001,902 (sin) of them.  Restated below operation for
// operation (public algorithm, constants from s_sincosf_data.c); tests/test_oracle.py sweeps EVERY angle against the host's own
// cosf / sinf.  The x86-64 multiarch build contracts the polynomials into FMAs on CPUs that have them, the baseline build does
// not: both forms are compiled here (USE_FMA) and both agree with each other after the rounding to f32 for every angle of the
// domain (same sweep), so one variant covers both.  The same reading decides pow(float, float) of the constructor (cpp:536).
int gLibmVariant = 1;  // ORBX_LIBM_FLOAT: the default since round 5 (include/orbx.h says why)
namespace glibc_sincosf {
struct SinCosT { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; };
const SinCosT kTab[2] = {
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5,
     -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5,
     0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
inline uint32_t abstop12(float x) { uint32_t u; std::memcpy(&u, &x, 4); return (u >> 20) & 0x7ff; }
template <bool FMA> inline double ma(double a, double b, double c) { return FMA ? std::fma(a, b, c) : a * b + c; }
// sinf_poly: sine polynomial for even n, cosine polynomial for odd n
template <bool FMA> inline float poly(double x, double x2, const SinCosT* p, int n) {
  if ((n & 1) == 0) {
    const double x3 = x * x2, s1 = ma<FMA>(x2, p->s3, p->s2), x7 = x3 * x2, s = ma<FMA>(x3, p->s1, x);
    return (float)ma<FMA>(x7, s1, s);
  }
  const double x4 = x2 * x2, c2 = ma<FMA>(x2, p->c4, p->c3), c1 = ma<FMA>(x2, p->c1, p->c0), x6 = x4 * x2, c = ma<FMA>(x4, p->c2, c1);
  return (float)ma<FMA>(x6, c2, c);
}
// reduce_fast (!TOINT_INTRINSICS: x86-64): quadrant from the scaled float-to-int conversion, one multiply-subtract
template <bool FMA> inline double reduceFast(double x, const SinCosT* p, int* np) {
  const double r = x * p->hpi_inv;
  const int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return FMA ? std::fma(-(double)n, p->hpi, x) : x - n * p->hpi;
}
// valid for 0 <= y < 120 (the keypoint angle in radians is below 6.2832); the huge-argument branch is not restated
template <bool FMA> float sinF(float y) {
  double x = y;
  const SinCosT* p = &kTab[0];
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    if (abstop12(y) < abstop12(0x1p-12f)) return y;
    return poly<FMA>(x, x * x, p, 0);
  }
  int n;
  x = reduceFast<FMA>(x, p, &n);
  const double s = p->sign[n & 3];
  if (n & 2) p = &kTab[1];
  return poly<FMA>(x * s, x * x, p, n);
}
template <bool FMA> float cosF(float y) {
  double x = y;
  const SinCosT* p = &kTab[0];
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
    return poly<FMA>(x, x * x, p, 1);
  }
  int n;
  x = reduceFast<FMA>(x, p, &n);
  const double s = p->sign[n & 3];
  if (n & 2) p = &kTab[1];
  return poly<FMA>(x * s, x * x, p, n ^ 1);
}
}  // namespace glibc_sincosf
inline void descSinCos(float angle, float* c, float* s) {
  if (gLibmVariant) { *c = glibc_sincosf::cosF<true>(angle); *s = glibc_sincosf::sinF<true>(angle); }
  else { *c = (float)std::cos((double)angle); *s = (float)std::sin((double)angle); }
}

// computeOrbDescriptor, cpp:169-228.  cos / sin of the float argument: see the libm variants above.
void orbDescriptor(const Image& blurred, const KP& kp, uint8_t* desc) {
  const float angle = kp.angle * factorPI;
  float c, s;
  descSinCos(angle, &c, &s);
  const int cx = cvRoundF(kp.x), cy = cvRoundF(kp.y);
  const uint8_t* center = &blurred.px[(size_t)cy * blurred.w + cx];
  const int step = blurred.w;
  for (int i = 0; i < 32; i++) {
    int val = 0;
    for (int b = 0; b < 8; b++) {
      const int8_t* p = &kPattern[(i * 8 + b) * 4];
      const float x0 = p[0], y0 = p[1], x1 = p[2], y1 = p[3];
      int t0 = center[cvRoundF(x0 * s + y0 * c) * step + cvRoundF(x0 * c - y0 * s)];
      int t1 = center[cvRoundF(x1 * s + y1 * c) * step + cvRoundF(x1 * c - y1 * s)];
      val |= (t0 < t1) << b;
    }
    desc[i] = (uint8_t)val;
  }
}

// ---------------------------------------------------------------------------------------------
// ORBextractor: ctor cpp:492-595, operator() cpp:1531-1653
// ---------------------------------------------------------------------------------------------
struct Extractor {
  int nfeatures;
  double scaleFactor;  // member is double (hpp:142)
  int nlevels, iniTh, minTh;
  std::vector<float> scale, invScale, sigma2, invSigma2;
  std::vector<int> quota;
  int umax[HALF_PATCH_SIZE + 1];
  // state of the last call (mvImagePyramid without the 19-px ring, plus test hooks)
  std::vector<Image> pyr;
  std::vector<std::vector<Cand>> cands;     // per level, coords relative to (minBorderX,minBorderY)
  std::vector<std::vector<KP>> selected;    // per level, level coords, with angle
  bool keepBlurred = false;
  std::vector<Image> blurred;

  Extractor(int nf, float sf, int nl, int ini, int mn) : nfeatures(nf), scaleFactor(sf), nlevels(nl), iniTh(ini), minTh(mn) {
    scale.resize(nl); sigma2.resize(nl); invScale.resize(nl); invSigma2.resize(nl);
    scale[0] = 1.0f; sigma2[0] = 1.0f;
    for (int i = 1; i < nl; i++) {
      scale[i] = (float)(scale[i - 1] * scaleFactor);  // float * double -> float
      sigma2[i] = scale[i] * scale[i];
    }
    for (int i = 0; i < nl; i++) { invScale[i] = 1.0f / scale[i]; invSigma2[i] = 1.0f / sigma2[i]; }
    quota.resize(nl);
    const float factor = (float)(1.0f / scaleFactor);
    // cpp:536 pow(float, float): the same overload question as cos / sin (libm variants above): ::pow(double, double) or powf.
    // powf is the host libm's (what a reference built on this host calls); it differs from the double reading for 89 k of the
    // 134 M (factor, nlevels) pairs with factor in [0.5, 1) and never for a two-decimal scale factor 1.01 .. 2.00
    const float powv = gLibmVariant ? powf(factor, (float)nl) : (float)std::pow((double)factor, (double)(float)nl);
    float desired = nfeatures * (1 - factor) / (1 - powv);
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
      quota[l] = cvRoundF(desired);
      sum += quota[l];
      desired *= factor;
    }
    quota[nl - 1] = std::max(nfeatures - sum, 0);
    // umax, cpp:562-594
    const int vmax = (int)std::floor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
    const int vmin = (int)std::ceil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (int v = 0; v <= HALF_PATCH_SIZE; v++) umax[v] = 0;
    for (int v = 0; v <= vmax; ++v) umax[v] = cvRoundD(std::sqrt(hp2 - v * v));
    for (int v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
  }

  // ComputePyramid, cpp:1660-1713 (19-px REFLECT_101 ring not materialised: never read)
  void computePyramid(const uint8_t* img, int w, int h, int stride) {
    pyr.assign(nlevels, Image());
    for (int l = 0; l < nlevels; l++) {
      Image& L = pyr[l];
      L.w = cvRoundF(w * invScale[l]);
      L.h = cvRoundF(h * invScale[l]);
      L.px.resize((size_t)L.w * L.h);
      if (l == 0) {
        for (int y = 0; y < h; y++) memcpy(&L.px[(size_t)y * w], img + (size_t)y * stride, w);
      } else {
        const Image& P = pyr[l - 1];
        resizeLinear(P.px.data(), P.w, P.h, P.w, L.px.data(), L.w, L.h, L.w);
      }
    }
  }

  // cell loops of ComputeKeyPointsOctTree, cpp:1051-1141
  void levelCandidates(int level, std::vector<Cand>& out) const {
    const Image& L = pyr[level];
    out.clear();
    const int minBX = EDGE_THRESHOLD - 3, minBY = minBX;
    const int maxBX = L.w - EDGE_THRESHOLD + 3, maxBY = L.h - EDGE_THRESHOLD + 3;
    const float W = 35;
    const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
    std::vector<Cand> cell;
    for (int i = 0; i < nRows; i++) {
      const float iniY = (float)(minBY + i * hCell);
      float maxY = iniY + hCell + 6;
      if (iniY >= maxBY - 3) continue;
      if (maxY > maxBY) maxY = (float)maxBY;
      for (int j = 0; j < nCols; j++) {
        const float iniX = (float)(minBX + j * wCell);
        float maxX = iniX + wCell + 6;
        if (iniX >= maxBX - 6) continue;
        if (maxX > maxBX) maxX = (float)maxBX;
        const int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY;
        const uint8_t* roi = &L.px[(size_t)y0 * L.w + x0];
        fastDetect(roi, x1 - x0, y1 - y0, L.w, iniTh, true, cell);
        if (cell.empty()) fastDetect(roi, x1 - x0, y1 - y0, L.w, minTh, true, cell);
        for (Cand c : cell) {
          c.x += j * wCell;
          c.y += i * hCell;
          out.push_back(c);
        }
      }
    }
  }

  // operator(), cpp:1531-1653.  Returns monoIndex (>=0) or -1 for an empty image, -3 if a level
  // is too small for the reference's cell arithmetic (UB upstream, SURVEY 8(b) "Errors").
  int extract(const uint8_t* img, int w, int h, int stride, int lap0, int lap1, std::vector<KP>& kpsOut,
              std::vector<uint8_t>& descOut) {
    kpsOut.clear(); descOut.clear();
    if (!img || w <= 0 || h <= 0) return -1;
    computePyramid(img, w, h, stride);
    for (int l = 0; l < nlevels; l++) {
      const float width = (float)(pyr[l].w - 2 * EDGE_THRESHOLD + 6), height = (float)(pyr[l].h - 2 * EDGE_THRESHOLD + 6);
      if ((int)(width / 35) < 1 || (int)(height / 35) < 1) return -3;
    }
    cands.assign(nlevels, {});
    selected.assign(nlevels, {});
    for (int l = 0; l < nlevels; l++) {  // ComputeKeyPointsOctTree cpp:1026-1189
      const int minBX = EDGE_THRESHOLD - 3, minBY = minBX;
      const int maxBX = pyr[l].w - EDGE_THRESHOLD + 3, maxBY = pyr[l].h - EDGE_THRESHOLD + 3;
      levelCandidates(l, cands[l]);
      std::vector<Cand> sel = distributeOctTree(cands[l], minBX, maxBX, minBY, maxBY, quota[l]);
      if ((int)sel.size() > quota[l]) sel.resize(quota[l]);
      const int scaledPatch = (int)(PATCH_SIZE * scale[l]);
      for (const Cand& c : sel) {
        KP k;
        k.x = c.x + minBX; k.y = c.y + minBY;
        k.size = (float)scaledPatch; k.angle = -1; k.response = c.response;
        k.octave = l; k.class_id = -1;
        selected[l].push_back(k);
      }
    }
    for (int l = 0; l < nlevels; l++)
      for (KP& k : selected[l]) k.angle = icAngle(pyr[l], k.x, k.y, umax);
    int nk = 0;
    for (int l = 0; l < nlevels; l++) nk += (int)selected[l].size();
    kpsOut.resize(nk);
    descOut.assign((size_t)nk * 32, 0);
    blurred.assign(nlevels, Image());
    int mono = 0, stereo = nk - 1;
    for (int l = 0; l < nlevels; l++) {
      if (selected[l].empty()) continue;
      Image B; B.w = pyr[l].w; B.h = pyr[l].h; B.px.resize(pyr[l].px.size());
      gaussian7(pyr[l].px.data(), B.w, B.h, B.w, B.px.data(), B.w);
      for (const KP& k0 : selected[l]) {
        uint8_t d[32];
        orbDescriptor(B, k0, d);
        KP k = k0;
        if (l != 0) { k.x *= scale[l]; k.y *= scale[l]; }
        int dstIdx;
        if (k.x >= lap0 && k.x <= lap1) dstIdx = stereo--;
        else dstIdx = mono++;
        kpsOut[dstIdx] = k;
        memcpy(&descOut[(size_t)dstIdx * 32], d, 32);
      }
      if (keepBlurred) blurred[l] = std::move(B);
    }
    return mono;
  }
};

// ---------------------------------------------------------------------------------------------
// Frame grid (SlamTypes/Frame.cpp:70-99,163-206) and ORBmatcher::SearchForInitialization
// (Features/ORBmatcher.cpp:11-150), DBoW2::FORB::distance (Thirdparty/DBoW2/src/FORB.cpp:77-101)
// ---------------------------------------------------------------------------------------------
const int GRID_COLS = 64, GRID_ROWS = 48;  // Frame.hpp:15-16
const int TH_LOW = 50, HISTO_LENGTH = 30;  // ORBmatcher.cpp:5-7

int hamming256(const uint8_t* a, const uint8_t* b) {
  uint64_t pa[4], pb[4], ret = 0;
  memcpy(pa, a, 32); memcpy(pb, b, 32);
  for (int i = 0; i < 4; i++) {
    uint64_t v = pa[i] ^ pb[i];
    v = v - ((v >> 1) & (uint64_t) ~(uint64_t)0 / 3);
    v = (v & (uint64_t) ~(uint64_t)0 / 15 * 3) + ((v >> 2) & (uint64_t) ~(uint64_t)0 / 15 * 3);
    v = (v + (v >> 4)) & (uint64_t) ~(uint64_t)0 / 255 * 15;
    ret += (uint64_t)(v * ((uint64_t) ~(uint64_t)0 / 255)) >> (sizeof(uint64_t) - 1) * 8;
  }
  return (int)(double)ret;
}

struct Bounds { int32_t minX, maxX, minY, maxY; };

struct FrameGrid {
  const KP* keys; int N; Bounds b; float wInv, hInv;
  std::vector<size_t> grid[GRID_COLS][GRID_ROWS];
  bool posInGrid(const KP& kp, int& px, int& py) const {  // Frame.cpp:89-99
    px = (int)std::round((kp.x - b.minX) * wInv);
    py = (int)std::round((kp.y - b.minY) * hInv);
    return !(px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS);
  }
  FrameGrid(const KP* k, int n, Bounds bb) : keys(k), N(n), b(bb) {
    wInv = (float)GRID_COLS / (float)(b.maxX - b.minX);  // Frame.cpp:46-47
    hInv = (float)GRID_ROWS / (float)(b.maxY - b.minY);
    for (int i = 0; i < n; i++) { int px, py; if (posInGrid(k[i], px, py)) grid[px][py].push_back(i); }
  }
  std::vector<size_t> featuresInArea(float x, float y, float r, int minLevel, int maxLevel) const {  // Frame.cpp:163-206
    std::vector<size_t> out;
    int minCX = std::max(0, (int)std::floor((x - b.minX - r) * wInv));
    if (minCX >= GRID_COLS) return out;
    int maxCX = std::min(GRID_COLS - 1, (int)std::ceil((x - b.minX + r) * wInv));
    if (maxCX < 0) return out;
    int minCY = std::max(0, (int)std::floor((y - b.minY - r) * hInv));
    if (minCY >= GRID_ROWS) return out;
    int maxCY = std::min(GRID_ROWS - 1, (int)std::ceil((y - b.minY + r) * hInv));
    if (maxCY < 0) return out;
    const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = minCX; ix <= maxCX; ix++)
      for (int iy = minCY; iy <= maxCY; iy++)
        for (size_t j : grid[ix][iy]) {
          const KP& kp = keys[j];
          if (checkLevels && !(kp.octave >= minLevel && kp.octave <= maxLevel)) continue;
          const float dx = kp.x - x, dy = kp.y - y;
          if (std::fabs(dx) < r && std::fabs(dy) < r) out.push_back(j);
        }
    return out;
  }
};

void threeMaxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {  // ORBmatcher.cpp:152-183
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = (int)histo[i].size();
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

int searchForInitialization(const KP* k1, const uint8_t* d1, int n1, const KP* k2, const uint8_t* d2, int n2,
                            Bounds bnd, int windowSize, float nnratio, bool checkOri, int* matches12, int* stats) {
  int nmatches = 0;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = HISTO_LENGTH / 360.0f;
  std::vector<int> matchedDist(n2, INT_MAX), matches21(n2, -1);
  int badDist = 0, badRatio = 0, badOri = 0;
  FrameGrid F2(k2, n2, bnd);
  for (int i1 = 0; i1 < n1; i1++) {
    const KP kp1 = k1[i1];
    if (kp1.octave > 0) continue;
    std::vector<size_t> idx2s = F2.featuresInArea(kp1.x, kp1.y, (float)windowSize, kp1.octave, kp1.octave);
    if (idx2s.empty()) continue;
    int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
    for (size_t idx2 : idx2s) {
      int dist = hamming256(d1 + (size_t)i1 * 32, d2 + idx2 * 32);
      if (matchedDist[idx2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = (int)idx2; }
      else if (dist < bestDist2) bestDist2 = dist;
    }
    if (bestDist > TH_LOW) { badDist++; continue; }
    if ((float)bestDist > nnratio * (float)bestDist2) { badRatio++; continue; }
    if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; nmatches--; }
    matches12[i1] = bestIdx2;
    matches21[bestIdx2] = i1;
    matchedDist[bestIdx2] = bestDist;
    nmatches++;
    if (checkOri) {
      float rot = k1[i1].angle - k2[bestIdx2].angle;
      if (rot < 0.0) rot += 360.0f;
      int bin = (int)std::round(rot * factor);
      if (bin == HISTO_LENGTH) bin = 0;
      if (bin >= 0 && bin < HISTO_LENGTH) rotHist[bin].push_back(i1);  // assert upstream (compiled out)
    }
  }
  if (checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    threeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (int idx1 : rotHist[i]) { matches12[idx1] = -1; nmatches--; badOri++; }
  }
  if (stats) { stats[0] = badDist; stats[1] = badRatio; stats[2] = badOri; }
  return nmatches;
}

}  // namespace

// =============================================================================================
// C entry points for ctypes (tests / smoke / cpu_baseline only)
// =============================================================================================
extern "C" {

void* orbo_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
  if (nlevels < 1 || nfeatures < 1 || (scaleFactor == 1.0f && nlevels > 1)) return nullptr;
  return new Extractor(nfeatures, scaleFactor, nlevels, iniTh, minTh);
}
void orbo_destroy(void* h) { delete (Extractor*)h; }

void orbo_get_tables(void* h, float* scale, float* invScale, float* sigma2, float* invSigma2, int* quota, int* umax16) {
  Extractor* e = (Extractor*)h;
  for (int i = 0; i < e->nlevels; i++) {
    if (scale) scale[i] = e->scale[i];
    if (invScale) invScale[i] = e->invScale[i];
    if (sigma2) sigma2[i] = e->sigma2[i];
    if (invSigma2) invSigma2[i] = e->invSigma2[i];
    if (quota) quota[i] = e->quota[i];
  }
  if (umax16) for (int i = 0; i < 16; i++) umax16[i] = e->umax[i];
}

int orbo_extract(void* h, const uint8_t* img, int w, int hh, int stride, int lap0, int lap1, KP* kps, uint8_t* desc,
                 int capacity, int* n_out) {
  Extractor* e = (Extractor*)h;
  std::vector<KP> k; std::vector<uint8_t> d;
  int r = e->extract(img, w, hh, stride, lap0, lap1, k, d);
  if (r < 0) { if (n_out) *n_out = 0; return r; }
  if ((int)k.size() > capacity) return -5;
  if (!k.empty()) { memcpy(kps, k.data(), k.size() * sizeof(KP)); memcpy(desc, d.data(), d.size()); }
  if (n_out) *n_out = (int)k.size();
  return r;
}

void orbo_keep_blurred(void* h, int on) { ((Extractor*)h)->keepBlurred = on != 0; }
int orbo_level_size(void* h, int level, int* w, int* hh) {
  Extractor* e = (Extractor*)h;
  if (level < 0 || level >= (int)e->pyr.size()) return -2;
  *w = e->pyr[level].w; *hh = e->pyr[level].h; return 0;
}
int orbo_level_image(void* h, int level, uint8_t* dst) {
  Extractor* e = (Extractor*)h;
  if (level < 0 || level >= (int)e->pyr.size()) return -2;
  memcpy(dst, e->pyr[level].px.data(), e->pyr[level].px.size()); return 0;
}
int orbo_level_blurred(void* h, int level, uint8_t* dst) {
  Extractor* e = (Extractor*)h;
  if (level < 0 || level >= (int)e->blurred.size() || e->blurred[level].px.empty()) return -2;
  memcpy(dst, e->blurred[level].px.data(), e->blurred[level].px.size()); return 0;
}
// candidates of the last extract: (x, y, response) relative to (minBorderX, minBorderY)
int orbo_level_candidates(void* h, int level, float* xyr, int cap) {
  Extractor* e = (Extractor*)h;
  if (level < 0 || level >= (int)e->cands.size()) return -2;
  int n = (int)e->cands[level].size();
  for (int i = 0; i < n && i < cap; i++) { xyr[3 * i] = e->cands[level][i].x; xyr[3 * i + 1] = e->cands[level][i].y; xyr[3 * i + 2] = e->cands[level][i].response; }
  return n;
}
int orbo_level_selected(void* h, int level, KP* out, int cap) {
  Extractor* e = (Extractor*)h;
  if (level < 0 || level >= (int)e->selected.size()) return -2;
  int n = (int)e->selected[level].size();
  for (int i = 0; i < n && i < cap; i++) out[i] = e->selected[level][i];
  return n;
}

// stand-alone primitives
void orbo_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
  resizeLinear(src, sw, sh, sstride, dst, dw, dh, dstride);
}
void orbo_gaussian7(const uint8_t* src, int w, int h, int stride, uint8_t* dst, int dstride) { gaussian7(src, w, h, stride, dst, dstride); }
int orbo_fast(const uint8_t* img, int w, int h, int stride, int t, int nms, float* xyr, int cap) {
  std::vector<Cand> out; fastDetect(img, w, h, stride, t, nms != 0, out);
  for (int i = 0; i < (int)out.size() && i < cap; i++) { xyr[3 * i] = out[i].x; xyr[3 * i + 1] = out[i].y; xyr[3 * i + 2] = out[i].response; }
  return (int)out.size();
}
int orbo_fast_strength(const uint8_t* img, int w, int h, int stride, int x, int y) {
  if (x < 3 || y < 3 || x >= w - 3 || y >= h - 3) return 0;
  return fastStrength(img + (size_t)y * stride + x, stride);
}
int orbo_distribute(const float* xyr, int n, int minX, int maxX, int minY, int maxY, int N, float* out_xyr, int cap) {
  std::vector<Cand> c(n);
  for (int i = 0; i < n; i++) c[i] = {xyr[3 * i], xyr[3 * i + 1], xyr[3 * i + 2]};
  std::vector<Cand> r = distributeOctTree(c, minX, maxX, minY, maxY, N);
  for (int i = 0; i < (int)r.size() && i < cap; i++) { out_xyr[3 * i] = r[i].x; out_xyr[3 * i + 1] = r[i].y; out_xyr[3 * i + 2] = r[i].response; }
  return (int)r.size();
}
// ---------------------------------------------------------------------------------------------
// keypoint undistortion + image bounds (SlamTypes/Frame.cpp:101-161).
// cv::undistortPoints(src, dst, K, dist(k1,k2,p1,p2), R = I, P = K) restated from OpenCV 4.x:
// K and dist are CV_32F (Config/Settings.hpp:32,39) converted to double; 5 fixed iterations
// (TermCriteria(MAX_ITER, 5, 0.01)); all arithmetic in double, in this exact operation order; result cast to float.
// ---------------------------------------------------------------------------------------------
struct Camera { float fx, fy, cx, cy, k1, k2, p1, p2; };

static void undistortPointD(const Camera& c, float xin, float yin, float* xo, float* yo) {
  const double fx = c.fx, fy = c.fy, cx = c.cx, cy = c.cy;
  const double k0 = c.k1, k1 = c.k2, k2 = c.p1, k3 = c.p2;  // OpenCV's k[0..3]; k[4..13] = 0
  const double ifx = 1. / fx, ify = 1. / fy;
  double x = xin, y = yin;
  const double u = x, v = y;
  x = (x - cx) * ifx;
  y = (y - cy) * ify;
  const double x0 = x, y0 = y;
  for (int j = 0; j < 5; j++) {
    const double r2 = x * x + y * y;
    const double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((0. * r2 + k1) * r2 + k0) * r2);
    if (icdist < 0) {
      x = (u - cx) * ifx;
      y = (v - cy) * ify;
      break;
    }
    const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + 0. * r2 + 0. * r2 * r2;
    const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y + 0. * r2 + 0. * r2 * r2;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  // RR = P * R = K:  xx = fx*x + 0*y + cx, yy = 0*x + fy*y + cy, ww = 1 / (0*x + 0*y + 1)
  const double xx = fx * x + 0. * y + cx, yy = 0. * x + fy * y + cy, ww = 1. / (0. * x + 0. * y + 1.);
  *xo = (float)(xx * ww);
  *yo = (float)(yy * ww);
}

void orbo_undistort_keypoints(const KP* in, int n, const float* cam8, KP* out) {  // Frame::UndistortKeyPoints, Frame.cpp:136-161
  Camera c{cam8[0], cam8[1], cam8[2], cam8[3], cam8[4], cam8[5], cam8[6], cam8[7]};
  for (int i = 0; i < n; i++) {
    out[i] = in[i];
    if (c.k1 != 0.0f) undistortPointD(c, in[i].x, in[i].y, &out[i].x, &out[i].y);
  }
}

void orbo_image_bounds(const float* cam8, int cols, int rows, int32_t* b4) {  // Frame::ComputeImageBounds, Frame.cpp:101-134
  Camera c{cam8[0], cam8[1], cam8[2], cam8[3], cam8[4], cam8[5], cam8[6], cam8[7]};
  if (c.k1 != 0.0f) {
    float m[4][2] = {{0.f, 0.f}, {(float)cols, 0.f}, {0.f, (float)rows}, {(float)cols, (float)rows}};
    for (int i = 0; i < 4; i++) undistortPointD(c, m[i][0], m[i][1], &m[i][0], &m[i][1]);
    b4[0] = (int)std::min(m[0][0], m[2][0]);  // mnMinX (float -> static int)
    b4[1] = (int)std::max(m[1][0], m[3][0]);  // mnMaxX
    b4[2] = (int)std::min(m[0][1], m[1][1]);  // mnMinY
    b4[3] = (int)std::max(m[2][1], m[3][1]);  // mnMaxY
  } else {
    b4[0] = 0; b4[1] = cols; b4[2] = 0; b4[3] = rows;
  }
}

// Converter::toGray (Utils/Converter.cpp:5-19; demo_initialization.cpp:67-68) = copy for one channel,
// cv::cvtColor(COLOR_RGB2GRAY / COLOR_BGR2GRAY) for three, false otherwise.  8-bit cvtColor restated with the 14-bit
// coefficients of OpenCV 3.x / early 4.x (Y = (R*4899 + G*9617 + B*1868 + 8192) >> 14; newer 4.x
// releases use 15-bit coefficients -- unpinned, like the rest of the OpenCV arithmetic).  Returns 1 / 0 like the bool.
int orbo_to_gray(const uint8_t* src, int w, int h, int stride, int channels, int rgb, uint8_t* dst, int dstride) {
  if (channels == 1) {
    for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * dstride, src + (size_t)y * stride, (size_t)w);
    return 1;
  }
  if (channels != 3) return 0;
  // variant 0: 14-bit coefficients (OpenCV 3.x .. 4.0); variant 1: 15-bit (OpenCV >= 4.1)  [from-knowledge]
  const int cr = gGrayVariant ? 9798 : 4899, cg = gGrayVariant ? 19235 : 9617, cb = gGrayVariant ? 3735 : 1868, sh = gGrayVariant ? 15 : 14;
  const int c0 = rgb ? cr : cb, c1 = cg, c2 = rgb ? cb : cr;
  for (int y = 0; y < h; y++) {
    const uint8_t* s = src + (size_t)y * stride;
    uint8_t* d = dst + (size_t)y * dstride;
    for (int x = 0; x < w; x++) d[x] = (uint8_t)((s[3 * x] * c0 + s[3 * x + 1] * c1 + s[3 * x + 2] * c2 + (1 << (sh - 1))) >> sh);
  }
  return 1;
}

// the Initializer's scoring loops (Initialization/Initializer.cpp:268-438), restated operation by
// operation: f32 arithmetic evaluated left to right without contraction, the reciprocals `1.0 / x` in double and rounded
// to float, the score a sequential f32 sum in match order.  H / F are row-major 3x3 (Eigen's (row, col) accessor).
// Returns the score; inliers[i] = vbMatchesInliers[i].
float orbo_check_homography(const float* H21, const float* H12, const KP* k1, const KP* k2, const int* first, const int* second,
                            int N, float sigma, uint8_t* inliers) {
  const float h11 = H21[0], h12 = H21[1], h13 = H21[2], h21 = H21[3], h22 = H21[4], h23 = H21[5], h31 = H21[6], h32 = H21[7],
              h33 = H21[8];
  const float h11inv = H12[0], h12inv = H12[1], h13inv = H12[2], h21inv = H12[3], h22inv = H12[4], h23inv = H12[5],
              h31inv = H12[6], h32inv = H12[7], h33inv = H12[8];
  float score = 0;
  const float th = 5.991;
  const float invSigmaSquare = 1.0 / (sigma * sigma);
  for (int i = 0; i < N; i++) {
    bool bIn = true;
    const float u1 = k1[first[i]].x, v1 = k1[first[i]].y, u2 = k2[second[i]].x, v2 = k2[second[i]].y;
    const float w2in1inv = 1.0 / (h31inv * u2 + h32inv * v2 + h33inv);
    const float u2in1 = (h11inv * u2 + h12inv * v2 + h13inv) * w2in1inv;
    const float v2in1 = (h21inv * u2 + h22inv * v2 + h23inv) * w2in1inv;
    const float squareDist1 = (u1 - u2in1) * (u1 - u2in1) + (v1 - v2in1) * (v1 - v2in1);
    const float chiSquare1 = squareDist1 * invSigmaSquare;
    if (chiSquare1 > th) bIn = false;
    else score += th - chiSquare1;
    const float w1in2inv = 1.0 / (h31 * u1 + h32 * v1 + h33);
    const float u1in2 = (h11 * u1 + h12 * v1 + h13) * w1in2inv;
    const float v1in2 = (h21 * u1 + h22 * v1 + h23) * w1in2inv;
    const float squareDist2 = (u2 - u1in2) * (u2 - u1in2) + (v2 - v1in2) * (v2 - v1in2);
    const float chiSquare2 = squareDist2 * invSigmaSquare;
    if (chiSquare2 > th) bIn = false;
    else score += th - chiSquare2;
    inliers[i] = bIn ? 1 : 0;
  }
  return score;
}

float orbo_check_fundamental(const float* F21, const KP* k1, const KP* k2, const int* first, const int* second, int N, float sigma,
                             uint8_t* inliers) {
  const float f11 = F21[0], f12 = F21[1], f13 = F21[2], f21 = F21[3], f22 = F21[4], f23 = F21[5], f31 = F21[6], f32 = F21[7],
              f33 = F21[8];
  float score = 0;
  const float th = 3.841;
  const float thScore = 5.991;
  const float invSigmaSquare = 1.0 / (sigma * sigma);
  for (int i = 0; i < N; i++) {
    bool bIn = true;
    const float u1 = k1[first[i]].x, v1 = k1[first[i]].y, u2 = k2[second[i]].x, v2 = k2[second[i]].y;
    const float a2 = f11 * u1 + f12 * v1 + f13;
    const float b2 = f21 * u1 + f22 * v1 + f23;
    const float c2 = f31 * u1 + f32 * v1 + f33;
    const float num2 = a2 * u2 + b2 * v2 + c2;
    const float squareDist1 = num2 * num2 / (a2 * a2 + b2 * b2);
    const float chiSquare1 = squareDist1 * invSigmaSquare;
    if (chiSquare1 > th) bIn = false;
    else score += thScore - chiSquare1;
    const float a1 = f11 * u2 + f21 * v2 + f31;
    const float b1 = f12 * u2 + f22 * v2 + f32;
    const float c1 = f13 * u2 + f23 * v2 + f33;
    const float num1 = a1 * u1 + b1 * v1 + c1;
    const float squareDist2 = num1 * num1 / (a1 * a1 + b1 * b1);
    const float chiSquare2 = squareDist2 * invSigmaSquare;
    if (chiSquare2 > th) bIn = false;
    else score += thScore - chiSquare2;
    inliers[i] = bIn ? 1 : 0;
  }
  return score;
}

// the std::sort call of cpp:912 in isolation: (count, UL.x, id) triples ordered with compareNodes (cpp:684-696)
void orbo_std_sort_sized(int* triples, int n) {
  struct T3 { int c, u, id; };
  std::vector<T3> v(n);
  for (int i = 0; i < n; i++) v[i] = {triples[3 * i], triples[3 * i + 1], triples[3 * i + 2]};
  std::sort(v.begin(), v.end(), [](const T3& a, const T3& b) {
    if (a.c < b.c) return true;
    if (a.c > b.c) return false;
    return a.u < b.u;
  });
  for (int i = 0; i < n; i++) { triples[3 * i] = v[i].c; triples[3 * i + 1] = v[i].u; triples[3 * i + 2] = v[i].id; }
}
float orbo_fast_atan2(float y, float x) { return fastAtan2(y, x); }
float orbo_ic_angle(const uint8_t* img, int w, int h, float x, float y, int* m10, int* m01) {
  Image im; im.w = w; im.h = h; im.px.assign(img, img + (size_t)w * h);
  Extractor e(1000, 1.2f, 8, 20, 7);
  return icAngle(im, x, y, e.umax, m10, m01);
}
void orbo_descriptor(const uint8_t* blurred, int w, int h, float x, float y, float angle, uint8_t* desc32) {
  Image im; im.w = w; im.h = h; im.px.assign(blurred, blurred + (size_t)w * h);
  KP k{}; k.x = x; k.y = y; k.angle = angle;
  orbDescriptor(im, k, desc32);
}
void orbo_sincos_deg(float angle_deg, float* c, float* s) { descSinCos(angle_deg * factorPI, c, s); }
// Sweep of the f32 angles (degrees) with bit patterns [lo_bits, hi_bits]: out[0] / out[1] = angles whose restated glibc cosf / sinf
// (FMA form) differs from the HOST's own cosf / sinf, out[2] / out[3] = the same for the form without FMAs, out[4] / out[5] =
// angles whose host cosf / sinf differs from the double reading (float)cos((double)a).  nthreads host threads.
void orbo_libm_sweep(uint32_t lo_bits, uint32_t hi_bits, int nthreads, long long* out6) {
  if (nthreads < 1) nthreads = 1;
  std::vector<std::array<long long, 6>> part((size_t)nthreads);
  std::vector<std::thread> th;
  const unsigned long long total = (unsigned long long)hi_bits - lo_bits + 1;
  for (int t = 0; t < nthreads; t++)
    th.emplace_back([&, t]() {
      std::array<long long, 6> a{};
      const unsigned long long b0 = lo_bits + total * t / nthreads, b1 = lo_bits + total * (t + 1) / nthreads;
      for (unsigned long long u = b0; u < b1; u++) {
        const uint32_t uu = (uint32_t)u;
        float deg; std::memcpy(&deg, &uu, 4);
        volatile float r = deg * factorPI;  // (volatile: the product is rounded to f32 exactly here)
        const float rr = r;
        const float hc = cosf(rr), hs = sinf(rr);
        a[0] += hc != glibc_sincosf::cosF<true>(rr); a[1] += hs != glibc_sincosf::sinF<true>(rr);
        a[2] += hc != glibc_sincosf::cosF<false>(rr); a[3] += hs != glibc_sincosf::sinF<false>(rr);
        a[4] += hc != (float)std::cos((double)rr); a[5] += hs != (float)std::sin((double)rr);
      }
      part[(size_t)t] = a;
    });
  for (auto& x : th) x.join();
  for (int k = 0; k < 6; k++) { out6[k] = 0; for (auto& a : part) out6[k] += a[(size_t)k]; }
}
void orbo_sincos_deg_batch(const float* angle_deg, int n, float* c, float* s) {  // cpp:173-174 for many angles
  for (int i = 0; i < n; i++) orbo_sincos_deg(angle_deg[i], &c[i], &s[i]);
}
int orbo_hamming(const uint8_t* a, const uint8_t* b) { return hamming256(a, b); }
void orbo_pos_in_grid(const KP* k, int n, const int32_t* bounds4, int* px, int* py, int* ok) {
  Bounds b{bounds4[0], bounds4[1], bounds4[2], bounds4[3]};
  FrameGrid g(k, 0, b);
  for (int i = 0; i < n; i++) ok[i] = g.posInGrid(k[i], px[i], py[i]) ? 1 : 0;
}
int orbo_features_in_area(const KP* k, int n, const int32_t* bounds4, float x, float y, float r, int minLevel, int maxLevel, int* out, int cap) {
  Bounds b{bounds4[0], bounds4[1], bounds4[2], bounds4[3]};
  FrameGrid g(k, n, b);
  std::vector<size_t> v = g.featuresInArea(x, y, r, minLevel, maxLevel);
  for (int i = 0; i < (int)v.size() && i < cap; i++) out[i] = (int)v[i];
  return (int)v.size();
}
int orbo_match_init(const KP* k1, const uint8_t* d1, int n1, const KP* k2, const uint8_t* d2, int n2, const int32_t* bounds4,
                    int windowSize, float nnratio, int checkOri, int* matches12, int* stats3) {
  Bounds b{bounds4[0], bounds4[1], bounds4[2], bounds4[3]};
  return searchForInitialization(k1, d1, n1, k2, d2, n2, b, windowSize, nnratio, checkOri != 0, matches12, stats3);
}

// ---------------------------------------------------------------------------------------------
// Initializer::CheckRT (Initialization/Initializer.cpp:569-713): triangulate the inlier matches with one (R21, t21)
// hypothesis and count the points that lie in front of both cameras with a small reprojection error.
//
// cv::triangulatePoints and the cv::Mat arithmetic are OpenCV, absent from this tree; restated
// [from-knowledge], PARITY UNPINNED like every OpenCV primitive here:
//   * triangulatePoints: per point the 4x4 DLT matrix A (f64), rows x*P(2,:) - P(0,:), y*P(2,:) - P(1,:) for the two
//     views; the point is the right singular vector of the smallest singular value of A, stored as f32.  OpenCV runs
//     its one-sided Jacobi SVD; here: the same Hestenes scheme (column pairs in (i, j) order, rotation chosen as OpenCV's
//     JacobiSVDImpl_ does, at most 30 sweeps, eps = DBL_EPSILON * 10), no library call, so that the device can repeat
//     it operation for operation.  The singular vector's sign is arbitrary and cancels in x / w.
//   * gemm on CV_32F accumulates in double and rounds once: P2 = K [R|t], O2 = -R^T t, x3Dc2 = R x + t.
//   * Mat / scalar = Mat * (1.0 / scalar) with the factor rounded to f32 first; cv::norm and Mat::dot accumulate in
//     double.
// Quirks of the reference that are kept (they change which points count):
//   * the i-th TRIANGULATED point (i counts inliers only) is booked under match i, not under the i-th inlier's match
//     (vbTriGood[vMatches12[i].first], vP3D[vMatches12[i].first] with the compacted index, :643-704);
//   * the depth test in camera 2 looks at z / z of the normalised point (:665-670), i.e. it never fires for finite z;
//     invZ2 is then 1 / (z * (1 / z)).
// Outputs: vbTriGood[n1], vP3D[n1 * 3] (zeros where nothing was booked), *parallax (degrees); returns nGood.
// ---------------------------------------------------------------------------------------------
namespace {
// right singular vector of the smallest singular value of the 4x4 matrix A (row-major), one-sided Jacobi in f64
void smallestRightSingularVector4(const double Ain[16], double x[4]) {
  double At[4][4], V[4][4], W[4];  // At[i] = column i of A (OpenCV works on the transposed matrix), V[i] = row i of V^T
  for (int i = 0; i < 4; i++)
    for (int k = 0; k < 4; k++) { At[i][k] = Ain[k * 4 + i]; V[i][k] = i == k ? 1.0 : 0.0; }
  for (int i = 0; i < 4; i++) { double sd = 0; for (int k = 0; k < 4; k++) sd += At[i][k] * At[i][k]; W[i] = sd; }
  const double eps = 2.2204460492503131e-16 * 10;
  for (int iter = 0; iter < 30; iter++) {
    bool changed = false;
    for (int i = 0; i < 3; i++)
      for (int j = i + 1; j < 4; j++) {
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < 4; k++) p += At[i][k] * At[j][k];
        if (std::fabs(p) <= eps * std::sqrt(a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = std::sqrt(p * p + beta * beta);
        double c, sn;
        if (beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          sn = std::sqrt(delta / gamma);
          c = p / (gamma * sn * 2);
        } else {
          c = std::sqrt((gamma + beta) / (gamma * 2));
          sn = p / (gamma * c * 2);
        }
        a = b = 0;
        for (int k = 0; k < 4; k++) {
          const double t0 = c * At[i][k] + sn * At[j][k], t1 = -sn * At[i][k] + c * At[j][k];
          At[i][k] = t0; At[j][k] = t1;
          a += t0 * t0; b += t1 * t1;
        }
        W[i] = a; W[j] = b;
        changed = true;
        for (int k = 0; k < 4; k++) {
          const double t0 = c * V[i][k] + sn * V[j][k], t1 = -sn * V[i][k] + c * V[j][k];
          V[i][k] = t0; V[j][k] = t1;
        }
      }
    if (!changed) break;
  }
  int best = 0;  // smallest squared column norm; the first one among equals
  for (int i = 1; i < 4; i++)
    if (W[i] < W[best]) best = i;
  for (int k = 0; k < 4; k++) x[k] = V[best][k];
}
}  // namespace

int orbo_check_rt(const float* R21, const float* t21, const float* K, const KP* k1, int n1, const KP* k2, const int* first,
                  const int* second, int N, const uint8_t* inliers, float th2, uint8_t* vbTriGood, float* vP3D, float* parallax) {
  // 1. projection matrices (:577-589)
  float P1[12] = {K[0], K[1], K[2], 0, K[3], K[4], K[5], 0, K[6], K[7], K[8], 0};
  float Rt[12], P2[12];
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rt[r * 4 + c] = R21[r * 3 + c]; Rt[r * 4 + 3] = t21[r]; }
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) {
      double sd = 0;
      for (int k = 0; k < 3; k++) sd += (double)K[r * 3 + k] * (double)Rt[k * 4 + c];
      P2[r * 4 + c] = (float)sd;
    }
  float O2[3];  // -R21^T t21 (:592)
  for (int r = 0; r < 3; r++) {
    double sd = 0;
    for (int k = 0; k < 3; k++) sd += (double)R21[k * 3 + r] * (double)t21[k];
    O2[r] = (float)(-1.0 * sd);
  }
  for (int i = 0; i < n1; i++) { vbTriGood[i] = 0; vP3D[3 * i] = vP3D[3 * i + 1] = vP3D[3 * i + 2] = 0.f; }
  std::vector<float> vCos;
  int nGood = 0, ci = 0;  // ci = compacted index (the i of the reference's loop :640)
  for (int m = 0; m < N; m++) {
    if (!inliers[m]) continue;
    const int i = ci++;
    const float u1 = k1[first[m]].x, v1 = k1[first[m]].y, u2 = k2[second[m]].x, v2 = k2[second[m]].y;
    // cv::triangulatePoints (:627): DLT in f64, result stored as f32
    double A[16];
    for (int k = 0; k < 4; k++) {
      A[0 * 4 + k] = (double)u1 * (double)P1[2 * 4 + k] - (double)P1[0 * 4 + k];
      A[1 * 4 + k] = (double)v1 * (double)P1[2 * 4 + k] - (double)P1[1 * 4 + k];
      A[2 * 4 + k] = (double)u2 * (double)P2[2 * 4 + k] - (double)P2[0 * 4 + k];
      A[3 * 4 + k] = (double)v2 * (double)P2[2 * 4 + k] - (double)P2[1 * 4 + k];
    }
    double xd[4];
    smallestRightSingularVector4(A, xd);
    const float X[4] = {(float)xd[0], (float)xd[1], (float)xd[2], (float)xd[3]};
    const int book = first[i];  // quirk: match i, not match m
    const float invW = (float)(1.0 / (double)X[3]);
    const float xn[3] = {X[0] * invW, X[1] * invW, X[2] * invW};
    if (!std::isfinite(X[0]) || !std::isfinite(X[1]) || !std::isfinite(X[2])) { vbTriGood[book] = 0; continue; }
    if (X[0] == 0 && X[1] == 0 && X[2] == 0) { vbTriGood[book] = 0; continue; }
    // 3.1 parallax (:659-667)
    const float oc2[3] = {xn[0] - O2[0], xn[1] - O2[1], xn[2] - O2[2]};
    const float dist1 = (float)std::sqrt((double)xn[0] * xn[0] + (double)xn[1] * xn[1] + (double)xn[2] * xn[2]);
    const float dist2 = (float)std::sqrt((double)oc2[0] * oc2[0] + (double)oc2[1] * oc2[1] + (double)oc2[2] * oc2[2]);
    const double dot = (double)xn[0] * oc2[0] + (double)xn[1] * oc2[1] + (double)xn[2] * oc2[2];
    const float cosParallax = (float)(dot / (double)(dist1 * dist2));
    // 3.2 the point in camera 2 (:671-672)
    float xc2[3];
    for (int r = 0; r < 3; r++) {
      double sd = 0;
      for (int k = 0; k < 3; k++) sd += (double)R21[r * 3 + k] * (double)xn[k];
      xc2[r] = (float)(1.0 * sd + 1.0 * (double)t21[r]);
    }
    const float invZc2 = (float)(1.0 / (double)xc2[2]);
    const float xc2n[3] = {xc2[0] * invZc2, xc2[1] * invZc2, xc2[2] * invZc2};
    if (xn[2] <= 0 && (double)cosParallax < 0.99998) continue;
    if (xc2n[2] <= 0 && (double)cosParallax < 0.99998) continue;
    // 3.3 reprojection errors (:682-695)
    const float invZ1 = (float)(1.0 / (double)xn[2]);
    const float im1x = K[0] * xn[0] * invZ1 + K[2], im1y = K[4] * xn[1] * invZ1 + K[5];
    const float e1 = (im1x - u1) * (im1x - u1) + (im1y - v1) * (im1y - v1);
    const float invZ2 = (float)(1.0 / (double)xc2n[2]);
    const float im2x = K[0] * xc2n[0] * invZ2 + K[2], im2y = K[4] * xc2n[1] * invZ2 + K[5];
    const float e2 = (im2x - u2) * (im2x - u2) + (im2y - v2) * (im2y - v2);
    if (e1 > th2 || e2 > th2) continue;
    // 3.4 (:698-704)
    vCos.push_back(cosParallax);
    vP3D[3 * book] = xn[0]; vP3D[3 * book + 1] = xn[1]; vP3D[3 * book + 2] = xn[2];
    nGood++;
    if ((double)cosParallax < 0.99998) vbTriGood[book] = 1;
  }
  if (nGood > 0) {  // :708-712
    std::sort(vCos.begin(), vCos.end());
    const size_t idx = (size_t)std::min(50, (int)vCos.size() - 1);
    *parallax = (float)(std::acos((double)vCos[idx]) * 180 / 3.14159265358979323846);
  } else {
    *parallax = 0;
  }
  return nGood;
}

// ---------------------------------------------------------------------------------------------
// CPU baseline: extract(A) + extract(B) + SearchForInitialization(A,B) per pair.
// `nthreads` independent workers each loop over the same pairs for `reps` repetitions; returns
// wall seconds, frames processed in *frames_out.
// ---------------------------------------------------------------------------------------------
double orbo_bench_pairs(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh, const uint8_t* imgs, int nimgs,
                        int w, int h, int windowSize, float nnratio, int nthreads, int reps, long* frames_out, int* checksum_out) {
  std::atomic<long> frames{0};
  std::atomic<int> checksum{0};
  auto worker = [&](int tid) {
    Extractor e(nfeatures, scaleFactor, nlevels, iniTh, minTh);
    std::vector<KP> ka, kb; std::vector<uint8_t> da, db; std::vector<int> m;
    Bounds b{0, w, 0, h};
    for (int r = 0; r < reps; r++)
      for (int p = 0; p + 1 < nimgs; p += 2) {
        e.extract(imgs + (size_t)p * w * h, w, h, w, 0, 0, ka, da);
        e.extract(imgs + (size_t)(p + 1) * w * h, w, h, w, 0, 0, kb, db);
        m.resize(ka.size() + 1);
        int nm = searchForInitialization(ka.data(), da.data(), (int)ka.size(), kb.data(), db.data(), (int)kb.size(), b,
                                         windowSize, nnratio, true, m.data(), nullptr);
        frames += 2;
        if (tid == 0 && r == 0) checksum += nm;
      }
  };
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++) th.emplace_back(worker, t);
  for (auto& t : th) t.join();
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (frames_out) *frames_out = frames.load();
  if (checksum_out) *checksum_out = checksum.load();
  return sec;
}


// ---------------------------------------------------------------------------------------------
// CPU baseline, measurement protocol of SURVEY.md 8(d): every worker thread is pinned to one core and times, with
// steady_clock, `reps` repetitions of  extract(frame A) + extract(frame B) + SearchForInitialization(A, B)  on its own
// pair of frames (pair (tid mod npairs)), after `warmups` untimed repetitions.  All workers run concurrently (the
// "all host cores" figure); nthreads == 1 is "the repo's CPU path".  times[(tid * reps + r) * 3 + {0, 1, 2}] = seconds of
// the whole repetition, of the two extractions alone, and of the matching alone.  Returns 0.
// ---------------------------------------------------------------------------------------------
int orbo_bench_protocol(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh, const uint8_t* imgs, int nimgs,
                        int w, int h, int windowSize, float nnratio, int nthreads, int warmups, int reps, double* times) {
  const int npairs = nimgs / 2;
  if (npairs < 1 || nthreads < 1 || reps < 1 || !times) return -1;
  std::atomic<int> ready{0};
  std::atomic<bool> go{false};
  auto worker = [&](int tid) {
#ifdef __linux__
    {  // pin to the tid-th core this process may use
      cpu_set_t all, one;
      CPU_ZERO(&all);
      if (sched_getaffinity(0, sizeof all, &all) == 0) {
        int seen = 0, want = tid % std::max(CPU_COUNT(&all), 1);
        for (int c = 0; c < CPU_SETSIZE; c++)
          if (CPU_ISSET(c, &all) && seen++ == want) {
            CPU_ZERO(&one);
            CPU_SET(c, &one);
            (void)pthread_setaffinity_np(pthread_self(), sizeof one, &one);
            break;
          }
      }
    }
#endif
    Extractor e(nfeatures, scaleFactor, nlevels, iniTh, minTh);
    std::vector<KP> ka, kb; std::vector<uint8_t> da, db; std::vector<int> m;
    Bounds b{0, w, 0, h};
    const int p = (tid % npairs) * 2;
    auto once = [&](double* t3) {
      const auto t0 = std::chrono::steady_clock::now();
      e.extract(imgs + (size_t)p * w * h, w, h, w, 0, 0, ka, da);
      e.extract(imgs + (size_t)(p + 1) * w * h, w, h, w, 0, 0, kb, db);
      const auto t1 = std::chrono::steady_clock::now();
      m.resize(ka.size() + 1);
      (void)searchForInitialization(ka.data(), da.data(), (int)ka.size(), kb.data(), db.data(), (int)kb.size(), b, windowSize,
                                    nnratio, true, m.data(), nullptr);
      const auto t2 = std::chrono::steady_clock::now();
      if (t3) {
        t3[0] = std::chrono::duration<double>(t2 - t0).count();
        t3[1] = std::chrono::duration<double>(t1 - t0).count();
        t3[2] = std::chrono::duration<double>(t2 - t1).count();
      }
    };
    for (int r = 0; r < warmups; r++) once(nullptr);
    ready++;
    while (!go.load()) std::this_thread::yield();
    for (int r = 0; r < reps; r++) once(times + ((size_t)tid * reps + r) * 3);
  };
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++) th.emplace_back(worker, t);
  while (ready.load() < nthreads) std::this_thread::yield();
  go = true;
  for (auto& t : th) t.join();
  return 0;
}

void orbo_set_libm_variant(int libm_variant) { gLibmVariant = libm_variant ? 1 : 0; }
void orbo_set_opencv_variant(int gaussian_variant, int gray_variant) {
  gGaussVariant = gaussian_variant ? 1 : 0;
  gGrayVariant = gray_variant ? 1 : 0;
}

}  // extern "C"
