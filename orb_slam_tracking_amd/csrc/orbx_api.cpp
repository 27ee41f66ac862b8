// orbx_api.cpp — host side of liborbx: context, per-batch pipeline and the C ABI of include/orbx.h.
//
// Pipeline of one batch (a large batch is cut into two halves that run as independent chains on two HIP streams):
//   k_pyramid_bands (or k_resize_dw x (nlevels-1))  ->  k_fast (all cells of all levels of all frames)
//   -> k_octree_lds / k_octree_global (quadtree selection, one workgroup per frame x level) -> k_sel_compact
//   -> k_describe_patch (orientation + blur + descriptors) -> results stay in HBM (device API) or D2H (host API)
// and, for matching, k_match_jacobi (one workgroup per frame pair) -> k_match_wide_lists -> k_match_wide_resolve for the
// pairs beyond its tables.  No host round trip inside a batch: the host only issues the launches and reads two error flags + the per-frame
// counts at the end.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "orbx_device.h"
#include "orbx_knobs.h"

namespace orbx {

// the diagnostic knobs (orbx_knobs.h), all unset: every launch choice is the library's own
std::atomic<long long> g_knob[KNOB_COUNT] = {
#define ORBX_KNOB_INIT(e, n) {KNOB_UNSET},
    ORBX_KNOB_LIST(ORBX_KNOB_INIT)
#undef ORBX_KNOB_INIT
};

hipError_t launch_resize(hipStream_t st, int nFrames, const uint8_t* src, long long srcFrameStride, int sw, int sh, int sstride,
                         uint8_t* dst, long long dstFrameStride, int dw, int dh, int dstride, const ResizeTab* xtab,
                         const ResizeTab* ytab, int dwordPath, int wideFrames);
hipError_t launch_fast(hipStream_t st, int nFrames, const uint8_t* img0, long long img0FrameStride, int img0Aligned,
                       const uint8_t* pyr, const Geom& g, uint32_t* cand, int* cellCount, const FastCell* cells, int waveOk,
                       int* usedWave);
hipError_t launch_pyramid_tiles(hipStream_t st, int nFrames, const uint8_t* img0, long long img0FrameStride, uint8_t* pyr,
                                const Geom& g, const PyrTileRect* rects, const PyrTileTap* taps, int nTiles, int buf0Bytes,
                                int bufBytes);
hipError_t launch_pyramid_bands(hipStream_t st, int nFrames, const uint8_t* img0, long long img0FrameStride, uint8_t* pyr,
                                const Geom& g, const ResizeTab* tab, const PyrBands& pb);
hipError_t launch_describe_patch(hipStream_t st, int nFrames, int maxSel, const uint8_t* img0, long long img0FrameStride,
                                 int img0Aligned, const uint8_t* pyr, const Geom& g, const SelKp* sel, const int* nsel,
                                 orbx_keypoint* kps, uint8_t* desc, int capacity, int gaussVariant, int libmFloat,
                                 const DescStage* staged);
hipError_t launch_match(hipStream_t st, int nPairs, const int* dFirst, const int* dSecond, const orbx_keypoint* kps,
                        const uint8_t* desc, const int* nkp, int capacity, orbx_bounds b, int window, float nnratio, int checkOri,
                        int* matches12, int* nmatches, int* stats, int* scratch, int pair0, int wideMode, int* hostWide,
                        unsigned int* diag);
hipError_t launch_octree(hipStream_t st, int nFrames, const uint32_t* cand, const int* cellCount, const OctLaunch& P,
                         SelKp* selStage, int* nselLevel, uint8_t* scratch, int* maxN, const int* hintL, int force, int* usedInstance);
hipError_t launch_sel_compact(hipStream_t st, int nFrames, const SelKp* selStage, const int* nselLevel, const OctLaunch& P,
                              SelKp* sel, int* nsel, int* nselUser, int* hostNsel, int selCap, int* hostErr, int* maxN,
                              int* hostMaxN);
size_t octScratchBytes(int nMax, int qMax);
hipError_t launch_to_gray(hipStream_t st, int nFrames, const uint8_t* src, long long srcFrameStride, int sstride, int w, int h,
                          int channels, int rgb, uint8_t* dst, long long dstFrameStride, int dstride, int grayVariant);
hipError_t launch_check_model(hipStream_t st, int nModels, const ScoreArgs& a);
hipError_t launch_copy_out(hipStream_t st, const CopyOut& c, int nseg, int maxRows);
hipError_t launch_check_rt(hipStream_t st, int nModels, const CheckRtArgs& a);
hipError_t launch_debug_sincos(hipStream_t st, const float* angle, int n, float* c, float* s, int libmFloat);
hipError_t launch_undistort(hipStream_t st, int nFrames, const orbx_keypoint* in, const int* nkp, int capacity, const CamD& c,
                            orbx_keypoint* out);

namespace {

inline int cvRoundF(float v) { return (int)lrintf(v); }  // round half to even
inline int cvRoundD(double v) { return (int)lrint(v); }
inline int alignUp(int v, int a) { return (v + a - 1) / a * a; }

}  // namespace
}  // namespace orbx

using namespace orbx;

// k_pyramid_bands' own tables (PyrXGroup per group of 4 output pixels, PyrYRow per output row), appended to the resize
// tables of the geometry; `ok` = every level meets the kernel's preconditions
struct PyrTabInfo {
  int32_t xoff[ORBX_MAX_LEVELS], yoff[ORBX_MAX_LEVELS];  // uint4 units from the table base
  int ok, dual2;
};

struct orbx_ctx {
  orbx_params p{};
  int device = 0;
  hipStream_t st = nullptr;
  bool ownStream = false;
  int maxW = 0, maxH = 0, maxB = 0;
  std::vector<float> scale, invScale, sigma2, invSigma2;
  std::vector<int> quota;
  int umax[16]{};
  int gaussVariant = 0, grayVariant = 0;  // orbx_set_opencv_variant
  int libmVariant = ORBX_LIBM_DEFAULT;    // orbx_set_libm_variant
  int selCap = 0;  // sum of the per-level quotas

  // geometry of the current frame size
  int curW = 0, curH = 0, curStride0 = 0;
  Geom g{};
  std::vector<ResizeTab> hTab;

  // device buffers
  uint8_t* dPyr = nullptr;
  size_t pyrBytes = 0;
  uint32_t* dCand = nullptr;
  size_t candEntries = 0;
  int* dCandCount = nullptr;   // only its tail is used: the two error flags (dOverflow)
  int* dCellCount = nullptr;   // [frame][cell of all levels]: candidates in the cell's segment
  int* dMaxN = nullptr;        // [frame][level] candidate count of the unit (k_octree_lds); reduced and reset by k_sel_compact
  int* hMaxN = nullptr;        // pinned mirror, written by k_sel_compact
  int* hMaxNDev = nullptr;
  int candHintL[ORBX_MAX_LEVELS]{};  // per level: largest candidate count of a unit in the previous batch of this geometry (0 = unknown)
  int maxSlotsUsed = 0;
  size_t cellCountEntries = 0;
  int* dOverflow = nullptr;
  ResizeTab* dTab = nullptr;
  size_t tabEntries = 0;
  PyrTabInfo pyrInfo{};   // k_pyramid_bands' tables inside dTab, for the current geometry
  std::vector<FastCell> hCells;  // k_fast_wave's per-cell records of the current geometry (buildFastCells)
  FastCell* dCells = nullptr;
  size_t cellEntries = 0;
  int fastWaveOk = 0;            // k_fast_wave usable for the current geometry: 0 no, 1 tiles of 16 dwords x 64 rows, 2 of 12 dwords
  SelKp* dSel = nullptr;
  int* dNsel = nullptr;
  // quadtree selection stage (device)
  OctLaunch oct{};
  int maxQuota = 0;
  SelKp* dSelStage = nullptr;
  int* dNselLevel = nullptr;
  uint8_t* dOctScratch = nullptr;
  PyrTileRect* dPyrTiles = nullptr;  // k_pyramid_tiles' rectangles of the current geometry (buildPyrTiles); nPyrTiles == 0: not usable
  PyrTileTap* dPyrTaps = nullptr;    // ... and the tiles' tap blobs (ORBX_PYR_TILE_TAPS entries per tile)
  size_t pyrTileCap = 0;             // tiles the two buffers hold
  std::vector<PyrTileRect> hPyrTiles;
  std::vector<PyrTileTap> hPyrTaps;
  int nPyrTiles = 0, pyrTileBuf0 = 0, pyrTileBuf = 0;
  uint32_t* dOctTab = nullptr;     // path-code tables of the current geometry's levels (OctLaunch::codeTab)
  size_t octTabEntries = 0;
  std::vector<uint32_t> hOctTab;
  size_t octScratchBytes = 0;
  int* hFlags = nullptr;  // pinned [2], written by k_sel_compact through hFlagsDev: selection error of the batch in flight with that parity
  int* hFlagsDev = nullptr;
  int* hNselDev = nullptr;  // device view of hNsel: k_sel_compact stores the per-frame counts straight to the host
  // The wide matcher kernels are issued with a batch only while batches need them (k_match_jacobi raises hWide[parity]
  // when it hands a pair on); a batch that needed them without having them gets them at its wait (late, prep included).
  int* hWide = nullptr;     // pinned [2]
  int* hWideDev = nullptr;
  bool wideExpected = false;  // a batch that needs the wide kernels without having them gets them at its wait, and its successors with them
  int wideIdle = 0;         // consecutive batches that had the wide kernels and did not need them
  bool wideLaunched[2]{};
  bool eventOrdered = false;  // orbx_order_before has been used: a consumer may read a batch's outputs without a host-side wait, so
                              // the wide kernels travel with every batch (nothing may be left to the wait)
  int lastLaunch[8]{};        // orbx_debug_last_launch: how the last batch was issued
  struct LateMatch {
    bool valid = false;
    int nPairs = 0, capacity = 0, window = 0, checkOri = 0;
    float nnratio = 0;
    orbx_bounds b{};
    const orbx_keypoint* dKps = nullptr;
    const uint8_t* dDesc = nullptr;
    const int* dN = nullptr;
    int32_t* dMatches12 = nullptr;
    int32_t* dNmatches = nullptr;
    int32_t* dStats = nullptr;
  } late[2];
  std::vector<int32_t> lastPairs;  // the pair list dPairs holds (first[], second[]): an unchanged list is not copied again
  uint8_t* dIn = nullptr;
  size_t inBytes = 0;
  orbx_keypoint* dKps = nullptr;
  uint8_t* dDesc = nullptr;
  // pinned host mirrors
  int* hNsel = nullptr;
  // results of a call on a few host frames are written by the descriptor kernel straight into mapped page-locked memory
  // (no copy commands, no second synchronisation behind them): orbx_extract_batch with at most pinFrames frames
  orbx_keypoint* hKpsPin = nullptr;
  orbx_keypoint* hKpsPinDev = nullptr;
  uint8_t* hDescPin = nullptr;
  uint8_t* hDescPinDev = nullptr;
  int pinFrames = 0;
  // matcher
  int* dMatchScratch = nullptr;
  unsigned int* dMatchDiag = nullptr;  // orbx_debug_match_counters: [0] blocks of 256 queries listed by k_match_bf_mfma since orbx_create
  size_t matchScratchInts = 0;
  int* dPairs = nullptr;
  size_t pairsCap = 0;
  // host-array entry points (orbx_match_init, orbx_undistort_keypoints): ONE device block [2 mCap keypoints | 2 mCap
  // descriptors | n[2]] with a page-locked mirror, so that a pair goes up with one copy command, and page-locked mapped
  // result words the matcher kernels write straight into (no copy commands back)
  uint8_t* dMblk = nullptr;
  uint8_t* hMblk = nullptr;
  size_t mBlkBytes = 0, mDescOff = 0, mCntOff = 0;
  int* hMo = nullptr;     // [mCap + 8]: n/a[2], nmatches, stats[3], pad[2], matches12[mCap]
  int* hMoDev = nullptr;
  orbx_keypoint* dMk = nullptr;  // = dMblk
  uint8_t* dMd = nullptr;        // = dMblk + mDescOff
  int* dMi = nullptr;  // n[2] + matches12[cap] + nmatches + stats[3]  (device ints of orbx_undistort_keypoints)
  size_t mCap = 0;
  uint8_t* dScore = nullptr;  // staging of orbx_check_homography / _fundamental
  size_t scoreBytes = 0;
  uint8_t* dColor = nullptr;  // staging of orbx_to_gray (host API): colour frame followed by its gray image
  size_t colorBytes = 0;

  // last extract call (for orbx_download_pyramid / debug hooks)
  const uint8_t* lastImg0 = nullptr;
  long long lastFrameStride0 = 0;
  int lastB = 0;

  hipStream_t st2 = nullptr;  // second stream: half-batches overlap (extractCore)
  hipEvent_t evFork = nullptr, evJoin = nullptr;
  hipEvent_t evOrder = nullptr;  // orbx_order_after / orbx_order_before: the caller's stream <-> the context's streams

  // profiling
  unsigned profMask = 0;  // stages whose launches are bracketed by events (bit = ORBX_STAGE_*)
  // [parity of the batch][stream slot][stage][begin/end]: two batches may be in flight (orbx_*_async)
  hipEvent_t ev[2][2][ORBX_STAGE_COUNT][2]{};
  bool used[2][2][ORBX_STAGE_COUNT]{};
  hipEvent_t evDone[2]{};   // end of the batch with that parity on st
  hipEvent_t evDone2[2]{};  // ... and on st2 (only when the batch used it: done2Used)
  bool done2Used[2]{};
  // orbx_set_pipeline_depth: the stream-ordered batches go, whole, to the next of `depth` LANES -- child contexts with their own
  // buffers and stream -- instead of as two half batches onto this context's two streams
  std::vector<orbx_ctx*> lanes;
  unsigned laneIssue = 0, laneDone = 0;  // batches issued to / waited for on the lanes (round-robin in issue order)
  bool noSplit = false;                  // (a lane) never cuts a batch into halves
  // orbx_extract_match_batch_host_async: the batch's frames go up into dIn and its results come down from these device
  // buffers behind its kernels, all stream-ordered; evOut = the copies back have landed (waited for by waitOldest)
  uint8_t* dPipeOut = nullptr;           // [kps | desc | n | matches12 | nmatches | stats]
  size_t pipeOutBytes = 0;
  hipEvent_t evOut[2]{};
  bool outUsed[2]{};
  bool hostInput = false;                // (while such a batch is issued) the second stream forks behind the upload
  unsigned seqIssue = 0;   // batches issued so far
  int pending = 0;         // issued and not yet waited for (0 .. 2)
  int parity = 0;          // of the batch being issued
  double ms[ORBX_STAGE_COUNT]{};
  int64_t launches[ORBX_STAGE_COUNT]{};

  std::string err;
};

namespace {

#define HIPCHK(expr)                                                                                     \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) {                                                                              \
      char buf_[512];                                                                                    \
      snprintf(buf_, sizeof buf_, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
      ctx->err = buf_;                                                                                   \
      return ORBX_E_HIP;                                                                                 \
    }                                                                                                    \
  } while (0)

// ORBextractor ctor arithmetic, Features/ORBextractor.cpp:508-594
void computeTables(orbx_ctx* c) {
  const int nl = c->p.nlevels;
  const double scaleFactor = (double)c->p.scale_factor;  // the member is a double (hpp:142)
  c->scale.assign(nl, 1.f);
  c->sigma2.assign(nl, 1.f);
  c->invScale.assign(nl, 1.f);
  c->invSigma2.assign(nl, 1.f);
  for (int i = 1; i < nl; i++) {
    c->scale[i] = (float)(c->scale[i - 1] * scaleFactor);
    c->sigma2[i] = c->scale[i] * c->scale[i];
  }
  for (int i = 0; i < nl; i++) {
    c->invScale[i] = 1.0f / c->scale[i];
    c->invSigma2[i] = 1.0f / c->sigma2[i];
  }
  c->quota.assign(nl, 0);
  const float factor = (float)(1.0f / scaleFactor);
  // cpp:536 pow(float, float): ::pow(double, double) or, with the float overloads visible, powf (orbx_set_libm_variant)
  const float powv = c->libmVariant ? powf(factor, (float)nl) : (float)std::pow((double)factor, (double)(float)nl);
  float desired = c->p.nfeatures * (1 - factor) / (1 - powv);
  int sum = 0;
  for (int l = 0; l < nl - 1; l++) {
    c->quota[l] = cvRoundF(desired);
    sum += c->quota[l];
    desired *= factor;
  }
  c->quota[nl - 1] = std::max(c->p.nfeatures - sum, 0);
  c->selCap = 0;
  for (int q : c->quota) c->selCap += q;
  const int HALF = 15;
  const int vmax = (int)std::floor(HALF * std::sqrt(2.f) / 2 + 1);
  const int vmin = (int)std::ceil(HALF * std::sqrt(2.f) / 2);
  const double hp2 = HALF * HALF;
  for (int v = 0; v <= HALF; v++) c->umax[v] = 0;
  for (int v = 0; v <= vmax; ++v) c->umax[v] = cvRoundD(std::sqrt(hp2 - v * v));
  for (int v = HALF, v0 = 0; v >= vmin; --v) {
    while (c->umax[v0] == c->umax[v0 + 1]) ++v0;
    c->umax[v] = v0;
    ++v0;
  }
}

void levelSize(const orbx_ctx* c, int w, int h, int l, int* lw, int* lh) {  // cpp:1662-1663
  *lw = cvRoundF(w * c->invScale[l]);
  *lh = cvRoundF(h * c->invScale[l]);
}

// fills c->g for a w x h frame whose level 0 has row stride `stride0`; returns 0 or an error code
void appendPyrTables(const Geom& g, std::vector<ResizeTab>* tab, PyrTabInfo* info) {
  static_assert(sizeof(ResizeTab) == 8 && sizeof(PyrXGroup) == 8 * sizeof(ResizeTab) && sizeof(PyrYRow) == 2 * sizeof(ResizeTab), "table units");
  *info = PyrTabInfo{};
  info->ok = g.nlevels > 1;
  for (int l = 1; l < g.nlevels; l++) {
    const LevelGeom& S = g.L[l - 1];
    const LevelGeom& D = g.L[l];
    const int sw = S.w, sh = S.h, ng = (D.w + 3) / 4;
    if (sw < 8 || D.h >= 32768 || D.w > 4096) info->ok = 0;  // (512 thread-columns of two 4-pixel groups: k_pyramid_bands' loadCols)
    if (tab->size() & 1) tab->push_back(ResizeTab{0, 0});
    info->xoff[l] = (int32_t)(tab->size() / 2);
    const int lim = (sw - 1) & ~3;  // last dword that holds a pixel of the row: never read beyond it
    for (int gx = 0; gx < ng; gx++) {
      PyrXGroup x{};
      const ResizeTab* e = tab->data() + D.xtabOff + 4 * gx;  // padded to a multiple of 4 entries (the pad repeats the last pixel)
      const int base = e[0].ofs & ~3;
      x.o[0] = (uint32_t)base;
      x.o[1] = (uint32_t)std::min(base + 4, lim);
      x.o[2] = (uint32_t)std::min(base + 8, lim);
      const uint32_t zero = 0x0c0c0c0cu;
      for (int i = 0; i < 4; i++) {
        const int k = e[i].ofs - base, c0 = e[i].coef & 0xffff, c1 = (int)((uint32_t)e[i].coef >> 16);
        // the right tap is needed only with a weight (cv::resize zeroes the fraction at the last column)
        const int k1 = c1 ? k + 1 : k;
        if (k < 0 || k1 > 11 || c0 + c1 > 2048) { info->ok = 0; continue; }
        const bool second = k1 > 7;            // bytes 4..11 = dwords (1, 2)
        if (second && k < 4) { info->ok = 0; continue; }
        const uint32_t sel = 0x0c000c00u | (uint32_t)(second ? k - 4 : k) | ((uint32_t)(second ? k1 - 4 : k1) << 16);
        if (i < 2) {
          if (second) info->ok = 0;            // pixels 0 and 1 have one selector (dwords 0, 1): scale <= 2 keeps them there
          x.sel[i] = sel;
        } else if (i == 2) {
          x.sel[2] = second ? zero : sel;
          x.sel[3] = second ? sel : zero;
          if (second) info->dual2 = 1;
        } else {
          x.sel[4] = second ? zero : sel;
          x.sel[5] = second ? sel : zero;
        }
        x.cf[i] = ((uint32_t)c0 << 4) | (((uint32_t)c1 << 4) << 16);
      }
      ResizeTab raw[8];
      memcpy(raw, &x, sizeof(x));
      tab->insert(tab->end(), raw, raw + 8);
    }
    info->yoff[l] = (int32_t)(tab->size() / 2);
    for (int dy = 0; dy < D.h; dy++) {
      const ResizeTab t = (*tab)[D.ytabOff + dy];
      const int sy0 = std::min(std::max(t.ofs, 0), sh - 1), sy1 = std::min(std::max(t.ofs + 1, 0), sh - 1);
      const int c0 = t.coef & 0xffff, c1 = (int)((uint32_t)t.coef >> 16);
      if (c0 + c1 > 2048) info->ok = 0;
      const PyrYRow y{(uint32_t)sy0 * (uint32_t)S.stride, (uint32_t)sy1 * (uint32_t)S.stride, (uint32_t)c0 << 8, (uint32_t)c1 << 8};
      ResizeTab raw[2];
      memcpy(raw, &y, sizeof(y));
      tab->insert(tab->end(), raw, raw + 2);
    }
  }
}

int buildGeometry(orbx_ctx* c, int w, int h, int stride0, Geom* out, std::vector<ResizeTab>* tab, PyrTabInfo* pyrInfo = nullptr) {
  Geom g{};
  g.nlevels = c->p.nlevels;
  g.iniTh = std::min(std::max(c->p.ini_th_fast, 0), 255);
  g.minTh = std::min(std::max(c->p.min_th_fast, 0), 255);
  g.selCap = c->selCap;
  int cellBase = 0;
  int64_t pyrOff = 0, candOff = 0;
  int tabOff = 0;
  if (tab) tab->clear();
  int pw = 0, ph = 0;
  for (int l = 0; l < g.nlevels; l++) {
    LevelGeom& L = g.L[l];
    levelSize(c, w, h, l, &L.w, &L.h);
    if (L.w > ORBX_MAX_FRAME_DIM || L.h > ORBX_MAX_FRAME_DIM) {  // (documented deviation, orbx.h: 12-bit candidate coordinates)
      c->err = "frame larger than ORBX_MAX_FRAME_DIM (4096) pixels in width or height";
      return ORBX_E_BADARG;
    }
    L.maxBX = L.w - ORBX_EDGE + 3;
    L.maxBY = L.h - ORBX_EDGE + 3;
    const float width = (float)(L.maxBX - ORBX_MIN_BORDER), height = (float)(L.maxBY - ORBX_MIN_BORDER);
    if (width < 35.f || height < 35.f) return ORBX_E_TOOSMALL;
    if ((int)std::round(width / height) < 1) return ORBX_E_TOOSMALL;  // nIni == 0 is UB upstream (cpp:706-709)
    L.nCols = (int)(width / 35.f);
    L.nRows = (int)(height / 35.f);
    L.wCell = (int)std::ceil(width / L.nCols);
    L.hCell = (int)std::ceil(height / L.nRows);
    L.cellBase = cellBase;
    L.colsInv24 = L.nCols > 0 ? (uint32_t)(((1u << 24) + (uint32_t)L.nCols - 1u) / (uint32_t)L.nCols) : 0u;
    // exact for index < 2^24 / nCols and index * colsInv24 < 2^32, i.e. nRows < 256: far beyond any frame the context accepts
    cellBase += L.nCols * L.nRows;
    // worst-case number of NMS survivors (no two are 8-neighbours)
    int cap = 0;
    for (int i = 0; i < L.nRows; i++) {
      const int iniY = ORBX_MIN_BORDER + i * L.hCell;
      if (iniY >= L.maxBY - 3) continue;
      const int ch = std::min(iniY + L.hCell + 6, L.maxBY) - iniY;
      for (int j = 0; j < L.nCols; j++) {
        const int iniX = ORBX_MIN_BORDER + j * L.wCell;
        if (iniX >= L.maxBX - 6) continue;
        const int cw = std::min(iniX + L.wCell + 6, L.maxBX) - iniX;
        if (cw < 7 || ch < 7) continue;
        cap += ((cw - 6 + 1) / 2) * ((ch - 6 + 1) / 2);
      }
    }
    // k_fast writes the survivors of a cell into the cell's own fixed segment (no atomics); the selection stage gathers
    L.candMax = std::max(cap, 1);
    L.segCap = alignUp(std::max(((L.wCell + 1) / 2) * ((L.hCell + 1) / 2), 1), 4);  // 16-byte aligned segments
    L.candCap = L.nCols * L.nRows * L.segCap;
    L.candOff = candOff;
    candOff += (int64_t)L.candCap * c->maxB;
    L.quota = c->quota[l];
    L.scale = c->scale[l];
    L.patchSize = (int)(31 * c->scale[l]);  // cpp:1165
    if (l == 0) {
      L.stride = stride0;
      L.imgOff = 0;
      L.frameStride = 0;
      L.xtabOff = L.ytabOff = 0;
    } else {
      L.stride = alignUp(L.w, 64);
      L.imgOff = pyrOff;
      L.frameStride = (int64_t)L.stride * L.h;
      pyrOff += L.frameStride * c->maxB;
      if (tab) {
        // cv::resize INTER_LINEAR coefficient tables
        const int sw = pw, sh = ph, dw = L.w, dh = L.h;
        const double scale_x = 1.0 / ((double)dw / sw), scale_y = 1.0 / ((double)dh / sh);
        L.xtabOff = tabOff;
        const int dwPad = alignUp(dw, 4);
        for (int dx = 0; dx < dwPad; dx++) {
          const int d = std::min(dx, dw - 1);
          float fx = (float)((d + 0.5) * scale_x - 0.5);
          int sx = (int)std::floor(fx);
          fx -= sx;
          if (sx < 0) { fx = 0; sx = 0; }
          if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
          const int c0 = cvRoundF((1.f - fx) * 2048), c1 = cvRoundF(fx * 2048);
          tab->push_back(ResizeTab{sx, (c0 & 0xffff) | (c1 << 16)});
        }
        int span = 0;
        for (int dx = 0; dx + 3 < dwPad; dx += 4)
          span = std::max(span, (*tab)[L.xtabOff + dx + 3].ofs - (*tab)[L.xtabOff + dx].ofs);
        L.resizeSpanOk = span <= 7 ? 1 : 0;
        tabOff += dwPad;
        L.ytabOff = tabOff;
        for (int dy = 0; dy < dh; dy++) {
          float fy = (float)((dy + 0.5) * scale_y - 0.5);
          int sy = (int)std::floor(fy);
          fy -= sy;
          const int c0 = cvRoundF((1.f - fy) * 2048), c1 = cvRoundF(fy * 2048);
          tab->push_back(ResizeTab{sy, (c0 & 0xffff) | (c1 << 16)});
        }
        tabOff += dh;
      }
    }
    pw = L.w;
    ph = L.h;
  }
  g.nCellsTotal = cellBase;
  if (tab) {
    PyrTabInfo info;
    appendPyrTables(g, tab, &info);
    if (pyrInfo) *pyrInfo = info;
  }
  *out = g;
  return ORBX_OK;
}

// k_fast_wave's cell records (FastCell): the cell rectangles of ComputeKeyPointsOctTree (cpp:1078-1103) for every level,
// with everything a wave would otherwise derive from its cell index.  Returns whether every cell fits the kernel's tile.
// 0 = some cell does not fit (k_fast runs), 1 = tiles of 16 dwords x 64 rows, 2 = every cell image is at most 12 dwords wide.
int buildFastCells(const Geom& g, std::vector<FastCell>* out) {
  out->assign((size_t)g.nCellsTotal, FastCell{});
  bool ok = true, narrow = true;
  for (int l = 0; l < g.nlevels; l++) {
    const LevelGeom& L = g.L[l];
    for (int ci = 0; ci < L.nRows; ci++)
      for (int cj = 0; cj < L.nCols; cj++) {
        const int local = ci * L.nCols + cj;
        FastCell& c = (*out)[(size_t)L.cellBase + local];
        c.xoff_level = (uint32_t)l << 16;
        c.stride = (uint32_t)L.stride;
        c.segOff = (uint32_t)local * (uint32_t)L.segCap;
        c.segCap = (uint32_t)L.segCap;
        const int iniY = ORBX_MIN_BORDER + ci * L.hCell, iniX = ORBX_MIN_BORDER + cj * L.wCell;
        if (iniY >= L.maxBY - 3 || iniX >= L.maxBX - 6) continue;  // cpp:1088, 1101 (rows stay 0)
        const int maxY = std::min(iniY + L.hCell + 6, L.maxBY), maxX = std::min(iniX + L.wCell + 6, L.maxBX);
        const int cw = maxX - iniX, ch = maxY - iniY;
        if (cw < 7 || ch < 7) continue;  // cv::FAST finds nothing in an image this small
        const int ax0 = iniX & ~3, xoff = iniX - ax0, nw = (maxX - ax0 + 3) >> 2;
        if (nw > 16 || ch > 64) ok = false;
        if (nw > 12) narrow = false;
        c.imgOff = (uint32_t)(iniY * L.stride + ax0);
        c.nw_ch = (uint32_t)nw | ((uint32_t)ch << 16);
        c.iw_ih = (uint32_t)(cw - 6) | ((uint32_t)(ch - 6) << 16);
        c.xoff_level |= (uint32_t)xoff;
        c.ox_oy = ((uint32_t)(cj * L.wCell - xoff) & 0xffffu) | ((uint32_t)(ci * L.hCell) << 16);
      }
  }
  return ok ? (narrow ? 2 : 1) : 0;
}

struct Sizes {
  size_t pyrBytes, candEntries, tabEntries;
};
Sizes sizesOf(const orbx_ctx* c, const Geom& g, size_t tabEntries) {
  Sizes s{};
  const LevelGeom& last = g.L[g.nlevels - 1];
  s.pyrBytes = g.nlevels > 1 ? (size_t)(last.imgOff + last.frameStride * c->maxB) : 0;
  s.candEntries = (size_t)(last.candOff + (int64_t)last.candCap * c->maxB);
  s.tabEntries = tabEntries;
  return s;
}

// OctLevel::depthBits D: after D DivideNode splits (cpp:617-676: the left / upper child gets ceil(extent / 2), the right / lower
// one floor) every cell of the level is one pixel, and from there on a key's quadrant digits are all 0 -- which is what lets the
// LDS kernel sort 32-bit keys (the path code's top rootBits + 2 D bits).  One case needs care: with a non-integral hX the
// reference's root = x / hX (cpp:747) against UL.x = hX * i (cpp:715-716) can put the column x = (int)(hX * i) into root i - 1,
// one pixel to the RIGHT of that root's rectangle, where it is routed right for ever (digits 1, 1, 1 ...).  It parts from the
// rectangle's last column one split after the rightmost cell has become one pixel wide, i.e. at depth floor(log2 w) + 1 =
// ceil(log2(w + 1)) for a rectangle of width w <= ceil(hX) -- so D is taken from ceil(hX) + 1, and the difference lies within the
// top D digits (tests: test_device_octree_root_boundary_columns).  A key can never lie left of its rectangle (x / hX >= i
// implies x >= floor(hX * i)), and y always lies inside [0, height).
int octDepthBits(int height, float hX) {
  int e = std::max((int)std::ceil(hX) + 1, height), d = 0;
  while (e > 1) { e = (e + 1) >> 1; d++; }
  return std::max(d, 1);
}

// Tiles of k_pyramid_tiles: TX x TY tiles per frame; tile (i, j) owns the pixels [i w / TX, (i + 1) w / TX) x [j h / TY, (j + 1) h / TY)
// of every level and needs, per level, the bounding rectangle of what it owns and of what its rectangle of the next level reads
// (k_resize's taps: columns ofs .. ofs + 1 and rows ofs .. ofs + 1, clamped as there); level 0's rectangle is what level 1 reads of
// the caller's image.  The taps of a tile go into one blob, positions relative to the source rectangle.  Returns false when a tile
// would not fit the kernel's budgets (then the levels are launched one by one).
constexpr long long kPyrTilesMaxPixels = 2500000ll;  // pixels per launch up to which k_pyramid_tiles is taken (issueExtract)
void pyrTileGrid(const Geom& g, int* TX, int* TY) {
  *TX = std::max(1, (g.L[1].w + 35) / 36);
  *TY = std::max(1, (g.L[1].h + 27) / 28);
}
bool buildPyrTiles(const Geom& g, const std::vector<ResizeTab>& tab, std::vector<PyrTileRect>* out, std::vector<PyrTileTap>* taps,
                   int* nTiles, int* buf0Bytes, int* bufBytes) {
  const int nl = g.nlevels;
  *nTiles = 0;
  *buf0Bytes = *bufBytes = 0;
  out->clear();
  taps->clear();
  if (nl < 2 || g.L[0].w > 32000 || g.L[0].h > 32000) return false;
  int TX, TY;
  pyrTileGrid(g, &TX, &TY);
  out->assign((size_t)TX * TY * nl, PyrTileRect{});
  taps->assign((size_t)TX * TY * ORBX_PYR_TILE_TAPS, PyrTileTap{0, 0});
  int maxPix = 0, maxPix0 = 0;
  for (int tj = 0; tj < TY; tj++)
    for (int ti = 0; ti < TX; ti++) {
      PyrTileRect* R = out->data() + (size_t)(tj * TX + ti) * nl;
      for (int l = 1; l < nl; l++) {
        R[l].ox0 = (int16_t)((long long)ti * g.L[l].w / TX); R[l].ox1 = (int16_t)((long long)(ti + 1) * g.L[l].w / TX);
        R[l].oy0 = (int16_t)((long long)tj * g.L[l].h / TY); R[l].oy1 = (int16_t)((long long)(tj + 1) * g.L[l].h / TY);
      }
      for (int l = nl - 1; l >= 0; l--) {
        int x0 = R[l].ox0, x1 = R[l].ox1, y0 = R[l].oy0, y1 = R[l].oy1;
        const bool ownEmpty = x1 <= x0 || y1 <= y0;
        if (l + 1 < nl && R[l + 1].nx1 > R[l + 1].nx0 && R[l + 1].ny1 > R[l + 1].ny0) {
          const ResizeTab* xt = tab.data() + g.L[l + 1].xtabOff;
          const ResizeTab* yt = tab.data() + g.L[l + 1].ytabOff;
          const int w = g.L[l].w, h = g.L[l].h;
          const int fx0 = std::min(std::max(xt[R[l + 1].nx0].ofs, 0), w - 1), fx1 = std::min(xt[R[l + 1].nx1 - 1].ofs + 1, w - 1) + 1;
          const int fy0 = std::min(std::max(yt[R[l + 1].ny0].ofs, 0), h - 1);
          const int fy1 = std::min(std::max(yt[R[l + 1].ny1 - 1].ofs + 1, 0), h - 1) + 1;
          if (ownEmpty) { x0 = fx0; x1 = fx1; y0 = fy0; y1 = fy1; }
          else { x0 = std::min(x0, fx0); x1 = std::max(x1, fx1); y0 = std::min(y0, fy0); y1 = std::max(y1, fy1); }
        } else if (ownEmpty) {
          x0 = x1 = y0 = y1 = 0;
        }
        R[l].nx0 = (int16_t)x0; R[l].nx1 = (int16_t)x1; R[l].ny0 = (int16_t)y0; R[l].ny1 = (int16_t)y1;
        (l == 0 ? maxPix0 : maxPix) = std::max(l == 0 ? maxPix0 : maxPix, (x1 - x0) * (y1 - y0));
        // (the kernel splits a pixel index into row and column with a 20-bit reciprocal of the width: exact while index * width < 2^20)
        if ((long long)(x1 - x0) * (y1 - y0) * (x1 - x0) >= (1ll << 20)) { out->clear(); taps->clear(); return false; }
      }
      // the tile's taps: per level the needed columns, then the needed rows, relative to the source rectangle (k_resize's clamps)
      PyrTileTap* T = taps->data() + (size_t)(tj * TX + ti) * ORBX_PYR_TILE_TAPS;
      int nT = 0;
      for (int l = 1; l < nl; l++) {
        const ResizeTab* xt = tab.data() + g.L[l].xtabOff;
        const ResizeTab* yt = tab.data() + g.L[l].ytabOff;
        const int sw = g.L[l - 1].w, sh = g.L[l - 1].h;
        const int nw = R[l].nx1 - R[l].nx0, nh = R[l].ny1 - R[l].ny0;
        if (nT + nw + nh > ORBX_PYR_TILE_TAPS) { out->clear(); taps->clear(); return false; }
        for (int x = R[l].nx0; x < R[l].nx1; x++) {
          const int sx = xt[x].ofs, sx1 = std::min(sx + 1, sw - 1);
          T[nT++] = PyrTileTap{(uint32_t)(sx - R[l - 1].nx0) | ((uint32_t)(sx1 - sx) << 16), (uint32_t)xt[x].coef};
        }
        for (int y = R[l].ny0; y < R[l].ny1; y++) {
          const int sy0 = std::min(std::max(yt[y].ofs, 0), sh - 1), sy1 = std::min(std::max(yt[y].ofs + 1, 0), sh - 1);
          T[nT++] = PyrTileTap{(uint32_t)(sy0 - R[l - 1].ny0) | ((uint32_t)(sy1 - sy0) << 16), (uint32_t)yt[y].coef};
        }
      }
    }
  const int buf0 = (maxPix0 + 63) & ~63, buf = (maxPix + 63) & ~63;
  if (buf0 <= 0 || buf <= 0 || ORBX_PYR_TILE_TAPS * (int)sizeof(PyrTileTap) + buf0 + 2 * buf > 60 * 1024) { out->clear(); taps->clear(); return false; }
  *nTiles = TX * TY;
  *buf0Bytes = buf0;
  *bufBytes = buf;
  return true;
}

// Path-code tables of one level (OctLevel::tabOff): DivideNode routes x and y independently (cpp:656-668: pt.x < n1.UR.x, then
// pt.y < n1.BR.y), so the 16 quadrant digits of a candidate are the OR of an x word and a y word, each a function of one
// coordinate: root = x / hX (cpp:747), the root's rectangle (cpp:715-716), then mid = UL + ceil(extent / 2) (cpp:620-621) 16 times.
// The selection units look the two words up instead of walking the 16 splits per candidate.  out: 2 tabW + tabH dwords.
void octCodeTable(const OctLevel& O, uint32_t* out) {
  for (int xi = 0; xi < O.tabW; xi++) {
    const float x = (float)xi;
    int root = (int)(x / O.hX);
    root = std::min(std::max(root, 0), O.nIni - 1);
    int ul = (int)(O.hX * (float)root), br = (int)(O.hX * (float)(root + 1));
    uint32_t s = 0;
    for (int d = 0; d < 16; d++) {
      const int mid = ul + ((br - ul + 1) >> 1);
      const uint32_t q = !(x < (float)mid);
      if (q) ul = mid; else br = mid;
      s = (s << 2) | q;
    }
    out[2 * xi] = s;
    out[2 * xi + 1] = (uint32_t)root;
  }
  uint32_t* ty = out + 2 * (size_t)O.tabW;
  for (int yi = 0; yi < O.tabH; yi++) {
    const float y = (float)yi;
    int ul = 0, br = O.height;
    uint32_t s = 0;
    for (int d = 0; d < 16; d++) {
      const int mid = ul + ((br - ul + 1) >> 1);
      const uint32_t q = !(y < (float)mid);
      if (q) ul = mid; else br = mid;
      s = (s << 2) | (q << 1);
    }
    ty[yi] = s;
  }
}
// Plan of the many-workgroup selection of large units (k_octree_buckets + k_octree_big) for one level: the deepest bucket depth
// the level allows; the depth of a launch is chosen below it from the previous batch's candidate counts (octBigChoose).  A
// bucket depth D0 must never exceed the depth the full passes are GUARANTEED to reach: the pass loop of DistributeOctTree
// (cpp:781-895) stops at the first depth k where the list holds N nodes or one more pass would overshoot (size + 3 nToExpand
// > N), and either needs 4 size_k > N with size_k <= nIni 4^k -- so k >= min{k : nIni 4^(k+1) > N}, and every node the list
// ever holds lies inside one bucket (the device checks k again).  At most ORBX_OCTB_MAX_BUCKETS buckets.
void octBigPlan(OctLevel* O, int nMax) {
  O->bigDMax = -1;
  O->bigD0 = 0; O->bigBuckets = 0; O->bigCapB = 0;
  if (O->nIni < 1 || O->nIni > ORBX_OCTB_MAX_BUCKETS || O->quota < 1) return;
  int kq = 0;
  while ((long long)O->nIni << (2 * (kq + 1)) <= (long long)O->quota && kq < 8) kq++;
  int d = kq;
  while (d > 0 && ((long long)O->nIni << (2 * d)) > ORBX_OCTB_MAX_BUCKETS) d--;
  O->bigDMax = d;
  octBigChoose(O, nMax, 0);
}
inline size_t octBigTabDwords(const OctLevel& O) {
  return O.bigDMax >= 0 ? ((size_t)O.nIni << O.bigDMax) + 1 + ((size_t)1 << O.bigDMax) + 1 : 0;
}
// places the levels' tables one behind the other (tabW / tabH and the big plan set by the caller); returns the dwords they take
size_t octTabLayout(OctLaunch* P) {
  size_t off = 0;
  for (int l = 0; l < P->nlevels; l++) {
    OctLevel& O = P->lev[l];
    O.tabOff = (int32_t)off;
    off += (2 * (size_t)O.tabW + (size_t)O.tabH + 1) & ~(size_t)1;  // (the x pairs are read as 8-byte words)
    O.bigTabOff = (int32_t)off;
    off += (octBigTabDwords(O) + 1) & ~(size_t)1;
  }
  return off;
}
// The buckets' coordinate intervals (OctLevel::bigTabOff), at the level's deepest bucket depth D = bigDMax: a key's top D x
// digits (and its root) grow with x, its top D y digits with y, so every prefix owns an interval of columns / rows;
// xs[(root << D | x prefix)] = its first column, one more entry = the end; ys likewise.  k_octree_buckets reads only the FAST cells that overlap its bucket's rectangle.  Returns false
// when a prefix sequence is not monotonic (it always is; the level then simply keeps the one-workgroup kernel).
bool octBigTable(const OctLevel& O, const uint32_t* codeTab, uint32_t* out) {
  const int d0 = O.bigDMax, nx = O.nIni << d0, ny = 1 << d0;
  auto topBits = [&](uint32_t s, int odd) {  // the digits of depths 1 .. d0: bit 2 (16 - d) + odd - 2 ... of the digit word
    uint32_t v = 0;
    for (int d = 1; d <= d0; d++) v = (v << 1) | ((s >> (2 * (16 - d) + odd)) & 1u);
    return v;
  };
  std::vector<int> first((size_t)std::max(nx, ny) + 1);
  auto fill = [&](int n, int count, auto keyOf, uint32_t* dst) {
    for (int i = 0; i <= n; i++) first[i] = -1;
    int prev = -1;
    for (int v = 0; v < count; v++) {
      const int key = keyOf(v);
      if (key < prev || key >= n) return false;
      if (first[key] < 0) first[key] = v;
      prev = key;
    }
    int nxt = count;
    for (int i = n; i >= 0; i--) {
      if (i < n && first[i] >= 0) nxt = first[i];
      dst[i] = (uint32_t)(i == n ? count : nxt);
    }
    return true;
  };
  if (!fill(nx, O.width, [&](int x) { return (int)((codeTab[2 * (size_t)x + 1] << d0) | topBits(codeTab[2 * (size_t)x], 0)); }, out)) return false;
  const uint32_t* ty = codeTab + 2 * (size_t)O.tabW;
  return fill(ny, O.height, [&](int y) { return (int)topBits(ty[y], 1); }, out + nx + 1);
}
void octCodeTables(OctLaunch* P, std::vector<uint32_t>* out) {
  size_t total = 0;
  for (int l = 0; l < P->nlevels; l++) total = std::max(total, (size_t)P->lev[l].bigTabOff + octBigTabDwords(P->lev[l]) + 1);
  out->assign(total, 0u);
  for (int l = 0; l < P->nlevels; l++) {
    OctLevel& O = P->lev[l];
    octCodeTable(O, out->data() + O.tabOff);
    if (O.bigDMax >= 0 && !octBigTable(O, out->data() + O.tabOff, out->data() + O.bigTabOff)) { O.bigDMax = -1; O.bigBuckets = 0; }
  }
}

// launch constants of the quadtree selection stage; returns the bytes of global scratch it needs
size_t buildOctLaunch(const orbx_ctx* c, const Geom& g, OctLaunch* out) {
  OctLaunch P{};
  P.nlevels = g.nlevels;
  P.selStride = g.selCap;
  int selOff = 0;
  int64_t scr = 0;
  for (int l = 0; l < g.nlevels; l++) {
    const LevelGeom& L = g.L[l];
    OctLevel& O = P.lev[l];
    O.width = L.maxBX - ORBX_MIN_BORDER;
    O.height = L.maxBY - ORBX_MIN_BORDER;
    O.nIni = (int)std::round((float)O.width / (float)O.height);  // cpp:706
    O.hX = (float)O.width / (float)O.nIni;                       // cpp:709
    O.depthBits = octDepthBits(O.height, O.hX);
    O.tabW = O.width;    // candidates lie inside the level's region (relative to minBorder)
    O.tabH = O.height;
    O.wCell = L.wCell;
    O.hCell = L.hCell;
    O.nCols = L.nCols;
    O.quota = L.quota;
    O.cellBase = L.cellBase;
    O.nCells = L.nCols * L.nRows;
    O.segCap = L.segCap;
    P.candOff[l] = L.candOff;
    P.candCap[l] = L.candCap;
    P.selOff[l] = selOff;
    selOff += L.quota;
    // Per-unit scratch: the device side (octreeGlobalUnit) derives its layout from the same two numbers through the same
    // formula, so stride and layout cannot drift apart: nMax = the level's worst-case candidate count (cells x segCap bound,
    // capped at what the sort keys can index), qMax = max(quota, nIni).  A unit with more candidates than nMax is refused on
    // the device (n > nMax -> -1) before anything is written.
    P.scrNMax[l] = std::min(L.candMax, ORBX_OCT_MAX_CAND);
    P.scrStride[l] = (int64_t)octScratchBytes(P.scrNMax[l], std::max(L.quota, O.nIni));  // = octreeGlobalUnit's qMax
    P.scrOff[l] = scr;
    scr += P.scrStride[l] * c->maxB;
    O.bigDMax = -1;
    if (O.nCells < 65536) octBigPlan(&O, P.scrNMax[l]);  // (a bucket's scores carry the FAST cell index in 16 bits)
  }
  P.nCellsTotal = g.nCellsTotal;
  octTabLayout(&P);
  P.codeTab = c->dOctTab;
  *out = P;
  return (size_t)scr;
}

// Every buffer whose size follows from (maxW, maxH, maxB): allocated at orbx_create and again by growTo when a call brings
// a larger frame or batch (ORBextractor::operator() takes any image, cpp:1531-1545).  Streams, events and the lazily sized
// matcher / staging buffers are not touched.
void freeAll(orbx_ctx* ctx) {
  void* dev[] = {ctx->dPyr, ctx->dCand, ctx->dCandCount, ctx->dCellCount, ctx->dMaxN, ctx->dTab, ctx->dCells, ctx->dSel, ctx->dNsel,
                 ctx->dSelStage, ctx->dNselLevel, ctx->dOctScratch, ctx->dOctTab, ctx->dPyrTiles, ctx->dPyrTaps, ctx->dIn, ctx->dKps, ctx->dDesc};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  void* host[] = {ctx->hNsel, ctx->hFlags, ctx->hMaxN, ctx->hWide, ctx->hKpsPin, ctx->hDescPin};
  for (void* p : host)
    if (p) (void)hipHostFree(p);
  ctx->dPyr = nullptr; ctx->dCand = nullptr; ctx->dCandCount = nullptr; ctx->dCellCount = nullptr; ctx->dMaxN = nullptr;
  ctx->dTab = nullptr; ctx->dCells = nullptr; ctx->dSel = nullptr; ctx->dNsel = nullptr; ctx->dSelStage = nullptr; ctx->dNselLevel = nullptr;
  ctx->dOctScratch = nullptr; ctx->dOctTab = nullptr; ctx->dPyrTiles = nullptr; ctx->dPyrTaps = nullptr; ctx->nPyrTiles = 0; ctx->dIn = nullptr; ctx->dKps = nullptr; ctx->dDesc = nullptr;
  ctx->hNsel = nullptr; ctx->hFlags = nullptr; ctx->hMaxN = nullptr; ctx->hWide = nullptr;
  ctx->hKpsPin = nullptr; ctx->hDescPin = nullptr; ctx->hKpsPinDev = nullptr; ctx->hDescPinDev = nullptr; ctx->pinFrames = 0;
  ctx->hNselDev = nullptr; ctx->hFlagsDev = nullptr; ctx->hMaxNDev = nullptr; ctx->hWideDev = nullptr;
  ctx->curW = ctx->curH = ctx->curStride0 = 0;  // the tables on the device are gone with dTab
  ctx->lastB = 0;
  ctx->lastImg0 = nullptr;
}

int allocAll(orbx_ctx* ctx) {
  const int max_width = ctx->maxW, max_height = ctx->maxH, max_batch = ctx->maxB;
  Geom g;
  std::vector<ResizeTab> tab;
  int r = buildGeometry(ctx, max_width, max_height, alignUp(max_width, 64), &g, &tab);
  if (r != ORBX_OK) return r;
  Sizes s = sizesOf(ctx, g, tab.size());
  // row strides grow by at most 63 bytes and tables by a few entries for smaller frames: keep headroom
  // (+ slack: k_pyramid_bands reads up to 11 bytes beyond the last row of a level)
  ctx->pyrBytes = s.pyrBytes + (size_t)64 * max_height * ctx->p.nlevels * max_batch + 4096;
  {
    // per-level bound of cells * segCap that holds for every frame size up to max_width x max_height:
    // cells * ceil(wCell / 2) * ceil(hCell / 2) <= (width / 2 + nCols + 1) * (height / 2 + nRows + 1)
    size_t bound = 0;
    for (int l = 0; l < g.nlevels; l++) {
      const size_t wd = (size_t)(g.L[l].maxBX - ORBX_MIN_BORDER), ht = (size_t)(g.L[l].maxBY - ORBX_MIN_BORDER);
      bound += ((wd / 2 + wd / 35 + 2) * (ht / 2 + ht / 35 + 2) + 4 * (wd / 35 + 1) * (ht / 35 + 1)) * (size_t)max_batch;
    }
    ctx->candEntries = std::max(s.candEntries, bound) + 1024;
    ctx->cellCountEntries = (size_t)g.nCellsTotal * max_batch + 1024;
    // cells of one frame over all levels, for every frame size up to the maximum: (wd / 35 + 1) * (ht / 35 + 1) per level
    size_t cells = 0;
    for (int l = 0; l < g.nlevels; l++) {
      const size_t wd = (size_t)(g.L[l].maxBX - ORBX_MIN_BORDER), ht = (size_t)(g.L[l].maxBY - ORBX_MIN_BORDER);
      cells += (wd / 35 + 1) * (ht / 35 + 1);
    }
    ctx->cellEntries = std::max(cells, (size_t)g.nCellsTotal) + 64;
  }
  ctx->tabEntries = s.tabEntries + 64 * ctx->p.nlevels;
  const size_t B = (size_t)max_batch, nl = (size_t)ctx->p.nlevels;
  const size_t cap = (size_t)std::max(ctx->selCap, 1);
  ctx->inBytes = (size_t)alignUp(max_width, 64) * max_height * B;
#define ALLOC(ptr, bytes)                                                                         \
  if (hipMalloc((void**)&(ptr), std::max<size_t>((bytes), 16)) != hipSuccess) {                  \
    ctx->err = "hipMalloc failed for " #ptr " (context sizing: max_width x max_height x max_batch)"; \
    return ORBX_E_HIP;                                                                            \
  }
#define ALLOCH(ptr, bytes)                                                                        \
  if (hipHostMalloc((void**)&(ptr), std::max<size_t>((bytes), 16), hipHostMallocDefault) != hipSuccess) {  \
    ctx->err = "hipHostMalloc failed for " #ptr;                                                  \
    return ORBX_E_HIP;                                                                            \
  }
  ALLOC(ctx->dPyr, ctx->pyrBytes);
  ALLOC(ctx->dCand, ctx->candEntries * 4);
  ALLOC(ctx->dCandCount, (B * nl + 2) * sizeof(int));
  ALLOC(ctx->dCellCount, ctx->cellCountEntries * sizeof(int));
  ctx->dOverflow = ctx->dCandCount + B * nl;  // (unused tail kept for layout compatibility)
  ALLOC(ctx->dTab, ctx->tabEntries * sizeof(ResizeTab));
  ALLOC(ctx->dCells, ctx->cellEntries * sizeof(FastCell));
  ALLOC(ctx->dSel, B * cap * sizeof(SelKp));
  ALLOC(ctx->dNsel, B * sizeof(int));
  ALLOC(ctx->dSelStage, B * cap * sizeof(SelKp));
  ALLOC(ctx->dNselLevel, B * nl * sizeof(int));
  {
    OctLaunch oct;
    ctx->octScratchBytes = buildOctLaunch(ctx, g, &oct) + 4096;
    // (no level of a smaller frame is larger than the same level of the largest one; its bucket tables may be: bounded by the most buckets)
    ctx->octTabEntries = octTabLayout(&oct) + 64 + (size_t)ctx->p.nlevels * (ORBX_OCTB_MAX_BUCKETS + 64);
    ctx->maxQuota = 0;
    for (int q : ctx->quota) ctx->maxQuota = std::max(ctx->maxQuota, q);
  }
  ALLOC(ctx->dOctScratch, ctx->octScratchBytes);
  ALLOC(ctx->dOctTab, ctx->octTabEntries * 4);
  {  // tiles of the largest geometry (no smaller frame has more) -- or of the largest frame k_pyramid_tiles is taken for at all
    int TX = 1, TY = 1;
    if (g.nlevels > 1) pyrTileGrid(g, &TX, &TY);
    ctx->pyrTileCap = std::min((size_t)TX * TY, (size_t)(kPyrTilesMaxPixels / (36 * 28) + 256));
  }
  ALLOC(ctx->dPyrTiles, ctx->pyrTileCap * g.nlevels * sizeof(PyrTileRect));
  ALLOC(ctx->dPyrTaps, ctx->pyrTileCap * ORBX_PYR_TILE_TAPS * sizeof(PyrTileTap));
  ALLOC(ctx->dIn, ctx->inBytes);
  ALLOC(ctx->dKps, B * cap * sizeof(orbx_keypoint));
  ALLOC(ctx->dDesc, B * cap * 32);
  ALLOCH(ctx->hNsel, B * sizeof(int))
  ALLOCH(ctx->hFlags, 2 * sizeof(int))
  ALLOC(ctx->dMaxN, (B * nl + 2) * sizeof(int));
  if (hipMemset(ctx->dMaxN, 0, (B * nl + 2) * sizeof(int)) != hipSuccess) return ORBX_E_HIP;
  ALLOCH(ctx->hMaxN, 2 * ORBX_MAX_LEVELS * sizeof(int))
  if (hipHostGetDevicePointer((void**)&ctx->hMaxNDev, ctx->hMaxN, 0) != hipSuccess) return ORBX_E_HIP;
  if (hipHostGetDevicePointer((void**)&ctx->hNselDev, ctx->hNsel, 0) != hipSuccess) return ORBX_E_HIP;
  if (hipHostGetDevicePointer((void**)&ctx->hFlagsDev, ctx->hFlags, 0) != hipSuccess) return ORBX_E_HIP;
  ctx->pinFrames = (int)std::min<size_t>(B, 4);
  ALLOCH(ctx->hKpsPin, (size_t)ctx->pinFrames * cap * sizeof(orbx_keypoint))
  ALLOCH(ctx->hDescPin, (size_t)ctx->pinFrames * cap * 32)
  if (hipHostGetDevicePointer((void**)&ctx->hKpsPinDev, ctx->hKpsPin, 0) != hipSuccess) return ORBX_E_HIP;
  if (hipHostGetDevicePointer((void**)&ctx->hDescPinDev, ctx->hDescPin, 0) != hipSuccess) return ORBX_E_HIP;
  ALLOCH(ctx->hWide, 16)
  ctx->hWide[0] = ctx->hWide[1] = 0;
  if (hipHostGetDevicePointer((void**)&ctx->hWideDev, ctx->hWide, 0) != hipSuccess) return ORBX_E_HIP;
  ctx->hFlags[0] = ctx->hFlags[1] = 0;
  for (int i = 0; i < 2 * ORBX_MAX_LEVELS; i++) ctx->hMaxN[i] = 0;
  for (int& v : ctx->candHintL) v = 0;
  ctx->maxSlotsUsed = 0;
#undef ALLOC
#undef ALLOCH
  return ORBX_OK;
}

// A call brought a larger frame or batch than the context was sized for: everything in flight is drained, the size
// dependent buffers are released and allocated again for the new maxima (never smaller than before).
int waitAll(orbx_ctx* ctx);
int growTo(orbx_ctx* ctx, int w, int h, int B) {
  int r = waitAll(ctx);
  // the outcome of the batches drained here (e.g. ORBX_E_CAPACITY of a stream-ordered batch nobody has waited for yet) is not
  // swallowed: like every call that has to wait for an earlier batch, this one returns that batch's error (nothing has been
  // freed or resized yet; everything is drained, so the caller's retry grows the context)
  if (r != ORBX_OK) return r;
  HIPCHK(hipStreamSynchronize(ctx->st));
  if (ctx->st2) HIPCHK(hipStreamSynchronize(ctx->st2));
  const int oldW = ctx->maxW, oldH = ctx->maxH, oldB = ctx->maxB;
  freeAll(ctx);
  ctx->maxW = std::max(oldW, w);
  ctx->maxH = std::max(oldH, h);
  ctx->maxB = std::max(oldB, B);
  r = allocAll(ctx);
  if (r != ORBX_OK) {  // (e.g. out of device memory) fall back to the old sizing so that the context stays usable
    freeAll(ctx);
    ctx->maxW = oldW; ctx->maxH = oldH; ctx->maxB = oldB;
    const std::string why = ctx->err;
    if (allocAll(ctx) != ORBX_OK) { ctx->err = "context could not be re-allocated after a failed growth"; return ORBX_E_HIP; }
    ctx->err = "context growth failed: " + why;
    return r;
  }
  return ORBX_OK;
}

int ensureGeometry(orbx_ctx* ctx, int w, int h, int stride0) {
  if (w == ctx->curW && h == ctx->curH && stride0 == ctx->curStride0) return ORBX_OK;
  if (w <= 0 || h <= 0) return ORBX_E_EMPTY;
  Geom g;
  std::vector<ResizeTab> tab;
  PyrTabInfo pyrInfo;
  int r = buildGeometry(ctx, w, h, stride0, &g, &tab, &pyrInfo);  // (before any growth: a frame the path cannot take must not cost one)
  if (r != ORBX_OK) return r;
  if (w > ctx->maxW || h > ctx->maxH) {  // operator() accepts any image (cpp:1531-1545): the context grows
    r = growTo(ctx, w, h, ctx->maxB);
    if (r != ORBX_OK) return r;
    r = buildGeometry(ctx, w, h, stride0, &g, &tab, &pyrInfo);  // (offsets depend on maxB only, but keep one source of truth)
    if (r != ORBX_OK) return r;
  }
  Sizes s = sizesOf(ctx, g, tab.size());
  OctLaunch oct;
  const size_t octBytes = buildOctLaunch(ctx, g, &oct);
  // every buffer sized at allocAll must hold this geometry's layout end (level offsets are computed for maxB frames)
  bool selFits = true;
  for (int l = 0; l < g.nlevels; l++)
    selFits = selFits && oct.selOff[l] + g.L[l].quota <= g.selCap && oct.scrOff[l] + oct.scrStride[l] * ctx->maxB <= (int64_t)octBytes;
  if (!selFits || s.pyrBytes > ctx->pyrBytes || s.candEntries > ctx->candEntries || s.tabEntries > ctx->tabEntries ||
      octBytes > ctx->octScratchBytes || (size_t)g.nCellsTotal * ctx->maxB > ctx->cellCountEntries) {
    ctx->err = "internal: geometry exceeds the buffers sized at orbx_create";
    return ORBX_E_BADARG;
  }
  if ((size_t)g.nCellsTotal > ctx->cellEntries) {
    ctx->err = "internal: geometry exceeds the buffers sized at orbx_create";
    return ORBX_E_BADARG;
  }
  oct.codeTab = ctx->dOctTab;  // (growTo may have replaced the buffer since buildOctLaunch read the pointer: it did not, but keep one source)
  octCodeTables(&oct, &ctx->hOctTab);
  if (ctx->hOctTab.size() > ctx->octTabEntries) {
    ctx->err = "internal: geometry exceeds the buffers sized at orbx_create";
    return ORBX_E_BADARG;
  }
  ctx->g = g;
  ctx->oct = oct;
  HIPCHK(hipMemcpyAsync(ctx->dOctTab, ctx->hOctTab.data(), ctx->hOctTab.size() * 4, hipMemcpyHostToDevice, ctx->st));
  if ((long long)w * h > kPyrTilesMaxPixels ||  // (never taken for such a frame: no tables, no 6 KB of taps per tile)
      !buildPyrTiles(g, tab, &ctx->hPyrTiles, &ctx->hPyrTaps, &ctx->nPyrTiles, &ctx->pyrTileBuf0, &ctx->pyrTileBuf) ||
      (size_t)ctx->nPyrTiles > ctx->pyrTileCap) {
    ctx->nPyrTiles = 0;
  } else {
    HIPCHK(hipMemcpyAsync(ctx->dPyrTiles, ctx->hPyrTiles.data(), ctx->hPyrTiles.size() * sizeof(PyrTileRect), hipMemcpyHostToDevice, ctx->st));
    HIPCHK(hipMemcpyAsync(ctx->dPyrTaps, ctx->hPyrTaps.data(), ctx->hPyrTaps.size() * sizeof(PyrTileTap), hipMemcpyHostToDevice, ctx->st));
  }
  ctx->hTab = tab;
  ctx->pyrInfo = pyrInfo;
  ctx->fastWaveOk = buildFastCells(g, &ctx->hCells);
  if (!tab.empty()) HIPCHK(hipMemcpyAsync(ctx->dTab, ctx->hTab.data(), tab.size() * sizeof(ResizeTab), hipMemcpyHostToDevice, ctx->st));
  if (!ctx->hCells.empty())
    HIPCHK(hipMemcpyAsync(ctx->dCells, ctx->hCells.data(), ctx->hCells.size() * sizeof(FastCell), hipMemcpyHostToDevice, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  ctx->curW = w;
  ctx->curH = h;
  ctx->curStride0 = stride0;
  for (int& v : ctx->candHintL) v = 0;  // candidate statistics of another frame size say nothing about this one
  return ORBX_OK;
}

struct StageTimer {
  orbx_ctx* c;
  int stage, si;
  hipStream_t st;
  bool on;
  StageTimer(orbx_ctx* c_, int s, int si_, hipStream_t st_) : c(c_), stage(s), si(si_), st(st_), on((c_->profMask >> s) & 1u) {
    if (on) (void)hipEventRecord(c->ev[c->parity][si][s][0], st);
  }
  void stop(int nLaunches) {
    if (on) {
      (void)hipEventRecord(c->ev[c->parity][si][stage][1], st);
      c->launches[stage] += nLaunches;
      c->used[c->parity][si][stage] = true;
    }
  }
};
void collectProfile(orbx_ctx* c, int parity) {
  for (int si = 0; si < 2; si++)
    for (int s = 0; s < ORBX_STAGE_COUNT; s++) {
      if (!c->used[parity][si][s]) continue;
      c->used[parity][si][s] = false;
      float ms = 0;
      if (hipEventElapsedTime(&ms, c->ev[parity][si][s][0], c->ev[parity][si][s][1]) == hipSuccess) c->ms[s] += ms;
    }
}

// candidate order of the reference: cell row, cell col, y, x (cpp:1078-1137; cv::FAST emits row-major)
inline uint64_t candOrderKey(const LevelGeom& L, uint32_t e) {
  const int x = e & 0xfff, y = (e >> 12) & 0xfff;
  const int cr = (y - 3) / L.hCell, cc = (x - 3) / L.wCell;
  return ((uint64_t)(cr * L.nCols + cc) << 24) | ((uint64_t)y << 12) | (uint64_t)x;
}

struct ExtractArgs {
  const uint8_t* dImg0;
  int stride0;
  long long frameStride0;
  int aligned0;
  int safeFrom;      // first frame of the batch behind whose level 0 nothing is known to follow (the last one, normally)
  orbx_keypoint* dKps;
  uint8_t* dDesc;
  int capacity;
  int* dNuser;  // the caller's per-frame count array (device), or nullptr
};

// Row bands of k_pyramid_bands.  Band b owns rows [b*h/K, (b+1)*h/K) of every level and, on top of that, every row of
// level l-1 that its rows of level l read (cv::resize's y table: source rows ofs and ofs+1, clamped), from the last
// level upwards.
PyrBands computePyrBands(const orbx_ctx* ctx, int K, int S = 1) {
  const Geom& g = ctx->g;
  const int nl = g.nlevels;
  PyrBands pb{};
  pb.nBands = K = std::min(K, ORBX_PYR_BANDS_MAX);
  // Column strips: strip s owns groups [ng * s / S, ng * (s + 1) / S) of every level and, on top of that, every group of level l - 1
  // that its groups of level l read: a group's taps lie in the 12 bytes from its first source dword (PyrXGroup::o), i.e. in source
  // groups base / 4 .. base / 4 + 2.  No strip may be empty on any level (the last level decides how many there can be).
  S = std::max(1, std::min(S, ORBX_PYR_STRIPS_MAX));
  while (S > 1 && (g.L[nl - 1].w + 3) / 4 < 16 * S) S--;
  pb.nStrips = S;
  for (int s = 0; s < S; s++) {
    int need0 = 0, need1 = 0;
    for (int l = nl - 1; l >= 1; l--) {
      const int ng = (g.L[l].w + 3) / 4;
      int a = (int)((long long)ng * s / S), b = (int)((long long)ng * (s + 1) / S);
      if (l < nl - 1 && need1 > need0) { a = std::min(a, need0); b = std::max(b, need1); }
      pb.g0[s][l] = (int16_t)a;
      pb.g1[s][l] = (int16_t)b;
      need0 = need1 = 0;
      if (b > a && l >= 2) {
        const ResizeTab* xt = ctx->hTab.data() + g.L[l].xtabOff;
        const int ngs = (g.L[l - 1].w + 3) / 4;
        need0 = std::min(std::max((xt[4 * a].ofs & ~3) / 4, 0), ngs - 1);
        need1 = std::min((xt[4 * (b - 1)].ofs & ~3) / 4 + 3, ngs);
      }
    }
  }
  pb.dual2 = ctx->pyrInfo.dual2;
  for (int l = 0; l < nl; l++) { pb.xoff[l] = ctx->pyrInfo.xoff[l]; pb.yoff[l] = ctx->pyrInfo.yoff[l]; }
  for (int b = 0; b < K; b++) {
    int need0 = 0, need1 = 0;  // rows of level l that the band's rows of level l + 1 read
    for (int l = nl - 1; l >= 1; l--) {
      const int h = g.L[l].h;
      int r0 = (int)((long long)b * h / K), r1 = (int)((long long)(b + 1) * h / K);
      if (l < nl - 1 && need1 > need0) { r0 = std::min(r0, need0); r1 = std::max(r1, need1); }
      pb.r0[b][l] = (int16_t)r0;
      pb.r1[b][l] = (int16_t)r1;
      pb.maxRows = std::max(pb.maxRows, r1 - r0);
      need0 = need1 = 0;
      if (r1 > r0 && l >= 2) {
        const ResizeTab* yt = ctx->hTab.data() + g.L[l].ytabOff;
        const int sh = g.L[l - 1].h;
        need0 = std::min(std::max(yt[r0].ofs, 0), sh - 1);
        need1 = std::min(std::max(yt[r1 - 1].ofs + 1, 0), sh - 1) + 1;
      }
    }
  }
  return pb;
}

// issues the kernels of the extraction of frames [f0, f0 + n) on stream `st` (stream slot si), no synchronisation:
// part 0 = pyramid + FAST, part 1 = selection + descriptors (two calls, so that the two half-batch chains can be issued
// alternately and the second chain does not wait for the host to have issued all of the first)
int issueExtract(orbx_ctx* ctx, int si, hipStream_t st, int f0, int n, const ExtractArgs& a, int part) {
  Geom g = ctx->g;
  g.frame0 = f0;
  OctLaunch oct = ctx->oct;
  oct.frame0 = f0;
  const int nl = g.nlevels;
  if (part == 0) {
  // whole pyramid in one launch (k_pyramid_bands) when every level meets the dword path's preconditions and the batch
  // is large enough to fill the device with (bands x frames) workgroups; otherwise one launch per level
  const bool noBands = knobOn(KNOB_NO_BANDS);  // diagnostics (orbx_debug_set)
  const bool noTiles = knobOn(KNOB_NO_TILES);  // diagnostics: small batches launch the levels one by one
  // k_pyramid_tiles up to this many frames per launch: measured at 640x480 (tools/batch_sweep.py), one frame 0.108 against 0.117 ms
  // per synchronous call and 22.1 k against 18.3 k frames/s on four lanes, 8 frames level, 16 frames 133 k against 147 k on lanes
  // (and up to eight VGA frames' worth of pixels: 8 frames of 3840x2160 take 0.81 ms in tiles, 0.41 ms level by level)
  const int tilesMax = (int)knob(KNOB_TILES_MAX_FRAMES, 8);
  const long long tilesMaxPx = std::min(knob(KNOB_TILES_MAX_PIXELS, kPyrTilesMaxPixels), kPyrTilesMaxPixels);
  const int bandsEnv = (int)knob(KNOB_PYR_BANDS, 0);  // diagnostics
  const int bandsMin = (int)knob(KNOB_BANDS_MIN_FRAMES, 0);  // diagnostics
  const int stripsEnv = (int)knob(KNOB_PYR_STRIPS, 0);       // diagnostics
  // rounds 2 - 5: from 32 frames per stream, or from 8 when the frames are large (16 frames 1080p: 0.43 -> 0.30 ms; 16 frames 640x480
  // were better off with the per-level launches then: 0.252 vs 0.263 ms per 32-frame call)
  // (round 5: with column strips also few large frames -- the four-frame halves of a 3840x2160 batch in 32 bands x 4 strips: 0.227 ms
  // per batch against 0.352 level by level and 0.456 in 32 bands without strips; tools/exp_pyr_strips.sh)
  // (round 6, with round 5's kernel -- row pairs, one group per thread, strips: from 9 frames, i.e. wherever the levels used to be
  // launched one by one.  Per synchronous call of 32 frames 640x480 = two halves of 16: pyramid stage 0.096 -> 0.068 ms, 136.5 k ->
  // 162.4 k frames/s; 24 frames: 107.6 k -> 128.9 k; 12 frames on one stream: 66.3 k -> 69.8 k, on four lanes 143 k -> 154 k;
  // up to 8 frames k_pyramid_tiles stays (8: 53.0 k against 53.8 k, level; tools/exp_c4_small.py))
  const bool enough = bandsMin > 0 ? n >= bandsMin : (n >= 9 || (n >= 2 && (long long)n * g.L[0].w * g.L[0].h >= (16ll << 20)));
  bool banded = nl > 1 && ctx->pyrInfo.ok && a.aligned0 && enough && !noBands;  // (dword loads: level 0 rows 4-byte aligned)
  PyrBands pb{};
  if (banded) {  // a band's rows of one level are staged by one pass of the workgroup: at most 256
    // Bands: neighbouring bands share ~8 rows of level 1 per band, so bands should be fat: 3 per 640x480 frame (512 threads
    // per workgroup, 72 VGPRs -> three workgroups per CU; 128 or 256 frames x 3 bands = one resident round or two).  Measured on
    // 256 frames: 2 bands 0.158 ms, 3 0.151, 4 0.170, 6 0.190 (with 256-thread workgroups: 4 0.168, 7 0.160, 8 0.192, 12 0.198).
    // A small batch takes more, thinner bands instead, to give every CU a workgroup
    const int rowBands = std::min(std::max(g.L[1].h / 130, 3), ORBX_PYR_BANDS_MAX);
    int K = n * rowBands >= 256 ? rowBands : std::min(ORBX_PYR_BANDS_MAX, std::max(rowBands, (256 + n - 1) / n));
    if (bandsEnv > 0) K = bandsEnv;
    // strips when the bands alone leave half the chip without a workgroup (n K <= 128): the fewest of 2 / 4 / 8 that give 384
    int S = 1;
    if (n * K <= 128)
      while (S < ORBX_PYR_STRIPS_MAX && n * K * S < 384) S *= 2;
    // rows of more than 256 groups (frames wider than 1228 pixels): strips of 100 .. 200 groups take the one-group-per-thread
    // instance (56 VGPRs, four workgroups per CU), 768 workgroups for up to four frames, 512 or more otherwise.  With the round's
    // final kernel, bands x strips (tools/exp_pyr_strips2.sh): 3840x2160 four-frame halves 32 x 4 0.203 ms, 16 x 8 0.194, 24 x 8
    // 0.180; eight frames on a lane 32 x 1 16.5 k frames/s, 16 x 4 17.3 k; 1920x1080 sixteen-frame halves 16 x 1 0.244 ms / 46.3 k
    // frames/s, 16 x 2 0.200 / 47.7 k, 8 x 4 0.195 / 48.4 k
    const int ng1 = (g.L[1].w + 3) / 4;
    if (ng1 > 256 && bandsEnv <= 0) {
      S = ng1 > 512 && n <= 4 ? 8 : 4;
      K = std::min(ORBX_PYR_BANDS_MAX, std::max((n <= 4 ? 768 : 512) / (n * S), 8));
    }
    if (stripsEnv > 0) S = stripsEnv;
    pb = computePyrBands(ctx, K, S);
    while (pb.maxRows > 256 && K < ORBX_PYR_BANDS_MAX) pb = computePyrBands(ctx, K = std::min(2 * K, ORBX_PYR_BANDS_MAX), S);
    banded = pb.maxRows <= 256;
    pb.safeFrom = a.safeFrom;
  }
  if (banded) {
    StageTimer tm(ctx, ORBX_STAGE_PYRAMID, si, st);
    HIPCHK(launch_pyramid_bands(st, n, a.dImg0, a.frameStride0, ctx->dPyr, g, ctx->dTab, pb));
    tm.stop(1);
    ctx->lastLaunch[0] = 1;
    ctx->lastLaunch[1] = pb.nBands * pb.nStrips;
  } else if (nl > 1 && ctx->nPyrTiles > 0 && !noTiles && n <= tilesMax && (long long)n * g.L[0].w * g.L[0].h <= tilesMaxPx) {
    // small batches: one launch, a workgroup per tile of a frame, the level chain through LDS
    StageTimer tm(ctx, ORBX_STAGE_PYRAMID, si, st);
    HIPCHK(launch_pyramid_tiles(st, n, a.dImg0, a.frameStride0, ctx->dPyr, g, ctx->dPyrTiles, ctx->dPyrTaps, ctx->nPyrTiles,
                                ctx->pyrTileBuf0, ctx->pyrTileBuf));
    tm.stop(1);
    ctx->lastLaunch[0] = 2;
    ctx->lastLaunch[1] = ctx->nPyrTiles;
  } else {
    StageTimer tm(ctx, ORBX_STAGE_PYRAMID, si, st);
    for (int l = 1; l < nl; l++) {
      const LevelGeom& S = g.L[l - 1];
      const LevelGeom& D = g.L[l];
      const uint8_t* src = l == 1 ? a.dImg0 : ctx->dPyr + S.imgOff;
      const long long sfs = l == 1 ? a.frameStride0 : S.frameStride;
      HIPCHK(launch_resize(st, n, src + (long long)f0 * sfs, sfs, S.w, S.h, S.stride, ctx->dPyr + D.imgOff + (long long)f0 * D.frameStride,
                           D.frameStride, D.w, D.h, D.stride, ctx->dTab + D.xtabOff, ctx->dTab + D.ytabOff,
                           D.resizeSpanOk && (l > 1 || a.aligned0),
                           // (behind the pyramid buffer's levels there is slack; behind the caller's last frame nothing is known)
                           l > 1 ? n : std::min(std::max(a.safeFrom - f0, 0), n)));
    }
    if (nl > 1) tm.stop(nl - 1);
    ctx->lastLaunch[0] = 0;
    ctx->lastLaunch[1] = 0;
  }
  {
    StageTimer tm(ctx, ORBX_STAGE_FAST, si, st);
    HIPCHK(launch_fast(st, n, a.dImg0, a.frameStride0, a.aligned0, ctx->dPyr, g, ctx->dCand, ctx->dCellCount, ctx->dCells,
                       ctx->fastWaveOk, &ctx->lastLaunch[2]));
    tm.stop(1);
    ctx->lastLaunch[5] = n;
  }
  return ORBX_OK;
  }  // part 0
  bool staged = false;
  {  // selection stage: quadtree per (frame, level), then level-major compaction
    StageTimer tm(ctx, ORBX_STAGE_SELECT, si, st);
    // per-level candidate maxima of this stream slot: device accumulators + their pinned host mirror (read after the sync)
    int* dMax = ctx->dMaxN;  // (indexed by frame: the two half batches do not meet)
    HIPCHK(launch_octree(st, n, ctx->dCand, ctx->dCellCount, oct, ctx->dSelStage, ctx->dNselLevel, ctx->dOctScratch, dMax,
                         ctx->candHintL, 0, &ctx->lastLaunch[3]));
    // small launches (the one-frame call): the descriptor kernel indexes the staging lists itself and does the bookkeeping
    // (DescStage) -- one kernel less on the call's critical path
    const bool noStaged = knobOn(KNOB_DESC_NO_STAGED);  // diagnostics
    const int stagedMax = (int)knob(KNOB_DESC_STAGED_MAX, ORBX_DESC_STAGED_MAX_UNITS);  // (experiments)
    staged = !noStaged && n * g.nlevels <= stagedMax;
    if (!staged)
      HIPCHK(launch_sel_compact(st, n, ctx->dSelStage, ctx->dNselLevel, oct, ctx->dSel, ctx->dNsel, a.dNuser, ctx->hNselDev, g.selCap,
                                ctx->hFlagsDev + ctx->parity, dMax, ctx->hMaxNDev + si * ORBX_MAX_LEVELS));
    ctx->maxSlotsUsed |= 1 << si;
    tm.stop(staged ? 1 : 2);
    ctx->lastLaunch[4] = (ctx->lastLaunch[4] & 1) | (staged ? 2 : 0);
  }
  {
    StageTimer tm(ctx, ORBX_STAGE_DESCRIBE, si, st);
    DescStage ds;
    ds.selStage = ctx->dSelStage; ds.nselLevel = ctx->dNselLevel; ds.nsel = ctx->dNsel; ds.nselUser = a.dNuser; ds.hostNsel = ctx->hNselDev;
    ds.hostErr = ctx->hFlagsDev + ctx->parity; ds.maxN = ctx->dMaxN; ds.hostMaxN = ctx->hMaxNDev + si * ORBX_MAX_LEVELS;
    ds.selStride = oct.selStride;
    for (int l = 0; l < ORBX_MAX_LEVELS; l++) ds.selOff[l] = oct.selOff[l];
    HIPCHK(launch_describe_patch(st, n, g.selCap, a.dImg0, a.frameStride0, a.aligned0, ctx->dPyr, g, ctx->dSel, ctx->dNsel,
                                 a.dKps, a.dDesc, a.capacity, ctx->gaussVariant, ctx->libmVariant, staged ? &ds : nullptr));
    tm.stop(1);
  }
  return ORBX_OK;
}

struct MatchArgs {  // optional matching fused behind the extraction (pairs of frames of the same batch)
  int nPairs = 0;
  const int32_t* hFirst = nullptr;
  const int32_t* hSecond = nullptr;
  orbx_bounds b{};
  int window = 0;
  float nnratio = 0;
  int checkOri = 0;
  int32_t* dMatches12 = nullptr;
  int32_t* dNmatches = nullptr;
  int32_t* dStats = nullptr;
};

int ensureMatchScratch(orbx_ctx* ctx, int nPairs, int capacity);

int issueMatch(orbx_ctx* ctx, int si, hipStream_t st, int pair0, int n, const MatchArgs& m, const orbx_keypoint* dKps,
               const uint8_t* dDesc, const int* dN, int capacity) {
  if (n <= 0) return ORBX_OK;
  StageTimer tm(ctx, ORBX_STAGE_MATCH, si, st);
  const int wide = ctx->wideLaunched[ctx->parity] ? 1 : 0;
  HIPCHK(launch_match(st, n, ctx->dPairs, ctx->dPairs + m.nPairs, dKps, dDesc, dN, capacity, m.b, m.window, m.nnratio, m.checkOri,
                      m.dMatches12, m.dNmatches, m.dStats, ctx->dMatchScratch, pair0, wide, ctx->hWideDev + ctx->parity, ctx->dMatchDiag));
  tm.stop(wide ? 3 : 1);  // k_match_jacobi (+ k_match_wide_lists + k_match_wide_resolve, for pending pairs only)
  return ORBX_OK;
}

// Before the matching of the batch being issued (ctx->parity): decide whether its wide kernels go with it, and keep what
// a late launch would need.
void armMatch(orbx_ctx* ctx, const MatchArgs& m, const orbx_keypoint* dKps, const uint8_t* dDesc, const int* dN, int capacity) {
  const int par = ctx->parity;
  ctx->hWide[par] = 0;  // (the previous batch of this parity has been waited for)
  ctx->wideLaunched[par] = ctx->wideExpected || ctx->eventOrdered;
  orbx_ctx::LateMatch& L = ctx->late[par];
  L.valid = true;
  L.nPairs = m.nPairs; L.capacity = capacity; L.window = m.window; L.checkOri = m.checkOri; L.nnratio = m.nnratio; L.b = m.b;
  L.dKps = dKps; L.dDesc = dDesc; L.dN = dN;
  L.dMatches12 = m.dMatches12; L.dNmatches = m.dNmatches; L.dStats = m.dStats;
}

// After the batch of `parity` has completed: if k_match_jacobi handed pairs on and the wide kernels were not issued with
// the batch, run them now (device drained first: later batches may be using the pairs' scratch), and adapt the expectation.
int settleMatch(orbx_ctx* ctx, int parity) {
  orbx_ctx::LateMatch& L = ctx->late[parity];
  if (!L.valid) return ORBX_OK;
  L.valid = false;
  const bool needed = ctx->hWide[parity] != 0;
  if (needed && !ctx->wideLaunched[parity]) {
    HIPCHK(hipStreamSynchronize(ctx->st));
    if (ctx->st2) HIPCHK(hipStreamSynchronize(ctx->st2));
    HIPCHK(launch_match(ctx->st, L.nPairs, ctx->dPairs, ctx->dPairs + L.nPairs, L.dKps, L.dDesc, L.dN, L.capacity, L.b, L.window,
                        L.nnratio, L.checkOri, L.dMatches12, L.dNmatches, L.dStats, ctx->dMatchScratch, 0, 2, ctx->hWideDev + parity, ctx->dMatchDiag));
    HIPCHK(hipStreamSynchronize(ctx->st));
  }
  if (needed) {
    ctx->wideExpected = true;
    ctx->wideIdle = 0;
  } else if (ctx->wideLaunched[parity] && ++ctx->wideIdle >= 3) {
    ctx->wideExpected = false;  // three batches in a row carried the wide kernels for nothing
  }
  return ORBX_OK;
}

// Pair list -> dPairs (first[], then second[]).  Trackers match the same pairs batch after batch, so an unchanged list
// is not copied again (two copy commands less at the head of the stream).
int waitAll(orbx_ctx* ctx);
int uploadPairs(orbx_ctx* ctx, int nPairs, const int32_t* hFirst, const int32_t* hSecond, hipStream_t st, bool* copied) {
  *copied = false;
  std::vector<int32_t>& lp = ctx->lastPairs;
  if ((int)lp.size() == 2 * nPairs && std::memcmp(lp.data(), hFirst, sizeof(int32_t) * nPairs) == 0 &&
      std::memcmp(lp.data() + nPairs, hSecond, sizeof(int32_t) * nPairs) == 0)
    return ORBX_OK;
  int r = waitAll(ctx);  // batches in flight (either stream) may still read the list dPairs holds
  if (r != ORBX_OK) return r;
  // the copy reads the context's own copy of the list, which stays put until the next change (the caller's arrays may be
  // gone by the time a stream-ordered call's copy runs)
  lp.assign(hFirst, hFirst + nPairs);
  lp.insert(lp.end(), hSecond, hSecond + nPairs);
  *copied = true;
  if (hipMemcpyAsync(ctx->dPairs, lp.data(), sizeof(int) * 2 * nPairs, hipMemcpyHostToDevice, st) != hipSuccess) {
    lp.clear();
    ctx->err = "hipMemcpyAsync (pair list)";
    return ORBX_E_HIP;
  }
  return ORBX_OK;
}

// Waits for the oldest batch in flight and does its host-side epilogue: stage times, the selection stage's instance hint
// for the next batch, the (sticky) selection error flag.
int waitOldest(orbx_ctx* ctx) {
  if (ctx->pending <= 0) return ORBX_OK;
  const int parity = (int)((ctx->seqIssue - (unsigned)ctx->pending) & 1u);
  HIPCHK(hipEventSynchronize(ctx->evDone[parity]));
  if (ctx->done2Used[parity]) HIPCHK(hipEventSynchronize(ctx->evDone2[parity]));
  if (ctx->outUsed[parity]) {  // a host-frame batch: its results are in the caller's buffers
    HIPCHK(hipEventSynchronize(ctx->evOut[parity]));
    ctx->outUsed[parity] = false;
  }
  ctx->pending--;
  {
    const int sm = settleMatch(ctx, parity);
    if (sm != ORBX_OK) return sm;
  }
  collectProfile(ctx, parity);
  {  // per level, the largest candidate count of a unit in the batch: picks the selection kernel's instances for the next one
    for (int l = 0; l < ctx->p.nlevels; l++) {
      int m = 0;
      for (int q = 0; q < 2; q++)
        if (ctx->maxSlotsUsed & (1 << q)) { const int v = ctx->hMaxN[q * ORBX_MAX_LEVELS + l]; m = ORBX_OCT_FB_MAX(m, v); }
      ctx->candHintL[l] = m;
    }
    if (ctx->pending == 0) ctx->maxSlotsUsed = 0;
  }
  if (ctx->hFlags[parity]) {  // raised by this batch's k_sel_compact through mapped host memory (one slot per parity: the
    ctx->hFlags[parity] = 0;   // other batch in flight reports into its own)
    ctx->err = "selection stage: more candidates or nodes than its scratch can hold";
    return ORBX_E_CAPACITY;
  }
  return ORBX_OK;
}
int laneWaitOne(orbx_ctx* ctx) {  // the oldest batch issued to the lanes
  if (ctx->lanes.empty() || ctx->laneDone == ctx->laneIssue) return ORBX_OK;
  orbx_ctx* c = ctx->lanes[ctx->laneDone % ctx->lanes.size()];
  const int r = waitOldest(c);
  if (r != ORBX_OK) ctx->err = c->err;
  if (r != ORBX_E_HIP || c->pending == 0) ctx->laneDone++;
  return r;
}

int waitAll(orbx_ctx* ctx) {
  int r = ORBX_OK;
  while (ctx->laneDone != ctx->laneIssue) {
    const unsigned before = ctx->laneDone;
    const int q = laneWaitOne(ctx);
    if (q == ORBX_E_HIP && ctx->laneDone == before) return q;
    if (r == ORBX_OK) r = q;
  }
  while (ctx->pending > 0) {
    const int q = waitOldest(ctx);
    if (q == ORBX_E_HIP) return q;  // (pending may not have moved)
    if (r == ORBX_OK) r = q;
  }
  return r;
}

// The whole extraction (and optionally the pair matching) of one batch.  d_img0: device pointer of frame 0 / level 0.
// Large batches are issued as two half-batches on two streams so that the latency-bound stages of one half (quadtree
// selection, matching) overlap the VALU-bound stages of the other (FAST, descriptors).
// async: return once the batch is issued (at most two in flight; see orbx_extract_match_batch_device_async).
int extractCore(orbx_ctx* ctx, int B, const uint8_t* dImg0, int w, int h, int stride0, long long frameStride0,
                orbx_keypoint* dKps, uint8_t* dDesc, int capacity, int* dNout, const MatchArgs* match, bool async = false) {
  if (B <= 0) return ORBX_E_BADARG;
  if (capacity < ctx->selCap) return ORBX_E_CAPACITY;
  int r = ORBX_OK;
  if (B > ctx->maxB) {  // a larger batch than the context was sized for: it grows (everything in flight is drained first)
    r = growTo(ctx, ctx->maxW, ctx->maxH, B);
    if (r != ORBX_OK) return r;
  }
  if (ctx->pending >= 2) {  // the events of this parity are still in use
    r = waitOldest(ctx);
    if (r != ORBX_OK) return r;
  }
  if (ctx->pending > 0 && !(w == ctx->curW && h == ctx->curH && stride0 == ctx->curStride0 && B == ctx->lastB)) {
    // a new geometry rewrites tables the batches in flight read; another batch size moves the border between the halves
    // of the internal buffers the two streams own
    r = waitAll(ctx);
    if (r != ORBX_OK) return r;
  }
  ctx->parity = (int)(ctx->seqIssue & 1u);
  r = ensureGeometry(ctx, w, h, stride0);
  if (r != ORBX_OK) return r;
  ctx->parity = (int)(ctx->seqIssue & 1u);  // (a growth waits for the batches in flight)
  hipStream_t st = ctx->st;
  ExtractArgs a;
  a.dImg0 = dImg0; a.stride0 = stride0; a.frameStride0 = frameStride0;
  a.aligned0 = (((uintptr_t)dImg0 | (uintptr_t)stride0 | (uintptr_t)frameStride0) & 3) == 0;
  a.dKps = dKps; a.dDesc = dDesc; a.capacity = capacity;
  // k_pyramid_bands may read a few bytes beyond a level-0 row: harmless while another frame follows in the caller's buffer
  a.safeFrom = frameStride0 >= (long long)stride0 * h ? B - 1 : 0;
  int* dN = dNout ? dNout : ctx->dNsel;
  const int nPairs = match ? match->nPairs : 0;
  bool pairsCopied = false;
  if (nPairs > 0) {
    if (capacity >= (1 << 20)) return ORBX_E_BADARG;
    for (int p = 0; p < nPairs; p++)
      if (match->hFirst[p] < 0 || match->hFirst[p] >= B || match->hSecond[p] < 0 || match->hSecond[p] >= B) return ORBX_E_BADARG;
    r = ensureMatchScratch(ctx, nPairs, capacity);
    if (r != ORBX_OK) return r;
    r = uploadPairs(ctx, nPairs, match->hFirst, match->hSecond, st, &pairsCopied);
    if (r != ORBX_OK) return r;
    ctx->parity = (int)(ctx->seqIssue & 1u);  // (uploadPairs may have waited)
  }
  a.dNuser = dNout;
  ctx->late[ctx->parity].valid = false;
  if (nPairs > 0) armMatch(ctx, *match, dKps, dDesc, dN, capacity);

  // two half batches from 16 frames on -- or from 4 when the frames are large (4 x 4K): the selection of such frames is a handful
  // of long latency-bound units (one 1024-thread workgroup per frame and level) under which the other half's pyramid, FAST and
  // descriptors find the chip almost empty (8 frames 4K / 8000 features: 6.0 k -> 8 k frames/s)
  const bool noSplitEnv = knobOn(KNOB_NO_SPLIT);
  const bool enoughToSplit = B >= 16 || (B >= 4 && (long long)B * w * h >= (32ll << 20));
  const bool split = ctx->st2 != nullptr && enoughToSplit && !noSplitEnv && !ctx->noSplit;
  const int n0 = split ? ((B / 2) & ~1) : B;
  // pairs whose frames both lie in one half can run right behind that half's extraction, on its stream; this needs
  // the pair list to be ordered [half 0][half 1][rest] (true for consecutive pairs (2k, 2k+1))
  int p0 = 0, p1 = 0;
  if (nPairs > 0 && split) {
    while (p0 < nPairs && match->hFirst[p0] < n0 && match->hSecond[p0] < n0) p0++;
    p1 = p0;
    while (p1 < nPairs && match->hFirst[p1] >= n0 && match->hSecond[p1] >= n0) p1++;
  }
  ctx->done2Used[ctx->parity] = split;
  ctx->lastLaunch[4] = split ? 1 : 0;
  ctx->lastLaunch[6] = nPairs > 0 && ctx->wideLaunched[ctx->parity] ? 1 : 0;
  // Everything below queues work; a failure in the middle must not leave launches in flight behind an error return (the
  // next call would reuse their buffers and events), so the issue is one unit with one exit.
  auto issue = [&]() -> int {
    int q = ORBX_OK;
    if (split) {
      // The two streams are independent pipelines (disjoint halves of every internal buffer).  The second stream waits for
      // the first whenever the first may carry something the second depends on: the producer of the frames (a copy into
      // dIn queued by the host-pointer entry points, work of any synchronous call, a caller's stream), a new pair list or
      // new tables.  Only a stream-ordered (_async) call on the context's own stream with another batch in flight and an
      // unchanged pair list skips the fork: nothing but the previous batches sits on st then (every other entry point
      // drains the batches in flight first), and the frames are the caller's own resident array.
      const bool needFork = !async || pairsCopied || ctx->pending == 0 || !ctx->ownStream || ctx->hostInput;
      if (needFork) {
        HIPCHK(hipEventRecord(ctx->evFork, st));
        HIPCHK(hipStreamWaitEvent(ctx->st2, ctx->evFork, 0));
      }
      // the two chains are issued alternately: the device starts on the second while the host still issues the first
      if ((q = issueExtract(ctx, 0, st, 0, n0, a, 0)) != ORBX_OK) return q;
      if ((q = issueExtract(ctx, 1, ctx->st2, n0, B - n0, a, 0)) != ORBX_OK) return q;
      if ((q = issueExtract(ctx, 0, st, 0, n0, a, 1)) != ORBX_OK) return q;
      if (nPairs > 0 && (q = issueMatch(ctx, 0, st, 0, p0, *match, dKps, dDesc, dN, capacity)) != ORBX_OK) return q;
      if ((q = issueExtract(ctx, 1, ctx->st2, n0, B - n0, a, 1)) != ORBX_OK) return q;
      if (nPairs > 0 && (q = issueMatch(ctx, 1, ctx->st2, p0, p1 - p0, *match, dKps, dDesc, dN, capacity)) != ORBX_OK) return q;
      if (nPairs > p1) {
        HIPCHK(hipEventRecord(ctx->evJoin, ctx->st2));
        HIPCHK(hipStreamWaitEvent(st, ctx->evJoin, 0));
        if ((q = issueMatch(ctx, 0, st, p1, nPairs - p1, *match, dKps, dDesc, dN, capacity)) != ORBX_OK) return q;
      }
      HIPCHK(hipEventRecord(ctx->evDone2[ctx->parity], ctx->st2));
    } else {
      if ((q = issueExtract(ctx, 0, st, 0, B, a, 0)) != ORBX_OK) return q;
      if ((q = issueExtract(ctx, 0, st, 0, B, a, 1)) != ORBX_OK) return q;
      if (nPairs > 0 && (q = issueMatch(ctx, 0, st, 0, nPairs, *match, dKps, dDesc, dN, capacity)) != ORBX_OK) return q;
    }
    // (the per-frame counts and the error flag arrive in pinned host memory straight from k_sel_compact)
    HIPCHK(hipEventRecord(ctx->evDone[ctx->parity], st));
    return ORBX_OK;
  };
  r = issue();
  if (r != ORBX_OK) {
    // part of the batch may be queued: drain both streams in-process, forget what was armed for it, and make the next
    // call upload the pair list again if this call's upload is the one that may have failed
    const std::string why = ctx->err;
    (void)hipStreamSynchronize(st);
    if (ctx->st2) (void)hipStreamSynchronize(ctx->st2);
    ctx->late[ctx->parity].valid = false;
    ctx->hWide[ctx->parity] = 0;
    ctx->hFlags[ctx->parity] = 0;
    for (int si = 0; si < 2; si++)
      for (int s2 = 0; s2 < ORBX_STAGE_COUNT; s2++) ctx->used[ctx->parity][si][s2] = false;
    if (pairsCopied) ctx->lastPairs.clear();
    ctx->err = why;
    return r;
  }
  ctx->seqIssue++;
  ctx->pending++;
  ctx->lastImg0 = dImg0;
  ctx->lastFrameStride0 = frameStride0;
  ctx->lastB = B;
  return async ? ORBX_OK : waitAll(ctx);
}

}  // namespace

// =================================================================================================
// the libm reading of a context (orbx_set_libm_variant; a new lane takes its parent's): the descriptor's cos / sin is a kernel
// argument; the constructor's pow (cpp:536) feeds the per-level quotas -- recomputed, and in the rare case that they change (never
// for a two-decimal scale factor) the buffers sized from them are re-planned
namespace {
int applyLibmVariant(orbx_ctx* c, int libm_variant) {
  if (c->libmVariant == libm_variant) return ORBX_OK;
  const std::vector<int> oldQuota = c->quota;
  c->libmVariant = libm_variant;
  computeTables(c);
  if (c->quota == oldQuota) return ORBX_OK;
  c->curW = c->curH = 0;
  c->curStride0 = -1;
  return growTo(c, c->maxW, c->maxH, c->maxB);
}
}  // namespace

// C ABI
// =================================================================================================
extern "C" {

int orbx_create(const orbx_params* params, int device_id, int max_width, int max_height, int max_batch, void* stream,
                orbx_ctx** out) {
  if (!params || !out) return ORBX_E_BADARG;
  *out = nullptr;
  if (params->nlevels < 1 || params->nlevels > ORBX_MAX_LEVELS || params->nfeatures < 1 || !(params->scale_factor >= 1.0f) ||
      (params->scale_factor == 1.0f && params->nlevels > 1) || max_width < 1 || max_height < 1 || max_batch < 1)
    return ORBX_E_BADARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return ORBX_E_HIP;
  if (hipSetDevice(device_id) != hipSuccess) return ORBX_E_HIP;
  orbx_ctx* ctx = new orbx_ctx();
  ctx->p = *params;
  ctx->device = device_id;
  ctx->maxW = max_width;
  ctx->maxH = max_height;
  ctx->maxB = max_batch;
  computeTables(ctx);
  auto fail = [&](int code) {
    orbx_destroy(ctx);
    return code;
  };
  if (stream) {
    ctx->st = (hipStream_t)stream;
  } else {
    if (hipStreamCreateWithFlags(&ctx->st, hipStreamNonBlocking) != hipSuccess) return fail(ORBX_E_HIP);
    ctx->ownStream = true;
  }
  {
    const int r = allocAll(ctx);
    if (r != ORBX_OK) return fail(r);
  }
  for (int si = 0; si < 2; si++)
    for (int s2 = 0; s2 < ORBX_STAGE_COUNT; s2++)
      for (int k = 0; k < 2; k++)
        for (int par = 0; par < 2; par++)
          if (hipEventCreate(&ctx->ev[par][si][s2][k]) != hipSuccess) return fail(ORBX_E_HIP);
  for (int par = 0; par < 2; par++)
    if (hipEventCreateWithFlags(&ctx->evDone[par], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->evDone2[par], hipEventDisableTiming) != hipSuccess)
      return fail(ORBX_E_HIP);
  if (hipStreamCreateWithFlags(&ctx->st2, hipStreamNonBlocking) != hipSuccess) return fail(ORBX_E_HIP);
  if (hipEventCreateWithFlags(&ctx->evFork, hipEventDisableTiming) != hipSuccess) return fail(ORBX_E_HIP);
  if (hipEventCreateWithFlags(&ctx->evJoin, hipEventDisableTiming) != hipSuccess) return fail(ORBX_E_HIP);
  if (hipEventCreateWithFlags(&ctx->evOrder, hipEventDisableTiming) != hipSuccess) return fail(ORBX_E_HIP);
  *out = ctx;
  return ORBX_OK;
}

void orbx_destroy(orbx_ctx* ctx) {
  if (!ctx) return;
  for (orbx_ctx* c : ctx->lanes) orbx_destroy(c);
  ctx->lanes.clear();
  (void)hipSetDevice(ctx->device);
  if (ctx->st) (void)hipStreamSynchronize(ctx->st);
  if (ctx->st2) (void)hipStreamSynchronize(ctx->st2);
  freeAll(ctx);
  void* dev[] = {ctx->dMatchScratch, ctx->dMatchDiag, ctx->dPairs, ctx->dMblk, ctx->dMi, ctx->dColor, ctx->dScore};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  if (ctx->hMblk) (void)hipHostFree(ctx->hMblk);
  if (ctx->hMo) (void)hipHostFree(ctx->hMo);
  for (int si = 0; si < 2; si++)
    for (int s = 0; s < ORBX_STAGE_COUNT; s++)
      for (int k = 0; k < 2; k++)
        for (int par = 0; par < 2; par++)
          if (ctx->ev[par][si][s][k]) (void)hipEventDestroy(ctx->ev[par][si][s][k]);
  for (int par = 0; par < 2; par++) {
    if (ctx->evDone[par]) (void)hipEventDestroy(ctx->evDone[par]);
    if (ctx->evDone2[par]) (void)hipEventDestroy(ctx->evDone2[par]);
  }
  if (ctx->evFork) (void)hipEventDestroy(ctx->evFork);
  if (ctx->evJoin) (void)hipEventDestroy(ctx->evJoin);
  if (ctx->evOrder) (void)hipEventDestroy(ctx->evOrder);
  for (int par = 0; par < 2; par++)
    if (ctx->evOut[par]) (void)hipEventDestroy(ctx->evOut[par]);
  if (ctx->dPipeOut) (void)hipFree(ctx->dPipeOut);
  if (ctx->st2) (void)hipStreamDestroy(ctx->st2);
  if (ctx->ownStream && ctx->st) (void)hipStreamDestroy(ctx->st);
  delete ctx;
}

int orbx_set_opencv_variant(orbx_ctx* ctx, int gaussian_variant, int gray_variant) {
  if (!ctx || gaussian_variant < 0 || gaussian_variant > 1 || gray_variant < 0 || gray_variant > 1) return ORBX_E_BADARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  const int w = waitAll(ctx);  // batches in flight keep the constants they were issued with
  if (w != ORBX_OK) return w;  // (an earlier batch's error is returned by the call that has to wait for it; retry)
  ctx->gaussVariant = gaussian_variant;
  ctx->grayVariant = gray_variant;
  for (orbx_ctx* c : ctx->lanes) { c->gaussVariant = gaussian_variant; c->grayVariant = gray_variant; }
  return ORBX_OK;
}

int orbx_set_libm_variant(orbx_ctx* ctx, int libm_variant) {
  if (!ctx || libm_variant < 0 || libm_variant > 1) return ORBX_E_BADARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  const int w = waitAll(ctx);  // batches in flight keep the reading they were issued with
  if (w != ORBX_OK) return w;
  int r = applyLibmVariant(ctx, libm_variant);
  for (orbx_ctx* c : ctx->lanes) {
    const int rl = applyLibmVariant(c, libm_variant);
    if (r == ORBX_OK) r = rl;
  }
  return r;
}

const char* orbx_last_error(const orbx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int orbx_get_levels(const orbx_ctx* ctx) { return ctx ? ctx->p.nlevels : ORBX_E_BADARG; }
float orbx_get_scale_factor(const orbx_ctx* ctx) { return ctx ? (float)(double)ctx->p.scale_factor : 0.f; }
int orbx_get_tables(const orbx_ctx* ctx, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                    int32_t* features_per_level) {
  if (!ctx) return ORBX_E_BADARG;
  for (int i = 0; i < ctx->p.nlevels; i++) {
    if (scale) scale[i] = ctx->scale[i];
    if (inv_scale) inv_scale[i] = ctx->invScale[i];
    if (sigma2) sigma2[i] = ctx->sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = ctx->invSigma2[i];
    if (features_per_level) features_per_level[i] = ctx->quota[i];
  }
  return ORBX_OK;
}
int orbx_get_umax(const orbx_ctx* ctx, int32_t* umax16) {
  if (!ctx || !umax16) return ORBX_E_BADARG;
  for (int i = 0; i < 16; i++) umax16[i] = ctx->umax[i];
  return ORBX_OK;
}

int orbx_extract_batch_device(orbx_ctx* ctx, int n_frames, const uint8_t* d_imgs, int width, int height, int stride,
                              size_t frame_stride_bytes, orbx_keypoint* d_kps, uint8_t* d_desc32, int capacity,
                              int32_t* d_n_out) {
  if (!ctx) return ORBX_E_BADARG;
  if (!d_imgs || width <= 0 || height <= 0) return ORBX_E_EMPTY;
  if (!d_kps || !d_desc32 || stride < width) return ORBX_E_BADARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  return extractCore(ctx, n_frames, d_imgs, width, height, stride, (long long)frame_stride_bytes, d_kps, d_desc32, capacity,
                     d_n_out, nullptr);
}

int orbx_extract_batch(orbx_ctx* ctx, int n_frames, const uint8_t* imgs, int width, int height, int stride,
                       size_t frame_stride_bytes, int lap0, int lap1, orbx_keypoint* kps, uint8_t* desc32, int capacity,
                       int* n_out, int* mono_out) {
  if (!ctx) return ORBX_E_BADARG;
  if (!imgs || width <= 0 || height <= 0) return ORBX_E_EMPTY;
  if (!kps || !desc32 || !n_out || stride < width || n_frames < 1) return ORBX_E_BADARG;
  if (capacity < ctx->selCap) return ORBX_E_CAPACITY;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  if (width > ctx->maxW || height > ctx->maxH) {  // operator() takes any image (cpp:1531-1545): grow before staging the frames
    Geom gChk;
    int rg = buildGeometry(ctx, width, height, alignUp(width, 64), &gChk, nullptr);
    if (rg != ORBX_OK) return rg;
    rg = growTo(ctx, width, height, ctx->maxB);
    if (rg != ORBX_OK) return rg;
  }
  const int dstride = alignUp(width, 64);
  const size_t dfs = (size_t)dstride * height;
  const int cap = std::max(ctx->selCap, 1);
  const bool latTrace = knobOn(KNOB_LAT_TRACE);
  static double latAcc[5] = {0, 0, 0, 0, 0};
  static long latN = 0;
  auto nowUs = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; };
  double tt[5] = {0, 0, 0, 0, 0};
  if (latTrace) tt[0] = nowUs();
  for (int f0 = 0; f0 < n_frames; f0 += ctx->maxB) {
    const int B = std::min(ctx->maxB, n_frames - f0);
    if (frame_stride_bytes == (size_t)stride * height || B == 1) {
      // frames stacked without gaps: the whole batch is one tall image -> one copy command instead of one per frame
      HIPCHK(hipMemcpy2DAsync(ctx->dIn, dstride, imgs + (size_t)f0 * frame_stride_bytes, stride, width, (size_t)height * B,
                              hipMemcpyHostToDevice, ctx->st));
    } else {
      for (int f = 0; f < B; f++)
        HIPCHK(hipMemcpy2DAsync(ctx->dIn + f * dfs, dstride, imgs + (size_t)(f0 + f) * frame_stride_bytes, stride, width, height,
                                hipMemcpyHostToDevice, ctx->st));
    }
    if (latTrace) tt[1] = nowUs();
    const bool noDirect = knobOn(KNOB_NO_DIRECT_OUT);  // diagnostics
    const bool direct = B <= ctx->pinFrames && !noDirect;
    int r = extractCore(ctx, B, ctx->dIn, width, height, dstride, (long long)dfs, direct ? ctx->hKpsPinDev : ctx->dKps,
                        direct ? ctx->hDescPinDev : ctx->dDesc, cap, nullptr, nullptr);
    if (r != ORBX_OK) return r;
    if (latTrace) tt[2] = nowUs();
    int maxN = 0;
    for (int f = 0; f < B; f++) {
      n_out[f0 + f] = ctx->hNsel[f];
      maxN = std::max(maxN, ctx->hNsel[f]);
    }
    if (direct) {  // the synchronous extractCore has waited for the descriptor kernel: the results are in host memory
      for (int f = 0; f < B; f++) {
        const size_t nf = (size_t)ctx->hNsel[f];
        memcpy(kps + (size_t)(f0 + f) * capacity, ctx->hKpsPin + (size_t)f * cap, nf * sizeof(orbx_keypoint));
        memcpy(desc32 + (size_t)(f0 + f) * capacity * 32, ctx->hDescPin + (size_t)f * cap * 32, nf * 32);
      }
      maxN = 0;
    }
    if (maxN > 0) {
      // one strided copy per array: row f = the first maxN entries of frame f (entries between n_out[f] and maxN are
      // unspecified, like everything beyond n_out[f] in the caller's capacity-sized rows)
      HIPCHK(hipMemcpy2DAsync(kps + (size_t)f0 * capacity, sizeof(orbx_keypoint) * (size_t)capacity, ctx->dKps,
                              sizeof(orbx_keypoint) * (size_t)cap, sizeof(orbx_keypoint) * (size_t)maxN, B, hipMemcpyDeviceToHost,
                              ctx->st));
      HIPCHK(hipMemcpy2DAsync(desc32 + (size_t)f0 * capacity * 32, (size_t)32 * capacity, ctx->dDesc, (size_t)32 * cap,
                              (size_t)32 * maxN, B, hipMemcpyDeviceToHost, ctx->st));
    }
    if (latTrace) tt[3] = nowUs();
    if (!direct) HIPCHK(hipStreamSynchronize(ctx->st));
    if (latTrace) {
      tt[4] = nowUs();
      for (int i = 1; i < 5; i++) latAcc[i] += tt[i] - tt[i - 1];
      if (++latN % 100 == 0)
        fprintf(stderr, "orbx lat trace (us, mean of %ld): h2d enqueue %.1f, extractCore %.1f, d2h enqueue %.1f, final sync %.1f\n", latN,
                latAcc[1] / latN, latAcc[2] / latN, latAcc[3] / latN, latAcc[4] / latN);
    }
    for (int f = 0; f < B; f++) {
      const int n = n_out[f0 + f];
      int mono = n;
      if (!(lap0 == 0 && lap1 == 0) && n > 0) {
        // stereo keypoints (x in [lap0, lap1]) go back-to-front, the rest front-to-back (cpp:1637-1646)
        orbx_keypoint* ko = kps + (size_t)(f0 + f) * capacity;
        uint8_t* dout = desc32 + (size_t)(f0 + f) * capacity * 32;
        std::vector<orbx_keypoint> k2(ko, ko + n);
        std::vector<uint8_t> d2(dout, dout + (size_t)n * 32);
        int mi = 0, si = n - 1;
        for (int i = 0; i < n; i++) {
          const int dst = (k2[i].x >= lap0 && k2[i].x <= lap1) ? si-- : mi++;
          ko[dst] = k2[i];
          memcpy(dout + (size_t)dst * 32, &d2[(size_t)i * 32], 32);
        }
        mono = mi;
      }
      if (mono_out) mono_out[f0 + f] = mono;
    }
  }
  return ORBX_OK;
}

int orbx_extract(orbx_ctx* ctx, const uint8_t* img, int width, int height, int stride, int lap0, int lap1, orbx_keypoint* kps,
                 uint8_t* desc32, int capacity, int* n_out) {
  int n = 0, mono = 0;
  int r = orbx_extract_batch(ctx, 1, img, width, height, stride, 0, lap0, lap1, kps, desc32, capacity, &n, &mono);
  if (n_out) *n_out = r == ORBX_OK ? n : 0;
  return r == ORBX_OK ? mono : r;
}

int orbx_host_register(orbx_ctx* ctx, void* ptr, size_t bytes) {
  if (!ctx || !ptr || bytes == 0) return ORBX_E_BADARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  HIPCHK(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  return ORBX_OK;
}
int orbx_host_unregister(orbx_ctx* ctx, void* ptr) {
  if (!ctx || !ptr) return ORBX_E_BADARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  HIPCHK(hipHostUnregister(ptr));
  return ORBX_OK;
}

int orbx_level_size(const orbx_ctx* ctx, int level, int* width, int* height) {
  if (!ctx || level < 0 || level >= ctx->p.nlevels || ctx->curW == 0) return ORBX_E_BADARG;
  if (width) *width = ctx->g.L[level].w;
  if (height) *height = ctx->g.L[level].h;
  return ORBX_OK;
}

int orbx_download_pyramid(orbx_ctx* ctx, int frame, int level, int border, uint8_t* dst, int dst_stride) {
  if (!ctx || !dst || level < 0 || level >= ctx->p.nlevels || frame < 0 || frame >= ctx->lastB || border < 0) return ORBX_E_BADARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  {  // stream-ordered batches may still be writing the pyramid on either stream
    const int w = waitAll(ctx);
    if (w == ORBX_E_HIP) return w;
  }
  const LevelGeom& L = ctx->g.L[level];
  if (dst_stride < L.w + 2 * border) return ORBX_E_BADARG;
  const uint8_t* src = level == 0 ? ctx->lastImg0 + (long long)frame * ctx->lastFrameStride0
                                  : ctx->dPyr + L.imgOff + (long long)frame * L.frameStride;
  HIPCHK(hipMemcpy2DAsync(dst + (size_t)border * dst_stride + border, dst_stride, src, L.stride, L.w, L.h, hipMemcpyDeviceToHost,
                          ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  if (border > 0) {  // cv::copyMakeBorder BORDER_REFLECT_101 (cpp:1689,1708)
    auto refl = [](int p, int n) {
      if (n == 1) return 0;
      while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
      return p;
    };
    const int W = L.w + 2 * border, H = L.h + 2 * border;
    for (int y = 0; y < H; y++) {
      const int sy = refl(y - border, L.h);
      uint8_t* row = dst + (size_t)y * dst_stride;
      const uint8_t* srow = dst + (size_t)(sy + border) * dst_stride + border;
      for (int x = 0; x < W; x++) {
        const int sx = refl(x - border, L.w);
        if (y - border != sy || x - border != sx) row[x] = srow[sx];
      }
    }
  }
  return ORBX_OK;
}

// ---- matching ------------------------------------------------------------------------------------
}  // extern "C"
namespace {
int ensureMatchScratch(orbx_ctx* ctx, int nPairs, int capacity) {
  if (ctx->pending > 0 && ((size_t)nPairs * (size_t)matchScratchStride(capacity) > ctx->matchScratchInts || (size_t)nPairs > ctx->pairsCap)) {
    const int w = waitAll(ctx);  // the buffers about to be replaced are in use
    if (w != ORBX_OK) return w;
  }
  // per pair: matchScratchStride(capacity) ints (orbx_device.h) -- the one function both this allocation and launch_match's
  // stride argument use; the kernels index a pair's area with [pair * stride, (pair + 1) * stride) and clamp their own
  // counts to the capacities the layout was computed from (queries / trains to matchWideCap(capacity), list entries per
  // query to MW_CP, per-frame counts to `capacity`)
  if (!ctx->dMatchDiag) {  // (diagnostic counters, orbx_debug_match_counters: 64 bytes, zero at first use)
    HIPCHK(hipMalloc((void**)&ctx->dMatchDiag, 64));
    HIPCHK(hipMemset(ctx->dMatchDiag, 0, 64));
  }
  const size_t need = (size_t)nPairs * (size_t)matchScratchStride(capacity);
  if (need > ctx->matchScratchInts) {
    if (ctx->dMatchScratch) (void)hipFree(ctx->dMatchScratch);
    ctx->dMatchScratch = nullptr;
    ctx->matchScratchInts = 0;
    HIPCHK(hipMalloc((void**)&ctx->dMatchScratch, need * sizeof(int)));
    ctx->matchScratchInts = need;
  }
  if ((size_t)nPairs > ctx->pairsCap) {
    if (ctx->dPairs) (void)hipFree(ctx->dPairs);
    ctx->dPairs = nullptr;
    ctx->lastPairs.clear();
    ctx->pairsCap = 0;
    HIPCHK(hipMalloc((void**)&ctx->dPairs, (size_t)nPairs * 2 * sizeof(int)));
    ctx->pairsCap = nPairs;
  }
  return ORBX_OK;
}

// device staging of the host-pointer entry points (orbx_match_init, orbx_undistort_keypoints): two keypoint /
// descriptor arrays of `cap` entries and a small int area
int ensureHostPairBuffers(orbx_ctx* ctx, size_t cap) {
  if (cap <= ctx->mCap) return ORBX_OK;
  if (ctx->dMblk) (void)hipFree(ctx->dMblk);
  if (ctx->dMi) (void)hipFree(ctx->dMi);
  if (ctx->hMblk) (void)hipHostFree(ctx->hMblk);
  if (ctx->hMo) (void)hipHostFree(ctx->hMo);
  ctx->dMblk = nullptr; ctx->hMblk = nullptr; ctx->hMo = nullptr; ctx->hMoDev = nullptr;
  ctx->dMk = nullptr; ctx->dMd = nullptr; ctx->dMi = nullptr; ctx->mCap = 0;
  const size_t descOff = (2 * cap * sizeof(orbx_keypoint) + 63) / 64 * 64, cntOff = descOff + 2 * cap * 32;
  const size_t bytes = cntOff + 64;
  HIPCHK(hipMalloc((void**)&ctx->dMblk, bytes));
  HIPCHK(hipMalloc((void**)&ctx->dMi, (cap + 8) * sizeof(int)));
  HIPCHK(hipHostMalloc((void**)&ctx->hMblk, bytes, hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&ctx->hMo, (cap + 8) * sizeof(int), hipHostMallocDefault));
  HIPCHK(hipHostGetDevicePointer((void**)&ctx->hMoDev, ctx->hMo, 0));
  memset(ctx->hMblk, 0, bytes);
  ctx->dMk = reinterpret_cast<orbx_keypoint*>(ctx->dMblk);
  ctx->dMd = ctx->dMblk + descOff;
  ctx->mBlkBytes = bytes; ctx->mDescOff = descOff; ctx->mCntOff = cntOff;
  ctx->mCap = cap;
  return ORBX_OK;
}

// mK / mDistCoef (CV_32F) -> the doubles cv::undistortPoints computes with
CamD makeCam(const orbx_camera& c) {
  CamD d{};
  d.fx = c.fx; d.fy = c.fy; d.cx = c.cx; d.cy = c.cy;
  d.ifx = 1. / d.fx; d.ify = 1. / d.fy;
  d.k0 = c.k1; d.k1 = c.k2; d.k2 = c.p1; d.k3 = c.p2;
  d.distorted = c.k1 != 0.0f;
  return d;
}
}  // namespace

extern "C" {

int orbx_match_init_batch_device(orbx_ctx* ctx, int n_pairs, const int32_t* h_first, const int32_t* h_second,
                                 const orbx_keypoint* d_kps, const uint8_t* d_desc32, const int32_t* d_n, int capacity,
                                 const orbx_bounds* bounds, int window_size, float nnratio, int check_orientation,
                                 int32_t* d_matches12, int32_t* d_nmatches, int32_t* d_stats) {
  if (!ctx || n_pairs < 0 || !h_first || !h_second || !d_kps || !d_desc32 || !d_n || !bounds || !d_matches12 || !d_nmatches ||
      capacity < 1 || capacity >= (1 << 20))
    return ORBX_E_BADARG;
  if (bounds->max_x <= bounds->min_x || bounds->max_y <= bounds->min_y) return ORBX_E_BADARG;
  if (n_pairs == 0) return ORBX_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  int r = waitAll(ctx);  // batches issued with the _async call
  if (r != ORBX_OK) return r;
  ctx->parity = (int)(ctx->seqIssue & 1u);
  r = ensureMatchScratch(ctx, n_pairs, capacity);
  if (r != ORBX_OK) return r;
  bool copied = false;
  r = uploadPairs(ctx, n_pairs, h_first, h_second, ctx->st, &copied);
  if (r != ORBX_OK) return r;
  MatchArgs m;
  m.nPairs = n_pairs; m.b = *bounds; m.window = window_size; m.nnratio = nnratio; m.checkOri = check_orientation;
  m.dMatches12 = d_matches12; m.dNmatches = d_nmatches; m.dStats = d_stats;
  armMatch(ctx, m, d_kps, d_desc32, d_n, capacity);
  r = issueMatch(ctx, 0, ctx->st, 0, n_pairs, m, d_kps, d_desc32, d_n, capacity);
  if (r != ORBX_OK) return r;
  HIPCHK(hipStreamSynchronize(ctx->st));
  r = settleMatch(ctx, ctx->parity);  // (runs the wide kernels now if the batch turned out to need them)
  collectProfile(ctx, ctx->parity);
  return r;
}

}  // extern "C"
namespace {
int extractMatch(orbx_ctx* ctx, int n_frames, const uint8_t* d_imgs, int width, int height, int stride,
                 size_t frame_stride_bytes, orbx_keypoint* d_kps, uint8_t* d_desc32, int capacity, int32_t* d_n_out, int n_pairs,
                 const int32_t* h_first, const int32_t* h_second, const orbx_bounds* bounds, int window_size, float nnratio,
                 int check_orientation, int32_t* d_matches12, int32_t* d_nmatches, int32_t* d_stats, bool async) {
  if (!ctx) return ORBX_E_BADARG;
  if (!d_imgs || width <= 0 || height <= 0) return ORBX_E_EMPTY;
  if (!d_kps || !d_desc32 || !d_n_out || stride < width || n_pairs < 0) return ORBX_E_BADARG;
  if (n_pairs > 0 && (!h_first || !h_second || !bounds || !d_matches12 || !d_nmatches || bounds->max_x <= bounds->min_x ||
                      bounds->max_y <= bounds->min_y))
    return ORBX_E_BADARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  MatchArgs m;
  m.nPairs = n_pairs; m.hFirst = h_first; m.hSecond = h_second;
  if (n_pairs > 0) { m.b = *bounds; m.window = window_size; m.nnratio = nnratio; m.checkOri = check_orientation; }
  m.dMatches12 = d_matches12; m.dNmatches = d_nmatches; m.dStats = d_stats;
  return extractCore(ctx, n_frames, d_imgs, width, height, stride, (long long)frame_stride_bytes, d_kps, d_desc32, capacity,
                     d_n_out, &m, async);
}
}  // namespace
extern "C" {

int orbx_extract_match_batch_device(orbx_ctx* ctx, int n_frames, const uint8_t* d_imgs, int width, int height, int stride,
                                    size_t frame_stride_bytes, orbx_keypoint* d_kps, uint8_t* d_desc32, int capacity,
                                    int32_t* d_n_out, int n_pairs, const int32_t* h_first, const int32_t* h_second,
                                    const orbx_bounds* bounds, int window_size, float nnratio, int check_orientation,
                                    int32_t* d_matches12, int32_t* d_nmatches, int32_t* d_stats) {
  return extractMatch(ctx, n_frames, d_imgs, width, height, stride, frame_stride_bytes, d_kps, d_desc32, capacity, d_n_out, n_pairs,
                      h_first, h_second, bounds, window_size, nnratio, check_orientation, d_matches12, d_nmatches, d_stats, false);
}

int orbx_extract_match_batch_device_async(orbx_ctx* ctx, int n_frames, const uint8_t* d_imgs, int width, int height, int stride,
                                          size_t frame_stride_bytes, orbx_keypoint* d_kps, uint8_t* d_desc32, int capacity,
                                          int32_t* d_n_out, int n_pairs, const int32_t* h_first, const int32_t* h_second,
                                          const orbx_bounds* bounds, int window_size, float nnratio, int check_orientation,
                                          int32_t* d_matches12, int32_t* d_nmatches, int32_t* d_stats) {
  if (ctx && !ctx->lanes.empty()) {  // pipeline mode: the whole batch to the next lane, at most one batch per lane in flight
    if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
    const unsigned L = (unsigned)ctx->lanes.size();
    if (ctx->laneIssue - ctx->laneDone >= L) {
      const int w = laneWaitOne(ctx);  // (an error of that earlier batch is returned by the call that has to wait for it)
      if (w != ORBX_OK) return w;
    }
    orbx_ctx* c = ctx->lanes[ctx->laneIssue % L];
    const int r = extractMatch(c, n_frames, d_imgs, width, height, stride, frame_stride_bytes, d_kps, d_desc32, capacity, d_n_out,
                               n_pairs, h_first, h_second, bounds, window_size, nnratio, check_orientation, d_matches12, d_nmatches,
                               d_stats, true);
    if (r != ORBX_OK) { ctx->err = c->err; return r; }
    memcpy(ctx->lastLaunch, c->lastLaunch, sizeof ctx->lastLaunch);
    ctx->lastLaunch[7] = (int)(ctx->laneIssue % L) + 1;
    ctx->laneIssue++;
    return ORBX_OK;
  }
  ctx->lastLaunch[7] = 0;
  return extractMatch(ctx, n_frames, d_imgs, width, height, stride, frame_stride_bytes, d_kps, d_desc32, capacity, d_n_out, n_pairs,
                      h_first, h_second, bounds, window_size, nnratio, check_orientation, d_matches12, d_nmatches, d_stats, true);
}

// The host-frame form of the stream-ordered call: frames come from (page-locked) host memory, results go to (page-locked) host
// memory.  Per batch, on the lane's stream: upload -> the kernels -> copies back; the uploads of the lanes' batches overlap the
// kernels of the batches in front of them, so a caller that keeps `depth` batches in flight is bound by the slower of PCIe and
// the kernels (640x480: PCIe).
}  // extern "C"
namespace {
int hostBatchIssue(orbx_ctx* c, int n_frames, const uint8_t* h_imgs, int width, int height, int stride, size_t frame_stride_bytes,
                   orbx_keypoint* h_kps, uint8_t* h_desc32, int capacity, int32_t* h_n_out, int n_pairs, const int32_t* h_first,
                   const int32_t* h_second, const orbx_bounds* bounds, int window_size, float nnratio, int check_orientation,
                   int32_t* h_matches12, int32_t* h_nmatches, int32_t* h_stats) {
  orbx_ctx* ctx = c;  // (HIPCHK reports into the context that issues)
  int r = waitAll(c);  // one batch per context: dIn and the result block are the batch's own
  if (r != ORBX_OK) return r;
  if (capacity < c->selCap) return ORBX_E_CAPACITY;
  if (width > c->maxW || height > c->maxH || n_frames > c->maxB) {  // any image size, any batch size: the context grows first
    Geom gChk;
    r = buildGeometry(c, width, height, alignUp(width, 64), &gChk, nullptr);
    if (r != ORBX_OK) return r;
    r = growTo(c, std::max(width, c->maxW), std::max(height, c->maxH), std::max(n_frames, c->maxB));
    if (r != ORBX_OK) return r;
  }
  const int dstride = alignUp(width, 64);
  const size_t dfs = (size_t)dstride * height;
  // the result block: [kps][desc][n][matches12][nmatches][stats], every part 256-byte aligned
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t oK = 0, oD = al(oK + sizeof(orbx_keypoint) * (size_t)capacity * n_frames), oN = al(oD + (size_t)32 * capacity * n_frames),
               oM = al(oN + sizeof(int32_t) * (size_t)n_frames), oNm = al(oM + sizeof(int32_t) * (size_t)capacity * n_pairs),
               oSt = al(oNm + sizeof(int32_t) * (size_t)n_pairs), total = al(oSt + sizeof(int32_t) * 3 * (size_t)n_pairs);
  if (total > c->pipeOutBytes) {
    if (c->dPipeOut) (void)hipFree(c->dPipeOut);
    c->dPipeOut = nullptr;
    c->pipeOutBytes = 0;
    HIPCHK(hipMalloc((void**)&c->dPipeOut, total));
    c->pipeOutBytes = total;
  }
  for (int par = 0; par < 2; par++)
    // (release to SYSTEM scope: k_copy_out's stores into the caller's page-locked arrays must be visible to the host when the
    // event has completed, also for arrays allocated non-coherent -- ADVICE r05)
    if (!c->evOut[par]) HIPCHK(hipEventCreateWithFlags(&c->evOut[par], hipEventDisableTiming | hipEventReleaseToSystem));
  // nothing of a batch may be left to the host-side wait (the copies back run behind the kernels): the wide matcher kernels travel
  // with every batch of this context from now on, as after orbx_order_before
  c->eventOrdered = true;
  // ---- upload (stream-ordered; asynchronous when the caller's memory is page-locked) ----
  if (frame_stride_bytes == (size_t)stride * height || n_frames == 1) {
    if (stride == dstride)
      HIPCHK(hipMemcpyAsync(c->dIn, h_imgs, dfs * n_frames, hipMemcpyHostToDevice, c->st));
    else
      HIPCHK(hipMemcpy2DAsync(c->dIn, dstride, h_imgs, stride, width, (size_t)height * n_frames, hipMemcpyHostToDevice, c->st));
  } else {
    for (int f = 0; f < n_frames; f++)
      HIPCHK(hipMemcpy2DAsync(c->dIn + f * dfs, dstride, h_imgs + (size_t)f * frame_stride_bytes, stride, width, height,
                              hipMemcpyHostToDevice, c->st));
  }
  orbx_keypoint* dK = reinterpret_cast<orbx_keypoint*>(c->dPipeOut + oK);
  uint8_t* dD = c->dPipeOut + oD;
  int32_t* dN = reinterpret_cast<int32_t*>(c->dPipeOut + oN);
  int32_t* dM = reinterpret_cast<int32_t*>(c->dPipeOut + oM);
  int32_t* dNm = reinterpret_cast<int32_t*>(c->dPipeOut + oNm);
  int32_t* dSt = h_stats ? reinterpret_cast<int32_t*>(c->dPipeOut + oSt) : nullptr;
  c->hostInput = true;
  r = extractMatch(c, n_frames, c->dIn, width, height, dstride, dfs, dK, dD, capacity, dN, n_pairs, h_first, h_second, bounds,
                   window_size, nnratio, check_orientation, dM, dNm, dSt, true);
  c->hostInput = false;
  if (r != ORBX_OK) return r;
  // ---- results back, behind the batch's kernels on both streams ----
  const int par = (int)((c->seqIssue - 1u) & 1u);
  if (c->done2Used[par]) HIPCHK(hipStreamWaitEvent(c->st, c->evDone2[par], 0));
  // Page-locked result arrays are filled by a KERNEL storing over the link (k_copy_out): a copy command behind the kernels would sit
  // at the head of its DMA queue until they have finished and hold up the next batches' uploads queued behind it.  Pageable
  // arrays (no device mapping) take copy commands.
  // (the WHOLE array must be mapped, not just its first byte: a caller may have registered part of a buffer, and the kernel's stores
  // beyond a mapped range would be a GPU memory fault where the copy command returns an error -- ADVICE r05)
  auto mapped = [&](void* h, size_t bytes) -> void* {
    void *d = nullptr, *dLast = nullptr;
    if (!h || bytes == 0) return nullptr;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipHostGetDevicePointer(&dLast, (char*)h + bytes - 1, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if ((char*)dLast - (char*)d != (ptrdiff_t)(bytes - 1)) return nullptr;  // (two registrations that happen to be adjacent on the host)
    return d;
  };
  const size_t nF = (size_t)n_frames, nP = (size_t)n_pairs, capz = (size_t)capacity;
  void* mK = mapped(h_kps, sizeof(orbx_keypoint) * capz * nF); void* mD = mapped(h_desc32, 32 * capz * nF);
  void* mN = mapped(h_n_out, sizeof(int32_t) * nF);
  void* mM = n_pairs > 0 ? mapped(h_matches12, sizeof(int32_t) * capz * nP) : nullptr;
  void* mNm = n_pairs > 0 ? mapped(h_nmatches, sizeof(int32_t) * nP) : nullptr;
  void* mSt = n_pairs > 0 && h_stats ? mapped(h_stats, sizeof(int32_t) * 3 * nP) : nullptr;
  const bool allMapped = mK && mD && mN && (n_pairs == 0 || (mM && mNm && (!h_stats || mSt)));
  if (allMapped) {
    CopyOut co{};
    int ns = 0;
    auto seg = [&](const void* src, void* dst, const int32_t* cnt, int rows, int rowDwords, int mult) {
      co.s[ns].src = (const uint32_t*)src; co.s[ns].dst = (uint32_t*)dst; co.s[ns].cnt = cnt;
      co.s[ns].rows = rows; co.s[ns].rowDwords = rowDwords; co.s[ns].mult = mult;
      ns++;
    };
    seg(dK, mK, dN, n_frames, capacity * 7, 7);
    seg(dD, mD, dN, n_frames, capacity * 8, 8);
    seg(dN, mN, nullptr, 1, n_frames, 0);
    if (n_pairs > 0) {
      seg(dM, mM, nullptr, n_pairs, capacity, 0);
      seg(dNm, mNm, nullptr, 1, n_pairs, 0);
      if (h_stats) seg(dSt, mSt, nullptr, 1, 3 * n_pairs, 0);
    }
    HIPCHK(launch_copy_out(c->st, co, ns, std::max(n_frames, n_pairs)));
  } else {
    HIPCHK(hipMemcpyAsync(h_kps, dK, sizeof(orbx_keypoint) * (size_t)capacity * n_frames, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipMemcpyAsync(h_desc32, dD, (size_t)32 * capacity * n_frames, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipMemcpyAsync(h_n_out, dN, sizeof(int32_t) * (size_t)n_frames, hipMemcpyDeviceToHost, c->st));
    if (n_pairs > 0) {
      HIPCHK(hipMemcpyAsync(h_matches12, dM, sizeof(int32_t) * (size_t)capacity * n_pairs, hipMemcpyDeviceToHost, c->st));
      HIPCHK(hipMemcpyAsync(h_nmatches, dNm, sizeof(int32_t) * (size_t)n_pairs, hipMemcpyDeviceToHost, c->st));
      if (h_stats) HIPCHK(hipMemcpyAsync(h_stats, dSt, sizeof(int32_t) * 3 * (size_t)n_pairs, hipMemcpyDeviceToHost, c->st));
    }
  }
  HIPCHK(hipEventRecord(c->evOut[par], c->st));
  c->outUsed[par] = true;
  return ORBX_OK;
}
}  // namespace
extern "C" {

int orbx_extract_match_batch_host_async(orbx_ctx* ctx, int n_frames, const uint8_t* h_imgs, int width, int height, int stride,
                                        size_t frame_stride_bytes, orbx_keypoint* h_kps, uint8_t* h_desc32, int capacity,
                                        int32_t* h_n_out, int n_pairs, const int32_t* h_first, const int32_t* h_second,
                                        const orbx_bounds* bounds, int window_size, float nnratio, int check_orientation,
                                        int32_t* h_matches12, int32_t* h_nmatches, int32_t* h_stats) {
  if (!ctx) return ORBX_E_BADARG;
  if (!h_imgs || width <= 0 || height <= 0) return ORBX_E_EMPTY;
  if (!h_kps || !h_desc32 || !h_n_out || stride < width || n_frames < 1 || n_pairs < 0 ||
      (n_pairs > 0 && (!h_first || !h_second || !bounds || !h_matches12 || !h_nmatches)))
    return ORBX_E_BADARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  if (!ctx->lanes.empty()) {  // pipeline mode: the whole batch to the next lane, at most one batch per lane in flight
    const unsigned L = (unsigned)ctx->lanes.size();
    if (ctx->laneIssue - ctx->laneDone >= L) {
      const int w = laneWaitOne(ctx);
      if (w != ORBX_OK) return w;
    }
    orbx_ctx* c = ctx->lanes[ctx->laneIssue % L];
    const int r = hostBatchIssue(c, n_frames, h_imgs, width, height, stride, frame_stride_bytes, h_kps, h_desc32, capacity, h_n_out,
                                 n_pairs, h_first, h_second, bounds, window_size, nnratio, check_orientation, h_matches12, h_nmatches,
                                 h_stats);
    if (r != ORBX_OK) { ctx->err = c->err; return r; }
    memcpy(ctx->lastLaunch, c->lastLaunch, sizeof ctx->lastLaunch);
    ctx->lastLaunch[7] = (int)(ctx->laneIssue % L) + 1;
    ctx->laneIssue++;
    return ORBX_OK;
  }
  ctx->lastLaunch[7] = 0;
  return hostBatchIssue(ctx, n_frames, h_imgs, width, height, stride, frame_stride_bytes, h_kps, h_desc32, capacity, h_n_out, n_pairs,
                        h_first, h_second, bounds, window_size, nnratio, check_orientation, h_matches12, h_nmatches, h_stats);
}

int orbx_set_pipeline_depth(orbx_ctx* ctx, int depth) {
  if (!ctx || depth < 0 || depth > 8) return ORBX_E_BADARG;
  if (depth > 0 && !ctx->ownStream) {  // the lanes run on streams of their own: a context tied to a caller's stream cannot promise its order
    ctx->err = "orbx_set_pipeline_depth: the context was created on a caller's stream";
    return ORBX_E_BADARG;
  }
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  const int w = waitAll(ctx);
  if (w != ORBX_OK) return w;  // (an earlier batch's error is returned by the call that has to wait for it; retry)
  for (orbx_ctx* c : ctx->lanes) orbx_destroy(c);
  ctx->lanes.clear();
  ctx->laneIssue = ctx->laneDone = 0;
  for (int i = 0; i < depth; i++) {
    orbx_ctx* c = nullptr;
    const int r = orbx_create(&ctx->p, ctx->device, ctx->maxW, ctx->maxH, ctx->maxB, nullptr, &c);
    if (r != ORBX_OK) {
      ctx->err = "orbx_set_pipeline_depth: a lane could not be created";
      for (orbx_ctx* q : ctx->lanes) orbx_destroy(q);
      ctx->lanes.clear();
      return r;
    }
    c->noSplit = true;
    c->eventOrdered = ctx->eventOrdered;
    c->gaussVariant = ctx->gaussVariant;
    c->grayVariant = ctx->grayVariant;
    c->profMask = ctx->profMask;
    ctx->lanes.push_back(c);
    {  // (the lane was created with the default reading: its quotas follow the parent's)
      const int rl = applyLibmVariant(c, ctx->libmVariant);
      if (rl != ORBX_OK) { ctx->err = c->err; return rl; }
    }
  }
  return ORBX_OK;
}

int orbx_wait_one(orbx_ctx* ctx) {
  if (!ctx) return ORBX_E_BADARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  if (ctx->laneDone != ctx->laneIssue) return laneWaitOne(ctx);
  return waitOldest(ctx);
}

int orbx_wait(orbx_ctx* ctx) {
  if (!ctx) return ORBX_E_BADARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  return waitAll(ctx);
}

// Stream ordering against a caller's stream (e.g. torch's current stream): the context works on its own two streams.
int orbx_order_after(orbx_ctx* ctx, void* stream) {
  if (!ctx) return ORBX_E_BADARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  hipStream_t s = (hipStream_t)stream;
  if (s == ctx->st) return ORBX_OK;
  HIPCHK(hipEventRecord(ctx->evOrder, s));
  HIPCHK(hipStreamWaitEvent(ctx->st, ctx->evOrder, 0));
  if (ctx->st2) HIPCHK(hipStreamWaitEvent(ctx->st2, ctx->evOrder, 0));
  for (orbx_ctx* c : ctx->lanes) HIPCHK(hipStreamWaitEvent(c->st, ctx->evOrder, 0));
  return ORBX_OK;
}

int orbx_order_before(orbx_ctx* ctx, void* stream) {
  if (!ctx) return ORBX_E_BADARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  hipStream_t s = (hipStream_t)stream;
  if (!ctx->eventOrdered) {
    // From now on a consumer may read outputs behind an event only, so nothing of a batch may be left to its host-side wait:
    // the wide matcher kernels travel with every batch of this context (armMatch).  Batches already in flight that were issued
    // without them are completed here, once, with a host-side wait (settleMatch runs their late wide kernels if they need them).
    ctx->eventOrdered = true;
    for (orbx_ctx* c : ctx->lanes) c->eventOrdered = true;
    std::vector<orbx_ctx*> all(ctx->lanes);
    all.push_back(ctx);
    for (orbx_ctx* c : all)
      for (int par = 0; par < 2; par++)
        if (c->late[par].valid && !c->wideLaunched[par]) {
          HIPCHK(hipEventSynchronize(c->evDone[par]));
          if (c->done2Used[par]) HIPCHK(hipEventSynchronize(c->evDone2[par]));
          const int sm = settleMatch(c, par);
          if (sm != ORBX_OK) { ctx->err = c->err; return sm; }
        }
  }
  if (s != ctx->st) {
    HIPCHK(hipEventRecord(ctx->evOrder, ctx->st));
    HIPCHK(hipStreamWaitEvent(s, ctx->evOrder, 0));
  }
  if (ctx->st2) {
    HIPCHK(hipEventRecord(ctx->evOrder, ctx->st2));
    HIPCHK(hipStreamWaitEvent(s, ctx->evOrder, 0));
  }
  for (orbx_ctx* c : ctx->lanes) {
    HIPCHK(hipEventRecord(c->evOrder, c->st));
    HIPCHK(hipStreamWaitEvent(s, c->evOrder, 0));
  }
  return ORBX_OK;
}

int orbx_match_init(orbx_ctx* ctx, const orbx_keypoint* k1, const uint8_t* d1, int n1, const orbx_keypoint* k2, const uint8_t* d2,
                    int n2, const orbx_bounds* bounds, int window_size, float nnratio, int check_orientation, int32_t* matches12,
                    int32_t* nmatches, orbx_match_stats* stats) {
  if (!ctx || n1 < 0 || n2 < 0 || !bounds || !nmatches || (n1 > 0 && (!k1 || !d1 || !matches12)) || (n2 > 0 && (!k2 || !d2))) return ORBX_E_BADARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  const size_t cap = (size_t)std::max(std::max(n1, n2), 1);
  if (cap >= (1u << 20)) return ORBX_E_BADARG;
  int er = ensureHostPairBuffers(ctx, cap);
  if (er != ORBX_OK) return er;
  const size_t c = ctx->mCap;
  hipStream_t st = ctx->st;
  // the pair is packed into the page-locked mirror of the device block and goes up with one copy command; the matcher
  // kernels store matches, count and statistics straight into mapped page-locked words (0.13 -> 0.09 ms per call)
  uint8_t* hb = ctx->hMblk;
  if (n1) {
    memcpy(hb, k1, sizeof(orbx_keypoint) * n1);
    memcpy(hb + ctx->mDescOff, d1, (size_t)32 * n1);
  }
  if (n2) {
    memcpy(hb + c * sizeof(orbx_keypoint), k2, sizeof(orbx_keypoint) * n2);
    memcpy(hb + ctx->mDescOff + c * 32, d2, (size_t)32 * n2);
  }
  int* hn = reinterpret_cast<int*>(hb + ctx->mCntOff);
  hn[0] = n1; hn[1] = n2;
  // (the block is private to the host-array entry points, which are synchronous: nothing in flight reads it)
  HIPCHK(hipMemcpyAsync(ctx->dMblk, hb, ctx->mBlkBytes, hipMemcpyHostToDevice, st));
  int* dN = reinterpret_cast<int*>(ctx->dMblk + ctx->mCntOff);  // [2]
  int* oNm = ctx->hMoDev + 2;     // [1]
  int* oSt = ctx->hMoDev + 3;     // [3]
  int* oM12 = ctx->hMoDev + 8;    // [cap]
  const int32_t first = 0, second = 1;
  int r = orbx_match_init_batch_device(ctx, 1, &first, &second, ctx->dMk, ctx->dMd, dN, (int)c, bounds, window_size, nnratio,
                                       check_orientation, oM12, oNm, oSt);
  if (r != ORBX_OK) return r;
  // (orbx_match_init_batch_device has synchronised the stream: the results are in host memory)
  const int* res = ctx->hMo + 2;
  if (n1) memcpy(matches12, ctx->hMo + 8, sizeof(int) * n1);
  if (stats) { stats->invalid_by_distance = res[1]; stats->invalid_by_ratio = res[2]; stats->invalid_by_orientation = res[3]; }
  *nmatches = res[0];
  return ORBX_OK;
}

// ---- Frame::UndistortKeyPoints / ComputeImageBounds (SlamTypes/Frame.cpp:101-161) ----------------
int orbx_undistort_batch_device(orbx_ctx* ctx, int n_frames, const orbx_keypoint* d_kps, const int32_t* d_n, int capacity,
                                const orbx_camera* cam, orbx_keypoint* d_kps_un) {
  if (!ctx || n_frames < 0 || !d_kps || !d_n || !cam || !d_kps_un || capacity < 1) return ORBX_E_BADARG;
  if (n_frames == 0) return ORBX_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  HIPCHK(launch_undistort(ctx->st, n_frames, d_kps, d_n, capacity, makeCam(*cam), d_kps_un));
  HIPCHK(hipStreamSynchronize(ctx->st));
  return ORBX_OK;
}

int orbx_undistort_keypoints(orbx_ctx* ctx, const orbx_keypoint* kps, int n, const orbx_camera* cam, orbx_keypoint* out) {
  if (!ctx || n < 0 || !cam || (n > 0 && (!kps || !out))) return ORBX_E_BADARG;
  if (n == 0) return ORBX_OK;
  if (n >= (1 << 20)) return ORBX_E_BADARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  int er = ensureHostPairBuffers(ctx, (size_t)n);
  if (er != ORBX_OK) return er;
  const size_t c = ctx->mCap;
  hipStream_t st = ctx->st;
  HIPCHK(hipMemcpyAsync(ctx->dMk, kps, sizeof(orbx_keypoint) * n, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(ctx->dMi, &n, sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(launch_undistort(st, 1, ctx->dMk, ctx->dMi, (int)c, makeCam(*cam), ctx->dMk + c));
  HIPCHK(hipMemcpyAsync(out, ctx->dMk + c, sizeof(orbx_keypoint) * n, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return ORBX_OK;
}

int orbx_image_bounds(orbx_ctx* ctx, const orbx_camera* cam, int width, int height, orbx_bounds* out) {
  if (!ctx || !cam || !out || width <= 0 || height <= 0) return ORBX_E_BADARG;
  if (cam->k1 == 0.0f) {  // Frame.cpp:127-133
    out->min_x = 0; out->max_x = width; out->min_y = 0; out->max_y = height;
    return ORBX_OK;
  }
  // Frame.cpp:105-113: corners (0,0) (cols,0) (0,rows) (cols,rows) through the same undistortion kernel
  orbx_keypoint c4[4]{}, u4[4]{};
  c4[1].x = (float)width; c4[2].y = (float)height; c4[3].x = (float)width; c4[3].y = (float)height;
  int r = orbx_undistort_keypoints(ctx, c4, 4, cam, u4);
  if (r != ORBX_OK) return r;
  out->min_x = (int)std::min(u4[0].x, u4[2].x);  // Frame.cpp:121-124, float -> static int
  out->max_x = (int)std::max(u4[1].x, u4[3].x);
  out->min_y = (int)std::min(u4[0].y, u4[1].y);
  out->max_y = (int)std::max(u4[2].y, u4[3].y);
  return ORBX_OK;
}

// ---- Initializer scoring loops (Initialization/Initializer.cpp:268-438) ---------------------------
namespace {
int checkModels(orbx_ctx* ctx, int kind, int n_models, const float* M21, const float* M12, const orbx_keypoint* k1, int n1,
                const orbx_keypoint* k2, int n2, const int32_t* matches12, float sigma, float* scores, uint8_t* inliers,
                int* n_matches_out, int* best) {
  if (!ctx || n_models < 0 || n1 < 0 || n2 < 0 || !n_matches_out || (n_models > 0 && (!M21 || (kind == 0 && !M12) || !scores)) ||
      (n1 > 0 && (!k1 || !matches12)) || (n2 > 0 && !k2))
    return ORBX_E_BADARG;
  // mvMatches12, Initializer.cpp:24-33
  std::vector<int32_t> fs;
  fs.reserve(2 * (size_t)n1);
  for (int i = 0; i < n1; i++)
    if (matches12[i] >= 0) {
      if (matches12[i] >= n2) return ORBX_E_BADARG;
      fs.push_back(i);
    }
  const int N = (int)fs.size();
  for (int i = 0; i < N; i++) fs.push_back(matches12[fs[i]]);
  *n_matches_out = N;
  if (best) *best = -1;
  if (n_models == 0) return ORBX_OK;
  if (N > 0 && !inliers) return ORBX_E_BADARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  // one staging allocation: models | keypoints | pairs | scores | inliers
  const size_t bM = (size_t)n_models * 9 * sizeof(float), bK1 = (size_t)n1 * sizeof(orbx_keypoint),
               bK2 = (size_t)n2 * sizeof(orbx_keypoint), bP = (size_t)N * sizeof(int32_t), bS = (size_t)n_models * sizeof(float),
               bI = (size_t)n_models * N;
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t need = 2 * al(bM) + al(bK1) + al(bK2) + 2 * al(bP) + al(bS) + al(bI) + 256;
  if (need > ctx->scoreBytes) {
    if (ctx->dScore) (void)hipFree(ctx->dScore);
    ctx->dScore = nullptr; ctx->scoreBytes = 0;
    HIPCHK(hipMalloc((void**)&ctx->dScore, need));
    ctx->scoreBytes = need;
  }
  uint8_t* p = ctx->dScore;
  ScoreArgs a{};
  hipStream_t st = ctx->st;
  a.M21 = (const float*)p; p += al(bM);
  a.M12 = (const float*)p; p += al(bM);
  a.k1 = (const orbx_keypoint*)p; p += al(bK1);
  a.k2 = (const orbx_keypoint*)p; p += al(bK2);
  a.first = (const int32_t*)p; p += al(bP);
  a.second = (const int32_t*)p; p += al(bP);
  a.scores = (float*)p; p += al(bS);
  a.inliers = p;
  a.N = N; a.kind = kind;
  a.invSigmaSquare = (float)(1.0 / (double)(sigma * sigma));  // `const float invSigmaSquare = 1.0 / (sigma * sigma)`
  HIPCHK(hipMemcpyAsync((void*)a.M21, M21, bM, hipMemcpyHostToDevice, st));
  if (kind == 0) HIPCHK(hipMemcpyAsync((void*)a.M12, M12, bM, hipMemcpyHostToDevice, st));
  if (n1) HIPCHK(hipMemcpyAsync((void*)a.k1, k1, bK1, hipMemcpyHostToDevice, st));
  if (n2) HIPCHK(hipMemcpyAsync((void*)a.k2, k2, bK2, hipMemcpyHostToDevice, st));
  if (N) {
    HIPCHK(hipMemcpyAsync((void*)a.first, fs.data(), bP, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync((void*)a.second, fs.data() + N, bP, hipMemcpyHostToDevice, st));
  }
  HIPCHK(launch_check_model(st, n_models, a));
  HIPCHK(hipMemcpyAsync(scores, a.scores, bS, hipMemcpyDeviceToHost, st));
  if (N) HIPCHK(hipMemcpyAsync(inliers, a.inliers, bI, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (best) {  // `if (currentScore > score)` with score starting at 0, Initializer.cpp:205-209 / 259-263
    float sc = 0;
    for (int m = 0; m < n_models; m++)
      if (scores[m] > sc) { sc = scores[m]; *best = m; }
  }
  return ORBX_OK;
}
}  // namespace

int orbx_check_homography(orbx_ctx* ctx, int n_models, const float* H21, const float* H12, const orbx_keypoint* k1, int n1,
                          const orbx_keypoint* k2, int n2, const int32_t* matches12, float sigma, float* scores, uint8_t* inliers,
                          int* n_matches_out, int* best) {
  return checkModels(ctx, 0, n_models, H21, H12, k1, n1, k2, n2, matches12, sigma, scores, inliers, n_matches_out, best);
}

int orbx_check_fundamental(orbx_ctx* ctx, int n_models, const float* F21, const orbx_keypoint* k1, int n1, const orbx_keypoint* k2,
                           int n2, const int32_t* matches12, float sigma, float* scores, uint8_t* inliers, int* n_matches_out,
                           int* best) {
  return checkModels(ctx, 1, n_models, F21, nullptr, k1, n1, k2, n2, matches12, sigma, scores, inliers, n_matches_out, best);
}

// ---- Initializer::CheckRT (Initialization/Initializer.cpp:569-713) -------------------------------------
int orbx_check_rt(orbx_ctx* ctx, int n_models, const float* R21, const float* t21, const float* K, const orbx_keypoint* k1, int n1,
                  const orbx_keypoint* k2, int n2, const int32_t* matches12, const uint8_t* matches_inliers, float th2, int32_t* n_good,
                  uint8_t* tri_good, float* p3d, float* parallax) {
  if (!ctx || n_models < 0 || n1 < 0 || n2 < 0 || !K || (n_models > 0 && (!R21 || !t21 || !n_good || !parallax)) ||
      (n1 > 0 && (!k1 || !matches12)) || (n2 > 0 && !k2) || (n_models > 0 && n1 > 0 && (!tri_good || !p3d)))
    return ORBX_E_BADARG;
  if (n_models == 0) return ORBX_OK;
  // mvMatches12 (Initializer.cpp:24-33), then the inliers in match order (:617-622); the i-th of them is booked under the
  // i-th MATCH's first keypoint (:643, :700: the reference indexes vMatches12 with the compacted index)
  std::vector<int32_t> fs, sc;
  for (int i = 0; i < n1; i++)
    if (matches12[i] >= 0) {
      if (matches12[i] >= n2) return ORBX_E_BADARG;
      fs.push_back(i);
      sc.push_back(matches12[i]);
    }
  const int N = (int)fs.size();
  if (N > 0 && !matches_inliers) return ORBX_E_BADARG;
  std::vector<float> pts;
  std::vector<int32_t> book;
  for (int m = 0; m < N; m++)
    if (matches_inliers[m]) {
      const int i = (int)book.size();
      book.push_back(fs[i]);
      pts.push_back(k1[fs[m]].x); pts.push_back(k1[fs[m]].y); pts.push_back(k2[sc[m]].x); pts.push_back(k2[sc[m]].y);
    }
  const int nInl = (int)book.size();
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t bR = (size_t)n_models * 9 * 4, bT = (size_t)n_models * 3 * 4, bP = (size_t)nInl * 16, bB = (size_t)nInl * 4,
               bG = (size_t)n_models * n1, bX = (size_t)n_models * n1 * 12, bC = (size_t)n_models * nInl * 4, bN = (size_t)n_models * 4;
  const size_t need = al(bR) + al(bT) + al(bP) + al(bB) + al(bG) + al(bX) + al(bC) + 2 * al(bN) + 256;
  if (need > ctx->scoreBytes) {
    if (ctx->dScore) (void)hipFree(ctx->dScore);
    ctx->dScore = nullptr; ctx->scoreBytes = 0;
    HIPCHK(hipMalloc((void**)&ctx->dScore, need));
    ctx->scoreBytes = need;
  }
  uint8_t* p = ctx->dScore;
  CheckRtArgs a{};
  hipStream_t st = ctx->st;
  a.R21 = (const float*)p; p += al(bR);
  a.t21 = (const float*)p; p += al(bT);
  a.pts = (const float*)p; p += al(bP);
  a.book = (const int32_t*)p; p += al(bB);
  a.good = p; p += al(bG);
  a.p3d = (float*)p; p += al(bX);
  a.cosBuf = (float*)p; p += al(bC);
  a.nGood = (int32_t*)p; p += al(bN);
  a.parallax = (float*)p;
  for (int i = 0; i < 9; i++) a.K[i] = K[i];
  a.th2 = th2; a.nInl = nInl; a.n1 = n1;
  HIPCHK(hipMemcpyAsync((void*)a.R21, R21, bR, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync((void*)a.t21, t21, bT, hipMemcpyHostToDevice, st));
  if (nInl) {
    HIPCHK(hipMemcpyAsync((void*)a.pts, pts.data(), bP, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync((void*)a.book, book.data(), bB, hipMemcpyHostToDevice, st));
  }
  HIPCHK(launch_check_rt(st, n_models, a));
  HIPCHK(hipMemcpyAsync(n_good, a.nGood, bN, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(parallax, a.parallax, bN, hipMemcpyDeviceToHost, st));
  if (n1) {
    HIPCHK(hipMemcpyAsync(tri_good, a.good, bG, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(p3d, a.p3d, bX, hipMemcpyDeviceToHost, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  return ORBX_OK;
}

// ---- Converter::toGray (Utils/Converter.cpp:5-19) ------------------------------------------------
int orbx_to_gray_batch_device(orbx_ctx* ctx, int n_frames, const uint8_t* d_src, int width, int height, int stride,
                              size_t frame_stride_bytes, int channels, int rgb, uint8_t* d_gray, int gray_stride,
                              size_t gray_frame_stride_bytes) {
  if (!ctx) return ORBX_E_BADARG;
  if (!d_src || width <= 0 || height <= 0) return ORBX_E_EMPTY;
  if (channels != 1 && channels != 3) { ctx->err = "Wrong image format"; return ORBX_E_BADARG; }  // Converter.cpp:17
  if (!d_gray || n_frames < 0 || n_frames > 65535 || height > 65535 || (long long)stride < (long long)width * channels ||
      gray_stride < width)
    return ORBX_E_BADARG;
  if (n_frames == 0) return ORBX_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  HIPCHK(launch_to_gray(ctx->st, n_frames, d_src, (long long)frame_stride_bytes, stride, width, height, channels, rgb, d_gray,
                        (long long)gray_frame_stride_bytes, gray_stride, ctx->grayVariant));
  HIPCHK(hipStreamSynchronize(ctx->st));
  return ORBX_OK;
}

int orbx_to_gray(orbx_ctx* ctx, const uint8_t* img, int width, int height, int stride, int channels, int rgb, uint8_t* gray,
                 int gray_stride) {
  if (!ctx) return ORBX_E_BADARG;
  if (!img || width <= 0 || height <= 0) return ORBX_E_EMPTY;
  if (channels != 1 && channels != 3) { ctx->err = "Wrong image format"; return ORBX_E_BADARG; }
  if (!gray || height > 65535 || (long long)stride < (long long)width * channels || gray_stride < width) return ORBX_E_BADARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  const int sstride = alignUp(width * channels, 64), gstride = alignUp(width, 64);
  const size_t need = ((size_t)sstride + gstride) * height;
  if (need > ctx->colorBytes) {
    if (ctx->dColor) (void)hipFree(ctx->dColor);
    ctx->dColor = nullptr; ctx->colorBytes = 0;
    HIPCHK(hipMalloc((void**)&ctx->dColor, need));
    ctx->colorBytes = need;
  }
  uint8_t* dG = ctx->dColor + (size_t)sstride * height;
  HIPCHK(hipMemcpy2DAsync(ctx->dColor, sstride, img, stride, (size_t)width * channels, height, hipMemcpyHostToDevice, ctx->st));
  HIPCHK(launch_to_gray(ctx->st, 1, ctx->dColor, 0, sstride, width, height, channels, rgb, dG, 0, gstride, ctx->grayVariant));
  HIPCHK(hipMemcpy2DAsync(gray, gray_stride, dG, gstride, width, height, hipMemcpyDeviceToHost, ctx->st));
  HIPCHK(hipStreamSynchronize(ctx->st));
  return ORBX_OK;
}

// ---- measurement hooks ------------------------------------------------------------------------
int orbx_profile_enable(orbx_ctx* ctx, int on) {
  if (!ctx) return ORBX_E_BADARG;
  ctx->profMask = on ? (1u << ORBX_STAGE_COUNT) - 1u : 0u;
  for (orbx_ctx* c : ctx->lanes) c->profMask = ctx->profMask;
  return ORBX_OK;
}
int orbx_profile_stages(orbx_ctx* ctx, unsigned stage_mask) {
  if (!ctx) return ORBX_E_BADARG;
  ctx->profMask = stage_mask & ((1u << ORBX_STAGE_COUNT) - 1u);
  for (orbx_ctx* c : ctx->lanes) c->profMask = ctx->profMask;
  return ORBX_OK;
}
int orbx_profile_reset(orbx_ctx* ctx) {
  if (!ctx) return ORBX_E_BADARG;
  for (int s = 0; s < ORBX_STAGE_COUNT; s++) { ctx->ms[s] = 0; ctx->launches[s] = 0; }
  for (orbx_ctx* c : ctx->lanes)
    for (int s = 0; s < ORBX_STAGE_COUNT; s++) { c->ms[s] = 0; c->launches[s] = 0; }
  return ORBX_OK;
}
int orbx_profile_get(orbx_ctx* ctx, double* ms, int64_t* launches) {
  if (!ctx) return ORBX_E_BADARG;
  for (int s = 0; s < ORBX_STAGE_COUNT; s++) {  // (this context's own calls plus what its lanes ran)
    double m = ctx->ms[s];
    int64_t n = ctx->launches[s];
    for (const orbx_ctx* c : ctx->lanes) { m += c->ms[s]; n += c->launches[s]; }
    if (ms) ms[s] = m;
    if (launches) launches[s] = n;
  }
  return ORBX_OK;
}

// ---- test hooks ---------------------------------------------------------------------------------
int orbx_debug_candidates(orbx_ctx* ctx, int frame, int level, float* xyr, int cap) {
  if (!ctx || frame < 0 || frame >= ctx->lastB || level < 0 || level >= ctx->p.nlevels) return ORBX_E_BADARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  {
    const int w = waitAll(ctx);
    if (w == ORBX_E_HIP) return w;
  }
  const LevelGeom& L = ctx->g.L[level];
  // gather the cells' segments (k_fast writes every cell's survivors into the cell's own segment)
  const int nCells = L.nCols * L.nRows;
  std::vector<int> cc(std::max(nCells, 1));
  HIPCHK(hipMemcpy(cc.data(), ctx->dCellCount + (size_t)frame * ctx->g.nCellsTotal + L.cellBase, sizeof(int) * (size_t)nCells,
                   hipMemcpyDeviceToHost));
  std::vector<uint32_t> seg((size_t)std::max(L.candCap, 1));
  HIPCHK(hipMemcpy(seg.data(), ctx->dCand + L.candOff + (int64_t)frame * L.candCap, (size_t)L.candCap * 4, hipMemcpyDeviceToHost));
  std::vector<uint32_t> e;
  for (int c = 0; c < nCells; c++)
    for (int k = 0; k < cc[c]; k++) e.push_back(seg[(size_t)c * L.segCap + k]);
  const int cnt = (int)e.size();
  std::vector<uint64_t> keyed(cnt);
  for (int i = 0; i < cnt; i++) keyed[i] = (candOrderKey(L, e[i]) << 8) | (e[i] >> 24);
  std::sort(keyed.begin(), keyed.end());
  for (int i = 0; i < cnt && i < cap; i++) {
    xyr[3 * i] = (float)((keyed[i] >> 8) & 0xfff);
    xyr[3 * i + 1] = (float)((keyed[i] >> 20) & 0xfff);
    xyr[3 * i + 2] = (float)(keyed[i] & 0xff);
  }
  return cnt;
}

int orbx_debug_selection_units(orbx_ctx* ctx, int frame, int32_t* counts, int32_t* redone) {
  if (!ctx || frame < 0 || frame >= ctx->lastB || !counts) return ORBX_E_BADARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  {
    const int w = waitAll(ctx);
    if (w == ORBX_E_HIP) return w;
  }
  const int nl = ctx->p.nlevels;
  std::vector<int32_t> raw((size_t)nl);
  HIPCHK(hipMemcpy(raw.data(), ctx->dNselLevel + (size_t)frame * nl, sizeof(int32_t) * (size_t)nl, hipMemcpyDeviceToHost));
  for (int l = 0; l < nl; l++) {
    counts[l] = raw[l] < 0 ? raw[l] : (raw[l] & ~ORBX_OCT_REDONE);
    if (redone) redone[l] = raw[l] >= 0 && (raw[l] & ORBX_OCT_REDONE) ? 1 : 0;
  }
  return ORBX_OK;
}

}  // extern "C"

// ---- device-side test hooks for the selection stage ------------------------------------------------------------
namespace orbx {
hipError_t launch_debug_sort(hipStream_t st, int* triples, int n, unsigned long long* a, unsigned long long* b);
}

extern "C" {

// DistributeOctTree on the device for caller-supplied candidates given in row-major (y, x) order, integer coordinates
// relative to (min_x, min_y) in [0, 4095], integer responses in [0, 255].  variant 0 = LDS kernel (falls through to
// the global-scratch kernel when it cannot take the unit), 1 = global-scratch kernel only, 2 / 3 = the 1024- / 512-candidate
// LDS instances (a unit that does not fit is redone by the same workgroup on global scratch), 4 = the 2048-candidate LDS instance
// with 64-bit sort keys (the others take 32-bit keys whenever the rectangle's path codes fit 21 bits); 5 = the many-workgroup
// kernels of large units (k_octree_buckets + k_octree_big, k_octree_global behind them for a unit they hand on), 6 = those two
// alone (ORBX_E_CAPACITY when they hand the unit on).
int orbx_debug_distribute_device(orbx_ctx* ctx, const float* xyr, int n, int min_x, int max_x, int min_y, int max_y,
                                 int n_features, int variant, float* out_xyr, int cap) {
  if (!ctx || n < 0 || (n > 0 && !xyr) || max_x <= min_x || max_y <= min_y || n_features < 0 || n > ORBX_OCT_MAX_CAND)
    return ORBX_E_BADARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  OctLaunch P{};
  P.nlevels = 1;
  P.selStride = std::max(n_features, 1);
  OctLevel& O = P.lev[0];
  O.width = max_x - min_x;
  O.height = max_y - min_y;
  O.nIni = (int)std::round((float)O.width / (float)O.height);
  if (O.nIni < 1 || O.nIni > 255) return ORBX_E_TOOSMALL;
  O.hX = (float)O.width / (float)O.nIni;
  O.depthBits = octDepthBits(O.height, O.hX);
  O.tabW = O.tabH = 4096;       // any coordinate the hook accepts
  O.wCell = O.hCell = 1 << 20;  // one "cell": candidate order = row-major
  O.nCols = 1;
  O.quota = n_features;
  O.cellBase = 0; O.nCells = 1; O.segCap = alignUp(std::max(n, 1), 4);  // all candidates in the segment of one cell
  P.nCellsTotal = 1;
  P.candCap[0] = std::max(n, 1);
  // (the pipeline sizes a unit's scratch for the level's worst case; here: eight times the candidates at hand, so that the
  // buckets of the many-workgroup kernels get slots with the same kind of headroom)
  P.scrNMax[0] = std::min(std::max(8 * n, 16384), ORBX_OCT_MAX_CAND);
  P.scrStride[0] = (int64_t)octScratchBytes(P.scrNMax[0], std::max(n_features, O.nIni));  // = octreeGlobalUnit's qMax
  std::vector<uint32_t> packed((size_t)alignUp(std::max(n, 1), 4));
  for (int i = 0; i < n; i++) {
    const int x = (int)xyr[3 * i], y = (int)xyr[3 * i + 1], r = (int)xyr[3 * i + 2];
    if (x < 0 || x > 4095 || y < 0 || y > 4095 || r < 0 || r > 255) return ORBX_E_BADARG;
    packed[i] = packCand(x, y, r);
  }
  O.tabW = std::max(O.tabW, O.width);  // (the bucket tables of the many-workgroup path walk the region's own columns / rows)
  O.tabH = std::max(O.tabH, O.height);
  octBigPlan(&O, P.scrNMax[0]);
  octTabLayout(&P);
  std::vector<uint32_t> codeTab;
  octCodeTables(&P, &codeTab);
  uint32_t* dC = nullptr;
  uint32_t* dT = nullptr;
  int* dI = nullptr;
  SelKp* dS = nullptr;
  uint8_t* dScr = nullptr;
  std::vector<SelKp> sel(std::max(n_features, 1));
  int res[2] = {0, 0};
  auto body = [&]() -> int {
    HIPCHK(hipMalloc((void**)&dC, packed.size() * 4));
    HIPCHK(hipMalloc((void**)&dI, 2 * sizeof(int)));
    HIPCHK(hipMalloc((void**)&dS, sel.size() * sizeof(SelKp)));
    HIPCHK(hipMalloc((void**)&dScr, (size_t)P.scrStride[0]));
    HIPCHK(hipMalloc((void**)&dT, codeTab.size() * 4));
    P.codeTab = dT;
    HIPCHK(hipMemcpyAsync(dT, codeTab.data(), codeTab.size() * 4, hipMemcpyHostToDevice, ctx->st));
    HIPCHK(hipMemcpyAsync(dC, packed.data(), packed.size() * 4, hipMemcpyHostToDevice, ctx->st));
    int hi[2] = {n, -7};
    HIPCHK(hipMemcpyAsync(dI, hi, sizeof hi, hipMemcpyHostToDevice, ctx->st));
    const int hintN[ORBX_MAX_LEVELS] = {n};  // (variant 5: the bucket depth from the candidate count, as in the pipeline; 6: from the area)
    HIPCHK(launch_octree(ctx->st, 1, dC, dI, P, dS, dI + 1, dScr, nullptr, variant == 5 ? hintN : nullptr,
                         variant == 1 ? -1 : variant == 2 ? 1024 : variant == 3 ? 512 : variant == 4 ? (2048 | 0x10000) : variant == 5 ? -2 :
                         variant == 6 ? -3 : 0, nullptr));
    HIPCHK(hipMemcpyAsync(res, dI, sizeof res, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(sel.data(), dS, sel.size() * sizeof(SelKp), hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    return ORBX_OK;
  };
  const int rc = body();
  if (dC) (void)hipFree(dC);
  if (dI) (void)hipFree(dI);
  if (dS) (void)hipFree(dS);
  if (dScr) (void)hipFree(dScr);
  if (dT) (void)hipFree(dT);
  if (rc != ORBX_OK) return rc;
  if (res[1] < 0) return ORBX_E_CAPACITY;
  res[1] &= ~0x40000000;  // (ORBX_OCT_REDONE: the many-workgroup kernels handed the unit to the one-workgroup code)
  for (int i = 0; i < res[1] && i < cap; i++) {
    out_xyr[3 * i] = (float)(sel[i].x - ORBX_MIN_BORDER);
    out_xyr[3 * i + 1] = (float)(sel[i].y - ORBX_MIN_BORDER);
    out_xyr[3 * i + 2] = (float)sel[i].response;
  }
  return res[1];
}

// Host only (no device is touched): the path codes of n points of a width x height candidate region, once from the per-level
// tables the selection kernels look up (octCodeTable: x and y digits separately) and once by walking DivideNode's 16 splits
// for the point (cpp:617-676, 747) -- the two must agree for every point.  Codes: root << 32 | 16 quadrant digits.
int orbx_debug_path_codes(int width, int height, int n, const int32_t* xs, const int32_t* ys, uint64_t* from_tables,
                          uint64_t* from_walk) {
  if (width <= 0 || height <= 0 || width > 4096 || height > 4096 || n < 0 || (n > 0 && (!xs || !ys || !from_tables || !from_walk)))
    return ORBX_E_BADARG;
  OctLevel O{};
  O.width = width;
  O.height = height;
  O.nIni = (int)std::round((float)width / (float)height);  // cpp:706
  if (O.nIni < 1 || O.nIni > 255) return ORBX_E_TOOSMALL;
  O.hX = (float)width / (float)O.nIni;                      // cpp:709
  O.tabW = width;
  O.tabH = height;
  std::vector<uint32_t> tab((size_t)2 * width + height + 2);
  octCodeTable(O, tab.data());
  for (int i = 0; i < n; i++) {
    const int xi = xs[i], yi = ys[i];
    if (xi < 0 || xi >= width || yi < 0 || yi >= height) return ORBX_E_BADARG;
    from_tables[i] = ((uint64_t)tab[2 * (size_t)xi + 1] << 32) | (uint64_t)(tab[2 * (size_t)xi] | tab[2 * (size_t)width + yi]);
    const float x = (float)xi, y = (float)yi;
    int root = (int)(x / O.hX);
    root = std::min(std::max(root, 0), O.nIni - 1);
    int ulx = (int)(O.hX * (float)root), brx = (int)(O.hX * (float)(root + 1)), uly = 0, bry = height;
    uint64_t code = (uint64_t)root;
    for (int d = 0; d < 16; d++) {  // DivideNode: half = ceil(extent / 2); the point goes right / down unless it is < the middle
      const int midX = ulx + ((brx - ulx + 1) >> 1), midY = uly + ((bry - uly + 1) >> 1);
      const int qx = !(x < (float)midX), qy = !(y < (float)midY);
      if (qx) ulx = midX; else brx = midX;
      if (qy) uly = midY; else bry = midY;
      code = (code << 2) | (uint64_t)(qy * 2 + qx);
    }
    from_walk[i] = code;
  }
  return ORBX_OK;
}

int orbx_debug_set(const char* key, long long value) {
  static const char* const names[KNOB_COUNT] = {
#define ORBX_KNOB_NAME(e, n) n,
      ORBX_KNOB_LIST(ORBX_KNOB_NAME)
#undef ORBX_KNOB_NAME
  };
  if (!key) return ORBX_E_BADARG;
  for (int k = 0; k < KNOB_COUNT; k++)
    if (std::strcmp(key, names[k]) == 0) {
      g_knob[k].store(value, std::memory_order_relaxed);
      return ORBX_OK;
    }
  return ORBX_E_BADARG;
}

int orbx_debug_last_launch(const orbx_ctx* ctx, int32_t* info8) {
  if (!ctx || !info8) return ORBX_E_BADARG;
  for (int i = 0; i < 8; i++) info8[i] = ctx->lastLaunch[i];
  return ORBX_OK;
}

int orbx_debug_match_counters(orbx_ctx* ctx, uint32_t* info4) {
  if (!ctx || !info4) return ORBX_E_BADARG;
  for (int i = 0; i < 4; i++) info4[i] = 0;
  if (!ctx->dMatchDiag) return ORBX_OK;  // no match has run on this context yet
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  const int w = waitAll(ctx);  // (the counters of every batch issued so far)
  if (w != ORBX_OK) return w;
  HIPCHK(hipStreamSynchronize(ctx->st));
  HIPCHK(hipMemcpy(info4, ctx->dMatchDiag, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return ORBX_OK;
}

int orbx_debug_sincos(orbx_ctx* ctx, const float* angle_deg, int n, float* cos_out, float* sin_out) {
  if (!ctx || n < 0 || (n > 0 && (!angle_deg || !cos_out || !sin_out))) return ORBX_E_BADARG;
  if (n == 0) return ORBX_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  float* d = nullptr;
  auto body = [&]() -> int {
    HIPCHK(hipMalloc((void**)&d, (size_t)n * 3 * sizeof(float)));
    HIPCHK(hipMemcpyAsync(d, angle_deg, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->st));
    HIPCHK(launch_debug_sincos(ctx->st, d, n, d + n, d + 2 * (size_t)n, ctx->libmVariant));
    HIPCHK(hipMemcpyAsync(cos_out, d + n, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(sin_out, d + 2 * (size_t)n, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    return ORBX_OK;
  };
  const int rc = body();
  if (d) (void)hipFree(d);
  return rc;
}

int orbx_debug_std_sort(orbx_ctx* ctx, int32_t* triples, int n) {
  if (!ctx || n < 0 || (n > 0 && !triples)) return ORBX_E_BADARG;
  if (n == 0) return ORBX_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return ORBX_E_HIP;
  int* d = nullptr;
  unsigned long long* w = nullptr;
  auto body = [&]() -> int {
    HIPCHK(hipMalloc((void**)&d, (size_t)n * 3 * sizeof(int)));
    HIPCHK(hipMalloc((void**)&w, (size_t)n * 2 * sizeof(unsigned long long)));
    HIPCHK(hipMemcpyAsync(d, triples, (size_t)n * 3 * sizeof(int), hipMemcpyHostToDevice, ctx->st));
    HIPCHK(launch_debug_sort(ctx->st, d, n, w, w + n));
    HIPCHK(hipMemcpyAsync(triples, d, (size_t)n * 3 * sizeof(int), hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    return ORBX_OK;
  };
  const int rc = body();
  if (d) (void)hipFree(d);
  if (w) (void)hipFree(w);
  return rc;
}

}  // extern "C"
