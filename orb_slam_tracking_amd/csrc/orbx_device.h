// orbx_device.h — POD structures shared between the host pipeline and the HIP kernels of liborbx.
#pragma once
#include <algorithm>
#include <cstdint>

#include "../../include/orbx.h"

#define ORBX_MAX_LEVELS 16
#define ORBX_EDGE 19          // EDGE_THRESHOLD, Features/ORBextractor.cpp:90
#define ORBX_MIN_BORDER 16    // EDGE_THRESHOLD - 3, cpp:1056
#define ORBX_CELL_MAX 76      // widest/tallest FAST cell image: ceil(width/nCols) <= 70, +6 overlap
#define ORBX_GRID_COLS 64     // FRAME_GRID_COLS, SlamTypes/Frame.hpp:16
#define ORBX_GRID_ROWS 48     // FRAME_GRID_ROWS, SlamTypes/Frame.hpp:15

// Wave priority of the latency-bound kernels (quadtree selection, Jacobi matcher, banded pyramid): their ~90 barrier-separated
// phases issue a few instructions each and then wait; on the lanes they share every SIMD with the issue-bound FAST / descriptor
// waves of other batches, behind which their few instructions queue.  ORBX_PRIO=n (build flag, experiment of round 4) raises them.
#if defined(__HIP_DEVICE_COMPILE__) && defined(ORBX_PRIO)
#define ORBX_SETPRIO() __builtin_amdgcn_s_setprio(ORBX_PRIO)
#else
#define ORBX_SETPRIO() do {} while (0)
#endif

namespace orbx {

// geometry of one pyramid level for the current frame size
struct LevelGeom {
  int32_t w, h, stride;       // level image, row stride in bytes (level 0: caller's stride)
  int32_t nCols, nRows;       // FAST cell grid (cpp:1067-1075)
  int32_t wCell, hCell;
  int32_t maxBX, maxBY;       // maxBorderX/Y (cpp:1058-1059); minBorder = 16
  int32_t cellBase;           // index of this level's first cell in the flattened (level, row, col) list
  int32_t candCap;            // entries of one frame's candidate area on this level = cells * segCap
  int32_t quota;              // mnFeaturesPerLevel
  int32_t xtabOff, ytabOff;   // offsets into the resize tables (entries)
  int32_t patchSize;          // (int)(31 * scale), cpp:1165
  float scale;                // mvScaleFactor
  int64_t imgOff;             // byte offset of frame 0's image inside the pyramid buffer (levels >= 1)
  int64_t frameStride;        // bytes between consecutive frames of this level
  int64_t candOff;            // entry offset of frame 0's candidate area (frame stride = candCap, cell stride = segCap)
  int32_t resizeSpanOk;       // 1 if the taps of any 4 consecutive outputs span <= 8 source pixels (k_resize_dw usable)
  uint32_t colsInv24;         // ceil(2^24 / nCols): cell index / nCols == (index * colsInv24) >> 24 (k_fast)
  int32_t segCap;             // entries of one cell's candidate segment = worst case of its NMS survivors
  int32_t candMax;            // worst-case number of candidates of the level (sum over its cells)
};

struct Geom {
  int32_t nlevels;
  int32_t nCellsTotal;
  int32_t iniTh, minTh;
  int32_t selCap;             // per-frame capacity of the selected-keypoint list (== output capacity)
  int32_t frame0;             // first frame of this launch (a batch may be issued as several sub-batches / streams)
  LevelGeom L[ORBX_MAX_LEVELS];
};

// k_fast_wave: what one wave needs to know about its FAST cell, precomputed on the host (buildFastCells) so that the kernel's
// prologue is one scalar load instead of a level search and a dozen divisions / multiplications per wave
struct FastCell {          // 32 bytes, eight dwords (16-bit fields are packed by hand: scalar loads come in dwords)
  uint32_t imgOff;         // byte offset, inside the frame's level image, of the first staged dword (row iniY, column iniX & ~3)
  uint32_t segOff;         // entry offset of the cell's candidate segment inside the frame's level area (local cell * segCap)
  uint32_t stride;         // row stride of the level image in bytes
  uint32_t nw_ch;          // staged dwords per row (<= 16) | rows (<= 64) << 16; rows == 0: the cell detects nothing (cpp:1088,1101)
  uint32_t iw_ih;          // detection area = cell image minus the 3-px FAST border: width | height << 16
  uint32_t xoff_level;     // iniX & 3 | pyramid level << 16
  uint32_t ox_oy;          // candidate (x, y) = (tile column + ox, tile row + oy), relative to minBorder: (uint16)ox | oy << 16
  uint32_t segCap;         // entries of the segment
};
static_assert(sizeof(FastCell) == 32, "FastCell is loaded with one s_load_dwordx8");

// candidate entry written by the FAST kernel: x | y<<12 | score<<24, x/y relative to minBorder
#if defined(__HIPCC__)
#define ORBX_HD __host__ __device__
#else
#define ORBX_HD
#endif
ORBX_HD static inline uint32_t packCand(int x, int y, int score) { return (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)score << 24); }

// a keypoint chosen by the quadtree stage, input of the orientation/descriptor kernel
struct SelKp {
  uint16_t x, y;        // level coordinates (already + minBorder)
  uint8_t level, response;  // response = FAST score (<= 254)
  uint16_t pad;
};

// quadtree selection stage (orbx_octree_kernel.hip): per-level constants of DistributeOctTree
struct OctLevel {
  int32_t width, height;   // maxBorder - minBorder
  int32_t nIni;            // cpp:706
  float hX;                // cpp:709
  int32_t wCell, hCell, nCols;  // FAST cell grid: defines the reference's candidate order
  int32_t quota;
  int32_t cellBase, nCells, segCap;  // the level's cells in the per-frame cell-count array; entries per cell segment
  int32_t depthBits;       // splits after which every cell of the level is one pixel (DivideNode halves with ceil): the quadrant
                           // digits of a path code beyond this depth are all 0
  int32_t tabOff, tabW, tabH;  // the level's path-code tables inside OctLaunch::codeTab (dword offset, even): tabW pairs
                           // {x digits of the 16 DivideNode splits at bit 2 (15 - d), root} by x, then tabH words of y digits
                           // (bit 2 (15 - d) + 1) by y (octCodeTable on the host); tabW >= width, tabH >= height
  // Large units on many workgroups (k_octree_buckets + k_octree_big): the level's keys are cut into nIni * 4^bigD0 BUCKETS, one
  // per tree node of depth bigD0 (0 = not usable for this level).  bigD0 never exceeds the depth the quota guarantees the full
  // passes to reach (octBigPlan), so no list node ever spans two buckets.  A bucket's keys are sorted in LDS by one workgroup
  // and land in the bucket's own slot of bigCapB entries of the unit's key / score / divergence arrays.
  int32_t bigD0, bigBuckets, bigCapB;  // chosen per launch (octBigChoose): depth <= bigDMax, nIni << 2 bigD0 buckets, keys per slot
  int32_t bigDMax;         // deepest bucket depth the level allows (-1: none): the depth of the interval tables
  int32_t bigTabOff;       // dword offset in codeTab: first x of every (root, top bigDMax x digits) prefix [nIni << bigDMax, + 1 end],
                           // then first y of every top-bigDMax-y-digits prefix [(1 << bigDMax) + 1]; a coarser prefix owns the union
};
#define ORBX_OCTB_CAP 1024        // keys of one bucket (one wave's register sort in k_octree_buckets)
#define ORBX_OCTB_MAX_BUCKETS 1024
#define ORBX_OCTB_INFO 40         // ints of a bucket's record: count, first / last inner divergence, overflow, 18 + 18 histogram bins

// Small launches (the one-frame call): k_describe_patch reads the selection's per-level staging lists directly and does
// k_sel_compact's bookkeeping in one designated wave per frame, so that no compaction kernel sits on the call's critical path.
struct DescStage {
  const SelKp* selStage;
  const int* nselLevel;
  int* nsel;
  int* nselUser;
  int* hostNsel;
  int* hostErr;
  int* maxN;
  int* hostMaxN;
  int32_t selStride;
  int32_t selOff[ORBX_MAX_LEVELS];
};
#define ORBX_DESC_STAGED_MAX_UNITS 256  // units (frames x levels) up to which the descriptor kernel takes the staged lists (round 5:
                                        // 64 -> 256: 32 frames of eight levels per launch; synchronous 32-frame call 132.7 k -> 143.7 k frames/s)

struct OctLaunch {
  OctLevel lev[ORBX_MAX_LEVELS];
  int64_t candOff[ORBX_MAX_LEVELS];
  int64_t scrOff[ORBX_MAX_LEVELS];     // global-scratch placement (k_octree_global)
  int64_t scrStride[ORBX_MAX_LEVELS];
  int32_t candCap[ORBX_MAX_LEVELS];
  int32_t selOff[ORBX_MAX_LEVELS];     // offset of the level inside one frame's SelKp staging area
  int32_t scrNMax[ORBX_MAX_LEVELS];
  int32_t nlevels, selStride;          // SelKp staging entries per frame
  int32_t frame0, nCellsTotal;         // first frame of this launch; cells of one frame over all levels
  const uint32_t* codeTab;             // path-code tables of the levels (device memory), see OctLevel::tabOff
};

#define ORBX_OCT_MAX_CAND ((1 << 19) - 1)  // candidates per (frame, level) the selection stage can index
// what a unit reports for the next batch's launch choices (maxN[], hostMaxN[], the host's candHintL[]): its candidate count and, from
// k_octree_big, the fill of its fullest bucket; the reducers take the maximum of each field
// (round 5: and the bucket depth the unit ran with -- a fill says nothing without it: ADVICE r04.)  Bits 0..18 count, 19..27 the
// fill in units of four keys (rounded up), 28..30 the depth.
#define ORBX_OCT_FEEDBACK(n, fill, depth) ((int)(n) | ((((int)(fill) + 3) >> 2) << 19) | ((int)(depth) << 28))
#define ORBX_OCT_FB_COUNT(v) ((int)(v) & 0x7ffff)
#define ORBX_OCT_FB_FILL(v) ((((int)(v) >> 19) & 0x1ff) << 2)
#define ORBX_OCT_FB_DEPTH(v) (((int)(v) >> 28) & 7)
#define ORBX_OCT_FB_MAXF(a, b, F) (F(a) > F(b) ? F(a) : F(b))
#define ORBX_OCT_FB_MAX(a, b) ORBX_OCT_FEEDBACK(ORBX_OCT_FB_MAXF(a, b, ORBX_OCT_FB_COUNT), ORBX_OCT_FB_MAXF(a, b, ORBX_OCT_FB_FILL), ORBX_OCT_FB_MAXF(a, b, ORBX_OCT_FB_DEPTH))
#define ORBX_OCT_REDONE 0x40000000  // bit of a (frame, level) count in nselLevel[]: k_octree_big redid the unit in place with the one-workgroup code

#if defined(__HIPCC__)
// ---- bookkeeping of the selection stage shared by k_sel_compact and the staged form of k_describe_patch (small launches, which
//      run no k_sel_compact): ONE definition of what a unit's raw count means and of what is published per frame ----
// a unit's raw count in nselLevel[]: negative = the unit failed; ORBX_OCT_REDONE set = k_octree_big redid it in place (same result)
__device__ __forceinline__ int selUnitCount(int raw, bool* failed) {
  *failed = *failed || raw < 0;
  return raw < 0 ? 0 : (raw & ~ORBX_OCT_REDONE);
}
// a frame's totals: to the context's array (k_describe_patch's launch bound), the caller's array, the host's mapped mirror; a failed
// unit raises the host's error flag (ORBX_E_CAPACITY at the batch's wait)
__device__ __forceinline__ void selPublishFrame(int f, int total, bool failed, int* __restrict__ nsel, int* __restrict__ nselUser,
                                                int* __restrict__ hostNsel, int* __restrict__ hostErr) {
  nsel[f] = total;
  if (nselUser) nselUser[f] = total;
  if (hostNsel) hostNsel[f] = total;
  if (failed) *hostErr = 1;
}
// per-level maxima (field by field, ORBX_OCT_FEEDBACK) of the units' reports of frames [frame0, frame0 + nFrames), by ONE wave:
// lane = (frame % 4, level); the reports are reset for the next launch
__device__ __forceinline__ void selReduceReports(int lane, int frame0, int nFrames, int nlevels, int* __restrict__ maxN,
                                                 int* __restrict__ hostMaxN) {
  static_assert(ORBX_MAX_LEVELS <= 16, "sixteen lanes per frame");
  int m = 0;
  if ((lane & 15) < nlevels)
    for (int fr = lane >> 4; fr < nFrames; fr += 4) {
      const int idx = (frame0 + fr) * nlevels + (lane & 15);
      const int v = maxN[idx];
      m = ORBX_OCT_FB_MAX(m, v);
      maxN[idx] = 0;
    }
  { const int t = __shfl_xor(m, 16); m = ORBX_OCT_FB_MAX(m, t); }
  { const int t = __shfl_xor(m, 32); m = ORBX_OCT_FB_MAX(m, t); }
  if (lane < nlevels) hostMaxN[lane] = m;
}
#endif


// Bucket depth of a level for one launch of the many-workgroup selection (k_octree_buckets / k_octree_big): the coarsest depth
// that leaves a bucket at most 256 keys on average of the `hint` candidates a unit of that level had in the previous batch (a
// bucket is one wave's sort of up to ORBX_OCTB_CAP keys: four times that average before a dense corner of the frame overfills
// its slot and the unit falls to the one-workgroup kernel; a wave spends ~1200 instructions on a bucket before its first key, so
// many small buckets cost more than few full ones: 40 us against 13 for the 600 k keys of four 4K frames) -- or, without a hint,
// about 16 k pixels: 250 keys at 1.5 %.  nMax = the candidates the unit's scratch is sized for (OctLaunch::scrNMax):
// the buckets' slots share its key array.  bigBuckets == 0: no plan, the level keeps the one-workgroup kernel.
// fillPrev / dPrev (round 5): the fullest bucket of the previous batch's units of this level and the depth they ran with.
// A bucket of a coarser depth d holds at most fillPrev * 4^(dPrev - d) keys, one of a finer depth at most fillPrev: the depth is
// deepened until that bound leaves a quarter of the bucket's slot free.  (Before, the depth came from the count alone; a unit
// whose bucket overflowed reported the largest count, got the deepest depth for one batch, reported its true count from there and
// returned to the depth that overflowed: every second batch of a scene with one dense cluster went through the in-place redo.)
// *estFill = that bound for the chosen depth (0 = unknown): picks k_octree_buckets' LDS slot count.
inline void octBigChoose(OctLevel* O, int nMax, int hint, int fillPrev = 0, int dPrev = 0, int* estFill = nullptr) {
  O->bigD0 = 0; O->bigBuckets = 0; O->bigCapB = 0;
  if (estFill) *estFill = 0;
  if (O->bigDMax < 0) return;
  int d = 0;
  if (hint > 0) {
    while (d < O->bigDMax && (long long)hint > 256ll * ((long long)O->nIni << (2 * d))) d++;
  } else {
    const double rootArea = (double)(O->width / O->nIni + 1) * (double)O->height;
    while (d < O->bigDMax && rootArea / (double)(1ll << (2 * d)) > 16384.0) d++;
  }
  long long nPad = 1024;
  while (nPad < nMax) nPad <<= 1;
  auto capOf = [&](int dd) {
    const long long nb = (long long)O->nIni << (2 * dd);
    int cap = ORBX_OCTB_CAP;
    while (cap > 0 && (long long)cap * nb > nPad) cap >>= 1;
    return cap;
  };
  auto bound = [&](int dd) -> long long {
    if (fillPrev <= 0) return 0;
    return dd <= dPrev ? (long long)fillPrev << (2 * (dPrev - dd)) : (long long)fillPrev;
  };
  while (d < O->bigDMax && bound(d) * 4 > 3ll * capOf(d)) d++;
  const long long nb = (long long)O->nIni << (2 * d);
  const int cap = capOf(d);
  if (cap < 256 || nb > ORBX_OCTB_MAX_BUCKETS) return;
  O->bigD0 = d; O->bigBuckets = (int32_t)nb; O->bigCapB = cap;
  if (estFill) *estFill = (int)std::min<long long>(bound(d), 1 << 20);
}

// k_copy_out: the result arrays of a host-frame batch, copied by a kernel into the caller's page-locked (device-mapped) arrays:
// segment s = `rows` rows of `rowDwords` dwords, of which the first cnt[row] * mult are copied (cnt == nullptr: the whole row)
struct CopySeg {
  const uint32_t* src;
  uint32_t* dst;
  const int32_t* cnt;
  int32_t rows, rowDwords, mult, pad;
};
struct CopyOut {
  CopySeg s[6];
};

struct ResizeTab {
  int32_t ofs;    // source index
  int32_t coef;   // c0 | c1 << 16 (Q11)
};

// k_pyramid_tiles: what tile t of a frame produces of level l -- the rectangle it owns (written to the pyramid) and the rectangle it
// needs (owned pixels plus what its rectangles of the higher levels read), both in level-l pixel coordinates; level 0: the part of
// the caller's image the tile stages (host: buildPyrTiles).  The tile's resize taps follow in one blob per tile (PyrTileTap):
// per level 1 .. nlevels - 1 the x taps of the needed columns, then the y taps of the needed rows.
struct PyrTileRect {
  int16_t nx0, nx1, ny0, ny1;  // needed: computed into LDS
  int16_t ox0, ox1, oy0, oy1;  // owned: also stored to the pyramid
};
struct PyrTileTap {
  uint32_t pos;   // first source column (row) relative to the previous level's needed rectangle | (second - first) << 16
  uint32_t coef;  // c0 | c1 << 16 (Q11), as ResizeTab
};
#define ORBX_PYR_TILE_TAPS 768  // taps of one tile over all levels (capacity of the blob and of its LDS copy)

// k_pyramid_bands: rows [r0, r1) of level l that band b of a frame produces (host: computePyrBands)
#define ORBX_PYR_BANDS_MAX 32
#define ORBX_PYR_STRIPS_MAX 8
struct PyrBands {
  int32_t nBands, dual2;                                 // dual2: pixel 2 of some group needs the second dword pair
  int32_t safeFrom;                                      // frames >= safeFrom (batch index): nothing is known to follow their level 0
  int32_t maxRows;                                       // most rows any band has on any level (<= 256)
  int32_t xoff[ORBX_MAX_LEVELS], yoff[ORBX_MAX_LEVELS];  // the level's PyrXGroup / PyrYRow tables, in uint4 units from the table base
  int16_t r0[ORBX_PYR_BANDS_MAX][ORBX_MAX_LEVELS], r1[ORBX_PYR_BANDS_MAX][ORBX_MAX_LEVELS];
  // column strips (round 5: few large frames -- four 3840x2160 frames are 128 workgroups of 32 bands, half the chip's CUs with one
  // workgroup each): strip s of a band produces the 4-pixel groups [g0, g1) of level l, its own share of the row plus the groups
  // its share of level l + 1 reads; workgroup = (band, strip).  One strip = the whole row (every 640x480 batch).
  int32_t nStrips;
  int16_t g0[ORBX_PYR_STRIPS_MAX][ORBX_MAX_LEVELS], g1[ORBX_PYR_STRIPS_MAX][ORBX_MAX_LEVELS];
};
// column constants of one group of 4 output pixels (host: appendPyrTables).  The group's taps lie in the 12 bytes of three
// dwords at byte offsets o[0] = (first tap & ~3), o[1], o[2] of a source row (o[1], o[2]: + 4, + 8, but never beyond the
// row's last dword, which can then only supply bytes of weight 0).
struct PyrXGroup {
  uint32_t o[3];
  uint32_t sel[6];   // v_perm_b32 selectors (bytes k, k + 1 as two u16 halves): pixel 0, 1, 2 in dwords (0, 1); pixel 2 in dwords
                     // (1, 2); pixel 3 in (0, 1); pixel 3 in (1, 2).  Of a pixel's two, the one that does not apply is 0x0c0c0c0c (zero)
  uint32_t cf[4];    // Q11 tap pair times 16: (c0 << 4) | (c1 << 4) << 16
  uint32_t pad[3];
};
static_assert(sizeof(PyrXGroup) == 64, "PyrXGroup is read as four uint4");
struct PyrYRow {     // one output row: byte offsets of its two source rows (already clamped), Q11 weights << 8
  uint32_t off0, off1, wy0, wy1;
};

// k_check_model: CheckHomography (kind 0) / CheckFundamental (kind 1) over nModels hypotheses
struct ScoreArgs {
  const float* M21;   // [nModels][9] row-major H21 / F21
  const float* M12;   // [nModels][9] H12 (kind 0 only)
  const orbx_keypoint* k1;
  const orbx_keypoint* k2;
  const int32_t* first;   // mvMatches12[i].first / .second
  const int32_t* second;
  int32_t N, kind;
  float invSigmaSquare;
  float* scores;      // [nModels]
  uint8_t* inliers;   // [nModels][N]
};

// k_check_rt: CheckRT (Initializer.cpp:569-713) over nModels (R21, t21) hypotheses
struct CheckRtArgs {
  const float* R21;      // [nModels][9]
  const float* t21;      // [nModels][3]
  const float* pts;      // [nInl][4] (u1, v1, u2, v2) of the inlier matches, in match order
  const int32_t* book;   // [nInl] keypoint of frame 1 the i-th triangulated point is booked under (the reference's quirk)
  float K[9];
  float th2;
  int32_t nInl, n1;
  uint8_t* good;         // [nModels][n1]
  float* p3d;            // [nModels][n1][3]
  float* cosBuf;         // [nModels][nInl] scratch: cosines of the counted points
  int32_t* nGood;        // [nModels]
  float* parallax;       // [nModels]
};


// camera of cv::undistortPoints in the doubles OpenCV converts mK / mDistCoef (CV_32F, Settings.hpp:32,39) to
struct CamD {
  double fx, fy, cx, cy, ifx, ify;  // ifx = 1./fx, ify = 1./fy
  double k0, k1, k2, k3;            // k1 k2 p1 p2 of the reference = OpenCV's k[0..3]
  int32_t distorted, pad_;          // mDistCoef.at<float>(0) != 0 (Frame.cpp:103,138)
};

// ---- SearchForInitialization scratch (ints per pair; one place for the host allocation and the launches) ----
// matchGeneral needs 4 * capacity; the wide path (k_match_wide_*) a header, one record per eligible train, the query index
// list, the per-query candidate counts and MW_CP list entries per query, all for min(capacity, MW_CAP) queries / trains.
constexpr int MW_CAP = 4096;  // octave-0 queries / eligible trains per pair the wide path takes
constexpr int MW_CP = 128;    // candidates listed per query (a fuller window hands the pair to k_match)
constexpr int MW_CXS = 80;    // behind the top rows and the queries' column order: start slot of every grid column's trains (65 used)
constexpr int MW_TOPK = 4;   // four spare rows behind every query's list (round 3 kept the best candidates there; the lists are sorted in place now)
constexpr int MW_HDR = 16;    // [0] nQ, [1] nT, [2] 1 = too large for the wide path, [3] != 0 = a list overflowed, [4..7] bounding box of the eligible trains (float bits: min x, max x, min y, max y), [8] 1 = trains stored by grid column (matchWidePrep)
inline int matchWideCap(int capacity) { return capacity < MW_CAP ? capacity : MW_CAP; }
inline long long matchScratchStride(int capacity) {
  const long long capl = matchWideCap(capacity);
  long long s = (long long)capacity * 4;
  const long long wide = MW_HDR + capl * (4 + 1 + 1) + capl * (MW_CP + MW_TOPK) + capl + MW_CXS;  // ... + query order by column + column starts
  if (s < wide) s = wide;
  return (s + 3) & ~3LL;  // the train records are uint4
}

}  // namespace orbx
