// orbx_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for the ORB tracking hot path.
//
//   k_resize    pyramid level l from level l-1      (ORBextractor::ComputePyramid, cpp:1660-1713 -> cv::resize)
//   k_fast      per-cell FAST-9-16 + in-cell NMS + threshold fallback
//                                                    (ComputeKeyPointsOctTree cell loops, cpp:1078-1141 -> cv::FAST)
//   k_describe  IC-angle + 7x7 Gaussian (patch-local) + steered BRIEF, one wave per keypoint
//                                                    (IC_Angle cpp:103-159, GaussianBlur cpp:1598-1606,
//                                                     computeOrbDescriptor cpp:169-228, assembly cpp:1557-1652)
//   k_match     SearchForInitialization, one workgroup per frame pair
//                                                    (ORBmatcher.cpp:11-183, Frame.cpp:89-99,163-206, FORB.cpp:77-101)
//
// All arithmetic is integer or uncontracted IEEE f32/f64 (compile with -ffp-contract=off) so the results are
// bit-identical to the CPU restatement in oracle/.  No MFMA: there is no dense contraction on this path.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/orbx.h"
#include "orbx_device.h"

namespace orbx {

// =================================================================================================
// K1  bilinear resize, Q11 fixed point (cv::resize INTER_LINEAR 8UC1)
// thread = 4 consecutive output pixels (one aligned u32 store), block = 64 x 4 threads
// =================================================================================================
__global__ __launch_bounds__(256) void k_resize(const uint8_t* __restrict__ src, long long srcFrameStride, int sw, int sh,
                                                int sstride, uint8_t* __restrict__ dst, long long dstFrameStride, int dw,
                                                int dh, int dstride, const ResizeTab* __restrict__ xtab,
                                                const ResizeTab* __restrict__ ytab) {
  const int f = blockIdx.z;
  const int dx0 = (blockIdx.x * 64 + threadIdx.x) * 4;
  const int dy = blockIdx.y * 4 + threadIdx.y;
  if (dx0 >= dw || dy >= dh) return;
  const ResizeTab ty = ytab[dy];
  const int sy0 = min(max(ty.ofs, 0), sh - 1), sy1 = min(max(ty.ofs + 1, 0), sh - 1);
  const int b0 = ty.coef & 0xffff, b1 = ty.coef >> 16;
  const uint8_t* S0 = src + (long long)f * srcFrameStride + (long long)sy0 * sstride;
  const uint8_t* S1 = src + (long long)f * srcFrameStride + (long long)sy1 * sstride;
  uint32_t packed = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const ResizeTab tx = xtab[dx0 + i];  // table is padded to a multiple of 4 entries
    const int sx = tx.ofs, sx1 = min(sx + 1, sw - 1);
    const int a0 = tx.coef & 0xffff, a1 = tx.coef >> 16;
    const int t0 = S0[sx] * a0 + S0[sx1] * a1;
    const int t1 = S1[sx] * a0 + S1[sx1] * a1;
    int v = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2;
    v = min(max(v, 0), 255);
    packed |= (uint32_t)v << (8 * i);
  }
  // rows are padded to a multiple of 64 bytes, so the full word may be stored even at the right edge
  *reinterpret_cast<uint32_t*>(dst + (long long)f * dstFrameStride + (long long)dy * dstride + dx0) = packed;
}

// =================================================================================================
// K2  FAST-9-16 per cell.  One workgroup (256 threads) per (cell, frame).
//   strength(p) = max over the 16 arcs of 9 contiguous ring pixels of min(+-(v - p_k))
//   corner at threshold t  <=>  strength > t ; score = strength - 1
//   in-cell NMS is threshold independent on the strength map:  keep <=> s > all 8 neighbours' s
//   (neighbours outside the cell's detection area count as 0) and s > 1
//   cell fallback (cpp:1117-1123): no survivor with s > iniTh  =>  use minTh for the whole cell
// =================================================================================================
#define TILE_STRIDE 84   // bytes per LDS tile row (21 words)
#define SMAP_STRIDE 72   // 70 + 2 zero apron
#define FAST_OUT_MAX 1296

__device__ __forceinline__ bool arc9(uint32_t m) {  // 16-bit circular mask has a run of >= 9 ones
  m |= m << 16;
  uint32_t r = m & (m >> 1);
  r &= r >> 2;
  r &= r >> 4;
  r &= m >> 8;
  return (r & 0xFFFFu) != 0;
}

__global__ __launch_bounds__(256) void k_fast(const uint8_t* __restrict__ img0, long long img0FrameStride, int img0Aligned,
                                              const uint8_t* __restrict__ pyr, const Geom g,
                                              uint32_t* __restrict__ cand, int* __restrict__ candCount,
                                              int* __restrict__ overflow) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[ORBX_CELL_MAX * TILE_STRIDE];
  __shared__ __attribute__((aligned(16))) uint8_t smap[SMAP_STRIDE * SMAP_STRIDE];
  __shared__ uint16_t list[70 * 70];
  __shared__ uint32_t outl[FAST_OUT_MAX];
  __shared__ int nList, nOut, outBase;

  const int t = threadIdx.x;
  const int f = blockIdx.y;
  int level = 0;
  const int cid = blockIdx.x;
  while (level + 1 < g.nlevels && cid >= g.L[level + 1].cellBase) level++;
  const LevelGeom& L = g.L[level];
  const int local = cid - L.cellBase;
  const int ci = local / L.nCols, cj = local - ci * L.nCols;
  // cell rectangle, cpp:1082-1103
  const int iniY = ORBX_MIN_BORDER + ci * L.hCell;
  const int iniX = ORBX_MIN_BORDER + cj * L.wCell;
  if (iniY >= L.maxBY - 3 || iniX >= L.maxBX - 6) return;
  const int maxY = min(iniY + L.hCell + 6, L.maxBY), maxX = min(iniX + L.wCell + 6, L.maxBX);
  const int cw = maxX - iniX, ch = maxY - iniY;
  if (cw < 7 || ch < 7) return;  // cv::FAST finds nothing in an image this small
  const int iw = cw - 6, ih = ch - 6;

  const uint8_t* base;
  int stride;
  bool aligned = true;
  if (level == 0) {
    base = img0 + (long long)f * img0FrameStride;
    stride = L.stride;
    aligned = img0Aligned != 0;
  } else {
    base = pyr + L.imgOff + (long long)f * L.frameStride;
    stride = L.stride;
  }
  // ---- stage the cell image in LDS (coalesced dword loads of the enclosing aligned span) ----
  const int ax0 = aligned ? (iniX & ~3) : iniX;
  const int xoff = iniX - ax0;
  if (aligned) {
    const int nw = (maxX - ax0 + 3) >> 2;
    uint32_t* tile32 = reinterpret_cast<uint32_t*>(tile);
    for (int idx = t; idx < nw * ch; idx += 256) {
      const int r = idx / nw, c = idx - r * nw;
      tile32[r * (TILE_STRIDE / 4) + c] =
          *reinterpret_cast<const uint32_t*>(base + (long long)(iniY + r) * stride + ax0 + 4 * c);
    }
  } else {
    for (int idx = t; idx < cw * ch; idx += 256) {
      const int r = idx / cw, c = idx - r * cw;
      tile[r * TILE_STRIDE + c] = base[(long long)(iniY + r) * stride + iniX + c];
    }
  }
  {
    uint32_t* smap32 = reinterpret_cast<uint32_t*>(smap);
    for (int idx = t; idx < SMAP_STRIDE * SMAP_STRIDE / 4; idx += 256) smap32[idx] = 0;
  }
  if (t == 0) { nList = 0; nOut = 0; }
  __syncthreads();

  // ring offsets inside the LDS tile, k = 0..15 (dx,dy) = (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)...
  constexpr int RS = TILE_STRIDE;
  constexpr int ro[16] = {3 * RS,      3 * RS + 1,  2 * RS + 2,  RS + 3,  3,       -RS + 3,     -2 * RS + 2, -3 * RS + 1,
                          -3 * RS,     -3 * RS - 1, -2 * RS - 2, -RS - 3, -3,      RS - 3,      2 * RS - 2,  3 * RS - 1};
  // per-thread pixel walk without divisions inside the loops: idx += 256  <=>  (px, py) += (256 % iw, 256 / iw)
  const int py0 = t / iw, px0 = t - py0 * iw;
  const int dpy = 256 / iw, dpx = 256 - dpy * iw;
  const int npix = iw * ih;
  // The reference runs cv::FAST at iniThFAST and, only if the cell yields nothing, again at minThFAST (cpp:1109-1123).
  // Same here: pass 0 at iniTh, pass 1 at minTh only for cells without a survivor.  The strength map is threshold
  // independent, so what pass 0 wrote stays valid for pass 1.
  for (int pass = 0; pass < 2; pass++) {
    const int th = pass == 0 ? g.iniTh : g.minTh;
    // ---- phase 0: necessary condition on the 4 compass pixels (an arc of 9 holds two adjacent ones) ----
    {
      int px = px0, py = py0;
      for (int idx = t; idx < npix; idx += 256) {
        const uint8_t* p = &tile[(py + 3) * TILE_STRIDE + xoff + px + 3];
        const int v = p[0], hi = v + th, lo = v - th;
        const int q0 = p[ro[0]], q4 = p[ro[4]], q8 = p[ro[8]], q12 = p[ro[12]];
        const bool b0 = q0 > hi, b4 = q4 > hi, b8 = q8 > hi, b12 = q12 > hi;
        const bool d0 = q0 < lo, d4 = q4 < lo, d8 = q8 < lo, d12 = q12 < lo;
        const bool cand0 = ((b0 | b8) & (b4 | b12)) | ((d0 | d8) & (d4 | d12));
        if (cand0) list[atomicAdd(&nList, 1)] = (uint16_t)((py << 7) | px);
        px += dpx;
        py += dpy;
        if (px >= iw) { px -= iw; py++; }
      }
    }
    __syncthreads();
    const int nl = nList;
    // ---- phase 1: exact strength of the remaining pixels; corners (s > th) enter the strength map ----
    for (int e = t; e < nl; e += 256) {
      const int code = list[e];
      const int py = code >> 7, px = code & 127;
      const uint8_t* p = &tile[(py + 3) * TILE_STRIDE + xoff + px + 3];
      const int v = p[0];
      int d[16];
#pragma unroll
      for (int k = 0; k < 16; k++) d[k] = v - (int)p[ro[k]];
      int mn3[16], mx3[16];
#pragma unroll
      for (int k = 0; k < 16; k++) {
        mn3[k] = min(min(d[k], d[(k + 1) & 15]), d[(k + 2) & 15]);
        mx3[k] = max(max(d[k], d[(k + 1) & 15]), d[(k + 2) & 15]);
      }
      int smn = -256, smx = 256;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        smn = max(smn, min(min(mn3[k], mn3[(k + 3) & 15]), mn3[(k + 6) & 15]));
        smx = min(smx, max(max(mx3[k], mx3[(k + 3) & 15]), mx3[(k + 6) & 15]));
      }
      const int s = max(smn, -smx);
      if (s > th) smap[(py + 1) * SMAP_STRIDE + px + 1] = (uint8_t)s;
      else list[e] = 0xFFFF;
    }
    __syncthreads();
    // ---- phase 2: in-cell NMS on the strength map; survivors are the cell's keypoints ----
    for (int e = t; e < nl; e += 256) {
      const int code = list[e];
      if (code == 0xFFFF) continue;
      const int py = code >> 7, px = code & 127;
      const uint8_t* q = &smap[(py + 1) * SMAP_STRIDE + px + 1];
      const int s = q[0];
      const bool keep = s > 1 && s > q[-SMAP_STRIDE - 1] && s > q[-SMAP_STRIDE] && s > q[-SMAP_STRIDE + 1] && s > q[-1] &&
                        s > q[1] && s > q[SMAP_STRIDE - 1] && s > q[SMAP_STRIDE] && s > q[SMAP_STRIDE + 1];
      if (keep) {
        const int slot = atomicAdd(&nOut, 1);
        if (slot < FAST_OUT_MAX) outl[slot] = packCand(px + 3 + cj * L.wCell, py + 3 + ci * L.hCell, s - 1);
      }
    }
    __syncthreads();
    if (nOut > 0 || pass == 1 || g.minTh >= g.iniTh) break;
    if (t == 0) nList = 0;  // retry the whole cell at minThFAST
    __syncthreads();
  }
  const int no = min(nOut, FAST_OUT_MAX);
  if (t == 0) outBase = no ? atomicAdd(&candCount[f * g.nlevels + level], no) : 0;
  __syncthreads();
  const int ob = outBase;
  uint32_t* dstc = cand + L.candOff + (long long)f * L.candCap;
  for (int e = t; e < no; e += 256) {
    if (ob + e < L.candCap) dstc[ob + e] = outl[e];
    else *overflow = 1;
  }
}

// =================================================================================================
// K4+K5+K6  orientation + patch-local Gaussian + steered BRIEF.  One wave (64 threads) per keypoint.
// The 7x7 blur is evaluated only on the 37x37 neighbourhood the 512 sample points can reach; the fixed-point
// arithmetic (Q8 taps [18,34,48,56,48,34,18], u16 horizontal sums, (sum + 2^15) >> 16) is exactly the separable
// whole-image blur of cpp:1598-1606, including BORDER_REFLECT_101 at the level's own edges.
// =================================================================================================
__device__ const int8_t d_pattern[256 * 4] = {
#include "orbx_pattern_data.inc"
};
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};  // cpp:562-594

__device__ __forceinline__ int reflect101(int p, int n) {
  if (p < 0) p = -p;
  if (p >= n) p = 2 * n - 2 - p;
  return p;
}

// cv::fastAtan2: plain f32 mul/add/div, no contraction
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  const float eps = (float)2.2204460492503131e-16;
  const float ax = fabsf(x), ay = fabsf(y);
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

#define RAW_N 43
#define RAW_STRIDE 44
#define BL_N 37
#define HB_STRIDE 38
#define BL_STRIDE 40

__global__ __launch_bounds__(64) void k_describe(const uint8_t* __restrict__ img0, long long img0FrameStride,
                                                 const uint8_t* __restrict__ pyr, const Geom g,
                                                 const SelKp* __restrict__ sel, const int* __restrict__ nsel,
                                                 orbx_keypoint* __restrict__ kps, uint8_t* __restrict__ desc, int capacity) {
  __shared__ uint8_t raw[RAW_N * RAW_STRIDE];
  __shared__ uint16_t hb[RAW_N * HB_STRIDE];
  __shared__ uint8_t bl[BL_N * BL_STRIDE];
  const int f = blockIdx.y, i = blockIdx.x, lane = threadIdx.x;
  if (i >= nsel[f]) return;
  const SelKp k = sel[(long long)f * g.selCap + i];
  const LevelGeom& L = g.L[k.level];
  const uint8_t* img = k.level == 0 ? img0 + (long long)f * img0FrameStride : pyr + L.imgOff + (long long)f * L.frameStride;
  const int kx = k.x, ky = k.y;
  // ---- 43x43 raw patch, REFLECT_101 at the level's edges ----
  for (int idx = lane; idx < RAW_N * RAW_N; idx += 64) {
    const int r = idx / RAW_N, c = idx - r * RAW_N;
    const int yy = reflect101(ky - 21 + r, L.h), xx = reflect101(kx - 21 + c, L.w);
    raw[r * RAW_STRIDE + c] = img[(long long)yy * L.stride + xx];
  }
  __syncthreads();
  // ---- IC_Angle (cpp:103-159): m10 = sum u*I, m01 = sum v*I over the 749-pixel disc, un-blurred image ----
  int m10 = 0, m01 = 0;
  {
    const int u = (lane & 31) - 15;  // lane 31 / 63 idle
    for (int it = 0; it < 16; it++) {
      const int v = -15 + 2 * it + (lane >> 5);
      if (v <= 15 && u <= 15) {
        const int av = v < 0 ? -v : v, au = u < 0 ? -u : u;
        if (au <= c_umax[av]) {
          const int I = raw[(21 + v) * RAW_STRIDE + 21 + u];
          m10 += u * I;
          m01 += v * I;
        }
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      m10 += __shfl_xor(m10, o);
      m01 += __shfl_xor(m01, o);
    }
  }
  const float angle = fast_atan2_deg((float)m01, (float)m10);
  // ---- horizontal 7-tap pass (Q8, exact in u16), rows 0..42, blur columns 0..36 ----
  for (int idx = lane; idx < RAW_N * BL_N; idx += 64) {
    const int r = idx / BL_N, c = idx - r * BL_N;
    const uint8_t* p = &raw[r * RAW_STRIDE + c];
    const int s = 18 * (p[0] + p[6]) + 34 * (p[1] + p[5]) + 48 * (p[2] + p[4]) + 56 * p[3];
    hb[r * HB_STRIDE + c] = (uint16_t)s;
  }
  __syncthreads();
  // ---- vertical pass (Q16) + round to nearest ----
  for (int idx = lane; idx < BL_N * BL_N; idx += 64) {
    const int r = idx / BL_N, c = idx - r * BL_N;
    const uint16_t* p = &hb[r * HB_STRIDE + c];
    const uint32_t s = 18u * (p[0] + p[6 * HB_STRIDE]) + 34u * (p[HB_STRIDE] + p[5 * HB_STRIDE]) +
                       48u * (p[2 * HB_STRIDE] + p[4 * HB_STRIDE]) + 56u * p[3 * HB_STRIDE];
    const uint32_t v = (s + 32768u) >> 16;
    bl[r * BL_STRIDE + c] = (uint8_t)(v > 255u ? 255u : v);
  }
  __syncthreads();
  // ---- steered BRIEF (cpp:169-228).  cos/sin of the f32 argument are evaluated in f64 and rounded to f32 ----
  const float factorPI = (float)(3.14159265358979323846 / 180.f);
  const float a = angle * factorPI;
  const float cs = (float)cos((double)a), sn = (float)sin((double)a);
  unsigned long long words[4];
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const int bit = w * 64 + lane;
    const int8_t* pt = &d_pattern[bit * 4];
    const float x0 = (float)pt[0], y0 = (float)pt[1], x1 = (float)pt[2], y1 = (float)pt[3];
    const int r0 = __float2int_rn(x0 * sn + y0 * cs), c0 = __float2int_rn(x0 * cs - y0 * sn);
    const int r1 = __float2int_rn(x1 * sn + y1 * cs), c1 = __float2int_rn(x1 * cs - y1 * sn);
    const int t0 = bl[(18 + r0) * BL_STRIDE + 18 + c0];
    const int t1 = bl[(18 + r1) * BL_STRIDE + 18 + c1];
    words[w] = __ballot(t0 < t1);
  }
  const long long o = (long long)f * capacity + i;
  if (lane < 4) reinterpret_cast<unsigned long long*>(desc + o * 32)[lane] = words[lane];
  if (lane == 0) {
    orbx_keypoint kp;
    // cpp:1631-1634: pt *= scale for level != 0 (scale[0] == 1 exactly)
    kp.x = k.level ? (float)kx * L.scale : (float)kx;
    kp.y = k.level ? (float)ky * L.scale : (float)ky;
    kp.size = (float)L.patchSize;
    kp.angle = angle;
    kp.response = (float)k.response;
    kp.octave = k.level;
    kp.class_id = -1;
    kps[o] = kp;
  }
}

// =================================================================================================
// K7+K8  ORBmatcher::SearchForInitialization.  One workgroup per frame pair.
// The reference walks the queries sequentially and lets earlier matches hide candidates from later queries
// (vMatchedDistance, ORBmatcher.cpp:67), so the query loop stays sequential inside the workgroup; the candidate
// scan of one query (window test + 256-bit Hamming + best / second best) runs across the workgroup's lanes.
// Tie-breaking equals the reference's candidate order (GetFeaturesInArea: cell x outer, cell y inner, index):
// the best candidate is the minimum of (distance, cellX*48+cellY, index).
// =================================================================================================
#define MATCH_T 256
#define TH_LOW 50
#define HISTO_LENGTH 30
#define INF_DIST 0x7fffffff
#define MATCH_NONE 0x7fffffffffffffffull

#define MATCH_PENDING ((int)0x80000000)  // nmatches value meaning "left for the general kernel"

struct MatchParams {
  int capacity;
  int window;
  float nnratio;
  int checkOri;
  int onlyPending;
  orbx_bounds b;
};

__device__ __forceinline__ int hamming256(const uint4 a0, const uint4 a1, const uint32_t* __restrict__ b) {
  const uint4 b0 = reinterpret_cast<const uint4*>(b)[0], b1 = reinterpret_cast<const uint4*>(b)[1];
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
         __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// -------------------------------------------------------------------------------------------------
// k_match_wave: one WAVE per frame pair, everything the sequential query loop touches lives in LDS:
// the octave-0 queries of F1 and the grid-eligible octave-0 trains of F2 (position, cell, angle, descriptor words
// stored word-major so that lane e reading word w is bank-conflict free), vMatchedDistance, vnMatches21 and the
// rotation bins.  No barriers inside the query loop (a single wave executes its LDS operations in order).
// Pairs that do not fit (more than MW_CAP eligible trains / octave-0 queries, or n1 > MW_N1) are marked
// MATCH_PENDING and done by the general kernel k_match right after.
// -------------------------------------------------------------------------------------------------
#define MW_CAP 512
#define MW_N1 4096

__global__ __launch_bounds__(64) void k_match_wave(const int* __restrict__ pairFirst, const int* __restrict__ pairSecond,
                                                   const orbx_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                   const int* __restrict__ nkp, const MatchParams mp,
                                                   int* __restrict__ matches12, int* __restrict__ nmatchesOut,
                                                   int* __restrict__ statsOut) {
  __shared__ float tX[MW_CAP], tY[MW_CAP], tAng[MW_CAP];
  __shared__ uint32_t tDesc[8][MW_CAP];
  __shared__ int tMd[MW_CAP], tM21[MW_CAP];
  __shared__ uint16_t tIdx[MW_CAP], tCell[MW_CAP];
  __shared__ float qX[MW_CAP], qY[MW_CAP], qAng[MW_CAP];
  __shared__ uint32_t qDesc[MW_CAP][8];
  __shared__ uint16_t qIdx[MW_CAP];
  __shared__ uint8_t accBin[MW_N1];  // rotation bin per F1 index, 255 = not in rotHist
  __shared__ int hist[HISTO_LENGTH];

  const int lane = threadIdx.x;
  const int pair = blockIdx.x;
  const int fa = pairFirst[pair], fb = pairSecond[pair];
  const int n1 = nkp[fa], n2 = nkp[fb];
  const int cap = mp.capacity;
  const orbx_keypoint* k1 = kps + (long long)fa * cap;
  const orbx_keypoint* k2 = kps + (long long)fb * cap;
  const uint32_t* d1 = reinterpret_cast<const uint32_t*>(desc + (long long)fa * cap * 32);
  const uint32_t* d2 = reinterpret_cast<const uint32_t*>(desc + (long long)fb * cap * 32);
  int* m12 = matches12 + (long long)pair * cap;

  const float wInv = (float)ORBX_GRID_COLS / (float)(mp.b.max_x - mp.b.min_x);  // Frame.cpp:46-47
  const float hInv = (float)ORBX_GRID_ROWS / (float)(mp.b.max_y - mp.b.min_y);
  const float fminX = (float)mp.b.min_x, fminY = (float)mp.b.min_y;

  // ---- stage the eligible trains of F2 (index order kept) ----
  int nT = 0;
  bool fits = n1 <= MW_N1 && n2 <= 65535;
  for (int j0 = 0; j0 < n2 && fits; j0 += 64) {
    const int j = j0 + lane;
    bool ok = false;
    orbx_keypoint kp;
    int cell = 0;
    if (j < n2) {
      kp = k2[j];
      // Frame::PosInGrid (Frame.cpp:89-99) + the octave filter of GetFeaturesInArea (Frame.cpp:179,191)
      const int px = (int)roundf((kp.x - fminX) * wInv), py = (int)roundf((kp.y - fminY) * hInv);
      ok = kp.octave == 0 && px >= 0 && px < ORBX_GRID_COLS && py >= 0 && py < ORBX_GRID_ROWS;
      cell = px * ORBX_GRID_ROWS + py;
    }
    const unsigned long long m = __ballot(ok);
    const int pos = nT + __popcll(m & ((1ull << lane) - 1ull));
    if (nT + __popcll(m) > MW_CAP) { fits = false; break; }
    if (ok) {
      tX[pos] = kp.x; tY[pos] = kp.y; tAng[pos] = kp.angle;
      tIdx[pos] = (uint16_t)j; tCell[pos] = (uint16_t)cell;
      tMd[pos] = INF_DIST; tM21[pos] = -1;
#pragma unroll
      for (int w = 0; w < 8; w++) tDesc[w][pos] = d2[(long long)j * 8 + w];
    }
    nT += __popcll(m);
  }
  // ---- stage the octave-0 queries of F1 ----
  int nQ = 0;
  for (int i0 = 0; i0 < n1 && fits; i0 += 64) {
    const int i = i0 + lane;
    bool ok = false;
    orbx_keypoint kp;
    if (i < n1) {
      kp = k1[i];
      ok = !(kp.octave > 0);  // ORBmatcher.cpp:38-39
      accBin[i] = 255;
      m12[i] = -1;
    }
    const unsigned long long m = __ballot(ok);
    const int pos = nQ + __popcll(m & ((1ull << lane) - 1ull));
    if (nQ + __popcll(m) > MW_CAP) { fits = false; break; }
    if (ok) {
      qX[pos] = kp.x; qY[pos] = kp.y; qAng[pos] = kp.angle; qIdx[pos] = (uint16_t)i;
#pragma unroll
      for (int w = 0; w < 8; w++) qDesc[pos][w] = d1[(long long)i * 8 + w];
    }
    nQ += __popcll(m);
  }
  if (!fits) {  // wave-uniform
    if (lane == 0) nmatchesOut[pair] = MATCH_PENDING;
    return;
  }
  if (lane < HISTO_LENGTH) hist[lane] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();

  int nm = 0, badDist = 0, badRatio = 0, badOri = 0;
  const float r = (float)mp.window;
  const float factor = HISTO_LENGTH / 360.0f;
  for (int q = 0; q < nQ; q++) {
    const float x1 = qX[q], y1 = qY[q];
    // cell window, Frame.cpp:167-177
    const int minCX = max(0, (int)floorf((x1 - fminX - r) * wInv));
    const int maxCX = min(ORBX_GRID_COLS - 1, (int)ceilf((x1 - fminX + r) * wInv));
    const int minCY = max(0, (int)floorf((y1 - fminY - r) * hInv));
    const int maxCY = min(ORBX_GRID_ROWS - 1, (int)ceilf((y1 - fminY + r) * hInv));
    if (minCX >= ORBX_GRID_COLS || maxCX < 0 || minCY >= ORBX_GRID_ROWS || maxCY < 0) continue;
    uint32_t qd[8];
#pragma unroll
    for (int w = 0; w < 8; w++) qd[w] = qDesc[q][w];
    unsigned long long best = MATCH_NONE;  // dist << 32 | cell << 20 | train index
    int second = INF_DIST, any = 0, bestE = -1;
    for (int e = lane; e < nT; e += 64) {
      const int c = tCell[e];
      const int cx = c / ORBX_GRID_ROWS, cy = c - cx * ORBX_GRID_ROWS;
      if (cx < minCX || cx > maxCX || cy < minCY || cy > maxCY) continue;
      const float dx = tX[e] - x1, dy = tY[e] - y1;
      if (!(fabsf(dx) < r && fabsf(dy) < r)) continue;
      any = 1;
      int dist = 0;
#pragma unroll
      for (int w = 0; w < 8; w++) dist += __popc(qd[w] ^ tDesc[w][e]);
      if (tMd[e] <= dist) continue;  // ORBmatcher.cpp:67
      const unsigned long long key = ((unsigned long long)dist << 32) | ((unsigned long long)c << 20) | (unsigned)tIdx[e];
      if (key < best) {
        second = min(second, (int)(best >> 32));
        best = key;
        bestE = e;
      } else {
        second = min(second, dist);
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const unsigned long long ob = __shfl_xor(best, o);
      const int os = __shfl_xor(second, o), oe = __shfl_xor(bestE, o);
      any |= __shfl_xor(any, o);
      const bool take = ob < best;
      second = min(min(second, os), (int)((take ? best : ob) >> 32));
      best = take ? ob : best;
      bestE = take ? oe : bestE;
    }
    if (!any) continue;  // vIndices2.empty(), ORBmatcher.cpp:45
    const int bestDist = (int)(best >> 32);
    if (best == MATCH_NONE || bestDist > TH_LOW) { badDist++; continue; }
    if ((float)bestDist > mp.nnratio * (float)second) { badRatio++; continue; }
    // all lanes hold the same decision; lane 0 commits it
    const int i1 = qIdx[q];
    const int bestIdx2 = (int)(best & 0xFFFFF);
    const int old = tM21[bestE];
    if (old >= 0) nm--;
    nm++;
    int bin = -1;
    if (mp.checkOri) {
      float rot = qAng[q] - tAng[bestE];
      if (rot < 0.0f) rot += 360.0f;
      bin = (int)roundf(rot * factor);
      if (bin == HISTO_LENGTH) bin = 0;
      if (bin < 0 || bin >= HISTO_LENGTH) bin = -1;
    }
    if (lane == 0) {
      if (old >= 0) m12[old] = -1;
      m12[i1] = bestIdx2;
      tM21[bestE] = i1;
      tMd[bestE] = bestDist;
      if (bin >= 0) { accBin[i1] = (uint8_t)bin; hist[bin]++; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  // ---- rotation histogram: keep the three largest bins (ComputeThreeMaxima, ORBmatcher.cpp:152-183) ----
  if (mp.checkOri) {
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < HISTO_LENGTH; i++) {
      const int s = hist[i];
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
      else if (s > max3) { max3 = s; ind3 = i; }
    }
    if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
    int dropped = 0;
    for (int i = lane; i < n1; i += 64) {
      const int b = accBin[i];
      if (b != 255 && b != ind1 && b != ind2 && b != ind3) {  // also hits queries whose match was stolen (quirk, :130-138)
        m12[i] = -1;
        dropped++;
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) dropped += __shfl_xor(dropped, o);
    nm -= dropped;
    badOri += dropped;
  }
  if (lane == 0) {
    nmatchesOut[pair] = nm;
    if (statsOut) { statsOut[pair * 3] = badDist; statsOut[pair * 3 + 1] = badRatio; statsOut[pair * 3 + 2] = badOri; }
  }
}

__global__ __launch_bounds__(MATCH_T) void k_match(const int* __restrict__ pairFirst, const int* __restrict__ pairSecond,
                                                   const orbx_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                   const int* __restrict__ nkp, const MatchParams mp,
                                                   int* __restrict__ matches12, int* __restrict__ nmatchesOut,
                                                   int* __restrict__ statsOut, int* __restrict__ scratch) {
  __shared__ unsigned long long sBest[MATCH_T / 64];
  __shared__ int sSecond[MATCH_T / 64], sAny[MATCH_T / 64];
  __shared__ int hist[HISTO_LENGTH];
  __shared__ int sNm, sBadDist, sBadRatio, sBadOri, sKeep[3];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int pair = blockIdx.x;
  if (mp.onlyPending && nmatchesOut[pair] != MATCH_PENDING) return;  // already done by k_match_wave
  const int fa = pairFirst[pair], fb = pairSecond[pair];
  const int n1 = nkp[fa], n2 = nkp[fb];
  const int cap = mp.capacity;
  const orbx_keypoint* k1 = kps + (long long)fa * cap;
  const orbx_keypoint* k2 = kps + (long long)fb * cap;
  const uint8_t* d1 = desc + (long long)fa * cap * 32;
  const uint8_t* d2 = desc + (long long)fb * cap * 32;
  int* m12 = matches12 + (long long)pair * cap;
  int* md = scratch + (long long)pair * cap * 4;  // vMatchedDistance
  int* m21 = md + cap;                            // vnMatches21
  int* accBin = m21 + cap;                        // rotation bin of every accepted query (rotHist membership)
  int* cell2 = accBin + cap;                      // F2 grid cell (cx*48+cy) of every eligible train, -1 otherwise

  const float wInv = (float)ORBX_GRID_COLS / (float)(mp.b.max_x - mp.b.min_x);  // Frame.cpp:46-47
  const float hInv = (float)ORBX_GRID_ROWS / (float)(mp.b.max_y - mp.b.min_y);
  const float fminX = (float)mp.b.min_x, fminY = (float)mp.b.min_y;
  for (int j = t; j < n2; j += MATCH_T) {
    md[j] = INF_DIST;
    m21[j] = -1;
    const orbx_keypoint kp = k2[j];
    // Frame::PosInGrid (Frame.cpp:89-99) + the octave filter of GetFeaturesInArea (Frame.cpp:179,191)
    const int px = (int)roundf((kp.x - fminX) * wInv), py = (int)roundf((kp.y - fminY) * hInv);
    const bool ok = kp.octave == 0 && px >= 0 && px < ORBX_GRID_COLS && py >= 0 && py < ORBX_GRID_ROWS;
    cell2[j] = ok ? px * ORBX_GRID_ROWS + py : -1;
  }
  for (int i = t; i < n1; i += MATCH_T) {
    m12[i] = -1;
    accBin[i] = -1;
  }
  if (t < HISTO_LENGTH) hist[t] = 0;
  if (t == 0) { sNm = 0; sBadDist = 0; sBadRatio = 0; sBadOri = 0; }
  __syncthreads();

  const float r = (float)mp.window;
  const float factor = HISTO_LENGTH / 360.0f;
  for (int i1 = 0; i1 < n1; i1++) {
    const orbx_keypoint kp1 = k1[i1];
    if (kp1.octave > 0) continue;  // ORBmatcher.cpp:38-39
    // cell window, Frame.cpp:167-177
    const int minCX = max(0, (int)floorf((kp1.x - fminX - r) * wInv));
    const int maxCX = min(ORBX_GRID_COLS - 1, (int)ceilf((kp1.x - fminX + r) * wInv));
    const int minCY = max(0, (int)floorf((kp1.y - fminY - r) * hInv));
    const int maxCY = min(ORBX_GRID_ROWS - 1, (int)ceilf((kp1.y - fminY + r) * hInv));
    if (minCX >= ORBX_GRID_COLS || maxCX < 0 || minCY >= ORBX_GRID_ROWS || maxCY < 0) continue;
    const uint4 q0 = reinterpret_cast<const uint4*>(d1 + (long long)i1 * 32)[0];
    const uint4 q1 = reinterpret_cast<const uint4*>(d1 + (long long)i1 * 32)[1];
    unsigned long long best = MATCH_NONE;  // (dist << 32) | (cell << 20 | index)  -- index < 2^20
    int second = INF_DIST, any = 0;
    for (int j = t; j < n2; j += MATCH_T) {
      const int c = cell2[j];
      if (c < 0) continue;
      const int cx = c / ORBX_GRID_ROWS, cy = c - cx * ORBX_GRID_ROWS;
      if (cx < minCX || cx > maxCX || cy < minCY || cy > maxCY) continue;
      const float dx = k2[j].x - kp1.x, dy = k2[j].y - kp1.y;
      if (!(fabsf(dx) < r && fabsf(dy) < r)) continue;
      any = 1;
      const int dist = hamming256(q0, q1, reinterpret_cast<const uint32_t*>(d2 + (long long)j * 32));
      if (md[j] <= dist) continue;  // ORBmatcher.cpp:67
      const unsigned long long key = ((unsigned long long)dist << 32) | ((unsigned long long)c << 20) | (unsigned)j;
      if (key < best) {
        const int prev = (int)(best >> 32);
        second = min(second, prev);
        best = key;
      } else {
        second = min(second, dist);
      }
    }
    // wave reduction of (best, second, any)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const unsigned long long ob = __shfl_xor(best, o);
      const int os = __shfl_xor(second, o);
      any |= __shfl_xor(any, o);
      const unsigned long long lo = ob < best ? ob : best, hi = ob < best ? best : ob;
      second = min(min(second, os), (int)(hi >> 32));
      best = lo;
    }
    if (lane == 0) { sBest[wave] = best; sSecond[wave] = second; sAny[wave] = any; }
    __syncthreads();
    if (t == 0) {
      unsigned long long bb = sBest[0];
      int ss = sSecond[0], aa = sAny[0];
      for (int w = 1; w < MATCH_T / 64; w++) {
        const unsigned long long ob = sBest[w];
        const unsigned long long lo = ob < bb ? ob : bb, hi = ob < bb ? bb : ob;
        ss = min(min(ss, sSecond[w]), (int)(hi >> 32));
        bb = lo;
        aa |= sAny[w];
      }
      if (aa) {  // vIndices2 not empty, ORBmatcher.cpp:45
        const int bestDist = (int)(bb >> 32);
        const int bestDist2 = ss;
        const int bestIdx2 = (int)(bb & 0xFFFFF);
        const bool none = bb == MATCH_NONE;
        if (none || bestDist > TH_LOW) {
          sBadDist++;
        } else if ((float)bestDist > mp.nnratio * (float)bestDist2) {
          sBadRatio++;
        } else {
          if (m21[bestIdx2] >= 0) { m12[m21[bestIdx2]] = -1; sNm--; }
          m12[i1] = bestIdx2;
          m21[bestIdx2] = i1;
          md[bestIdx2] = bestDist;
          sNm++;
          if (mp.checkOri) {
            float rot = kp1.angle - k2[bestIdx2].angle;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            if (bin >= 0 && bin < HISTO_LENGTH) { accBin[i1] = bin; hist[bin]++; }
          }
        }
      }
    }
    __syncthreads();
  }
  // ---- rotation histogram: keep the three largest bins (ComputeThreeMaxima, ORBmatcher.cpp:152-183) ----
  if (mp.checkOri) {
    if (t == 0) {
      int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
      for (int i = 0; i < HISTO_LENGTH; i++) {
        const int s = hist[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
      }
      if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
      else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
      sKeep[0] = ind1; sKeep[1] = ind2; sKeep[2] = ind3;
    }
    __syncthreads();
    const int i1k = sKeep[0], i2k = sKeep[1], i3k = sKeep[2];
    int dropped = 0;
    for (int i = t; i < n1; i += MATCH_T) {
      const int b = accBin[i];
      if (b >= 0 && b != i1k && b != i2k && b != i3k) {  // also hits queries whose match was stolen (quirk, :130-138)
        m12[i] = -1;
        dropped++;
      }
    }
    if (dropped) { atomicSub(&sNm, dropped); atomicAdd(&sBadOri, dropped); }
    __syncthreads();
  }
  if (t == 0) {
    nmatchesOut[pair] = sNm;
    if (statsOut) { statsOut[pair * 3] = sBadDist; statsOut[pair * 3 + 1] = sBadRatio; statsOut[pair * 3 + 2] = sBadOri; }
  }
}

// =================================================================================================
// launch wrappers (called from orbx_api.cpp)
// =================================================================================================
hipError_t launch_resize(hipStream_t st, int nFrames, const uint8_t* src, long long srcFrameStride, int sw, int sh, int sstride,
                         uint8_t* dst, long long dstFrameStride, int dw, int dh, int dstride, const ResizeTab* xtab,
                         const ResizeTab* ytab) {
  dim3 block(64, 4, 1), grid((dw + 255) / 256, (dh + 3) / 4, nFrames);
  hipLaunchKernelGGL(k_resize, grid, block, 0, st, src, srcFrameStride, sw, sh, sstride, dst, dstFrameStride, dw, dh, dstride,
                     xtab, ytab);
  return hipGetLastError();
}

hipError_t launch_fast(hipStream_t st, int nFrames, const uint8_t* img0, long long img0FrameStride, int img0Aligned,
                       const uint8_t* pyr, const Geom& g, uint32_t* cand, int* candCount, int* overflow) {
  dim3 block(256, 1, 1), grid(g.nCellsTotal, nFrames, 1);
  hipLaunchKernelGGL(k_fast, grid, block, 0, st, img0, img0FrameStride, img0Aligned, pyr, g, cand, candCount, overflow);
  return hipGetLastError();
}

hipError_t launch_describe(hipStream_t st, int nFrames, int maxSel, const uint8_t* img0, long long img0FrameStride,
                           const uint8_t* pyr, const Geom& g, const SelKp* sel, const int* nsel, orbx_keypoint* kps,
                           uint8_t* desc, int capacity) {
  if (maxSel <= 0) return hipSuccess;
  dim3 block(64, 1, 1), grid(maxSel, nFrames, 1);
  hipLaunchKernelGGL(k_describe, grid, block, 0, st, img0, img0FrameStride, pyr, g, sel, nsel, kps, desc, capacity);
  return hipGetLastError();
}

hipError_t launch_match(hipStream_t st, int nPairs, const int* dFirst, const int* dSecond, const orbx_keypoint* kps,
                        const uint8_t* desc, const int* nkp, int capacity, orbx_bounds b, int window, float nnratio, int checkOri,
                        int* matches12, int* nmatches, int* stats, int* scratch) {
  if (nPairs <= 0) return hipSuccess;
  MatchParams mp;
  mp.capacity = capacity; mp.window = window; mp.nnratio = nnratio; mp.checkOri = checkOri; mp.b = b;
  mp.onlyPending = 1;
  // small pairs: one wave each, LDS resident; whatever it marks MATCH_PENDING is done by the general kernel
  hipLaunchKernelGGL(k_match_wave, dim3(nPairs), dim3(64), 0, st, dFirst, dSecond, kps, desc, nkp, mp, matches12, nmatches,
                     stats);
  hipLaunchKernelGGL(k_match, dim3(nPairs), dim3(MATCH_T), 0, st, dFirst, dSecond, kps, desc, nkp, mp, matches12, nmatches,
                     stats, scratch);
  return hipGetLastError();
}

}  // namespace orbx
