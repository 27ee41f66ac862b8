// orbx_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for the ORB tracking hot path.
//
//   k_pyramid_bands   the whole pyramid of a batch in one launch, row bands with halos (k_resize_dw / k_resize: one
//                     level per launch, for small batches / unaligned input)
//                                                    (ORBextractor::ComputePyramid, cpp:1660-1713 -> cv::resize)
//   k_fast            per-cell FAST-9-16 + in-cell NMS + threshold fallback, survivors into per-cell segments
//                                                    (ComputeKeyPointsOctTree cell loops, cpp:1078-1141 -> cv::FAST)
//   k_describe_patch  IC-angle + 7x7 Gaussian (patch-local, v_dot4/v_dot2 fixed point) + steered BRIEF, one wave per
//                     keypoint                       (IC_Angle cpp:103-159, GaussianBlur cpp:1598-1606,
//                                                     computeOrbDescriptor cpp:169-228, assembly cpp:1557-1652)
//   k_match_jacobi / k_match_wide_lists / k_match_wide_resolve   SearchForInitialization: parallel fixpoint sweeps (one
//                     workgroup per pair up to 256 queries; wide path up to 4096), the sequential loop for the rest
//                                                    (ORBmatcher.cpp:11-183, Frame.cpp:89-99,163-206, FORB.cpp:77-101)
//   k_match_bf_mfma   the wide path's brute-force case (windows that cover every train) when a launch holds enough pairs: all-pairs
//                     Hamming distances as int8 matrix products (v_mfma_i32_32x32x32_i8 on +-1 bytes)
//   k_undistort, k_to_gray, k_check_model     the steps around the path (Frame.cpp:101-161, Converter.cpp:5-19,
//                                                     Initializer.cpp:268-438)
//   (the quadtree selection lives in orbx_octree_kernel.hip)
//
// All arithmetic is integer or uncontracted IEEE f32/f64 (compile with -ffp-contract=off) so the results are
// bit-identical to the CPU restatement in oracle/.  The matrix cores serve k_match_bf_mfma only (exact int8 / int32 arithmetic): the
// all-pairs distances of a brute-force match are the path's one dense contraction; everything else is byte and integer work.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "../../include/orbx.h"
#include "orbx_device.h"
#include "orbx_knobs.h"

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

#define RESIZE_DW_ROWS 4  // output rows per thread of k_resize_dw
struct __attribute__((aligned(4))) PyrU3 { uint32_t a, b, c; };  // 12 bytes from a 4-aligned address: one global_load_dwordx3
// Same arithmetic, but the source taps of the 4 outputs are fetched as 3 aligned dwords per source row (the taps of
// 4 consecutive outputs span at most 11 bytes from the aligned start when the scale is <= 2) instead of 8 byte loads,
// and extracted with v_alignbyte.  Needs 4-byte aligned source rows; the launcher falls back to k_resize otherwise.
__global__ __launch_bounds__(256) void k_resize_dw(const uint8_t* __restrict__ src, long long srcFrameStride, int sw, int sh,
                                                   int sstride, uint8_t* __restrict__ dst, long long dstFrameStride, int dw,
                                                   int dh, int dstride, const ResizeTab* __restrict__ xtab,
                                                   const ResizeTab* __restrict__ ytab, int wideFrames) {
  // RESIZE_DW_ROWS output rows per thread (rows dy, dy + 4, ...: a wave still covers whole row segments): the column taps -- 32
  // bytes of table per thread, eight times the bytes it stores per row -- are loaded once, and the rows' source dwords are all
  // in flight together
  const int f = blockIdx.z;
  const int dx0 = (blockIdx.x * 64 + threadIdx.x) * 4;
  const int dyBase = blockIdx.y * (4 * RESIZE_DW_ROWS) + threadIdx.y;
  if (dx0 >= dw || dyBase >= dh) return;
  const uint4 tA = reinterpret_cast<const uint4*>(xtab + dx0)[0];  // entries dx0, dx0+1 (ofs, coef, ofs, coef)
  const uint4 tB = reinterpret_cast<const uint4*>(xtab + dx0)[1];  // entries dx0+2, dx0+3
  const int sxs[4] = {(int)tA.x, (int)tA.z, (int)tB.x, (int)tB.z};
  const uint32_t cfs[4] = {tA.y, tA.w, tB.y, tB.w};
  const int base = sxs[0] & ~3;
  const int lim = (sw - 1) & ~3;  // last dword that holds a pixel of the row: never read beyond it
  const int o0 = base, o1 = min(base + 4, lim), o2 = min(base + 8, lim);
  const uint8_t* S = src + (long long)f * srcFrameStride;
  uint32_t r0a[RESIZE_DW_ROWS], r0b[RESIZE_DW_ROWS], r0c[RESIZE_DW_ROWS], r1a[RESIZE_DW_ROWS], r1b[RESIZE_DW_ROWS], r1c[RESIZE_DW_ROWS];
  int b0[RESIZE_DW_ROWS], b1[RESIZE_DW_ROWS];
#pragma unroll
  for (int k = 0; k < RESIZE_DW_ROWS; k++) {
    const int dy = min(dyBase + 4 * k, dh - 1);  // (a row beyond the image repeats the last one and is not stored)
    const ResizeTab ty = ytab[dy];
    const int sy0 = min(max(ty.ofs, 0), sh - 1), sy1 = min(max(ty.ofs + 1, 0), sh - 1);
    b0[k] = ty.coef & 0xffff; b1[k] = ty.coef >> 16;
    const uint8_t* S0 = S + (long long)sy0 * sstride;
    const uint8_t* S1 = S + (long long)sy1 * sstride;
    // The group's taps lie in 12 bytes from the aligned start: ONE 12-byte load per source row (the texture addresser charges per
    // instruction and lane -- 18 cycles for this load, 3 x 12 for three dwords at this pitch, tools/microbench/mem_rate.hip --
    // and at 4K the launch was bound by it: 24 us per level, 0.9 TB/s).  It may read up to 11 bytes beyond the row's last pixel:
    // the next row, the next frame, or the slack the buffers end with -- only behind the LAST row of the caller's last frame
    // nothing is known to follow, so frames >= wideFrames take that row (a uniform branch per row) through clamped dwords.
    if (f < wideFrames || sy1 < sh - 1) {
      const PyrU3 u0 = *reinterpret_cast<const PyrU3*>(S0 + o0), u1 = *reinterpret_cast<const PyrU3*>(S1 + o0);
      r0a[k] = u0.a; r0b[k] = u0.b; r0c[k] = u0.c;
      r1a[k] = u1.a; r1b[k] = u1.b; r1c[k] = u1.c;
    } else {
      r0a[k] = *reinterpret_cast<const uint32_t*>(S0 + o0); r0b[k] = *reinterpret_cast<const uint32_t*>(S0 + o1);
      r0c[k] = *reinterpret_cast<const uint32_t*>(S0 + o2);
      r1a[k] = *reinterpret_cast<const uint32_t*>(S1 + o0); r1b[k] = *reinterpret_cast<const uint32_t*>(S1 + o1);
      r1c[k] = *reinterpret_cast<const uint32_t*>(S1 + o2);
    }
  }
#pragma unroll
  for (int k = 0; k < RESIZE_DW_ROWS; k++) {
    const int dy = dyBase + 4 * k;
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int kk = sxs[i] - base;  // 0..10
      const int sh8 = kk & 3;
      // 32-bit window starting at byte kk: low byte = S[sx], next byte = S[sx+1] (a clamped or missing dword can only
      // supply bytes beyond the last pixel, whose weight is 0)
      const uint32_t lo0 = kk < 4 ? r0a[k] : (kk < 8 ? r0b[k] : r0c[k]), hi0 = kk < 4 ? r0b[k] : r0c[k];
      const uint32_t lo1 = kk < 4 ? r1a[k] : (kk < 8 ? r1b[k] : r1c[k]), hi1 = kk < 4 ? r1b[k] : r1c[k];
      const uint32_t w0 = __builtin_amdgcn_alignbyte(hi0, lo0, sh8), w1 = __builtin_amdgcn_alignbyte(hi1, lo1, sh8);
      const int a0 = cfs[i] & 0xffff, a1 = cfs[i] >> 16;
      const int t0 = (int)(w0 & 255) * a0 + (int)((w0 >> 8) & 255) * a1;
      const int t1 = (int)(w1 & 255) * a0 + (int)((w1 >> 8) & 255) * a1;
      int v = (((b0[k] * (t0 >> 4)) >> 16) + ((b1[k] * (t1 >> 4)) >> 16) + 2) >> 2;
      v = min(max(v, 0), 255);
      packed |= (uint32_t)v << (8 * i);
    }
    if (dy < dh) *reinterpret_cast<uint32_t*>(dst + (long long)f * dstFrameStride + (long long)dy * dstride + dx0) = packed;
  }
}

// The whole pyramid of a frame in one launch for SMALL batches (one frame per call is what the reference's Frame constructor
// hands over, SlamTypes/Frame.cpp:58-60): seven per-level launches are a 26 us chain of 3.7 us kernels there.  A workgroup owns
// one tile of the frame on every level; the chain level l <- level l - 1 (cpp:1660-1713) runs inside the workgroup through LDS:
// it stages the part of the caller's image its level-1 rectangle reads and its resize taps (one blob per tile, positions already
// relative to the source rectangle: buildPyrTiles), then computes, level by level, the rectangle its higher levels need (a halo of
// a few pixels around what it owns), keeps it in LDS as the source of the next level, and stores the owned part to the pyramid.
// Global memory is read once, at the start.  Same Q11 arithmetic as k_resize; neighbouring tiles compute their shared halo pixels
// twice and get the same bytes.  LDS: taps | level-0 rectangle | two level buffers.
__global__ __launch_bounds__(256) void k_pyramid_tiles(const uint8_t* __restrict__ img0, long long img0FrameStride,
                                                       uint8_t* __restrict__ pyr, const Geom g, const PyrTileRect* __restrict__ rects,
                                                       const PyrTileTap* __restrict__ taps, int buf0Bytes, int bufBytes) {
  extern __shared__ __attribute__((aligned(16))) uint8_t tileLds[];
  PyrTileTap* T = reinterpret_cast<PyrTileTap*>(tileLds);
  uint8_t* buf0 = tileLds + ORBX_PYR_TILE_TAPS * sizeof(PyrTileTap);
  uint8_t* bufA = buf0 + buf0Bytes;
  uint8_t* bufB = bufA + bufBytes;
  const int f = blockIdx.y + g.frame0, t = blockIdx.x;
  const PyrTileRect* __restrict__ R = rects + (size_t)t * g.nlevels;
  {  // taps and the level-0 rectangle: the only reads from global memory, all in flight together
    int nT = 0;
    for (int l = 1; l < g.nlevels; l++) nT += (R[l].nx1 - R[l].nx0) + (R[l].ny1 - R[l].ny0);
    const uint2* __restrict__ src = reinterpret_cast<const uint2*>(taps + (size_t)t * ORBX_PYR_TILE_TAPS);
    uint2* dst = reinterpret_cast<uint2*>(T);
    uint2 tv[ORBX_PYR_TILE_TAPS / 256];
#pragma unroll
    for (int q = 0; q < ORBX_PYR_TILE_TAPS / 256; q++) tv[q] = (int)threadIdx.x + q * 256 < nT ? src[threadIdx.x + q * 256] : make_uint2(0, 0);
    const PyrTileRect R0 = R[0];
    const int w0 = R0.nx1 - R0.nx0, n0 = w0 * (R0.ny1 - R0.ny0);
    const uint32_t inv0 = ((1u << 20) + (uint32_t)w0 - 1) / (uint32_t)max(w0, 1);
    const uint8_t* G0 = img0 + (long long)f * img0FrameStride + (long long)R0.ny0 * g.L[0].stride + R0.nx0;
    for (int base = 0; base < n0; base += 8 * 256) {
      uint8_t pv[8];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int idx = min(base + q * 256 + (int)threadIdx.x, n0 - 1);
        const int y = (int)(((uint32_t)idx * inv0) >> 20), x = idx - y * w0;  // (exact: idx < 2^11 * 8, w0 < 512)
        pv[q] = G0[(long long)y * g.L[0].stride + x];
      }
#pragma unroll
      for (int q = 0; q < 8; q++)
        if (base + q * 256 + (int)threadIdx.x < n0) buf0[base + q * 256 + threadIdx.x] = pv[q];
    }
#pragma unroll
    for (int q = 0; q < ORBX_PYR_TILE_TAPS / 256; q++)
      if ((int)threadIdx.x + q * 256 < nT) dst[threadIdx.x + q * 256] = tv[q];
  }
  __syncthreads();
  const uint8_t* srcL = buf0;
  int ppitch = R[0].nx1 - R[0].nx0;
  uint8_t* dstL = bufA;
  int tapOff = 0;
  for (int l = 1; l < g.nlevels; l++) {
    const LevelGeom& D = g.L[l];
    const PyrTileRect Rl = R[l];
    const int nw = Rl.nx1 - Rl.nx0, nh = Rl.ny1 - Rl.ny0, npx = nw * nh;
    const uint32_t inv = ((1u << 20) + (uint32_t)nw - 1) / (uint32_t)max(nw, 1);
    const PyrTileTap* TX = T + tapOff;
    const PyrTileTap* TY = TX + nw;
    uint8_t* out = pyr + D.imgOff + (long long)f * D.frameStride;
    for (int base = 0; base < npx; base += 4 * 256) {  // four pixels per thread in flight: each is a chain tap -> source pixels
      int pi[4], pj[4];
      PyrTileTap ex[4], ey[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int idx = min(base + q * 256 + (int)threadIdx.x, npx - 1);
        pj[q] = (int)(((uint32_t)idx * inv) >> 20);
        pi[q] = idx - pj[q] * nw;
        ex[q] = TX[pi[q]];
        ey[q] = TY[pj[q]];
      }
      int p00[4], p01[4], p10[4], p11[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint8_t* S0 = srcL + (int)(ey[q].pos & 0xffff) * ppitch + (int)(ex[q].pos & 0xffff);
        const uint8_t* S1 = S0 + (int)(ey[q].pos >> 16) * ppitch;
        const int d = (int)(ex[q].pos >> 16);
        p00[q] = S0[0]; p01[q] = S0[d]; p10[q] = S1[0]; p11[q] = S1[d];
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int a0 = ex[q].coef & 0xffff, a1 = ex[q].coef >> 16, b0 = ey[q].coef & 0xffff, b1 = ey[q].coef >> 16;
        const int t0 = p00[q] * a0 + p01[q] * a1;
        const int t1 = p10[q] * a0 + p11[q] * a1;
        int v = (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2;
        v = min(max(v, 0), 255);
        if (base + q * 256 + (int)threadIdx.x < npx) {
          dstL[pj[q] * nw + pi[q]] = (uint8_t)v;
          const int x = Rl.nx0 + pi[q], y = Rl.ny0 + pj[q];
          if (y >= Rl.oy0 && y < Rl.oy1 && x >= Rl.ox0 && x < Rl.ox1) out[(long long)y * D.stride + x] = (uint8_t)v;
        }
      }
    }
    __syncthreads();
    tapOff += nw + nh;
    ppitch = nw;
    srcL = dstL;
    dstL = (dstL == bufA) ? bufB : bufA;
  }
}

typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t dot2u16(uint32_t a, uint32_t b, uint32_t c) {  // v_dot2_u32_u16
  return __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, a), __builtin_bit_cast(ushort2v, b), c, false);
}

// The whole pyramid of a frame in ONE launch.  The chain level l <- level l-1 is kept (cpp:1660-1713), but instead of one
// launch per level a workgroup owns a horizontal band of the frame: the band's rows of the last level need a slightly
// larger band of the level before, and so on up to level 0 (PyrBands, computed on the host from the same y tables).  The
// workgroup produces its bands level by level, re-reading what it wrote itself a moment ago (L2 hits) behind a fence +
// barrier; bands of neighbouring workgroups overlap by a few rows and write identical bytes there.  No inter-workgroup
// synchronisation, no launch gaps, and the small upper levels no longer pay a launch each.
//
// Arithmetic = k_resize's (cv::resize INTER_LINEAR 8UC1, Q11 taps), arranged for the two budgets that
// bound it (tools/microbench: valu_rate = vector issue, add / and / shift-right 2 cycles per wave64 instruction, everything
// else 4; mem_rate = the CU's one texture addresser, 6 cycles per dword load of a wave, 34 per byte-aligned 8-byte load):
//   * the taps of 4 consecutive outputs lie within 12 source bytes from a 4-aligned start (scale <= 2): three aligned
//     dword loads per source row and group (measured cheaper than one unaligned 8-byte window: the addresser, not the
//     vector pipe, bounded that variant).  The byte pair of pixels 0..2 always sits in dwords (0, 1); pixel 3 (and pixel 2
//     at scales above ~1.3, template DUAL2) may need dwords (1, 2): it takes one v_perm_b32 per dword pair, the one that does
//     not apply with an all-zero selector, and an OR.  All selectors depend on the column only (host table PyrXGroup)
//   * horizontal: v_dot2_u32_u16 with the Q11 pair pre-multiplied by 16, t' = 16 t
//   * vertical: (wy * (t >> 4)) >> 16 == v_mul_hi_u32_u24(wy << 8, t' & ~0xff)  (both operands < 2^24; bits 47:32 of the
//     product); the row offsets and wy << 8 come ready-made from the host table PyrYRow
//   * the four sums of a group are rounded, shifted and packed two at a time; no clamp: with tap pairs summing to 2048
//     (checked on the host, else the per-level kernels run) the result is <= 255
__device__ __forceinline__ uint32_t mulHiU24(uint32_t a, uint32_t b) {  // v_mul_hi_u32_u24: bits 47:32 of the 24 x 24 bit product
  return (uint32_t)(((unsigned long long)(a & 0xffffffu) * (unsigned long long)(b & 0xffffffu)) >> 32);
}
// Diagnostic build only (-DORBX_PYR_STAMPS): per workgroup, s_memtime at the start, after the first barrier, and per level after
// the rows and after the barrier; s_memrealtime at start and end; tools/pyr_stamps.py prints where a workgroup's time goes.
#ifdef ORBX_PYR_STAMPS
#define PYR_STAMP_WGS (1 << 14)
__device__ uint32_t g_pyrStamps[PYR_STAMP_WGS * 40];
#define PYR_STAMP(k)                                                                                          \
  do {                                                                                                        \
    if (threadIdx.x == 0 && stampWg_ < PYR_STAMP_WGS) g_pyrStamps[stampWg_ * 40 + (k)] = (uint32_t)__builtin_amdgcn_s_memtime(); \
  } while (0)
#define PYR_STAMP_RT(k)                                                                                       \
  do {                                                                                                        \
    if (threadIdx.x == 0 && stampWg_ < PYR_STAMP_WGS) g_pyrStamps[stampWg_ * 40 + (k)] = (uint32_t)__builtin_amdgcn_s_memrealtime(); \
  } while (0)
extern "C" int orbx_diag_pyr_stamps(uint32_t* out, int nWgs) {  // out: nWgs x 40 dwords; nWgs < 0: clear the buffer
  if (nWgs < 0) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_pyrStamps)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(uint32_t) * 40 * PYR_STAMP_WGS);
  }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pyrStamps), sizeof(uint32_t) * 40 * (size_t)nWgs);
}
#else
#define PYR_STAMP(k) do { } while (0)
#define PYR_STAMP_RT(k) do { } while (0)
#endif
#ifndef PYR_T
#define PYR_T 512   // threads per workgroup: 512 fill their lanes with whole rows better than 256 (level 1 of 640x480: 469 of 512
                    // against 201 of 256) and take three fat bands per frame (3.5 % shared rows instead of 11 % with seven)
#endif
#define PYR_R 2   // output rows in flight per thread and step
// GMAX = groups of 4 pixels a thread may own: 2 in general; 1 (round 5) when no level (strip) is wider than PYR_T groups -- every
// frame up to 2457 pixels wide, and the strips of larger ones -- which keeps one set of column constants and row registers
template <int DUAL2, int GMAX>
__global__ __launch_bounds__(PYR_T) void k_pyramid_bands(const uint8_t* __restrict__ img0, long long img0FrameStride,
                                                       uint8_t* __restrict__ pyr, const Geom g,
                                                       const uint4* __restrict__ ptab, const PyrBands pb) {
  // the band's PyrYRow entries of the current level (and, filled meanwhile, of the next one): a step's row constants then
  // cost an LDS read instead of a global load in front of the source loads that depend on them
  ORBX_SETPRIO();
  extern __shared__ uint4 yrows[];  // [2][pb.maxRows]
  const int f = blockIdx.y + g.frame0, strip = (int)blockIdx.x / pb.nBands, band = (int)blockIdx.x - strip * pb.nBands, tid = threadIdx.x;
  const int nl = g.nlevels;
#ifdef ORBX_PYR_STAMPS
  const unsigned stampWg_ = blockIdx.y * gridDim.x + blockIdx.x;
#endif
  PYR_STAMP(0);
  PYR_STAMP_RT(36);
  // column constants of this thread for level l
  struct Cols {
    uint32_t o[GMAX], sel[GMAX][6], cf[GMAX][4];
    int G, rpp, rsub, col, gOff;
    bool lanes, haveG[GMAX];
  };
  auto loadCols = [&](int l, Cols& c) {
    c.gOff = pb.g0[strip][l];
    const int ng = max(pb.g1[strip][l] - c.gOff, 1);   // groups of 4 output pixels per row of this workgroup's strip
    // A thread keeps its column(s): it owns G adjacent groups (8 or 4 output pixels) of one thread-column and walks down the
    // band's rows, so the column constants of its pixels are loaded ONCE per level and stay in registers.  rpp rows are
    // covered per pass; G (1 or 2) = whichever fills more of the 256 lanes with whole rows (134 groups: 67 thread-columns
    // x 3 rows = 201 of 256 lanes).  (The host takes this kernel only when every level >= 1 is at most 4096 pixels wide,
    // i.e. at most PYR_T = 512 thread-columns of two groups: 3840x2160 frames qualify.)
    const int ngt2 = (ng + 1) >> 1;
    const int use1 = ng <= PYR_T ? ng * (PYR_T / ng) : 0, use2 = ngt2 * (PYR_T / ngt2);
    c.G = GMAX == 2 && use2 > use1 ? 2 : 1;
    const int ngt = c.G == 2 ? ngt2 : ng;
    c.rpp = max(PYR_T / ngt, 1);
    c.rsub = tid / ngt;
    c.col = tid - c.rsub * ngt;
    c.lanes = c.rsub < c.rpp;
    const uint4* xg = ptab + pb.xoff[l];
#pragma unroll
    for (int j = 0; j < GMAX; j++) {
      const int gx = c.col * c.G + j;
      c.haveG[j] = c.lanes && j < c.G && gx < pb.g1[strip][l] - c.gOff;
      const uint4* e = xg + (unsigned)(c.haveG[j] ? c.gOff + gx : 0) * 4u;
      const uint4 X0 = e[0], X1 = e[1], X2 = e[2];
      const uint32_t X3 = e[3].x;
      c.o[j] = X0.x;
      c.sel[j][0] = X0.w; c.sel[j][1] = X1.x; c.sel[j][2] = X1.y; c.sel[j][3] = X1.z; c.sel[j][4] = X1.w; c.sel[j][5] = X2.x;
      c.cf[j][0] = X2.y; c.cf[j][1] = X2.z; c.cf[j][2] = X2.w; c.cf[j][3] = X3;
    }
  };
  Cols C;
  loadCols(1, C);
  {
    const int r0 = pb.r0[band][1], n = pb.r1[band][1] - r0;
    if (tid < n) yrows[pb.maxRows + tid] = (ptab + pb.yoff[1])[r0 + tid];
  }
  __syncthreads();
  PYR_STAMP(1);
  for (int l = 1; l < nl; l++) {
    const LevelGeom& S = g.L[l - 1];
    const LevelGeom& D = g.L[l];
    const uint8_t* src = l == 1 ? img0 + (long long)f * img0FrameStride : pyr + S.imgOff + (long long)f * S.frameStride;
    uint8_t* dst = pyr + D.imgOff + (long long)f * D.frameStride;
    const uint4* yr = yrows + (l & 1) * pb.maxRows;
    const int r0 = pb.r0[band][l], r1 = pb.r1[band][l];
    const int dstride = D.stride;
    // the next level's row constants are fetched now and used after the barrier: their latency runs under this level's rows
    uint4 nextY = make_uint4(0, 0, 0, 0);
    int nextN = 0;
    if (l + 1 < nl) {
      const int q0 = pb.r0[band][l + 1];
      nextN = pb.r1[band][l + 1] - q0;
      if (tid < nextN) nextY = (ptab + pb.yoff[l + 1])[q0 + tid];
    }
    const int G = C.G, rpp = C.rpp;
    // The three dwords of a group come as ONE 12-byte load (the addresser's cost is per instruction and lane: 18 cycles, against
    // 3 x 12 for single dwords at this pitch).  At the end of a row that reads up to 11 bytes beyond the last pixel: the next
    // row, the next level, the next frame, or the slack the pyramid buffer ends with; only behind the last row of the
    // caller's last frame nothing is known to follow, so frames f >= pb.safeFrom take level 1 with clamped single dwords.
    // Round 5: a step's two rows are ADJACENT output rows dy, dy + 1.  At the pyramid's scale 1.2 the lower source row of dy is
    // the upper source row of dy + 1 for five row pairs of six (same byte offset in PyrYRow): that row is loaded once and its
    // horizontal interpolation (perms, dot2, mask: 14 of a group's 50 instructions) is computed once for both outputs -- three
    // loads and three interpolations per row pair and group instead of four.  A pair that does not share (every pair at scale 2)
    // takes the fourth behind one wave-uniform test; values are those of the separate evaluation either way (same source bytes,
    // same column constants).  (Before: rows dy and dy + rpp, four of each; a timing-only build with a fifth less pyramid work
    // gave the step +3 %.)
    static_assert(PYR_R == 2, "the row loop pairs adjacent rows");
    auto rows = [&](auto safeTag) {
    constexpr bool SAFE = decltype(safeTag)::value;
    auto load3 = [&](const uint32_t rowOff, const int j) -> PyrU3 {
      PyrU3 v;
      if (SAFE) {
        // a dword beyond the row's last one is replaced by the last one and can only supply bytes of weight 0
        const uint32_t lim = (uint32_t)(S.w - 1) & ~3u, o0 = C.o[j], o1 = min(o0 + 4u, lim), o2 = min(o0 + 8u, lim);
        v.a = *reinterpret_cast<const uint32_t*>(src + (rowOff + o0));
        v.b = *reinterpret_cast<const uint32_t*>(src + (rowOff + o1));
        v.c = *reinterpret_cast<const uint32_t*>(src + (rowOff + o2));
      } else {
        v = *reinterpret_cast<const PyrU3*>(src + (rowOff + C.o[j]));
      }
      return v;
    };
    // horizontal interpolation of one source row for the group's four pixels: 16 t, low byte masked (see mulHiU24 above)
    auto hrow = [&](const PyrU3 a, const int j, uint32_t (&t)[4]) {
      uint32_t p[4];
      p[0] = __builtin_amdgcn_perm(a.b, a.a, C.sel[j][0]);
      p[1] = __builtin_amdgcn_perm(a.b, a.a, C.sel[j][1]);
      p[2] = __builtin_amdgcn_perm(a.b, a.a, C.sel[j][2]);
      if (DUAL2) p[2] |= __builtin_amdgcn_perm(a.c, a.b, C.sel[j][3]);
      p[3] = __builtin_amdgcn_perm(a.b, a.a, C.sel[j][4]) | __builtin_amdgcn_perm(a.c, a.b, C.sel[j][5]);
#pragma unroll
      for (int i = 0; i < 4; i++) t[i] = dot2u16(p[i], C.cf[j][i], 0u) & ~0xffu;
    };
    for (int dy0 = r0 + 2 * C.rsub; dy0 < r1; dy0 += 2 * rpp) {  // two adjacent rows per step: their loads are in flight together
      const bool live0 = C.lanes && dy0 < r1, live1 = C.lanes && dy0 + 1 < r1;
      const uint4 ty0 = yr[live0 ? dy0 - r0 : 0], ty1 = yr[live1 ? dy0 + 1 - r0 : 0];
      const bool shared = ty1.x == ty0.y;                               // the rows' common source row
      const bool fourth = __ballot(live1 && !shared) != 0ull;           // (wave-uniform) some lane's pair does not share one
      PyrU3 ra0[GMAX], rb0[GMAX], ra1[GMAX], rb1[GMAX];
#pragma unroll
      for (int j = 0; j < GMAX; j++)
        if (j < G) {
          ra0[j] = load3(ty0.x, j);
          rb0[j] = load3(ty0.y, j);
          rb1[j] = load3(ty1.y, j);
          if (fourth) ra1[j] = load3(ty1.x, j);
        }
#pragma unroll
      for (int j = 0; j < GMAX; j++) {
        if (j >= G) continue;
        uint32_t ta0[4], tb0[4], ta1[4], tb1[4];
        hrow(ra0[j], j, ta0);
        hrow(rb0[j], j, tb0);
        hrow(rb1[j], j, tb1);
        if (fourth) {
          hrow(ra1[j], j, ta1);
#pragma unroll
          for (int i = 0; i < 4; i++) ta1[i] = shared ? tb0[i] : ta1[i];
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++) ta1[i] = tb0[i];
        }
        uint32_t u[4], v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          u[i] = mulHiU24(ty0.z, ta0[i]) + mulHiU24(ty0.w, tb0[i]);
          v[i] = mulHiU24(ty1.z, ta1[i]) + mulHiU24(ty1.w, tb1[i]);
        }
        // (u + 2) >> 2 on two 16-bit fields at a time; bytes 0 and 2 of each pair are the pixels
        const uint32_t s01 = ((u[0] | (u[1] << 16)) + 0x00020002u) >> 2, s23 = ((u[2] | (u[3] << 16)) + 0x00020002u) >> 2;
        const uint32_t q01 = ((v[0] | (v[1] << 16)) + 0x00020002u) >> 2, q23 = ((v[2] | (v[3] << 16)) + 0x00020002u) >> 2;
        const unsigned col = (unsigned)((C.gOff + C.col * G + j) * 4);
        if (live0 && C.haveG[j]) *reinterpret_cast<uint32_t*>(dst + ((unsigned)(dy0 * dstride) + col)) = __builtin_amdgcn_perm(s23, s01, 0x06040200u);
        if (live1 && C.haveG[j]) *reinterpret_cast<uint32_t*>(dst + ((unsigned)((dy0 + 1) * dstride) + col)) = __builtin_amdgcn_perm(q23, q01, 0x06040200u);
      }
    }
    };
    if (l == 1 && f >= pb.safeFrom) rows(std::true_type{});
    else rows(std::false_type{});
    PYR_STAMP(2 * l);
    if (tid < nextN) yrows[((l + 1) & 1) * pb.maxRows + tid] = nextY;
    if (l + 1 < nl) loadCols(l + 1, C);
    // this band of level l is the source of the band of level l + 1 in the same workgroup: the barrier's workgroup-scope
    // release/acquire is all that is needed (the CU's vector L1 is coherent for its own workgroup; an agent-scope
    // fence would write back / invalidate L2 once per level and workgroup and was measured 7x slower)
    __syncthreads();
    PYR_STAMP(2 * l + 1);
  }
  PYR_STAMP_RT(37);
#ifdef ORBX_PYR_STAMPS
  if (threadIdx.x == 0 && stampWg_ < PYR_STAMP_WGS) {
    g_pyrStamps[stampWg_ * 40 + 38] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
    g_pyrStamps[stampWg_ * 40 + 39] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
  }
#endif
}

// =================================================================================================
// K2  FAST-9-16 per cell.  One workgroup (256 threads) per (cell, frame).
//   strength(p) = max over the 16 arcs of 9 contiguous ring pixels of min(+-(v - p_k))
//   corner at threshold t  <=>  strength > t ; score = strength - 1
//   in-cell NMS is threshold independent on the strength map:  keep <=> s > all 8 neighbours' s
//   (neighbours outside the cell's detection area count as 0) and s > 1
//   cell fallback (cpp:1117-1123): no survivor with s > iniTh  =>  use minTh for the whole cell
// =================================================================================================
#define TILE_STRIDE 84   // bytes per LDS tile row (21 words); k_fast divides tile offsets by it with (x * 3121) >> 18
static_assert(84 * 3121 > (1 << 18) && 83 * 3121 < (1 << 18), "reciprocal of TILE_STRIDE");
#define SMAP_STRIDE 72   // 70 + 2 zero apron
// threads per (cell, frame) workgroup: 256 measured best (128 and 512 are 7 % and 60 % slower)
#define FAST_T 256
struct FastLds {
  int32_t tileBytes, smapBytes, listBytes, outCap;
};

// ceil(2^20 / n) for n = 1..79: x / n == (x * c_inv20[n]) >> 20 for x < 2^20 / n, so the per-thread index splits of k_fast
// cost a multiply and a shift instead of an integer division (uses: x < 256 for any n; x < 21 * 76 for n <= 21;
// x < 76 * 76 for 7 <= n <= 76 -- the products stay below 2^32)
struct Inv20Table {
  uint32_t v[80];
};
constexpr Inv20Table makeInv20() {
  Inv20Table t{};
  for (int n = 1; n < 80; n++) t.v[n] = ((1u << 20) + (uint32_t)n - 1u) / (uint32_t)n;
  return t;
}
__constant__ Inv20Table c_inv20 = makeInv20();

__device__ __forceinline__ bool arc9(uint32_t m) {  // 16-bit circular mask has a run of >= 9 ones
  m |= m << 16;
  uint32_t r = m & (m >> 1);
  r &= r >> 2;
  r &= r >> 4;
  r &= m >> 8;
  return (r & 0xFFFFu) != 0;
}

__global__ __launch_bounds__(FAST_T) void k_fast(const uint8_t* __restrict__ img0, long long img0FrameStride, int img0Aligned,
                                              const uint8_t* __restrict__ pyr, const Geom g,
                                              uint32_t* __restrict__ cand, int* __restrict__ cellCount,
                                              const FastLds fl) {
  // LDS carved to the largest cell of this geometry (launch_fast), so that occupancy is bound by waves, not by LDS:
  // tile[tileBytes] | strength map[smapBytes] | quick-reject list u16[listCap] | survivors u32[outCap]
  extern __shared__ __attribute__((aligned(16))) uint8_t fastLds[];
  uint8_t* const tile = fastLds;
  uint8_t* const smap = fastLds + fl.tileBytes;
  uint16_t* const list = reinterpret_cast<uint16_t*>(fastLds + fl.tileBytes + fl.smapBytes);
  uint32_t* const outl = reinterpret_cast<uint32_t*>(fastLds + fl.tileBytes + fl.smapBytes + fl.listBytes);
  __shared__ int nList, nOut;

  const int t = threadIdx.x;
  const int f = blockIdx.y + g.frame0;
  int level = 0;
  const int cid = blockIdx.x;
  while (level + 1 < g.nlevels && cid >= g.L[level + 1].cellBase) level++;
  const LevelGeom& L = g.L[level];
  const int local = cid - L.cellBase;
  const int ci = (int)(((uint32_t)local * L.colsInv24) >> 24), cj = local - ci * L.nCols;  // local / nCols
  // cell rectangle, cpp:1082-1103
  const int iniY = ORBX_MIN_BORDER + ci * L.hCell;
  const int iniX = ORBX_MIN_BORDER + cj * L.wCell;
  // every cell writes its counter (also 0), so the counters need no clearing between batches
  int* const myCount = cellCount + (long long)f * g.nCellsTotal + cid;
  if (iniY >= L.maxBY - 3 || iniX >= L.maxBX - 6) {
    if (t == 0) *myCount = 0;
    return;
  }
  const int maxY = min(iniY + L.hCell + 6, L.maxBY), maxX = min(iniX + L.wCell + 6, L.maxBX);
  const int cw = maxX - iniX, ch = maxY - iniY;
  if (cw < 7 || ch < 7) {  // cv::FAST finds nothing in an image this small
    if (t == 0) *myCount = 0;
    return;
  }
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
    const int nw = (maxX - ax0 + 3) >> 2;  // <= 21
    const uint32_t invNw = c_inv20.v[nw];
    uint32_t* tile32 = reinterpret_cast<uint32_t*>(tile);
    // (row, word) of this thread's first dword, then idx += FAST_T  <=>  (c, r) += (FAST_T % nw, FAST_T / nw): one 32-bit
    // offset from the uniform base pointer and one LDS index, both stepped by constants (no per-iteration address math)
    const int r0 = (int)(((uint32_t)t * invNw) >> 20), c0 = t - r0 * nw;
    const int dr = (int)(((uint32_t)FAST_T * invNw) >> 20), dc = FAST_T - dr * nw;
    uint32_t goff = (uint32_t)((iniY + r0) * stride + ax0 + 4 * c0);
    const uint32_t dGoff = (uint32_t)(dr * stride + 4 * dc), wrapG = (uint32_t)(stride - 4 * nw);
    int lidx = r0 * (TILE_STRIDE / 4) + c0, c = c0;
    const int dLidx = dr * (TILE_STRIDE / 4) + dc, wrapL = TILE_STRIDE / 4 - nw;
#pragma unroll 1
    for (int idx = t; idx < nw * ch; idx += FAST_T) {
      tile32[lidx] = *reinterpret_cast<const uint32_t*>(base + goff);
      c += dc; goff += dGoff; lidx += dLidx;
      if (c >= nw) { c -= nw; goff += wrapG; lidx += wrapL; }
    }
  } else {
    const uint32_t invCw = c_inv20.v[cw];  // 7 <= cw <= 76, idx < 76 * 76
    for (int idx = t; idx < cw * ch; idx += FAST_T) {
      const int r = (int)(((uint32_t)idx * invCw) >> 20), c = idx - r * cw;
      tile[r * TILE_STRIDE + c] = base[(long long)(iniY + r) * stride + iniX + c];
    }
  }
  {
    uint4* smap128 = reinterpret_cast<uint4*>(smap);  // smapBytes is a multiple of 16
    for (int idx = t; idx < fl.smapBytes / 16; idx += FAST_T) smap128[idx] = make_uint4(0u, 0u, 0u, 0u);
  }
  if (t == 0) { nList = 0; nOut = 0; }
  __syncthreads();

  // ring offsets inside the LDS tile, k = 0..15 (dx,dy) = (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)...
  constexpr int RS = TILE_STRIDE;
  constexpr int ro[16] = {3 * RS,      3 * RS + 1,  2 * RS + 2,  RS + 3,  3,       -RS + 3,     -2 * RS + 2, -3 * RS + 1,
                          -3 * RS,     -3 * RS - 1, -2 * RS - 2, -RS - 3, -3,      RS - 3,      2 * RS - 2,  3 * RS - 1};
  // per-thread pixel walk without divisions inside the loops: idx += FAST_T  <=>  (px, py) += (FAST_T % iw, FAST_T / iw)
  const int py0 = (int)(((uint32_t)t * c_inv20.v[iw]) >> 20), px0 = t - py0 * iw;  // t / iw, iw <= 70
  const int dpy = (int)(((uint32_t)FAST_T * c_inv20.v[iw]) >> 20), dpx = FAST_T - dpy * iw;
  const int npix = iw * ih;
  const int off0 = (py0 + 3) * TILE_STRIDE + xoff + px0 + 3, dOff = dpy * TILE_STRIDE + dpx;
  // The reference runs cv::FAST at iniThFAST and, only if the cell yields nothing, again at minThFAST (cpp:1109-1123).
  // Same here: pass 0 at iniTh, pass 1 at minTh only for cells without a survivor.  The strength map is threshold
  // independent, so what pass 0 wrote stays valid for pass 1.
  for (int pass = 0; pass < 2; pass++) {
    const int th = pass == 0 ? g.iniTh : g.minTh;
    // ---- phase 0: necessary condition on the 4 compass pixels (an arc of 9 holds two adjacent ones) ----
    {
      // the pixel is carried as its byte offset in the LDS tile (row py + 3, column xoff + px + 3): one add per step, and
      // the same offset is the list entry that phase 1 dereferences directly
      int px = px0, off = off0;
      for (int idx = t; idx < npix; idx += FAST_T) {
        const uint8_t* p = &tile[off];
        const int v = p[0], hi = v + th, lo = v - th;
        const int q0 = p[ro[0]], q4 = p[ro[4]], q8 = p[ro[8]], q12 = p[ro[12]];
        const bool b0 = q0 > hi, b4 = q4 > hi, b8 = q8 > hi, b12 = q12 > hi;
        const bool d0 = q0 < lo, d4 = q4 < lo, d8 = q8 < lo, d12 = q12 < lo;
        const bool cand0 = ((b0 | b8) & (b4 | b12)) | ((d0 | d8) & (d4 | d12));
        if (cand0) list[atomicAdd(&nList, 1)] = (uint16_t)off;
        px += dpx;
        off += dOff;
        if (px >= iw) { px -= iw; off += TILE_STRIDE - iw; }
      }
    }
    __syncthreads();
    const int nl = nList;
    // ---- phase 1: exact strength of the remaining pixels; corners (s > th) enter the strength map ----
    for (int e = t; e < nl; e += FAST_T) {
      const int off = list[e];
      const uint8_t* p = &tile[off];
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
      if (s > th) {
        const int row = (off * 3121) >> 18;  // off / TILE_STRIDE (84): exact for off < 6500
        smap[(row - 2) * SMAP_STRIDE + (off - row * TILE_STRIDE - xoff - 2)] = (uint8_t)s;  // [(py + 1)][px + 1]
      } else {
        list[e] = 0xFFFF;
      }
    }
    __syncthreads();
    // ---- phase 2: in-cell NMS on the strength map; survivors are the cell's keypoints ----
    for (int e = t; e < nl; e += FAST_T) {
      const int off = list[e];
      if (off == 0xFFFF) continue;
      const int row = (off * 3121) >> 18;
      const int py = row - 3, px = off - row * TILE_STRIDE - xoff - 3;
      const uint8_t* q = &smap[(py + 1) * SMAP_STRIDE + px + 1];
      // all nine reads are issued together (short-circuit tests would chain nine LDS round trips)
      const int s = q[0];
      const int n0 = q[-SMAP_STRIDE - 1], n1 = q[-SMAP_STRIDE], n2 = q[-SMAP_STRIDE + 1], n3 = q[-1], n4 = q[1],
                n5 = q[SMAP_STRIDE - 1], n6 = q[SMAP_STRIDE], n7 = q[SMAP_STRIDE + 1];
      const int nmax = max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
      const bool keep = s > 1 && s > nmax;
      if (keep) {
        const int slot = atomicAdd(&nOut, 1);
        if (slot < fl.outCap) outl[slot] = packCand(px + 3 + cj * L.wCell, py + 3 + ci * L.hCell, s - 1);
      }
    }
    __syncthreads();
    if (nOut > 0 || pass == 1 || g.minTh >= g.iniTh) break;
    if (t == 0) nList = 0;  // retry the whole cell at minThFAST
    __syncthreads();
  }
  // The cell's survivors go to the cell's own segment of the (frame, level) candidate area, and their number to the cell's
  // counter: plain stores, no atomic and no barrier (a returning global atomic per cell kept the workgroup's slot
  // occupied for a memory round trip and serialised the cells of a level).  The selection stage gathers the segments.
  const int no = min(nOut, fl.outCap);  // nOut <= outCap = segCap: NMS survivors are never 8-neighbours
  if (t == 0) *myCount = no;
  if (no > 0) {
    uint32_t* dstc = cand + L.candOff + (long long)f * L.candCap + (long long)local * L.segCap;
    for (int e = t; e < no; e += FAST_T) dstc[e] = outl[e];
  }
}

// -------------------------------------------------------------------------------------------------
// k_fast_wave: the same stage with ONE WAVE PER CELL and no workgroup barrier anywhere.  k_fast above spends a 256-thread
// workgroup on a cell of ~1400 pixels: five pixels per thread, and around them every wave pays the whole scalar prologue
// (level search, cell rectangle, 64-bit addresses), six workgroup barriers and the loop control -- 0.83 scalar instructions
// per vector instruction (r01 counters) -- and its 147,712 workgroups per 256 frames arrive at the dispatcher's limit
// (~400 workgroups per microsecond chip-wide), which is why nothing done inside that kernel ever moved its 0.34 ms.
// Here a wave owns the cell:
//   * prologue = one s_load_dwordx8 of the cell record the host precomputed (FastCell);
//   * the cell image (<= 64 rows x 16 dwords) is staged lane = tile row, three 16-byte loads per lane in flight;
//   * the quick reject works on quads of four horizontally adjacent pixels per lane (dword LDS reads, packed 16-bit
//     arithmetic); its survivors go onto a stack in LDS (ballot + mbcnt compaction, no atomics) and are evaluated 64 at a
//     time as soon as a full wave of them is waiting, so no list of all pixels is kept;
//   * corners enter the strength map and a corner list (<= 256; a fuller cell is scanned instead), in-cell NMS runs over
//     that list, survivors are stored straight into the cell's segment;
//   * list appends are stores under exec = the ballot (round 6; before: unconditional stores, a lane with nothing to append
//     writing its own dummy slot), so the hot loops select no addresses and rebuild no lane bits;
//   * workgroup -> cell map is XCD-aware (see the kernel).
// Same arithmetic and the same outputs (a cell's survivors in another order, which the selection stage does not see).
// Preconditions (launch_fast): 4-byte aligned level-0 rows, every cell image <= 61 x 64 pixels; otherwise k_fast runs.
// -------------------------------------------------------------------------------------------------
#define FW_RING 192   // survivor stack, entries (u16 LDS addresses): fewer than 64 waiting + at most 128 new ones per half step
                      // (LDS is granted in pieces of 1280 bytes on gfx950: at 640x480 the kernel's 6208 bytes are five of them, 25
                      // waves per CU; with a stack of 320 entries it took six and ran 21)
#define FW_CORN 256   // corner list, entries
#ifndef FW_CPW
#define FW_CPW 1      // consecutive cells per wave (measured per 256 frames: 1: 0.275 ms, 2: 0.288, 4: 0.316, 8: 0.361: the tail grows)
#endif
#ifndef ORBX_FAST_FLATSKIP
#define ORBX_FAST_FLATSKIP 1  // 0 = comparison build: every cell without a survivor is swept again at minThFAST (tools/bench_real_images.py)
#endif
#ifndef FW_XK
#define FW_XK 16      // groups of FW_CPW cells per run of the XCD-aware order (8 .. 64 measured equal)
#endif

__device__ __forceinline__ int fwMbcnt(unsigned long long m) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// LDS through 32-bit addresses (round 6): k_fast_wave carries a pixel as the LDS ADDRESS of the top-left byte of the 7 x 7 block
// around it, so that every read of the quick reject, of the arc strength and of the NMS is an immediate offset from one
// register (the 16-bit offset field of the DS instructions is unsigned: with centre-relative offsets half of the ring reads
// needed an address register each)
typedef __attribute__((address_space(3))) uint8_t fw_lds_u8;
typedef __attribute__((address_space(3))) uint16_t fw_lds_u16;
typedef __attribute__((address_space(3))) uint32_t fw_lds_u32;
__device__ __forceinline__ const fw_lds_u8* fwLds8(const uint32_t a) { return reinterpret_cast<const fw_lds_u8*>((size_t)a); }
__device__ __forceinline__ const fw_lds_u16* fwLds16(const uint32_t a) { return reinterpret_cast<const fw_lds_u16*>((size_t)a); }
__device__ __forceinline__ const fw_lds_u32* fwLds32(const uint32_t a) { return reinterpret_cast<const fw_lds_u32*>((size_t)a); }
// store by the lanes of a SCALAR mask (a ballot, cut down by scalar masks): exec = m around one DS write -- no vector
// instruction selects an address or rebuilds the lane's own bit (all 64 lanes are active where these are used)
__device__ __forceinline__ void fwStoreB16(const uint32_t addr, const uint32_t val, const unsigned long long m) {
  unsigned long long saved;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tds_write_b16 %2, %3\n\ts_mov_b64 exec, %0"
               : "=&s"(saved)
               : "s"(m), "v"(addr), "v"(val)
               : "memory");
}
__device__ __forceinline__ void fwStoreB8(const uint32_t addr, const uint32_t val, const unsigned long long m) {
  unsigned long long saved;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tds_write_b8 %2, %3\n\ts_mov_b64 exec, %0"
               : "=&s"(saved)
               : "s"(m), "v"(addr), "v"(val)
               : "memory");
}
// ballots of the sign bits of a packed pair of 16-bit values: one compare each
__device__ __forceinline__ unsigned long long fwNegLo(const uint32_t x) {
  unsigned long long m;
  asm("v_cmp_gt_i16_e64 %0, 0, %1" : "=s"(m) : "v"(x));
  return m;
}
__device__ __forceinline__ unsigned long long fwNegHi(const uint32_t x) {
  unsigned long long m;
  asm("v_cmp_gt_i32_e64 %0, 0, %1" : "=s"(m) : "v"(x));
  return m;
}

// arc strength of the pixel whose 7 x 7 block starts at LDS address e (tile row stride TS): max over the 16 arcs of 9
// contiguous ring pixels of the smallest |difference| with one sign, as in k_fast
template <int TS>
__device__ __forceinline__ int fwStrength(const uint32_t e) {
  constexpr int C = 3 * TS + 3;
  constexpr int ro[16] = {C + 3 * TS,  C + 3 * TS + 1, C + 2 * TS + 2, C + TS + 3, C + 3,  C - TS + 3, C - 2 * TS + 2, C - 3 * TS + 1,
                          C - 3 * TS,  C - 3 * TS - 1, C - 2 * TS - 2, C - TS - 3, C - 3,  C + TS - 3, C + 2 * TS - 2, C + 3 * TS - 1};
  const fw_lds_u8* const p = fwLds8(e);
  const int v = p[C];
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
  return max(smn, -smx);
}

typedef short short2v __attribute__((ext_vector_type(2)));

// Diagnostic build only (-DORBX_FAST_STAMPS): per wave, s_memtime deltas of k_fast_wave's phases, HW_ID / XCC_ID and start /
// end in s_memtime and s_memrealtime ticks, stored to a buffer nothing else reads (cdna_hip_programming.md section 7,
// in-kernel stamps); tools/fast_stamps.py prints phase shares, residency per CU and the chip-wide occupancy timeline.  The
// production build contains no stamp.
#ifdef ORBX_FAST_STAMPS
#define FW_STAMP_WAVES (1 << 18)
__device__ uint32_t g_fastStamps[FW_STAMP_WAVES * 12];
#define FW_STAMP(k)                                                                        \
  do {                                                                                     \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                          \
    stampAcc_[k] += (uint32_t)(now_ - tPrev_);                                             \
    tPrev_ = now_;                                                                         \
  } while (0)
#define FW_STAMP_INIT()                                                  \
  unsigned long long tPrev_ = __builtin_amdgcn_s_memtime();              \
  const uint32_t tStart_ = (uint32_t)tPrev_;                             \
  const uint32_t rStart_ = (uint32_t)__builtin_amdgcn_s_memrealtime();   \
  uint32_t stampAcc_[4] = {0u, 0u, 0u, 0u}
#define FW_STAMP_FLUSH()                                                                                                  \
  do {                                                                                                                    \
    const unsigned wid_ = (blockIdx.y * gridDim.x + blockIdx.x) & (FW_STAMP_WAVES - 1);                                   \
    if (lane == 0) {                                                                                                     \
      *reinterpret_cast<uint4*>(&g_fastStamps[wid_ * 12]) = make_uint4(stampAcc_[0], stampAcc_[1], stampAcc_[2], stampAcc_[3]); \
      *reinterpret_cast<uint4*>(&g_fastStamps[wid_ * 12 + 4]) =                                                          \
          make_uint4(__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)),                                        \
                     __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)), tStart_, (uint32_t)__builtin_amdgcn_s_memtime()); \
      *reinterpret_cast<uint4*>(&g_fastStamps[wid_ * 12 + 8]) = make_uint4(rStart_, (uint32_t)__builtin_amdgcn_s_memrealtime(), 0u, 0u); \
    }                                                                                                                    \
  } while (0)
#else
#define FW_STAMP(k) do {} while (0)
#define FW_STAMP_INIT() do {} while (0)
#define FW_STAMP_FLUSH() do {} while (0)
#endif

// TS = bytes per LDS tile row and per strength-map row: 48 when every cell image of the geometry is at most 12 dwords
// wide (cells of ~36 px: every frame size from VGA up), else 64.
// One wave = one workgroup = FW_CPW consecutive cells: single-wave workgroups retire on their own (a 4-wave workgroup holds
// its LDS until its slowest wave is done).
//
// Round 6 (the kernel is bound by vector-instruction issue, so only fewer instructions help; the sweep's step
// went from ~100 vector instructions with survivors / 63 without to ~60 / 40, an evaluation of 64 survivors from ~155 to ~125):
//   * a lane's position in the sweep is ONE register, quad column << 16 | LDS address: the step to the lane's next quad is an
//     add, the wrap into the next tile row an add and an unsigned min (a position whose column has not passed the row's end
//     underflows in the second add), nine vector instructions before;
//   * the quick reject's nine dword reads and the strength's seventeen byte reads are immediate offsets from one address
//     register each (a pixel is carried as the LDS address of the top-left byte of its 7 x 7 block);
//   * the verdicts are not folded into four bits per lane any more: the four pixels' ballots are four sign compares of the two
//     packed test words, and pixels beyond a row's end and the idle lanes of the last step are taken out of the BALLOTS by
//     scalar masks (the last quad of a row is one compare per step that has survivors) instead of out of the lanes' bits;
//   * survivors are appended by exec-masked stores (exec = the ballot: fwStoreB16) to a STACK of LDS addresses -- no ring index
//     to wrap, no per-lane slot select -- and evaluated 64 from the top as soon as 64 are waiting, twice per step;
//   * corners enter the strength map and the corner list the same way.
template <int TS>
__global__ __launch_bounds__(64) void k_fast_wave(const uint8_t* __restrict__ img0, long long img0FrameStride,
                                                  const uint8_t* __restrict__ pyr, const Geom g,
                                                  const FastCell* __restrict__ cells, uint32_t* __restrict__ cand,
                                                  int* __restrict__ cellCount, const int tileBytes, const int smapBytes) {
  // 16 bytes (a lane's address may point one dword in front of the tile) | tile[tileBytes] | strength map[smapBytes] |
  // survivor stack u16[FW_RING] | corner list u16[FW_CORN]
  extern __shared__ __attribute__((aligned(16))) uint8_t fwLds[];
  const int lane = threadIdx.x;
  uint8_t* const tile = fwLds + 16;
  uint8_t* const smap = tile + tileBytes;
  const uint32_t ldsBase = (uint32_t)(size_t)(fw_lds_u8*)fwLds;
  const uint32_t tileAddr = ldsBase + 16u, smapAddr = tileAddr + (uint32_t)tileBytes;
  const uint32_t ringAddr = smapAddr + (uint32_t)smapBytes, cornAddr = ringAddr + 2u * FW_RING;
  const uint32_t ringLane = ringAddr + 2u * (uint32_t)lane;
  // XCD-aware group order.  Workgroups go round-robin to the 8 XCDs (the grid's x size is a multiple of 8 * FW_XK), so the
  // workgroups with equal (blockIdx.x & 7) share an L2.  Each XCD takes every eighth RUN of FW_XK consecutive groups (16
  // cells, about one cell row of the lowest level): horizontally adjacent cells, which overlap by 6 px and share cache lines,
  // are fetched through one L2 (HBM reads 1.89 -> 0.5 MB per frame), while every XCD still gets cells of all levels -- a
  // contiguous eighth of the cell list per XCD left the XCD with the small, corner-dense upper levels working twice as long
  // as the others (measured: tools/fast_stamps.py).  The run -> XCD assignment rotates with the frame.
  const int nGroups = (g.nCellsTotal + FW_CPW - 1) / FW_CPW;
  const int xq = (int)((blockIdx.x + blockIdx.y) & 7u), xi = (int)(blockIdx.x >> 3);
  const int grp = ((xi / FW_XK) * 8 + xq) * FW_XK + (xi % FW_XK);
  if (grp >= nGroups) return;
  const int f = blockIdx.y + g.frame0;
  // (a frame's groups from the LAST one down: the cells of the small top levels hold about twice the corners of a level-0 cell, and
  // the workgroups dispatched last are the launch's tail -- with the light level-0 cells there, 0.202 - 0.203 against 0.206 - 0.207 ms
  // alone per 256 frames, round 6)
  const int cid0 = (nGroups - 1 - grp) * FW_CPW, cid1 = min(cid0 + FW_CPW, g.nCellsTotal);
  FW_STAMP_INIT();
  for (int cid = cid0; cid < cid1; cid++) {
  const FastCell c = cells[cid];  // wave-uniform: one s_load_dwordx8
  int* const myCount = cellCount + (long long)f * g.nCellsTotal + cid;
  const int ch = (int)(c.nw_ch >> 16);
  if (ch == 0) {  // every cell writes its counter (also 0), so the counters need no clearing between batches
    if (lane == 0) *myCount = 0;
    continue;
  }
  const int level = (int)(c.xoff_level >> 16);
  const LevelGeom& L = g.L[level];
  const uint8_t* const base =
      (level == 0 ? img0 + (long long)f * img0FrameStride : pyr + L.imgOff + (long long)f * L.frameStride) + c.imgOff;
  // ---- stage the cell image: LANE = TILE ROW (a cell image has at most 64 rows), the row's TS bytes as TS / 16 16-byte loads
  //      (4-byte aligned global_load_dwordx4) and as many ds_write_b128 -- 3 + 3 instructions per cell instead of 9 + 9 with
  //      lane = (row mod 5, dword) (round 5: the CU's vector-memory path charges per instruction and lane, k_describe_patch).
  //      Dwords beyond the cell's own nw hold the pixels to its right (inside the level's row or its padding: a cell image ends at
  //      least ten bytes before the row does, and thirteen rows before the level does); nothing reads them unmasked ----
  {
    typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
    constexpr int NQ = TS / 16;
    const int stride = (int)c.stride;
    if (lane < ch) {
      const uint8_t* p = base + (unsigned)(lane * stride);
      u32x4_a4 q[NQ];
#pragma unroll
      for (int j = 0; j < NQ; j++) q[j] = *reinterpret_cast<const u32x4_a4*>(p + 16 * j);
      uint4* const dst = reinterpret_cast<uint4*>(tile + lane * TS);
#pragma unroll
      for (int j = 0; j < NQ; j++) dst[j] = make_uint4(q[j].x, q[j].y, q[j].z, q[j].w);
    }
  }
  const int iw = (int)(c.iw_ih & 0xffffu), ih = (int)(c.iw_ih >> 16), xoff = (int)(c.xoff_level & 0xffffu);
  const int ox = (int)(int16_t)(c.ox_oy & 0xffffu), oy = (int)(c.ox_oy >> 16);
  {  // strength map rows 0 .. ih + 1
    uint4* const s128 = reinterpret_cast<uint4*>(smap);
    for (int i = lane; i < (ih + 2) * (TS / 16); i += 64) s128[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  __builtin_amdgcn_wave_barrier();  // (LDS operations of one wave execute in order; this only pins the compiler's order)
#ifdef ORBX_FAST_STAMPS
  __builtin_amdgcn_s_waitcnt(0);  // charge the staging phase with its loads
#endif
  FW_STAMP(0);

  constexpr int RS = TS;
  // a pixel's tile offset is o = (py + 3) * TS + xoff + px + 3; it is carried as e = tileAddr + o - (3 * TS + 3), the LDS address
  // of the top-left byte of its 7 x 7 block; its strength-map byte is at smap + (py + 1) * TS + px + 1 = LDS address e + kS, the
  // 3 x 3 block around that at e + kN + {0, 1, 2} + {0, TS, 2 TS}
  const uint32_t kN = smapAddr - tileAddr - (uint32_t)xoff, kS = kN + (uint32_t)(TS + 1);
  const uint32_t eIdle = tileAddr + (uint32_t)xoff;  // the first pixel: what idle lanes evaluate
  // The quick reject works on QUADS of four horizontally adjacent pixels per lane: the quad's rows come from LDS as dwords
  // (8 reads per 4 pixels instead of 20 byte reads) and the test runs on packed 16-bit pairs.  Item = (quad column, row);
  // idx += 64  <=>  (qc, qy) += (dqc, dqy) with one wrap at most.
  const int nqc = (iw + 3) >> 2, nItems = nqc * ih;
  const uint32_t inv = c_inv20.v[nqc];  // nqc <= 15
  const int qy0 = (int)(((uint32_t)lane * inv) >> 20), qc0 = lane - qy0 * nqc;
  const int dqy = (int)((64u * inv) >> 20), dqc = 64 - dqy * nqc;  // dqc = 64 mod nqc < nqc
  const int sh = (xoff + 3) & 3;  // byte phase of a quad's first pixel inside its dword (wave-uniform)
  // a lane's position: quad column << 16 | LDS address a of the dword in front of the quad's first dword, three tile rows up (the
  // reads below are a[1], a[2] (top), a[3 TS / 4 .. + 2] (centre row) and a[6 TS / 4 + 1, + 2] (bottom); the quad's first pixel is
  // at a + 3 TS + 4 + sh).  Next item: += dPos; a column at or beyond nqc wraps into the next row: += wrapK -- for a column that
  // has not passed the row's end that sum underflows into a huge value, so the unsigned minimum of the two keeps the right one.
  const uint32_t pos0 = ((uint32_t)qc0 << 16) | (tileAddr + (uint32_t)(qy0 * TS + xoff + 3 - sh + 4 * qc0 - 4));
  const uint32_t dPos = ((uint32_t)dqc << 16) | (uint32_t)(dqy * TS + 4 * dqc);
  const uint32_t wrapK = (uint32_t)(TS - 4 * nqc) - ((uint32_t)nqc << 16);
  const uint32_t lastThr = (uint32_t)(nqc - 1) << 16;          // positions at or above: the row's last quad
  const int nvLast = iw - 4 * (nqc - 1);                        // pixels of a row's last quad that lie inside the row (1 .. 4)
  const uint32_t aLast = tileAddr + (uint32_t)((ih - 1) * TS + xoff + 3 - sh + 4 * (nqc - 1) - 4);  // the last item's address
  // v_perm_b32 selectors (SGPRs): two tap bytes -> the halves of a u16 pair (0x0c = constant 0).  Centre / top / bottom bytes
  // sh + {0..3} of the dword pair at the quad; left compass bytes sh + 1 + {0..3} of the pair one dword to the left; right
  // compass bytes sh + 3 + {0..3} of the pair at the quad (sh <= 1) or sh - 1 + {0..3} of the pair one dword to the right
  const uint32_t selC0 = 0x0c000c00u + (uint32_t)sh * 0x00010001u + 0x00010000u, selC1 = selC0 + 0x00020002u;
  const uint32_t selL0 = selC0 + 0x00010001u, selL1 = selL0 + 0x00020002u;
  const bool rNear = sh <= 1;
  const uint32_t rOff4 = rNear ? 0u : 4u;
  const uint32_t selR0 = 0x0c000c00u + (uint32_t)(rNear ? sh + 3 : sh - 1) * 0x00010001u + 0x00010000u, selR1 = selR0 + 0x00020002u;
  uint32_t* const dstc = cand + L.candOff + (long long)f * L.candCap + c.segOff;
  const int segCap = (int)c.segCap;
  int nOut = 0;
  for (int pass = 0; pass < 2; pass++) {
    const int th = pass == 0 ? g.iniTh : g.minTh;
    const uint32_t th2 = (uint32_t)th * 0x00010001u;
    int nList = 0, nCorn = 0;  // wave-uniform (SGPRs)
    // A cell without a survivor at iniThFAST is swept again at minThFAST (cpp:1117-1123) -- on real images that is every second
    // cell, and in a flat region the second sweep lists nothing either.  While the first sweep has not listed a pixel yet (`flat`,
    // wave-uniform), it also keeps the minimum of its two test values e = (v + th) - max side and (min side + th) - v: a pixel
    // passes the quick reject at minThFAST iff e < iniTh - minTh, so a cell whose sweep ends `flat` with every minimum at or above
    // that difference could not list a single pixel at minThFAST and is not swept again.  Pixels beyond a row's end and idle lanes
    // are not masked out of the minimum: a false "some pixel passes" only costs the second sweep, which then decides as before.
    bool flat = pass == 0 && ORBX_FAST_FLATSKIP;
    short2v looseMin = {0x7fff, 0x7fff};
    // evaluates the `cnt` newest entries of the survivor stack (64, or what is left at the end): exact strength; corners
    // (s > th) enter the strength map and the corner list
    auto flush = [&](const int cnt) {
      __builtin_amdgcn_wave_barrier();
      const int keepN = nList - cnt;
      uint32_t e = (uint32_t)fwLds16(ringLane)[keepN];
      unsigned long long act = ~0ull;
      if (cnt < 64) {  // (uniform)
        asm volatile("" ::: "memory");
        act = (1ull << cnt) - 1ull;
        e = lane < cnt ? e : eIdle;
      }
      const int sv = fwStrength<TS>(e);
      const unsigned long long mc = __ballot(sv > th) & act;
      if (mc != 0ull) {
        fwStoreB8(e + kS, (uint32_t)sv, mc);
        const int nc = (int)__popcll(mc);
        if (nCorn + nc <= FW_CORN) fwStoreB16((cornAddr + 2u * (uint32_t)nCorn) + 2u * (uint32_t)fwMbcnt(mc), e, mc);
        nCorn += nc;
      }
      nList = keepN;
    };
    // ---- quick reject on the 4 compass pixels (an arc of 9 holds two adjacent ones), survivors onto the stack ----
    {
      uint32_t pos = pos0;
      for (int idx0 = 0; idx0 < nItems; idx0 += 64) {
        uint32_t a = pos & 0xffffu;
        const bool tail = idx0 + 64 > nItems;  // (uniform) the step with idle lanes: they read the last item again, masked below
        if (tail) {
          asm volatile("" ::: "memory");
          a = min(a, aLast);
        }
        const fw_lds_u32* const pa = fwLds32(a);
        const uint32_t t0 = pa[1], t1 = pa[2];
        const uint32_t cM = pa[3 * (TS / 4)], c0 = pa[3 * (TS / 4) + 1], c1 = pa[3 * (TS / 4) + 2];
        const uint32_t b0 = pa[6 * (TS / 4) + 1], b1 = pa[6 * (TS / 4) + 2];
        // the dword pair that holds the right compass pixels: read again at a wave-uniform offset (an LDS read instead of two
        // v_cndmask on the vector port, which is the port this kernel is bound by)
        const fw_lds_u32* const pr = fwLds32(a + rOff4);
        const uint32_t rlo = pr[3 * (TS / 4) + 1], rhi = pr[3 * (TS / 4) + 2];
        uint32_t fl[2];
        short2v em[2];
#pragma unroll
        for (int hp = 0; hp < 2; hp++) {  // pixels (0, 1) and (2, 3) of the quad as u16 pairs
          const uint32_t V = __builtin_amdgcn_perm(c1, c0, hp ? selC1 : selC0);
          const uint32_t QL = __builtin_amdgcn_perm(c0, cM, hp ? selL1 : selL0), QR = __builtin_amdgcn_perm(rhi, rlo, hp ? selR1 : selR0);
          const uint32_t QT = __builtin_amdgcn_perm(t1, t0, hp ? selC1 : selC0), QB = __builtin_amdgcn_perm(b1, b0, hp ? selC1 : selC0);
          const ushort2v vV = __builtin_bit_cast(ushort2v, V), vL = __builtin_bit_cast(ushort2v, QL), vR = __builtin_bit_cast(ushort2v, QR),
                         vT = __builtin_bit_cast(ushort2v, QT), vB = __builtin_bit_cast(ushort2v, QB);
          // corner needs (max(T, B) > hi and max(L, R) > hi) or (min(T, B) < lo and min(L, R) < lo)
          const ushort2v mx = __builtin_elementwise_min(__builtin_elementwise_max(vT, vB), __builtin_elementwise_max(vL, vR));
          const ushort2v mn = __builtin_elementwise_max(__builtin_elementwise_min(vT, vB), __builtin_elementwise_min(vL, vR));
          const ushort2v hi = vV + __builtin_bit_cast(ushort2v, th2);
          const short2v e1 = __builtin_bit_cast(short2v, hi) - __builtin_bit_cast(short2v, mx);                      // < 0: mx > hi
          const short2v e2 = __builtin_bit_cast(short2v, mn) + __builtin_bit_cast(short2v, th2) - __builtin_bit_cast(short2v, vV);  // < 0: mn < lo
          fl[hp] = __builtin_bit_cast(uint32_t, e1) | __builtin_bit_cast(uint32_t, e2);  // sign bits: the two pixels' verdicts
          em[hp] = __builtin_elementwise_min(e1, e2);
        }
        if (flat) {  // (uniform; the empty asm keeps it a branch -- as a select it would cost every step of every cell five instructions)
          asm volatile("" ::: "memory");
          looseMin = __builtin_elementwise_min(looseMin, __builtin_elementwise_min(em[0], em[1]));
        }
        if (__ballot(((fl[0] | fl[1]) & 0x80008000u) != 0u) != 0ull) {  // (wave-uniform) a step without any survivor appends nothing
          flat = false;
          // what the ballots must not count: pixels beyond the row's end (the last quad of a row holds nvLast pixels of it) and the
          // idle lanes of the last step
          unsigned long long keepEnd = ~0ull;
          if (nvLast < 4) keepEnd = ~__ballot(pos >= lastThr);
          const unsigned long long live = tail ? (1ull << (nItems - idx0)) - 1ull : ~0ull;
          const uint32_t q0 = a + (uint32_t)(sh + 1);  // the quad's first pixel as a stack entry
#pragma unroll
          for (int j = 0; j < 4; j++) {
            unsigned long long m = ((j & 1) ? fwNegHi(fl[j >> 1]) : fwNegLo(fl[j >> 1])) & live;
            if (j >= nvLast) m &= keepEnd;
            if (m != 0ull) {  // (wave-uniform)
              fwStoreB16((ringAddr + 2u * (uint32_t)nList) + 2u * (uint32_t)fwMbcnt(m), q0 + (uint32_t)j, m);
              nList += (int)__popcll(m);
            }
            if (j & 1)
              while (nList >= 64) flush(64);
          }
        }
        pos += dPos;
        pos = min(pos, pos + wrapK);
      }
      FW_STAMP(1);
      if (nList > 0) flush(nList);
    }
    __builtin_amdgcn_wave_barrier();
    FW_STAMP(2);
    // ---- in-cell NMS on the strength map; survivors are the cell's keypoints ----
    auto nms = [&](const uint32_t eIn, const bool act) {
      const uint32_t e = act ? eIn : eIdle;
      const fw_lds_u8* const q = fwLds8(e + kN);  // top-left of the 3 x 3 block in the strength map
      // all nine reads are issued together (short-circuit tests would chain nine LDS round trips)
      const int sv = q[RS + 1];
      const int n0 = q[0], n1 = q[1], n2 = q[2], n3 = q[RS], n4 = q[RS + 2], n5 = q[2 * RS], n6 = q[2 * RS + 1], n7 = q[2 * RS + 2];
      const int nmax = max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
      const bool keep = act && sv > 1 && sv > nmax;
      const unsigned long long mk = __ballot(keep);
      if (keep) {
        const int slot = nOut + fwMbcnt(mk);
        // tile row = y - oy, tile column = x - ox; o / TS by multiply-shift (exact for o < 2^12)
        const int o = (int)(e - tileAddr) + 3 * TS + 3;
        const int row = TS == 64 ? (o >> 6) : (int)(((uint32_t)o * 43691u) >> 21);
        static_assert(TS == 64 || TS == 48, "row split of a tile offset");
        if (slot < segCap) dstc[slot] = packCand(o - row * TS + ox, row + oy, sv - 1);
      }
      nOut += (int)__popcll(mk);
    };
    if (nCorn <= FW_CORN) {
      for (int e0 = 0; e0 < nCorn; e0 += 64) nms((uint32_t)fwLds16(cornAddr)[min(e0 + lane, FW_CORN - 1)], e0 + lane < nCorn);
    } else {  // more corners than the list holds (noise at a low threshold): scan the strength map instead
      const uint32_t invw = c_inv20.v[iw];
      const int py0 = (int)(((uint32_t)lane * invw) >> 20), px0 = lane - py0 * iw;
      const int dpy = (int)((64u * invw) >> 20), dpx = 64 - dpy * iw;
      int px = px0;
      uint32_t e = eIdle + (uint32_t)(py0 * TS + px0);
      const int npix = iw * ih;
      for (int idx0 = 0; idx0 < npix; idx0 += 64) {
        const bool live = idx0 + lane < npix;
        const int sv = live ? (int)fwLds8(e + kS)[0] : 0;
        const unsigned long long any = __ballot(sv > 0);
        if (any) nms(e, sv > 0);
        px += dpx;
        e += (uint32_t)(dpy * TS + dpx);
        if (px >= iw) { px -= iw; e += (uint32_t)(TS - iw); }
      }
    }
    FW_STAMP(3);
    // cpp:1109-1123: the cell is retried at minThFAST only if it yielded nothing at iniThFAST
    if (nOut > 0 || pass == 1 || g.minTh >= g.iniTh) break;
    {  // no pixel was listed and none passes the quick reject at minThFAST either
      const short2v dth = {(short)(g.iniTh - g.minTh), (short)(g.iniTh - g.minTh)};
      if (flat && __ballot((__builtin_bit_cast(uint32_t, looseMin - dth) & 0x80008000u) != 0u) == 0ull) break;
    }
  }
  if (lane == 0) *myCount = min(nOut, segCap);  // nOut <= segCap: NMS survivors are never 8-neighbours
  __builtin_amdgcn_wave_barrier();  // the next cell's staging stores come after this cell's last LDS reads
  }  // cells of this wave
  FW_STAMP_FLUSH();
}
#ifdef ORBX_FAST_STAMPS
extern "C" int orbx_diag_fast_stamps(uint32_t* out, int nWaves) {  // out: nWaves x 12 dwords; nWaves < 0: clear the buffer
  if (nWaves < 0) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_fastStamps)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(uint32_t) * 12 * FW_STAMP_WAVES);
  }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fastStamps), sizeof(uint32_t) * 12 * (size_t)nWaves);
}
#endif

// =================================================================================================
// K4+K5+K6  orientation + patch-local Gaussian + steered BRIEF per keypoint.
// =================================================================================================
constexpr int8_t k_pattern[256 * 4] = {
#include "orbx_pattern_data.inc"
};
// the same 256 point pairs as f32 (x0, y0, x1, y1): the steered-BRIEF loop needs them as floats (cpp:184-188)
struct PatternF {
  float v[256 * 4];
};
constexpr PatternF makePatternF() {
  PatternF t{};
  for (int i = 0; i < 256 * 4; i++) t.v[i] = (float)k_pattern[i];
  return t;
}
__device__ const PatternF d_patternf = makePatternF();
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};  // cpp:562-594

__device__ __forceinline__ int reflect101(int p, int n) {
  if (p < 0) p = -p;
  if (p >= n) p = 2 * n - 2 - p;
  return p;
}

// cos / sin of x in [0, 2 pi + eps] (a keypoint angle in radians) in f64, ~1 ulp: quadrant reduction with a two-term pi/2
// and the fdlibm kernel polynomials, fused multiply-adds written out.  The general-range sincos of the device library costs
// ~110 f64 instructions per call (it also carries the huge-argument path); this one ~35.  What the descriptor needs is the
// value rounded to f32, cpp:173-174; tests/test_gpu_parity.py::test_sincos_matches_libm sweeps 3 M angles against libm.
__device__ __forceinline__ void sincosSmall(double x, double* sn, double* cs) {
  const double k = rint(x * 6.36619772367581382433e-01);                   // nearest multiple of pi/2: 0 .. 4
  double r = fma(-k, 1.57079632679489655800e+00, x);                       // pi/2 high part: exact product for k <= 4
  r = fma(-k, 6.12323399573676603587e-17, r);                              // pi/2 low part
  const double z = r * r;
  // __kernel_sin / __kernel_cos (fdlibm), |r| <= pi/4
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2), S1);
  const double sr = fma(z * r, ps, r);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  const double cr = w + (((1.0 - w) - hz) + fma(z * z, pc, 0.0));
  const int q = (int)k & 3;
  const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
  *sn = (q & 2) ? -s0 : s0;
  *cs = ((q + 1) & 2) ? -c0 : c0;
}

// The same values from a 256-entry table (round 4): x = k pi / 128 + r with |r| <= pi / 256, sin / cos of k pi / 128 as correctly
// rounded doubles (orbx_sincos_tab.inc), three-term Taylor polynomials for sin r and cos r - 1 (next terms r^9 / 9! and
// r^8 / 8!: below 2^-80) and the addition theorems -- ~17 f64 instructions instead of ~33 (no 6-term kernels, no quadrant
// selects).  In k_describe_patch the angle is the same in all 64 lanes (one keypoint per wave), so the table entry comes through
// one scalar load (UNIFORM).  What matters is the value ROUNDED TO F32: tools/sincos_exhaustive.py compares it with the
// oracle's (float)cos((double)a) / (float)sin((double)a) for EVERY f32 angle in [0, 360] (1.13e9 values).
#include "orbx_sincos_tab.inc"
template <bool UNIFORM>
__device__ __forceinline__ void sincosTable(double x, double* sn, double* cs) {
  const double k = rint(x * ORBX_128OPI);                  // nearest multiple of pi / 128: 0 .. 256
  double r = fma(-k, ORBX_PIO128_HI, x);                   // exact (cancellation: the difference has at most 52 significant bits)
  r = fma(-k, ORBX_PIO128_LO, r);
  int ki = (int)k & 255;
  if (UNIFORM) ki = __builtin_amdgcn_readfirstlane(ki);
  const double sk = d_sincosTab[ki][0], ck = d_sincosTab[ki][1];
  const double z = r * r;
  const double ps = fma(z, fma(z, -1.0 / 5040.0, 1.0 / 120.0), -1.0 / 6.0);   // (sin r - r) / r^3
  const double pc = fma(z, fma(z, -1.0 / 720.0, 1.0 / 24.0), -0.5);           // (cos r - 1) / r^2
  const double sr = fma(z * r, ps, r), cm1 = z * pc;
  *sn = fma(ck, sr, fma(sk, cm1, sk));
  *cs = fma(-sk, sr, fma(ck, cm1, ck));
}
#ifdef ORBX_SINCOS_POLY  // (build flag: the round-2 / 3 evaluation, for comparison)
#define ORBX_SINCOS(x, sn, cs, uniform) sincosSmall(x, sn, cs)
#else
#define ORBX_SINCOS(x, sn, cs, uniform) sincosTable<uniform>(x, sn, cs)
#endif

// ORBX_LIBM_FLOAT (orbx_set_libm_variant): cos(angle) / sin(angle) of cpp:174 read as cosf / sinf -- glibc >= 2.28's f64-polynomial
// algorithm (sysdeps/ieee754/flt-32/s_sincosf.h, s_sinf.c, s_cosf.c; constants of s_sincosf_data.c), which is not correctly
// rounded, repeated operation for operation in the FMA form of the x86-64 multiarch build.  Valid for 0 <= y < 120 (a keypoint
// angle in radians is below 6.2832).  The oracle holds the same restatement and sweeps every f32 angle of [0, 360] against the
// host's libm (tests/test_oracle.py); tests/test_gpu_parity.py::test_sincos_matches_libm compares this one with the oracle.
// Branch-free form (the keypoint's angle is one value per wave, but a branch costs the wave its scalar hops either way): every
// call evaluates the sine polynomial on x * sign and the cosine polynomial on x^2 once and hands them out by the quadrant's parity.
//   * glibc's first branch (|y| < pi / 4 by its top-12-bit test, i.e. y < 0.75) is the reduction with n = 0: x * (2 / pi) < 0.48
//     rounds to quadrant 0, fma(-0, hpi, x) == x, sign[0] == 1, table 0 -- the same operations on the same values;
//   * its shortcut for |y| < 2^-12 (sinf returns y, cosf 1.0f) is what the polynomials round to there: x^3 / 6 is below 2^-26.6 x
//     and x^2 / 2 below 2^-25, less than half an ulp of the f32 results;
//   * table 1 holds the negated cosine coefficients, so its cosine polynomial is the exact negative of table 0's.
// tools/sincos_exhaustive.py 1 compares it with the oracle's literal restatement for EVERY f32 angle in [0, 360].
__device__ __forceinline__ void sincosfGlibc(float y, float* sn, float* cs) {
  const double x = (double)y;
  // reduce_fast (!TOINT_INTRINSICS): hpi_inv is 2 / pi * 2^24, the quadrant ends up in bits 24..31
  const double r = x * 0x1.45F306DC9C883p+23;
  const int n = ((int)r + 0x800000) >> 24;
  const double xr = fma(-(double)n, 0x1.921FB54442D18p0, x);
  const double sg = ((n ^ (n >> 1)) & 1) ? -1.0 : 1.0;  // sign[n & 3] = {1, -1, -1, 1}
  const double xs = xr * sg, x2 = xr * xr;
  // sinf_poly, even n: the sine polynomial (the same coefficients in both tables)
  const double x3 = xs * x2, s1 = fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7), x7 = x3 * x2;
  const double S = fma(x7, s1, fma(x3, -0x1.555545995a603p-3, xs));
  // sinf_poly, odd n: the cosine polynomial of table 0; table 1 (n & 2) negates it
  const double x4 = x2 * x2, c2 = fma(x2, 0x1.99343027bf8c3p-16, -0x1.6c087e89a359dp-10);
  const double c1 = fma(x2, -0x1.ffffffd0c621cp-2, 0x1p0), x6 = x4 * x2;
  const double C0 = fma(x6, c2, fma(x4, 0x1.55553e1068f19p-5, c1));
  const float Sf = (float)S, C0f = (float)C0;
  const float Cf = (n & 2) ? -C0f : C0f;
  *sn = (n & 1) ? Cf : Sf;   // sinf: sinf_poly(x * s, x * x, p, n)
  *cs = (n & 1) ? Sf : Cf;   // cosf: sinf_poly(x * s, x * x, p, n ^ 1)
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


// -------------------------------------------------------------------------------------------------
// K4+K5+K6 fused, patch-local: one wave per keypoint.  The 43x43 raw window (REFLECT_101 at the level's edges) is
// staged in LDS with dword loads; the 7x7 Gaussian is evaluated only on the 37x37 neighbourhood the 512 rotated
// sample points can reach, with the same exact fixed-point arithmetic as the whole-level blur (v_dot4_u32_u8 for the
// horizontal pass, v_dot2_u32_u16 on row pairs for the vertical pass), so no blurred pyramid is written or re-read.
// -------------------------------------------------------------------------------------------------
#define PW_ROWS 43
#define PW_WORDS 13    // 52 bytes per staged row: window columns kx-21 .. kx+21 start at byte (kx-21)&3
#define PW_PAIRS 22    // row pairs of horizontal sums (rows 0..43, the last one is a dummy)
#define PW_COLS 40     // 37 blurred columns padded to 10 groups of 4
#define BL_ROWS_PAD 38
#define PW_RAW_WORDS 560  // PW_ROWS * PW_WORDS = 559, padded so that the row-pair sums behind it are 16-byte aligned
#define PW_WAVE_WORDS (PW_RAW_WORDS + PW_PAIRS * PW_COLS + 4)  // the blurred bytes reuse the raw window's space; + 2 moment sums

// IC_Angle disc as dot4 weights: row |v| of the 31x31 disc covers u = -umax[|v|] .. umax[|v|]; the row's 32 bytes
// u = -15 .. 16 are 8 dwords j, and for each the weights are w1 = (1 per valid byte) and wu = (u + 15 per valid byte), so
// sum I = sum_j dot4(B_j, w1_j),  sum u*I = sum_j dot4(B_j, wu_j) - 15 * sum I   (exact integer identities).
struct IcTables {
  uint32_t w1[16 * 8], wu[16 * 8];
};
constexpr IcTables makeIcTables() {
  IcTables t{};
  constexpr int um[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};  // cpp:562-594 (== c_umax)
  for (int av = 0; av < 16; av++)
    for (int j = 0; j < 8; j++) {
      uint32_t m1 = 0, mu = 0;
      for (int b = 0; b < 4; b++) {
        const int u = -15 + 4 * j + b, au = u < 0 ? -u : u;
        if (au <= um[av]) {
          m1 |= 1u << (8 * b);
          mu |= (uint32_t)(u + 15) << (8 * b);
        }
      }
      t.w1[av * 8 + j] = m1;
      t.wu[av * 8 + j] = mu;
    }
  return t;
}
__device__ const IcTables d_ic = makeIcTables();

#ifndef ORBX_DESC_EXP
#define ORBX_DESC_EXP 0      // 3 = diagnostic build without the window fetch (timing only: the kernel's issue bound; tools/exp_desc_fetch.sh)
#endif
#ifndef ORBX_DESC_LOADMAP
#define ORBX_DESC_LOADMAP 0  // window staging: 0 = lane is a window row (three 16-byte loads per lane), 1 = lane is a (row, 16-byte piece) slot
#endif
#ifndef DESC_WAVES
#define DESC_WAVES 3   // keypoints (= waves) per workgroup: consecutive keypoints of a frame's list are spatially close, so
                       // putting them on one CU lets their overlapping windows hit in that CU's L1 (1: 0.44 ms, 2: 0.38,
                       // 3: 0.365, 4: 0.39, 8: 0.44, 16: 0.63 per 256 frames; 3 slices of 5.8 KB keep 27 waves per CU)
#endif
// GV = Gaussian Q8 tap set (orbx_set_opencv_variant): 0 = [18,34,48,56,48,34,18] (error diffusion, sum 256: OpenCV >= 4.1.1 /
// 3.4.7), 1 = [18,34,49,55,49,34,18] (every tap rounded, sum 257: the bit-exact path of 3.4.1 .. 4.1.0 and the integer filter
// before it; a sum of 2^24 or more saturates to 255)
// Diagnostic build only (-DORBX_DESC_STAMPS): per wave, s_memtime at the start, when the window loads have landed, after
// IC_Angle, after the horizontal / vertical blur passes and at the end; tools/desc_stamps.py prints the shares.
#ifdef ORBX_DESC_STAMPS
#define DS_STAMP_WAVES (1 << 18)
__device__ uint32_t g_descStamps[DS_STAMP_WAVES * 8];
#define DS_STAMP(k)                                                                                     \
  do {                                                                                                  \
    if (lane == 0 && dsWave_ < DS_STAMP_WAVES) g_descStamps[dsWave_ * 8 + (k)] = (uint32_t)__builtin_amdgcn_s_memtime(); \
  } while (0)
extern "C" int orbx_diag_desc_stamps(uint32_t* out, int nWaves) {  // out: nWaves x 8 dwords; nWaves < 0: clear the buffer
  if (nWaves < 0) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_descStamps)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(uint32_t) * 8 * DS_STAMP_WAVES);
  }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_descStamps), sizeof(uint32_t) * 8 * (size_t)nWaves);
}
#else
#define DS_STAMP(k) do { } while (0)
#endif

// The 512 rotated sample points lie in the disc r^2 + c^2 <= 365 around the keypoint ((13, 13) is the farthest pattern point:
// 18.38, and rounding moves a point by at most 0.71), not in the whole 37 x 37 square.  A (row pair, 4-column) item of the
// horizontal pass is needed only if one of the four vertical items it feeds holds a point of the disc: 190 of 220, i.e. three
// steps of 64 lanes instead of four of 60.  Entry = row pair << 4 | column group, 0xffff = idle.
struct DescHItems {
  uint16_t v[192];
};
constexpr DescHItems makeDescHItems() {
  DescHItems t{};
  bool vneed[19 + 3][10] = {};
  for (int q = 0; q < 19; q++)
    for (int g = 0; g < 10; g++)
      for (int r = 2 * q; r < 2 * q + 2; r++)
        for (int c = 4 * g; c < 4 * g + 4; c++)
          if (r <= 36 && c <= 36 && (r - 18) * (r - 18) + (c - 18) * (c - 18) <= 365) vneed[q][g] = true;
  int n = 0;
  for (int rp = 0; rp < 22; rp++)
    for (int g = 0; g < 10; g++) {
      bool need = false;
      for (int q = rp - 3; q <= rp; q++)
        if (q >= 0 && q < 19 && vneed[q][g]) need = true;
      if (need && n < 192) t.v[n++] = (uint16_t)(rp << 4 | g);
    }
  for (; n < 192; n++) t.v[n] = 0xffff;
  return t;
}
__device__ const DescHItems d_descHItems = makeDescHItems();
constexpr int descHItemCount() {
  const DescHItems t = makeDescHItems();
  int n = 0;
  for (int i = 0; i < 192; i++) n += t.v[i] != 0xffff;
  return n;
}
static_assert(descHItemCount() == 190, "horizontal items of the sampling disc (all of them must fit the three steps)");

// horizontal Gaussian taps as v_dot4 weights: the 7 taps applied to window bytes t .. t + 6 (t = byte shift + column, 0 .. 6),
// restricted to dword m of the 16 staged bytes
template <int GV>
__host__ __device__ constexpr uint32_t descHTap(int t, int m) {
  const uint32_t T[7] = {18u, 34u, GV ? 49u : 48u, GV ? 55u : 56u, GV ? 49u : 48u, 34u, 18u};
  uint32_t k = 0;
  for (int i = 0; i < 4; i++) {
    const int j = 4 * m + i - t;
    if (j >= 0 && j < 7) k |= T[j] << (8 * i);
  }
  return k;
}
template <int GV, bool STAGED = false>
__global__ __launch_bounds__(64 * DESC_WAVES) void k_describe_patch(const uint8_t* __restrict__ img0, long long img0FrameStride,
                                                       int img0Aligned, const uint8_t* __restrict__ pyr, const Geom g,
                                                       const SelKp* __restrict__ sel, const int* __restrict__ nsel,
                                                       orbx_keypoint* __restrict__ kps, uint8_t* __restrict__ desc,
                                                       int capacity, const DescStage ds, const int libmFloat) {
  __shared__ __attribute__((aligned(16))) uint32_t ldsAll[DESC_WAVES][PW_WAVE_WORDS];
  static_assert(PW_WAVE_WORDS % 4 == 0, "every wave's LDS slice must stay 16-byte aligned");
  const int f = blockIdx.y + g.frame0, lane = threadIdx.x & 63;
  uint32_t* const lds = ldsAll[threadIdx.x >> 6];
  // XCD-aware order: workgroups go round-robin to the 8 XCDs (the grid's x size is a multiple of 8); workgroup b takes
  // keypoint group (b % 8) * chunk + b / 8, so that one XCD works on a contiguous eighth of the frame's keypoint list
  // (mostly one pyramid level) and its L2 holds that part of the pyramid only.  Every keypoint costs the same, so the
  // XCDs stay balanced.
  const int grp = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  // (the wave's index as a scalar: the keypoint record then comes through a scalar load, beside the count's)
  const int i = grp * DESC_WAVES + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // (the keypoint record is fetched together with the frame's count, not behind it: count, record and window were three
  // dependent global loads and, by tools/desc_stamps.py, 63 % of a wave's lifetime)
  SelKp k;
  if constexpr (STAGED) {
    // keypoint i of the level-major order = entry i - (keypoints of the levels below) of its level's staging list; the counts of
    // the frame's levels come through scalar loads (a unit the selection redid carries a tag bit, a failed one a negative count)
    int off = 0, total = 0, kl = 0;
    bool bad = false;
    for (int l = 0; l < g.nlevels; l++) {
      const int c = selUnitCount(ds.nselLevel[f * g.nlevels + l], &bad);
      if (i >= total) { off = total; kl = l; }
      total += c;
    }
    if (i == 0) {  // the frame's designated wave: k_sel_compact's bookkeeping (orbx_device.h: one definition for both)
      if (lane == 0) selPublishFrame(f, total, bad, ds.nsel, ds.nselUser, ds.hostNsel, ds.hostErr);
      if (blockIdx.y == 0 && ds.maxN) selReduceReports(lane, g.frame0, (int)gridDim.y, g.nlevels, ds.maxN, ds.hostMaxN);
    }
    if (i >= total) return;  // wave-uniform
    k = ds.selStage[(long long)f * ds.selStride + ds.selOff[kl] + (i - off)];
  } else {
    // (both loads are issued before either is waited for: the compiler sinks the record's load behind the count's branch otherwise,
    // and the wave's chain of dependent loads is what its lifetime is made of -- tools/desc_stamps.py)
    const unsigned long long kraw = *reinterpret_cast<const unsigned long long*>(&sel[(long long)f * g.selCap + min(i, g.selCap - 1)]);
    const int cnt = nsel[f];
    unsigned long long kuse = kraw;
    asm volatile("" : "+s"(kuse));  // (the record is a value here, not a load to be moved)
    if (i >= cnt) return;  // wave-uniform; the waves of a workgroup never synchronise with each other
    k.x = (uint16_t)(kuse & 0xffff); k.y = (uint16_t)((kuse >> 16) & 0xffff); k.level = (uint8_t)((kuse >> 32) & 0xff);
    k.response = (uint8_t)((kuse >> 40) & 0xff); k.pad = 0;
  }
#ifdef ORBX_DESC_STAMPS
  const unsigned dsWave_ = (unsigned)((blockIdx.y * gridDim.x + blockIdx.x) * DESC_WAVES + (threadIdx.x >> 6));
#endif
  DS_STAMP(0);
  // Loads that do not depend on the keypoint are issued first, so that their latency runs under the window fetch: the
  // disc-row weights of IC_Angle (lane = disc row) and this lane's four point pairs of the BRIEF pattern.
  const int icRow = min(lane, 30), icAv = icRow < 15 ? 15 - icRow : icRow - 15;
  const uint4 w1a = reinterpret_cast<const uint4*>(d_ic.w1 + icAv * 8)[0], w1b = reinterpret_cast<const uint4*>(d_ic.w1 + icAv * 8)[1];
  const uint4 wua = reinterpret_cast<const uint4*>(d_ic.wu + icAv * 8)[0], wub = reinterpret_cast<const uint4*>(d_ic.wu + icAv * 8)[1];
  float4 pat[4];
#pragma unroll
  for (int wq = 0; wq < 4; wq++) pat[wq] = reinterpret_cast<const float4*>(d_patternf.v)[wq * 64 + lane];
  uint32_t hitem[3];  // this lane's items of the horizontal pass
#pragma unroll
  for (int it = 0; it < 3; it++) hitem[it] = d_descHItems.v[it * 64 + lane];
  uint32_t* raw = lds;                               // [43][13] dwords
  uint32_t* hz2 = raw + PW_RAW_WORDS;                // [22][40] row-pair packed horizontal sums (16-byte aligned rows)
  uint32_t* bl32 = raw;                              // [38][10] dwords = blurred bytes, row stride 40 (raw is dead by then)
  int* msum = reinterpret_cast<int*>(hz2 + PW_PAIRS * PW_COLS);  // [2] moment sums of IC_Angle
  static_assert(BL_ROWS_PAD * (PW_COLS / 4) <= PW_ROWS * PW_WORDS, "blurred bytes must fit in the raw window");
  static_assert(PW_ROWS * PW_WORDS <= PW_RAW_WORDS && PW_RAW_WORDS % 4 == 0, "raw window padding");
  // the keypoint is wave-uniform: keep its fields in SGPRs so that the level geometry comes through scalar loads
  const int level = __builtin_amdgcn_readfirstlane((int)k.level);
  const int kx = __builtin_amdgcn_readfirstlane((int)k.x), ky = __builtin_amdgcn_readfirstlane((int)k.y);
  const LevelGeom& L = g.L[level];
  const uint8_t* img = level == 0 ? img0 + (long long)f * img0FrameStride : pyr + L.imgOff + (long long)f * L.frameStride;
  const bool aligned = level > 0 || img0Aligned != 0;
  const int w = L.w, h = L.h, stride = L.stride;
  const int ax = (kx - 21) & ~3;          // may be -4
  const int s = (kx - 21) - ax;           // byte offset of window column 0 inside a staged row, 0..3
  // ---- stage the raw window.  Round 5: LANE = WINDOW ROW, the row as three 16-byte loads (4-byte aligned global_load_dwordx4):
  //      3 vector-memory instructions per keypoint instead of 11 single-dword ones over (4 rows x 13 dwords) -- the CU's address /
  //      return path charges per instruction and lane, and the window fetch was what kept the kernel off its issue bound (alone
  //      per 256 frames: 0.334 ms; without any fetch 0.231 = the issue bound; this form 0.265; docs/history.md, round 5).
  //      Columns kx-21 .. kx+21 end at byte s + 42 <= 45 of the staged row: 48 bytes hold them; dword 12 of a row only ever meets
  //      zero taps (blurred columns 37..39, the padding of the last group of four, are never sampled) and is left as it is. ----
  {
    if (lane == 0) { msum[0] = 0; msum[1] = 0; }
    const bool rowsInside = aligned && ax >= 0 && ax + 48 <= w;  // (uniform) no staged dword crosses the level's left / right side
#if ORBX_DESC_EXP == 3  // TIMING ONLY: no window fetch at all (the kernel's issue bound)
    if (false)
#endif
    if (rowsInside) {
      typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
#if ORBX_DESC_LOADMAP == 0
      if (lane < PW_ROWS) {
        int yy = ky - 21 + lane;
        yy = yy < 0 ? -yy : yy; yy = yy >= h ? 2 * h - 2 - yy : yy;  // REFLECT_101 (a keypoint is at least 19 px from the border)
        // (uniform base + 32-bit lane offset: global_load with an SGPR address)
        const uint8_t* p = img + (uint32_t)(yy * stride + ax);
        const u32x4_a4 q0 = *reinterpret_cast<const u32x4_a4*>(p), q1 = *reinterpret_cast<const u32x4_a4*>(p + 16),
                       q2 = *reinterpret_cast<const u32x4_a4*>(p + 32);
        uint32_t* dst = raw + lane * PW_WORDS;
        dst[0] = q0.x; dst[1] = q0.y; dst[2] = q0.z; dst[3] = q0.w; dst[4] = q1.x; dst[5] = q1.y; dst[6] = q1.z; dst[7] = q1.w;
        dst[8] = q2.x; dst[9] = q2.y; dst[10] = q2.z; dst[11] = q2.w;
      }
#else  // (experiment: slot = (row, 16-byte piece), 129 slots over the lanes of three instructions)
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const int idx = t * 64 + lane, r = idx / 3, c = idx - 3 * r;
        if (idx < 3 * PW_ROWS) {
          int yy = ky - 21 + r;
          yy = yy < 0 ? -yy : yy; yy = yy >= h ? 2 * h - 2 - yy : yy;
          const u32x4_a4 q = *reinterpret_cast<const u32x4_a4*>(img + (uint32_t)(yy * stride + ax + 16 * c));
          uint32_t* dst = raw + r * PW_WORDS + 4 * c;
          dst[0] = q.x; dst[1] = q.y; dst[2] = q.z; dst[3] = q.w;
        }
      }
#endif
    } else {
      // the window crosses the level's left / right side (or level 0 is not dword-aligned): lanes 0..51 = 4 rows x 13 dwords per
      // step; dwords inside the level by dword loads, the others byte by byte with REFLECT_101
      const int rsub = lane / PW_WORDS, d = lane - rsub * PW_WORDS;
      const int xs = ax + 4 * d;
      const bool active = lane < 4 * PW_WORDS;
      const bool fastx = aligned && xs >= 0 && xs + 4 <= w;
#pragma unroll
      for (int it = 0; it < 11; it++) {
        const int r = it * 4 + rsub;
        if (active && fastx && r < PW_ROWS) {
          int yy = ky - 21 + r;
          yy = yy < 0 ? -yy : yy; yy = yy >= h ? 2 * h - 2 - yy : yy;
          raw[it * (4 * PW_WORDS) + lane] = *reinterpret_cast<const uint32_t*>(img + (yy * stride + xs));
        }
      }
      if (active && !fastx) {
        for (int r = rsub; r < PW_ROWS; r += 4) {
          int yy = ky - 21 + r;
          yy = yy < 0 ? -yy : yy; yy = yy >= h ? 2 * h - 2 - yy : yy;
          const uint8_t* row = img + (long long)yy * stride;
          uint32_t word = 0;
#pragma unroll
          for (int b = 0; b < 4; b++) {
            int xx = xs + b;
            xx = xx < 0 ? -xx : xx; xx = xx >= w ? 2 * w - 2 - xx : xx; xx = min(max(xx, 0), w - 1);
            word |= (uint32_t)row[xx] << (8 * b);
          }
          raw[r * PW_WORDS + d] = word;
        }
      }
    }
  }
#ifdef ORBX_DESC_STAMPS
  __builtin_amdgcn_s_waitcnt(0);  // charge the staging phase with its loads
#endif
  DS_STAMP(1);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  // ---- IC_Angle (cpp:103-159) on the un-blurred window: pixel (u, v) is row 21+v, byte s+21+u.  Lane = disc row v;
  //      the row's bytes u = -15..16 are 8 dwords, each weighted with v_dot4_u32_u8 (tables d_ic) ----
  if (lane < 31) {
    const int v = lane - 15;
    const uint32_t w1[8] = {w1a.x, w1a.y, w1a.z, w1a.w, w1b.x, w1b.y, w1b.z, w1b.w};
    const uint32_t wu[8] = {wua.x, wua.y, wua.z, wua.w, wub.x, wub.y, wub.z, wub.w};
    const uint32_t* rowp = raw + (6 + lane) * PW_WORDS + ((s + 6) >> 2);  // row 21 + v, first dword holding u = -15
    const uint32_t sh = (uint32_t)(s + 6) & 3u;
    uint32_t src[9];
#pragma unroll
    for (int j = 0; j < 9; j++) src[j] = rowp[j];
    uint32_t sumI = 0, sumU = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t B = __builtin_amdgcn_alignbyte(src[j + 1], src[j], sh);
      sumI = __builtin_amdgcn_udot4(B, w1[j], sumI, false);
      sumU = __builtin_amdgcn_udot4(B, wu[j], sumU, false);
    }
    atomicAdd(&msum[0], (int)sumU - 15 * (int)sumI);  // m10 = sum u*I
    atomicAdd(&msum[1], v * (int)sumI);                // m01 = sum v*I
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  const int m10 = msum[0], m01 = msum[1];
  const float angle = fast_atan2_deg((float)m01, (float)m10);
  DS_STAMP(2);
  // ---- horizontal pass: blurred column c (x = kx-18+c) uses window bytes s+c .. s+c+6 of the staged row.
  //      Item = (row pair, group of 4 columns): 4 dwords per row cover the 10 bytes the 4 columns need; the 190 items of
  //      the sampling disc (d_descHItems) in 3 steps.
  //      The byte shift s of the window inside its dwords is the same for the whole wave, so instead of shifting the data
  //      (9 v_alignbyte per row) the TAPS are shifted: for each s the weights of the four dwords are compile-time
  //      constants (descHTap), zero ones are skipped, and every s costs exactly 10 v_dot4_u32_u8 per row ----
  {
    auto hpass = [&](auto sTag) {
      constexpr int SH = decltype(sTag)::value;
#pragma unroll
      for (int it = 0; it < 3; it++) {
        const int rp = (int)(hitem[it] >> 4), gq = (int)(hitem[it] & 15u);
        if (hitem[it] != 0xffffu) {
          uint32_t hs[2][4];
#pragma unroll
          for (int h2 = 0; h2 < 2; h2++) {
            const int row = h2 ? min(2 * rp + 1, PW_ROWS - 1) : 2 * rp;
            const uint32_t* p = raw + row * PW_WORDS + gq;
            const uint32_t dd[4] = {p[0], p[1], p[2], p[3]};
#pragma unroll
            for (int c = 0; c < 4; c++) {
              uint32_t acc = 0;
#pragma unroll
              for (int m = 0; m < 4; m++)
                if (descHTap<GV>(SH + c, m) != 0u) acc = __builtin_amdgcn_udot4(dd[m], descHTap<GV>(SH + c, m), acc, false);
              hs[h2][c] = acc;
            }
          }
          uint4 o4;
          // (both sums are < 2^16: one v_perm_b32 packs the pair instead of a shift and an or)
          o4.x = __builtin_amdgcn_perm(hs[1][0], hs[0][0], 0x05040100u); o4.y = __builtin_amdgcn_perm(hs[1][1], hs[0][1], 0x05040100u);
          o4.z = __builtin_amdgcn_perm(hs[1][2], hs[0][2], 0x05040100u); o4.w = __builtin_amdgcn_perm(hs[1][3], hs[0][3], 0x05040100u);
          *reinterpret_cast<uint4*>(&hz2[rp * PW_COLS + 4 * gq]) = o4;
        }
      }
    };
    switch (s) {  // wave-uniform
      case 0: hpass(std::integral_constant<int, 0>{}); break;
      case 1: hpass(std::integral_constant<int, 1>{}); break;
      case 2: hpass(std::integral_constant<int, 2>{}); break;
      default: hpass(std::integral_constant<int, 3>{}); break;
    }
  }
  DS_STAMP(3);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  // ---- vertical pass + rounding with v_dot2_u32_u16 on row pairs: blurred rows 2q and 2q+1 both use pairs q..q+3,
  //      even row taps (18,34)(48,56)(48,34)(18,0), odd row taps (0,18)(34,48)(56,48)(34,18).  Item = (q, group of 4
  //      columns), 190 items over 3 steps; the blurred bytes overwrite the raw window (no longer needed) ----
  {
    constexpr uint32_t T0 = 18u, T1 = 34u, T2 = GV ? 49u : 48u, T3 = GV ? 55u : 56u;
    constexpr uint32_t E0 = T0 | (T1 << 16), E1 = T2 | (T3 << 16), E2 = T2 | (T1 << 16), E3 = T0;
    constexpr uint32_t O0 = T0 << 16, O1 = T1 | (T2 << 16), O2 = T3 | (T2 << 16), O3 = T1 | (T0 << 16);
    int q = lane / 10, gq = lane - q * 10;
#pragma unroll
    for (int it = 0; it < 3; it++) {
      if (q < BL_ROWS_PAD / 2) {
        const uint4 P0 = *reinterpret_cast<const uint4*>(&hz2[(q + 0) * PW_COLS + 4 * gq]);
        const uint4 P1 = *reinterpret_cast<const uint4*>(&hz2[(q + 1) * PW_COLS + 4 * gq]);
        const uint4 P2 = *reinterpret_cast<const uint4*>(&hz2[(q + 2) * PW_COLS + 4 * gq]);
        const uint4 P3 = *reinterpret_cast<const uint4*>(&hz2[(q + 3) * PW_COLS + 4 * gq]);
#define ORBX_VE(c) dot2u16(P0.c, E0, dot2u16(P1.c, E1, dot2u16(P2.c, E2, dot2u16(P3.c, E3, 32768u))))
#define ORBX_VO(c) dot2u16(P0.c, O0, dot2u16(P1.c, O1, dot2u16(P2.c, O2, dot2u16(P3.c, O3, 32768u))))
        uint32_t e0 = ORBX_VE(x), e1 = ORBX_VE(y), e2 = ORBX_VE(z), e3 = ORBX_VE(w);
        uint32_t o0 = ORBX_VO(x), o1 = ORBX_VO(y), o2 = ORBX_VO(z), o3 = ORBX_VO(w);
        if (GV) {  // taps that sum to 257: saturate_cast<uchar>
          e0 = min(e0, 0xffffffu); e1 = min(e1, 0xffffffu); e2 = min(e2, 0xffffffu); e3 = min(e3, 0xffffffu);
          o0 = min(o0, 0xffffffu); o1 = min(o1, 0xffffffu); o2 = min(o2, 0xffffffu); o3 = min(o3, 0xffffffu);
        }
#undef ORBX_VE
#undef ORBX_VO
        // each sum is < 2^24: its blurred byte is bits 16..23; v_perm_b32 gathers byte 2 of four sums into one dword
        bl32[(2 * q) * (PW_COLS / 4) + gq] =
            __builtin_amdgcn_perm(e1, e0, 0x0c0c0602u) | __builtin_amdgcn_perm(e3, e2, 0x06020c0cu);
        bl32[(2 * q + 1) * (PW_COLS / 4) + gq] =
            __builtin_amdgcn_perm(o1, o0, 0x0c0c0602u) | __builtin_amdgcn_perm(o3, o2, 0x06020c0cu);
      }
      q += 6; gq += 4;  // item + 64
      if (gq >= 10) { gq -= 10; q++; }
    }
  }
  DS_STAMP(4);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  // ---- steered BRIEF (cpp:169-228).  cos/sin of the f32 argument are evaluated in f64 and rounded to f32 ----
  const uint8_t* bl = reinterpret_cast<const uint8_t*>(bl32);
  const float factorPI = (float)(3.14159265358979323846 / 180.f);
  float cs, sn;
  if (libmFloat) {  // (uniform) ORBX_LIBM_FLOAT
    sincosfGlibc(angle * factorPI, &sn, &cs);
  } else {
    double sd, cd;
    ORBX_SINCOS((double)(angle * factorPI), &sd, &cd, true);
    cs = (float)cd; sn = (float)sd;
  }
  unsigned long long words[4];
#pragma unroll
  for (int wq = 0; wq < 4; wq++) {
    const float4 pt = pat[wq];
    const float x0 = pt.x, y0 = pt.y, x1 = pt.z, y1 = pt.w;
    // cvRound of the rotated coordinates (cpp:184-188) and the byte address (18 + r) * 40 + 18 + c in one go: v + 1.5 * 2^23
    // rounds to nearest-even at integer granularity and leaves 0x4B400000 + rint(v) in the float's bits; the low 24 bits
    // (0x400000 + r) go through v_mad_u32_u24, the constants are taken off at the end
    constexpr float MAGIC = 12582912.f;
    constexpr uint32_t OFF = 40u * 0x400000u + 0x4B400000u - (18u * PW_COLS + 18u);
    const uint32_t ir0 = __float_as_uint((x0 * sn + y0 * cs) + MAGIC), ic0 = __float_as_uint((x0 * cs - y0 * sn) + MAGIC);
    const uint32_t ir1 = __float_as_uint((x1 * sn + y1 * cs) + MAGIC), ic1 = __float_as_uint((x1 * cs - y1 * sn) + MAGIC);
    const int t0 = bl[(ir0 & 0xffffffu) * (uint32_t)PW_COLS + ic0 - OFF];
    const int t1 = bl[(ir1 & 0xffffffu) * (uint32_t)PW_COLS + ic1 - OFF];
    words[wq] = __ballot(t0 < t1);
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
  DS_STAMP(5);
}

// =================================================================================================
// K7+K8  ORBmatcher::SearchForInitialization.  One workgroup per frame pair.
// The reference walks the queries sequentially and lets earlier matches hide candidates from later queries
// (vMatchedDistance, ORBmatcher.cpp:67), so the query loop stays sequential inside the workgroup; the candidate
// scan of one query (window test + 256-bit Hamming + best / second best) runs across the workgroup's lanes.
// Tie-breaking equals the reference's candidate order (GetFeaturesInArea: cell x outer, cell y inner, index):
// the best candidate is the minimum of (distance, cellX*48+cellY, index).
// =================================================================================================
#define TH_LOW 50
#define HISTO_LENGTH 30
#define INF_DIST 0x7fffffff
#define MATCH_NONE 0x7fffffffffffffffull

#define MATCH_PENDING ((int)0x80000000)  // nmatches value meaning "handed on to the wide path"

struct MatchParams {
  int capacity;
  int window;
  float nnratio;
  int checkOri;
  int noGeneral;  // diagnostics: leave the pairs the parallel paths cannot take at MATCH_PENDING
  int pair0;  // first pair of this launch
  int dmax;   // the wide path lists only candidates with a smaller distance (launch_match)
  int noMfma; // diagnostics: no k_match_bf_mfma -- the brute-force case stays on the vector ALU (k_match_wide_lists, xor / bcnt)
  orbx_bounds b;
};

// 16 descriptor bits (the low half of w) as 16 signed bytes, +1 for a clear bit and -1 for a set one: the operand form of
// v_mfma_i32_32x32x32_i8 in which the dot product of two 256-bit descriptors is 256 - 2 x their Hamming distance.
// nibble * 0x00204081 has bit j of the nibble at bit 8 j (its other copies are masked off); the 0 / 1 bytes then select byte 0
// (0x01) or byte 1 (0xff) of a constant through v_perm_b32: four instructions per four bytes.
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ v4i_t pm1Bytes16(const uint32_t w) {
  v4i_t r;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t n = (w >> (4 * j)) & 0xfu;
    r[j] = (int)__builtin_amdgcn_perm(0u, 0x0000ff01u, (n * 0x00204081u) & 0x01010101u);
  }
  return r;
}

__device__ __forceinline__ int hamming256(const uint4 a0, const uint4 a1, const uint32_t* __restrict__ b) {
  const uint4 b0 = reinterpret_cast<const uint4*>(b)[0], b1 = reinterpret_cast<const uint4*>(b)[1];
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
         __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

#define MW_T 1024
#define MW_R (MW_CAP / MW_T)
#define MW_END 0xffff
#define MW_SWEEPS 1024  // after sweep i the first i queries are final; in practice a handful of sweeps

// matchWidePrep: first step of the wide path (see k_match_wide_lists / k_match_wide_resolve below), run by the workgroup
// of k_match_jacobi that hands its pair on: ordered compaction of the octave-0 queries (F1 index) and of the eligible
// trains (x, y, grid cell, cell << 20 | F2 index) into the pair's scratch, and matches12 = -1.  The trains are stored by grid
// column (a counting sort; the slot numbers are mere names: nothing downstream depends on their order), with the columns' start
// slots behind them, and the queries get a second order by THEIR column (qPerm): k_match_wide_lists then gives a wave 64
// queries of neighbouring columns and walks only the trains of the columns their windows reach -- a twelfth of them at
// 3840x2160 with the 100-pixel window.  Every thread of the workgroup must call it (it has barriers).
template <int T>
__device__ void matchWidePrep(const int pair, const int* __restrict__ pairFirst, const int* __restrict__ pairSecond,
                              const orbx_keypoint* __restrict__ kps, const int* __restrict__ nkp, const MatchParams& mp,
                              int* __restrict__ matches12, int* __restrict__ scratch, long long scratchStride, int capl) {
  __shared__ int wq[T / 64], wt[T / 64];
  __shared__ int sBaseQ, sBaseT;
  __shared__ int colT[ORBX_GRID_COLS + 1], colQ[ORBX_GRID_COLS + 1];  // per grid column: count, then start / fill position
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fa = pairFirst[pair], fb = pairSecond[pair];
  const int cap = mp.capacity;
  const int n1 = min(max(nkp[fa], 0), cap), n2 = min(max(nkp[fb], 0), cap);  // (a count beyond the frame's slots is clamped)
  const orbx_keypoint* k1 = kps + (long long)fa * cap;
  const orbx_keypoint* k2 = kps + (long long)fb * cap;
  int* m12 = matches12 + (long long)pair * cap;
  int* S = scratch + (long long)pair * scratchStride;
  uint4* trec = reinterpret_cast<uint4*>(S + MW_HDR);
  int* qIdx = S + MW_HDR + 4 * capl;
  // staging (both regions are written by the later kernels only): the trains in index order, the queries' columns
  uint4* tmpRec = reinterpret_cast<uint4*>(S + MW_HDR + ((6 * capl + 3) & ~3));  // (the lists' space: MW_CP >= 5 rows; 16-byte aligned also with an odd capacity)
  int* tmpQc = S + MW_HDR + 5 * capl;                                 // (the counts' space)
  int* qPerm = S + MW_HDR + (6 + MW_CP + MW_TOPK) * capl;
  int* cxStart = qPerm + capl;
  static_assert(MW_CP >= 5, "the train records are staged in the lists' space");
  if (t <= ORBX_GRID_COLS) { colT[t] = 0; colQ[t] = 0; }
  // (only when a window spans less than half of the grid's columns: a window that reaches most columns prunes nothing, and the
  // brute-force configurations read the descriptors faster in index order -- header [8] tells k_match_wide_lists)
  const bool byCol = 2.f * (float)mp.window * ((float)ORBX_GRID_COLS / (float)(mp.b.max_x - mp.b.min_x)) + 3.f < 0.5f * (float)ORBX_GRID_COLS;
  const float wInv = (float)ORBX_GRID_COLS / (float)(mp.b.max_x - mp.b.min_x);  // Frame.cpp:46-47
  const float hInv = (float)ORBX_GRID_ROWS / (float)(mp.b.max_y - mp.b.min_y);
  const float fminX = (float)mp.b.min_x, fminY = (float)mp.b.min_y;
  if (t == 0) { sBaseQ = 0; sBaseT = 0; }
  __syncthreads();
  const int n = max(n1, n2);
  // bounding box of the eligible trains (header [4..7]): lets k_match_wide_lists recognise the brute-force case -- a window that
  // covers every train for every query of a workgroup -- and skip the per-pair window tests there
  float bbx0 = 3.0e38f, bbx1 = -3.0e38f, bby0 = 3.0e38f, bby1 = -3.0e38f;
  for (int i0 = 0; i0 < n; i0 += T) {
    const int i = i0 + t;
    bool okQ = false, okT = false;
    float tx = 0.f, ty = 0.f;
    int px = 0, py = 0;
    int qc = 0;
    if (i < n1) {
      const orbx_keypoint kq = k1[i];
      okQ = !(kq.octave > 0);  // ORBmatcher.cpp:38-39
      qc = min(max((int)floorf((kq.x - fminX) * wInv), 0), ORBX_GRID_COLS - 1);  // the query's own column (an ordering aid only)
      m12[i] = -1;
    }
    if (i < n2) {
      const orbx_keypoint kp = k2[i];
      // Frame::PosInGrid (Frame.cpp:89-99) + the octave filter of GetFeaturesInArea (Frame.cpp:179,191)
      px = (int)roundf((kp.x - fminX) * wInv);
      py = (int)roundf((kp.y - fminY) * hInv);
      okT = kp.octave == 0 && px >= 0 && px < ORBX_GRID_COLS && py >= 0 && py < ORBX_GRID_ROWS;
      tx = kp.x; ty = kp.y;
      if (okT) { bbx0 = fminf(bbx0, tx); bbx1 = fmaxf(bbx1, tx); bby0 = fminf(bby0, ty); bby1 = fmaxf(bby1, ty); }
    }
    const unsigned long long bq = __ballot(okQ), bt = __ballot(okT);
    if (lane == 0) { wq[wave] = __popcll(bq); wt[wave] = __popcll(bt); }
    __syncthreads();
    int beforeQ = sBaseQ, beforeT = sBaseT;
    for (int w2 = 0; w2 < wave; w2++) { beforeQ += wq[w2]; beforeT += wt[w2]; }
    const unsigned long long below = (1ull << lane) - 1ull;
    const int posQ = beforeQ + __popcll(bq & below), posT = beforeT + __popcll(bt & below);
    if (okQ && posQ < capl) {
      qIdx[posQ] = i;
      if (byCol) { tmpQc[posQ] = qc; atomicAdd(&colQ[qc], 1); }
    }
    if (okT && posT < capl) {
      (byCol ? tmpRec : trec)[posT] = make_uint4(__float_as_uint(tx), __float_as_uint(ty), (uint32_t)px | ((uint32_t)py << 8),
                                                  ((uint32_t)(px * ORBX_GRID_ROWS + py) << 20) | (uint32_t)i);
      if (byCol) atomicAdd(&colT[px], 1);
    }
    __syncthreads();
    if (t == 0) {
      int tq = sBaseQ, tt = sBaseT;
      for (int w2 = 0; w2 < T / 64; w2++) { tq += wq[w2]; tt += wt[w2]; }
      sBaseQ = tq; sBaseT = tt;
    }
    __syncthreads();
  }
  {  // workgroup reduction of the bounding box (wq / wt are free again: reused as float slots through LDS arrays of their own)
    __shared__ float bbW[4][T / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      bbx0 = fminf(bbx0, __shfl_xor(bbx0, o)); bbx1 = fmaxf(bbx1, __shfl_xor(bbx1, o));
      bby0 = fminf(bby0, __shfl_xor(bby0, o)); bby1 = fmaxf(bby1, __shfl_xor(bby1, o));
    }
    if (lane == 0) { bbW[0][wave] = bbx0; bbW[1][wave] = bbx1; bbW[2][wave] = bby0; bbW[3][wave] = bby1; }
    __syncthreads();
    if (t == 0) {
      for (int w2 = 1; w2 < T / 64; w2++) {
        bbx0 = fminf(bbx0, bbW[0][w2]); bbx1 = fmaxf(bbx1, bbW[1][w2]); bby0 = fminf(bby0, bbW[2][w2]); bby1 = fmaxf(bby1, bbW[3][w2]);
      }
      S[4] = (int)__float_as_uint(bbx0); S[5] = (int)__float_as_uint(bbx1); S[6] = (int)__float_as_uint(bby0); S[7] = (int)__float_as_uint(bby1);
    }
  }
  if (t == 0) {
    S[0] = sBaseQ; S[1] = sBaseT;
    S[2] = (sBaseQ > capl || sBaseT > capl || n2 > 0xfffff) ? 1 : 0;
    S[3] = 0;
    S[8] = byCol ? 1 : 0;
    S[9] = 0; S[10] = 0;  // k_match_bf_mfma's mask of the 64-query blocks it has listed
  }
  if (!byCol) return;  // (uniform)
  // ---- counting sort by grid column: exclusive starts (one wave), then every staged record / query takes the next free slot of
  //      its column ----
  __syncthreads();  // (the staged records and the column counts are complete; global writes of this workgroup are visible to it)
  if (wave == 0) {
    const int cT = colT[lane], cQ = colQ[lane];
    int iT = cT, iQ = cQ;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int uT = __shfl_up(iT, o), uQ = __shfl_up(iQ, o);
      if (lane >= o) { iT += uT; iQ += uQ; }
    }
    colT[lane] = iT - cT; colQ[lane] = iQ - cQ;
    cxStart[lane] = iT - cT;
    if (lane == 63) { colT[64] = iT; cxStart[64] = iT; }
  }
  __syncthreads();
  const int nTs = min(sBaseT, capl), nQs = min(sBaseQ, capl);
  for (int e = t; e < nTs; e += T) {
    const uint4 r = tmpRec[e];
    trec[atomicAdd(&colT[r.z & 0xff], 1)] = r;
  }
  for (int q = t; q < nQs; q += T) qPerm[atomicAdd(&colQ[tmpQc[q]], 1)] = q;
}


// Diagnostic build only (-DORBX_MJ_STAMPS): per workgroup of k_match_jacobi, s_memtime behind each phase (a barrier first);
// tools/mj_stamps.py prints the shares.
#ifdef ORBX_MJ_STAMPS
__device__ unsigned long long g_mjStamps[1024 * 16];
#define MJ_STAMP(k)                                                                                      \
  do {                                                                                                   \
    __syncthreads();                                                                                     \
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_mjStamps[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
extern "C" int orbx_diag_mj_stamps(unsigned long long* out, int nWgs) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mjStamps), sizeof(unsigned long long) * 16 * (size_t)nWgs);
}
#else
#define MJ_STAMP(k) do { } while (0)
#endif

// -------------------------------------------------------------------------------------------------
// k_match_jacobi: one workgroup per frame pair, ONE THREAD PER QUERY.
// The reference's query loop is sequential only through vMatchedDistance: query q sees, for every train t, the smallest
// distance with which an EARLIER query claimed t (each accepted match overwrites it with a smaller value,
// ORBmatcher.cpp:67,103).  So the outcome of q is a function of the outcomes of the queries < q, and the sequential
// result is the unique fixpoint of "evaluate all queries in parallel against the claims of the previous sweep":
// after sweep i the first i queries are final, and in practice a handful of sweeps suffice (the loop stops when a
// sweep changes nothing).  Claims are kept per train (up to MJ_K; more, or no convergence within MJ_SWEEPS, hands
// the pair to the sequential kernels).  Everything else (vnMatches21 stealing, nmatches, rotation histogram with its
// double-decrement quirk) follows from the final outcomes: a train belongs to its LAST claimant.
// All threads walk the trains together, so train data are LDS broadcasts.
// -------------------------------------------------------------------------------------------------
// Up to MJ_CAP (256) octave-0 queries / eligible trains per pair, MJ_P (4) threads per query (thread t works for query
// t % MJ_CAP on the trains e with e % MJ_P == t / MJ_CAP), MJ_CP (16) listed trains (distance below dmax) of one query
// per part - 64 KB of lists.  (8 per part, 32 KB, let a pair or two of every bench batch overflow into the wide path, whose
// kernels then travelled with every batch: 302 k frames/s against 305 k with 16; 24 per part = 98 KB was measured 0.3 %
// slower in round 1.)  A larger or fuller pair is marked MATCH_PENDING for the wide path below.
#define MJ_K 8   // (4 let two or three pairs of every other bench batch overflow: a train with five claimants)
#define MJ_SWEEPS 64

template <int MJ_CAP, int MJ_P, int MJ_CP>
__global__ __launch_bounds__(MJ_CAP * MJ_P) void k_match_jacobi(const int* __restrict__ pairFirst, const int* __restrict__ pairSecond,
                                                               const orbx_keypoint* __restrict__ kps,
                                                               const uint8_t* __restrict__ desc, const int* __restrict__ nkp,
                                                               const MatchParams mp, int* __restrict__ matches12,
                                                               int* __restrict__ nmatchesOut, int* __restrict__ statsOut,
                                                               int* __restrict__ scratch, long long scratchStride, int capl,
                                                               int* __restrict__ hostWide) {
  constexpr int MJ_T = MJ_CAP * MJ_P;
  ORBX_SETPRIO();
  __shared__ float2 tXY[MJ_CAP];
  __shared__ float tAng[MJ_CAP];
  __shared__ uint4 tDesc[MJ_CAP][2];  // a train's 256 bits: two 16-byte reads per distance
  __shared__ uint16_t tIdx[MJ_CAP];
  __shared__ uint8_t tCx[MJ_CAP], tCy[MJ_CAP];
  __shared__ int clCount[MJ_CAP], lastQ[MJ_CAP];
  __shared__ uint16_t clQ[MJ_CAP][MJ_K], clD[MJ_CAP][MJ_K];
  __shared__ uint32_t candList[MJ_P * MJ_CP * MJ_CAP];  // [part][k][query]: dist << 16 | train slot
  __shared__ uint32_t tOrd[MJ_CAP];                      // cell << 20 | F2 index: the reference's candidate order
  __shared__ unsigned long long pBest[MJ_P][MJ_CAP];     // per part: best key of the sweep
  __shared__ uint32_t pAux[MJ_P][MJ_CAP];                // per part: second-best distance (0xffff = none) | best train slot << 16
  __shared__ uint8_t pCnt[MJ_P][MJ_CAP];                 // per part: candidates listed for the query
  __shared__ int hist[HISTO_LENGTH];
  __shared__ int sChangedSw[MJ_SWEEPS];
  __shared__ int sNT, sBase, sOverflow, sNm, sBadDist, sBadRatio, sBadOri, sKeep[3], sKeepV[3];
  __shared__ int sTooMany[2];  // per parity of the train-staging chunk: more than MJ_CAP eligible trains so far
  __shared__ int wcnt[MJ_T / 64];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int part = t / MJ_CAP, q = t - part * MJ_CAP;  // waves 0-3 = part 0, ...
  const int pair = blockIdx.x + mp.pair0;
  const int fa = pairFirst[pair], fb = pairSecond[pair];
  const int cap = mp.capacity;
  const int n1 = min(max(nkp[fa], 0), cap), n2 = min(max(nkp[fb], 0), cap);  // (a count beyond the frame's slots is clamped)
  const orbx_keypoint* k1 = kps + (long long)fa * cap;
  const orbx_keypoint* k2 = kps + (long long)fb * cap;
  const uint32_t* d1 = reinterpret_cast<const uint32_t*>(desc + (long long)fa * cap * 32);
  const uint32_t* d2 = reinterpret_cast<const uint32_t*>(desc + (long long)fb * cap * 32);
  int* m12 = matches12 + (long long)pair * cap;

  const float wInv = (float)ORBX_GRID_COLS / (float)(mp.b.max_x - mp.b.min_x);  // Frame.cpp:46-47
  const float hInv = (float)ORBX_GRID_ROWS / (float)(mp.b.max_y - mp.b.min_y);
  const float fminX = (float)mp.b.min_x, fminY = (float)mp.b.min_y;
  MJ_STAMP(0);
  if (t == 0) { sNT = 0; sBase = 0; sOverflow = n2 > 65535 ? 1 : 0; sNm = 0; sBadDist = 0; sBadRatio = 0; sBadOri = 0; sTooMany[0] = 0; sTooMany[1] = 0; }
  if (t < 3) { sKeep[t] = -1; sKeepV[t] = 0; }
  if (t < MJ_SWEEPS) sChangedSw[t] = 0;
  if (t < HISTO_LENGTH) hist[t] = 0;
  __syncthreads();
  // F1's keypoint t and its descriptor are requested now: their latency runs under the staging of F2's trains
  orbx_keypoint pk1{};
  uint32_t pd1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (t < n1) {
    pk1 = k1[t];
#pragma unroll
    for (int w = 0; w < 8; w++) pd1[w] = d1[(long long)t * 8 + w];
  }
  // ---- stage the grid-eligible octave-0 trains of F2 (slot order is irrelevant: ties are broken by cell and index);
  //      keypoint and descriptor are loaded together (one round trip, not two) ----
  for (int j0 = 0; j0 < n2; j0 += MJ_T) {
    const int j = j0 + t;
    if (j < n2) {
      const orbx_keypoint kp = k2[j];
      uint32_t dw[8];
#pragma unroll
      for (int w = 0; w < 8; w++) dw[w] = d2[(long long)j * 8 + w];
      // Frame::PosInGrid (Frame.cpp:89-99) + the octave filter of GetFeaturesInArea (Frame.cpp:179,191)
      const int px = (int)roundf((kp.x - fminX) * wInv), py = (int)roundf((kp.y - fminY) * hInv);
      if (kp.octave == 0 && px >= 0 && px < ORBX_GRID_COLS && py >= 0 && py < ORBX_GRID_ROWS) {
        const int slot = atomicAdd(&sNT, 1);
        if (slot < MJ_CAP) {
          tXY[slot] = make_float2(kp.x, kp.y); tAng[slot] = kp.angle;
          tIdx[slot] = (uint16_t)j; tCx[slot] = (uint8_t)px; tCy[slot] = (uint8_t)py;
          tDesc[slot][0] = make_uint4(dw[0], dw[1], dw[2], dw[3]);
          tDesc[slot][1] = make_uint4(dw[4], dw[5], dw[6], dw[7]);
        }
      }
    }
    // more eligible trains than this kernel takes: no need to look at the rest.  The decision must not be read from sNT behind
    // the barrier (a wave that has passed it may already be adding the next chunk's trains while a slower one still compares):
    // every thread looks at the counter BEFORE the barrier -- the thread whose add came last sees the chunk's total -- and raises
    // the chunk's own flag, which nothing writes again before every wave has read it (the other parity's flag is the next chunk's)
    if (sNT > MJ_CAP) sTooMany[(j0 / MJ_T) & 1] = 1;
    __syncthreads();
    if (sTooMany[(j0 / MJ_T) & 1]) break;  // (uniform)
  }
  if (sNT > MJ_CAP || sOverflow) {  // block-uniform: the pair goes to the wide path (matchWidePrep starts from the keypoints again)
    if (t == 0) { nmatchesOut[pair] = MATCH_PENDING; *hostWide = 1; }  // (mapped host memory: the batch needs the wide path)
    matchWidePrep<MJ_T>(pair, pairFirst, pairSecond, kps, nkp, mp, matches12, scratch, scratchStride, capl);
    return;
  }
  MJ_STAMP(1);
  // ---- the q-th octave-0 keypoint of F1 in index order is query q ----
  float qx = 0, qy = 0, qang = 0;
  int qi = -1;
  uint32_t qd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // the queries' data reach their threads through LDS (candList's space, free until the lists are built)
  float* const sQx = reinterpret_cast<float*>(candList);
  float* const sQy = sQx + MJ_CAP;
  float* const sQa = sQy + MJ_CAP;
  uint32_t* const sQd = reinterpret_cast<uint32_t*>(sQa + MJ_CAP);  // [8][MJ_CAP]
  static_assert(MJ_P * MJ_CP >= 11, "the query staging area must fit into candList");
  for (int i0 = 0; i0 < n1; i0 += MJ_T) {
    const int i = i0 + t;
    bool ok = false;
    orbx_keypoint kq = pk1;
    uint32_t dq[8];
#pragma unroll
    for (int w = 0; w < 8; w++) dq[w] = pd1[w];
    if (i0 > 0 && i < n1) {  // (more than MJ_T keypoints in F1: the later ones are loaded here)
      kq = k1[i];
#pragma unroll
      for (int w = 0; w < 8; w++) dq[w] = d1[(long long)i * 8 + w];
    }
    if (i < n1) {
      ok = !(kq.octave > 0);  // ORBmatcher.cpp:38-39
      m12[i] = -1;
    }
    const unsigned long long bm = __ballot(ok);
    if (lane == 0) wcnt[wave] = __popcll(bm);
    __syncthreads();
    int before = sBase;
    for (int w2 = 0; w2 < wave; w2++) before += wcnt[w2];
    const int pos = before + __popcll(bm & ((1ull << lane) - 1ull));
    if (ok && pos < MJ_CAP) {
      clCount[pos] = i;  // clCount is free until the sweeps start: F1 index of query `pos`
      sQx[pos] = kq.x; sQy[pos] = kq.y; sQa[pos] = kq.angle;
#pragma unroll
      for (int w = 0; w < 8; w++) sQd[w * MJ_CAP + pos] = dq[w];
    }
    __syncthreads();
    if (t == 0) {
      int tot = sBase;
      for (int w2 = 0; w2 < MJ_T / 64; w2++) tot += wcnt[w2];
      sBase = tot;
    }
    __syncthreads();
  }
  MJ_STAMP(2);
  const int nQ = sBase, nT = sNT;
  if (nQ > MJ_CAP || nT > MJ_CAP || sOverflow) {  // block-uniform: the pair goes to the wide path
    if (t == 0) { nmatchesOut[pair] = MATCH_PENDING; *hostWide = 1; }  // (mapped host memory: the batch needs the wide path)
    matchWidePrep<MJ_T>(pair, pairFirst, pairSecond, kps, nkp, mp, matches12, scratch, scratchStride, capl);
    return;
  }
  if (q < nQ) {  // every part takes its query
    qi = clCount[q];
    qx = sQx[q]; qy = sQy[q]; qang = sQa[q];
#pragma unroll
    for (int w = 0; w < 8; w++) qd[w] = sQd[w * MJ_CAP + q];
  }
  __syncthreads();
  if (t < MJ_CAP) clCount[t] = 0;
  __syncthreads();
  MJ_STAMP(3);
  // cell window of my query, Frame.cpp:167-177
  const float r = (float)mp.window;
  const int minCX = max(0, (int)floorf((qx - fminX - r) * wInv));
  const int maxCX = min(ORBX_GRID_COLS - 1, (int)ceilf((qx - fminX + r) * wInv));
  const int minCY = max(0, (int)floorf((qy - fminY - r) * hInv));
  const int maxCY = min(ORBX_GRID_ROWS - 1, (int)ceilf((qy - fminY + r) * hInv));
  const bool hasWindow = q < nQ && !(minCX >= ORBX_GRID_COLS || maxCX < 0 || minCY >= ORBX_GRID_ROWS || maxCY < 0);

  // ---- candidate list of (my query, my part of the trains), built once: every train inside the window, with its
  //      Hamming distance.  The order inside a list is irrelevant: the comparison key is a total order.
  //      Two steps: the distance test |dx| < r && |dy| < r of every train of the part (all threads of a wave walk the same
  //      trains: one 8-byte LDS broadcast and five vector instructions per train) leaves a 64-bit mask per thread; the cell
  //      test of GetFeaturesInArea (Frame.cpp:167-177 -- it can only drop a train whose rounded cell lies outside the window's
  //      cell range) and the 256-bit distance are then evaluated for the set bits only.  (All three in one loop made every
  //      wave pay the distance for every train some lane had in its window -- nearly all, at 13 % useful lanes: 56 k of the
  //      kernel's 86 k cycles; the cell test in the first loop still cost 25 k.) ----
  static_assert(MJ_CAP / MJ_P <= 64, "one mask bit per train of a part");
  uint32_t* const myList = candList + (size_t)part * MJ_CP * MJ_CAP + q;
  int nCand = 0;
  unsigned long long win = 0ull;  // bit i: train part + MJ_P * i passes the distance test of my query
  if (hasWindow) {
    for (int e0 = part, s4 = 0; e0 < nT; e0 += 4 * MJ_P, s4 += 4) {  // four trains per step: their LDS broadcasts are in flight together
      float2 txy[4];
#pragma unroll
      for (int j = 0; j < 4; j++) txy[j] = tXY[min(e0 + j * MJ_P, MJ_CAP - 1)];
      uint32_t nib = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bool in = (e0 + j * MJ_P) < nT && (fabsf(txy[j].x - qx) < r && fabsf(txy[j].y - qy) < r);
        nib |= (uint32_t)in << j;
      }
      win |= (unsigned long long)nib << s4;
    }
  }
  bool anyIn = false;
  MJ_STAMP(7);
  while (win) {
    const int e = part + MJ_P * __builtin_ctzll(win);
    win &= win - 1ull;
    const int cx = tCx[e], cy = tCy[e];
    if (cx < minCX || cx > maxCX || cy < minCY || cy > maxCY) continue;
    anyIn = true;
    const uint4 ta = tDesc[e][0], tb = tDesc[e][1];
    const int dist = __popc(qd[0] ^ ta.x) + __popc(qd[1] ^ ta.y) + __popc(qd[2] ^ ta.z) + __popc(qd[3] ^ ta.w) +
                     __popc(qd[4] ^ tb.x) + __popc(qd[5] ^ tb.y) + __popc(qd[6] ^ tb.z) + __popc(qd[7] ^ tb.w);
    if (dist >= mp.dmax) continue;  // can neither be accepted nor fail the ratio test of a nearer train (launch_match)
    if (nCand < MJ_CP) myList[nCand * MJ_CAP] = ((uint32_t)dist << 16) | (uint32_t)e;
    nCand++;
  }
  if (nCand > MJ_CP) sOverflow = 1;
  pCnt[part][q] = (uint8_t)(min(nCand, 127) | (anyIn ? 0x80 : 0));  // bit 7: the part saw a train in the window
  if (t < nT) tOrd[t] = ((uint32_t)(tCx[t] * ORBX_GRID_ROWS + tCy[t]) << 20) | (uint32_t)tIdx[t];
  __syncthreads();
  if (sOverflow) {  // a window too full for the lists: the pair goes to the wide path
    if (t == 0) { nmatchesOut[pair] = MATCH_PENDING; *hostWide = 1; }  // (mapped host memory: the batch needs the wide path)
    matchWidePrep<MJ_T>(pair, pairFirst, pairSecond, kps, nkp, mp, matches12, scratch, scratchStride, capl);
    return;
  }
  MJ_STAMP(4);
  // a query has a candidate in its window iff some part listed one (vIndices2.empty() -> continue, ORBmatcher.cpp:46-47)
  bool hasCand = false;
#pragma unroll
  for (int pp = 0; pp < MJ_P; pp++) hasCand |= pCnt[pp][q] != 0;
  // outcome of my query: 0 = no candidate in the window, 1 = invalid by distance, 2 = invalid by ratio, 3 = accepted
  // (kept by the part-0 thread of the query)
  int outcome = 0, bestT = -1, bestD = 0;
  bool converged = false;
#ifdef ORBX_MJ_STAMPS
  int sweepsDone_ = 0;
#endif
  for (int sweep = 0; sweep < MJ_SWEEPS; sweep++) {
#ifdef ORBX_MJ_STAMPS
    sweepsDone_ = sweep + 1;
#endif
    // ---- every part scans its share of the query's candidates ----
    unsigned long long best = MATCH_NONE;  // dist << 32 | cell << 20 | train index
    int second = INF_DIST, bt = 0;
    // four candidates per step: their list entries, claim counts and order keys are independent LDS reads that
    // are in flight together (this loop is bound by LDS round trips)
    for (int k = 0; k < nCand; k += 4) {
      uint32_t ce[4], ord[4];
      int cnt[4];
#pragma unroll
      for (int j = 0; j < 4; j++) ce[j] = myList[min(k + j, MJ_CP - 1) * MJ_CAP];
#pragma unroll
      for (int j = 0; j < 4; j++) { cnt[j] = clCount[ce[j] & 0xffff]; ord[j] = tOrd[ce[j] & 0xffff]; }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (k + j >= nCand) break;
        const int e = ce[j] & 0xffff, dist = (int)(ce[j] >> 16);
        // vMatchedDistance[e] as query q sees it: smallest distance of an earlier accepted query that chose e
        int md = INF_DIST;
        const int nc = min(cnt[j], MJ_K);
        for (int c = 0; c < nc; c++)
          if ((int)clQ[e][c] < q) md = min(md, (int)clD[e][c]);
        if (md <= dist) continue;  // ORBmatcher.cpp:67
        const unsigned long long key = ((unsigned long long)dist << 32) | ord[j];
        if (key < best) {
          second = min(second, (int)(best >> 32));
          best = key;
          bt = e;
        } else {
          second = min(second, dist);
        }
      }
    }
    pBest[part][q] = best;
    pAux[part][q] = (uint32_t)min(second, 0xffff) | ((uint32_t)bt << 16);
    __syncthreads();
    // ---- part 0 merges the four partial results: best = smallest key, second = second smallest distance overall ----
    bool changed = false;
    if (part == 0) {
      int nOutcome = 0, nBestT = -1, nBestD = 0;
#pragma unroll
      for (int pp = 0; pp < MJ_P; pp++) {
        const unsigned long long kb = pBest[pp][q];
        const uint32_t ax = pAux[pp][q];
        const int sec = (int)(ax & 0xffff);
        if (pp == 0) { best = kb; bt = (int)(ax >> 16); second = sec == 0xffff ? INF_DIST : sec; continue; }
        if (sec != 0xffff) second = min(second, sec);
        if (kb == MATCH_NONE) continue;
        if (kb < best) {
          if (best != MATCH_NONE) second = min(second, (int)(best >> 32));
          best = kb;
          bt = (int)(ax >> 16);
        } else {
          second = min(second, (int)(kb >> 32));
        }
      }
      const int bd = (int)(best >> 32);
      if (hasCand) {
        if (best == MATCH_NONE || bd > TH_LOW) nOutcome = 1;
        else if ((float)bd > mp.nnratio * (float)second) nOutcome = 2;
        else { nOutcome = 3; nBestT = bt; nBestD = bd; }
      }
      changed = nOutcome != outcome || nBestT != bestT || nBestD != bestD;
      outcome = nOutcome; bestT = nBestT; bestD = nBestD;
    }
    // (one flag per sweep, cleared at the start of the kernel: no barrier to reset it; the claim counters are cleared here, behind
    // the barrier that ended their last readers)
    if (changed) sChangedSw[sweep] = 1;
    if (t < MJ_CAP) clCount[t] = 0;
    __syncthreads();
    if (!sChangedSw[sweep]) { converged = true; break; }
    if (outcome == 3) {
      const int slot = atomicAdd(&clCount[bestT], 1);
      if (slot < MJ_K) { clQ[bestT][slot] = (uint16_t)q; clD[bestT][slot] = (uint16_t)bestD; }
      else sOverflow = 1;
    }
    __syncthreads();
    if (sOverflow) break;
  }
  if (!converged) {  // block-uniform (the sweep's flag / sOverflow are read after barriers)
    if (t == 0) { nmatchesOut[pair] = MATCH_PENDING; *hostWide = 1; }  // (mapped host memory: the batch needs the wide path)
    matchWidePrep<MJ_T>(pair, pairFirst, pairSecond, kps, nkp, mp, matches12, scratch, scratchStride, capl);
    return;
  }
  MJ_STAMP(5);
#ifdef ORBX_MJ_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_mjStamps[blockIdx.x * 16 + 8] = (unsigned long long)sweepsDone_;
#endif
  // ---- final bookkeeping from the converged outcomes (part-0 threads hold them; the others have outcome 0) ----
  if (t < MJ_CAP) lastQ[t] = -1;
  __syncthreads();
  if (outcome == 3) atomicMax(&lastQ[bestT], q);
  if (outcome == 1) atomicAdd(&sBadDist, 1);
  if (outcome == 2) atomicAdd(&sBadRatio, 1);
  int bin = -1;
  if (outcome == 3 && mp.checkOri) {
    float rot = qang - tAng[bestT];
    if (rot < 0.0f) rot += 360.0f;
    bin = (int)roundf(rot * (HISTO_LENGTH / 360.0f));
    if (bin == HISTO_LENGTH) bin = 0;
    if (bin < 0 || bin >= HISTO_LENGTH) bin = -1;
    if (bin >= 0) atomicAdd(&hist[bin], 1);
  }
  __syncthreads();
  // nmatches before pruning = trains that ended up with a claimant (every steal took one match away again)
  if (t < nT && lastQ[t] >= 0) atomicAdd(&sNm, 1);
  if (outcome == 3 && lastQ[bestT] == q) m12[qi] = (int)tIdx[bestT];
  if (mp.checkOri) {
    // ComputeThreeMaxima, ORBmatcher.cpp:152-183: its strict-greater cascade keeps the three largest bins by (size descending,
    // index ascending) among the non-empty ones -- bin t's place is the number of bins that come before it in that order
    if (t < HISTO_LENGTH) {
      const int v = hist[t];
      int place = 0;
      for (int i = 0; i < HISTO_LENGTH; i++) {
        const int o = hist[i];
        place += (o > v) || (o == v && i < t);
      }
      if (v > 0 && place < 3) { sKeep[place] = t; sKeepV[place] = v; }
    }
    __syncthreads();
    int ind1 = sKeep[0], ind2 = sKeep[1], ind3 = sKeep[2];
    {
      const int max1 = sKeepV[0], max2 = sKeepV[1], max3 = sKeepV[2];
      if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
      else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
    }
    // every accepted query sits in rotHist, also one whose match was stolen later (double decrement, :130-138)
    if (bin >= 0 && bin != ind1 && bin != ind2 && bin != ind3) {
      m12[qi] = -1;
      atomicSub(&sNm, 1);
      atomicAdd(&sBadOri, 1);
    }
  }
  __syncthreads();
  if (t == 0) {
    nmatchesOut[pair] = sNm;
    if (statsOut) { statsOut[pair * 3] = sBadDist; statsOut[pair * 3 + 1] = sBadRatio; statsOut[pair * 3 + 2] = sBadOri; }
  }
  MJ_STAMP(6);
}

// -------------------------------------------------------------------------------------------------
// matchGeneral: the reference's loop as written - queries one after the other, the workgroup's threads share the trains
// of one query (any size; vMatchedDistance, vnMatches21 and the rotation bins in the pair's global scratch).  Run by the
// workgroup of k_match_wide_resolve for a pair the parallel paths cannot take.  Every thread must call it.
// -------------------------------------------------------------------------------------------------
template <int T>
__device__ void matchGeneral(const int pair, const int* __restrict__ pairFirst, const int* __restrict__ pairSecond,
                             const orbx_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                             const int* __restrict__ nkp, const MatchParams& mp, int* __restrict__ matches12,
                             int* __restrict__ nmatchesOut, int* __restrict__ statsOut, int* __restrict__ scratch,
                             long long scratchStride) {
  __shared__ unsigned long long sBest[T / 64];
  __shared__ int sSecond[T / 64], sAny[T / 64];
  __shared__ int hist[HISTO_LENGTH];
  __shared__ int sNm, sBadDist, sBadRatio, sBadOri, sKeep[3];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fa = pairFirst[pair], fb = pairSecond[pair];
  const int cap = mp.capacity;
  const int n1 = min(max(nkp[fa], 0), cap), n2 = min(max(nkp[fb], 0), cap);  // (a count beyond the frame's slots is clamped)
  const orbx_keypoint* k1 = kps + (long long)fa * cap;
  const orbx_keypoint* k2 = kps + (long long)fb * cap;
  const uint8_t* d1 = desc + (long long)fa * cap * 32;
  const uint8_t* d2 = desc + (long long)fb * cap * 32;
  int* m12 = matches12 + (long long)pair * cap;
  int* md = scratch + (long long)pair * scratchStride;  // vMatchedDistance (the pair's scratch holds >= 4 * cap ints)
  int* m21 = md + cap;                            // vnMatches21
  int* accBin = m21 + cap;                        // rotation bin of every accepted query (rotHist membership)
  int* cell2 = accBin + cap;                      // F2 grid cell (cx*48+cy) of every eligible train, -1 otherwise

  const float wInv = (float)ORBX_GRID_COLS / (float)(mp.b.max_x - mp.b.min_x);  // Frame.cpp:46-47
  const float hInv = (float)ORBX_GRID_ROWS / (float)(mp.b.max_y - mp.b.min_y);
  const float fminX = (float)mp.b.min_x, fminY = (float)mp.b.min_y;
  for (int j = t; j < n2; j += T) {
    md[j] = INF_DIST;
    m21[j] = -1;
    const orbx_keypoint kp = k2[j];
    // Frame::PosInGrid (Frame.cpp:89-99) + the octave filter of GetFeaturesInArea (Frame.cpp:179,191)
    const int px = (int)roundf((kp.x - fminX) * wInv), py = (int)roundf((kp.y - fminY) * hInv);
    const bool ok = kp.octave == 0 && px >= 0 && px < ORBX_GRID_COLS && py >= 0 && py < ORBX_GRID_ROWS;
    cell2[j] = ok ? px * ORBX_GRID_ROWS + py : -1;
  }
  for (int i = t; i < n1; i += T) {
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
    for (int j = t; j < n2; j += T) {
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
      for (int w = 1; w < T / 64; w++) {
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
    for (int i = t; i < n1; i += T) {
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


// -------------------------------------------------------------------------------------------------
// The wide path: the same fixpoint as k_match_jacobi for pairs with up to MW_CAP (4096) octave-0 queries / eligible
// trains (the 1920x1080 / 4000-feature and 3840x2160 / 8000-feature configurations: ~870 and ~1740 per frame), cut into
// three steps so that the expensive part - the window test of every (query, train) and the Hamming distances - is
// spread over the whole GPU instead of one workgroup per pair:
//   matchWidePrep         (above; run by k_match_jacobi's workgroup when it hands the pair on) compaction of the queries
//                         and of the eligible trains into the pair's scratch
//   k_match_wide_lists    one workgroup per 64 queries: lane = query, the four waves split the trains; a train record
//                         and its descriptor are wave-uniform (scalar) loads; every in-window train is appended to the
//                         query's candidate list (dist << 16 | train slot; the order inside a list is irrelevant)
//   k_match_wide_resolve  one workgroup per pair: the sweeps and the final bookkeeping of k_match_jacobi, up to four
//                         queries per thread, lists read from the scratch (L2); the claims of a sweep are one linked
//                         list per train through its claimants (LDS), so any number of queries may claim a train
// Only candidates with a distance below MatchParams::dmax are listed (see launch_match): a farther train can neither be
// accepted nor make a nearer one fail the ratio test, so the lists stay short even when the window covers the frame
// (the brute-force configurations).
// A pair the path cannot take (more than MW_CAP queries or trains, more than MW_CP listed candidates of one query) is
// matched by the same workgroup of k_match_wide_resolve with matchGeneral.
// -------------------------------------------------------------------------------------------------
// matchWidePrep as a kernel of its own: only for a batch whose wide kernels were not issued with it (launch_match,
// wideMode 2) - the scratch of its pending pairs may have been reused by a later batch in the meantime.
__global__ __launch_bounds__(MW_T) void k_match_wide_prep(const int* __restrict__ pairFirst, const int* __restrict__ pairSecond,
                                                         const orbx_keypoint* __restrict__ kps, const int* __restrict__ nkp,
                                                         const MatchParams mp, int* __restrict__ matches12,
                                                         const int* __restrict__ nmatchesOut, int* __restrict__ scratch,
                                                         long long scratchStride, int capl) {
  ORBX_SETPRIO();
  const int pair = blockIdx.x + mp.pair0;
  if (nmatchesOut[pair] != MATCH_PENDING) return;
  matchWidePrep<MW_T>(pair, pairFirst, pairSecond, kps, nkp, mp, matches12, scratch, scratchStride, capl);
}

// NW = waves per workgroup that share the trains of the workgroup's 64 queries: 4 when the launch has workgroups enough to fill the
// chip (64 sets of 2000 x 2000: 2048), 8 when it has not (round 6: sixteen 1080p pairs are 112 workgroups, ONE 2000 x 2000
// match is 32 -- with four waves each the list kernel ran on a fraction of the CUs for 38 and ~16 us)
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_match_wide_lists(const int* __restrict__ pairFirst, const int* __restrict__ pairSecond,
                                                         const orbx_keypoint* __restrict__ kps,
                                                         const uint8_t* __restrict__ desc, const int* __restrict__ nkp,
                                                         const MatchParams mp, const int* __restrict__ nmatchesOut,
                                                         const int* __restrict__ scratchR, int* __restrict__ scratchW,
                                                         long long scratchStride, int capl) {
  __shared__ int cnt[64], anyIn[64];  // per query: candidates listed, and whether its window holds any train at all
  const int t = threadIdx.x, lane = t & 63;
  const int part = __builtin_amdgcn_readfirstlane(t >> 6);
  const int pair = blockIdx.y + mp.pair0;
  if (nmatchesOut[pair] != MATCH_PENDING) return;
  const int* S = scratchR + (long long)pair * scratchStride;  // read side: header [0..2], train records, query indices
  if (S[2]) return;
  const int nQ = S[0], nT = S[1];
  const int q0 = blockIdx.x * 64;
  if (q0 >= nQ) return;
  if ((S[9 + (blockIdx.x >> 5)] >> (blockIdx.x & 31)) & 1) return;  // these 64 queries were listed by k_match_bf_mfma
  const uint4* trec = reinterpret_cast<const uint4*>(S + MW_HDR);
  const int* qIdx = S + MW_HDR + 4 * capl;
  int* W = scratchW + (long long)pair * scratchStride;        // write side: header [3], counts, lists
  int* cntOut = W + MW_HDR + 5 * capl;
  uint32_t* lists = reinterpret_cast<uint32_t*>(W + MW_HDR + 6 * capl);
  const int fa = pairFirst[pair], fb = pairSecond[pair];
  const int cap = mp.capacity;
  const orbx_keypoint* k1 = kps + (long long)fa * cap;
  const uint32_t* d1 = reinterpret_cast<const uint32_t*>(desc + (long long)fa * cap * 32);
  const uint32_t* d2 = reinterpret_cast<const uint32_t*>(desc + (long long)fb * cap * 32);
  const float wInv = (float)ORBX_GRID_COLS / (float)(mp.b.max_x - mp.b.min_x);
  const float hInv = (float)ORBX_GRID_ROWS / (float)(mp.b.max_y - mp.b.min_y);
  const float fminX = (float)mp.b.min_x, fminY = (float)mp.b.min_y;
  if (t < 64) { cnt[t] = 0; anyIn[t] = 0; }
  __syncthreads();
  // lane -> query through the queries' order by grid column (matchWidePrep): the 64 queries of a workgroup are neighbours
  const bool valid = q0 + lane < nQ;
  const bool byCol = S[8] != 0;  // (uniform) trains stored by grid column, queries handed out in column order
  const int q = !valid ? 0 : (byCol ? (S + MW_HDR + (6 + MW_CP + MW_TOPK) * capl)[q0 + lane] : q0 + lane);
  const int* cxStart = S + MW_HDR + (6 + MW_CP + MW_TOPK) * capl + capl;
  float qx = 0.f, qy = 0.f;
  uint32_t qd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (valid) {
    const int qi = qIdx[q];
    qx = k1[qi].x; qy = k1[qi].y;
    const uint4 a0 = reinterpret_cast<const uint4*>(d1 + (long long)qi * 8)[0];
    const uint4 a1 = reinterpret_cast<const uint4*>(d1 + (long long)qi * 8)[1];
    qd[0] = a0.x; qd[1] = a0.y; qd[2] = a0.z; qd[3] = a0.w; qd[4] = a1.x; qd[5] = a1.y; qd[6] = a1.z; qd[7] = a1.w;
  }
  // cell window of my query, Frame.cpp:167-177 (an empty window makes every test below fail)
  const float r = (float)mp.window;
  const int minCX = max(0, (int)floorf((qx - fminX - r) * wInv));
  const int maxCX = min(ORBX_GRID_COLS - 1, (int)ceilf((qx - fminX + r) * wInv));
  const int minCY = max(0, (int)floorf((qy - fminY - r) * hInv));
  const int maxCY = min(ORBX_GRID_ROWS - 1, (int)ceilf((qy - fminY + r) * hInv));
  // the trains lie sorted by grid column: only the columns some window of this workgroup's queries reaches are walked (the cell
  // test of every query stays as it is), an NW-th of that range per wave
  int cLo = valid ? minCX : ORBX_GRID_COLS, cHi = valid ? maxCX : -1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { cLo = min(cLo, __shfl_xor(cLo, o)); cHi = max(cHi, __shfl_xor(cHi, o)); }
  cLo = __builtin_amdgcn_readfirstlane(cLo);
  cHi = __builtin_amdgcn_readfirstlane(cHi);
  const int rLo = !byCol || cHi < cLo ? 0 : cxStart[min(max(cLo, 0), ORBX_GRID_COLS)];
  const int rHi = !byCol ? nT : (cHi < cLo ? 0 : min(nT, cxStart[min(max(cHi + 1, 0), ORBX_GRID_COLS)]));
  const int chunk = (rHi - rLo + NW - 1) / NW;
  // (the walk's bounds in scalar registers: the train records and descriptors below must stay wave-uniform scalar loads)
  const int e0 = __builtin_amdgcn_readfirstlane(rLo + part * chunk), e1 = __builtin_amdgcn_readfirstlane(min(rHi, rLo + part * chunk + chunk));
  bool any = false;
  // Brute force (BASELINE config 5's 2000 x 2000 match: the window covers the frame): when, for every query of this wave, the
  // cell range is the whole grid and both ends of the trains' bounding box lie within r -- float subtraction is monotonic, so
  // then fabsf(tx - qx) < r for EVERY train -- the six comparisons per (query, train) are dropped and the loop is the 16
  // xor / bcnt operations of the Hamming distance plus one compare (k_match_wide_lists 410 -> see profiles us per 64 x 2000^2)
  const float bbx0 = __uint_as_float((uint32_t)S[4]), bbx1 = __uint_as_float((uint32_t)S[5]);
  const float bby0 = __uint_as_float((uint32_t)S[6]), bby1 = __uint_as_float((uint32_t)S[7]);
  const bool lanePass = !valid || (minCX == 0 && maxCX == ORBX_GRID_COLS - 1 && minCY == 0 && maxCY == ORBX_GRID_ROWS - 1 &&
                                   fabsf(bbx0 - qx) < r && fabsf(bbx1 - qx) < r && fabsf(bby0 - qy) < r && fabsf(bby1 - qy) < r);
  const bool allPass = __ballot(!lanePass) == 0ull;  // wave-uniform
  auto scan = [&](auto tag) {
    constexpr bool ALL = decltype(tag)::value;
    for (int eb = e0; eb < e1; eb += 4) {  // four wave-uniform train records in flight
      uint4 rec[4];
#pragma unroll
      for (int j = 0; j < 4; j++) rec[j] = trec[min(eb + j, e1 - 1)];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int e = eb + j;
        if (e >= e1) break;
        bool in = valid;
        if (!ALL) {
          const int cx = (int)(rec[j].z & 0xff), cy = (int)(rec[j].z >> 8);
          const float dx = __uint_as_float(rec[j].x) - qx, dy = __uint_as_float(rec[j].y) - qy;
          in = valid && cx >= minCX && cx <= maxCX && cy >= minCY && cy <= maxCY && fabsf(dx) < r && fabsf(dy) < r;
          if (__ballot(in) == 0ull) continue;  // wave-uniform
        }
        const uint32_t* b = d2 + (long long)(rec[j].w & 0xfffff) * 8;  // wave-uniform address: scalar loads
        const uint4 b0 = reinterpret_cast<const uint4*>(b)[0], b1 = reinterpret_cast<const uint4*>(b)[1];
        if (in) {
          const int dist = __popc(qd[0] ^ b0.x) + __popc(qd[1] ^ b0.y) + __popc(qd[2] ^ b0.z) + __popc(qd[3] ^ b0.w) +
                           __popc(qd[4] ^ b1.x) + __popc(qd[5] ^ b1.y) + __popc(qd[6] ^ b1.z) + __popc(qd[7] ^ b1.w);
          any = true;
          if (dist < mp.dmax) {  // a farther train can neither be accepted nor fail the ratio test of a nearer one
            const int slot = atomicAdd(&cnt[lane], 1);
            if (slot < MW_CP) lists[(size_t)slot * capl + q] = ((uint32_t)dist << 16) | (uint32_t)e;
          }
        }
      }
    }
  };
  // Brute force: the wave stages the descriptors of 64 trains at a time in LDS (lane = train: two 16-byte loads, the next chunk's
  // in flight while this one is matched) and every query lane then reads a train's 256 bits as two LDS broadcasts -- the scalar
  // loads of the general path left the wave waiting for every group of four trains (0.55 of the issue cycles; 217 us per
  // 64 x 2000^2 pairs).  The loop is the 16 xor / bcnt operations, a compare and, rarely, an append.
  __shared__ uint4 stageD[NW][2][64][2];  // [wave][buffer][train][half]
  auto scanAllLds = [&]() {
    uint4 (*st)[64][2] = stageD[part];
    auto fetch = [&](const int eb, uint4& a, uint4& b) {
      a = make_uint4(0, 0, 0, 0); b = a;
      const int e = eb + lane;
      if (e < e1) {
        const uint32_t idx = reinterpret_cast<const uint32_t*>(trec + e)[3] & 0xfffffu;
        const uint4* p = reinterpret_cast<const uint4*>(d2 + (long long)idx * 8);
        a = p[0]; b = p[1];
      }
    };
    uint4 na, nb;
    fetch(e0, na, nb);
    int cur = 0;
    st[0][lane][0] = na; st[0][lane][1] = nb;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int eb = e0; eb < e1; eb += 64) {
      const bool more = eb + 64 < e1;  // (uniform)
      if (more) fetch(eb + 64, na, nb);
      const int cntT = min(64, e1 - eb);
      for (int j0 = 0; j0 < cntT; j0 += 4) {
        uint4 b0[4], b1[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { b0[j] = st[cur][min(j0 + j, 63)][0]; b1[j] = st[cur][min(j0 + j, 63)][1]; }
        // v_bcnt_u32_b32 adds its count to an accumulator: eight of them in a chain per train, the four trains' chains
        // interleaved (the compiler builds a tree of counts and three-operand adds instead: three more 4-cycle instructions a train)
        int dd[4] = {0, 0, 0, 0};
#define ORBX_BCNT4(QW, F)                                                                                          \
  _Pragma("unroll") for (int j = 0; j < 4; j++)                                                                    \
      asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(dd[j]) : "v"(qd[QW] ^ F), "v"(dd[j]));
        ORBX_BCNT4(0, b0[j].x) ORBX_BCNT4(1, b0[j].y) ORBX_BCNT4(2, b0[j].z) ORBX_BCNT4(3, b0[j].w)
        ORBX_BCNT4(4, b1[j].x) ORBX_BCNT4(5, b1[j].y) ORBX_BCNT4(6, b1[j].z) ORBX_BCNT4(7, b1[j].w)
#undef ORBX_BCNT4
        // (round 5) one test for the four trains: a random descriptor lies ~128 bits away and dmax is ~56, so a lane finds a
        // candidate among four trains once in hundreds of steps -- four compares and exec-mask branches per step became one
        const int dmin = min(min(dd[0], dd[1]), min(dd[2], dd[3]));
        if (__ballot(valid && dmin < mp.dmax) != 0ull) {  // (wave-uniform)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (j0 + j >= cntT) break;  // (uniform)
            const int dist = dd[j];
            if (valid && dist < mp.dmax) {  // a farther train can neither be accepted nor fail the ratio test of a nearer one
              const int slot = atomicAdd(&cnt[lane], 1);
              if (slot < MW_CP) lists[(size_t)slot * capl + q] = ((uint32_t)dist << 16) | (uint32_t)(eb + j0 + j);
            }
          }
        }
      }
      if (valid && cntT > 0) any = true;
      if (more) {
        __builtin_amdgcn_wave_barrier();  // (this chunk's LDS reads are issued before the other buffer is overwritten: it is the other one)
        st[cur ^ 1][lane][0] = na; st[cur ^ 1][lane][1] = nb;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        cur ^= 1;
      }
    }
  };
  if (allPass) scanAllLds();
  else scan(std::false_type{});
  if (any) anyIn[lane] = 1;  // vIndices2 of the query is not empty (the four parts may all store the same 1)
  __syncthreads();
  if (t < 64 && valid) {
    const int c = cnt[t];
    if (c > MW_CP) atomicOr(&W[3], 1);
    cntOut[q] = min(c, MW_CP) | (anyIn[t] << 16);
  }
}

// -------------------------------------------------------------------------------------------------
// k_match_bf_mfma: the BRUTE-FORCE case of k_match_wide_lists on the matrix cores (round 5).  When the window of every query of a
// block covers every train (BASELINE config 5's 2000 x 2000 match, config 3's all-pairs match), the work is all pairs of
// 256-bit Hamming distances and one compare -- a dense contraction over the bits: with the bits as +-1 bytes (pm1Bytes16) the
// dot product of two descriptors is 256 - 2 x distance, so "distance < dmax" is "accumulator > 256 - 2 dmax", one threshold.
//   * workgroup = 256 queries, eight waves of 32 (one 32-row fragment x 8 k-steps = 32 registers, built once; BF_NT = 2: four waves
//     of 64 -- half the waves per SIMD, 2.50 against 2.30 us per 2000 x 2000: the kernel is bound by a wave's own chain of LDS reads,
//     MFMAs, reduction, appends and barrier, ~2800 cycles per tile, so more and lighter waves are what helps);
//   * the trains go by in tiles of 32: thread (train t >> 3, dword t & 7) loads one dword of a train's descriptor (the record two
//     tiles ahead, the dword one tile ahead) and expands it into the tile's LDS image (two 16-byte stores), laid out so that a
//     wave reads the B fragment of k-step c as 64 consecutive 16-byte pieces; two images, one LDS-only barrier per tile;
//   * per tile and wave 8 v_mfma_i32_32x32x32_i8 (32 x 32 distances), then the maxima of four consecutive query rows, their
//     maximum and one compare;
//   * a candidate -- about one per wave and tile in the synthetic sets, one in thousands of pairs -- is appended from the lane that
//     holds its accumulator (column = train, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) to the lists k_match_wide_lists fills
//     (slots through the queries' counters in LDS); only a group of four rows with a hit looks at its accumulators again;
//   * both operands take their k order from the same function of (lane >> 5, byte), so the instruction's own k map is irrelevant.
// The vector form costs 16 xor / bcnt per 64 pairs (1.1 cycles of a SIMD per pair), the matrix form 0.25 (16 MFMAs of 32 cycles
// per 2048 pairs).  Measured per 64 sets of 2000 x 2000: 81 us (with 64 queries per wave: 96) against 160 for the vector form; the
// MFMAs with their LDS reads and the barrier alone would take 31.5 us -- the int8 rate (tools/exp_bf_parts.sh: timing-only builds
// without one part; with the reduction 55, the staging 65, the appends 100 at 64 queries per wave).  Cycle stamps
// (tools/bf_stamps.py) show why the parts add up: a wave needs ~2800 cycles per tile whatever shares its CU -- LDS reads + MFMA
// issue 800, drain + reduction 290, appends 750 (a hit every second tile, 1400 cycles of compare -> scalar test -> branch round
// trips), staging 420, barrier 670 (the wave that took the append path is waited for) -- so the kernel is bound by the number of
// waves in flight: hence the light waves.  Tried against the chain itself, none faster (docs/history.md): the MFMAs of tile
// i + 1 issued before the vector work on tile i (two accumulator sets), per-wave hit buffers filled without atomics, lane masks
// instead of maxima, a rolled hit loop over indexed registers, scalar hit handling by v_readlane, all B fragments read ahead, the
// vector work interleaved into the MFMAs by sched_group_barrier (280 registers: one wave per SIMD), unequal wave priorities.
// A block whose queries do not all pass the brute-force test, and a launch with too few blocks to fill the chip, is left to
// k_match_wide_lists (header [9..10]: one bit per 64 queries listed here).
// Built for candidates that are one pair in thousands.  Where they come by the hundred per query (BASELINE config 3's synthetic
// frames repeat their corners: every ninth pair is nearer than dmax) the appends dominate and the vector form is faster -- 64 such
// pairs of 1080p frames in one launch (measured with four waves of 64 queries and the rule at 128 blocks): 0.47 against 0.32 ms for the
// matching stage, 54.7 k against 56.2 k frames/s
// (tools/exp_c3_dense.py; config 3 as benchmarked, 16 pairs per call, stays below the 256 blocks).  Giving such blocks back was tried
// (a wave counting its first tiles' candidates: the first tile says nothing, those frames' candidates sit further down the train
// list; counting while listing cost the sparse case 7 %) and left out.
// -------------------------------------------------------------------------------------------------
#ifndef ORBX_BF_EXP
#define ORBX_BF_EXP 0  // diagnostic builds of k_match_bf_mfma without one of its parts (timing only; tools/exp_bf_parts.sh)
#endif
// Diagnostic build only (-DORBX_BF_STAMPS): per wave of k_match_bf_mfma, s_memtime deltas of a tile's phases summed over its tiles
// (tools/bf_stamps.py prints the shares): 0 LDS reads + MFMA issue, 1 drain + reduction, 2 appends, 3 staging, 4 barrier.
#ifdef ORBX_BF_STAMPS
__device__ uint32_t g_bfStamps[4096 * 8];
#define BF_STAMP(k)                                                        \
  do {                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                     \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();          \
    bfAcc_[k] += (uint32_t)(now_ - bfPrev_);                               \
    bfPrev_ = now_;                                                        \
    __builtin_amdgcn_sched_barrier(0);                                     \
  } while (0)
extern "C" int orbx_diag_bf_stamps(uint32_t* out, int nWaves) {
  if (nWaves < 0) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_bfStamps)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(uint32_t) * 8 * 4096);
  }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bfStamps), sizeof(uint32_t) * 8 * (size_t)nWaves);
}
#else
#define BF_STAMP(k) do { } while (0)
#endif
#ifndef BF_NT
#define BF_NT 1  // 32-row query fragments per wave: 1 = 32 queries per wave, eight waves per workgroup; 2 = 64 queries, four waves
#endif
#ifndef BF_WAVES
#define BF_WAVES (8 / BF_NT)  // waves per workgroup
#endif
#define BF_T (64 * BF_WAVES)
#define BF_QWG (32 * BF_NT * BF_WAVES)  // queries per workgroup (a multiple of 64)
static_assert(BF_T == 256 || BF_T == 512, "k_match_bf_mfma: the staging maps 512 half dwords of a tile onto 256 or 512 threads");
__global__ __launch_bounds__(BF_T) void k_match_bf_mfma(const int* __restrict__ pairFirst, const int* __restrict__ pairSecond,
                                                      const orbx_keypoint* __restrict__ kps,
                                                      const uint8_t* __restrict__ desc, const int* __restrict__ nkp,
                                                      const MatchParams mp, const int* __restrict__ nmatchesOut,
                                                      const int* __restrict__ scratchR, int* __restrict__ scratchW,
                                                      long long scratchStride, int capl, unsigned int* __restrict__ diag) {
  // [image][k-step = dword][bit half][train]: 2 x 8 KB; rows of 33 pieces, so that the sixteen (train, dword) stores of a quarter
  // wave fall into sixteen different groups of four banks (with 32 the eight dwords of a train share one)
  __shared__ __attribute__((aligned(16))) v4i_t tileB[2][8][2][33];
  __shared__ uint32_t qdS[BF_QWG][9];
  __shared__ int cnt[BF_QWG];
  __shared__ int sAll;
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int pair = blockIdx.y + mp.pair0;
  if (nmatchesOut[pair] != MATCH_PENDING) return;
  const int* S = scratchR + (long long)pair * scratchStride;
  if (S[2] || S[8]) return;  // (trains stored by grid column: the windows are narrow, no block can be brute force)
  const int nQ = S[0], nT = S[1];
  const int Q0 = blockIdx.x * BF_QWG;
  if (Q0 >= nQ || nT <= 0) return;
  // (a launch whose pairs have too few 256-query blocks to fill the chip with stays on k_match_wide_lists, 64 queries x a quarter of
  // the trains per wave: 1080p frame pairs, 869 octave-0 queries each, matched four or eight pairs per call -- 0.24 ms there, 0.36 here;
  // 2000 x 2000 sets: 16 per call 6.5 us per pair of sets there, 8.3 here; 32 per call 4.3 there, 3.8 here)
  if ((int)gridDim.y * ((nQ + 255) >> 8) < 256) return;  // (counted in blocks of 256 queries whatever the workgroup's size)
  const int nTl = (nT + 31) >> 5;
  const uint4* trec = reinterpret_cast<const uint4*>(S + MW_HDR);
  const int* qIdx = S + MW_HDR + 4 * capl;
  int* W = scratchW + (long long)pair * scratchStride;
  int* cntOut = W + MW_HDR + 5 * capl;
  uint32_t* lists = reinterpret_cast<uint32_t*>(W + MW_HDR + 6 * capl);
  const int fa = pairFirst[pair], fb = pairSecond[pair];
  const int cap = mp.capacity;
  const orbx_keypoint* k1 = kps + (long long)fa * cap;
  const uint32_t* d1 = reinterpret_cast<const uint32_t*>(desc + (long long)fa * cap * 32);
  const uint32_t* d2 = reinterpret_cast<const uint32_t*>(desc + (long long)fb * cap * 32);
  // ---- thread = query: the brute-force test of k_match_wide_lists (its comment there), for the whole block ----
  const int q = Q0 + t;
  const bool valid = t < BF_QWG && q < nQ;
  {
    const float wInv = (float)ORBX_GRID_COLS / (float)(mp.b.max_x - mp.b.min_x);
    const float hInv = (float)ORBX_GRID_ROWS / (float)(mp.b.max_y - mp.b.min_y);
    const float fminX = (float)mp.b.min_x, fminY = (float)mp.b.min_y;
    float qx = 0.f, qy = 0.f;
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
    if (valid) {
      const int qi = qIdx[q];
      qx = k1[qi].x; qy = k1[qi].y;
      a0 = reinterpret_cast<const uint4*>(d1 + (long long)qi * 8)[0];
      a1 = reinterpret_cast<const uint4*>(d1 + (long long)qi * 8)[1];
    }
    const float r = (float)mp.window;
    const int minCX = max(0, (int)floorf((qx - fminX - r) * wInv));
    const int maxCX = min(ORBX_GRID_COLS - 1, (int)ceilf((qx - fminX + r) * wInv));
    const int minCY = max(0, (int)floorf((qy - fminY - r) * hInv));
    const int maxCY = min(ORBX_GRID_ROWS - 1, (int)ceilf((qy - fminY + r) * hInv));
    const float bbx0 = __uint_as_float((uint32_t)S[4]), bbx1 = __uint_as_float((uint32_t)S[5]);
    const float bby0 = __uint_as_float((uint32_t)S[6]), bby1 = __uint_as_float((uint32_t)S[7]);
    const bool lanePass = !valid || (minCX == 0 && maxCX == ORBX_GRID_COLS - 1 && minCY == 0 && maxCY == ORBX_GRID_ROWS - 1 &&
                                     fabsf(bbx0 - qx) < r && fabsf(bbx1 - qx) < r && fabsf(bby0 - qy) < r && fabsf(bby1 - qy) < r);
    if (t == 0) sAll = 1;
    __syncthreads();
    if (!lanePass) sAll = 0;
    if (t < BF_QWG) {
      qdS[t][0] = a0.x; qdS[t][1] = a0.y; qdS[t][2] = a0.z; qdS[t][3] = a0.w;
      qdS[t][4] = a1.x; qdS[t][5] = a1.y; qdS[t][6] = a1.z; qdS[t][7] = a1.w;
      cnt[t] = 0;
    }
    __syncthreads();
    if (!sAll) return;  // (uniform) k_match_wide_lists takes the block
  }
  if (t == 0 && diag) atomicAdd(diag, 1u);  // orbx_debug_match_counters: blocks of 256 queries this kernel has listed
  // ---- the queries' fragments: rows = query 64 wv + 32 T + (lane & 31), k = the 16 bits (lane >> 5) of dword c ----
  const int h = lane >> 5, n = lane & 31;
  const int qw = 32 * BF_NT * wv;  // the wave's first query of the block
  const bool waveLive = Q0 + qw < nQ;  // (uniform) a wave without a query only helps to stage the trains
  v4i_t aq[BF_NT][8];
#pragma unroll
  for (int T = 0; T < BF_NT; T++)
#pragma unroll
    for (int c = 0; c < 8; c++) aq[T][c] = pm1Bytes16(qdS[qw + 32 * T + n][c] >> (16 * h));
  const int thr = 256 - 2 * mp.dmax;
  // ---- staging: thread = (train j of the tile, dword c).  A loaded value is first TOUCHED a tile later (the record word is masked
  //      where it is used, not where it is loaded) and both loads are unconditional (a slot beyond the last train reads the last
  //      train again; the appends mask it): a wait right behind a load, or the register copy a conditional load ends in, costs
  //      every tile a memory round trip ----
#if BF_T == 256
  const int sj = t >> 3, sc = t & 7;
#else  // (eight waves: a thread expands one half of a dword)
  const int sj = t >> 4, sc = (t >> 1) & 7, sh = t & 1;
#endif
  auto loadIdx = [&](const int tile) -> uint32_t { return reinterpret_cast<const uint32_t*>(trec + min(32 * tile + sj, nT - 1))[3]; };
  auto loadDw = [&](const uint32_t rec) -> uint32_t { return d2[(long long)(rec & 0xfffffu) * 8 + sc]; };
  auto expand = [&](const int img, const uint32_t w) {
#if BF_T == 256
    tileB[img][sc][0][sj] = pm1Bytes16(w);
    tileB[img][sc][1][sj] = pm1Bytes16(w >> 16);
#else
    tileB[img][sc][sh][sj] = pm1Bytes16(w >> (16 * sh));
#endif
  };
  auto ldsBarrier = [&]() {  // (over the LDS images only: __syncthreads() would also wait for the global loads in flight)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  };
  // image 0 = tile 0; wN = this thread's dword of tile 1, idxN its record of tile 2
  uint32_t idxN = loadIdx(0);
  uint32_t wN = loadDw(idxN);
  idxN = loadIdx(1);
  expand(0, wN);
  wN = loadDw(idxN);
  idxN = loadIdx(2);
  ldsBarrier();
#ifdef ORBX_BF_STAMPS
  unsigned long long bfPrev_ = __builtin_amdgcn_s_memtime();
  uint32_t bfAcc_[5] = {0u, 0u, 0u, 0u, 0u};
#endif
  for (int i = 0; i < nTl; i++) {
    const int img = i & 1;
    if (waveLive) {
      v16i_t acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
#pragma unroll
      for (int c = 0; c < 8; c++) {
#if !(ORBX_BF_EXP & 2)  // (2: TIMING ONLY, no matrix instructions)
        const v4i_t bt = tileB[img][c][h][n];
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq[0][c], bt, acc0, 0, 0, 0);
#if BF_NT == 2
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq[1][c], bt, acc1, 0, 0, 0);
#endif
#endif
      }
      BF_STAMP(0);
#if !(ORBX_BF_EXP & 8)  // (8: TIMING ONLY, no reduction)
      int gm[2][4];
#pragma unroll
      for (int T = 0; T < 2; T++)
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
          const v16i_t& A = T ? acc1 : acc0;
          gm[T][gq] = T < BF_NT ? max(max(A[4 * gq], A[4 * gq + 1]), max(A[4 * gq + 2], A[4 * gq + 3])) : (int)0x80000000;
        }
      const int m = max(max(max(gm[0][0], gm[0][1]), max(gm[0][2], gm[0][3])), max(max(gm[1][0], gm[1][1]), max(gm[1][2], gm[1][3])));
      const int e = 32 * i + n;
      const bool tOk = e < nT;
      asm volatile("" :: "v"(m));
      BF_STAMP(1);
#if ORBX_BF_EXP & 1  // (1: TIMING ONLY, no appends)
      if (m == 0x7fffffff)
#endif
      if (__ballot(tOk && m > thr) != 0ull) {  // (wave-uniform) some pair of the tile is nearer than dmax
#pragma unroll
        for (int T = 0; T < BF_NT; T++)
#pragma unroll
          for (int gq = 0; gq < 4; gq++) {
            if (__ballot(tOk && gm[T][gq] > thr) == 0ull) continue;  // (wave-uniform)
            const v16i_t& A = T ? acc1 : acc0;
#pragma unroll
            for (int r = 4 * gq; r < 4 * gq + 4; r++) {
              const int a = A[r];
              const int ql = qw + 32 * T + (r & 3) + 8 * (r >> 2) + 4 * h;
              if (tOk && a > thr && Q0 + ql < nQ) {
                const int slot = atomicAdd(&cnt[ql], 1);
                if (slot < MW_CP) lists[(size_t)slot * capl + Q0 + ql] = ((uint32_t)((256 - a) >> 1) << 16) | (uint32_t)e;
              }
            }
          }
      }
#else
      if (acc0[0] == 0x7fffffff) cnt[0] = 1;
#endif
      BF_STAMP(2);
    }
#if !(ORBX_BF_EXP & 4)  // (4: TIMING ONLY, no staging)
    if (i + 1 < nTl) {  // (uniform) the next tile into the other image: its last readers passed the barrier of the previous step
      expand(img ^ 1, wN);
      wN = loadDw(idxN);
      idxN = loadIdx(i + 3);
    }
#endif
    BF_STAMP(3);
    ldsBarrier();
    BF_STAMP(4);
  }
#ifdef ORBX_BF_STAMPS
  {
    const unsigned wid_ = ((blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv) & 4095u;
    if (lane == 0) {
      for (int j = 0; j < 5; j++) g_bfStamps[wid_ * 8 + j] = bfAcc_[j];
      g_bfStamps[wid_ * 8 + 5] = (uint32_t)nTl;
    }
  }
#endif
  __syncthreads();
  if (valid) {
    const int c = cnt[t];
    if (c > MW_CP) atomicOr(&W[3], 1);
    cntOut[q] = min(c, MW_CP) | (1 << 16);  // (bit 16: vIndices2 of the query is not empty -- nT > 0 here)
  }
  if (t < BF_QWG / 64 && Q0 + 64 * t < nQ) {
    const int blk = (BF_QWG / 64) * (int)blockIdx.x + t;  // the 64-query block of k_match_wide_lists
    atomicOr(&W[9 + (blk >> 5)], 1 << (blk & 31));
  }
}

// k_match_wide_sort: every query's candidate list (k_match_wide_lists: at most MW_CP entries in no particular order) sorted by
// the comparison key (distance, reference candidate order = cell << 20 | F2 index), a wave per query (sixteen queries per
// workgroup), over the whole GPU.  k_match_wide_resolve then walks a list from its head and stops at the second candidate that no
// earlier query hides: a sweep touches a few entries per query instead of all ~100 (the synthetic 1080p scene repeats its corners:
// half of the queries found fewer than two visible candidates among their four best and re-read their whole list in every sweep
// -- 140 us for eight pairs on eight CUs).
static_assert(MW_CP == 128, "k_match_wide_sort: two list entries per lane");
__global__ __launch_bounds__(256) void k_match_wide_sort(const MatchParams mp, const int* __restrict__ nmatchesOut, int* __restrict__ scratch,
                                                        long long scratchStride, int capl, int qpw) {
  ORBX_SETPRIO();
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.y + mp.pair0;
  if (nmatchesOut[pair] != MATCH_PENDING) return;
  int* S = scratch + (long long)pair * scratchStride;
  if (S[2] | S[3]) return;  // (the reference's loop takes the pair: k_match_wide_resolve)
  const uint4* trec = reinterpret_cast<const uint4*>(S + MW_HDR);
  const int nQ = S[0];
  // qpw queries per wave (launch_match: 4, or 16 in a large launch -- sparse lists leave most waves nothing to sort, and 8000
  // workgroups that only look and leave cost 12 us per 64 sets of 2000): the lanes look at the wave's counts together, the wave
  // then visits the lists with two entries or more
  const int qBase = ((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6)) * qpw;
  if (qBase >= nQ) return;
  const int ncMine = lane < qpw && qBase + lane < nQ ? (S[MW_HDR + 5 * capl + qBase + lane] & 0xffff) : 0;
  unsigned long long todo = __ballot(ncMine >= 2);
  while (todo != 0ull) {
  const int qi = (int)__builtin_ctzll(todo);
  todo &= todo - 1ull;
  const int q = qBase + qi;
  const int nc = __builtin_amdgcn_readlane(ncMine, qi);
  uint32_t* myList = reinterpret_cast<uint32_t*>(S + MW_HDR + 6 * capl) + q;
  unsigned long long v[2];
#pragma unroll
  for (int e = 0; e < 2; e++) {
    const int k = 2 * lane + e;
    v[e] = ~0ull;
    if (k < nc) {
      const uint32_t ce = myList[(size_t)k * capl];
      const uint32_t slot = ce & 0xffffu;
      v[e] = ((unsigned long long)(ce >> 16) << 44) | ((unsigned long long)trec[slot].w << 12) | (unsigned long long)slot;
    }
  }
  static_assert(MW_CAP <= 4096, "twelve slot bits in k_match_wide_sort's keys");
  // bitonic sort of 128 keys, two per lane (lane t owns 2 t, 2 t + 1): partner lanes by shuffle, the pair in registers
  for (int k = 2; k <= 128; k <<= 1) {
    for (int j = k >> 1; j >= 2; j >>= 1) {
      const int lm = j >> 1;
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const int i = 2 * lane + e;
        const unsigned long long o = ((unsigned long long)__shfl_xor((uint32_t)(v[e] >> 32), lm) << 32) | __shfl_xor((uint32_t)v[e], lm);
        const bool keepMin = ((i & j) == 0) == ((i & k) == 0);
        v[e] = keepMin ? (o < v[e] ? o : v[e]) : (o > v[e] ? o : v[e]);
      }
    }
    const bool asc = ((2 * lane) & k) == 0;
    const unsigned long long x = v[0], y = v[1];
    const bool sw = (x > y) == asc;
    v[0] = sw ? y : x;
    v[1] = sw ? x : y;
  }
#pragma unroll
  for (int e = 0; e < 2; e++) {
    const int k = 2 * lane + e;
    if (k < nc) myList[(size_t)k * capl] = ((uint32_t)(v[e] >> 44) << 16) | (uint32_t)(v[e] & 0xfffu);
  }
  }
}

__global__ __launch_bounds__(MW_T) void k_match_wide_resolve(const int* __restrict__ pairFirst, const int* __restrict__ pairSecond,
                                                            const orbx_keypoint* __restrict__ kps,
                                                            const uint8_t* __restrict__ desc, const int* __restrict__ nkp,
                                                            const MatchParams mp, int* __restrict__ matches12,
                                                            int* __restrict__ nmatchesOut, int* __restrict__ statsOut,
                                                            int* scratch, long long scratchStride, int capl) {
  ORBX_SETPRIO();
  extern __shared__ __align__(16) unsigned char mwLds[];
  // claims of a sweep = one linked list per train through its claimant queries (any number of claimants)
  int* head = reinterpret_cast<int*>(mwLds);                              // [capl] a claimant of the train, -1 = none; lastQ at the end
  uint32_t* tOrd = reinterpret_cast<uint32_t*>(mwLds) + capl;             // [capl] cell << 20 | F2 index
  uint16_t* nextQ = reinterpret_cast<uint16_t*>(mwLds + (size_t)capl * 8);  // [capl] next claimant of the same train, MW_END = none
  uint16_t* outD = nextQ + capl;                                          // [capl] distance of the query's claim
  __shared__ int hist[HISTO_LENGTH];
  __shared__ int sNm, sBadDist, sBadRatio, sBadOri, sKeep[3];

  const int t = threadIdx.x;
  const int pair = blockIdx.x + mp.pair0;
  if (nmatchesOut[pair] != MATCH_PENDING) return;
  const int* S = scratch + (long long)pair * scratchStride;
  if (S[2] | S[3]) {  // too large or a full list (block-uniform): the reference's loop as written
    if (!mp.noGeneral)
      matchGeneral<MW_T>(pair, pairFirst, pairSecond, kps, desc, nkp, mp, matches12, nmatchesOut, statsOut, scratch, scratchStride);
    return;
  }
  const int nQ = S[0], nT = S[1];
  const uint4* trec = reinterpret_cast<const uint4*>(S + MW_HDR);
  const int* qIdx = S + MW_HDR + 4 * capl;
  const int* cntIn = S + MW_HDR + 5 * capl;
  const uint32_t* lists = reinterpret_cast<const uint32_t*>(S + MW_HDR + 6 * capl);
  const int fa = pairFirst[pair], fb = pairSecond[pair];
  const int cap = mp.capacity;
  const orbx_keypoint* k1 = kps + (long long)fa * cap;
  const orbx_keypoint* k2 = kps + (long long)fb * cap;
  int* m12 = matches12 + (long long)pair * cap;

  if (t == 0) { sNm = 0; sBadDist = 0; sBadRatio = 0; sBadOri = 0; }
  if (t < HISTO_LENGTH) hist[t] = 0;
  for (int e = t; e < nT; e += MW_T) { tOrd[e] = trec[e].w; head[e] = -1; }
  int nCand[MW_R], outcome[MW_R], bestT[MW_R], bestD[MW_R];
  bool hasCand[MW_R];
#pragma unroll
  for (int rr = 0; rr < MW_R; rr++) {
    const int q = t + rr * MW_T;
    const int c = q < nQ ? cntIn[q] : 0;
    nCand[rr] = c & 0xffff;
    hasCand[rr] = c != 0;
    outcome[rr] = 0; bestT[rr] = -1; bestD[rr] = 0;
  }
  __syncthreads();
  // outcome of a query: 0 = no candidate in the window, 1 = invalid by distance, 2 = invalid by ratio, 3 = accepted
  bool converged = false;
  MJ_STAMP(0);
#ifdef ORBX_MJ_STAMPS
  int sweepsDone_ = 0;
#endif
  for (int sweep = 0; sweep < MW_SWEEPS; sweep++) {
#ifdef ORBX_MJ_STAMPS
    sweepsDone_ = sweep + 1;
    if (sweep == 1) MJ_STAMP(1);
#endif
    bool changed = false;
#pragma unroll
    for (int rr = 0; rr < MW_R; rr++) {
      const int q = t + rr * MW_T;
      const int nc = nCand[rr];
      if (!hasCand[rr]) continue;  // vIndices2.empty() -> continue (ORBmatcher.cpp:46-47): outcome stays 0
      const uint32_t* myList = lists + q;
      // vMatchedDistance[e] as query q sees it: smallest distance of an earlier accepted query that chose e
      auto hiddenBelow = [&](int hd) {
        int md = INF_DIST;
        for (int c = hd; c >= 0;) {
          if (c < q) md = min(md, (int)outD[c]);
          const int nx = nextQ[c];
          c = nx == MW_END ? -1 : nx;
        }
        return md;
      };
      // the list is sorted by (distance, candidate order) (k_match_wide_sort): the first candidate no earlier query hides is the
      // best one, the next such carries the second-smallest distance -- nothing behind it matters
      int found = 0, bd = 0, bt = 0, second = INF_DIST;
      for (int k = 0; k < nc && found < 2; k += 4) {
        uint32_t ce[4];
        int hd[4];
#pragma unroll
        for (int j = 0; j < 4; j++) ce[j] = myList[(size_t)min(k + j, nc - 1) * capl];
#pragma unroll
        for (int j = 0; j < 4; j++) hd[j] = head[ce[j] & 0xffff];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (k + j >= nc || found == 2) break;
          const int e = ce[j] & 0xffff, dist = (int)(ce[j] >> 16);
          if (hiddenBelow(hd[j]) <= dist) continue;  // ORBmatcher.cpp:67
          if (found == 0) { bd = dist; bt = e; } else second = dist;
          found++;
        }
      }
      int nOutcome, nBestT = -1, nBestD = 0;
      if (found == 0 || bd > TH_LOW) nOutcome = 1;
      else if ((float)bd > mp.nnratio * (float)second) nOutcome = 2;
      else { nOutcome = 3; nBestT = bt; nBestD = bd; }
      changed |= nOutcome != outcome[rr] || nBestT != bestT[rr] || nBestD != bestD[rr];
      outcome[rr] = nOutcome; bestT[rr] = nBestT; bestD[rr] = nBestD;
    }
    if (!__syncthreads_or(changed ? 1 : 0)) { converged = true; break; }  // also: every scan of the sweep is done
    for (int e = t; e < nT; e += MW_T) head[e] = -1;
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < MW_R; rr++) {
      if (outcome[rr] == 3) {
        const int q = t + rr * MW_T;
        outD[q] = (uint16_t)bestD[rr];
        const int prev = atomicExch(&head[bestT[rr]], q);
        nextQ[q] = prev < 0 ? (uint16_t)MW_END : (uint16_t)prev;
      }
    }
    __syncthreads();
  }
  if (!converged) {  // block-uniform; cannot happen for nQ <= MW_SWEEPS (after sweep i the first i queries are final)
    if (!mp.noGeneral)
      matchGeneral<MW_T>(pair, pairFirst, pairSecond, kps, desc, nkp, mp, matches12, nmatchesOut, statsOut, scratch, scratchStride);
    return;
  }
  MJ_STAMP(2);
#ifdef ORBX_MJ_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 1024) g_mjStamps[blockIdx.x * 16 + 8] = (unsigned long long)sweepsDone_;
#endif
  // ---- final bookkeeping from the converged outcomes: a train belongs to its LAST claimant ----
  int* lastQ = head;
  for (int e = t; e < nT; e += MW_T) lastQ[e] = -1;
  __syncthreads();
  int bin[MW_R], nBadDist = 0, nBadRatio = 0;
#pragma unroll
  for (int rr = 0; rr < MW_R; rr++) {
    const int q = t + rr * MW_T;
    bin[rr] = -1;
    if (outcome[rr] == 3) atomicMax(&lastQ[bestT[rr]], q);
    nBadDist += outcome[rr] == 1;
    nBadRatio += outcome[rr] == 2;
    if (outcome[rr] == 3 && mp.checkOri) {
      float rot = k1[qIdx[q]].angle - k2[tOrd[bestT[rr]] & 0xfffff].angle;
      if (rot < 0.0f) rot += 360.0f;
      int b = (int)roundf(rot * (HISTO_LENGTH / 360.0f));
      if (b == HISTO_LENGTH) b = 0;
      if (b < 0 || b >= HISTO_LENGTH) b = -1;
      if (b >= 0) atomicAdd(&hist[b], 1);
      bin[rr] = b;
    }
  }
  if (nBadDist) atomicAdd(&sBadDist, nBadDist);
  if (nBadRatio) atomicAdd(&sBadRatio, nBadRatio);
  __syncthreads();
  // nmatches before pruning = trains that ended up with a claimant (every steal took one match away again)
  int owned = 0;
  for (int e = t; e < nT; e += MW_T) owned += lastQ[e] >= 0;
  if (owned) atomicAdd(&sNm, owned);
#pragma unroll
  for (int rr = 0; rr < MW_R; rr++) {
    const int q = t + rr * MW_T;
    if (outcome[rr] == 3 && lastQ[bestT[rr]] == q) m12[qIdx[q]] = (int)(tOrd[bestT[rr]] & 0xfffff);
  }
  if (mp.checkOri) {
    if (t == 0) {  // ComputeThreeMaxima, ORBmatcher.cpp:152-183
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
    // every accepted query sits in rotHist, also one whose match was stolen later (double decrement, :130-138)
    int dropped = 0;
#pragma unroll
    for (int rr = 0; rr < MW_R; rr++) {
      if (bin[rr] >= 0 && bin[rr] != sKeep[0] && bin[rr] != sKeep[1] && bin[rr] != sKeep[2]) {
        m12[qIdx[t + rr * MW_T]] = -1;
        dropped++;
      }
    }
    if (dropped) { atomicSub(&sNm, dropped); atomicAdd(&sBadOri, dropped); }
  }
  __syncthreads();
  if (t == 0) {
    nmatchesOut[pair] = sNm;
    if (statsOut) { statsOut[pair * 3] = sBadDist; statsOut[pair * 3 + 1] = sBadRatio; statsOut[pair * 3 + 2] = sBadOri; }
  }
  MJ_STAMP(3);
}

// =================================================================================================
// Keypoint undistortion: Frame::UndistortKeyPoints (SlamTypes/Frame.cpp:136-161) = cv::undistortPoints with R = I, P = K
// (SURVEY appendix A8).  One thread per keypoint, f64 arithmetic in OpenCV's operation order (the build has
// -ffp-contract=off, so no fused multiply-adds), 5 fixed iterations, result stored as f32.  Every other field of the
// keypoint is copied (Frame.cpp:155-160: `kp = mvKeys[i]; kp.pt = ...`).
// =================================================================================================
__device__ __forceinline__ void undistortPoint(const CamD& c, float xin, float yin, float* xo, float* yo) {
  double x = xin, y = yin;
  const double u = x, v = y;
  x = (x - c.cx) * c.ifx;
  y = (y - c.cy) * c.ify;
  const double x0 = x, y0 = y;
#pragma unroll 1
  for (int j = 0; j < 5; j++) {
    const double r2 = x * x + y * y;
    // numerator 1 + ((k[7]*r2 + k[6])*r2 + k[5])*r2 with k[5..7] = 0; denominator with k[4] = 0
    const double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((0. * r2 + c.k1) * r2 + c.k0) * r2);
    if (icdist < 0) {
      x = (u - c.cx) * c.ifx;
      y = (v - c.cy) * c.ify;
      break;
    }
    const double deltaX = 2 * c.k2 * x * y + c.k3 * (r2 + 2 * x * x) + 0. * r2 + 0. * r2 * r2;
    const double deltaY = c.k2 * (r2 + 2 * y * y) + 2 * c.k3 * x * y + 0. * r2 + 0. * r2 * r2;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  const double xx = c.fx * x + 0. * y + c.cx, yy = 0. * x + c.fy * y + c.cy, ww = 1. / (0. * x + 0. * y + 1.);
  *xo = (float)(xx * ww);
  *yo = (float)(yy * ww);
}

__global__ __launch_bounds__(256) void k_undistort(const orbx_keypoint* __restrict__ in, const int* __restrict__ nkp, int capacity,
                                                   CamD c, orbx_keypoint* __restrict__ out) {
  const int f = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = min(nkp[f], capacity);
  if (i >= n) return;
  orbx_keypoint kp = in[(size_t)f * capacity + i];
  if (c.distorted) undistortPoint(c, kp.x, kp.y, &kp.x, &kp.y);
  out[(size_t)f * capacity + i] = kp;
}

// =================================================================================================
// Model scoring of the Initializer: CheckHomography / CheckFundamental (Initialization/Initializer.cpp:268-438), one wave
// per RANSAC hypothesis.  The per-match terms are independent (lanes); the score is a sequential f32 sum in match
// order, so lane 0 adds the 128 terms of each step of 64 matches one after the other (a skipped term is an exact + 0.0f).
// f32 arithmetic as written upstream, uncontracted; `1.0 / x` in double, rounded to float; f32 division correctly
// rounded (build flags).
// =================================================================================================
__global__ __launch_bounds__(64) void k_check_model(const ScoreArgs a) {
  __shared__ __attribute__((aligned(16))) float add[128];
  const int hyp = blockIdx.x, lane = threadIdx.x;
  float m[9], mi[9];
#pragma unroll
  for (int q = 0; q < 9; q++) {
    m[q] = a.M21[hyp * 9 + q];
    mi[q] = a.kind == 0 ? a.M12[hyp * 9 + q] : 0.f;
  }
  const float th = a.kind == 0 ? 5.991f : 3.841f, thScore = 5.991f;
  const float invSigmaSquare = a.invSigmaSquare;
  float score = 0;
  for (int i0 = 0; i0 < a.N; i0 += 64) {
    const int i = i0 + lane;
    float c1 = 0.f, c2 = 0.f;
    if (i < a.N) {
      bool bIn = true;
      const orbx_keypoint kp1 = a.k1[a.first[i]], kp2 = a.k2[a.second[i]];
      const float u1 = kp1.x, v1 = kp1.y, u2 = kp2.x, v2 = kp2.y;
      float chiSquare1, chiSquare2;
      if (a.kind == 0) {  // CheckHomography :300-343
        const float w2in1inv = (float)(1.0 / (double)(mi[6] * u2 + mi[7] * v2 + mi[8]));
        const float u2in1 = (mi[0] * u2 + mi[1] * v2 + mi[2]) * w2in1inv;
        const float v2in1 = (mi[3] * u2 + mi[4] * v2 + mi[5]) * w2in1inv;
        const float squareDist1 = (u1 - u2in1) * (u1 - u2in1) + (v1 - v2in1) * (v1 - v2in1);
        chiSquare1 = squareDist1 * invSigmaSquare;
        const float w1in2inv = (float)(1.0 / (double)(m[6] * u1 + m[7] * v1 + m[8]));
        const float u1in2 = (m[0] * u1 + m[1] * v1 + m[2]) * w1in2inv;
        const float v1in2 = (m[3] * u1 + m[4] * v1 + m[5]) * w1in2inv;
        const float squareDist2 = (u2 - u1in2) * (u2 - u1in2) + (v2 - v1in2) * (v2 - v1in2);
        chiSquare2 = squareDist2 * invSigmaSquare;
      } else {            // CheckFundamental :385-428
        const float a2 = m[0] * u1 + m[1] * v1 + m[2];
        const float b2 = m[3] * u1 + m[4] * v1 + m[5];
        const float c2f = m[6] * u1 + m[7] * v1 + m[8];
        const float num2 = a2 * u2 + b2 * v2 + c2f;
        const float squareDist1 = num2 * num2 / (a2 * a2 + b2 * b2);
        chiSquare1 = squareDist1 * invSigmaSquare;
        const float a1 = m[0] * u2 + m[3] * v2 + m[6];
        const float b1 = m[1] * u2 + m[4] * v2 + m[7];
        const float c1f = m[2] * u2 + m[5] * v2 + m[8];
        const float num1 = a1 * u1 + b1 * v1 + c1f;
        const float squareDist2 = num1 * num1 / (a1 * a1 + b1 * b1);
        chiSquare2 = squareDist2 * invSigmaSquare;
      }
      if (chiSquare1 > th) bIn = false;
      else c1 = thScore - chiSquare1;
      if (chiSquare2 > th) bIn = false;
      else c2 = thScore - chiSquare2;
      a.inliers[(size_t)hyp * a.N + i] = bIn ? 1 : 0;
    }
    add[2 * lane] = c1;
    add[2 * lane + 1] = c2;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
#pragma unroll 8
      for (int q = 0; q < 128; q++) score += add[q];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) a.scores[hyp] = score;
}

// =================================================================================================
// Colour -> gray: Converter::toGray (Utils/Converter.cpp:5-19) = cv::cvtColor(COLOR_RGB2GRAY / COLOR_BGR2GRAY) on 8-bit,
// Y = (c0*ch0 + c1*ch1 + c2*ch2 + 2^(shift-1)) >> shift with (c0, c1, c2) = (R, G, B weights) for RGB order and reversed for BGR:
// (4899, 9617, 1868) >> 14 (OpenCV 3.x .. 4.0, the default) or (9798, 19235, 3735) >> 15 (OpenCV >= 4.1), orbx_set_opencv_variant.
// Thread = 4 output pixels: 12 source bytes as 3 aligned dwords -> one dword store (byte paths when unaligned / at the
// row tail).  HBM streaming: 3 B read + 1 B written per pixel.
// =================================================================================================
__global__ __launch_bounds__(256) void k_to_gray(const uint8_t* __restrict__ src, long long srcFrameStride, int sstride, int w,
                                                 int h, int c0, int c1, int c2, int shift, int srcAligned, uint8_t* __restrict__ dst,
                                                 long long dstFrameStride, int dstride, int dstAligned) {
  const uint32_t half = 1u << (shift - 1);
  const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int y = blockIdx.y;
  if (x >= w) return;
  const uint8_t* s = src + (size_t)blockIdx.z * srcFrameStride + (size_t)y * sstride + (size_t)3 * x;
  uint8_t* d = dst + (size_t)blockIdx.z * dstFrameStride + (size_t)y * dstride + x;
  uint32_t out = 0;
  if (x + 4 <= w && srcAligned) {
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(s);
    const uint32_t a = s4[0], b = s4[1], c = s4[2];  // bytes 0..11 = p0(0,1,2) p1(3,4,5) p2(6,7,8) p3(9,10,11)
    const uint32_t y0 = ((a & 255) * c0 + ((a >> 8) & 255) * c1 + ((a >> 16) & 255) * c2 + half) >> shift;
    const uint32_t y1 = ((a >> 24) * c0 + (b & 255) * c1 + ((b >> 8) & 255) * c2 + half) >> shift;
    const uint32_t y2 = (((b >> 16) & 255) * c0 + (b >> 24) * c1 + (c & 255) * c2 + half) >> shift;
    const uint32_t y3 = (((c >> 8) & 255) * c0 + ((c >> 16) & 255) * c1 + (c >> 24) * c2 + half) >> shift;
    out = y0 | y1 << 8 | y2 << 16 | y3 << 24;
    if (dstAligned) {
      *reinterpret_cast<uint32_t*>(d) = out;
      return;
    }
    d[0] = (uint8_t)y0; d[1] = (uint8_t)y1; d[2] = (uint8_t)y2; d[3] = (uint8_t)y3;
    return;
  }
  const int n = min(4, w - x);
  for (int i = 0; i < n; i++) d[i] = (uint8_t)((s[3 * i] * c0 + s[3 * i + 1] * c1 + s[3 * i + 2] * c2 + half) >> shift);
}

__global__ __launch_bounds__(256) void k_copy_rows(const uint8_t* __restrict__ src, long long srcFrameStride, int sstride, int w,
                                                   int h, uint8_t* __restrict__ dst, long long dstFrameStride, int dstride) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= w) return;
  dst[(size_t)blockIdx.z * dstFrameStride + (size_t)blockIdx.y * dstride + x] =
      src[(size_t)blockIdx.z * srcFrameStride + (size_t)blockIdx.y * sstride + x];
}

// =================================================================================================
// launch wrappers (called from orbx_api.cpp)
// =================================================================================================
hipError_t launch_to_gray(hipStream_t st, int nFrames, const uint8_t* src, long long srcFrameStride, int sstride, int w, int h,
                          int channels, int rgb, uint8_t* dst, long long dstFrameStride, int dstride, int grayVariant) {
  if (nFrames <= 0) return hipSuccess;
  if (channels == 1) {  // Converter.cpp:6-8: copyTo
    dim3 block(256, 1, 1), grid((w + 255) / 256, h, nFrames);
    hipLaunchKernelGGL(k_copy_rows, grid, block, 0, st, src, srcFrameStride, sstride, w, h, dst, dstFrameStride, dstride);
    return hipGetLastError();
  }
  const int srcAligned = ((uintptr_t)src % 4 == 0) && (srcFrameStride % 4 == 0) && (sstride % 4 == 0);
  const int dstAligned = ((uintptr_t)dst % 4 == 0) && (dstFrameStride % 4 == 0) && (dstride % 4 == 0);
  dim3 block(256, 1, 1), grid((w + 1023) / 1024, h, nFrames);
  const int cr = grayVariant ? 9798 : 4899, cg = grayVariant ? 19235 : 9617, cb = grayVariant ? 3735 : 1868, shift = grayVariant ? 15 : 14;
  hipLaunchKernelGGL(k_to_gray, grid, block, 0, st, src, srcFrameStride, sstride, w, h, rgb ? cr : cb, cg, rgb ? cb : cr, shift,
                     srcAligned, dst, dstFrameStride, dstride, dstAligned);
  return hipGetLastError();
}

// Results of a host-frame batch (orbx_extract_match_batch_host_async) into the caller's page-locked arrays, by stores over the link
// instead of copy commands: a copy command queued behind a batch's kernels sits at the head of its DMA queue until they have
// finished, and the NEXT batches' uploads queue behind it (measured: uploads and kernels of four lanes strictly alternate, 0.61 of
// the link's rate; tools/host_pipeline_timeline.py).  grid = (chunks, rows, segments); only the entries a frame really has travel.
__global__ __launch_bounds__(256) void k_copy_out(const CopyOut c) {
  const CopySeg& s = c.s[blockIdx.z];
  for (int row = blockIdx.y; row < s.rows; row += gridDim.y) {
    const int n = s.cnt ? min(max(s.cnt[row], 0) * s.mult, s.rowDwords) : s.rowDwords;
    const uint32_t* src = s.src + (long long)row * s.rowDwords;
    uint32_t* dst = s.dst + (long long)row * s.rowDwords;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dst[i] = src[i];
  }
}
hipError_t launch_copy_out(hipStream_t st, const CopyOut& c, int nseg, int maxRows) {
  if (nseg <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_copy_out, dim3(8, (unsigned)std::min(std::max(maxRows, 1), 1024), (unsigned)nseg), dim3(256), 0, st, c);
  return hipGetLastError();
}

// test hook: the cos / sin pair of k_describe_patch for arbitrary angles (degrees)
__global__ __launch_bounds__(256) void k_debug_sincos(const float* __restrict__ angle, int n, float* __restrict__ c, float* __restrict__ s,
                                                      const int libmFloat) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float factorPI = (float)(3.14159265358979323846 / 180.f);
  if (libmFloat) {
    sincosfGlibc(angle[i] * factorPI, &s[i], &c[i]);
    return;
  }
  double sd, cd;
  ORBX_SINCOS((double)(angle[i] * factorPI), &sd, &cd, false);
  c[i] = (float)cd;
  s[i] = (float)sd;
}
hipError_t launch_debug_sincos(hipStream_t st, const float* angle, int n, float* c, float* s, int libmFloat) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_debug_sincos, dim3((n + 255) / 256), dim3(256), 0, st, angle, n, c, s, libmFloat);
  return hipGetLastError();
}

hipError_t launch_check_model(hipStream_t st, int nModels, const ScoreArgs& a) {
  if (nModels <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_check_model, dim3(nModels), dim3(64), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_undistort(hipStream_t st, int nFrames, const orbx_keypoint* in, const int* nkp, int capacity, const CamD& c,
                            orbx_keypoint* out) {
  if (nFrames <= 0 || capacity <= 0) return hipSuccess;
  dim3 block(256, 1, 1), grid((capacity + 255) / 256, nFrames, 1);
  hipLaunchKernelGGL(k_undistort, grid, block, 0, st, in, nkp, capacity, c, out);
  return hipGetLastError();
}

hipError_t launch_pyramid_tiles(hipStream_t st, int nFrames, const uint8_t* img0, long long img0FrameStride, uint8_t* pyr,
                                const Geom& g, const PyrTileRect* rects, const PyrTileTap* taps, int nTiles, int buf0Bytes,
                                int bufBytes) {
  const size_t lds = ORBX_PYR_TILE_TAPS * sizeof(PyrTileTap) + (size_t)buf0Bytes + 2 * (size_t)bufBytes;
  hipLaunchKernelGGL(k_pyramid_tiles, dim3(nTiles, nFrames, 1), dim3(256, 1, 1), lds, st, img0, img0FrameStride, pyr, g, rects, taps,
                     buf0Bytes, bufBytes);
  return hipGetLastError();
}

hipError_t launch_resize(hipStream_t st, int nFrames, const uint8_t* src, long long srcFrameStride, int sw, int sh, int sstride,
                         uint8_t* dst, long long dstFrameStride, int dw, int dh, int dstride, const ResizeTab* xtab,
                         const ResizeTab* ytab, int dwordPath, int wideFrames) {
  // wideFrames: frames of this launch behind whose last source row more memory is known to follow (k_resize_dw's 12-byte loads)
  dim3 block(64, 4, 1), grid((dw + 255) / 256, (dh + 3) / 4, nFrames);
  if (dwordPath)
    hipLaunchKernelGGL(k_resize_dw, dim3(grid.x, (dh + 4 * RESIZE_DW_ROWS - 1) / (4 * RESIZE_DW_ROWS), nFrames), block, 0, st, src, srcFrameStride, sw, sh, sstride, dst, dstFrameStride, dw, dh,
                       dstride, xtab, ytab, wideFrames);
  else
    hipLaunchKernelGGL(k_resize, grid, block, 0, st, src, srcFrameStride, sw, sh, sstride, dst, dstFrameStride, dw, dh, dstride,
                       xtab, ytab);
  return hipGetLastError();
}

hipError_t launch_pyramid_bands(hipStream_t st, int nFrames, const uint8_t* img0, long long img0FrameStride, uint8_t* pyr,
                                const Geom& g, const ResizeTab* tab, const PyrBands& pb) {
  if (nFrames <= 0 || g.nlevels <= 1) return hipSuccess;
  dim3 block(PYR_T, 1, 1), grid(pb.nBands * pb.nStrips, nFrames, 1);
  if (pb.maxRows > 256 || pb.maxRows < 1 || pb.nStrips < 1 || pb.nStrips > ORBX_PYR_STRIPS_MAX) return hipErrorInvalidValue;  // (the host picks the band count accordingly)
  // one group per thread when no level's strip is wider than PYR_T groups (GMAX = 1: fewer registers)
  bool one = true;
  for (int l = 1; l < g.nlevels; l++)
    for (int s2 = 0; s2 < pb.nStrips; s2++) one = one && pb.g1[s2][l] - pb.g0[s2][l] <= PYR_T;
  // ... and when rows of at most 256 groups leave 512 threads at least two rows per pass: with one row of 270 .. 400 groups per
  // pass (1920x1080: levels 1 .. 3) two groups per thread fill the lanes so much better that they win (config 3: pyramid 0.242
  // against 0.285 ms per batch; 640x480 and the strips of 3840x2160 the other way round: 425.2 k against 416.8 k frames/s,
  // 0.206 against 0.243 ms)
  one = one && g.nlevels > 1 && pb.g1[0][1] - pb.g0[0][1] <= PYR_T / 2;
  const int gmaxKnob = (int)knob(KNOB_PYR_GMAX, 0);  // diagnostics
  if (gmaxKnob == 2) one = false;
  if (gmaxKnob == 1) {
    one = true;
    for (int l = 1; l < g.nlevels; l++)
      for (int s2 = 0; s2 < pb.nStrips; s2++) one = one && pb.g1[s2][l] - pb.g0[s2][l] <= PYR_T;
  }
  const size_t lds = 2 * (size_t)pb.maxRows * sizeof(uint4);
  const uint4* tab4 = reinterpret_cast<const uint4*>(tab);
  if (pb.dual2 && one) hipLaunchKernelGGL((k_pyramid_bands<1, 1>), grid, block, lds, st, img0, img0FrameStride, pyr, g, tab4, pb);
  else if (pb.dual2) hipLaunchKernelGGL((k_pyramid_bands<1, 2>), grid, block, lds, st, img0, img0FrameStride, pyr, g, tab4, pb);
  else if (one) hipLaunchKernelGGL((k_pyramid_bands<0, 1>), grid, block, lds, st, img0, img0FrameStride, pyr, g, tab4, pb);
  else hipLaunchKernelGGL((k_pyramid_bands<0, 2>), grid, block, lds, st, img0, img0FrameStride, pyr, g, tab4, pb);
  return hipGetLastError();
}

hipError_t launch_fast(hipStream_t st, int nFrames, const uint8_t* img0, long long img0FrameStride, int img0Aligned,
                       const uint8_t* pyr, const Geom& g, uint32_t* cand, int* cellCount, const FastCell* cells, int waveOk,
                       int* usedWave) {
  int cw = 7, ch = 7;  // largest cell image of this geometry (cell + 6 px overlap, cpp:1094-1103)
  for (int l = 0; l < g.nlevels; l++) {
    cw = std::max(cw, std::min(g.L[l].wCell + 6, ORBX_CELL_MAX));
    ch = std::max(ch, std::min(g.L[l].hCell + 6, ORBX_CELL_MAX));
  }
  const bool forceOld = knobOn(KNOB_FAST_WG);  // diagnostics (orbx_debug_set): the workgroup-per-cell kernel
  // a lone wave needs 17 us for its cell; the four waves of k_fast's workgroup 8.8 us: the latter for the one-frame call, whose
  // cells cannot fill the chip either way (measured up to eight 640x480 frames = 4616 cells per launch: tools/exp_fast_small.sh)
  // (the parity tests run their small batches through both kernels: knob fast_wg_max_cells = 0)
  const int wgMaxCells = (int)knob(KNOB_FAST_WG_MAX_CELLS, 5000);
  const bool small = (long long)nFrames * g.nCellsTotal <= wgMaxCells;
  if (usedWave) *usedWave = (waveOk && img0Aligned && cells && !forceOld && !small) ? 1 : 0;
  if (waveOk && img0Aligned && cells && !forceOld && !small) {
    // one wave per workgroup, FW_CPW cells per wave; x size padded to whole rounds of 8 runs (XCD-aware order)
    const int groups = ((g.nCellsTotal + FW_CPW - 1) / FW_CPW + 8 * FW_XK - 1) / (8 * FW_XK) * (8 * FW_XK);
    const int ts = waveOk == 2 ? 48 : 64;  // tile / strength-map row stride: 48 when every cell image is <= 12 dwords wide
    // (+ 16: the quick reject's dword reads reach a few bytes beyond the last tile row)
    const int tileBytes = ch * ts + 16, smapBytes = (ch - 6 + 2) * ts;
    const int fastPad = (int)knob(KNOB_FAST_LDS_PAD, 0);  // diagnostics: fewer waves per CU
    const size_t lds = (size_t)(16 + tileBytes + smapBytes + FW_RING * 2 + FW_CORN * 2) + fastPad;  // k_fast_wave's layout
    const bool dbg = knobOn(KNOB_FAST_DEBUG);
    if (dbg) {
      int nb48 = -1, nb64 = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb48, k_fast_wave<48>, 64, lds);
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb64, k_fast_wave<64>, 64, lds);
      fprintf(stderr, "k_fast_wave: ts %d lds %zu B, occupancy API: %d / %d workgroups per CU\n", ts, lds, nb48, nb64);
    }
    if (ts == 48)
      hipLaunchKernelGGL(k_fast_wave<48>, dim3(groups, nFrames, 1), dim3(64, 1, 1), lds, st, img0, img0FrameStride, pyr, g, cells,
                         cand, cellCount, tileBytes, smapBytes);
    else
      hipLaunchKernelGGL(k_fast_wave<64>, dim3(groups, nFrames, 1), dim3(64, 1, 1), lds, st, img0, img0FrameStride, pyr, g, cells,
                         cand, cellCount, tileBytes, smapBytes);
    return hipGetLastError();
  }
  dim3 block(FAST_T, 1, 1), grid(g.nCellsTotal, nFrames, 1);
  FastLds fl;
  fl.tileBytes = (ch * TILE_STRIDE + 15) & ~15;
  fl.smapBytes = (SMAP_STRIDE * (ch - 6 + 2) + 15) & ~15;
  fl.listBytes = (2 * (cw - 6) * (ch - 6) + 15) & ~15;
  fl.outCap = ((cw - 6 + 1) / 2) * ((ch - 6 + 1) / 2);  // NMS survivors are never 8-neighbours
  const size_t lds = (size_t)fl.tileBytes + fl.smapBytes + fl.listBytes + (size_t)fl.outCap * 4;
  hipLaunchKernelGGL(k_fast, grid, block, lds, st, img0, img0FrameStride, img0Aligned, pyr, g, cand, cellCount, fl);
  return hipGetLastError();
}

hipError_t launch_describe_patch(hipStream_t st, int nFrames, int maxSel, const uint8_t* img0, long long img0FrameStride,
                                 int img0Aligned, const uint8_t* pyr, const Geom& g, const SelKp* sel, const int* nsel,
                                 orbx_keypoint* kps, uint8_t* desc, int capacity, int gaussVariant, int libmFloat,
                                 const DescStage* staged) {
  // staged (optional): the selection's staging lists -- the kernel indexes them itself and writes the frames' totals (the caller
  // launched no k_sel_compact; see DescStage)
  if (maxSel <= 0 && !staged) return hipSuccess;
  dim3 block(64 * DESC_WAVES, 1, 1), grid(((maxSel + DESC_WAVES - 1) / DESC_WAVES + 7) / 8 * 8, nFrames, 1);  // x: multiple of 8
  const int descPad = (int)knob(KNOB_DESC_LDS_PAD, 0);  // diagnostics: fewer waves per CU
  const DescStage none = {};
  if (staged) {
    if (gaussVariant)
      hipLaunchKernelGGL((k_describe_patch<1, true>), grid, block, descPad, st, img0, img0FrameStride, img0Aligned, pyr, g, sel, nsel, kps,
                         desc, capacity, *staged, libmFloat);
    else
      hipLaunchKernelGGL((k_describe_patch<0, true>), grid, block, descPad, st, img0, img0FrameStride, img0Aligned, pyr, g, sel, nsel, kps,
                         desc, capacity, *staged, libmFloat);
  } else if (gaussVariant)
    hipLaunchKernelGGL((k_describe_patch<1, false>), grid, block, descPad, st, img0, img0FrameStride, img0Aligned, pyr, g, sel, nsel, kps,
                       desc, capacity, none, libmFloat);
  else
    hipLaunchKernelGGL((k_describe_patch<0, false>), grid, block, descPad, st, img0, img0FrameStride, img0Aligned, pyr, g, sel, nsel, kps,
                       desc, capacity, none, libmFloat);
  return hipGetLastError();
}

hipError_t launch_match(hipStream_t st, int nPairs, const int* dFirst, const int* dSecond, const orbx_keypoint* kps,
                        const uint8_t* desc, const int* nkp, int capacity, orbx_bounds b, int window, float nnratio, int checkOri,
                        int* matches12, int* nmatches, int* stats, int* scratch, int pair0, int wideMode, int* hostWide,
                        unsigned int* diag) {
  // wideMode 1: k_match_jacobi and the wide kernels behind it; 0: k_match_jacobi only (the caller expects no pair to need the
  // wide path and checks *hostWide afterwards); 2: the wide path alone, prep included, for the pairs still pending
  if (nPairs <= 0) return hipSuccess;
  MatchParams mp;
  mp.capacity = capacity; mp.window = window; mp.nnratio = nnratio; mp.checkOri = checkOri; mp.b = b;
  mp.noGeneral = knobOn(KNOB_MATCH_NO_GENERAL) ? 1 : 0;  // tests: see which pairs the parallel paths complete
  mp.noMfma = knobOn(KNOB_MATCH_NO_MFMA) ? 1 : 0;
  mp.pair0 = pair0;
  // With bestDist <= TH_LOW required, a train at distance s can matter as best only if s <= TH_LOW and as second-best
  // only if nnratio * s < TH_LOW (ORBmatcher.cpp:84-87: accepted iff bestDist <= nnratio * bestDist2 in f32, and f32
  // multiplication is monotonic): the wide path lists the candidates below the first s that satisfies neither.
  mp.dmax = TH_LOW + 1;
  while (mp.dmax < 257 && !(nnratio * (float)mp.dmax >= (float)TH_LOW)) mp.dmax++;
  // k_match_jacobi: pairs with <= 256 octave-0 queries and eligible trains, candidate lists and claims in LDS; a pair it
  // cannot take is prepared there for the wide path (<= 4096; lists in the pair's scratch, built by k_match_wide_lists
  // over the whole GPU, resolved by k_match_wide_resolve, which also runs the reference's sequential loop for what is
  // left).  The two wide kernels return at once for the pairs that are already done.
  const long long stride = matchScratchStride(capacity);  // = ensureMatchScratch
  const int capl = matchWideCap(capacity);
  const size_t lds = (size_t)capl * 12;  // head + tOrd + nextQ + outD: 48 KB at MW_CAP
  if (wideMode != 2)
    hipLaunchKernelGGL((k_match_jacobi<256, 4, 16>), dim3(nPairs), dim3(1024), 0, st, dFirst, dSecond, kps, desc, nkp, mp,
                       matches12, nmatches, stats, scratch, stride, capl, hostWide);
  if (wideMode == 2)
    hipLaunchKernelGGL(k_match_wide_prep, dim3(nPairs), dim3(MW_T), 0, st, dFirst, dSecond, kps, nkp, mp, matches12, nmatches,
                       scratch, stride, capl);
  if (wideMode != 0) {
    // (the kernel takes brute-force blocks only in a launch with at least 256 blocks of 256 queries, and a block is brute force only
    // if its queries' windows reach across the whole bounds: where the launch shape or the window rules that out for every pair --
    // 16 pairs of 1080p frames, a window of 100 pixels -- the launch would be 5.6 us of workgroups that look and leave)
    // (a query's window reaches grid column 0 and the last one only if 2 r > (COLS - 3) cell widths, rows alike: what is not launched
    // here is listed by k_match_wide_lists, so this is a matter of time only)
    const long long w2 = 2ll * window, ex = b.max_x - b.min_x, ey = b.max_y - b.min_y;
    const bool bfPossible = (long long)nPairs * ((capl + 255) >> 8) >= 256 && w2 * ORBX_GRID_COLS > (ORBX_GRID_COLS - 3) * ex &&
                            w2 * ORBX_GRID_ROWS > (ORBX_GRID_ROWS - 3) * ey;
    if (!mp.noMfma && bfPossible)
      hipLaunchKernelGGL(k_match_bf_mfma, dim3((capl + BF_QWG - 1) / BF_QWG, nPairs), dim3(BF_T), 0, st, dFirst, dSecond, kps, desc, nkp, mp,
                         nmatches, scratch, scratch, stride, capl, diag);
    {
      const long long wgs = (long long)nPairs * ((capl + 63) / 64);
      if (wgs <= 512)
        hipLaunchKernelGGL(k_match_wide_lists<8>, dim3((capl + 63) / 64, nPairs), dim3(512), 0, st, dFirst, dSecond, kps, desc, nkp, mp,
                           nmatches, scratch, scratch, stride, capl);
      else
        hipLaunchKernelGGL(k_match_wide_lists<4>, dim3((capl + 63) / 64, nPairs), dim3(256), 0, st, dFirst, dSecond, kps, desc, nkp, mp,
                           nmatches, scratch, scratch, stride, capl);
    }
    const int qpw = (long long)nPairs * capl >= 100000 ? 16 : 4;  // queries per wave of k_match_wide_sort
    hipLaunchKernelGGL(k_match_wide_sort, dim3((capl + 4 * qpw - 1) / (4 * qpw), nPairs), dim3(256), 0, st, mp, nmatches, scratch, stride,
                       capl, qpw);
    hipLaunchKernelGGL(k_match_wide_resolve, dim3(nPairs), dim3(MW_T), lds, st, dFirst, dSecond, kps, desc, nkp, mp, matches12,
                       nmatches, stats, scratch, stride, capl);
  }
  return hipGetLastError();
}

}  // namespace orbx
