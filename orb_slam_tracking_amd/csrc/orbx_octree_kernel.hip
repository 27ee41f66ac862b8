// orbx_octree_kernel.hip — device version of the keypoint quadtree selection
// (reference: ORBextractor::DistributeOctTree, Features/ORBextractor.cpp:698-1011; DivideNode cpp:617-676;
//  compareNodes cpp:684-696; truncation to the level quota cpp:1159-1161; coordinate fix-up cpp:1165-1179).
//
// One workgroup per (frame, pyramid level) -- large units: a wave per bucket of keys, then one workgroup per unit (below).  The
// array formulation (its host prototype lives under tests/cpp/host_quadtree.cpp, test infrastructure):
// every candidate gets a path code (root index + one quadrant digit per depth); with the codes sorted, every tree
// node is a contiguous range and the reference's std::list bookkeeping becomes arithmetic on common-prefix lengths.
//
// The partial last pass ("split the biggest nodes first until N nodes exist", cpp:897-965) needs the exact
// permutation libstdc++'s UNSTABLE std::sort produces on (count, UL.x).  libstdc++'s sort is
//   __introsort_loop (median-of-3 quicksort partitions down to ranges of <= 16, heapsort on depth exhaustion)
//   followed by __final_insertion_sort, and an insertion sort is a STABLE sort of whatever it is given.
// So only the partition phase is replayed literally (one lane, O(E log(E/16)) steps); the final insertion sort is
// computed as a parallel stable rank sort.  Everything else (node creation, cut point, list order) is parallel.
//
// The candidates arrive unordered from k_fast; the reference's candidate order (cell row, cell col, y, x) only
// matters as the tie-breaker "first of the highest responses", so it is carried as an order key, never materialised.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "../../include/orbx.h"
#include "orbx_device.h"
#include "orbx_knobs.h"

namespace orbx {

#define OCT_DEPTH 16

// Diagnostic build only (-DORBX_OCT_STAMPS): s_memtime stamps of the selection kernel's phases, per workgroup, into a
// buffer nothing else reads (cdna_hip_programming.md section 7).  The production build contains no stamp.
#ifdef ORBX_OCT_STAMPS
#define OCT_NSTAMP 16
__device__ unsigned long long g_octStamps[4096 * OCT_NSTAMP];
#define OCT_STAMP(k)                                                                                         \
  do {                                                                                                       \
    __syncthreads();                                                                                         \
    if (threadIdx.x == 0) {                                                                                  \
      const int b_ = (blockIdx.y * gridDim.x + blockIdx.x) & 4095;                                           \
      g_octStamps[b_ * OCT_NSTAMP + (k)] = __builtin_amdgcn_s_memtime();                                     \
    }                                                                                                        \
  } while (0)
#define OCT_STAMP_ACC(k, t0)                                                                                 \
  do {                                                                                                       \
    __syncthreads();                                                                                         \
    if (threadIdx.x == 0) {                                                                                  \
      const int b_ = (blockIdx.y * gridDim.x + blockIdx.x) & 4095;                                           \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                          \
      g_octStamps[b_ * OCT_NSTAMP + (k)] += now_ - (t0);                                                     \
      (t0) = now_;                                                                                           \
    }                                                                                                        \
  } while (0)
// ... the std::sort replay's phases, summed over its recursion levels (slot 7 = levels), per workgroup
__device__ unsigned long long g_octReplay[4096 * 8];
#define OCT_REPLAY_INIT() unsigned long long tRep_ = __builtin_amdgcn_s_memtime(); \
  if (threadIdx.x == 0) for (int k_ = 0; k_ < 8; k_++) g_octReplay[((blockIdx.y * gridDim.x + blockIdx.x) & 4095) * 8 + k_] = 0
#define OCT_REPLAY_ACC(k)                                                                                   \
  do {                                                                                                      \
    if (threadIdx.x == 0) {                                                                                 \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                         \
      g_octReplay[((blockIdx.y * gridDim.x + blockIdx.x) & 4095) * 8 + (k)] += now_ - tRep_;                \
      tRep_ = now_;                                                                                         \
    }                                                                                                       \
  } while (0)
#define OCT_REPLAY_COUNT(k) do { if (threadIdx.x == 0) g_octReplay[((blockIdx.y * gridDim.x + blockIdx.x) & 4095) * 8 + (k)] += 1; } while (0)
extern "C" int orbx_diag_oct_replay(unsigned long long* out, int nBlocks) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_octReplay), sizeof(unsigned long long) * 8 * nBlocks);
}
// ... and which XCD the bucket waves / the unit's workgroup of the many-workgroup selection ran on (placement check)
__device__ unsigned int g_octXcc[4096 * 9];
#define OCT_XCC_B1(u) atomicAdd(&g_octXcc[((u) & 4095) * 9 + (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 7)], 1u)
#define OCT_XCC_B2(u) g_octXcc[((u) & 4095) * 9 + 8] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 7
// ... and per bucket wave of k_octree_buckets: s_memtime behind its phases, the bucket's key count (tools/octb_stamps.py)
#define OCTB_NSTAMP 8
__device__ unsigned long long g_octbStamps[65536 * OCTB_NSTAMP];
#define OCTB_STAMP(k) do { if (lane == 0) g_octbStamps[((blockIdx.x * OCTB_WAVES + wv) & 65535) * OCTB_NSTAMP + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define OCTB_STAMP_V(k, v) do { if (lane == 0) g_octbStamps[((blockIdx.x * OCTB_WAVES + wv) & 65535) * OCTB_NSTAMP + (k)] = (unsigned long long)(v); } while (0)
extern "C" int orbx_diag_octb_stamps(unsigned long long* out, int nWaves) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_octbStamps), sizeof(unsigned long long) * 8 * (size_t)nWaves);
}
#else
#define OCT_REPLAY_INIT() do {} while (0)
#define OCT_REPLAY_ACC(k) do {} while (0)
#define OCT_REPLAY_COUNT(k) do {} while (0)
#define OCTB_STAMP(k) do {} while (0)
#define OCTB_STAMP_V(k, v) do {} while (0)
#define OCT_XCC_B1(u) do {} while (0)
#define OCT_XCC_B2(u) do {} while (0)
#define OCT_STAMP(k) do {} while (0)
#define OCT_STAMP_ACC(k, t0) do {} while (0)
#endif

typedef unsigned long long u64;

// sizes shared by both instances of the device code
#define OCT_SORT_LDS 2048
#define OCT_GLOBAL_XCHG 4096  // keys of k_octree_global's exchange buffer (32 KB: one such workgroup per CU anyway)
#define SLESS(a, b) (((a) >> 20) < ((b) >> 20))
#define OCT_PAR_MAX 512
#define OCT_PAR_BIG 2048  // k_octree_global: two positions per thread of its 1024
#define OCT_PAR_RANGES_FOR(capN) ((capN) / 16 < 64 ? 64 : (capN) / 16)  // ranges of more than 16 keys alive at a time: < capN / 16
#define OCT_PAR_SCR_FOR(capN) (4 * (((capN) + 8) / 2) + 5 * OCT_PAR_RANGES_FOR(capN) + 4)
#define OCT_PAR_SCR OCT_PAR_SCR_FOR(OCT_PAR_MAX)
// what stdSortPartitionPhasePar lays out in it: LR[capN + 8], SLR[capN + 2] / the wave tasks' 130 dwords each, the task list
#define OCT_PAR_NEED(capN) (((capN) + 8) + ((capN) + 2 > 130 * ((capN) / 128 < 4 ? 4 : (capN) / 128) ? (capN) + 2 : 130 * ((capN) / 128 < 4 ? 4 : (capN) / 128)) + (capN) / 16 + 2)
static_assert(OCT_PAR_NEED(256) <= OCT_PAR_SCR_FOR(256) && OCT_PAR_NEED(512) <= OCT_PAR_SCR_FOR(512) && OCT_PAR_NEED(2048) <= OCT_PAR_SCR_FOR(2048),
              "scratch of the parallel std::sort replay");
#define OCTBIG_NODES 8192   // k_octree_big: list nodes of a unit (LDS tables; M < 4 N)
#define OCTBIG_PEND 2048    // ... pending nodes of a partial-pass round (< N)
#define OCTBIG_XCHG 4096    // ... u64 entries of its sort exchange buffer (8192 32-bit node keys, 4096 64-bit ones)

#define OCT_T 256
namespace t256 {
#include "orbx_octree_body.inc"
}  // namespace t256
#undef OCT_T
#define OCT_T 1024
namespace t1024 {
#include "orbx_octree_body.inc"
}  // namespace t1024
#undef OCT_T
#define OCT_T 64
namespace t64 {  // single-wave workgroups (k_octree_buckets): the register sort without any cross-wave stage
#include "orbx_octree_body.inc"
}  // namespace t64
#undef OCT_T
#define OCT_T 256  // the kernels below run 256 threads unless they say otherwise
using namespace t256;

// ---- kernels ---------------------------------------------------------------------------------------------------------

// LDS-resident variant: n <= NMAX candidates, quota <= QMAX; K32: 32-bit sort keys (OctScratchT), for levels whose path codes
// need at most 21 bits
template <int NMAX, int QMAX, bool K32>
__global__ __launch_bounds__(OCT_T) void k_octree_lds(const uint32_t* __restrict__ cand, const int* __restrict__ cellCount,
                                                     const OctLaunch P, SelKp* __restrict__ selStage,
                                                     int* __restrict__ nselLevel, uint8_t* __restrict__ scratch,
                                                     int* __restrict__ maxN, int deferBig, int level0) {
  ORBX_SETPRIO();
  constexpr int MCAP = 4 * QMAX, FCAP = 2 * QMAX;
  static_assert((NMAX & (NMAX - 1)) == 0 && (MCAP & (MCAP - 1)) == 0, "sort buffers must be powers of two");
  // LDS budget (QMAX 256): NMAX 2048: 16 + 8 + 4 + 6 + 2 + 3 KB = 39 KB -> FOUR workgroups per CU (it was 51 KB and three
  //                        until sorted positions became 16-bit here and alone[] a function of div[]);
  //                        NMAX 1024:  8 + 8 + 3.4 + 6 + 1 + 3 KB = 29.4 KB -> five;
  //            (QMAX 128)  NMAX  512:  4 + 4 + 3.4 + 3 + 0.5 + 1.5 KB = 16.5 KB -> eight (the wave slots of the CU), for the upper
  //                        levels, whose quotas and candidate counts are small (launch_octree picks the instance per level).
  //   nodes[] is dead once the node records exist, so the partial pass's buffers (sized, pending, childCnt) live in it, and
  //   it is not yet written while step 1 reads the candidate position list, which therefore lives there too;
  //   hiOf[] (steps 3-4) shares its space with the parallel std::sort replay's scratch (partial pass).
  using KEY = typename std::conditional<K32, uint32_t, u64>::type;
  // LDS budget with 32-bit keys: NMAX 2048: 8 + 8 + 4 + 6 + 2 + 3 KB = 31.5 KB -> FIVE workgroups per CU
  constexpr int KEYW = K32 ? (NMAX / 2 < 1024 - MCAP ? 1024 - MCAP : NMAX / 2) : NMAX;  // u64 words of the key array
  __shared__ __attribute__((aligned(16))) u64 keysNodes[KEYW + MCAP];
  KEY* keys = reinterpret_cast<KEY*>(keysNodes);
  u64* nodes = keysNodes + KEYW;
  static_assert(MCAP * 8 >= NMAX * 4, "the candidate position list must fit into nodes[]");
  uint32_t* candL = reinterpret_cast<uint32_t*>(nodes);
  // the sort replay never sees more than QMAX entries here (the partial pass's list is shorter than the quota): 256 keys
  constexpr int PARCAP = 256;
  static_assert(QMAX <= PARCAP, "parallel-replay capacity");
  constexpr int HI_DWORDS = (NMAX / 2 > OCT_PAR_SCR_FOR(PARCAP)) ? NMAX / 2 : OCT_PAR_SCR_FOR(PARCAP);
  __shared__ __attribute__((aligned(16))) uint32_t hiPar[HI_DWORDS];
  uint16_t* hiOf = reinterpret_cast<uint16_t*>(hiPar);
  static_assert(NMAX <= 65535, "16-bit sorted positions");
  __shared__ uint16_t nodeLo[MCAP + FCAP], nodeHi[MCAP + FCAP];
  __shared__ __attribute__((aligned(16))) uint8_t div[NMAX + 16];  // (octChildBounds reads whole 16-byte chunks)
  __shared__ uint8_t nodeDepth[MCAP + FCAP];
  static_assert(2 * QMAX * 8 + 2 * QMAX * 4 + QMAX * 4 <= MCAP * 8, "partial-pass buffers must fit in nodes[]");
  u64* sized = nodes;                                             // [2 * QMAX]
  int* pending = reinterpret_cast<int*>(nodes + 2 * QMAX);        // [2 * QMAX]
  int* childCnt = pending + 2 * QMAX;                             // [QMAX]
  const int level = blockIdx.y + level0, f = blockIdx.x + P.frame0;  // level-major dispatch, see launch_octree
  int* nOut = &nselLevel[f * P.nlevels + level];
  // keys[] + nodes[] double as the sort exchange buffer of the global-scratch path: a workgroup's worth of padded keys, the radix
  // sort's 256 x 4 digit counters (4 KB) and the parallel std::sort replay's scratch must fit
  static_assert(KEYW + MCAP >= 1024 && (KEYW + MCAP) * 2 >= OCT_PAR_SCR_FOR(OCT_PAR_MAX), "exchange buffer of the global-scratch path");
  static_assert(NMAX <= 2048, "11-bit candidate index of the 32-bit keys");
  __shared__ int redo;
  __shared__ int gws[OCT_T / 64];
  const uint32_t* segBase = cand + P.candOff[level] + (int64_t)f * P.candCap[level];
  const int* cellCnt = cellCount + (int64_t)f * P.nCellsTotal + P.lev[level].cellBase;
  const int n = gatherCandidates(cellCnt, P.lev[level].nCells, P.lev[level].segCap, candL, NMAX, threadIdx.x, gws);
  // feedback for the next batch's choice of instance: one plain store per unit, reduced by k_sel_compact (an atomicMax on the
  // level's counter serialised the 256 units of a level behind each other at the memory side: 9 us before the first barrier)
  if (threadIdx.x == 0 && maxN) maxN[f * P.nlevels + level] = n;
  // 32-bit keys: the level's path codes must fit 21 bits (the host picks this instance only then; checked again here)
  const int depthBits = P.lev[level].depthBits;
  const int rootBits = P.lev[level].nIni > 1 ? 32 - __builtin_clz((unsigned)(P.lev[level].nIni - 1)) : 0;
  const bool keysFit = !K32 || (depthBits >= 1 && depthBits <= OCT_DEPTH && rootBits + 2 * depthBits <= 21);
  if (n <= NMAX && P.lev[level].quota <= QMAX && keysFit) {
    // the candidate words of the unit, in candidate-list order: head of the unit's global scratch area (octScratchBytes >= 8 KB)
    uint32_t* candE = reinterpret_cast<uint32_t*>(scratch + P.scrOff[level] + (int64_t)f * P.scrStride[level]);
    OctScratchT<uint16_t, KEY, (NMAX / OCT_T + 3) / 4 * 4> S{keys, 2 * (OCT_DEPTH - depthBits), nodes, div, hiOf, nodeLo, nodeHi, nodeDepth, sized, pending,
                                 childCnt, candL, segBase, nullptr, 0, hiPar /* hiOf's space: dead during the partial pass */, PARCAP,
                                 nullptr, candE,
                                 reinterpret_cast<uint32_t*>(nodes) /* step 6 only: nodes[] and what aliases it are dead by then */};
    static_assert(MCAP * 8 >= NMAX * 4, "the sorted candidate copy of step 6 must fit into nodes[]");
    octreeSelect(S, n, P.lev[level], P.codeTab, level, selStage + (int64_t)f * P.selStride + P.selOff[level], nOut, MCAP, FCAP, QMAX);
    __syncthreads();
    if (threadIdx.x == 0) redo = (*nOut == -2);  // a node table overflowed the LDS layout
    __syncthreads();
    if (!redo) return;
  }
  // the unit does not fit the LDS layout.  deferBig (the launch expects such units: large quotas / candidate counts): mark
  // it for k_octree_global, which runs them with 1024 threads; otherwise (a rare outlier of a configuration that fits) the
  // same workgroup handles it on global scratch, so that no second kernel sits on the stream's critical path
  if (deferBig) {
    if (threadIdx.x == 0) *nOut = -2;
    return;
  }
  __syncthreads();
  octreeGlobalUnit(cand, cellCount, P, selStage, nselLevel, scratch, level, f, keysNodes, KEYW + MCAP, nullptr, 0);
}

// global-scratch variant, 1024 threads, for the (frame, level) units the LDS variant left (nselLevel == -2), or for all
// units when `all` is set.  Scratch of unit (f, level) starts at scrOff[level] + f * scrStride[level]; layout:
// octScratchBytes().
static_assert(OCT_SORT_LDS * 2 >= OCT_PAR_SCR_FOR(OCT_PAR_MAX), "the sort exchange buffer doubles as the parallel-replay scratch of the global-scratch units");
static_assert(OCT_SORT_LDS * 2 >= 256 * (1024 / 64), "... and as the 256 x nWaves digit counters of the radix sort (1024-thread instance)");
static_assert(OCT_SORT_LDS >= 1024, "the register sort of the 1024-thread instance exchanges 1024 padded keys through it");
__global__ __launch_bounds__(1024) void k_octree_global(const uint32_t* __restrict__ cand, const int* __restrict__ cellCount,
                                                       const OctLaunch P, SelKp* __restrict__ selStage,
                                                       int* __restrict__ nselLevel, uint8_t* __restrict__ scratch, int all, int level0) {
  ORBX_SETPRIO();
  // exchange buffer: 2048 keys for the register sorts, and 512 x 16 digit counters (9-bit digits) for the radix sort
  __shared__ u64 xchg[OCT_GLOBAL_XCHG];
  // scratch of the workgroup-parallel std::sort replay for up to 2048 pending nodes (the level-0 quota of 1080p / 4000 features is
  // 869, of 4K / 8000 features 1737: the one-lane replay took 96 k cycles of such a unit)
  __shared__ __attribute__((aligned(16))) uint32_t parScr[OCT_PAR_SCR_FOR(OCT_PAR_BIG)];
  // a second set of the radix sort's 512 x 16 digit counters: a pass's scatter sweep counts the next pass's digits
  __shared__ uint32_t radixCnt2[512 * (1024 / 64)];
  static_assert(sizeof(radixCnt2) == OCT_GLOBAL_XCHG * sizeof(u64), "as many counters as the exchange buffer holds");
  const int level = blockIdx.y + level0, f = blockIdx.x + P.frame0;  // level-major dispatch, see launch_octree
  if (!all && nselLevel[f * P.nlevels + level] != -2) return;
  t1024::octreeGlobalUnit(cand, cellCount, P, selStage, nselLevel, scratch, level, f, xchg, OCT_GLOBAL_XCHG, parScr, OCT_PAR_BIG, radixCnt2);
}

// ---- large units on many workgroups -----------------------------------------------------------------------------------
// k_octree_buckets: ONE WAVE per BUCKET of a (frame, level) unit -- the keys of one tree node of depth bigD0 (octBigChoose).
// A key's path depends on its coordinates only, so the bucket owns a rectangle of the level, and its candidates lie in the
// segments of the FAST cells that overlap it: the wave lists those cells' survivors, keeps the ones whose path code has the
// bucket's prefix, sorts them (registers + shuffles, no barrier) by the rest of the code and writes, into the bucket's slot of
// the unit's arrays, the sorted keys, a score per key (response << 40 | inverted reference candidate order: the emit's "first
// of the highest responses" is a maximum) and the divergence depth between neighbours; the bucket's record gets its count,
// the histograms of the inner divergences / first lonely depths, and the first and last inner divergence (what the
// neighbouring BUCKETS decide -- the divergence at the slot's two ends and the lonely depth of its first and last key -- is
// left to k_octree_big).  A bucket is a chain of memory round trips (cell counts, candidates, code tables) and ~40 sort stages
// for a few hundred keys; single-wave workgroups with 13 KB of LDS put twelve of them on a CU (four-wave workgroups around a
// 2048-key LDS sort: 54-78 us per launch instead of ~20).
// Placement: workgroups go round-robin to the 8 XCDs, and the buckets of unit u = (level - level0) * nFrames + frame are given to
// the workgroups of XCD u % 8 -- the XCD k_octree_big's workgroup of that unit runs on (its grid index is u), so what is
// written here is read there from the same L2 instead of through the fabric.
#define OCTB_WAVES 4      // buckets (= independent waves) per workgroup: the dispatcher starts ~400 workgroups a microsecond
#define OCTB_T (64 * OCTB_WAVES)
#define OCTB_MAXCELLS 255

// bitonic sort of 64 E keys of ONE wave, E per lane (lane t owns a[t E .. t E + E)); partners at distance < E are registers of
// the same lane, the others lanes of the wave (shuffles); no workgroup barrier anywhere (the waves of k_octree_buckets are
// independent).  `a` is LDS private to the wave.
template <int E>
__device__ __forceinline__ void waveSort32(uint32_t* a, int lane) {
  constexpr int NTOT = 64 * E;
  uint32_t v[E];
#pragma unroll
  for (int e = 0; e < E; e++) v[e] = a[lane * E + e];
  for (int k = 2; k <= NTOT; k <<= 1) {
    t64::bitonicLaneStages<E>(v, lane, k);
#pragma unroll
    for (int jj = E / 2; jj > 0; jj >>= 1) {
      if (jj < k) {
#pragma unroll
        for (int e = 0; e < E; e++) {
          const int pe = e ^ jj;
          if (pe > e) {
            const bool asc = ((lane * E + e) & k) == 0;
            const uint32_t x = v[e], y = v[pe];
            const bool sw = (x > y) == asc;
            v[e] = sw ? y : x;
            v[pe] = sw ? x : y;
          }
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < E; e++) a[lane * E + e] = v[e];
}
// (orders a wave's own LDS traffic: what its lanes wrote is what its lanes read behind this point)
#define OCTB_WAVE_SYNC()                                        \
  do {                                                          \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      \
    __builtin_amdgcn_wave_barrier();                            \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      \
  } while (0)

template <int CAP>  // LDS slots per wave (512 / 1024): chosen from the previous batch's fullest bucket -- 22 / 38 KB per workgroup
__global__ __launch_bounds__(OCTB_T) void k_octree_buckets(const uint32_t* __restrict__ cand, const int* __restrict__ cellCount,
                                                          const OctLaunch P, uint8_t* __restrict__ scratch, int level0, int level1,
                                                          int nFrames) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  OCTB_STAMP(0);
  OCTB_STAMP_V(6, 0);
  int level, f, b;
  {
    const int nUnits = (level1 - level0) * nFrames;
    // (the levels' bucket counts first, all scalar loads in flight together: the walk below then costs no memory round trips)
    int nbL[ORBX_MAX_LEVELS];
#pragma unroll
    for (int l = 0; l < ORBX_MAX_LEVELS; l++) nbL[l] = level0 + l < level1 ? P.lev[min(level0 + l, ORBX_MAX_LEVELS - 1)].bigBuckets : 0;
    int u = blockIdx.x & 7, k = (blockIdx.x >> 3) * OCTB_WAVES + wv;
    for (;; u += 8) {
      if (u >= nUnits) return;
      const int li = u / nFrames;
      int nb = 0;
#pragma unroll
      for (int l = 0; l < ORBX_MAX_LEVELS; l++) nb = li == l ? nbL[l] : nb;
      if (k < nb) break;
      k -= nb;
    }
    level = level0 + u / nFrames; f = P.frame0 + u % nFrames; b = k;
    if (lane == 0) OCT_XCC_B1(u);
  }
  const OctLevel& L = P.lev[level];
  static_assert(CAP <= ORBX_OCTB_CAP && CAP >= 64, "bucket slots in LDS");
  __shared__ uint32_t keysAll[OCTB_WAVES][CAP];
  __shared__ uint32_t ceAll[OCTB_WAVES][CAP];
  __shared__ int cpreAll[OCTB_WAVES][OCTB_MAXCELLS + 1];
  __shared__ uint16_t cidxAll[OCTB_WAVES][OCTB_MAXCELLS + 1];
  uint32_t* keysL = keysAll[wv];
  uint32_t* ceL = ceAll[wv];
  int* cpre = cpreAll[wv];
  uint16_t* cidx = cidxAll[wv];  // FAST cell (row-major index in the level's grid) of the rectangle's c-th cell
  uint32_t* candBuf;
  int mCap, fCap, qMax;
  t64::OctScratch S = t64::octCarve(P, level, f, scratch, &candBuf, &mCap, &fCap, &qMax);
  int* info = reinterpret_cast<int*>(candBuf) + (size_t)b * ORBX_OCTB_INFO;
  u64* score = S.sortTmp;
  const int d0 = L.bigD0, capB = L.bigCapB;
  const size_t base = (size_t)b * capB;
  // the bucket's rectangle: root, x prefix and y prefix from the bucket id (root << 2 D0 | D0 quadrant digits, y bit above x bit)
  const int root = b >> (2 * d0);
  uint32_t xp = 0, yp = 0;
  for (int d = 0; d < d0; d++) {
    xp |= ((uint32_t)(b >> (2 * d)) & 1u) << d;
    yp |= ((uint32_t)(b >> (2 * d + 1)) & 1u) << d;
  }
  // (the interval tables are kept at the level's deepest bucket depth: a coarser prefix owns the union of its refinements)
  const uint32_t* bt = P.codeTab + L.bigTabOff;
  const int dm = L.bigDMax, ds = dm - d0;
  const int xlo = (int)bt[(root << dm) + ((int)xp << ds)], xhi = (int)bt[(root << dm) + (((int)xp + 1) << ds)];
  const uint32_t* bty = bt + (L.nIni << dm) + 1;
  const int ylo = (int)bty[(int)yp << ds], yhi = (int)bty[((int)yp + 1) << ds];
  // FAST cell (ci, cj) holds the candidates with (x - 3) / wCell == cj and (y - 3) / hCell == ci (k_fast: detection areas tile)
  const int nRowsC = L.nCells / L.nCols;
  int cx0 = 0, cx1 = -1, cy0 = 0, cy1 = -1;
  if (xhi > xlo && yhi > ylo) {
    cx0 = min(max(xlo - 3, 0) / L.wCell, L.nCols - 1); cx1 = min(max(xhi - 1 - 3, 0) / L.wCell, L.nCols - 1);
    cy0 = min(max(ylo - 3, 0) / L.hCell, nRowsC - 1); cy1 = min(max(yhi - 1 - 3, 0) / L.hCell, nRowsC - 1);
  }
  const int cw = cx1 - cx0 + 1, nc = cw * (cy1 - cy0 + 1);
  const int* cellCnt = cellCount + (int64_t)f * P.nCellsTotal + L.cellBase;
  // the sort key of a candidate: the digits of its path code between the bucket's depth and the level's one-pixel depth D (all
  // below are 0, all above the same for the whole bucket) -- at most 2 (D - D0) <= 22 bits -- then its slot (10 bits)
  const int dBits = L.depthBits >= 1 && L.depthBits <= OCT_DEPTH ? L.depthBits : OCT_DEPTH;
  const int remBits = 2 * (dBits - d0);
  bool over = nc > OCTB_MAXCELLS || remBits > 22 || remBits < 0;
  int nRaw = 0;
  if (!over && nc > 0) {
    // exclusive prefix of the cells' counts, row-major over the rectangle (a wave scan per 64 cells)
    for (int c0 = 0; c0 < nc; c0 += 64) {
      const int c = c0 + lane;
      const int cell = (cy0 + c / cw) * L.nCols + cx0 + c % cw;
      const int v = c < nc ? cellCnt[cell] : 0;
      if (c < nc) cidx[c] = (uint16_t)cell;
      const int inc = t64::waveScanIncl(v);
      if (c < nc) cpre[c] = nRaw + inc - v;
      nRaw += __builtin_amdgcn_readlane(inc, 63);
    }
    if (lane == 0) cpre[nc] = nRaw;
  }
  OCTB_WAVE_SYNC();
  OCTB_STAMP(1);
  const uint32_t* segBase = cand + P.candOff[level] + (int64_t)f * P.candCap[level];
  const uint2* __restrict__ tabX = reinterpret_cast<const uint2*>(P.codeTab + L.tabOff);
  const uint32_t* __restrict__ tabY = P.codeTab + L.tabOff + 2 * L.tabW;
  const int pshift = 2 * (OCT_DEPTH - d0);  // the bucket id = code >> pshift
  const int rshift = 2 * (OCT_DEPTH - dBits);
  const uint32_t remMask = remBits > 0 ? (1u << remBits) - 1u : 0u;
  int n = 0;
  if (!over) {
    // four raw candidates per lane and step: their searches, candidate words and table words are in flight together
    for (int r0 = 0; r0 < nRaw; r0 += 256) {
      uint32_t ce[4], digits[4];
      uint2 tx[4];
      int cellOf[4];
      bool valid[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int r = r0 + 64 * j + lane;
        valid[j] = r < nRaw;
        int lo = 0, hi = nc;  // cpre[lo] <= r < cpre[hi]
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (cpre[mid] <= r) lo = mid; else hi = mid;
        }
        cellOf[j] = cidx[lo];
        ce[j] = valid[j] ? segBase[(size_t)cellOf[j] * L.segCap + (r - cpre[lo])] : 0u;
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        tx[j] = tabX[min((int)(ce[j] & 0xfff), L.tabW - 1)];
        digits[j] = tabY[min((int)((ce[j] >> 12) & 0xfff), L.tabH - 1)];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        digits[j] |= tx[j].x;
        const u64 code = ((u64)tx[j].y << 32) | (u64)digits[j];
        const bool in = valid[j] && (int)(code >> pshift) == b;
        const unsigned long long m = __ballot(in);
        if (in) {
          const int slot = n + __popcll(m & ((1ull << lane) - 1ull));
          if (slot < CAP) {
            keysL[slot] = (((digits[j] >> rshift) & remMask) << 10) | (uint32_t)slot;
            ceL[slot] = ce[j];
          }
        }
        n += __popcll(m);
      }
    }
  }
  if (over || n >= capB || n > CAP) {  // (n == capB: the slot's last entry holds the divergence to the next bucket)
    if (lane == 0) { info[0] = 0; info[3] = 1; }
    return;
  }
  static_assert(ORBX_OCTB_CAP <= 1024, "ten slot bits in a sort key");
  OCTB_STAMP(2);
  OCTB_STAMP_V(6, n + 1);
  OCTB_STAMP_V(7, nRaw);
  static_assert(ORBX_OCTB_CAP <= 16 * 64, "register sort: at most sixteen keys per lane");
  int nPad = 64;
  while (nPad < n) nPad <<= 1;
  for (int i = n + lane; i < nPad; i += 64) keysL[i] = ~0u;
  OCTB_WAVE_SYNC();
  if (n > 1) {
    const int e = nPad / 64;
    if (e == 1) waveSort32<1>(keysL, lane);
    else if (e == 2) waveSort32<2>(keysL, lane);
    else if (e == 4) waveSort32<4>(keysL, lane);
    else if (e == 8) waveSort32<8>(keysL, lane);
    else waveSort32<16>(keysL, lane);
  }
  OCTB_WAVE_SYNC();
  OCTB_STAMP(3);
  // the 16-level path code of sorted key i: root | the bucket's digits | the sorted digits | zeros
  const u64 codeTop = ((u64)root << 32) | (d0 > 0 ? (u64)((uint32_t)(b & ((1 << (2 * d0)) - 1)) << (32 - 2 * d0)) : 0ull);
  auto codeAt = [&](int i) { return codeTop | ((u64)(keysL[i] >> 10) << rshift); };
  const uint32_t wInv = (uint32_t)(((1u << 24) + (uint32_t)L.wCell - 1u) / (uint32_t)L.wCell);
  const uint32_t hInv = (uint32_t)(((1u << 24) + (uint32_t)L.hCell - 1u) / (uint32_t)L.hCell);
  int hD[OCT_DEPTH + 2], hA[OCT_DEPTH + 2];  // (wave-uniform counters: ballots, no LDS atomics on a handful of addresses)
#pragma unroll
  for (int d = 0; d < OCT_DEPTH + 2; d++) { hD[d] = 0; hA[d] = 0; }
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    int dl = -1, al = -1;
    if (i < n) {
      const u64 code = codeAt(i);
      const int slot = (int)(keysL[i] & 1023u);
      const uint32_t ce = ceL[slot];
      S.keys[base + i] = (code >> S.keyShift) << 24;
      // reference candidate order (cell row, cell col, y, x), cpp:1078-1137; the FAST cell of a candidate from its coordinates,
      // (x - 3) / wCell and (y - 3) / hCell (k_fast: the cells' detection areas tile the level), by 2^24 reciprocals: exact for
      // coordinates below 4096 and cells of at least 16 pixels (the error term x (m - 2^24 / w) stays below 2^24 / w)
      const uint32_t cx = ce & 0xfffu, cy = (ce >> 12) & 0xfffu;
      const uint32_t cc = (uint32_t)(((u64)(cx > 3u ? cx - 3u : 0u) * (u64)wInv) >> 24), cr = (uint32_t)(((u64)(cy > 3u ? cy - 3u : 0u) * (u64)hInv) >> 24);
      const u64 rank = ((u64)(cr * (uint32_t)L.nCols + cc) << 24) | (u64)(ce & 0xffffffu);
      score[base + i] = ((u64)(ce >> 24) << 40) | (((1ull << 40) - 1ull) - rank);
      if (i > 0) {
        dl = t64::divDepth(codeAt(i - 1), code);
        S.div[base + i] = (uint8_t)dl;
        if (i + 1 < n) al = max(dl, t64::divDepth(code, codeAt(i + 1)));
        if (i == 1) info[1] = dl;
        if (i == n - 1) info[2] = dl;
      }
    }
#pragma unroll
    for (int d = 1; d < OCT_DEPTH + 2; d++) {  // (inside a bucket keys part below the bucket's depth, down to the one-pixel depth)
      if (d > d0 && d <= dBits + 1) {            // (uniform)
        hD[d] += __popcll(__ballot(dl == d));
        hA[d] += __popcll(__ballot(al == d));
      }
    }
  }
  if (lane < OCT_DEPTH + 2) {
    int vd = 0, va = 0;
#pragma unroll
    for (int d = 0; d < OCT_DEPTH + 2; d++) { vd = lane == d ? hD[d] : vd; va = lane == d ? hA[d] : va; }
    info[4 + lane] = vd; info[4 + OCT_DEPTH + 2 + lane] = va;
  }
  if (lane == 0) {
    info[0] = n; info[3] = 0;
    if (n < 2) { info[1] = 255; info[2] = 255; }
  }
  OCTB_STAMP(4);
}
static_assert(ORBX_OCTB_INFO >= 4 + 2 * (OCT_DEPTH + 2), "a bucket's record holds both histograms");

// k_octree_big: the tree arithmetic of the units k_octree_buckets has prepared, one workgroup of 1024 threads per unit
// (octreeSelectBig).  A unit it cannot take (a bucket overflowed, the full passes stopped above the bucket depth, node tables
// beyond its LDS) is redone by the same workgroup with the one-workgroup code (octreeGlobalUnit).
__global__ __launch_bounds__(1024) void k_octree_big(const uint32_t* __restrict__ cand, const int* __restrict__ cellCount, const OctLaunch P,
                                                    SelKp* __restrict__ selStage, int* __restrict__ nselLevel,
                                                    uint8_t* __restrict__ scratch, int* __restrict__ maxN, int level0, int fallback) {
  ORBX_SETPRIO();
  __shared__ __attribute__((aligned(16))) u64 xchg[OCTBIG_XCHG];
  __shared__ __attribute__((aligned(16))) uint32_t parScr[OCT_PAR_SCR_FOR(OCT_PAR_BIG)];
  __shared__ __attribute__((aligned(16))) uint32_t nodeLH[OCTBIG_NODES];
  __shared__ uint16_t nodeUlx[OCTBIG_NODES];
  __shared__ int pNd[OCTBIG_PEND], pLo[OCTBIG_PEND], pHiD[OCTBIG_PEND];
  __shared__ uint16_t pUlx[OCTBIG_PEND];
  static_assert(OCT_PAR_BIG <= OCTBIG_PEND && OCTBIG_PEND <= OCTBIG_XCHG, "the replay's keys live in the exchange buffer");
  const int level = blockIdx.y + level0, f = blockIdx.x + P.frame0;  // level-major dispatch, see launch_octree
  int* nOut = &nselLevel[f * P.nlevels + level];
  uint32_t* candBuf;
  int mCap, fCap, qMax;
  t1024::OctScratch S = t1024::octCarve(P, level, f, scratch, &candBuf, &mCap, &fCap, &qMax);
  S.xchg = xchg; S.xchgCap = OCTBIG_XCHG;
  S.parScr = parScr; S.parCap = OCT_PAR_BIG;
  S.nodeLH = nodeLH; S.nodeUlx = nodeUlx; S.pNd = pNd; S.pLo = pLo; S.pHiD = pHiD; S.pUlx = pUlx;
  if (threadIdx.x == 0) OCT_XCC_B2(blockIdx.y * gridDim.x + blockIdx.x);
  __shared__ int handedOn, redoneCount;
  if (threadIdx.x == 0) handedOn = 1;
  __syncthreads();
  if (P.lev[level].bigBuckets > 0) {  // (uniform; no bucket plan for the level: straight to the one-workgroup code)
    t1024::octreeSelectBig(S, S.sortTmp, reinterpret_cast<const int*>(candBuf), P.lev[level], level,
                           selStage + (int64_t)f * P.selStride + P.selOff[level], nOut, mCap, fCap, qMax,
                           maxN ? maxN + (f * P.nlevels + level) : nullptr);
    __syncthreads();
    if (threadIdx.x == 0) handedOn = *nOut == -2;
    __syncthreads();
  }
  if (!handedOn || !fallback) return;  // (block-uniform)
  // The unit could not be taken (a bucket overflowed, the full passes stopped above the bucket depth, node tables beyond the LDS
  // ones): this workgroup redoes it from the candidates with the one-workgroup code of k_octree_global, in place -- no second
  // kernel waits behind every batch for a case that hardly ever occurs.  The count is published with ORBX_OCT_REDONE set:
  // k_octree_emit must not take it for an output list's (k_sel_compact strips the bit).
  static_assert(OCTBIG_XCHG >= OCT_GLOBAL_XCHG && OCTBIG_NODES * 4 >= 512 * (1024 / 64) * 4, "the one-workgroup code's LDS buffers fit this kernel's");
  const int idx = f * P.nlevels + level;
  t1024::octreeGlobalUnit(cand, cellCount, P, selStage, &redoneCount - idx, scratch, level, f, xchg, OCT_GLOBAL_XCHG, parScr, OCT_PAR_BIG, nodeLH);
  __syncthreads();
  if (threadIdx.x == 0) *nOut = redoneCount >= 0 ? (redoneCount | ORBX_OCT_REDONE) : redoneCount;
}

// k_octree_emit: the units' output lists (k_octree_big: key ranges of the first N alive nodes in list order) -> SelKp records: a
// range's first key with the highest response (cpp:984-1007) is the key with the largest score.  One thread per range, 256 per
// workgroup (many workgroups: one CU's L1 takes ~4 cycles per scattered line, 45 k cycles for the 1737 ranges of a 4K level 0),
// every range's scores in flight sixteen at a time.
#ifndef ORBX_TAIL_T
#define ORBX_TAIL_T 64  // threads per workgroup of k_octree_emit / k_sel_compact: ONE wave (see k_sel_compact)
#endif
__global__ __launch_bounds__(ORBX_TAIL_T) void k_octree_emit(const OctLaunch P, SelKp* __restrict__ selStage, const int* __restrict__ nselLevel,
                                                    uint8_t* __restrict__ scratch, int level0) {
  ORBX_SETPRIO();
  const int level = blockIdx.y + level0, f = blockIdx.x + P.frame0;
  const int nOutNodes = nselLevel[f * P.nlevels + level];
  const int p = blockIdx.z * ORBX_TAIL_T + threadIdx.x;
  if (nOutNodes < 0 || (nOutNodes & ORBX_OCT_REDONE) || p >= nOutNodes) return;  // (a redone unit's records are in place already)
  uint32_t* candBuf;
  int mCap, fCap, qMax;
  const OctScratch S = octCarve(P, level, f, scratch, &candBuf, &mCap, &fCap, &qMax);
  const u64* __restrict__ score = S.sortTmp;
  const uint32_t e = (uint32_t)S.pending[p];
  const int lo = (int)(e & 0x7ffffu), hi = lo + (int)(e >> 19);
  u64 best = 0ull;
  for (int i = lo; i < hi; i += 16) {
    u64 sv[16];
#pragma unroll
    for (int q = 0; q < 16; q++) sv[q] = score[min(i + q, hi - 1)];
#pragma unroll
    for (int q = 0; q < 16; q++) best = sv[q] > best ? sv[q] : best;
  }
  const u64 rank = ((1ull << 40) - 1ull) - (best & ((1ull << 40) - 1ull));
  SelKp kp;
  kp.x = (uint16_t)((uint32_t)(rank & 0xfff) + ORBX_MIN_BORDER);  // cpp:1171-1172
  kp.y = (uint16_t)((uint32_t)((rank >> 12) & 0xfff) + ORBX_MIN_BORDER);
  kp.level = (uint8_t)level;
  kp.response = (uint8_t)(best >> 40);
  kp.pad = 0;
  selStage[(int64_t)f * P.selStride + P.selOff[level] + p] = kp;
}

size_t octScratchBytes(int nMax, int qMax) {
  size_t nPad = 1024;
  while ((int)nPad < nMax) nPad <<= 1;
  qMax = qMax < 1 ? 1 : qMax;
  const size_t mCap = 4 * (size_t)qMax, fCap = 16 * (size_t)qMax;
  size_t mPad = 1024;
  while (mPad < mCap) mPad <<= 1;
  size_t b = nPad * 8 + mPad * 8 + (size_t)2 * qMax * 8 + nPad * 4 + (mCap + fCap) * 8 + (size_t)3 * qMax * 4 + 2 * (nPad + 8) +
             2 * (mCap + fCap + 8) + 16 + nPad * 4 /* gathered candidates */ + nPad * 8 /* radix sort buffer */ +
             16 /* the divergence bytes start 16-byte aligned */;
  return (b + 255) / 256 * 256;
}

// compacts the per-level staging lists of every frame into level-major order and writes the per-frame totals: to the
// context's array (k_describe_patch reads it), to the caller's array if there is one, and to the host (mapped pinned
// memory), so that no copy command follows on the stream.  A unit the selection could not handle raises the host's error
// flag the same way.
__global__ __launch_bounds__(ORBX_TAIL_T) void k_sel_compact(const SelKp* __restrict__ selStage, const int* __restrict__ nselLevel,
                                                     const OctLaunch P, SelKp* __restrict__ sel, int* __restrict__ nsel,
                                                     int* __restrict__ nselUser, int* __restrict__ hostNsel, int selCap,
                                                     int* __restrict__ hostErr, int* __restrict__ maxN,
                                                     int* __restrict__ hostMaxN) {
  ORBX_SETPRIO();
  // grid (frames, parts): the workgroups (frame, 0 .. parts - 1) share the frame's copy; (frame, 0) writes its totals
  constexpr int T = ORBX_TAIL_T;
  const int f = blockIdx.x + P.frame0, part = blockIdx.y, parts = gridDim.y;
  // per-level maxima of the units' candidate counts (k_octree_lds: maxN[frame * nlevels + level]) of this launch go to pinned
  // host memory; the counts are reset for the next launch (levels no LDS unit ran on report 0)
  // (small launches: the first workgroup, one count per thread; else workgroup l takes level l)
  const int nUnits = (int)gridDim.x * P.nlevels;
  if (maxN && part == 0 && (nUnits <= 256 ? blockIdx.x == 0 : (int)blockIdx.x < P.nlevels)) {
    __shared__ int red[ORBX_MAX_LEVELS], redFill[ORBX_MAX_LEVELS], redDepth[ORBX_MAX_LEVELS];  // (ORBX_OCT_FEEDBACK: the maximum of each field)
    if (threadIdx.x < P.nlevels) { red[threadIdx.x] = 0; redFill[threadIdx.x] = 0; redDepth[threadIdx.x] = 0; }
    __syncthreads();
    if (nUnits <= 256) {  // (one wave, the code the staged descriptor kernel runs: orbx_device.h)
      if (threadIdx.x < 64) selReduceReports((int)threadIdx.x, P.frame0, (int)gridDim.x, P.nlevels, maxN, hostMaxN);
    } else {
      const int l = blockIdx.x;
      int m = 0;
      for (int i = threadIdx.x; i < (int)gridDim.x; i += T) {
        const int idx = (P.frame0 + i) * P.nlevels + l;
        const int v = maxN[idx];
        m = ORBX_OCT_FB_MAX(m, v);
        maxN[idx] = 0;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(m, o); m = ORBX_OCT_FB_MAX(m, t); }
      if ((threadIdx.x & 63) == 0 && m > 0) {
        atomicMax(&red[l], ORBX_OCT_FB_COUNT(m)); atomicMax(&redFill[l], ORBX_OCT_FB_FILL(m)); atomicMax(&redDepth[l], ORBX_OCT_FB_DEPTH(m));
      }
      __syncthreads();
      if (threadIdx.x == 0) hostMaxN[l] = ORBX_OCT_FEEDBACK(red[l], redFill[l], redDepth[l]);
    }
  }
  __shared__ int off[ORBX_MAX_LEVELS + 1];
  if (threadIdx.x == 0) {
    int acc = 0;
    bool bad = false;
    for (int l = 0; l < P.nlevels; l++) {
      off[l] = acc;
      acc += selUnitCount(nselLevel[f * P.nlevels + l], &bad);
    }
    off[P.nlevels] = acc;
    if (part == 0) selPublishFrame(f, acc, bad, nsel, nselUser, hostNsel, hostErr);
  }
  __syncthreads();
  for (int l = 0; l < P.nlevels; l++) {
    const int c = off[l + 1] - off[l];
    const SelKp* src = selStage + (int64_t)f * P.selStride + P.selOff[l];
    SelKp* dst = sel + (int64_t)f * selCap + off[l];
    for (int i = part * T + threadIdx.x; i < c; i += parts * T) dst[i] = src[i];
  }
}

// Instance of the LDS kernel for one level: the smallest one that holds the level's quota and, with 6 % headroom, the largest
// candidate count a unit of that level had in the previous batch of this geometry (0 = unknown: the 2048-candidate instance).
// 0 = the level expects units beyond the LDS layout (a quota above 256, more than 2048 candidates): those go to k_octree_global.
static int octInstanceFor(int quota, int hint) {
  if (quota > 256 || hint > 2048) return 0;
  if (hint > 0 && hint <= 480 && quota <= 128) return 512;
  if (hint > 0 && hint <= 960) return 1024;
  return 2048;
}

hipError_t launch_octree(hipStream_t st, int nFrames, const uint32_t* cand, const int* cellCount, const OctLaunch& P,
                         SelKp* selStage, int* nselLevel, uint8_t* scratch, int* maxN, const int* hintL, int force,
                         int* usedInstance) {
  // workgroups are dispatched x-fastest: all frames of level 0 first, then level 1, ...  The units of the lowest levels
  // run longest (most candidates, largest quota), so this is longest-processing-time-first and keeps the tail short.
  // Units the LDS variant cannot take (more than NMAX candidates, quota above 256, node-table overflow) run on global
  // scratch: when a level expects them (a quota above 256, or more than 2048 candidates in a unit of the previous
  // batch) they are deferred to k_octree_global, which gives each of them 1024 threads; otherwise the LDS kernel's own
  // workgroup handles the rare outlier and no second kernel is launched.
  // force: 0 = choose per level from hintL (nullptr = unknown); 2048 / 1024 / 512 = that LDS instance for every level (test
  // hook, diagnostics); -1 = every unit on global scratch (test hook).
  const bool noSmall = knobOn(KNOB_OCT_NO_SMALL);  // diagnostics (orbx_debug_set): always the 2048-candidate instance
  const bool key64Env = knobOn(KNOB_OCT_KEY64);    // diagnostics: always 64-bit sort keys
  const bool key64 = key64Env || (force & 0x10000) != 0;               // (test hook: force | 0x10000)
  force = force < 0 ? force : (force & 0xffff);
  // knob oct_split_min (diagnostics) = batch size from which every group of consecutive levels with the same instance gets its
  // own launch.  Off by default: measured on the bench workload (256 frames, four lanes) 2048 | 1024 x 2 | 512 x 5 gives 295.9 k
  // frames/s, 2048 x 3 | 512 x 5 300.2 k, 2048 | 1024 x 7 302.3 k against 306.6 k with ONE launch on the largest instance --
  // consecutive launches of a stream do not overlap (hipExtAnyOrderLaunch is ignored on gfx9: tools/microbench/any_order.hip),
  // so every group adds its own tail, and that costs more than the smaller units' LDS gives back to the other lanes.
  const int splitMin = (int)knob(KNOB_OCT_SPLIT_MIN, 1 << 30);
  if (usedInstance) *usedInstance = 0;
  // Large units (levels that expect them): k_octree_buckets sorts every unit's keys bucket by bucket on many workgroups,
  // k_octree_big does the tree arithmetic with one workgroup per unit (and redoes a unit it cannot take with the one-workgroup
  // code of k_octree_global, the round-1..3 path), k_octree_emit picks the keypoints.  knob oct_no_big (diagnostics): the old path.
  const bool noBig = knobOn(KNOB_OCT_NO_BIG);
  OctLaunch Q = P;  // the launch's bucket depths: from the candidate counts of the previous batch (octBigChoose)
  // (hintL: ORBX_OCT_FEEDBACK values -- the previous batch's largest candidate count and fullest bucket per level)
  auto hintOf = [&](int l) { return hintL ? ORBX_OCT_FB_COUNT(hintL[l]) : 0; };
  int estFill[ORBX_MAX_LEVELS] = {};  // (bound of the fullest bucket at the chosen depth, from the previous batch's fill and depth)
  for (int l = 0; l < Q.nlevels; l++)
    octBigChoose(&Q.lev[l], Q.scrNMax[l], hintOf(l), hintL ? ORBX_OCT_FB_FILL(hintL[l]) : 0, hintL ? ORBX_OCT_FB_DEPTH(hintL[l]) : 0, &estFill[l]);
  const int depthKnob = (int)knob(KNOB_OCT_BIG_DEPTH, -1);  // diagnostics / tests: this depth whatever the previous batch reported
  if (depthKnob >= 0)
    for (int l = 0; l < Q.nlevels; l++)
      if (Q.lev[l].bigBuckets > 0 && depthKnob <= Q.lev[l].bigDMax) {
        OctLevel& O = Q.lev[l];
        long long nPad = 1024;
        while (nPad < Q.scrNMax[l]) nPad <<= 1;
        const long long nb = (long long)O.nIni << (2 * depthKnob);
        int cap = ORBX_OCTB_CAP;
        while (cap > 0 && (long long)cap * nb > nPad) cap >>= 1;
        if (cap >= 256 && nb <= ORBX_OCTB_MAX_BUCKETS) { O.bigD0 = depthKnob; O.bigBuckets = (int32_t)nb; O.bigCapB = cap; estFill[l] = 0; }
      }
  auto launchBig = [&](int l0, int l1, bool fallback) {
    int nBuckets = 0;
    for (int l = l0; l < l1; l++) nBuckets += Q.lev[l].bigBuckets;
    if (nBuckets > 0) {
      // workgroup i of k_octree_buckets runs on XCD i % 8 and takes the (i / 8)-th bucket of the units u = i % 8, i % 8 + 8, ...
      long long perXcd[8] = {0, 0, 0, 0, 0, 0, 0, 0}, most = 0;
      for (int u = 0; u < (l1 - l0) * nFrames; u++) perXcd[u & 7] += Q.lev[l0 + u / nFrames].bigBuckets;
      for (int x = 0; x < 8; x++) most = std::max(most, perXcd[x]);
      // the waves' LDS slots: 512 per bucket when the previous batch's fullest bucket of these levels leaves a quarter of that free
      // (five workgroups per CU instead of four: 41 -> 36 us per half batch at 1080p / 4K); a fuller bucket overflows into the
      // in-place redo of its unit and reports ORBX_OCTB_CAP, which brings the 1024-slot instance back for the next batch
      int fillMost = 0;
      bool known = hintL != nullptr;
      for (int l = l0; l < l1; l++)
        if (Q.lev[l].bigBuckets > 0) {
          const int fl = estFill[l];  // (scaled to this launch's depth)
          known = known && fl > 0;
          fillMost = std::max(fillMost, fl);
        }
      const bool noSmallSlots = knobOn(KNOB_OCTB_NO_512);  // diagnostics
      const dim3 bgrid((unsigned)(8 * ((most + OCTB_WAVES - 1) / OCTB_WAVES)), 1, 1);
      if (known && !noSmallSlots && fillMost * 4 <= 512 * 3)
        hipLaunchKernelGGL(k_octree_buckets<512>, bgrid, dim3(OCTB_T), 0, st, cand, cellCount, Q, scratch, l0, l1, nFrames);
      else
        hipLaunchKernelGGL(k_octree_buckets<ORBX_OCTB_CAP>, bgrid, dim3(OCTB_T), 0, st, cand, cellCount, Q, scratch, l0, l1, nFrames);
      hipLaunchKernelGGL(k_octree_big, dim3(nFrames, l1 - l0, 1), dim3(1024), 0, st, cand, cellCount, Q, selStage, nselLevel, scratch, maxN, l0,
                         fallback ? 1 : 0);
      int qMost = 1;
      for (int l = l0; l < l1; l++) qMost = std::max(qMost, Q.lev[l].quota);
      hipLaunchKernelGGL(k_octree_emit, dim3(nFrames, l1 - l0, (qMost + ORBX_TAIL_T - 1) / ORBX_TAIL_T), dim3(ORBX_TAIL_T), 0, st, Q, selStage, nselLevel, scratch, l0);
    } else {
      hipLaunchKernelGGL(k_octree_global, dim3(nFrames, l1 - l0, 1), dim3(1024), 0, st, cand, cellCount, P, selStage, nselLevel, scratch, 1, l0);
    }
  };
  if (force == -2 || force == -3) {  // (test hook) the many-workgroup path for every unit, with / without the fallback behind it
    launchBig(0, P.nlevels, force == -2);
    return hipGetLastError();
  }
  if (force < 0) {
    hipLaunchKernelGGL(k_octree_global, dim3(nFrames, P.nlevels, 1), dim3(1024), 0, st, cand, cellCount, P, selStage, nselLevel, scratch,
                       1, 0);
    return hipGetLastError();
  }
  int inst[ORBX_MAX_LEVELS];
  int largest = 512;
  bool anyBig = false;
  for (int l = 0; l < P.nlevels; l++) {
    inst[l] = force > 0 ? force : octInstanceFor(P.lev[l].quota, hintOf(l));
    if (noSmall && inst[l] != 0 && force == 0) inst[l] = 2048;
    if (inst[l] == 0) anyBig = true;
    largest = std::max(largest, inst[l] == 0 ? 2048 : inst[l]);
  }
  // One launch on the largest instance any level needs (the levels of the reference's own 640x480 images stay below 960
  // candidates -> 30 KB units; sparse scenes with small quotas -> 17 KB units).
  if (nFrames < splitMin && force == 0)
    for (int l = 0; l < P.nlevels; l++) inst[l] = anyBig ? 0 : largest;
  const long long instKnob = knob(KNOB_OCT_INST, 0);  // diagnostics: one hex digit per level, 1 = 512, 2 = 1024, 3 = 2048
  if (instKnob && force == 0 && !anyBig && hintL && hintOf(0) > 0) {
    for (int l = 0; l < P.nlevels && l < 16; l++) {
      const int dgt = (int)((instKnob >> (4 * l)) & 15);
      const int v = dgt == 1 ? 512 : dgt == 2 ? 1024 : dgt == 3 ? 2048 : 0;
      if (v) inst[l] = std::max(inst[l] == largest ? octInstanceFor(P.lev[l].quota, hintOf(l)) : inst[l], v);
    }
  }
  if (usedInstance) {  // the smallest instance any level runs on (0: some level goes to k_octree_global)
    int m = 1 << 30;
    for (int l = 0; l < P.nlevels; l++) m = std::min(m, inst[l]);
    *usedInstance = m;
  }
  for (int l0 = 0; l0 < P.nlevels;) {
    int l1 = l0 + 1;
    while (l1 < P.nlevels && inst[l1] == inst[l0]) l1++;
    const dim3 grid(nFrames, l1 - l0, 1), block(OCT_T, 1, 1);
    // 32-bit sort keys when the path codes of every level of the launch fit 21 bits (frames up to ~1024 px per root and
    // side: VGA, 752x480, 1080p; not 4K)
    bool k32 = !key64;
    for (int l = l0; l < l1; l++) {
      const int rootBits = P.lev[l].nIni > 1 ? 32 - __builtin_clz((unsigned)(P.lev[l].nIni - 1)) : 0;
      k32 = k32 && P.lev[l].depthBits >= 1 && rootBits + 2 * P.lev[l].depthBits <= 21;
    }
    const int ldsPad = (int)knob(KNOB_OCT_LDS_PAD, 0);  // diagnostics: fewer units per CU
#define ORBX_OCT_LAUNCH(N_, Q_, DEFER_)                                                                                             \
  do {                                                                                                                              \
    if (k32)                                                                                                                        \
      hipLaunchKernelGGL((k_octree_lds<N_, Q_, true>), grid, block, ldsPad, st, cand, cellCount, P, selStage, nselLevel, scratch, maxN, DEFER_, l0);  \
    else                                                                                                                            \
      hipLaunchKernelGGL((k_octree_lds<N_, Q_, false>), grid, block, ldsPad, st, cand, cellCount, P, selStage, nselLevel, scratch, maxN, DEFER_, l0); \
  } while (0)
    switch (inst[l0]) {
      case 512: ORBX_OCT_LAUNCH(512, 128, 0); break;
      case 1024: ORBX_OCT_LAUNCH(1024, 256, 0); break;
      case 2048: ORBX_OCT_LAUNCH(2048, 256, 0); break;
      default: {  // the level expects large units
        bool planned = !noBig;
        for (int l = l0; l < l1; l++) planned = planned && Q.lev[l].bigBuckets > 0;
        if (planned) {
          launchBig(l0, l1, !knobOn(KNOB_OCT_BIG_NO_FALLBACK));
        } else {  // what fits the LDS layout is done there, the rest is deferred to the one-workgroup kernel
          ORBX_OCT_LAUNCH(2048, 256, 1);
          hipLaunchKernelGGL(k_octree_global, grid, dim3(1024), 0, st, cand, cellCount, P, selStage, nselLevel, scratch, 0, l0);
        }
        break;
      }
    }
#undef ORBX_OCT_LAUNCH
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    l0 = l1;
  }
  return hipSuccess;
}

hipError_t launch_sel_compact(hipStream_t st, int nFrames, const SelKp* selStage, const int* nselLevel, const OctLaunch& P,
                              SelKp* sel, int* nsel, int* nselUser, int* hostNsel, int selCap, int* hostErr, int* maxN,
                              int* hostMaxN) {
  hipLaunchKernelGGL(k_sel_compact, dim3(nFrames, 256 / ORBX_TAIL_T), dim3(ORBX_TAIL_T), 0, st, selStage, nselLevel, P, sel, nsel, nselUser, hostNsel, selCap,
                     hostErr, maxN, hostMaxN);
  return hipGetLastError();
}

#ifdef ORBX_OCT_STAMPS
extern "C" int orbx_diag_oct_stamps(unsigned long long* out, int nBlocks) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_octStamps), sizeof(unsigned long long) * OCT_NSTAMP * nBlocks);
}
extern "C" int orbx_diag_oct_xcc(unsigned int* out, int nUnits) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_octXcc), sizeof(unsigned int) * 9 * nUnits);
}
#endif

// ---- test hook: the std::sort replay alone (partition phase on one lane + parallel stable rank sort) ---------------
__global__ __launch_bounds__(OCT_T) void k_debug_sort(int* triples, int n, u64* a, u64* b) {
  const int tid = threadIdx.x;
  for (int j = tid; j < n; j += OCT_T)
    a[j] = ((u64)(uint32_t)triples[3 * j] << 40) | ((u64)((uint32_t)triples[3 * j + 1] & 0xfffff) << 20) |
           (u64)((uint32_t)triples[3 * j + 2] & 0xfffff);
  __syncthreads();
  __shared__ __attribute__((aligned(16))) uint32_t parScr[OCT_PAR_SCR];
  __shared__ u64 parKeys[OCT_PAR_MAX];
  __shared__ int parWs[4];
  if (n <= OCT_PAR_MAX) {  // the workgroup-parallel replay works on LDS keys, as in the selection kernels
    for (int j = tid; j < n; j += OCT_T) parKeys[j] = a[j];
    __syncthreads();
    if (n & 2) {  // ... with the whole sort from the partition phase's own ranges (k_octree_big's form): in-range ranks, no rank pass
      __shared__ u64 parRanked[OCT_PAR_MAX];
      stdSortPartitionPhasePar(parKeys, n, tid, parScr, parWs, n <= 256 && (n & 1) ? 256 : OCT_PAR_MAX, parRanked);
      __syncthreads();
      for (int j = tid; j < n; j += OCT_T) {
        triples[3 * j] = (int)(parRanked[j] >> 40);
        triples[3 * j + 1] = (int)((parRanked[j] >> 20) & 0xfffff);
        triples[3 * j + 2] = (int)(parRanked[j] & 0xfffff);
      }
      return;
    }
    stdSortPartitionPhasePar(parKeys, n, tid, parScr, parWs, n <= 256 && (n & 1) ? 256 : OCT_PAR_MAX);  // both layouts get exercised
    __syncthreads();
    for (int j = tid; j < n; j += OCT_T) a[j] = parKeys[j];
  } else {
    stdSortPartitionPhase(a, n, tid);
  }
  __syncthreads();
  for (int j = tid; j < n; j += OCT_T) {
    const u64 v = a[j], kv = v >> 20;
    int rank = 0;
    for (int i = 0; i < n; i++) {
      const u64 ki = a[i] >> 20;
      rank += (ki < kv) || (ki == kv && i < j);
    }
    b[rank] = v;
  }
  __syncthreads();
  for (int j = tid; j < n; j += OCT_T) {
    triples[3 * j] = (int)(b[j] >> 40);
    triples[3 * j + 1] = (int)((b[j] >> 20) & 0xfffff);
    triples[3 * j + 2] = (int)(b[j] & 0xfffff);
  }
}
hipError_t launch_debug_sort(hipStream_t st, int* triples, int n, unsigned long long* a, unsigned long long* b) {
  hipLaunchKernelGGL(k_debug_sort, dim3(1), dim3(OCT_T), 0, st, triples, n, a, b);
  return hipGetLastError();
}

}  // namespace orbx
