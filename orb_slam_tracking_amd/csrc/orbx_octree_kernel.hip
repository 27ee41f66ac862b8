// orbx_octree_kernel.hip — device version of the keypoint quadtree selection
// (reference: ORBextractor::DistributeOctTree, Features/ORBextractor.cpp:698-1011; DivideNode cpp:617-676;
//  compareNodes cpp:684-696; truncation to the level quota cpp:1159-1161; coordinate fix-up cpp:1165-1179).
//
// One workgroup per (frame, pyramid level).  Same array formulation as the host prototype in orbx_octree.cpp:
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

#include <cstdint>
#include <cstdlib>

#include "../../include/orbx.h"
#include "orbx_device.h"

namespace orbx {

#define OCT_T 256
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
#else
#define OCT_STAMP(k) do {} while (0)
#define OCT_STAMP_ACC(k, t0) do {} while (0)
#endif

typedef unsigned long long u64;

struct OctScratch {
  u64* keys;          // [nPad]   (code << 24) | candidate position in the unit's candidate area, sorted ascending
  u64* nodes;         // [mPad]   (17 - blockDepth) << 59 | orderKey << 19 | lo, sorted ascending = std::list order
  uint8_t* div;       // [n + 1]  divergence depth between sorted neighbours (255 at both ends = "separated")
  uint8_t* alone;     // [n]      first depth at which a key is the only key of its node
  uint32_t* hiOf;     // [n]      end of the node that starts at sorted position lo
  int* nodeLo;        // [mCap + fCap]  node records: list nodes first, then the nodes pushed during the partial pass
  int* nodeHi;
  uint8_t* nodeDepth;
  uint8_t* nodeAlive;
  u64* sized;         // [2 * qCap]  count << 40 | UL.x << 20 | node, two buffers (partitioned / stably sorted)
  int* pending;       // [2 * qCap]  two buffers
  int* childCnt;      // [qCap]      per sorted entry: children | multi-key children << 8
  const uint32_t* cand;     // [n] positions of the unit's candidates inside its candidate area (gatherCandidates); only step 1
                            // reads it, so in the LDS kernel it shares its space with hiOf
  const uint32_t* segBase;  // the unit's candidate area: cell c's survivors at segBase + c * segCap
  u64* xchg;          // LDS exchange buffer for the register sort when keys/nodes live in global memory, else nullptr
  uint32_t* parScr;   // OCT_PAR_SCR_FOR(parCap) dwords of LDS that are free during the partial pass (parallel std::sort replay)
  int parCap;         // largest array the parallel replay may take there (512 or 256 keys); larger ones use the one-lane replay
  u64* sortTmp;       // [n] second key buffer of the radix sort (global-scratch units only, else nullptr)
};

__device__ __forceinline__ int divDepth(u64 a, u64 b) {  // first depth at which two path codes differ
  const u64 x = a ^ b;
  if (!x) return OCT_DEPTH + 1;
  const int hb = 63 - __builtin_clzll(x);
  return hb >= 2 * OCT_DEPTH ? 0 : OCT_DEPTH - hb / 2;
}

// bitonic sort in memory, one barrier per stage: only for arrays too large for the register version below
__device__ __forceinline__ void bitonicSortMem(u64* a, int nPow2, int tid) {
  for (int k = 2; k <= nPow2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < nPow2; i += OCT_T) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const u64 x = a[i], y = a[ixj];
          const bool asc = (i & k) == 0;
          if ((x > y) == asc) { a[i] = y; a[ixj] = x; }
        }
      }
      __syncthreads();
    }
}

__device__ __forceinline__ u64 shflXor64(u64 v, int laneMask) {
  const uint32_t lo = __shfl_xor((uint32_t)v, laneMask), hi = __shfl_xor((uint32_t)(v >> 32), laneMask);
  return ((u64)hi << 32) | lo;
}

// Bitonic sort of 256*E keys held E per thread (thread t owns elements t*E .. t*E+E-1).  Compare-exchange partners at
// distance < E are registers of the same thread, at distance < 64*E lanes of the same wave (shuffles), and only the
// last stages (distance >= 64*E, three of them) go through memory with a barrier.
template <int E>
__device__ void bitonicSortRegs(u64* a, int tid) {
  constexpr int NTOT = OCT_T * E;
  u64 v[E];
#pragma unroll
  for (int e = 0; e < E; e++) v[e] = a[tid * E + e];
  for (int k = 2; k <= NTOT; k <<= 1) {
    for (int j = k >> 1; j >= E; j >>= 1) {
      if (j >= 64 * E) {  // partner in another wave
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; e++) a[tid * E + e] = v[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; e++) {
          const int i = tid * E + e;
          const u64 o = a[i ^ j];
          const bool lower = (i & j) == 0, asc = (i & k) == 0;
          const bool keepMin = lower == asc;
          v[e] = keepMin ? (o < v[e] ? o : v[e]) : (o > v[e] ? o : v[e]);
        }
      } else {  // partner lane in the same wave
        const int lm = j / E;
#pragma unroll
        for (int e = 0; e < E; e++) {
          const int i = tid * E + e;
          const u64 o = shflXor64(v[e], lm);
          const bool lower = (i & j) == 0, asc = (i & k) == 0;
          const bool keepMin = lower == asc;
          v[e] = keepMin ? (o < v[e] ? o : v[e]) : (o > v[e] ? o : v[e]);
        }
      }
    }
    // partner register of the same thread: distances E/2 .. 1 (compile-time, so v[] stays in registers)
#pragma unroll
    for (int jj = E / 2; jj > 0; jj >>= 1) {
      if (jj < k) {
#pragma unroll
        for (int e = 0; e < E; e++) {
          const int pe = e ^ jj;
          if (pe > e) {
            const bool asc = ((tid * E + e) & k) == 0;
            const u64 x = v[e], y = v[pe];
            const bool sw = (x > y) == asc;
            v[e] = sw ? y : x;
            v[pe] = sw ? x : y;
          }
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; e++) a[tid * E + e] = v[e];
  __syncthreads();
}

// sorts a[0 .. nPow2) ascending; entries beyond the real data must be padded with ~0 up to max(nPow2, 256).
// `xchg`: LDS buffer of OCT_SORT_LDS keys used when `a` itself is not in LDS (the register sort exchanges its last
// stages through LDS only); arrays larger than that are sorted in place, one barrier per stage.
#define OCT_SORT_LDS 2048
__device__ void bitonicSort(u64* a, int nPow2, int tid, u64* xchg) {
  if (nPow2 > OCT_SORT_LDS) {
    bitonicSortMem(a, nPow2, tid);
    return;
  }
  const int np = nPow2 < 256 ? 256 : nPow2;
  u64* s = a;
  if (xchg) {
    for (int i = tid; i < np; i += OCT_T) xchg[i] = a[i];
    __syncthreads();
    s = xchg;
  }
  if (np == 256) bitonicSortRegs<1>(s, tid);
  else if (np == 512) bitonicSortRegs<2>(s, tid);
  else if (np == 1024) bitonicSortRegs<4>(s, tid);
  else bitonicSortRegs<8>(s, tid);
  if (xchg) {
    for (int i = tid; i < np; i += OCT_T) a[i] = xchg[i];
    __syncthreads();
  }
}

// exclusive prefix sum of one int per thread over the workgroup; *total = sum.  `ws` = 4 ints of LDS.
__device__ __forceinline__ int blockScanExcl(int v, int tid, int* ws, int* total) {
  const int lane = tid & 63, wave = tid >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  __syncthreads();  // ws may still be read from a previous call
  if (lane == 63) ws[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < OCT_T / 64; w++) {
    const int s = ws[w];
    if (w < wave) base += s;
    tot += s;
  }
  *total = tot;
  return base + inc - v;
}

// Stable LSD radix sort of a[0 .. n) by the path code (bits 24 .. 24 + codeBits of the key), 4 bits per pass, for the
// units whose keys live in global memory (more than OCT_SORT_LDS candidates: the in-memory bitonic network needs
// log2(n)^2 / 2 barrier-separated sweeps, 120 at n = 32768).  The keys arrive ordered by their low 24 bits (the candidate
// position grows with the index), so a stable sort by the code alone yields the full key order.  Thread t owns the
// contiguous block [t * chunk, (t + 1) * chunk): per pass it counts its digits into cnt[digit][t], the 16 x 256 counters
// are scanned digit-major, and the block is scattered in order.  A pass in which every key has the same digit is
// skipped.  `cnt` = 4096 dwords of LDS, `ws` = 4 ints of LDS, b = second buffer of n keys.  Result in a.
__device__ void radixSortCodes(u64* a, u64* b, int n, int tid, uint32_t* cnt, int* ws, int codeBits) {
  const int chunk = (n + OCT_T - 1) / OCT_T;
  const int lo = min(tid * chunk, n), hi = min(lo + chunk, n);
  u64* src = a;
  u64* dst = b;
  for (int shift = 24; shift < 24 + codeBits; shift += 4) {
#pragma unroll
    for (int d = 0; d < 16; d++) cnt[d * OCT_T + tid] = 0;
    const int d0 = (int)((src[0] >> shift) & 15);
    bool same = true;
    for (int i = lo; i < hi; i++) {
      const int d = (int)((src[i] >> shift) & 15);
      same &= d == d0;
      cnt[d * OCT_T + tid]++;
    }
    if (__syncthreads_and(same ? 1 : 0)) continue;  // (also the barrier after the counting)
    // exclusive scan of the counters in (digit, thread) order: thread t owns entries [16 t, 16 t + 16)
    uint32_t loc[16];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { loc[k] = cnt[16 * tid + k]; sum += (int)loc[k]; }
    int total;
    uint32_t run = (uint32_t)blockScanExcl(sum, tid, ws, &total);
#pragma unroll
    for (int k = 0; k < 16; k++) { cnt[16 * tid + k] = run; run += loc[k]; }
    __syncthreads();
    for (int i = lo; i < hi; i++) {
      const u64 key = src[i];
      const int d = (int)((key >> shift) & 15);
      dst[cnt[d * OCT_T + tid]++] = key;
    }
    __syncthreads();
    u64* t2 = src; src = dst; dst = t2;
  }
  if (src != a) {
    for (int i = tid; i < n; i += OCT_T) a[i] = src[i];
    __syncthreads();
  }
}

// ---- literal replay of the PARTITION PHASE of libstdc++'s std::sort (bits/stl_algo.h: __introsort_loop) on packed
//      keys; the comparator (compareNodes: count, then UL.x) is "a >> 20 < b >> 20" -----------------------------------
#define SLESS(a, b) (((a) >> 20) < ((b) >> 20))

// key array accessor for the replay (LDS or global memory)
struct MemKeys {
  u64* p;
  __device__ __forceinline__ u64 get(int i) const { return p[i]; }
  __device__ __forceinline__ void set(int i, u64 v) { p[i] = v; }
};
template <class A>
__device__ void stdAdjustHeap(A& p, int first, int holeIndex, int len, u64 value) {
  const int topIndex = holeIndex;
  int secondChild = holeIndex;
  while (secondChild < (len - 1) / 2) {
    secondChild = 2 * (secondChild + 1);
    if (SLESS(p.get(first + secondChild), p.get(first + (secondChild - 1)))) secondChild--;
    p.set(first + holeIndex, p.get(first + secondChild));
    holeIndex = secondChild;
  }
  if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
    secondChild = 2 * (secondChild + 1);
    p.set(first + holeIndex, p.get(first + (secondChild - 1)));
    holeIndex = secondChild - 1;
  }
  int parent = (holeIndex - 1) / 2;  // __push_heap
  while (holeIndex > topIndex && SLESS(p.get(first + parent), value)) {
    p.set(first + holeIndex, p.get(first + parent));
    holeIndex = parent;
    parent = (holeIndex - 1) / 2;
  }
  p.set(first + holeIndex, value);
}

template <class A>
__device__ void stdHeapSortRange(A& p, int first, int last) {  // __partial_sort(first, last, last)
  const int len = last - first;
  if (len >= 2) {  // __make_heap
    int parent = (len - 2) / 2;
    for (;;) {
      stdAdjustHeap(p, first, parent, len, p.get(first + parent));
      if (parent == 0) break;
      parent--;
    }
  }
  int l = last;  // __sort_heap
  while (l - first > 1) {
    --l;
    const u64 v = p.get(l);  // __pop_heap(first, l, l)
    p.set(l, p.get(first));
    stdAdjustHeap(p, first, 0, l - first, v);
  }
}

template <class A>
__device__ void stdIntrosortLoop(A& p, int n, u64* stack /* LDS, >= 40 entries: at most 2*log2(n) <= 38 pending ranges */) {
  if (n <= 16) return;
  int sp = 0, first = 0, last = n;
  int depth = 2 * (31 - __builtin_clz((unsigned)n));
  for (;;) {
    while (last - first > 16) {
      if (depth == 0) { stdHeapSortRange(p, first, last); break; }
      --depth;
      const int mid = first + (last - first) / 2;
      {  // __move_median_to_first(first, first + 1, mid, last - 1)
        const int ia = first + 1, ib = mid, ic = last - 1;
        const u64 a = p.get(ia), b = p.get(ib), c = p.get(ic), f = p.get(first);
        int sel;
        if (SLESS(a, b)) sel = SLESS(b, c) ? ib : (SLESS(a, c) ? ic : ia);
        else sel = SLESS(a, c) ? ia : (SLESS(b, c) ? ic : ib);
        p.set(first, p.get(sel));
        p.set(sel, f);
      }
      const u64 pivot = p.get(first);
      int lo = first + 1, hi = last;
      for (;;) {  // __unguarded_partition(first + 1, last, first)
        u64 vlo = p.get(lo);
        while (SLESS(vlo, pivot)) vlo = p.get(++lo);
        --hi;
        u64 vhi = p.get(hi);
        while (SLESS(pivot, vhi)) vhi = p.get(--hi);
        if (!(lo < hi)) break;
        p.set(lo, vhi);
        p.set(hi, vlo);
        ++lo;
      }
      // the library recurses on [cut, last) and loops on [first, cut): disjoint ranges, same depth budget
      stack[sp++] = (u64)first | ((u64)lo << 20) | ((u64)depth << 40);  // private arrays would live in (slow) scratch memory
      first = lo;
    }
    if (sp == 0) break;
    const u64 top = stack[--sp];
    first = (int)(top & 0xfffff); last = (int)((top >> 20) & 0xfffff); depth = (int)(top >> 40);
  }
}

// partition phase of std::sort on p[0..n), replayed by one lane (a register-resident variant read through v_readlane
// was measured slower: a lone wave spends ~100 cycles per serial step either way).  Call with the whole workgroup.
__device__ void stdSortPartitionPhase(u64* p, int n, int tid) {
  __shared__ u64 sortStack[40];
  if (n <= 16 || tid != 0) return;
  MemKeys A{p};
  stdIntrosortLoop(A, n, sortStack);
}

// The same partition phase with the whole workgroup, for n <= OCT_PAR_MAX keys in LDS.  std::sort's recursion is a tree of
// disjoint ranges that are partitioned independently, so all ranges of one recursion level are processed together
// (breadth first), and one __unguarded_partition is evaluated in closed form: with L[k] = position of the k-th element that
// stops the upward scan (!(a < pivot), ascending) and R[k] = position of the k-th element that stops the downward scan
// (!(pivot < a), descending), the library swaps exactly the pairs (L[k], R[k]) with L[k] < R[k] -- a prefix k < K because L
// increases and R decreases -- and returns cut = min(L[K], R[K-1]) (the upward scan then stops at the next original stop
// or at the element the last swap put at R[K-1], whichever comes first).  Ranks come from two workgroup prefix sums.
// Median-of-3, the depth budget and the heapsort fallback stay literal, per range, on the range's owner thread.
// `scr` = OCT_PAR_SCR dwords of LDS, `ws` = 4 ints of LDS (blockScanExcl).
#define OCT_PAR_MAX 512
#define OCT_PAR_RANGES 64
// scratch dwords for at most capN keys (capN = 512 or 256): four u16 arrays of capN + 8 entries, then the range tables
#define OCT_PAR_SCR_FOR(capN) (4 * (((capN) + 8) / 2) + 2 * OCT_PAR_RANGES + 2 * OCT_PAR_RANGES + OCT_PAR_RANGES + 4)
#define OCT_PAR_SCR OCT_PAR_SCR_FOR(OCT_PAR_MAX)
__device__ void stdSortPartitionPhasePar(u64* p, int n, int tid, uint32_t* scr, int* ws, int capN) {
  if (n <= 16) return;
  const int seg = (capN + 8) / 2;                              // dwords per u16 array
  uint16_t* Lpos = reinterpret_cast<uint16_t*>(scr);           // [capN + 8]
  uint16_t* Rpos = reinterpret_cast<uint16_t*>(scr + seg);
  uint16_t* sl = reinterpret_cast<uint16_t*>(scr + 2 * seg);   // exclusive prefix of the upward-stop flags, [capN] = total
  uint16_t* sr = reinterpret_cast<uint16_t*>(scr + 3 * seg);   // exclusive prefix of the downward-stop flags
  uint32_t* rngA = scr + 4 * seg;                              // ranges: first | last << 10 | depth << 20, sorted by first
  uint32_t* rngB = rngA + OCT_PAR_RANGES;
  u64* rPivot = reinterpret_cast<u64*>(rngB + OCT_PAR_RANGES);  // [OCT_PAR_RANGES]
  int* rK = reinterpret_cast<int*>(rngB + OCT_PAR_RANGES + 2 * OCT_PAR_RANGES);  // swaps of the range; -1 = not partitioned
  int* sN = rK + OCT_PAR_RANGES;
  MemKeys A{p};
  if (tid == 0) {
    rngA[0] = 0u | ((uint32_t)n << 10) | ((uint32_t)(2 * (31 - __builtin_clz((unsigned)n))) << 20);
    *sN = 1;
  }
  __syncthreads();
  uint32_t* cur = rngA;
  uint32_t* nxt = rngB;
  for (;;) {
    const int nAct = *sN;
    if (nAct == 0) break;
    // ---- owners: depth check, median of three, pivot ----
    if (tid < nAct) {
      const uint32_t rg = cur[tid];
      const int first = rg & 1023, last = (rg >> 10) & 1023, depth = (int)(rg >> 20);
      if (depth == 0) {
        stdHeapSortRange(A, first, last);
        rK[tid] = -1;
      } else {
        const int mid = first + (last - first) / 2;
        const int ia = first + 1, ib = mid, ic = last - 1;  // __move_median_to_first(first, first + 1, mid, last - 1)
        const u64 a = p[ia], b = p[ib], c = p[ic], f = p[first];
        int sel;
        if (SLESS(a, b)) sel = SLESS(b, c) ? ib : (SLESS(a, c) ? ic : ia);
        else sel = SLESS(a, c) ? ia : (SLESS(b, c) ? ic : ib);
        const u64 pv = p[sel];
        p[first] = pv;
        p[sel] = f;
        rPivot[tid] = pv;
        rK[tid] = 0;
      }
    }
    __syncthreads();
    // ---- every position: its range and its stop flags ----
    int myR[2], gl[2], ll[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int i = 2 * tid + j;
      int lo = 0, hi = nAct;  // last range whose first <= i
      while (hi - lo > 1) {
        const int m = (lo + hi) >> 1;
        if ((int)(cur[m] & 1023) <= i) lo = m; else hi = m;
      }
      const uint32_t rg = cur[lo];
      const int first = rg & 1023, last = (rg >> 10) & 1023;
      const bool act = i > first && i < last && rK[lo] >= 0;
      myR[j] = act ? lo : -1;
      gl[j] = 0; ll[j] = 0;
      if (act) {
        const u64 v = p[i], pv = rPivot[lo];
        gl[j] = !SLESS(v, pv);
        ll[j] = !SLESS(pv, v);
      }
    }
    int totL, totR;
    const int exL = blockScanExcl(gl[0] + gl[1], tid, ws, &totL);
    const int exR = blockScanExcl(ll[0] + ll[1], tid, ws, &totR);
    if (2 * tid < capN) {  // (flags beyond n are 0: the prefix stays at the total there)
      sl[2 * tid] = (uint16_t)exL; sl[2 * tid + 1] = (uint16_t)(exL + gl[0]);
      sr[2 * tid] = (uint16_t)exR; sr[2 * tid + 1] = (uint16_t)(exR + ll[0]);
    }
    if (tid == 0) { sl[capN] = (uint16_t)totL; sr[capN] = (uint16_t)totR; }
    __syncthreads();
    // ---- scatter the stop positions by rank: upward stops ascending, downward stops descending ----
#pragma unroll
    for (int j = 0; j < 2; j++) {
      if (myR[j] < 0) continue;
      const int i = 2 * tid + j;
      const uint32_t rg = cur[myR[j]];
      const int first = rg & 1023, last = (rg >> 10) & 1023;
      if (gl[j]) Lpos[first + 1 + ((int)sl[i] - (int)sl[first + 1])] = (uint16_t)i;
      if (ll[j]) {
        const int nR = (int)sr[last] - (int)sr[first + 1];
        Rpos[first + 1 + (nR - 1 - ((int)sr[i] - (int)sr[first + 1]))] = (uint16_t)i;
      }
    }
    __syncthreads();
    // ---- the swaps: pair k of the range lives at index first + 1 + k ----
#pragma unroll
    for (int j = 0; j < 2; j++) {
      if (myR[j] < 0) continue;
      const int i = 2 * tid + j;
      const uint32_t rg = cur[myR[j]];
      const int first = rg & 1023, last = (rg >> 10) & 1023;
      const int k = i - (first + 1);
      const int nL = (int)sl[last] - (int)sl[first + 1], nR = (int)sr[last] - (int)sr[first + 1];
      if (k < min(nL, nR)) {
        const int a = Lpos[i], b = Rpos[i];
        if (a < b) {
          const u64 va = p[a], vb = p[b];
          p[a] = vb;
          p[b] = va;
          atomicAdd(&rK[myR[j]], 1);
        }
      }
    }
    __syncthreads();
    // ---- owners: cut point and the ranges of the next level (both halves inherit depth - 1) ----
    int cnt = 0, cFirst = 0, cCut = 0, cLast = 0, cDepth = 0;
    if (tid < nAct && rK[tid] >= 0) {
      const uint32_t rg = cur[tid];
      cFirst = rg & 1023; cLast = (rg >> 10) & 1023; cDepth = (int)(rg >> 20) - 1;
      const int K = rK[tid];
      const int nL = (int)sl[cLast] - (int)sl[cFirst + 1], nR = (int)sr[cLast] - (int)sr[cFirst + 1];
      const int cutL = K < nL ? (int)Lpos[cFirst + 1 + K] : 4096;
      const int cutR = (K > 0 && K - 1 < nR) ? (int)Rpos[cFirst + K] : 4096;
      cCut = min(cutL, cutR);
      cnt = (cCut - cFirst > 16) + (cLast - cCut > 16);
    }
    int tot;
    int base = blockScanExcl(cnt, tid, ws, &tot);
    if (cnt) {
      if (cCut - cFirst > 16) nxt[base++] = (uint32_t)cFirst | ((uint32_t)cCut << 10) | ((uint32_t)cDepth << 20);
      if (cLast - cCut > 16) nxt[base] = (uint32_t)cCut | ((uint32_t)cLast << 10) | ((uint32_t)cDepth << 20);
    }
    if (tid == 0) *sN = tot;
    __syncthreads();
    uint32_t* tmp = cur; cur = nxt; nxt = tmp;
  }
}

// reference candidate order (cell row, cell col, y, x) of a packed candidate (cpp:1078-1137; cv::FAST is row-major)
__device__ __forceinline__ u64 candRank(uint32_t e, const OctLevel& L) {
  const int x = e & 0xfff, y = (e >> 12) & 0xfff;
  const int cr = max(y - 3, 0) / L.hCell, cc = max(x - 3, 0) / L.wCell;
  return ((u64)(cr * L.nCols + cc) << 24) | ((u64)y << 12) | (u64)x;
}

__device__ __forceinline__ void rootRect(const OctLevel& L, int root, int& ulx, int& uly, int& brx, int& bry) {
  ulx = (int)(L.hX * (float)root);
  brx = (int)(L.hX * (float)(root + 1));
  uly = 0;
  bry = L.height;
}

// Lists the candidates of one (frame, level) unit: k_fast writes every cell's survivors into the cell's own segment
// (without atomics); dst[0 .. n) receives the positions (cell * segCap + k) of the n survivors inside the unit's candidate
// area, cells in index order.  The position doubles as the candidate's index in the sort keys, so the candidates themselves
// are read from the segments where they lie (L2) and never copied.  Returns n to every thread; nothing is written when
// n > cap.  `ws` = 4 ints of LDS.  Call with the whole workgroup.
__device__ int gatherCandidates(const int* __restrict__ cellCnt, int nCells, int segCap, uint32_t* dst, int cap, int tid, int* ws) {
  const int chunk = (nCells + OCT_T - 1) / OCT_T;
  const int b = min(tid * chunk, nCells), e = min(b + chunk, nCells);
  int c = 0;
  for (int i = b; i < e; i++) c += cellCnt[i];
  int n;
  int pos = blockScanExcl(c, tid, ws, &n);
  if (n <= cap)
    for (int i = b; i < e; i++) {
      const int k = chunk == 1 ? c : cellCnt[i];
      for (int j = 0; j < k; j++) dst[pos++] = (uint32_t)(i * segCap + j);
    }
  __syncthreads();
  return n;
}

// The whole selection for one (frame, level).  S.cand: n unordered packed candidates.  Writes min(#nodes, quota) SelKp
// records (list order) to `out` and the count to *nOut (-2: scratch too small for this unit).
__device__ void octreeSelect(const OctScratch S, int n, const OctLevel L, int level, SelKp* __restrict__ out,
                             int* __restrict__ nOut, int mCap, int fCap, int qCap) {
  __shared__ int cntDiv[OCT_DEPTH + 2], cntAlone[OCT_DEPTH + 2];
  __shared__ int sK, sPhase2, sM, sFront, sSize, sCut, sFinish;
  __shared__ int ws[OCT_T / 64];
  const int tid = threadIdx.x;
  const int N = L.quota;
  if (n <= 0 || N <= 0) {  // nothing to select (an empty quota truncates everything, cpp:1159-1161)
    if (tid == 0) *nOut = 0;
    return;
  }
  int nPad = 1;
  while (nPad < n) nPad <<= 1;
  if (nPad < 256) nPad = 256;  // the register sort works on at least 256 (padded) keys
  OCT_STAMP(0);

  // ---- 1. path codes -----------------------------------------------------------------------------------------
  for (int i = tid; i < nPad; i += OCT_T) {
    u64 key = ~0ull;
    if (i < n) {
      const uint32_t gi = S.cand[i];  // position of candidate i in the unit's candidate area
      const uint32_t e = S.segBase[gi];
      const float x = (float)(e & 0xfff), y = (float)((e >> 12) & 0xfff);
      int root = (int)(x / L.hX);  // cpp:747
      root = min(max(root, 0), L.nIni - 1);
      int ulx, uly, brx, bry;
      rootRect(L, root, ulx, uly, brx, bry);
      u64 code = (u64)root;
#pragma unroll 4
      for (int d = 0; d < OCT_DEPTH; d++) {  // DivideNode, cpp:617-676: half = ceil(extent / 2)
        const int midX = ulx + ((brx - ulx + 1) >> 1), midY = uly + ((bry - uly + 1) >> 1);
        const int qx = !(x < (float)midX), qy = !(y < (float)midY);
        if (qx) ulx = midX; else brx = midX;
        if (qy) uly = midY; else bry = midY;
        code = (code << 2) | (u64)(qy * 2 + qx);
      }
      key = (code << 24) | (u64)gi;  // gi grows with i: same order among equal codes as the list index
    }
    S.keys[i] = key;
  }
  if (tid < OCT_DEPTH + 2) { cntDiv[tid] = 0; cntAlone[tid] = 0; }
  __syncthreads();
  OCT_STAMP(1);
  if (nPad > OCT_SORT_LDS && S.sortTmp && S.xchg) {
    const int rootBits = L.nIni > 1 ? 32 - __builtin_clz((unsigned)(L.nIni - 1)) : 0;
    radixSortCodes(S.keys, S.sortTmp, n, tid, reinterpret_cast<uint32_t*>(S.xchg), ws, 2 * OCT_DEPTH + rootBits);
  } else {
    bitonicSort(S.keys, nPad, tid, S.xchg);
  }
  OCT_STAMP(2);

  // ---- 2. divergence depths, S_d (distinct depth-d prefixes), singles_d -----------------------------------------
  for (int i = tid; i <= n; i += OCT_T) {
    int d = 255;
    if (i > 0 && i < n) {
      d = divDepth(S.keys[i - 1] >> 24, S.keys[i] >> 24);
      atomicAdd(&cntDiv[d], 1);
    }
    S.div[i] = (uint8_t)d;
  }
  __syncthreads();
  for (int i = tid; i < n; i += OCT_T) {
    const int dl = S.div[i] == 255 ? -1 : (int)S.div[i], dr = S.div[i + 1] == 255 ? -1 : (int)S.div[i + 1];
    const int a = max(max(dl, dr), 0);
    S.alone[i] = (uint8_t)a;
    atomicAdd(&cntAlone[a], 1);
  }
  __syncthreads();
  OCT_STAMP(3);
  // ---- 3. replay the pass loop on sizes only (cpp:781-895) ------------------------------------------------------
  if (tid == 0) {
    int Sd[OCT_DEPTH + 2], sg[OCT_DEPTH + 2];
    int accD = 0, accA = 0;
    for (int d = 0; d <= OCT_DEPTH; d++) {
      accD += cntDiv[d];
      accA += cntAlone[d];
      Sd[d] = 1 + accD;
      sg[d] = accA;
    }
    Sd[OCT_DEPTH + 1] = Sd[OCT_DEPTH];
    sg[OCT_DEPTH + 1] = sg[OCT_DEPTH];
    int k = 0, phase2 = 0;
    for (;;) {
      const int prevSize = Sd[k];
      if (k < OCT_DEPTH) k++;
      const int size = Sd[k], nToExpand = size - sg[k];
      if (size >= N || size == prevSize) break;
      if (size + 3 * nToExpand > N) { phase2 = 1; break; }
    }
    sK = k;
    sPhase2 = phase2;
    sM = Sd[k];
    sFront = 0;
  }
  __syncthreads();
  const int k = sK;
  const int M = sM;  // nodes in the list after k full passes
  if (M > mCap) {    // cannot happen for capacities sized from the quota (M < 4N) unless N < nIni; guard anyway
    if (tid == 0) *nOut = -2;
    return;
  }
  OCT_STAMP(4);
  // ---- 4. node list in std::list order ---------------------------------------------------------------------------
  // node starts: a leaf key (alone < k) or the first key of a depth-k group
  {
    const int chunk = (n + OCT_T - 1) / OCT_T;
    const int b = min(tid * chunk, n), e = min(b + chunk, n);
    int c = 0;
    for (int i = b; i < e; i++) c += (S.alone[i] < k) || (S.div[i] == 255 || (int)S.div[i] <= k);
    int tot;
    int m = blockScanExcl(c, tid, ws, &tot);
    for (int i = b; i < e; i++) {
      const bool leaf = S.alone[i] < k;
      if (leaf || (S.div[i] == 255 || (int)S.div[i] <= k)) {
        const int j = leaf ? (int)S.alone[i] : k;  // block depth
        const u64 code = S.keys[i] >> 24;
        // order key: first j digits, digit m flipped when (j - m) is even, root flipped when j is odd
        const u64 prefix = code >> (2 * (OCT_DEPTH - j));
        u64 flip = 0;
        for (int mm = j; mm >= 1; mm -= 2) flip |= (u64)3 << (2 * (j - mm));
        u64 okey = prefix ^ flip;
        if (j & 1) {
          const u64 root = okey >> (2 * j);
          okey = (okey & (((u64)1 << (2 * j)) - 1)) | ((u64)(255 - root) << (2 * j));
        }
        S.nodes[m++] = ((u64)(OCT_DEPTH + 1 - j) << 59) | (okey << 19) | (u64)i;
      }
    }
    __syncthreads();
  }
  int mPad = 1;
  while (mPad < M) mPad <<= 1;
  if (mPad < 256) mPad = 256;
  for (int m = tid; m < M; m += OCT_T) {  // ends of the nodes, from the position-ordered list before it is re-sorted
    const int lo = (int)(S.nodes[m] & 0x7ffff);
    const int hi = (m + 1 < M) ? (int)(S.nodes[m + 1] & 0x7ffff) : n;
    S.hiOf[lo] = (uint32_t)hi;
  }
  __syncthreads();
  for (int m = M + tid; m < mPad; m += OCT_T) S.nodes[m] = ~0ull;
  __syncthreads();
  OCT_STAMP(5);
  bitonicSort(S.nodes, mPad, tid, S.xchg);
  OCT_STAMP(6);
  for (int m = tid; m < M; m += OCT_T) {
    const u64 v = S.nodes[m];
    const int lo = (int)(v & 0x7ffff);
    S.nodeLo[m] = lo;
    S.nodeHi[m] = (int)S.hiOf[lo];
    S.nodeDepth[m] = (uint8_t)(OCT_DEPTH + 1 - (int)(v >> 59));
    S.nodeAlive[m] = 1;
  }
  __syncthreads();

  OCT_STAMP(7);
#ifdef ORBX_OCT_STAMPS
  unsigned long long tAcc = __builtin_amdgcn_s_memtime();
  if (tid == 0) for (int k_ = 8; k_ < 14; k_++) g_octStamps[((blockIdx.y * gridDim.x + blockIdx.x) & 4095) * OCT_NSTAMP + k_] = 0;
#endif
  // ---- 5. partial pass(es), cpp:897-965 ---------------------------------------------------------------------------
  if (sPhase2) {
    int* pendA = S.pending;
    int* pendB = S.pending + qCap;
    u64* sizedA = S.sized;
    u64* sizedB = S.sized + qCap;
    // pending = multi-key depth-k nodes in creation order = reverse of the depth-k block order
    int nPend;
    {
      const int chunk = (M + OCT_T - 1) / OCT_T;
      const int b = min(tid * chunk, M), e = min(b + chunk, M);
      int c = 0;
      for (int r = b; r < e; r++) {  // r = reversed list index
        const int i = M - 1 - r;
        c += (S.nodeDepth[i] == k) && (S.nodeHi[i] - S.nodeLo[i] > 1);
      }
      int p = blockScanExcl(c, tid, ws, &nPend);
      if (nPend <= qCap)
        for (int r = b; r < e; r++) {
          const int i = M - 1 - r;
          if ((S.nodeDepth[i] == k) && (S.nodeHi[i] - S.nodeLo[i] > 1)) pendA[p++] = i;
        }
    }
    if (tid == 0) { sSize = M; sFinish = 0; }
    __syncthreads();
    if (nPend > qCap) {  // E_k < N <= qCap always; guard anyway
      if (tid == 0) *nOut = -2;
      return;
    }
    int nFront = 0;
    for (;;) {
      // (a) sort keys: count << 40 | UL.x << 20 | node
      for (int j = tid; j < nPend; j += OCT_T) {
        const int nd = pendA[j];
        const int lo = S.nodeLo[nd];
        const u64 code = S.keys[lo] >> 24;
        int ulx, uly, brx, bry;
        rootRect(L, (int)(code >> (2 * OCT_DEPTH)), ulx, uly, brx, bry);
        const int depth = S.nodeDepth[nd];
        for (int d = 1; d <= depth; d++) {
          const int q = (int)((code >> (2 * (OCT_DEPTH - d))) & 3);
          const int halfX = (brx - ulx + 1) >> 1, halfY = (bry - uly + 1) >> 1;
          if (q & 1) ulx += halfX; else brx = ulx + halfX;
          if (q & 2) uly += halfY; else bry = uly + halfY;
        }
        sizedA[j] = ((u64)(S.nodeHi[nd] - lo) << 40) | ((u64)(ulx & 0xfffff) << 20) | (u64)nd;
      }
      __syncthreads();
      OCT_STAMP_ACC(8, tAcc);
      // (b) std::sort (cpp:912): partition phase on one lane, final insertion sort as a parallel stable rank sort
      if (nPend <= S.parCap && S.parScr) stdSortPartitionPhasePar(sizedA, nPend, tid, S.parScr, ws, S.parCap);
      else stdSortPartitionPhase(sizedA, nPend, tid);
      __syncthreads();
      OCT_STAMP_ACC(9, tAcc);
      for (int j = tid; j < nPend; j += OCT_T) {
        const u64 v = sizedA[j];
        const u64 kv = v >> 20;
        int rank = 0;
        for (int i = 0; i < nPend; i++) {
          const u64 ki = sizedA[i] >> 20;
          rank += (ki < kv) || (ki == kv && i < j);
        }
        sizedB[rank] = v;
      }
      __syncthreads();
      OCT_STAMP_ACC(10, tAcc);
      // (c) children of every pending node (in sorted order)
      for (int j = tid; j < nPend; j += OCT_T) {
        const int nd = (int)(sizedB[j] & 0xfffff);
        const int lo = S.nodeLo[nd], hi = S.nodeHi[nd], pd = S.nodeDepth[nd];
        int nch = 1, nmulti = 1;
        if (pd < OCT_DEPTH) {
          nch = 0;
          nmulti = 0;
          int start = lo;
          for (int i = lo + 1; i <= hi; i++)
            if (i == hi || (int)S.div[i] <= pd + 1) {
              nch++;
              nmulti += (i - start > 1);
              start = i;
            }
        }
        S.childCnt[j] = nch | (nmulti << 8);
      }
      __syncthreads();
      OCT_STAMP_ACC(11, tAcc);
      // (d) cut point: nodes are split from the back of the sorted array until the list holds N nodes
      {
        // growth of the list when the last t+1 sorted nodes are split: inclusive prefix over t = nPend-1-j
        const int prevSize = sSize;
        const int chunk = (nPend + OCT_T - 1) / OCT_T;
        const int b = min(tid * chunk, nPend), e = min(b + chunk, nPend);
        int c = 0;
        for (int t = b; t < e; t++) c += (S.childCnt[nPend - 1 - t] & 0xff) - 1;
        int totalGrowth;
        int acc = blockScanExcl(c, tid, ws, &totalGrowth);
        if (tid == 0) { sCut = 0; sSize = prevSize + totalGrowth; sFinish = (totalGrowth == 0 || prevSize + totalGrowth >= N) ? 1 : 0; }
        __syncthreads();
        // growth is >= 0 per split, so the first t whose inclusive sum reaches N - prevSize is unique: that thread reports
        for (int t = b; t < e; t++) {
          const int before = acc;
          acc += (S.childCnt[nPend - 1 - t] & 0xff) - 1;
          if (prevSize + before < N && prevSize + acc >= N) { sCut = nPend - 1 - t; sSize = prevSize + acc; sFinish = 1; }
        }
      }
      __syncthreads();
      const int cut = sCut, finish = sFinish;
      OCT_STAMP_ACC(12, tAcc);
      // (e) create the children: processing order t = nPend-1-j; children are push_front'ed in quadrant order
      {
        const int nProc = nPend - cut;
        const int chunk = (nProc + OCT_T - 1) / OCT_T;
        const int b = min(tid * chunk, nProc), e = min(b + chunk, nProc);
        int cc = 0, cm = 0;
        for (int t = b; t < e; t++) {
          const int v = S.childCnt[nPend - 1 - t];
          cc += v & 0xff;
          cm += v >> 8;
        }
        int totC, totM;
        int offC = blockScanExcl(cc, tid, ws, &totC);
        int offM = blockScanExcl(cm, tid, ws, &totM);
        if (nFront + totC > fCap || totM > qCap) {  // scratch too small: the caller re-runs the unit with more
          if (tid == 0) *nOut = -2;
          return;
        }
        for (int t = b; t < e; t++) {
          const int j = nPend - 1 - t;
          const int nd = (int)(sizedB[j] & 0xfffff);
          const int lo = S.nodeLo[nd], hi = S.nodeHi[nd], pd = S.nodeDepth[nd];
          if (pd >= OCT_DEPTH) {  // coincident keys: cannot be split further, stays one node
            const int id = mCap + nFront + offC++;
            S.nodeLo[id] = lo; S.nodeHi[id] = hi; S.nodeDepth[id] = (uint8_t)pd; S.nodeAlive[id] = 1;
            pendB[offM++] = id;
          } else {
            int start = lo;
            for (int i = lo + 1; i <= hi; i++)
              if (i == hi || (int)S.div[i] <= pd + 1) {
                const int id = mCap + nFront + offC++;
                S.nodeLo[id] = start; S.nodeHi[id] = i; S.nodeDepth[id] = (uint8_t)(pd + 1); S.nodeAlive[id] = 1;
                if (i - start > 1) pendB[offM++] = id;
                start = i;
              }
          }
          S.nodeAlive[nd] = 0;
        }
        nFront += totC;
        nPend = totM;
      }
      __syncthreads();
      OCT_STAMP_ACC(13, tAcc);
      if (finish) break;
      { int* t = pendA; pendA = pendB; pendB = t; }
    }
    if (tid == 0) sFront = nFront;
    __syncthreads();
  }
  OCT_STAMP(14);
  // ---- 6. output positions: reverse(front alive) ++ list alive; keep the first `quota` ------------------------------
  const int nFront = sFront;
  const int total = nFront + M;  // virtual sequence: pushed nodes in reverse push order, then the list
  {
    const int chunk = (total + OCT_T - 1) / OCT_T;
    const int b = min(tid * chunk, total), e = min(b + chunk, total);
    auto nodeAt = [&](int v) { return v < nFront ? mCap + (nFront - 1 - v) : v - nFront; };
    int c = 0;
    for (int v = b; v < e; v++) c += S.nodeAlive[nodeAt(v)];
    int alive;
    int p = blockScanExcl(c, tid, ws, &alive);
    for (int v = b; v < e && p < N; v++) {
      const int nd = nodeAt(v);
      if (!S.nodeAlive[nd]) continue;
      // first key with the highest response (cpp:984-1007); "first" = reference candidate order
      const int lo = S.nodeLo[nd], hi = S.nodeHi[nd];
      uint32_t bestE = S.segBase[(int)(S.keys[lo] & 0xffffff)];
      if (hi - lo > 1) {
        u64 bestRank = candRank(bestE, L);
        for (int i = lo + 1; i < hi; i++) {
          const uint32_t e2 = S.segBase[(int)(S.keys[i] & 0xffffff)];
          const uint32_t s1 = bestE >> 24, s2 = e2 >> 24;
          if (s2 < s1) continue;
          const u64 r2 = candRank(e2, L);
          if (s2 > s1 || r2 < bestRank) { bestE = e2; bestRank = r2; }
        }
      }
      SelKp kp;
      kp.x = (uint16_t)((bestE & 0xfff) + ORBX_MIN_BORDER);  // cpp:1171-1172
      kp.y = (uint16_t)(((bestE >> 12) & 0xfff) + ORBX_MIN_BORDER);
      kp.level = (uint8_t)level;
      kp.response = (uint8_t)(bestE >> 24);
      kp.pad = 0;
      out[p++] = kp;
    }
    if (tid == 0) *nOut = min(alive, N);
  }
  OCT_STAMP(15);
}

// One (frame, level) unit on the global-scratch layout (octScratchBytes()); `xchg` = OCT_SORT_LDS u64 of LDS for the sorts.
// Scratch of unit (f, level) starts at scrOff[level] + f * scrStride[level].
__device__ void octreeGlobalUnit(const uint32_t* __restrict__ cand, const int* __restrict__ cellCount, const OctLaunch& P,
                                 SelKp* __restrict__ selStage, int* __restrict__ nselLevel, uint8_t* __restrict__ scratch,
                                 int level, int f, u64* xchg) {
  __shared__ int gws[OCT_T / 64];
  int* nOut = &nselLevel[f * P.nlevels + level];
  const int nMax = P.scrNMax[level], qMax = max(P.lev[level].quota, 1);
  size_t nPad = 256;
  while ((int)nPad < nMax) nPad <<= 1;
  const int mCap = 4 * qMax, fCap = 16 * qMax;
  size_t mPad = 256;
  while ((int)mPad < mCap) mPad <<= 1;
  uint8_t* p = scratch + P.scrOff[level] + (int64_t)f * P.scrStride[level];
  OctScratch S;
  S.keys = (u64*)p; p += nPad * 8;
  S.nodes = (u64*)p; p += mPad * 8;
  S.sized = (u64*)p; p += (size_t)2 * qMax * 8;
  S.hiOf = (uint32_t*)p; p += nPad * 4;
  S.nodeLo = (int*)p; p += (size_t)(mCap + fCap) * 4;
  S.nodeHi = (int*)p; p += (size_t)(mCap + fCap) * 4;
  S.pending = (int*)p; p += (size_t)2 * qMax * 4;
  S.childCnt = (int*)p; p += (size_t)qMax * 4;
  S.div = p; p += nPad + 8;
  S.alone = p; p += nPad + 8;
  S.nodeDepth = p; p += (size_t)(mCap + fCap + 8);
  S.nodeAlive = p; p += (size_t)(mCap + fCap + 8);
  p = (uint8_t*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
  uint32_t* candBuf = (uint32_t*)p;  // [nPad] positions of the unit's candidates
  p += nPad * 4;
  S.sortTmp = (u64*)p;               // [nPad] (p stays 16-byte aligned: nPad is a multiple of 256)
  S.segBase = cand + P.candOff[level] + (int64_t)f * P.candCap[level];
  const int n = gatherCandidates(cellCount + (int64_t)f * P.nCellsTotal + P.lev[level].cellBase, P.lev[level].nCells,
                                 P.lev[level].segCap, candBuf, nMax, threadIdx.x, gws);
  if (n > nMax) {  // more candidates than the selection stage can index (2^19 - 1)
    if (threadIdx.x == 0) *nOut = -1;
    return;
  }
  S.cand = candBuf;
  S.xchg = xchg;
  S.parScr = reinterpret_cast<uint32_t*>(xchg);  // the sort exchange buffer is idle during the partial pass
  S.parCap = OCT_PAR_MAX;
  octreeSelect(S, n, P.lev[level], level, selStage + (int64_t)f * P.selStride + P.selOff[level], nOut, mCap, fCap, qMax);
  __syncthreads();
  if (threadIdx.x == 0 && *nOut == -2) *nOut = -1;  // even the large scratch was too small: hard error
}

// ---- kernels ---------------------------------------------------------------------------------------------------------

// LDS-resident variant: n <= NMAX candidates, quota <= QMAX
template <int NMAX, int QMAX>
__global__ __launch_bounds__(OCT_T) void k_octree_lds(const uint32_t* __restrict__ cand, const int* __restrict__ cellCount,
                                                     const OctLaunch P, SelKp* __restrict__ selStage,
                                                     int* __restrict__ nselLevel, uint8_t* __restrict__ scratch,
                                                     int* __restrict__ maxN) {
  constexpr int MCAP = 4 * QMAX, FCAP = 2 * QMAX;
  static_assert((NMAX & (NMAX - 1)) == 0 && (MCAP & (MCAP - 1)) == 0, "sort buffers must be powers of two");
  // LDS budget (QMAX 256): NMAX 2048: 16 + 8 + 8 + 12 + 4 + 3 KB = 51 KB -> three workgroups per CU;
  //                        NMAX 1024:  8 + 8 + 4 + 12 + 2 + 3 KB = 37 KB -> four (launch_octree picks the instance).
  //   nodes[] is dead once the node records exist, so the partial pass's buffers (sized, pending, childCnt) live in it;
  //   hiOf[] (steps 3-4) shares its space with the candidate position list (step 1 only).
  __shared__ u64 keysNodes[NMAX + MCAP];
  u64* keys = keysNodes;
  u64* nodes = keysNodes + NMAX;
  __shared__ uint32_t candL[NMAX];
  __shared__ int nodeLo[MCAP + FCAP], nodeHi[MCAP + FCAP];
  __shared__ uint8_t div[NMAX + 4], alone[NMAX];
  __shared__ uint8_t nodeDepth[MCAP + FCAP], nodeAlive[MCAP + FCAP];
  static_assert(2 * QMAX * 8 + 2 * QMAX * 4 + QMAX * 4 <= MCAP * 8, "partial-pass buffers must fit in nodes[]");
  u64* sized = nodes;                                             // [2 * QMAX]
  int* pending = reinterpret_cast<int*>(nodes + 2 * QMAX);        // [2 * QMAX]
  int* childCnt = pending + 2 * QMAX;                             // [QMAX]
  uint32_t* hiOf = candL;
  const int level = blockIdx.y, f = blockIdx.x + P.frame0;  // level-major dispatch, see launch_octree
  int* nOut = &nselLevel[f * P.nlevels + level];
  static_assert(NMAX + MCAP >= OCT_SORT_LDS, "keys[] + nodes[] double as the sort exchange buffer of the global-scratch path");
  __shared__ int redo;
  __shared__ int gws[OCT_T / 64];
  const uint32_t* segBase = cand + P.candOff[level] + (int64_t)f * P.candCap[level];
  const int* cellCnt = cellCount + (int64_t)f * P.nCellsTotal + P.lev[level].cellBase;
  const int n = gatherCandidates(cellCnt, P.lev[level].nCells, P.lev[level].segCap, candL, NMAX, threadIdx.x, gws);
  if (threadIdx.x == 0 && maxN) atomicMax(&maxN[level], n);  // feedback for the next batch's choice of instance
  if (n <= NMAX && P.lev[level].quota <= QMAX) {
    constexpr int PARCAP = OCT_PAR_SCR_FOR(512) <= NMAX ? 512 : 256;  // what fits into candL[NMAX]
    static_assert(OCT_PAR_SCR_FOR(PARCAP) <= NMAX, "parallel-replay scratch must fit into the position list's space");
    OctScratch S{keys, nodes, div, alone, hiOf, nodeLo, nodeHi, nodeDepth, nodeAlive, sized, pending, childCnt, candL, segBase, nullptr,
                 candL /* hiOf / position list space: dead during the partial pass */, PARCAP};
    octreeSelect(S, n, P.lev[level], level, selStage + (int64_t)f * P.selStride + P.selOff[level], nOut, MCAP, FCAP, QMAX);
    __syncthreads();
    if (threadIdx.x == 0) redo = (*nOut == -2);  // a node table overflowed the LDS layout
    __syncthreads();
    if (!redo) return;
  }
  // the unit does not fit the LDS layout: same workgroup, global scratch (no second kernel on the stream's critical path)
  __syncthreads();
  octreeGlobalUnit(cand, cellCount, P, selStage, nselLevel, scratch, level, f, keysNodes);
}

// global-scratch variant for the (frame, level) units the LDS variant left (nselLevel == -2), or for all units when
// `all` is set.  Scratch of unit (f, level) starts at scrOff[level] + f * scrStride[level]; layout: octScratchBytes().
__global__ __launch_bounds__(OCT_T) void k_octree_global(const uint32_t* __restrict__ cand, const int* __restrict__ cellCount,
                                                        const OctLaunch P, SelKp* __restrict__ selStage,
                                                        int* __restrict__ nselLevel, uint8_t* __restrict__ scratch, int all) {
  __shared__ u64 xchg[OCT_SORT_LDS];
  const int level = blockIdx.y, f = blockIdx.x + P.frame0;  // level-major dispatch, see launch_octree
  if (!all && nselLevel[f * P.nlevels + level] != -2) return;
  octreeGlobalUnit(cand, cellCount, P, selStage, nselLevel, scratch, level, f, xchg);
}

size_t octScratchBytes(int nMax, int qMax) {
  size_t nPad = 256;
  while ((int)nPad < nMax) nPad <<= 1;
  qMax = qMax < 1 ? 1 : qMax;
  const size_t mCap = 4 * (size_t)qMax, fCap = 16 * (size_t)qMax;
  size_t mPad = 256;
  while (mPad < mCap) mPad <<= 1;
  size_t b = nPad * 8 + mPad * 8 + (size_t)2 * qMax * 8 + nPad * 4 + (mCap + fCap) * 8 + (size_t)3 * qMax * 4 + 2 * (nPad + 8) +
             2 * (mCap + fCap + 8) + 16 + nPad * 4 /* gathered candidates */ + nPad * 8 /* radix sort buffer */;
  return (b + 255) / 256 * 256;
}

// compacts the per-level staging lists of every frame into level-major order and writes the per-frame totals
__global__ __launch_bounds__(256) void k_sel_compact(const SelKp* __restrict__ selStage, const int* __restrict__ nselLevel,
                                                     const OctLaunch P, SelKp* __restrict__ sel, int* __restrict__ nsel,
                                                     int selCap, int* __restrict__ err, int* __restrict__ maxN,
                                                     int* __restrict__ hostMaxN) {
  const int f = blockIdx.x + P.frame0;
  // per-level candidate maxima of this launch go to pinned host memory and are reset for the next one
  if (blockIdx.x == 0 && maxN && threadIdx.x < P.nlevels) {
    hostMaxN[threadIdx.x] = maxN[threadIdx.x];
    maxN[threadIdx.x] = 0;
  }
  __shared__ int off[ORBX_MAX_LEVELS + 1];
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int l = 0; l < P.nlevels; l++) {
      off[l] = acc;
      const int c = nselLevel[f * P.nlevels + l];
      if (c < 0) *err = 1;
      acc += max(c, 0);
    }
    off[P.nlevels] = acc;
    nsel[f] = acc;
  }
  __syncthreads();
  for (int l = 0; l < P.nlevels; l++) {
    const int c = off[l + 1] - off[l];
    const SelKp* src = selStage + (int64_t)f * P.selStride + P.selOff[l];
    SelKp* dst = sel + (int64_t)f * selCap + off[l];
    for (int i = threadIdx.x; i < c; i += 256) dst[i] = src[i];
  }
}

hipError_t launch_octree(hipStream_t st, int nFrames, const uint32_t* cand, const int* cellCount, const OctLaunch& P,
                         SelKp* selStage, int* nselLevel, uint8_t* scratch, int maxQuota, int* maxN, int nHint) {
  // workgroups are dispatched x-fastest: all frames of level 0 first, then level 1, ...  The units of the lowest levels
  // run longest (most candidates, largest quota), so this is longest-processing-time-first and keeps the tail short.
  dim3 grid(nFrames, P.nlevels, 1), block(OCT_T, 1, 1);
  // the LDS variant handles the units it cannot take (more than NMAX candidates, quota above 256, node-table overflow)
  // itself on global scratch, so it is launched whatever the largest quota is: the higher levels of a large configuration
  // still fit.
  // nHint = largest candidate count of a unit in the previous batch (0 = unknown): with 6 % headroom below 1024 the
  // smaller instance runs four workgroups per CU instead of three; a unit that outgrows it is still handled correctly
  // (global scratch), only slower
  static const bool noSmall = getenv("ORBX_OCT_NO_SMALL") != nullptr;  // diagnostics: always the 2048-candidate instance
  if (maxQuota >= (1 << 30))  // test hook (orbx_debug_distribute_device variant 1): every unit on global scratch
    hipLaunchKernelGGL(k_octree_global, grid, block, 0, st, cand, cellCount, P, selStage, nselLevel, scratch, 1);
  else if (nHint > 0 && nHint <= 960 && !noSmall)
    hipLaunchKernelGGL((k_octree_lds<1024, 256>), grid, block, 0, st, cand, cellCount, P, selStage, nselLevel, scratch, maxN);
  else
    hipLaunchKernelGGL((k_octree_lds<2048, 256>), grid, block, 0, st, cand, cellCount, P, selStage, nselLevel, scratch, maxN);
  return hipGetLastError();
}

hipError_t launch_sel_compact(hipStream_t st, int nFrames, const SelKp* selStage, const int* nselLevel, const OctLaunch& P,
                              SelKp* sel, int* nsel, int selCap, int* err, int* maxN, int* hostMaxN) {
  hipLaunchKernelGGL(k_sel_compact, dim3(nFrames), dim3(256), 0, st, selStage, nselLevel, P, sel, nsel, selCap, err, maxN, hostMaxN);
  return hipGetLastError();
}

#ifdef ORBX_OCT_STAMPS
extern "C" int orbx_diag_oct_stamps(unsigned long long* out, int nBlocks) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_octStamps), sizeof(unsigned long long) * OCT_NSTAMP * nBlocks);
}
#endif

// ---- test hook: the std::sort replay alone (partition phase on one lane + parallel stable rank sort) ---------------
__global__ __launch_bounds__(OCT_T) void k_debug_sort(int* triples, int n, u64* a, u64* b) {
  const int tid = threadIdx.x;
  for (int j = tid; j < n; j += OCT_T)
    a[j] = ((u64)(uint32_t)triples[3 * j] << 40) | ((u64)((uint32_t)triples[3 * j + 1] & 0xfffff) << 20) |
           (u64)((uint32_t)triples[3 * j + 2] & 0xfffff);
  __syncthreads();
  __shared__ uint32_t parScr[OCT_PAR_SCR];
  __shared__ u64 parKeys[OCT_PAR_MAX];
  __shared__ int parWs[4];
  if (n <= OCT_PAR_MAX) {  // the workgroup-parallel replay works on LDS keys, as in the selection kernels
    for (int j = tid; j < n; j += OCT_T) parKeys[j] = a[j];
    __syncthreads();
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
