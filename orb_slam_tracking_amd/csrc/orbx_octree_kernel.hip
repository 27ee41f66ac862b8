// orbx_octree_kernel.hip — device version of the keypoint quadtree selection
// (reference: ORBextractor::DistributeOctTree, Features/ORBextractor.cpp:698-1011; DivideNode cpp:617-676;
//  compareNodes cpp:684-696; truncation to the level quota cpp:1159-1161; coordinate fix-up cpp:1165-1179).
//
// One workgroup per (frame, pyramid level).  Same array formulation as the host implementation in orbx_octree.cpp:
// every candidate gets a path code (root index + one quadrant digit per depth); with the codes sorted, every tree
// node is a contiguous range and the reference's std::list bookkeeping becomes arithmetic on common-prefix lengths.
// The partial last pass ("split the biggest nodes first until N nodes exist") needs the exact permutation produced by
// libstdc++'s UNSTABLE std::sort on (count, UL.x); it is replayed by one lane with a literal re-implementation of
// libstdc++'s introsort (median-of-3 quicksort, heapsort fallback, final insertion sort, threshold 16).
// The candidates arrive unordered from k_fast; the reference's candidate order (cell row, cell col, y, x) only
// matters as the tie-breaker "first of the highest responses", so it is carried as an order key, never materialised.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/orbx.h"
#include "orbx_device.h"

namespace orbx {

#define OCT_T 256
#define OCT_DEPTH 16

typedef unsigned long long u64;

struct OctScratch {
  u64* keys;        // [nPad]   (code << 24) | candidate index, sorted ascending
  u64* nodes;       // [mPad]   (17 - blockDepth) << 59 | orderKey << 19 | lo, sorted ascending = std::list order
  uint8_t* div;     // [n + 1]  divergence depth between sorted neighbours (div[0] = div[n] = 255 -> "separated")
  uint8_t* alone;   // [n]
  uint32_t* hiOf;   // [n]      end of the node that starts at sorted position lo
  int* nodeLo;      // [mCap + fCap]  node records: list nodes first, then the nodes pushed during the partial pass
  int* nodeHi;
  uint8_t* nodeDepth;
  uint8_t* nodeAlive;
  int* sized;       // [3 * qCap]  (count, ulx, node) triples for the emulated std::sort
  int* pending;     // [2 * qCap]
  int* pos;         // [mCap + fCap] output position of every node
};

__device__ __forceinline__ int divDepth(u64 a, u64 b) {  // first depth at which two path codes differ
  const u64 x = a ^ b;
  if (!x) return OCT_DEPTH + 1;
  const int hb = 63 - __builtin_clzll(x);
  return hb >= 2 * OCT_DEPTH ? 0 : OCT_DEPTH - hb / 2;
}

__device__ __forceinline__ void bitonicSort(u64* a, int nPow2, int tid) {
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

// ---- literal replay of libstdc++'s std::sort (bits/stl_algo.h) on (count, ulx, node) triples -------------------
struct SizedRef {
  int* p;  // triples
  __device__ __forceinline__ bool less(int i, int j) const {  // compareNodes
    const int ci = p[3 * i], cj = p[3 * j];
    if (ci < cj) return true;
    if (ci > cj) return false;
    return p[3 * i + 1] < p[3 * j + 1];
  }
  __device__ __forceinline__ void swap(int i, int j) const {
    for (int k = 0; k < 3; k++) { const int t = p[3 * i + k]; p[3 * i + k] = p[3 * j + k]; p[3 * j + k] = t; }
  }
  __device__ __forceinline__ void move(int dst, int src) const {
    p[3 * dst] = p[3 * src]; p[3 * dst + 1] = p[3 * src + 1]; p[3 * dst + 2] = p[3 * src + 2];
  }
};
struct SizedVal {
  int c, u, n;
};
__device__ __forceinline__ bool valLess(const SizedVal& a, const int* p, int j) {  // comp(val, *j)
  if (a.c < p[3 * j]) return true;
  if (a.c > p[3 * j]) return false;
  return a.u < p[3 * j + 1];
}
__device__ __forceinline__ bool lessVal(const int* p, int i, const SizedVal& b) {  // comp(*i, val)
  if (p[3 * i] < b.c) return true;
  if (p[3 * i] > b.c) return false;
  return p[3 * i + 1] < b.u;
}

__device__ void stdAdjustHeap(int* p, int first, int holeIndex, int len, SizedVal value) {
  SizedRef r{p};
  const int topIndex = holeIndex;
  int secondChild = holeIndex;
  while (secondChild < (len - 1) / 2) {
    secondChild = 2 * (secondChild + 1);
    if (r.less(first + secondChild, first + (secondChild - 1))) secondChild--;
    r.move(first + holeIndex, first + secondChild);
    holeIndex = secondChild;
  }
  if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
    secondChild = 2 * (secondChild + 1);
    r.move(first + holeIndex, first + (secondChild - 1));
    holeIndex = secondChild - 1;
  }
  // __push_heap
  int parent = (holeIndex - 1) / 2;
  while (holeIndex > topIndex && lessVal(p, first + parent, value)) {
    r.move(first + holeIndex, first + parent);
    holeIndex = parent;
    parent = (holeIndex - 1) / 2;
  }
  p[3 * (first + holeIndex)] = value.c; p[3 * (first + holeIndex) + 1] = value.u; p[3 * (first + holeIndex) + 2] = value.n;
}

__device__ void stdHeapSortRange(int* p, int first, int last) {  // __partial_sort(first, last, last)
  const int len = last - first;
  if (len >= 2) {  // __make_heap
    int parent = (len - 2) / 2;
    for (;;) {
      SizedVal v{p[3 * (first + parent)], p[3 * (first + parent) + 1], p[3 * (first + parent) + 2]};
      stdAdjustHeap(p, first, parent, len, v);
      if (parent == 0) break;
      parent--;
    }
  }
  // __heap_select's loop over [middle, last) is empty; __sort_heap:
  int l = last;
  while (l - first > 1) {
    --l;
    SizedVal v{p[3 * l], p[3 * l + 1], p[3 * l + 2]};  // __pop_heap(first, l, l)
    SizedRef{p}.move(l, first);
    stdAdjustHeap(p, first, 0, l - first, v);
  }
}

__device__ void stdSortSized(int* p, int n) {
  if (n <= 1) return;
  SizedRef r{p};
  // __introsort_loop with an explicit stack of (first, last, depth) for the recursive right halves
  int stackF[64], stackL[64], stackD[64];
  int sp = 0;
  int first = 0, last = n;
  int depth = 2 * (31 - __builtin_clz((unsigned)n));
  for (;;) {
    while (last - first > 16) {
      if (depth == 0) { stdHeapSortRange(p, first, last); break; }
      --depth;
      // __unguarded_partition_pivot
      const int mid = first + (last - first) / 2;
      {  // __move_median_to_first(first, first + 1, mid, last - 1)
        const int a = first + 1, b = mid, c = last - 1;
        if (r.less(a, b)) {
          if (r.less(b, c)) r.swap(first, b);
          else if (r.less(a, c)) r.swap(first, c);
          else r.swap(first, a);
        } else if (r.less(a, c)) r.swap(first, a);
        else if (r.less(b, c)) r.swap(first, c);
        else r.swap(first, b);
      }
      int lo = first + 1, hi = last;
      for (;;) {  // __unguarded_partition(first + 1, last, first)
        while (r.less(lo, first)) ++lo;
        --hi;
        while (r.less(first, hi)) --hi;
        if (!(lo < hi)) break;
        r.swap(lo, hi);
        ++lo;
      }
      const int cut = lo;
      // recurse on [cut, last) first (as the library does), then continue with [first, cut)
      stackF[sp] = first; stackL[sp] = cut; stackD[sp] = depth; sp++;
      first = cut;
    }
    if (sp == 0) break;
    --sp;
    first = stackF[sp]; last = stackL[sp]; depth = stackD[sp];
  }
  // __final_insertion_sort
  auto insertionSort = [&](int f, int l) {
    if (f == l) return;
    for (int i = f + 1; i != l; ++i) {
      if (r.less(i, f)) {
        SizedVal v{p[3 * i], p[3 * i + 1], p[3 * i + 2]};
        for (int k = i; k > f; --k) r.move(k, k - 1);  // move_backward
        p[3 * f] = v.c; p[3 * f + 1] = v.u; p[3 * f + 2] = v.n;
      } else {  // __unguarded_linear_insert
        SizedVal v{p[3 * i], p[3 * i + 1], p[3 * i + 2]};
        int lastPos = i, next = i - 1;
        while (valLess(v, p, next)) { r.move(lastPos, next); lastPos = next; --next; }
        p[3 * lastPos] = v.c; p[3 * lastPos + 1] = v.u; p[3 * lastPos + 2] = v.n;
      }
    }
  };
  if (n > 16) {
    insertionSort(0, 16);
    for (int i = 16; i != n; ++i) {  // __unguarded_insertion_sort
      SizedVal v{p[3 * i], p[3 * i + 1], p[3 * i + 2]};
      int lastPos = i, next = i - 1;
      while (valLess(v, p, next)) { r.move(lastPos, next); lastPos = next; --next; }
      p[3 * lastPos] = v.c; p[3 * lastPos + 1] = v.u; p[3 * lastPos + 2] = v.n;
    }
  } else {
    insertionSort(0, n);
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

// The whole selection for one (frame, level).  cand: n unordered packed candidates.  Writes min(#nodes, quota)
// SelKp records (list order) to `out` and returns that count in *nOut (by thread 0).
__device__ void octreeSelect(const OctScratch S, const uint32_t* __restrict__ cand, int n, const OctLevel L, int level,
                             SelKp* __restrict__ out, int* __restrict__ nOut, int mCap, int fCap, int qCap) {
  __shared__ int cntDiv[OCT_DEPTH + 2], cntAlone[OCT_DEPTH + 2];
  __shared__ int sK, sPhase2, sM, sFront, sTotal;
  __shared__ int partial[OCT_T];
  const int tid = threadIdx.x;
  const int N = L.quota;
  if (n <= 0 || N <= 0) {  // nothing to select (an empty quota truncates everything, cpp:1159-1161)
    if (tid == 0) *nOut = 0;
    return;
  }
  int nPad = 1;
  while (nPad < n) nPad <<= 1;

  // ---- 1. path codes --------------------------------------------------------------------------------------
  for (int i = tid; i < nPad; i += OCT_T) {
    u64 key = ~0ull;
    if (i < n) {
      const uint32_t e = cand[i];
      const float x = (float)(e & 0xfff), y = (float)((e >> 12) & 0xfff);
      int root = (int)(x / L.hX);  // cpp:747
      root = min(max(root, 0), L.nIni - 1);
      int ulx, uly, brx, bry;
      rootRect(L, root, ulx, uly, brx, bry);
      u64 code = (u64)root;
      for (int d = 0; d < OCT_DEPTH; d++) {  // DivideNode, cpp:617-676
        const int midX = ulx + ((brx - ulx + 1) >> 1), midY = uly + ((bry - uly + 1) >> 1);
        const int qx = !(x < (float)midX), qy = !(y < (float)midY);
        if (qx) ulx = midX; else brx = midX;
        if (qy) uly = midY; else bry = midY;
        code = (code << 2) | (u64)(qy * 2 + qx);
      }
      key = (code << 24) | (u64)i;
    }
    S.keys[i] = key;
  }
  if (tid < OCT_DEPTH + 2) { cntDiv[tid] = 0; cntAlone[tid] = 0; }
  __syncthreads();
  bitonicSort(S.keys, nPad, tid);

  // ---- 2. divergence depths, S_d, singles_d ---------------------------------------------------------------
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
  // ---- 3. replay the pass loop on sizes only (cpp:781-895) ----------------------------------------------
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
  // ---- 4. node list in std::list order ---------------------------------------------------------------------
  // node starts: a leaf key (alone < k) or the first key of a depth-k group
  {
    const int chunk = (n + OCT_T - 1) / OCT_T;
    const int b = tid * chunk, e = min(b + chunk, n);
    int c = 0;
    for (int i = b; i < e; i++) c += (S.alone[i] < k) || (S.div[i] == 255 || (int)S.div[i] <= k);
    partial[tid] = c;
    __syncthreads();
    if (tid == 0) {
      int acc = 0;
      for (int i = 0; i < OCT_T; i++) { const int v = partial[i]; partial[i] = acc; acc += v; }
    }
    __syncthreads();
    int m = partial[tid];
    for (int i = b; i < e; i++) {
      const bool leaf = S.alone[i] < k;
      if (leaf || (S.div[i] == 255 || (int)S.div[i] <= k)) {
        const int j = leaf ? (int)S.alone[i] : k;  // block depth
        const u64 code = S.keys[i] >> 24;
        // order key: first j digits, digit m flipped when (j - m) is even, root flipped when j is odd
        u64 prefix = code >> (2 * (OCT_DEPTH - j));
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
    // hiOf[start] = next start
    __syncthreads();
  }
  int mPad = 1;
  while (mPad < M) mPad <<= 1;
  // ends of the nodes, found from the position-ordered list before it is re-sorted
  for (int m = tid; m < M; m += OCT_T) {
    const int lo = (int)(S.nodes[m] & 0x7ffff);
    const int hi = (m + 1 < M) ? (int)(S.nodes[m + 1] & 0x7ffff) : n;
    S.hiOf[lo] = (uint32_t)hi;
  }
  for (int m = M + tid; m < mPad; m += OCT_T) S.nodes[m] = ~0ull;
  __syncthreads();
  bitonicSort(S.nodes, mPad, tid);
  for (int m = tid; m < M; m += OCT_T) {
    const u64 v = S.nodes[m];
    const int lo = (int)(v & 0x7ffff);
    S.nodeLo[m] = lo;
    S.nodeHi[m] = (int)S.hiOf[lo];
    S.nodeDepth[m] = (uint8_t)(OCT_DEPTH + 1 - (int)(v >> 59));
    S.nodeAlive[m] = 1;
  }
  __syncthreads();

  // ---- 5. partial pass(es), cpp:897-965: replayed by one lane ----------------------------------------------
  if (sPhase2 && tid == 0) {
    int size = M;
    int nPend = 0;
    for (int i = M - 1; i >= 0; i--)  // creation order = reverse of the depth-k block order
      if (S.nodeDepth[i] == k && S.nodeHi[i] - S.nodeLo[i] > 1) S.pending[nPend++] = i;  // < N <= qCap entries
    int nFront = 0;
    bool finish = false, overflow = false;
    while (!finish) {
      const int prevSize = size;
      const int np = min(nPend, qCap);
      for (int i = 0; i < np; i++) {
        const int nd = S.pending[i];
        const int lo = S.nodeLo[nd];
        // UL.x of the node: walk its digits from the root rectangle
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
        S.sized[3 * i] = S.nodeHi[nd] - lo;
        S.sized[3 * i + 1] = ulx;
        S.sized[3 * i + 2] = nd;
      }
      nPend = 0;
      stdSortSized(S.sized, np);  // cpp:912
      for (int j = np - 1; j >= 0; j--) {
        const int parent = S.sized[3 * j + 2];
        const int plo = S.nodeLo[parent], phi = S.nodeHi[parent], pd = S.nodeDepth[parent];
        int nChildren = 0;
        if (pd >= OCT_DEPTH) {  // coincident keys: cannot be split further
          if (nFront < fCap && nPend < 2 * qCap) {
            const int id = mCap + nFront++;
            S.nodeLo[id] = plo; S.nodeHi[id] = phi; S.nodeDepth[id] = (uint8_t)pd; S.nodeAlive[id] = 1;
            S.pending[nPend++] = id;
          } else {
            overflow = true;
          }
          nChildren = 1;
        } else {
          int lo = plo;
          while (lo < phi) {
            int hi = lo + 1;
            while (hi < phi && (int)S.div[hi] > pd + 1) hi++;
            if (nFront < fCap && (hi - lo == 1 || nPend < 2 * qCap)) {
              const int id = mCap + nFront++;
              S.nodeLo[id] = lo; S.nodeHi[id] = hi; S.nodeDepth[id] = (uint8_t)(pd + 1); S.nodeAlive[id] = 1;
              if (hi - lo > 1) S.pending[nPend++] = id;
            } else {
              overflow = true;
            }
            nChildren++;
            lo = hi;
          }
        }
        S.nodeAlive[parent] = 0;
        size += nChildren - 1;
        if (size >= N) break;
      }
      if (size >= N || size == prevSize || overflow) finish = true;
    }
    sFront = overflow ? -1 : nFront;
  }
  __syncthreads();
  if (sFront < 0) {  // scratch too small for this unit: report, the caller re-runs it with larger scratch
    if (tid == 0) *nOut = -2;
    return;
  }
  // ---- 6. output positions: reverse(front alive) ++ list alive; keep the first `quota` ----------------------
  const int nFront = sFront;
  const int total = nFront + M;  // virtual sequence: front nodes in reverse push order, then the list
  {
    const int chunk = (total + OCT_T - 1) / OCT_T;
    const int b = tid * chunk, e = min(b + chunk, total);
    auto nodeAt = [&](int v) { return v < nFront ? mCap + (nFront - 1 - v) : v - nFront; };
    int c = 0;
    for (int v = b; v < e; v++) c += S.nodeAlive[nodeAt(v)];
    partial[tid] = c;
    __syncthreads();
    if (tid == 0) {
      int acc = 0;
      for (int i = 0; i < OCT_T; i++) { const int v = partial[i]; partial[i] = acc; acc += v; }
      sTotal = acc;
    }
    __syncthreads();
    int p = partial[tid];
    for (int v = b; v < e; v++) {
      const int nd = nodeAt(v);
      if (!S.nodeAlive[nd]) continue;
      if (p < N) {
        // first key with the highest response (cpp:984-1007); "first" = reference candidate order
        const int lo = S.nodeLo[nd], hi = S.nodeHi[nd];
        uint32_t bestE = cand[(int)(S.keys[lo] & 0xffffff)];
        u64 bestRank = candRank(bestE, L);
        for (int i = lo + 1; i < hi; i++) {
          const uint32_t e2 = cand[(int)(S.keys[i] & 0xffffff)];
          const u64 r2 = candRank(e2, L);
          const uint32_t s1 = bestE >> 24, s2 = e2 >> 24;
          if (s2 > s1 || (s2 == s1 && r2 < bestRank)) { bestE = e2; bestRank = r2; }
        }
        SelKp kp;
        kp.x = (uint16_t)((bestE & 0xfff) + ORBX_MIN_BORDER);        // cpp:1171-1172
        kp.y = (uint16_t)(((bestE >> 12) & 0xfff) + ORBX_MIN_BORDER);
        kp.level = (uint8_t)level;
        kp.response = (uint8_t)(bestE >> 24);
        kp.pad = 0;
        out[p] = kp;
      }
      p++;
    }
  }
  __syncthreads();
  if (tid == 0) *nOut = min(sTotal, N);
}

// ---- kernels -------------------------------------------------------------------------------------------------

// LDS-resident variant: n <= NMAX candidates, quota <= QMAX
template <int NMAX, int QMAX>
__global__ __launch_bounds__(OCT_T) void k_octree_lds(const uint32_t* __restrict__ cand, const int* __restrict__ candCount,
                                                     const OctLaunch P, SelKp* __restrict__ selStage,
                                                     int* __restrict__ nselLevel) {
  constexpr int MCAP = 4 * QMAX, FCAP = 4 * QMAX;
  static_assert((NMAX & (NMAX - 1)) == 0 && (MCAP & (MCAP - 1)) == 0, "sort buffers must be powers of two");
  __shared__ u64 keys[NMAX];
  __shared__ u64 nodes[MCAP];
  __shared__ uint8_t div[NMAX + 4], alone[NMAX];
  __shared__ uint32_t hiOf[NMAX];
  __shared__ int nodeLo[MCAP + FCAP], nodeHi[MCAP + FCAP];
  __shared__ uint8_t nodeDepth[MCAP + FCAP], nodeAlive[MCAP + FCAP];
  __shared__ int sized[3 * QMAX], pending[2 * QMAX];
  const int level = blockIdx.x, f = blockIdx.y;
  const int n = candCount[f * P.nlevels + level];
  int* nOut = &nselLevel[f * P.nlevels + level];
  if (n > NMAX || P.lev[level].quota > QMAX) {  // handled by the global-scratch variant
    if (threadIdx.x == 0) *nOut = -2;
    return;
  }
  OctScratch S{keys, nodes, div, alone, hiOf, nodeLo, nodeHi, nodeDepth, nodeAlive, sized, pending, nullptr};
  octreeSelect(S, cand + P.candOff[level] + (int64_t)f * P.candCap[level], n, P.lev[level], level,
               selStage + (int64_t)f * P.selStride + P.selOff[level], nOut, MCAP, FCAP, QMAX);
}

// global-scratch variant for the (frame, level) units the LDS variant left (nselLevel == -2), or for all units when
// `all` is set.  Scratch of unit (f, level) starts at scrOff[level] + f * scrStride[level]; layout: octScratchBytes().
__global__ __launch_bounds__(OCT_T) void k_octree_global(const uint32_t* __restrict__ cand, const int* __restrict__ candCount,
                                                        const OctLaunch P, SelKp* __restrict__ selStage,
                                                        int* __restrict__ nselLevel, uint8_t* __restrict__ scratch, int all) {
  const int level = blockIdx.x, f = blockIdx.y;
  int* nOut = &nselLevel[f * P.nlevels + level];
  if (!all && *nOut != -2) return;
  const int nMax = P.scrNMax[level], qMax = max(P.lev[level].quota, 1);
  const int n = candCount[f * P.nlevels + level];
  if (n > nMax) {  // more candidates than the selection stage can index (2^19 - 1)
    if (threadIdx.x == 0) *nOut = -1;
    return;
  }
  size_t nPad = 1;
  while ((int)nPad < nMax) nPad <<= 1;
  const int mCap = 4 * qMax, fCap = 16 * qMax;
  size_t mPad = 1;
  while ((int)mPad < mCap) mPad <<= 1;
  uint8_t* p = scratch + P.scrOff[level] + (int64_t)f * P.scrStride[level];
  OctScratch S;
  S.keys = (u64*)p; p += nPad * 8;
  S.nodes = (u64*)p; p += mPad * 8;
  S.hiOf = (uint32_t*)p; p += nPad * 4;
  S.nodeLo = (int*)p; p += (size_t)(mCap + fCap) * 4;
  S.nodeHi = (int*)p; p += (size_t)(mCap + fCap) * 4;
  S.sized = (int*)p; p += (size_t)3 * qMax * 4;
  S.pending = (int*)p; p += (size_t)2 * qMax * 4;
  S.div = p; p += nPad + 8;
  S.alone = p; p += nPad + 8;
  S.nodeDepth = p; p += (size_t)(mCap + fCap + 8);
  S.nodeAlive = p;
  S.pos = nullptr;
  octreeSelect(S, cand + P.candOff[level] + (int64_t)f * P.candCap[level], n, P.lev[level], level,
               selStage + (int64_t)f * P.selStride + P.selOff[level], nOut, mCap, fCap, qMax);
  if (threadIdx.x == 0 && *nOut == -2) *nOut = -1;  // even the large scratch was too small: hard error
}

size_t octScratchBytes(int nMax, int qMax) {
  size_t nPad = 1;
  while ((int)nPad < nMax) nPad <<= 1;
  qMax = qMax < 1 ? 1 : qMax;
  const size_t mCap = 4 * (size_t)qMax, fCap = 16 * (size_t)qMax;
  size_t mPad = 1;
  while (mPad < mCap) mPad <<= 1;
  size_t b = nPad * 8 + mPad * 8 + nPad * 4 + (mCap + fCap) * 8 + (size_t)5 * qMax * 4 + 2 * (nPad + 8) + 2 * (mCap + fCap + 8);
  return (b + 255) / 256 * 256;
}

// compacts the per-level staging lists of every frame into level-major order and writes the per-frame totals
__global__ __launch_bounds__(256) void k_sel_compact(const SelKp* __restrict__ selStage, const int* __restrict__ nselLevel,
                                                     const OctLaunch P, SelKp* __restrict__ sel, int* __restrict__ nsel,
                                                     int selCap, int* __restrict__ err) {
  const int f = blockIdx.x;
  __shared__ int off[ORBX_MAX_LEVELS + 1];
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int l = 0; l < P.nlevels; l++) {
      off[l] = acc;
      const int c = nselLevel[f * P.nlevels + l];
      if (c < 0) { *err = 1; }
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

hipError_t launch_octree(hipStream_t st, int nFrames, const uint32_t* cand, const int* candCount, const OctLaunch& P,
                         SelKp* selStage, int* nselLevel, uint8_t* scratch, int maxQuota) {
  dim3 grid(P.nlevels, nFrames, 1), block(OCT_T, 1, 1);
  const bool lds = maxQuota <= 256;
  if (lds) hipLaunchKernelGGL((k_octree_lds<1024, 256>), grid, block, 0, st, cand, candCount, P, selStage, nselLevel);
  // units the LDS variant could not take (more than 1024 candidates, or scratch overflow) exit at once otherwise
  hipLaunchKernelGGL(k_octree_global, grid, block, 0, st, cand, candCount, P, selStage, nselLevel, scratch, lds ? 0 : 1);
  return hipGetLastError();
}

hipError_t launch_sel_compact(hipStream_t st, int nFrames, const SelKp* selStage, const int* nselLevel, const OctLaunch& P,
                              SelKp* sel, int* nsel, int selCap, int* err) {
  hipLaunchKernelGGL(k_sel_compact, dim3(nFrames), dim3(256), 0, st, selStage, nselLevel, P, sel, nsel, selCap, err);
  return hipGetLastError();
}

// ---- test hook: the std::sort replay alone ---------------------------------------------------------------------
__global__ void k_debug_sort(int* p, int n) {
  if (threadIdx.x == 0 && blockIdx.x == 0) stdSortSized(p, n);
}
hipError_t launch_debug_sort(hipStream_t st, int* triples, int n) {
  hipLaunchKernelGGL(k_debug_sort, dim3(1), dim3(64), 0, st, triples, n);
  return hipGetLastError();
}

}  // namespace orbx
