// host_quadtree.cpp — host prototype of the keypoint quadtree selection (TEST INFRASTRUCTURE, built as
// tests/cpp/libhostquadtree.so by tests/cpp/Makefile; the product's selection stage is the device code in
// orb_slam_tracking_amd/csrc/orbx_octree_kernel.hip, and liborbx.so contains no CPU path)
// (reference: ORBextractor::DistributeOctTree, Features/ORBextractor.cpp:698-1011;
//  ExtractorNode::DivideNode cpp:617-676; compareNodes cpp:684-696).
//
// Formulation (array based, no linked lists): the way a key travels down the quadtree depends only
// on its own coordinates (the split lines are fixed by the root rectangle), so every key gets a
// "path code" = root index followed by one quadrant digit (2 bits) per depth.  With the keys sorted
// by path code every tree node is a contiguous range, and the reference's std::list bookkeeping
// reduces to arithmetic on common-prefix lengths:
//   * size of the node list after pass k   = number of distinct depth-k prefixes           (S_k)
//   * expandable nodes created in pass k   = depth-k prefix groups with >= 2 keys          (E_k)
//   * list order: children are push_front'ed while the list is walked front to back, so the
//     depth-k block is ordered by the path digits with alternating direction, followed by the
//     single-key leaves of depth k-1, k-2, ... in their own (older) block orders.
// The last, partial pass ("split the biggest nodes first until N is reached", cpp:897-965) is
// replayed literally, including the UNSTABLE std::sort on (count, UL.x).
// The device version of this stage uses the same formulation; this file lets CPU-only tests check the formulation
// against the oracle's literal std::list restatement.
#include "host_quadtree.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace orbx {

namespace {

const int kDepth = 16;  // quadrant digits kept per key (image side <= 4096 needs 12-13)

struct Rect {
  int ulx, uly, brx, bry;
};

// one DivideNode step for a key: returns the quadrant (0=n1 UL, 1=n2 UR, 2=n3 BL, 3=n4 BR)
inline int descend(Rect& r, float x, float y) {
  const int halfX = (int)std::ceil((float)(r.brx - r.ulx) / 2);
  const int halfY = (int)std::ceil((float)(r.bry - r.uly) / 2);
  const int midX = r.ulx + halfX, midY = r.uly + halfY;
  const int qx = !(x < (float)midX), qy = !(y < (float)midY);
  if (qx) r.ulx = midX; else r.brx = midX;
  if (qy) r.uly = midY; else r.bry = midY;
  return qy * 2 + qx;
}

struct Key {
  uint64_t code;
  int idx;  // position in the candidate list (= order of vToDistributeKeys)
};

struct NodeRec {
  int lo, hi;    // range in the code-sorted key array
  int depth;     // number of quadrant digits that define this node
  bool alive;
};

struct Sized {
  int count, ulx, node;
};

}  // namespace

int octree_select(const OctCand* c, int n, int minX, int maxX, int minY, int maxY, int N, std::vector<int>& out) {
  out.clear();
  if (n <= 0) return 0;
  const int nIni = (int)std::round((float)(maxX - minX) / (float)(maxY - minY));  // cpp:706
  if (nIni < 1 || nIni > 255) return ORBX_E_TOOSMALL_INTERNAL;
  const float hX = (float)(maxX - minX) / (float)nIni;  // cpp:709

  auto rootRect = [&](int root) {
    Rect r;
    r.ulx = (int)(hX * (float)root);
    r.brx = (int)(hX * (float)(root + 1));
    r.uly = 0;
    r.bry = maxY - minY;
    return r;
  };

  // 1. path codes
  std::vector<Key> keys(n);
  for (int i = 0; i < n; i++) {
    int root = (int)(c[i].x / hX);  // cpp:747
    if (root < 0) root = 0;
    if (root >= nIni) root = nIni - 1;
    Rect r = rootRect(root);
    uint64_t code = (uint64_t)root;
    for (int d = 0; d < kDepth; d++) code = (code << 2) | (uint64_t)descend(r, c[i].x, c[i].y);
    keys[i].code = code;
    keys[i].idx = i;
  }
  std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) { return a.code != b.code ? a.code < b.code : a.idx < b.idx; });

  // 2. divergence depth between sorted neighbours (0 = different roots, kDepth+1 = never)
  std::vector<int> div(n + 1, -1);  // div[0] = div[n] = -1: "always separated"
  for (int i = 1; i < n; i++) {
    uint64_t x = keys[i - 1].code ^ keys[i].code;
    if (!x) { div[i] = kDepth + 1; continue; }
    int hb = 63 - __builtin_clzll(x);
    div[i] = hb >= 2 * kDepth ? 0 : kDepth - hb / 2;
  }
  // alone[i]: first depth at which key i is the only key of its node
  std::vector<int> alone(n);
  std::vector<int> S(kDepth + 2, 1), singles(kDepth + 2, 0);
  {
    std::vector<int> cntDiv(kDepth + 2, 0), cntAlone(kDepth + 2, 0);
    for (int i = 1; i < n; i++) cntDiv[div[i]]++;
    for (int i = 0; i < n; i++) {
      alone[i] = std::max(std::max(div[i], div[i + 1]), 0);
      cntAlone[alone[i]]++;
    }
    int accD = 0, accA = 0;
    for (int d = 0; d <= kDepth; d++) {
      accD += cntDiv[d];
      accA += cntAlone[d];
      S[d] = 1 + accD;    // distinct depth-d prefixes
      singles[d] = accA;  // keys that are alone at depth <= d
    }
    S[kDepth + 1] = S[kDepth];
    singles[kDepth + 1] = singles[kDepth];
  }

  // 3. replay the pass loop on sizes only (cpp:781-895)
  int k = 0;
  bool phase2 = false;
  for (;;) {
    const int prevSize = S[k];
    if (k < kDepth) k++;  // beyond kDepth nothing can split any further: size stays
    const int size = S[k];
    const int nToExpand = size - singles[k];
    if (size >= N || size == prevSize) break;
    if (size + 3 * nToExpand > N) { phase2 = true; break; }
  }

  // 4. node list after k full passes, in std::list order:
  //    [depth-k block] ++ [single-key leaves of depth k-1] ++ ... ++ [depth 0]
  // order inside the depth-j block: lexicographic on (root, q1..qj) with alternating direction.
  auto orderKey = [&](uint64_t code, int j) {
    // keep the first j digits, flip digit m (1-based) when (j-m) is even (descending), root when j is odd
    uint64_t prefix = code >> (2 * (kDepth - j));
    uint64_t flip = 0;
    for (int m = j; m >= 1; m -= 2) flip |= (uint64_t)3 << (2 * (j - m));
    uint64_t key = prefix ^ flip;
    if (j & 1) {
      uint64_t root = key >> (2 * j);
      key = (key & (((uint64_t)1 << (2 * j)) - 1)) | ((uint64_t)(255 - root) << (2 * j));
    }
    return key;
  };
  std::vector<NodeRec> lst;
  {
    struct Tmp { uint64_t okey; int blockDepth; NodeRec nd; };
    std::vector<Tmp> tmp;
    int i = 0;
    while (i < n) {
      if (alone[i] < k) {  // leaf created (as a single-key node) at depth alone[i]
        Tmp t; t.blockDepth = alone[i]; t.okey = orderKey(keys[i].code, alone[i]);
        t.nd = NodeRec{i, i + 1, alone[i], true};
        tmp.push_back(t);
        i++;
      } else {  // depth-k node: all following keys with the same k-prefix
        int j = i + 1;
        while (j < n && div[j] > k) j++;
        Tmp t; t.blockDepth = k; t.okey = orderKey(keys[i].code, k);
        t.nd = NodeRec{i, j, k, true};
        tmp.push_back(t);
        i = j;
      }
    }
    std::sort(tmp.begin(), tmp.end(), [](const Tmp& a, const Tmp& b) {
      if (a.blockDepth != b.blockDepth) return a.blockDepth > b.blockDepth;
      return a.okey < b.okey;
    });
    lst.reserve(tmp.size());
    for (const Tmp& t : tmp) lst.push_back(t.nd);
  }

  // 5. partial pass(es): cpp:897-965
  std::vector<NodeRec> front;  // nodes push_front'ed during phase 2, in push order
  if (phase2) {
    auto nodeUlx = [&](const NodeRec& nd) {
      const uint64_t code = keys[nd.lo].code;
      Rect r = rootRect((int)(code >> (2 * kDepth)));
      for (int d = 1; d <= nd.depth; d++) {
        const int q = (int)((code >> (2 * (kDepth - d))) & 3);
        const int halfX = (int)std::ceil((float)(r.brx - r.ulx) / 2);
        const int halfY = (int)std::ceil((float)(r.bry - r.uly) / 2);
        if (q & 1) r.ulx += halfX; else r.brx = r.ulx + halfX;
        if (q & 2) r.uly += halfY; else r.bry = r.uly + halfY;
      }
      return r.ulx;
    };
    // handles: >= 0 -> index into lst, < 0 -> ~index into front
    auto nodeAt = [&](int h) -> NodeRec& { return h >= 0 ? lst[h] : front[~h]; };
    std::vector<int> pending;  // creation order = reverse of the depth-k block order
    for (int i = (int)lst.size() - 1; i >= 0; i--)
      if (lst[i].depth == k && lst[i].hi - lst[i].lo > 1) pending.push_back(i);
    int size = (int)lst.size();
    bool finish = false;
    while (!finish) {
      const int prevSize = size;
      std::vector<Sized> prev(pending.size());
      for (size_t i = 0; i < pending.size(); i++) {
        const NodeRec& nd = nodeAt(pending[i]);
        prev[i] = Sized{nd.hi - nd.lo, nodeUlx(nd), pending[i]};
      }
      pending.clear();
      std::sort(prev.begin(), prev.end(), [](const Sized& a, const Sized& b) {  // compareNodes, cpp:684-696
        if (a.count < b.count) return true;
        if (a.count > b.count) return false;
        return a.ulx < b.ulx;
      });
      for (int j = (int)prev.size() - 1; j >= 0; j--) {
        const NodeRec parent = nodeAt(prev[j].node);
        int nChildren = 0;
        if (parent.depth >= kDepth) {  // cannot be split any further (coincident keys)
          front.push_back(NodeRec{parent.lo, parent.hi, parent.depth, true});
          pending.push_back(~((int)front.size() - 1));
          nChildren = 1;
        } else {
          int lo = parent.lo;
          while (lo < parent.hi) {
            int hi = lo + 1;
            while (hi < parent.hi && div[hi] > parent.depth + 1) hi++;
            front.push_back(NodeRec{lo, hi, parent.depth + 1, true});
            if (hi - lo > 1) pending.push_back(~((int)front.size() - 1));
            nChildren++;
            lo = hi;
          }
        }
        nodeAt(prev[j].node).alive = false;
        size += nChildren - 1;
        if (size >= N) break;
      }
      if (size >= N || size == prevSize) finish = true;
    }
  }

  // 6. one key per node: highest response, the earliest candidate wins ties (cpp:984-1007)
  auto emit = [&](const NodeRec& nd) {
    int best = keys[nd.lo].idx;
    for (int i = nd.lo + 1; i < nd.hi; i++) {
      const int id = keys[i].idx;
      if (c[id].response > c[best].response || (c[id].response == c[best].response && id < best)) best = id;
    }
    out.push_back(best);
  };
  for (int i = (int)front.size() - 1; i >= 0; i--)
    if (front[i].alive) emit(front[i]);
  for (const NodeRec& nd : lst)
    if (nd.alive) emit(nd);
  return (int)out.size();
}

}  // namespace orbx

// C entry point for the CPU tests (ctypes): DistributeOctTree (cpp:698-1011) on caller-supplied candidates.
extern "C" int hostquadtree_distribute(const float* xyr, int n, int min_x, int max_x, int min_y, int max_y, int n_features,
                                       float* out_xyr, int cap) {
  if (n < 0 || (n > 0 && !xyr) || max_x <= min_x || max_y <= min_y) return ORBX_E_BADARG;
  std::vector<orbx::OctCand> c(n);
  for (int i = 0; i < n; i++) c[i] = orbx::OctCand{xyr[3 * i], xyr[3 * i + 1], xyr[3 * i + 2]};
  std::vector<int> chosen;
  int r = orbx::octree_select(c.data(), n, min_x, max_x, min_y, max_y, n_features, chosen);
  if (r < 0) return r;
  for (int i = 0; i < (int)chosen.size() && i < cap; i++) {
    out_xyr[3 * i] = c[chosen[i]].x;
    out_xyr[3 * i + 1] = c[chosen[i]].y;
    out_xyr[3 * i + 2] = c[chosen[i]].response;
  }
  return (int)chosen.size();
}
