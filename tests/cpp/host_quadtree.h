// host_quadtree.h — the host prototype of the path-code quadtree selection (TEST INFRASTRUCTURE: not part of liborbx.so).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/orbx.h"  // error codes only

namespace orbx {

const int ORBX_E_TOOSMALL_INTERNAL = ORBX_E_TOOSMALL;

// a FAST candidate of one pyramid level, relative to (minBorderX, minBorderY)
// (vToDistributeKeys, Features/ORBextractor.cpp:1134-1137)
struct OctCand {
  float x, y, response;
};

// DistributeOctTree (cpp:698-1011): `out` receives indices into c[] in the order of the reference's
// final node list (NOT truncated to N; the caller truncates like cpp:1159-1161).
// Returns the number of selected keys or a negative error.
int octree_select(const OctCand* c, int n, int minX, int maxX, int minY, int maxY, int N, std::vector<int>& out);

}  // namespace orbx
