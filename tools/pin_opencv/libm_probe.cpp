// libm_probe.cpp — which function do the reference's unqualified cos(angle) / sin(angle) / pow(factor, float) calls on floats
// resolve to (Features/ORBextractor.cpp:174, :536)?  The translation unit has NO `using namespace std` (cpp:69-71 import list / pair /
// vector), so the answer depends on which headers are in the include chain: with <cmath> alone only C's ::cos(double) is in the
// global namespace (the float is promoted: ORBX_LIBM_DOUBLE); once libstdc++'s <math.h> wrapper is included anywhere it does
// `using std::cos;` and overload resolution picks std::cos(float) = cosf (ORBX_LIBM_FLOAT).  This probe includes what the reference's
// ORBextractor.cpp includes and lets the COMPILER say which one it is: the type of cos(1.0f).
//   with a real OpenCV (the case that decides the library's default):
//     g++ -std=c++17 $(pkg-config --cflags opencv4) tools/pin_opencv/libm_probe.cpp -o /tmp/libm_probe && /tmp/libm_probe
//   without one (-DORBX_PROBE_NO_OPENCV: the standard headers of the reference alone -- on libstdc++ 11 they do not even declare
//   cos, so OpenCV's headers decide --; -DORBX_PROBE_CMATH adds <cmath>, -DORBX_PROBE_MATH_H adds <math.h>: tests/test_host.py
//   compiles both to show the mechanism on this image's libstdc++: DOUBLE and FLOAT)
#include <algorithm>
#include <iostream>
#include <list>
#include <utility>
#include <vector>
#ifndef ORBX_PROBE_NO_OPENCV
#include <opencv2/opencv.hpp>  // Features/ORBextractor.hpp:24
#include "opencv2/core/core.hpp"
#include "opencv2/features2d/features2d.hpp"
#include "opencv2/highgui/highgui.hpp"
#include "opencv2/imgproc/imgproc.hpp"
#endif
#ifdef ORBX_PROBE_CMATH
#include <cmath>
#endif
#ifdef ORBX_PROBE_MATH_H
#include <math.h>
#endif
#include <type_traits>

using std::list;
using std::pair;
using std::vector;

int main() {
  float angle = 1.0f, factor = 0.8f;
  const bool cosFloat = std::is_same<decltype(cos(angle)), float>::value;
  const bool sinFloat = std::is_same<decltype(sin(angle)), float>::value;
  const bool powFloat = std::is_same<decltype(pow(factor, static_cast<float>(8))), float>::value;
  std::cout << "cos(float) -> " << (cosFloat ? "float: cosf" : "double: cos((double)x)") << "\n"
            << "sin(float) -> " << (sinFloat ? "float: sinf" : "double: sin((double)x)") << "\n"
            << "pow(float, float) -> " << (powFloat ? "float: powf" : "double: pow((double)x, (double)y)") << "\n"
            << "orbx_set_libm_variant: " << (cosFloat && sinFloat && powFloat ? "ORBX_LIBM_FLOAT" : (!cosFloat && !sinFloat && !powFloat ? "ORBX_LIBM_DOUBLE" : "MIXED"))
            << std::endl;
  return 0;
}
