// MOCK (compile check only, see ../core/core.hpp): declaration of cv::FAST as Features/ORBextractor.cpp:1109,1119 calls it
#pragma once
#include "../core/core.hpp"
namespace cv {
void FAST(InputArray image, std::vector<KeyPoint>& keypoints, int threshold, bool nonmaxSuppression = true);
}  // namespace cv
