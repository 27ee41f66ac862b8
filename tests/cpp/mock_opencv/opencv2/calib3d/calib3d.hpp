// MOCK (compile check only, see ../core/core.hpp): declaration of cv::undistortPoints as SlamTypes/Frame.cpp:119,150 calls it
#pragma once
#include "../core/core.hpp"
namespace cv {
void undistortPoints(InputArray src, OutputArray dst, InputArray cameraMatrix, InputArray distCoeffs, InputArray R, InputArray P);
}  // namespace cv
