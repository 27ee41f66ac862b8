// MOCK (compile check only, see ../core/core.hpp): declarations of the imgproc calls tools/pin_opencv/pin_opencv.cpp makes
#pragma once
#include "../core/core.hpp"
namespace cv {
void resize(InputArray src, OutputArray dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR);
void GaussianBlur(InputArray src, OutputArray dst, Size ksize, double sigmaX, double sigmaY = 0, int borderType = BORDER_REFLECT_101);
void cvtColor(InputArray src, OutputArray dst, int code, int dstCn = 0);
}  // namespace cv
