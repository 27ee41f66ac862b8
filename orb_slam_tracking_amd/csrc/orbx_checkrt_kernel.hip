// orbx_checkrt_kernel.hip — Initializer::CheckRT on the device (Initialization/Initializer.cpp:569-713): for every (R21, t21)
// hypothesis of ReconstructHF (:440-567) triangulate the inlier matches and count the points that lie in front of the cameras
// with a small reprojection error; parallax = the 51st smallest cosine.
//
// One wave per hypothesis, lane = inlier match.  cv::triangulatePoints is a 4x4 DLT per point: the right singular vector of
// the smallest singular value, found with the same one-sided Jacobi scheme (f64, column pairs in (i, j) order, at most 30
// sweeps) as the CPU restatement in oracle/, operation for operation, uncontracted -- see the oracle's header for what is
// [from-knowledge] about OpenCV here, and for the two quirks of the reference that are kept (booking under the compacted
// index; the camera-2 depth test on z / z).  Microseconds of work: built for completeness of the consumer side.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/orbx.h"
#include "orbx_device.h"

namespace orbx {

__device__ void smallestRightSingularVector4(const double Ain[16], double x[4]) {
  double At[4][4], V[4][4], W[4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int k = 0; k < 4; k++) { At[i][k] = Ain[k * 4 + i]; V[i][k] = i == k ? 1.0 : 0.0; }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    double sd = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) sd += At[i][k] * At[i][k];
    W[i] = sd;
  }
  const double eps = 2.2204460492503131e-16 * 10;
  for (int iter = 0; iter < 30; iter++) {
    bool changed = false;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = i + 1; j < 4; j++) {
        double a = W[i], p = 0, b = W[j];
#pragma unroll
        for (int k = 0; k < 4; k++) p += At[i][k] * At[j][k];
        if (!(fabs(p) <= eps * sqrt(a * b))) {
          p *= 2;
          const double beta = a - b, gamma = sqrt(p * p + beta * beta);
          double c, sn;
          if (beta < 0) {
            const double delta = (gamma - beta) * 0.5;
            sn = sqrt(delta / gamma);
            c = p / (gamma * sn * 2);
          } else {
            c = sqrt((gamma + beta) / (gamma * 2));
            sn = p / (gamma * c * 2);
          }
          a = b = 0;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const double t0 = c * At[i][k] + sn * At[j][k], t1 = -sn * At[i][k] + c * At[j][k];
            At[i][k] = t0; At[j][k] = t1;
            a += t0 * t0; b += t1 * t1;
          }
          W[i] = a; W[j] = b;
          changed = true;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const double t0 = c * V[i][k] + sn * V[j][k], t1 = -sn * V[i][k] + c * V[j][k];
            V[i][k] = t0; V[j][k] = t1;
          }
        }
      }
    if (!changed) break;
  }
  // smallest squared column norm, the first one among equals (selects instead of a dynamic index: the arrays stay in registers)
  double wb = W[0];
#pragma unroll
  for (int k = 0; k < 4; k++) x[k] = V[0][k];
#pragma unroll
  for (int i = 1; i < 4; i++) {
    const bool lt = W[i] < wb;
    wb = lt ? W[i] : wb;
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = lt ? V[i][k] : x[k];
  }
}

__global__ __launch_bounds__(64) void k_check_rt(const CheckRtArgs a) {
  const int m = blockIdx.x, lane = threadIdx.x;
  const float* R = a.R21 + 9 * m;
  const float* t = a.t21 + 3 * m;
  // 1. P1 = [K | 0], P2 = K [R | t] (gemm on CV_32F: double accumulation, one rounding), O2 = -R^T t
  float P2[12], O2[3], Rl[9], tl[3];
#pragma unroll
  for (int i = 0; i < 9; i++) Rl[i] = R[i];
#pragma unroll
  for (int i = 0; i < 3; i++) tl[i] = t[i];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 4; c++) {
      double sd = 0;
#pragma unroll
      for (int k = 0; k < 3; k++) sd += (double)a.K[r * 3 + k] * (double)(c < 3 ? Rl[k * 3 + c] : tl[k]);
      P2[r * 4 + c] = (float)sd;
    }
#pragma unroll
  for (int r = 0; r < 3; r++) {
    double sd = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) sd += (double)Rl[k * 3 + r] * (double)tl[k];
    O2[r] = (float)(-1.0 * sd);
  }
  uint8_t* good = a.good + (long long)m * a.n1;
  float* p3d = a.p3d + (long long)m * a.n1 * 3;
  float* cosBuf = a.cosBuf + (long long)m * a.nInl;
  for (int i = lane; i < a.n1; i += 64) { good[i] = 0; p3d[3 * i] = 0.f; p3d[3 * i + 1] = 0.f; p3d[3 * i + 2] = 0.f; }
  __syncthreads();  // (one wave: orders the zeroing before the bookings below)
  int nGood = 0;
  for (int i0 = 0; i0 < a.nInl; i0 += 64) {
    const int i = i0 + lane;
    bool counted = false;
    float cosParallax = 0.f;
    if (i < a.nInl) {
      const float4 p = reinterpret_cast<const float4*>(a.pts)[i];
      const float u1 = p.x, v1 = p.y, u2 = p.z, v2 = p.w;
      double A[16];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const double p1r0 = k < 3 ? (double)a.K[k] : 0.0, p1r1 = k < 3 ? (double)a.K[3 + k] : 0.0, p1r2 = k < 3 ? (double)a.K[6 + k] : 0.0;
        A[0 * 4 + k] = (double)u1 * p1r2 - p1r0;
        A[1 * 4 + k] = (double)v1 * p1r2 - p1r1;
        A[2 * 4 + k] = (double)u2 * (double)P2[2 * 4 + k] - (double)P2[0 * 4 + k];
        A[3 * 4 + k] = (double)v2 * (double)P2[2 * 4 + k] - (double)P2[1 * 4 + k];
      }
      double xd[4];
      smallestRightSingularVector4(A, xd);
      const float X0 = (float)xd[0], X1 = (float)xd[1], X2 = (float)xd[2], X3 = (float)xd[3];
      const int book = a.book[i];
      const float invW = (float)(1.0 / (double)X3);
      const float xn0 = X0 * invW, xn1 = X1 * invW, xn2 = X2 * invW;
      const bool finite3 = isfinite(X0) && isfinite(X1) && isfinite(X2);
      const bool zero3 = X0 == 0 && X1 == 0 && X2 == 0;
      if (finite3 && !zero3) {
        const float oc0 = xn0 - O2[0], oc1 = xn1 - O2[1], oc2 = xn2 - O2[2];
        const float dist1 = (float)sqrt((double)xn0 * xn0 + (double)xn1 * xn1 + (double)xn2 * xn2);
        const float dist2 = (float)sqrt((double)oc0 * oc0 + (double)oc1 * oc1 + (double)oc2 * oc2);
        const double dot = (double)xn0 * oc0 + (double)xn1 * oc1 + (double)xn2 * oc2;
        cosParallax = (float)(dot / (double)(dist1 * dist2));
        float xc2[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
          const double sd = (double)Rl[r * 3] * (double)xn0 + (double)Rl[r * 3 + 1] * (double)xn1 + (double)Rl[r * 3 + 2] * (double)xn2;
          xc2[r] = (float)(1.0 * sd + 1.0 * (double)tl[r]);
        }
        const float invZc2 = (float)(1.0 / (double)xc2[2]);
        const float xc2n0 = xc2[0] * invZc2, xc2n1 = xc2[1] * invZc2, xc2n2 = xc2[2] * invZc2;
        const bool far = !((double)cosParallax < 0.99998);
        if ((xn2 > 0 || far) && (xc2n2 > 0 || far)) {
          const float invZ1 = (float)(1.0 / (double)xn2);
          const float im1x = a.K[0] * xn0 * invZ1 + a.K[2], im1y = a.K[4] * xn1 * invZ1 + a.K[5];
          const float e1 = (im1x - u1) * (im1x - u1) + (im1y - v1) * (im1y - v1);
          const float invZ2 = (float)(1.0 / (double)xc2n2);
          const float im2x = a.K[0] * xc2n0 * invZ2 + a.K[2], im2y = a.K[4] * xc2n1 * invZ2 + a.K[5];
          const float e2 = (im2x - u2) * (im2x - u2) + (im2y - v2) * (im2y - v2);
          if (!(e1 > a.th2 || e2 > a.th2)) {
            counted = true;
            p3d[3 * book] = xn0; p3d[3 * book + 1] = xn1; p3d[3 * book + 2] = xn2;
            if (!far) good[book] = 1;
          }
        }
      }
    }
    const unsigned long long mc = __ballot(counted);
    if (counted) cosBuf[nGood + __popcll(mc & ((1ull << lane) - 1ull))] = cosParallax;
    nGood += (int)__popcll(mc);
  }
  __syncthreads();  // the cosines are read by other lanes below
  // 4. parallax = acos of the idx-th smallest cosine, idx = min(50, nGood - 1): rank selection (any order of equal values
  //    gives the same element)
  if (lane == 0) a.nGood[m] = nGood;
  if (nGood == 0) {
    if (lane == 0) a.parallax[m] = 0.f;
    return;
  }
  const int idx = min(50, nGood - 1);
  for (int j = lane; j < nGood; j += 64) {
    const float v = cosBuf[j];
    int rank = 0;
    for (int k = 0; k < nGood; k++) {
      const float w = cosBuf[k];
      rank += (w < v) || (w == v && k < j);
    }
    if (rank == idx) a.parallax[m] = (float)(acos((double)v) * 180 / 3.14159265358979323846);
  }
}

hipError_t launch_check_rt(hipStream_t st, int nModels, const CheckRtArgs& a) {
  if (nModels <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_check_rt, dim3(nModels), dim3(64), 0, st, a);
  return hipGetLastError();
}

}  // namespace orbx
