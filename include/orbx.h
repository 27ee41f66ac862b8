/* orbx.h — C ABI of the MI355X-native ORB extract + initialization-match hot path.
 *
 * Drop-in boundary for zeal-up/ORB_SLAM_Tracking's
 *   ORBextractor::ORBextractor / operator()            (Features/ORBextractor.hpp:68-85, .cpp:492-595, 1531-1653)
 *   ORBextractor getters + mvImagePyramid              (Features/ORBextractor.hpp:87-111)
 *   ORBmatcher::SearchForInitialization                (Features/ORBmatcher.hpp:36, .cpp:11-150)
 *   Frame grid rules the matcher depends on            (SlamTypes/Frame.cpp:70-99, 163-206)
 * Plain pointers and sizes only; no C++ or torch types cross this boundary.  The C++ classes with the
 * reference's own signatures live in include/orbx_shim.hpp and call only the functions below.
 *
 * All compute runs in hand-written HIP kernels for gfx950 (liborbx.so).  There is no CPU fallback:
 * every entry point that computes returns ORBX_E_HIP when no usable device / kernel image exists.
 *
 * Threading: one orbx_ctx is bound to one device; calls on one ctx must be
 * serialised by the caller; different ctxs are independent (multi-GPU = one ctx per device/process).
 */
#ifndef ORBX_H_
#define ORBX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes (negative).  ORBX_E_EMPTY == -1 keeps `operator()`'s "return -1 on empty image"
 * (Features/ORBextractor.cpp:1536). */
#define ORBX_OK 0
#define ORBX_E_EMPTY (-1)
#define ORBX_E_BADARG (-2)
#define ORBX_E_TOOSMALL (-3) /* a pyramid level is narrower than one FAST cell: UB upstream (cpp:1071-1074) */
/* Documented deviation: a frame wider or taller than ORBX_MAX_FRAME_DIM pixels returns ORBX_E_BADARG (orbx_last_error says so).
 * The reference has no such limit (Features/ORBextractor.cpp:1531-1545 takes any cv::Mat); here a FAST candidate is one 32-bit
 * word -- x and y in 12 bits each, the score in 8 -- and the selection kernels' path-code tables are indexed by those
 * coordinates.  4096 x 4096 covers every BASELINE configuration (the largest is 3840 x 2160). */
#define ORBX_MAX_FRAME_DIM 4096
#define ORBX_E_HIP (-4)
#define ORBX_E_CAPACITY (-5)
#define ORBX_E_RCCL (-6)     /* the multi-device context could not load / initialise RCCL, or a collective failed */

/* mirrors cv::KeyPoint (28 bytes): pt.x pt.y size angle response octave class_id */
typedef struct orbx_keypoint {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} orbx_keypoint;

/* ORBextractor ctor arguments (Features/ORBextractor.hpp:68-69) */
typedef struct orbx_params {
  int32_t nfeatures;
  float scale_factor;
  int32_t nlevels;
  int32_t ini_th_fast;
  int32_t min_th_fast;
} orbx_params;

/* Frame::mnMinX/mnMaxX/mnMinY/mnMaxY (SlamTypes/Frame.hpp, static ints) */
typedef struct orbx_bounds {
  int32_t min_x, max_x, min_y, max_y;
} orbx_bounds;

/* the three counters SearchForInitialization prints (Features/ORBmatcher.cpp:144-147) */
typedef struct orbx_match_stats {
  int32_t invalid_by_distance, invalid_by_ratio, invalid_by_orientation;
} orbx_match_stats;

typedef struct orbx_ctx orbx_ctx;

/* ---- lifetime ---------------------------------------------------------------------------- */
/* Creates a context on `device_id` sized for batches of up to `max_batch` frames of up to
 * max_width x max_height pixels.  The sizes are an initial reservation, not a limit: like
 * ORBextractor::operator() (cpp:1531-1545), every extraction entry point takes any frame size up to ORBX_MAX_FRAME_DIM in either
 * direction (see there), and a call
 * that brings a larger frame or batch drains what is in flight and re-allocates the context's buffers
 * (a one-off cost of milliseconds; results are unaffected).  `stream` is an optional hipStream_t (as
 * void*) the caller wants the work issued on (e.g. torch's current stream); NULL = the ctx creates its own.
 * Replaces ORBextractor::ORBextractor (cpp:492-595); scaleFactor==1 with nlevels>1 returns
 * ORBX_E_BADARG instead of exit(1) (cpp:502-505). */
int orbx_create(const orbx_params* params, int device_id, int max_width, int max_height, int max_batch,
                void* stream, orbx_ctx** out);
void orbx_destroy(orbx_ctx* ctx);

/* The two constants of the path that depend on the OpenCV release the reference is linked against (its CMakeLists.txt:13 pins
 * none), selectable per context so that the library can be diffed against any OpenCV-linked build without touching a kernel:
 *   gaussian_variant  Q8 taps of cv::GaussianBlur(7x7, sigma 2) on 8-bit images (ORBextractor.cpp:1601):
 *     ORBX_GAUSS_ERROR_DIFFUSION (default)  [18,34,48,56,48,34,18], sum 256: getGaussianKernelFixedPoint_ED, OpenCV >= 4.1.1 / 3.4.7
 *     ORBX_GAUSS_ROUNDED                    [18,34,49,55,49,34,18], sum 257: each tap rounded, the bit-exact path of OpenCV
 *                                           3.4.1 .. 3.4.6 / 4.0 .. 4.1.0 and the integer filter of older releases
 *   gray_variant      cv::cvtColor RGB/BGR -> gray on 8-bit (Converter.cpp:11-13):
 *     ORBX_GRAY_14BIT (default)             (R*4899 + G*9617 + B*1868 + 2^13) >> 14: OpenCV 3.x .. 4.0
 *     ORBX_GRAY_15BIT                       (R*9798 + G*19235 + B*3735 + 2^14) >> 15: OpenCV >= 4.1
 * Takes effect for the calls that follow (batches in flight are waited for; like every call that has to wait for an earlier
 * stream-ordered batch, this one returns that batch's error WITHOUT having done its own work: call it again). */
#define ORBX_GAUSS_ERROR_DIFFUSION 0
#define ORBX_GAUSS_ROUNDED 1
#define ORBX_GRAY_14BIT 0
#define ORBX_GRAY_15BIT 1
int orbx_set_opencv_variant(orbx_ctx* ctx, int gaussian_variant, int gray_variant);

/* The libm reading of the reference's two unqualified <cmath> calls on floats: cos(angle) / sin(angle) of computeOrbDescriptor
 * (Features/ORBextractor.cpp:174) and pow(factor, (float)nlevels) of the constructor (cpp:536).  The file has no `using namespace
 * std` (cpp:69-71 import list / pair / vector only), so which function a call resolves to depends on the headers in the
 * translation unit (DESIGN.md section 2 derives it from the reference's includes):
 *   ORBX_LIBM_DOUBLE            only ::cos(double) / ::pow(double, double) are visible in the global namespace (<cmath> alone):
 *                               the float is promoted, the double result converted back -- (float)cos((double)angle).  The
 *                               default of rounds 1-4.
 *   ORBX_LIBM_FLOAT (default)   libstdc++'s <math.h> wrapper is in the include chain (`using std::cos;` ...): std::cos(float) =
 *                               cosf, std::pow(float, float) = powf.  The default since round 5: <opencv2/opencv.hpp>
 *                               (Features/ORBextractor.hpp:24) brings <math.h> in through opencv2/flann/lsh_table.h [from knowledge of
 *                               OpenCV 3.x / 4.x; tools/pin_opencv/libm_probe.cpp lets the compiler confirm it against a real one],
 *                               and the ORB-SLAM sources this file descends from say `using namespace std`.  cosf / sinf = glibc >= 2.28's algorithm, restated in the
 *                               kernel operation for operation (it is not correctly rounded: of the 1,135,869,953 f32 angles
 *                               in [0, 360], 1,484,894 give a different (cos, sin) pair than ORBX_LIBM_DOUBLE, and for 96 of
 *                               them a rotated sample point lands on another pixel); powf = the host libm's.
 *                               SCOPE OF THE CLAIM: "FLOAT" means the sinf / cosf of glibc >= 2.28 (2018; the
 *                               s_sincosf.h algorithm) on an x86-64 host -- with or without FMA contraction, both forms were
 *                               checked against the host's libm for every angle.  A reference built on an OLDER glibc (Ubuntu
 *                               16.04 / 18.04: 2.23 / 2.27) ran another sinf / cosf, which NEITHER reading reproduces exactly;
 *                               and powf(factor, nlevels) in the constructor is the DEPLOY host's libm, so for the rare
 *                               (factor, nlevels, nfeatures) combinations above two hosts may plan different per-level quotas.
 *                               The default stays a reasoned choice, not a measured one, until tools/pin_opencv/libm_probe has
 *                               been compiled against a real OpenCV (neither this image nor the GPU
 *                               box has one).
 * Takes effect for the calls that follow (batches in flight are waited for; an earlier batch's error is returned as by
 * orbx_set_opencv_variant).  If the constructor's pow changes a per-level quota (it does for 138 of 8e8 tried (factor, nlevels,
 * nfeatures) combinations and never for a scale factor with two decimals), the context's buffers are re-planned. */
#define ORBX_LIBM_DOUBLE 0
#define ORBX_LIBM_FLOAT 1
#define ORBX_LIBM_DEFAULT ORBX_LIBM_FLOAT
int orbx_set_libm_variant(orbx_ctx* ctx, int libm_variant);
const char* orbx_last_error(const orbx_ctx* ctx);

/* ---- getters (Features/ORBextractor.hpp:87-108) ------------------------------------------- */
int orbx_get_levels(const orbx_ctx* ctx);
float orbx_get_scale_factor(const orbx_ctx* ctx);
/* each out array has nlevels entries; any pointer may be NULL */
int orbx_get_tables(const orbx_ctx* ctx, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                    int32_t* features_per_level);
int orbx_get_umax(const orbx_ctx* ctx, int32_t* umax16);

/* ---- extraction: ORBextractor::operator() (cpp:1531-1653) --------------------------------- */
/* One frame from host memory.  Returns monoIndex (>= 0; == *n_out when lap0 == lap1 == 0, as in
 * Frame.cpp:58-60) or a negative error.  `kps`/`desc32` must hold `capacity` >= nfeatures entries
 * (keypoints 28 B, descriptors 32 B each). */
int orbx_extract(orbx_ctx* ctx, const uint8_t* img, int width, int height, int stride, int lap0, int lap1,
                 orbx_keypoint* kps, uint8_t* desc32, int capacity, int* n_out);

/* Optional, for a host that keeps its frames in buffers of its own (a camera ring, a decoder's pool): page-locks [ptr, ptr + bytes)
 * so that orbx_extract / orbx_extract_batch copy a frame out of it with one asynchronous DMA instead of through the runtime's
 * staging copy of pageable memory (measured at 640x480: 1-2 us of the call's ~85, BENCH_r04 single_frame.cpp_shim 0.0849 against
 * 0.0836 ms; more for large frames, and it is what keeps orbx_extract_match_batch_host_async's uploads asynchronous).  Thin wrappers over hipHostRegister / hipHostUnregister
 * (a C++ host need not link HIP); the buffer must be unregistered before it is freed.  Results are the same either way. */
int orbx_host_register(orbx_ctx* ctx, void* ptr, size_t bytes);
int orbx_host_unregister(orbx_ctx* ctx, void* ptr);

/* `n_frames` same-sized frames from host memory (frame f at imgs + f*frame_stride_bytes).
 * Outputs: frame f's keypoints at kps + f*capacity, descriptors at desc32 + f*capacity*32,
 * counts n_out[f], return values (monoIndex) mono_out[f] (may be NULL). */
int orbx_extract_batch(orbx_ctx* ctx, int n_frames, const uint8_t* imgs, int width, int height, int stride,
                       size_t frame_stride_bytes, int lap0, int lap1, orbx_keypoint* kps, uint8_t* desc32,
                       int capacity, int* n_out, int* mono_out);

/* Same, but the frames are already resident in device memory (HBM) and results stay there:
 * d_kps / d_desc32 / d_n_out are device pointers (d_n_out: int32[n_frames]).  Nothing is copied
 * to the host except two error flags and the per-frame counts; there is no host round trip between
 * the stages.  The call returns after the work has completed on the ctx stream. */
int orbx_extract_batch_device(orbx_ctx* ctx, int n_frames, const uint8_t* d_imgs, int width, int height,
                              int stride, size_t frame_stride_bytes, orbx_keypoint* d_kps, uint8_t* d_desc32,
                              int capacity, int32_t* d_n_out);

/* mvImagePyramid (hpp:111): copies level `level` of frame `frame` of the last extract call to host.
 * `border` = 0 copies the w x h level; border = 19 reproduces the reference's REFLECT_101 ring
 * (cpp:1689,1708), dst then is (w+38) x (h+38).  dst_stride in bytes. */
int orbx_level_size(const orbx_ctx* ctx, int level, int* width, int* height);
int orbx_download_pyramid(orbx_ctx* ctx, int frame, int level, int border, uint8_t* dst, int dst_stride);

/* ---- matching: ORBmatcher::SearchForInitialization (ORBmatcher.cpp:11-150) ----------------- */
/* k1/d1 = F1.mvKeysUn / F1.mDescriptors (n1 entries), k2/d2 = F2's; `bounds` = Frame::mnMin/Max*;
 * the 64x48 grid of F2 is rebuilt inside with Frame::PosInGrid's rule (Frame.cpp:89-99), so callers
 * do not pass mGrid.  matches12 has n1 entries (-1 = none).  *nmatches receives the reference's return value
 * exactly as it computes it (its double-decrement quirk can make it differ from the number of non-negative
 * entries, even negative), which is why it is an out-parameter; the function returns ORBX_OK or an error. */
int orbx_match_init(orbx_ctx* ctx, const orbx_keypoint* k1, const uint8_t* d1, int n1, const orbx_keypoint* k2,
                    const uint8_t* d2, int n2, const orbx_bounds* bounds, int window_size, float nnratio,
                    int check_orientation, int32_t* matches12, int32_t* nmatches, orbx_match_stats* stats);

/* Batched, device-resident: pair p matches frame first[p] against frame second[p] of the arrays a
 * previous orbx_extract_batch_device call filled (same capacity / layout).  d_matches12 is
 * int32[n_pairs*capacity], d_nmatches int32[n_pairs], d_stats (nullable) int32[n_pairs*3];
 * h_first/h_second are host arrays. */
int orbx_match_init_batch_device(orbx_ctx* ctx, int n_pairs, const int32_t* h_first, const int32_t* h_second,
                                 const orbx_keypoint* d_kps, const uint8_t* d_desc32, const int32_t* d_n,
                                 int capacity, const orbx_bounds* bounds, int window_size, float nnratio,
                                 int check_orientation, int32_t* d_matches12, int32_t* d_nmatches,
                                 int32_t* d_stats);

/* The whole hot path of one batch in one call: orbx_extract_batch_device followed by
 * orbx_match_init_batch_device on pairs of frames of the same batch (first[p], second[p] in [0, n_frames)).
 * Large batches are issued as two half-batches on two HIP streams, and a pair whose frames lie in the same half
 * is matched right behind that half's extraction, so the latency-bound stages (quadtree selection, sequential
 * matching) of one half overlap the arithmetic-bound stages of the other.  Results are identical to the two
 * separate calls.  n_pairs may be 0. */
int orbx_extract_match_batch_device(orbx_ctx* ctx, int n_frames, const uint8_t* d_imgs, int width, int height,
                                    int stride, size_t frame_stride_bytes, orbx_keypoint* d_kps, uint8_t* d_desc32,
                                    int capacity, int32_t* d_n_out, int n_pairs, const int32_t* h_first,
                                    const int32_t* h_second, const orbx_bounds* bounds, int window_size, float nnratio,
                                    int check_orientation, int32_t* d_matches12, int32_t* d_nmatches, int32_t* d_stats);

/* Stream-ordered form of the call above for callers that keep the device busy across batches (a tracker that prepares
 * batch k+1 while batch k runs): issues the batch and returns.  At most two batches are in flight (orbx_set_pipeline_depth
 * changes that) - a third call first waits for the oldest.  The outputs, counts and errors of a batch are valid once an orbx_wait_one / orbx_wait has covered
 * it (an error of an earlier batch can also be returned by the call that has to wait for it); batches in flight together
 * must be given different output arrays; the input frames must stay untouched until their batch has been waited for.
 * Every other call on the context may be used in between (the synchronous ones simply queue behind). */
int orbx_extract_match_batch_device_async(orbx_ctx* ctx, int n_frames, const uint8_t* d_imgs, int width, int height,
                                          int stride, size_t frame_stride_bytes, orbx_keypoint* d_kps, uint8_t* d_desc32,
                                          int capacity, int32_t* d_n_out, int n_pairs, const int32_t* h_first,
                                          const int32_t* h_second, const orbx_bounds* bounds, int window_size,
                                          float nnratio, int check_orientation, int32_t* d_matches12,
                                          int32_t* d_nmatches, int32_t* d_stats);

/* The same stream-ordered call for frames and results in HOST memory -- what the reference's call site hands over (a host
 * cv::Mat per frame, SlamTypes/Frame.cpp:58-60), batched: per batch the frames are uploaded, the kernels run and every result array
 * is copied back, all stream-ordered on the lane the batch goes to, so that with orbx_set_pipeline_depth(N >= 2) the upload of one
 * batch runs under the kernels of the batches in front of it and the call is bound by the slower of the PCIe link and the
 * kernels (640x480: the link; bench.py reports the rate beside the H2D rate of the box as `host_pipeline`).  h_imgs and the
 * result arrays should be page-locked (hipHostMalloc, orbx_host_register): from pageable memory the runtime stages every copy
 * synchronously and the call degenerates to orbx_extract_batch's rate.  All arrays are host pointers with the layout of the
 * device call (h_kps [n_frames][capacity], h_desc32 [n_frames][capacity][32], h_n_out [n_frames], h_matches12 [n_pairs][capacity],
 * h_nmatches [n_pairs], h_stats nullable [n_pairs][3]); entries beyond a frame's count are unspecified.  Valid once an
 * orbx_wait_one / orbx_wait has covered the batch; inputs and outputs must stay untouched until then.  With pipeline depth 0 one
 * batch is in flight at a time (the call first waits for the previous one).  From its first use on, every batch of the context
 * carries all of its matcher kernels (as after orbx_order_before): the copies back run behind the kernels, not behind a wait. */
int orbx_extract_match_batch_host_async(orbx_ctx* ctx, int n_frames, const uint8_t* h_imgs, int width, int height,
                                        int stride, size_t frame_stride_bytes, orbx_keypoint* h_kps, uint8_t* h_desc32,
                                        int capacity, int32_t* h_n_out, int n_pairs, const int32_t* h_first,
                                        const int32_t* h_second, const orbx_bounds* bounds, int window_size,
                                        float nnratio, int check_orientation, int32_t* h_matches12,
                                        int32_t* h_nmatches, int32_t* h_stats);

/* Pipeline depth of the stream-ordered call (default 0).  depth 0: a batch is cut into two halves that run on the context's two
 * streams, at most two batches in flight.  depth >= 1: the context keeps `depth` lanes (each with its own stream and its own
 * copy of the internal buffers, sized like the context) and every stream-ordered batch goes, whole, to the next lane; at most
 * `depth` batches are in flight (the call that would exceed it first waits for the oldest), so batches in flight together need
 * `depth` different output arrays.  Whole batches on four lanes measure 341 k frames/s against 323 k with the two halves at
 * 256 frames 640x480 per batch, and 239 k against 109 k at 32 frames per batch on three lanes (MI355X, rounds 3-4).  Waits for
 * everything in flight (and returns an earlier batch's error without changing the depth: call it again); 0 <= depth <= 8; costs depth times the context's
 * device memory; not available (ORBX_E_BADARG) on a context created on a caller's stream, whose order the lanes' own streams could
 * not keep.  orbx_download_pyramid and the profile / debug hooks of the context keep referring to the batches the context
 * ran itself (synchronous calls, depth 0); orbx_profile_get adds what the lanes ran.  (The reference has no counterpart: it
 * extracts one frame per call, Frame.cpp:58-60.) */
int orbx_set_pipeline_depth(orbx_ctx* ctx, int depth);
int orbx_wait_one(orbx_ctx* ctx); /* the oldest batch in flight (ORBX_OK if there is none) */
int orbx_wait(orbx_ctx* ctx);     /* all batches in flight */

/* Stream ordering against a caller's stream.  The context issues its work on its own HIP streams, which nothing
 * orders against the stream the caller produces frames / consumes results on (e.g. torch's current stream):
 *   orbx_order_after(ctx, s):  work issued on ctx from now on starts after everything queued on `s` so far
 *                              (call it between producing the frames on `s` and the device-resident entry points);
 *   orbx_order_before(ctx, s): work queued on `s` from now on starts after everything issued on ctx so far
 *                              (for a consumer that reads the outputs of an _async batch without a host-side wait).
 * `s` is a hipStream_t as void* (NULL = the legacy default stream).  Both only record / wait on events, with one exception:
 * the FIRST orbx_order_before on a context switches it to "event-ordered" mode -- from then on every batch carries all of its
 * matcher kernels (normally the kernels for pairs beyond the fast matcher's tables are issued only while batches need them, and
 * a batch that needed them without having them is completed inside its orbx_wait), and batches already in flight that were
 * issued without them are completed right there with a host-side wait.  After it, outputs (keypoints, descriptors, counts,
 * matches12, nmatches, statistics) read behind the event are final.  Error codes (ORBX_E_CAPACITY of the selection stage) are
 * still reported by orbx_wait_one / orbx_wait only, which an event-ordered consumer calls later, at its leisure. */
int orbx_order_after(orbx_ctx* ctx, void* stream);
int orbx_order_before(orbx_ctx* ctx, void* stream);

/* ---- between extractor and matcher: Frame::UndistortKeyPoints / ComputeImageBounds ---------- */
/* (SlamTypes/Frame.cpp:101-161)  The camera as the reference holds it: mK's four entries and
 * mDistCoef = (k1, k2, p1, p2), all CV_32F (Config/Settings.hpp:28-39). */
typedef struct orbx_camera {
  float fx, fy, cx, cy;
  float k1, k2, p1, p2;
} orbx_camera;

/* mvKeysUn of one frame, host memory: out[i] = kps[i] with pt replaced by cv::undistortPoints(pt, K, dist, R = I, P = K)
 * (5 fixed iterations in f64, result stored as f32).  k1 == 0 copies the keypoints unchanged (Frame.cpp:138-142). */
int orbx_undistort_keypoints(orbx_ctx* ctx, const orbx_keypoint* kps, int n, const orbx_camera* cam, orbx_keypoint* out);

/* Batched, device-resident: frame f's d_n[f] keypoints at d_kps + f*capacity -> d_kps_un + f*capacity (the layout of
 * orbx_extract_batch_device; d_kps_un may equal d_kps).  Feed d_kps_un to orbx_match_init_batch_device. */
int orbx_undistort_batch_device(orbx_ctx* ctx, int n_frames, const orbx_keypoint* d_kps, const int32_t* d_n, int capacity,
                                const orbx_camera* cam, orbx_keypoint* d_kps_un);

/* Frame::ComputeImageBounds (Frame.cpp:101-134): the four image corners undistorted on the device, min/max truncated
 * to the reference's static ints; k1 == 0 gives (0, width, 0, height). */
int orbx_image_bounds(orbx_ctx* ctx, const orbx_camera* cam, int width, int height, orbx_bounds* out);

/* ---- in front of the extractor: Converter::toGray (Utils/Converter.cpp:5-19) -------- */
/* channels == 1 copies, channels == 3 is cv::cvtColor(COLOR_RGB2GRAY if rgb else COLOR_BGR2GRAY) on 8-bit pixels,
 * Y = (R*4899 + G*9617 + B*1868 + 8192) >> 14; any other channel count returns ORBX_E_BADARG ("Wrong image format",
 * the reference's `return false`).  Strides in bytes.  Host buffers: */
int orbx_to_gray(orbx_ctx* ctx, const uint8_t* img, int width, int height, int stride, int channels, int rgb, uint8_t* gray,
                 int gray_stride);
/* device-resident batch: frame f at d_src + f*frame_stride_bytes -> d_gray + f*gray_frame_stride_bytes, ready to be
 * handed to orbx_extract_batch_device / orbx_extract_match_batch_device. */
int orbx_to_gray_batch_device(orbx_ctx* ctx, int n_frames, const uint8_t* d_src, int width, int height, int stride,
                              size_t frame_stride_bytes, int channels, int rgb, uint8_t* d_gray, int gray_stride,
                              size_t gray_frame_stride_bytes);

/* ---- behind the matcher: the Initializer's model scoring (Initialization/Initializer.cpp:268-438) -- */
/* CheckHomography / CheckFundamental for `n_models` RANSAC hypotheses at once (the loops of FindHomography :180-211 and
 * FindFundamental :231-265 call them once per iteration).  H21 / H12 / F21: n_models row-major 3x3 matrices (Eigen's
 * (row, col)); k1 / k2 = mvKeys1 / mvKeys2 (the frames' mvKeysUn); matches12 = the matcher's vnMatches12 (n1 entries),
 * from which mvMatches12 is built as in Initializer::Initialize :24-33 (pairs (i, matches12[i]) with matches12[i] >= 0, in
 * order).  scores[m] = the returned score, inliers[m * N + i] = vbMatchesInliers[i] for the N = *n_matches_out pairs (the
 * caller provides n_models * n1 bytes).  best (nullable) = the hypothesis the RANSAC loop would keep: first maximum of the
 * scores, -1 if none is > 0.  Host pointers. */
int orbx_check_homography(orbx_ctx* ctx, int n_models, const float* H21, const float* H12, const orbx_keypoint* k1, int n1,
                          const orbx_keypoint* k2, int n2, const int32_t* matches12, float sigma, float* scores,
                          uint8_t* inliers, int* n_matches_out, int* best);
int orbx_check_fundamental(orbx_ctx* ctx, int n_models, const float* F21, const orbx_keypoint* k1, int n1,
                           const orbx_keypoint* k2, int n2, const int32_t* matches12, float sigma, float* scores,
                           uint8_t* inliers, int* n_matches_out, int* best);

/* Initializer::CheckRT (Initialization/Initializer.cpp:569-713) for `n_models` (R21, t21) hypotheses at once -- ReconstructHF
 * (:440-567) calls it once per candidate of cv::decomposeEssentialMat / cv::decomposeHomographyMat with the same matches and
 * inlier flags.  R21: n_models row-major 3x3, t21: n_models x 3, K: row-major 3x3 (all CV_32F upstream); k1 / k2 / matches12 as
 * above; matches_inliers[i] = vbMatchesInliers[i] for the N pairs of mvMatches12; th2 = the squared reprojection threshold
 * (4 * sigma^2 upstream).  Outputs per model m: n_good[m] = the return value, tri_good[m * n1 + k] = vbTriGood[k],
 * p3d[(m * n1 + k) * 3 ..] = vP3D[k] (zeros where the reference books nothing), parallax[m] in degrees.  cv::triangulatePoints
 * is restated as a DLT with a fixed one-sided Jacobi SVD in f64; the reference's two quirks in this function (booking under
 * the compacted index, camera-2 depth test on z / z) are kept: see oracle/orbx_oracle.cpp.  Floating point: n_good and
 * tri_good equal the CPU restatement, points and parallax agree with it to 1e-4 relative (in practice bitwise).  Host pointers. */
int orbx_check_rt(orbx_ctx* ctx, int n_models, const float* R21, const float* t21, const float* K, const orbx_keypoint* k1, int n1,
                  const orbx_keypoint* k2, int n2, const int32_t* matches12, const uint8_t* matches_inliers, float th2,
                  int32_t* n_good, uint8_t* tri_good, float* p3d, float* parallax);

/* ---- measurement hooks (bench.py; HIP events on the ctx stream) ---------------------------- */
#define ORBX_STAGE_PYRAMID 0
#define ORBX_STAGE_FAST 1
#define ORBX_STAGE_SELECT 2 /* quadtree selection incl. its transfers */
#define ORBX_STAGE_DESCRIBE 3
#define ORBX_STAGE_MATCH 4
#define ORBX_STAGE_COUNT 5
/* enable: record hipEvents around every stage; accumulated device ms and launch counts since the
 * last reset are returned by orbx_profile_get (arrays of ORBX_STAGE_COUNT). */
int orbx_profile_enable(orbx_ctx* ctx, int on);
/* the same for a subset of the stages (bit s = ORBX_STAGE_s): every bracketed stage costs two event records on its stream */
int orbx_profile_stages(orbx_ctx* ctx, unsigned stage_mask);
int orbx_profile_reset(orbx_ctx* ctx);
int orbx_profile_get(orbx_ctx* ctx, double* ms, int64_t* launches);

/* ---- several MI355X from one host process ---------------------------------------- */
/* One orbx_ctx per device; a batch is cut into contiguous even-sized blocks, one per device (consecutive pairs (2k, 2k + 1)
 * never straddle devices); the only exchange is an ncclAllGather of the per-frame keypoint counts over xGMI.  RCCL is loaded
 * with dlopen when a context with more than one device is created (ORBX_E_RCCL if that fails); device ids must be distinct.
 * ORBX_MULTI_FORCE_RCCL=1 in the environment makes a ONE-device context go through RCCL as well (dlopen, ncclCommInitAll(1),
 * grouped ncclAllGather on the collective stream, ncclCommDestroy): the way to exercise these code paths on a one-GPU box.
 * If a device fails to queue its block of a batch, the batches issued before it are still completed as documented below
 * (blocks waited for, counts gathered into their counts_all) before the call returns the error; only the partly issued batch
 * is dropped. */
typedef struct orbx_multi orbx_multi;
int orbx_multi_create(const orbx_params* params, int n_devices, const int* device_ids, int max_width, int max_height,
                      int max_batch_per_device, orbx_multi** out);
void orbx_multi_destroy(orbx_multi* m);
int orbx_multi_size(const orbx_multi* m);
orbx_ctx* orbx_multi_ctx(orbx_multi* m, int r); /* device r's context, for every other call of this header */
const char* orbx_multi_last_error(const orbx_multi* m);
/* the block [*lo, *hi) of a batch of n_frames that device r of n_devices owns */
int orbx_multi_shard_range(int n_frames, int n_devices, int r, int* lo, int* hi);
/* d_*[r] = device r's arrays for ITS block (frames already resident in its HBM; layout of orbx_extract_match_batch_device).
 * Every device extracts its block and matches the block's consecutive pairs; counts_all (host, n_frames entries) receives the
 * all-gathered keypoint counts in global frame order (every device holds the same copy). */
int orbx_multi_extract_match_batch_device(orbx_multi* m, int n_frames, const uint8_t* const* d_imgs, int width, int height, int stride,
                                          size_t frame_stride_bytes, orbx_keypoint* const* d_kps, uint8_t* const* d_desc32, int capacity,
                                          int32_t* const* d_n_out, const orbx_bounds* bounds, int window_size, float nnratio,
                                          int check_orientation, int32_t* const* d_matches12, int32_t* const* d_nmatches,
                                          int32_t* counts_all);

/* The throughput form (what bench.py does with one process per GPU, for a C / C++ host): one issuing host thread per device, the
 * blocks of a batch issued stream-ordered on every device's context, several batches in flight.
 *   orbx_multi_set_pipeline_depth: orbx_set_pipeline_depth(depth) on every device's context (whole blocks on `depth` lanes;
 *     0 = two half blocks on two streams); at most max(depth, 2) batches are in flight;
 *   orbx_multi_extract_match_batch_device_async: returns once every device has queued its block; a call that would exceed the
 *     batches in flight first completes the oldest one (whose error, if any, it returns);
 *   orbx_multi_wait_one: completes the oldest batch -- waits for its blocks, all-gathers its counts on the collective streams
 *     (RCCL; the devices meanwhile run the batches issued after it) and fills ITS counts_all; orbx_multi_wait: all of them.
 * Batches in flight together need different output arrays (and counts_all arrays); the contexts returned by orbx_multi_ctx must
 * not be used directly while batches are in flight.  orbx_multi_extract_match_batch_device == the async call + orbx_multi_wait. */
int orbx_multi_set_pipeline_depth(orbx_multi* m, int depth);
int orbx_multi_extract_match_batch_device_async(orbx_multi* m, int n_frames, const uint8_t* const* d_imgs, int width, int height,
                                                int stride, size_t frame_stride_bytes, orbx_keypoint* const* d_kps,
                                                uint8_t* const* d_desc32, int capacity, int32_t* const* d_n_out,
                                                const orbx_bounds* bounds, int window_size, float nnratio, int check_orientation,
                                                int32_t* const* d_matches12, int32_t* const* d_nmatches, int32_t* counts_all);
int orbx_multi_wait_one(orbx_multi* m);
int orbx_multi_wait(orbx_multi* m);

/* ---- test hooks (used by tests/ only; stable but not part of the reference surface) -------- */
/* candidates of (frame, level) of the last extract call, as produced by the FAST kernel, sorted
 * into the reference's order (cell row, cell col, y, x): xyr = (x, y, response) triples relative to
 * (minBorderX, minBorderY) like vToDistributeKeys (cpp:1134-1137).  Returns the count. */
int orbx_debug_candidates(orbx_ctx* ctx, int frame, int level, float* xyr, int cap);
/* The selection units (frame, level) of the context's last batch: counts[level] = keypoints the unit selected (negative: it
 * failed), redone[level] (nullable) = 1 if the many-workgroup kernels could not take the unit (a bucket of keys overflowed its slot)
 * and its workgroup redid it with the one-workgroup code -- results are the same, only the time differs. */
int orbx_debug_selection_units(orbx_ctx* ctx, int frame, int32_t* counts, int32_t* redone);
/* The quadtree selection alone, DistributeOctTree (cpp:698-1011), on the device (the kernels the pipeline uses).  Candidates must be given in row-major (y, x)
 * order with integer coordinates in [0, 4095] relative to (min_x, min_y) and integer responses in [0, 255].
 * variant 0 = LDS-resident kernel (redoing a unit that does not fit on global scratch), variant 1 = global-scratch kernel
 * only, variant 2 = the smaller LDS instance (<= 1024 candidates) and variant 3 = the smallest one (<= 512 candidates, quota
 * <= 128) that the pipeline picks per pyramid level when the previous batch allows; variant 4 = the 2048-candidate LDS instance
 * with 64-bit sort keys (variants 0, 2, 3 use 32-bit keys whenever the rectangle's quadtree path codes fit 21 bits); variant 5 =
 * the many-workgroup kernels the pipeline takes for levels with large units (a workgroup per bucket of keys sorts, one per unit
 * does the tree arithmetic; a unit they cannot take falls to the global-scratch kernel), variant 6 = the same without that
 * fallback (ORBX_E_CAPACITY instead). */
int orbx_debug_distribute_device(orbx_ctx* ctx, const float* xyr, int n, int min_x, int max_x, int min_y, int max_y,
                                 int n_features, int variant, float* out_xyr, int cap);
/* The device's literal replay of libstdc++ std::sort with the reference's compareNodes (cpp:684-696, 912) on n
 * (count, UL.x, id) int32 triples, in place. */
int orbx_debug_std_sort(orbx_ctx* ctx, int32_t* triples, int n);
/* The (cos, sin) pair the descriptor kernel uses for a keypoint angle in degrees (f64 evaluation rounded to f32,
 * cpp:173-174), for n angles; host pointers. */
int orbx_debug_sincos(orbx_ctx* ctx, const float* angle_deg, int n, float* cos_out, float* sin_out);

/* How the context's last extraction batch was issued (the kernels behind one entry point depend on batch size, frame size,
 * alignment and the previous batch's statistics): info8 = { [0] pyramid: 1 = k_pyramid_bands (one launch, large batches), 2 = k_pyramid_tiles
 * (one launch, small batches), 0 = one resize launch per level; [1] row bands (tiles) per frame of that kernel; [2] FAST: 1 = k_fast_wave, 0 = k_fast; [3] selection: candidate
 * capacity of the smallest LDS instance a pyramid level of the batch ran on (2048 / 1024 / 512; 0 = a level expected units beyond
 * the LDS layout and went to the global-scratch kernel); [4] bit 0: the batch was cut into two
 * halves on two streams, bit 1: the descriptor kernel took the selection's staging lists itself (small launches: no k_sel_compact); [5] frames per kernel launch; [6] 1 = the wide matcher kernels went with the batch; [7] lane the batch
 * went to (1-based; 0 = the context itself) }. */
int orbx_debug_last_launch(const orbx_ctx* ctx, int32_t* info8);
/* Which matcher kernels did the work (cumulative since orbx_create, for the context itself -- a lane has its own): info4 = { [0] blocks
 * of 256 queries whose candidate lists k_match_bf_mfma built on the matrix cores (a brute-force block of a launch with at least 256
 * such blocks; everything else is listed by k_match_wide_lists on the vector ALU), [1..3] reserved (0) }.  Waits for the context's
 * batches in flight.  Tests take the difference around a call to see that the matrix path ran (or, under the knob match_no_mfma,
 * did not). */
int orbx_debug_match_counters(orbx_ctx* ctx, uint32_t* info4);
/* Diagnostic knobs (process-wide; the library itself reads NO environment variable): a named integer that makes the launches that
 * follow take a particular kernel or launch shape -- e.g. "fast_wg_max_cells" = 0 sends the FAST cells of small batches through
 * k_fast_wave, "no_split" = 1 keeps a synchronous call on one stream.  None changes a result; tests use them to run one input
 * through every kernel, the profiling tools to look at a kernel alone.  The names are listed in csrc/orbx_knobs.h; the Python
 * loader forwards ORBX_<NAME> environment variables here.  value == LLONG_MIN unsets a knob.  ORBX_E_BADARG: unknown name. */
int orbx_debug_set(const char* key, long long value);
/* Host only, no device needed: the quadtree path codes (root << 32 | 16 quadrant digits of ExtractorNode::DivideNode,
 * Features/ORBextractor.cpp:617-676, 747) of n points (xs[i], ys[i]) of a width x height candidate region, from the per-level
 * x / y tables the selection kernels look up and by walking the 16 splits per point; the two agree for every point. */
int orbx_debug_path_codes(int width, int height, int n, const int32_t* xs, const int32_t* ys, uint64_t* from_tables,
                          uint64_t* from_walk);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H_ */
