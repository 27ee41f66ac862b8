"""ctypes wrapper around oracle/liborbx_oracle.so — the CPU restatement of the reference algorithm.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, and only as
the checker / timed CPU baseline.  The product (orb_slam_tracking_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SO = os.environ.get("ORBX_ORACLE_SO") or os.path.join(ORACLE_DIR, "liborbx_oracle.so")  # (the sanitizer test points this at the ASan build)

KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"),
               ("class_id", "<i4")])

_L = None


def build(force: bool = False) -> str:
    src = os.path.join(ORACLE_DIR, "orbx_oracle.cpp")
    if os.environ.get("ORBX_ORACLE_SO"):
        return SO
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return SO


def lib() -> ctypes.CDLL:
    global _L
    if _L is not None:
        return _L
    build()
    L = ctypes.CDLL(SO)
    vp, i32, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    L.orbo_create.restype = vp
    L.orbo_create.argtypes = [i32, f32, i32, i32, i32]
    L.orbo_destroy.argtypes = [vp]
    L.orbo_destroy.restype = None
    L.orbo_get_tables.argtypes = [vp] * 7
    L.orbo_get_tables.restype = None
    L.orbo_extract.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp]
    L.orbo_keep_blurred.argtypes = [vp, i32]
    L.orbo_keep_blurred.restype = None
    L.orbo_level_size.argtypes = [vp, i32, vp, vp]
    L.orbo_level_image.argtypes = [vp, i32, vp]
    L.orbo_level_blurred.argtypes = [vp, i32, vp]
    L.orbo_level_candidates.argtypes = [vp, i32, vp, i32]
    L.orbo_level_selected.argtypes = [vp, i32, vp, i32]
    L.orbo_resize_linear.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32]
    L.orbo_resize_linear.restype = None
    L.orbo_gaussian7.argtypes = [vp, i32, i32, i32, vp, i32]
    L.orbo_gaussian7.restype = None
    L.orbo_fast.argtypes = [vp, i32, i32, i32, i32, i32, vp, i32]
    L.orbo_fast_strength.argtypes = [vp, i32, i32, i32, i32, i32]
    L.orbo_distribute.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, i32]
    L.orbo_fast_atan2.argtypes = [f32, f32]
    L.orbo_fast_atan2.restype = f32
    L.orbo_ic_angle.argtypes = [vp, i32, i32, f32, f32, vp, vp]
    L.orbo_ic_angle.restype = f32
    L.orbo_descriptor.argtypes = [vp, i32, i32, f32, f32, f32, vp]
    L.orbo_descriptor.restype = None
    L.orbo_sincos_deg.argtypes = [f32, vp, vp]
    L.orbo_sincos_deg.restype = None
    L.orbo_hamming.argtypes = [vp, vp]
    L.orbo_pos_in_grid.argtypes = [vp, i32, vp, vp, vp, vp]
    L.orbo_pos_in_grid.restype = None
    L.orbo_features_in_area.argtypes = [vp, i32, vp, f32, f32, f32, i32, i32, vp, i32]
    L.orbo_match_init.argtypes = [vp, vp, i32, vp, vp, i32, vp, i32, f32, i32, vp, vp]
    L.orbo_bench_pairs.argtypes = [i32, f32, i32, i32, i32, vp, i32, i32, i32, i32, f32, i32, i32, vp, vp]
    L.orbo_bench_pairs.restype = ctypes.c_double
    _L = L
    return L


def _p(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


class Extractor:
    """CPU oracle of ORBextractor (reference Features/ORBextractor.cpp:492-595, 1531-1653)."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7):
        self.L = lib()
        self.h = self.L.orbo_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
        if not self.h:
            raise ValueError("bad extractor parameters")
        self.nlevels = nlevels
        self.nfeatures = nfeatures

    def __del__(self):
        try:
            if self.h:
                self.L.orbo_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def tables(self):
        n = self.nlevels
        sc, inv, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        q = np.zeros(n, np.int32)
        um = np.zeros(16, np.int32)
        self.L.orbo_get_tables(self.h, _p(sc), _p(inv), _p(s2), _p(is2), _p(q), _p(um))
        return dict(scale=sc, inv_scale=inv, sigma2=s2, inv_sigma2=is2, quota=q, umax=um)

    def __call__(self, img: np.ndarray, lap=(0, 0), cap: int | None = None):
        img = np.ascontiguousarray(img)
        cap = cap or max(self.nfeatures + 64, 64)
        k = np.zeros(cap, KP)
        d = np.zeros((cap, 32), np.uint8)
        n = ctypes.c_int(0)
        r = self.L.orbo_extract(self.h, _p(img), img.shape[1], img.shape[0], img.strides[0], lap[0], lap[1], _p(k), _p(d), cap,
                                ctypes.byref(n))
        return r, k[:n.value].copy(), d[:n.value].copy()

    def level_size(self, level):
        w, h = ctypes.c_int(0), ctypes.c_int(0)
        assert self.L.orbo_level_size(self.h, level, ctypes.byref(w), ctypes.byref(h)) == 0
        return w.value, h.value

    def level_image(self, level):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        assert self.L.orbo_level_image(self.h, level, _p(out)) == 0
        return out

    def keep_blurred(self, on=True):
        self.L.orbo_keep_blurred(self.h, int(on))

    def level_blurred(self, level):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        r = self.L.orbo_level_blurred(self.h, level, _p(out))
        return out if r == 0 else None

    def level_candidates(self, level):
        n = self.L.orbo_level_candidates(self.h, level, None, 0)
        out = np.zeros((max(n, 1), 3), np.float32)
        self.L.orbo_level_candidates(self.h, level, _p(out), n)
        return out[:n]

    def level_selected(self, level):
        out = np.zeros(self.nfeatures + 64, KP)
        n = self.L.orbo_level_selected(self.h, level, _p(out), len(out))
        return out[:n]


def set_opencv_variant(gaussian_variant=0, gray_variant=0):
    """Process-wide: Gaussian Q8 taps (0 = error diffusion [18,34,48,56,..], 1 = rounded [18,34,49,55,..]) and BGR2GRAY
    coefficients (0 = 14-bit, 1 = 15-bit) of the oracle; mirrors the product's orbx_set_opencv_variant."""
    L = lib()
    L.orbo_set_opencv_variant.argtypes = [ctypes.c_int, ctypes.c_int]
    L.orbo_set_opencv_variant.restype = None
    L.orbo_set_opencv_variant(int(gaussian_variant), int(gray_variant))


LIBM_DEFAULT = 1  # ORBX_LIBM_FLOAT (include/orbx.h)


def set_libm_variant(libm_variant=LIBM_DEFAULT):
    """Process-wide: the libm reading of cos / sin (cpp:174) and pow (cpp:536): 0 = through double, 1 = cosf / sinf / powf;
    mirrors the product's orbx_set_libm_variant.  Extractors built afterwards take the constructor's pow from it."""
    L = lib()
    L.orbo_set_libm_variant.argtypes = [ctypes.c_int]
    L.orbo_set_libm_variant.restype = None
    L.orbo_set_libm_variant(int(libm_variant))


def libm_sweep(lo_bits: int, hi_bits: int, nthreads: int = 8):
    """Angles (degrees, f32 bit patterns lo_bits .. hi_bits) for which [0, 1] the restated glibc cosf / sinf (FMA form), [2, 3] the
    form without FMAs differ from the host's own cosf / sinf, and [4, 5] the host's cosf / sinf differ from (float)cos((double)a)."""
    L = lib()
    out = (ctypes.c_longlong * 6)()
    L.orbo_libm_sweep.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
    L.orbo_libm_sweep.restype = None
    L.orbo_libm_sweep(lo_bits, hi_bits, nthreads, out)
    return list(out)


def resize_linear(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = np.ascontiguousarray(src)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orbo_resize_linear(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    return dst


def gaussian7(src: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(src)
    dst = np.zeros_like(src)
    lib().orbo_gaussian7(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dst.strides[0])
    return dst


def fast(img: np.ndarray, t: int, nms: bool = True) -> np.ndarray:
    img = np.ascontiguousarray(img)
    n = lib().orbo_fast(_p(img), img.shape[1], img.shape[0], img.strides[0], t, int(nms), None, 0)
    out = np.zeros((max(n, 1), 3), np.float32)
    lib().orbo_fast(_p(img), img.shape[1], img.shape[0], img.strides[0], t, int(nms), _p(out), n)
    return out[:n]


def fast_strength(img: np.ndarray, x: int, y: int) -> int:
    img = np.ascontiguousarray(img)
    return lib().orbo_fast_strength(_p(img), img.shape[1], img.shape[0], img.strides[0], x, y)


def distribute(xyr: np.ndarray, min_x, max_x, min_y, max_y, n_features) -> np.ndarray:
    xyr = np.ascontiguousarray(xyr, np.float32).reshape(-1, 3)
    out = np.zeros((len(xyr) + 8, 3), np.float32)
    n = lib().orbo_distribute(_p(xyr), len(xyr), min_x, max_x, min_y, max_y, n_features, _p(out), len(out))
    return out[:n]


def fast_atan2(y: float, x: float) -> float:
    return float(lib().orbo_fast_atan2(y, x))


def ic_angle(img: np.ndarray, x: float, y: float):
    img = np.ascontiguousarray(img)
    m10, m01 = ctypes.c_int(0), ctypes.c_int(0)
    a = lib().orbo_ic_angle(_p(img), img.shape[1], img.shape[0], x, y, ctypes.byref(m10), ctypes.byref(m01))
    return float(a), m10.value, m01.value


def descriptor(blurred: np.ndarray, x: float, y: float, angle: float) -> np.ndarray:
    blurred = np.ascontiguousarray(blurred)
    d = np.zeros(32, np.uint8)
    lib().orbo_descriptor(_p(blurred), blurred.shape[1], blurred.shape[0], x, y, angle, _p(d))
    return d


def hamming(a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orbo_hamming(_p(a), _p(b))


def pos_in_grid(kps: np.ndarray, bounds):
    kps = np.ascontiguousarray(kps, KP)
    b = np.array(bounds, np.int32)
    n = len(kps)
    px, py, ok = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    lib().orbo_pos_in_grid(_p(kps), n, _p(b), _p(px), _p(py), _p(ok))
    return px, py, ok.astype(bool)


def features_in_area(kps: np.ndarray, bounds, x, y, r, min_level=-1, max_level=-1) -> np.ndarray:
    kps = np.ascontiguousarray(kps, KP)
    b = np.array(bounds, np.int32)
    out = np.zeros(len(kps) + 1, np.int32)
    n = lib().orbo_features_in_area(_p(kps), len(kps), _p(b), x, y, r, min_level, max_level, _p(out), len(out))
    return out[:n]


def match_init(k1, d1, k2, d2, bounds, window=100, nnratio=0.9, check_ori=True):
    """Oracle of ORBmatcher::SearchForInitialization.  Returns (nmatches, matches12, stats[3])."""
    k1 = np.ascontiguousarray(k1, KP)
    k2 = np.ascontiguousarray(k2, KP)
    d1 = np.ascontiguousarray(d1, np.uint8)
    d2 = np.ascontiguousarray(d2, np.uint8)
    b = np.array(bounds, np.int32)
    m = np.full(max(len(k1), 1), -1, np.int32)
    st = np.zeros(3, np.int32)
    nm = lib().orbo_match_init(_p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2), _p(b), window, nnratio, int(check_ori), _p(m),
                               _p(st))
    return nm, m[:len(k1)].copy(), st


def bench_pairs(params, imgs: np.ndarray, window=100, nnratio=0.9, nthreads=1, reps=1):
    """CPU baseline: extract(A) + extract(B) + SearchForInitialization per pair.  Returns (seconds, frames, checksum)."""
    imgs = np.ascontiguousarray(imgs, np.uint8)
    n, h, w = imgs.shape
    frames = ctypes.c_long(0)
    chk = ctypes.c_int(0)
    sec = lib().orbo_bench_pairs(params[0], params[1], params[2], params[3], params[4], _p(imgs), n, w, h, window, nnratio,
                                 nthreads, reps, ctypes.byref(frames), ctypes.byref(chk))
    return float(sec), int(frames.value), int(chk.value)


def bench_protocol(params, imgs: np.ndarray, window=100, nnratio=0.9, nthreads=1, warmups=5, reps=30, native=False):
    """CPU baseline by the protocol of SURVEY.md 8(d): pinned worker threads, each timing `reps` repetitions of extract(A) +
    extract(B) + SearchForInitialization on its own frame pair after `warmups` untimed ones.  Returns an array
    [nthreads, reps, 3] of seconds (whole repetition, extraction alone, matching alone).  native=True uses the
    -march=native build of the same restatement."""
    imgs = np.ascontiguousarray(imgs, np.uint8)
    n, h, w = imgs.shape
    L = lib()
    if native:
        # -march=native means THIS host's instruction set: always compiled where it runs (the in-tree .so files travel from the
        # build container to the GPU box, whose host CPU may differ), into a scratch directory
        import tempfile
        path = os.path.join(tempfile.mkdtemp(prefix="orbx_oracle_native_"), "liborbx_oracle_native.so")
        subprocess.run(["g++", "-O3", "-DNDEBUG", "-std=c++17", "-ffp-contract=off", "-fPIC", "-pthread", "-march=native", "-shared",
                        "-o", path, os.path.join(ORACLE_DIR, "orbx_oracle.cpp")], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        L = ctypes.CDLL(path)
    L.orbo_bench_protocol.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p]
    out = np.zeros((nthreads, reps, 3), np.float64)
    r = L.orbo_bench_protocol(params[0], params[1], params[2], params[3], params[4], _p(imgs), n, w, h, window, nnratio, nthreads,
                              warmups, reps, _p(out))
    assert r == 0
    return out


def std_sort_sized(triples: np.ndarray) -> np.ndarray:
    """libstdc++ std::sort with the reference's compareNodes on (count, UL.x, id) triples (cpp:684-696, 912)."""
    t = np.ascontiguousarray(triples, np.int32).reshape(-1, 3).copy()
    L = lib()
    L.orbo_std_sort_sized.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.orbo_std_sort_sized.restype = None
    L.orbo_std_sort_sized(_p(t), len(t))
    return t


SETTINGS_CAMERA = (609.2855, 609.3422, 351.4274, 237.7324, -0.3492, 0.1363, 0.0, 0.0)  # reference Settings.yaml


def _cam(camera) -> np.ndarray:
    return np.ascontiguousarray(camera, np.float32).reshape(8)


def undistort_keypoints(kps: np.ndarray, camera) -> np.ndarray:
    """Oracle of Frame::UndistortKeyPoints (Frame.cpp:136-161); camera = (fx, fy, cx, cy, k1, k2, p1, p2)."""
    kps = np.ascontiguousarray(kps, KP)
    out = np.zeros(len(kps), KP)
    L = lib()
    L.orbo_undistort_keypoints.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.orbo_undistort_keypoints.restype = None
    L.orbo_undistort_keypoints(_p(kps), len(kps), _p(_cam(camera)), _p(out))
    return out


def image_bounds(camera, cols: int, rows: int):
    """Oracle of Frame::ComputeImageBounds (Frame.cpp:101-134) -> (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    b = np.zeros(4, np.int32)
    L = lib()
    L.orbo_image_bounds.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.orbo_image_bounds.restype = None
    L.orbo_image_bounds(_p(_cam(camera)), cols, rows, _p(b))
    return tuple(int(v) for v in b)


def to_gray(image: np.ndarray, rgb: bool = False):
    """Oracle of Converter::toGray (Utils/Converter.cpp:5-19).  Returns the gray image, or None for `return false`."""
    im = np.ascontiguousarray(image, np.uint8)
    ch = 1 if im.ndim == 2 else im.shape[2]
    h, w = im.shape[:2]
    out = np.zeros((h, w), np.uint8)
    L = lib()
    L.orbo_to_gray.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_int]
    ok = L.orbo_to_gray(_p(im), w, h, w * ch, ch, int(rgb), _p(out), w)
    return out if ok else None


def _pairs(matches12):
    m = np.ascontiguousarray(matches12, np.int32)
    first = np.nonzero(m >= 0)[0].astype(np.int32)  # mvMatches12, Initializer.cpp:24-33
    return first, m[first].copy()


def check_homography(H21, H12, k1, k2, matches12, sigma=1.0):
    """Oracle of Initializer::CheckHomography (Initializer.cpp:268-352) -> (score, inliers)."""
    k1, k2 = np.ascontiguousarray(k1, KP), np.ascontiguousarray(k2, KP)
    first, second = _pairs(matches12)
    H21 = np.ascontiguousarray(H21, np.float32).reshape(9)
    H12 = np.ascontiguousarray(H12, np.float32).reshape(9)
    inl = np.zeros(max(len(first), 1), np.uint8)
    L = lib()
    L.orbo_check_homography.restype = ctypes.c_float
    L.orbo_check_homography.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
    sc = L.orbo_check_homography(_p(H21), _p(H12), _p(k1), _p(k2), _p(first), _p(second), len(first), sigma, _p(inl))
    return np.float32(sc), inl[:len(first)].astype(bool)


def check_fundamental(F21, k1, k2, matches12, sigma=1.0):
    """Oracle of Initializer::CheckFundamental (Initializer.cpp:355-438) -> (score, inliers)."""
    k1, k2 = np.ascontiguousarray(k1, KP), np.ascontiguousarray(k2, KP)
    first, second = _pairs(matches12)
    F21 = np.ascontiguousarray(F21, np.float32).reshape(9)
    inl = np.zeros(max(len(first), 1), np.uint8)
    L = lib()
    L.orbo_check_fundamental.restype = ctypes.c_float
    L.orbo_check_fundamental.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
    sc = L.orbo_check_fundamental(_p(F21), _p(k1), _p(k2), _p(first), _p(second), len(first), sigma, _p(inl))
    return np.float32(sc), inl[:len(first)].astype(bool)


def check_rt(R21, t21, K, k1, k2, matches12, inliers, th2=4.0):
    """Oracle of Initializer::CheckRT (Initializer.cpp:569-713) -> (nGood, vbTriGood[n1], vP3D[n1, 3], parallax)."""
    k1, k2 = np.ascontiguousarray(k1, KP), np.ascontiguousarray(k2, KP)
    first, second = _pairs(matches12)
    R21 = np.ascontiguousarray(R21, np.float32).reshape(9)
    t21 = np.ascontiguousarray(t21, np.float32).reshape(3)
    K = np.ascontiguousarray(K, np.float32).reshape(9)
    inl = np.ascontiguousarray(inliers, np.uint8)
    assert len(inl) == len(first)
    good = np.zeros(max(len(k1), 1), np.uint8)
    p3d = np.zeros((max(len(k1), 1), 3), np.float32)
    par = ctypes.c_float(0)
    L = lib()
    L.orbo_check_rt.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_float,
                                                                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    n = L.orbo_check_rt(_p(R21), _p(t21), _p(K), _p(k1), len(k1), _p(k2), _p(first), _p(second), len(first), _p(inl), th2, _p(good), _p(p3d),
                        ctypes.byref(par))
    return int(n), good[:len(k1)].astype(bool), p3d[:len(k1)], np.float32(par.value)


def two_view_case(seed=0, n=500, outliers=0.2, noise=0.5):
    """Synthetic two-view geometry for CheckRT: 3-D points in front of camera 1, camera 2 = (R, t), pixel noise, a share of
    wrong matches, matches12 with holes.  Returns K, R, t (float64 truth), k1, k2, matches12 and the four (R, t) candidates
    an essential-matrix decomposition yields (the true one first)."""
    rng = np.random.default_rng(seed)
    K = np.array([[609.2855, 0, 351.4274], [0, 609.3422, 237.7324], [0, 0, 1.0]])
    ang = np.deg2rad(rng.uniform(2, 8, 3)) * rng.choice([-1, 1], 3)
    cx, sx, cy, sy, cz, sz = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
    R = (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
         @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
    t = rng.normal(0, 1, 3); t /= np.linalg.norm(t)
    X = np.stack([rng.uniform(-4, 4, n), rng.uniform(-3, 3, n), rng.uniform(4, 20, n)], 1)
    p1 = (K @ X.T).T; p1 = p1[:, :2] / p1[:, 2:]
    X2 = (R @ X.T).T + t
    p2 = (K @ X2.T).T; p2 = p2[:, :2] / p2[:, 2:]
    k1, k2 = np.zeros(n, KP), np.zeros(n + 20, KP)
    k1["x"], k1["y"] = (p1[:, 0] + rng.normal(0, noise, n)).astype(np.float32), (p1[:, 1] + rng.normal(0, noise, n)).astype(np.float32)
    perm = rng.permutation(n + 20)
    k2["x"][perm[:n]] = (p2[:, 0] + rng.normal(0, noise, n)).astype(np.float32)
    k2["y"][perm[:n]] = (p2[:, 1] + rng.normal(0, noise, n)).astype(np.float32)
    k2["x"][perm[n:]], k2["y"][perm[n:]] = rng.uniform(0, 640, 20).astype(np.float32), rng.uniform(0, 480, 20).astype(np.float32)
    m12 = perm[:n].astype(np.int32)
    wrong = rng.random(n) < outliers
    m12[wrong] = rng.integers(0, n + 20, wrong.sum())
    m12[rng.random(n) < 0.1] = -1  # unmatched keypoints
    # the twisted-pair candidates of decomposeEssentialMat: (R, t), (R, -t), (R', t), (R', -t), R' = R rotated by pi about t
    tx = t / np.linalg.norm(t)
    Rpi = 2 * np.outer(tx, tx) - np.eye(3)
    cands = [(R, t), (R, -t), (Rpi @ R, t), (Rpi @ R, -t)]
    return K, R, t, k1, k2, m12, cands


def scoring_case(seed=0, n=400, n_models=24):
    """Synthetic two-view case for the scoring loops: keypoints of frame 1, their images under a homography plus noise and
    outliers, a matches12 vector with holes, and stacks of perturbed H / F hypotheses (float32)."""
    rng = np.random.default_rng(seed)
    k1, k2 = np.zeros(n, KP), np.zeros(n + 30, KP)
    k1["x"], k1["y"] = rng.uniform(20, 620, n).astype(np.float32), rng.uniform(20, 460, n).astype(np.float32)
    H = np.array([[1.02, 0.03, 5.0], [-0.02, 0.99, -3.0], [1e-5, -2e-5, 1.0]])
    p = H @ np.stack([k1["x"].astype(np.float64), k1["y"].astype(np.float64), np.ones(n)])
    perm = rng.permutation(n + 30)
    xy = (p[:2] / p[2]).T + rng.normal(0, 0.7, (n, 2))
    out = rng.random(n) < 0.2
    xy[out] += rng.uniform(-40, 40, (int(out.sum()), 2))
    k2["x"][perm[:n]], k2["y"][perm[:n]] = xy[:, 0].astype(np.float32), xy[:, 1].astype(np.float32)
    m12 = perm[:n].astype(np.int32)
    m12[rng.random(n) < 0.3] = -1
    H21 = np.stack([H + rng.normal(0, 1, (3, 3)) * np.array([[2e-3, 2e-3, 0.5], [2e-3, 2e-3, 0.5], [1e-6, 1e-6, 0]]) * (i > 0)
                    for i in range(n_models)])
    H12 = np.stack([np.linalg.inv(h) for h in H21])
    # fundamental matrices of a translating + rotating camera, perturbed
    K = np.array([[520.0, 0, 320], [0, 520, 240], [0, 0, 1]])
    Ki = np.linalg.inv(K)
    F21 = []
    for i in range(n_models):
        t = np.array([1.0, 0.1, 0.05]) + rng.normal(0, 0.05, 3) * (i > 0)
        a = rng.normal(0, 0.01, 3) * (i > 0)
        R = np.eye(3) + np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        F21.append(Ki.T @ tx @ R @ Ki)
    return k1, k2, m12, H21.astype(np.float32), H12.astype(np.float32), np.stack(F21).astype(np.float32)


def sincos_deg_batch(angles: np.ndarray):
    """cos / sin of computeOrbDescriptor (cpp:173-174) for many angles: (float)cos((double)(angle * factorPI))."""
    a = np.ascontiguousarray(angles, np.float32)
    c, sn = np.zeros(len(a), np.float32), np.zeros(len(a), np.float32)
    L = lib()
    L.orbo_sincos_deg_batch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.orbo_sincos_deg_batch.restype = None
    L.orbo_sincos_deg_batch(_p(a), len(a), _p(c), _p(sn))
    return c, sn
