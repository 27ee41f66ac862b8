"""orb_slam_tracking_amd — MI355X-native ORB extraction + initialization matching.

Thin ctypes mirror of the C ABI in ``include/orbx.h`` (``liborbx.so``: hand-written HIP kernels for gfx950 plus the
C++ host pipeline).  The classes keep the names and argument meaning of the reference's C++ interface for this path:

* ``ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)`` / ``__call__``  -- Features/ORBextractor.hpp:68-85
* ``ORBmatcher(nnratio=0.6, checkOri=True).SearchForInitialization(F1, F2, windowSize=100)`` -- Features/ORBmatcher.hpp:15,36
* ``Frame`` -- the part of SlamTypes/Frame.{hpp,cpp} the matcher reads (mvKeys, mvKeysUn, mDescriptors, N, image bounds)

There is NO CPU fallback: importing works anywhere, but every compute call raises ``OrbxError`` unless liborbx.so is
built and a HIP device is usable.  (The C++ drop-in classes live in include/orbx_shim.hpp.)
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import numpy as np

__all__ = ["KEYPOINT_DTYPE", "OrbxError", "ORBextractor", "ORBmatcher", "Frame", "lib", "lib_path",
           "STAGES", "E_EMPTY", "E_BADARG", "E_TOOSMALL", "E_HIP", "E_CAPACITY", "E_RCCL"]

# mirrors cv::KeyPoint / orbx_keypoint (28 bytes)
KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4"), ("class_id", "<i4")])
assert KEYPOINT_DTYPE.itemsize == 28

E_EMPTY, E_BADARG, E_TOOSMALL, E_HIP, E_CAPACITY, E_RCCL = -1, -2, -3, -4, -5, -6
_ERRNAMES = {-1: "ORBX_E_EMPTY", -2: "ORBX_E_BADARG", -3: "ORBX_E_TOOSMALL", -4: "ORBX_E_HIP", -5: "ORBX_E_CAPACITY", -6: "ORBX_E_RCCL"}
STAGES = ("pyramid", "fast", "select", "describe", "match")


class OrbxError(RuntimeError):
    def __init__(self, code: int, what: str = ""):
        self.code = code
        super().__init__("%s (%d)%s" % (_ERRNAMES.get(code, "ORBX_E_?"), code, (": " + what) if what else ""))


class _Params(ctypes.Structure):
    _fields_ = [("nfeatures", ctypes.c_int32), ("scale_factor", ctypes.c_float), ("nlevels", ctypes.c_int32),
                ("ini_th_fast", ctypes.c_int32), ("min_th_fast", ctypes.c_int32)]


class _Bounds(ctypes.Structure):
    _fields_ = [("min_x", ctypes.c_int32), ("max_x", ctypes.c_int32), ("min_y", ctypes.c_int32), ("max_y", ctypes.c_int32)]


class _Camera(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2")]


class _Stats(ctypes.Structure):
    _fields_ = [("invalid_by_distance", ctypes.c_int32), ("invalid_by_ratio", ctypes.c_int32),
                ("invalid_by_orientation", ctypes.c_int32)]


def lib_path() -> str:
    # ORBX_LIB (diagnostics): an instrumented build of the same sources, e.g. `make -C csrc VARIANT=octstamps EXTRA=-DORBX_OCT_STAMPS`
    return os.environ.get("ORBX_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "liborbx.so")


_LIB = None


def _one_hip_runtime() -> None:
    """One HIP runtime per process.  liborbx.so needs `libamdhip64.so.7`; PyTorch-ROCm ships its own copy of that library and loads
    it by path.  If torch is imported first, liborbx resolves to torch's copy (same SONAME) and both share devices, streams and
    memory.  The other order used to give the process two runtimes -- the second one finds no device (`orbx_create` ->
    ORBX_E_HIP).  So when torch is installed but not yet imported, its copy is loaded here, before liborbx, without importing torch."""
    import sys
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(p):
            ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)
    except Exception:  # no torch, or an unusual layout: liborbx takes the system's runtime
        pass


def hip_runtimes_mapped() -> list:
    """The distinct libamdhip64 files mapped into this process (Linux).  More than one means two HIP runtimes: the pre-load above
    only helps when torch's copy carries the SONAME liborbx.so was linked against (libamdhip64.so.7) -- a torch wheel built against
    another ROCm major leaves the process with both, and the second one finds no device."""
    try:
        return sorted(set(ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln))
    except OSError:
        return []


def lib() -> ctypes.CDLL:
    """Loads liborbx.so (built in-tree by ``__graft_entry__.build()`` / ``make -C orb_slam_tracking_amd/csrc``)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise OrbxError(E_HIP, "liborbx.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
    _one_hip_runtime()
    L = ctypes.CDLL(path)
    vp, i32, f32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    L.orbx_create.argtypes = [ctypes.POINTER(_Params), i32, i32, i32, i32, vp, ctypes.POINTER(vp)]
    L.orbx_destroy.argtypes = [vp]
    L.orbx_destroy.restype = None
    L.orbx_last_error.argtypes = [vp]
    L.orbx_last_error.restype = ctypes.c_char_p
    L.orbx_get_levels.argtypes = [vp]
    L.orbx_get_scale_factor.argtypes = [vp]
    L.orbx_get_scale_factor.restype = f32
    L.orbx_get_tables.argtypes = [vp, vp, vp, vp, vp, vp]
    L.orbx_get_umax.argtypes = [vp, vp]
    L.orbx_extract.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp]
    L.orbx_host_register.argtypes = [vp, vp, sz]
    L.orbx_host_unregister.argtypes = [vp, vp]
    L.orbx_extract_batch.argtypes = [vp, i32, vp, i32, i32, i32, sz, i32, i32, vp, vp, i32, vp, vp]
    L.orbx_extract_batch_device.argtypes = [vp, i32, vp, i32, i32, i32, sz, vp, vp, i32, vp]
    L.orbx_level_size.argtypes = [vp, i32, vp, vp]
    L.orbx_download_pyramid.argtypes = [vp, i32, i32, i32, vp, i32]
    L.orbx_match_init.argtypes = [vp, vp, vp, i32, vp, vp, i32, ctypes.POINTER(_Bounds), i32, f32, i32, vp,
                                  ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(_Stats)]
    L.orbx_match_init_batch_device.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, ctypes.POINTER(_Bounds), i32, f32, i32,
                                               vp, vp, vp]
    L.orbx_extract_match_batch_device.argtypes = [vp, i32, vp, i32, i32, i32, sz, vp, vp, i32, vp, i32, vp, vp,
                                                  ctypes.POINTER(_Bounds), i32, f32, i32, vp, vp, vp]
    L.orbx_extract_match_batch_device_async.argtypes = [vp, i32, vp, i32, i32, i32, sz, vp, vp, i32, vp, i32, vp, vp,
                                                  ctypes.POINTER(_Bounds), i32, f32, i32, vp, vp, vp]
    L.orbx_extract_match_batch_host_async.argtypes = L.orbx_extract_match_batch_device_async.argtypes
    L.orbx_set_opencv_variant.argtypes = [vp, i32, i32]
    L.orbx_set_libm_variant.argtypes = [vp, i32]
    L.orbx_set_pipeline_depth.argtypes = [vp, i32]
    L.orbx_wait_one.argtypes = [vp]
    L.orbx_wait.argtypes = [vp]
    L.orbx_order_after.argtypes = [vp, vp]
    L.orbx_order_before.argtypes = [vp, vp]
    L.orbx_profile_stages.argtypes = [vp, ctypes.c_uint]
    L.orbx_undistort_keypoints.argtypes = [vp, vp, i32, ctypes.POINTER(_Camera), vp]
    L.orbx_undistort_batch_device.argtypes = [vp, i32, vp, vp, i32, ctypes.POINTER(_Camera), vp]
    L.orbx_image_bounds.argtypes = [vp, ctypes.POINTER(_Camera), i32, i32, ctypes.POINTER(_Bounds)]
    L.orbx_to_gray.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, i32]
    L.orbx_to_gray_batch_device.argtypes = [vp, i32, vp, i32, i32, i32, sz, i32, i32, vp, i32, sz]
    L.orbx_check_homography.argtypes = [vp, i32, vp, vp, vp, i32, vp, i32, vp, f32, vp, vp, vp, vp]
    L.orbx_check_fundamental.argtypes = [vp, i32, vp, vp, i32, vp, i32, vp, f32, vp, vp, vp, vp]
    L.orbx_check_rt.argtypes = [vp, i32, vp, vp, vp, vp, i32, vp, i32, vp, vp, f32, vp, vp, vp, vp]
    L.orbx_multi_create.argtypes = [ctypes.POINTER(_Params), i32, vp, i32, i32, i32, ctypes.POINTER(vp)]
    L.orbx_multi_destroy.argtypes = [vp]
    L.orbx_multi_destroy.restype = None
    L.orbx_multi_size.argtypes = [vp]
    L.orbx_multi_ctx.argtypes = [vp, i32]
    L.orbx_multi_ctx.restype = vp
    L.orbx_multi_last_error.argtypes = [vp]
    L.orbx_multi_last_error.restype = ctypes.c_char_p
    L.orbx_multi_shard_range.argtypes = [i32, i32, i32, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.orbx_multi_extract_match_batch_device.argtypes = [vp, i32, vp, i32, i32, i32, sz, vp, vp, i32, vp, ctypes.POINTER(_Bounds), i32,
                                                        f32, i32, vp, vp, vp]
    L.orbx_multi_extract_match_batch_device_async.argtypes = L.orbx_multi_extract_match_batch_device.argtypes
    L.orbx_multi_set_pipeline_depth.argtypes = [vp, i32]
    L.orbx_multi_wait_one.argtypes = [vp]
    L.orbx_multi_wait.argtypes = [vp]
    L.orbx_profile_enable.argtypes = [vp, i32]
    L.orbx_profile_reset.argtypes = [vp]
    L.orbx_profile_get.argtypes = [vp, vp, vp]
    L.orbx_debug_candidates.argtypes = [vp, i32, i32, vp, i32]
    L.orbx_debug_distribute_device.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32]
    L.orbx_debug_std_sort.argtypes = [vp, vp, i32]
    L.orbx_debug_sincos.argtypes = [vp, vp, i32, vp, vp]
    L.orbx_debug_set.argtypes = [ctypes.c_char_p, ctypes.c_longlong]
    L.orbx_debug_last_launch.argtypes = [vp, vp]
    L.orbx_debug_match_counters.argtypes = [vp, vp]
    L.orbx_debug_selection_units.argtypes = [vp, i32, vp, vp]
    L.orbx_debug_path_codes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
    _LIB = L
    # The library reads no environment variable; the profiling scripts' ORBX_<KNOB> settings are forwarded to orbx_debug_set here
    # (diagnostic knobs, csrc/orbx_knobs.h: they choose among kernels / launch shapes with identical results).
    for name in KNOBS:
        v = os.environ.get("ORBX_" + name.upper())
        if v is not None:
            debug_set(name, _knob_value(name, v))
    return L


KNOBS = ("no_bands", "no_tiles", "tiles_max_frames", "tiles_max_pixels", "pyr_bands", "pyr_gmax", "pyr_strips", "bands_min_frames", "desc_no_staged",
         "desc_staged_max", "no_split", "lat_trace", "no_direct_out", "fast_wg", "fast_wg_max_cells", "fast_lds_pad", "fast_debug",
         "desc_lds_pad", "match_no_general", "match_no_mfma", "oct_no_small", "oct_key64", "oct_split_min", "oct_no_big", "oct_big_depth", "oct_big_no_fallback", "octb_no_512", "oct_inst",
         "oct_lds_pad", "multi_force_rccl")
KNOB_UNSET = -(1 << 63)


def _knob_value(name: str, text: str) -> int:
    if name == "oct_inst":  # "2048,2048,1024,512,..." -> one hex digit per level, lowest digit = level 0
        v = 0
        for l, t in enumerate(text.split(",")[:16]):
            v |= {"512": 1, "1024": 2, "2048": 3}.get(t.strip(), 0) << (4 * l)
        return v
    try:
        return int(text)
    except ValueError:
        return 1  # (a flag set to anything: on)


def debug_set(name: str, value: Optional[int]) -> None:
    """Diagnostic knob of the library (orbx_debug_set, include/orbx.h): `None` unsets it."""
    r = lib().orbx_debug_set(name.encode(), KNOB_UNSET if value is None else int(value))
    if r != 0:
        raise OrbxError(r, "orbx_debug_set(%r)" % name)


def _ptr(a) -> ctypes.c_void_p:
    if a is None:
        return ctypes.c_void_p(0)
    if isinstance(a, np.ndarray):
        return ctypes.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):  # torch tensor
        return ctypes.c_void_p(a.data_ptr())
    return ctypes.c_void_p(int(a))


def _nbytes(a):
    """Size in bytes of a numpy array / torch tensor, None for a raw pointer (an int: its size is the caller's business)."""
    if isinstance(a, np.ndarray):
        return int(a.nbytes)
    if hasattr(a, "data_ptr") and hasattr(a, "numel"):
        return int(a.numel()) * int(a.element_size())
    return None


def _need(what: str, a, nbytes: int) -> None:
    """A batch call reads / writes `nbytes` of `a`: an array that is shorter is a ValueError HERE -- behind the C ABI it would be a
    GPU memory fault (round 6: bench.py handed 64 frames' worth of arguments over 16-frame tensors)."""
    have = _nbytes(a)
    if have is not None and have < nbytes:
        raise ValueError("%s holds %d bytes, the call needs %d" % (what, have, nbytes))


def _need_batch(imgs, n_frames, width, height, stride, frame_stride, kps, desc, n, capacity, first=None, second=None, matches12=None,
                nmatches=None, stats=None) -> None:
    if n_frames > 0 and imgs is not None:
        _need("the frames", imgs, (n_frames - 1) * frame_stride + (height - 1) * stride + width)
    _need("the keypoint array", kps, n_frames * capacity * 28)
    _need("the descriptor array", desc, n_frames * capacity * 32)
    _need("the count array", n, n_frames * 4)
    if first is not None and len(first):
        if len(second) != len(first):
            raise ValueError("first / second: %d and %d pairs" % (len(first), len(second)))
        # (one pass over both: as unsigned, a negative index is a huge one)
        if n_frames > 0 and int(np.maximum(first.view(np.uint32), second.view(np.uint32)).max()) >= n_frames:
            raise ValueError("a pair names a frame outside the batch of %d" % n_frames)
        _need("matches12", matches12, len(first) * capacity * 4)
        _need("nmatches", nmatches, len(first) * 4)
        if stats is not None:
            _need("stats", stats, len(first) * 12)


def _torch_stream(*arrays):
    """The current torch stream (as an int handle) if any of the arguments is a torch CUDA tensor, else None: work
    issued on the context's own HIP streams must start after what torch has queued for those tensors."""
    for a in arrays:
        if a is not None and hasattr(a, "data_ptr") and getattr(a, "is_cuda", False):
            import torch
            st = torch.cuda.current_stream(a.device)
            if st.query():  # nothing queued on it: nothing to order behind (saves the event record + two stream waits)
                return None
            return int(st.cuda_stream)
    return None


class ORBextractor:
    """Features/ORBextractor.hpp:55-158.  One instance == one orbx_ctx bound to one device; calls are serialised by
    the caller (the reference's operator() is not re-entrant either, cpp:1669)."""

    HARRIS_SCORE, FAST_SCORE = 0, 1

    def __init__(self, nfeatures: int, scaleFactor: float, nlevels: int, iniThFAST: int, minThFAST: int, *,
                 max_width: int = 1920, max_height: int = 1080, max_batch: int = 1, device: int = 0,
                 stream: Optional[int] = None):
        self._L = lib()
        self._h = ctypes.c_void_p(0)
        p = _Params(int(nfeatures), float(scaleFactor), int(nlevels), int(iniThFAST), int(minThFAST))
        r = self._L.orbx_create(ctypes.byref(p), int(device), int(max_width), int(max_height), int(max_batch),
                                ctypes.c_void_p(stream or 0), ctypes.byref(self._h))
        if r != 0:
            self._h = ctypes.c_void_p(0)
            what = "orbx_create"
            if r == E_HIP:
                what = "orbx_create failed (no usable HIP device / kernel image?)"
                rts = hip_runtimes_mapped()
                if len(rts) > 1:  # the diagnosis that used to take a debugger (gpurun_out/r05/create2.log)
                    what += ("; TWO HIP runtimes are mapped into this process (%s): liborbx.so and PyTorch each brought their own "
                             "libamdhip64 -- import torch before orb_slam_tracking_amd, or build liborbx.so against the ROCm that "
                             "torch ships" % ", ".join(rts))
            raise OrbxError(r, what)
        self.nfeatures, self.nlevels, self.device, self.max_batch = int(nfeatures), int(nlevels), int(device), int(max_batch)
        q = np.zeros(self.nlevels, np.int32)
        self._L.orbx_get_tables(self._h, None, None, None, None, _ptr(q))
        self.capacity = max(int(q.sum()), 1)
        self.order_with_torch = True  # device-resident calls given torch tensors are ordered behind torch's current stream

    # -- lifetime ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.orbx_destroy(self._h)
            self._h = ctypes.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    GAUSS_ERROR_DIFFUSION, GAUSS_ROUNDED, GRAY_14BIT, GRAY_15BIT = 0, 1, 0, 1

    def set_opencv_variant(self, gaussian_variant: int = 0, gray_variant: int = 0) -> None:
        """The two OpenCV-release dependent constants of the path (include/orbx.h): Gaussian Q8 taps, BGR2GRAY coefficients."""
        self._check(self._L.orbx_set_opencv_variant(self._h, int(gaussian_variant), int(gray_variant)), "orbx_set_opencv_variant")

    LIBM_DOUBLE, LIBM_FLOAT, LIBM_DEFAULT = 0, 1, 1

    def set_libm_variant(self, libm_variant: int = 1) -> None:
        """The libm reading of the reference's unqualified cos / sin / pow on floats (include/orbx.h): 0 = through double,
        1 = cosf / sinf / powf (glibc >= 2.28's algorithm; the default)."""
        self._check(self._L.orbx_set_libm_variant(self._h, int(libm_variant)), "orbx_set_libm_variant")

    def order_after(self, stream: Optional[int]) -> None:
        """Work issued on this context from now on starts after everything queued on `stream` (hipStream_t handle)."""
        self._check(self._L.orbx_order_after(self._h, ctypes.c_void_p(stream or 0)), "orbx_order_after")

    def order_before(self, stream: Optional[int]) -> None:
        """Work queued on `stream` from now on starts after everything issued on this context so far."""
        self._check(self._L.orbx_order_before(self._h, ctypes.c_void_p(stream or 0)), "orbx_order_before")

    def _order_torch(self, *arrays) -> None:
        # torch tensors in: the producer of the frames (and the last reader of the output arrays) runs on torch's current
        # stream, which the context's private streams know nothing about
        if self.order_with_torch:
            s = _torch_stream(*arrays)
            if s is not None:
                self.order_after(s)

    def _check(self, r: int, what: str = "") -> int:
        if r < 0:
            raise OrbxError(r, (self._L.orbx_last_error(self._h) or b"").decode() or what)
        return r

    # -- getters (hpp:87-108) --------------------------------------------------------------------
    def GetLevels(self) -> int:
        return self._L.orbx_get_levels(self._h)

    def GetScaleFactor(self) -> float:
        return float(self._L.orbx_get_scale_factor(self._h))

    def _table(self, idx: int, dtype=np.float32) -> np.ndarray:
        out = np.zeros(self.nlevels, dtype)
        args = [None] * 5
        args[idx] = _ptr(out)
        self._L.orbx_get_tables(self._h, *args)
        return out

    def GetScaleFactors(self):
        return self._table(0)

    def GetInverseScaleFactors(self):
        return self._table(1)

    def GetScaleSigmaSquares(self):
        return self._table(2)

    def GetInverseScaleSigmaSquares(self):
        return self._table(3)

    def GetNumFeaturesPerLevel(self):
        return self._table(4, np.int32)

    def umax(self) -> np.ndarray:
        out = np.zeros(16, np.int32)
        self._L.orbx_get_umax(self._h, _ptr(out))
        return out

    # -- operator() (cpp:1531-1653) ----------------------------------------------------------------
    def __call__(self, image: np.ndarray, mask=None, vLappingArea: Sequence[int] = (0, 0)) -> Tuple[int, np.ndarray, np.ndarray]:
        """Returns (monoIndex, keypoints[KEYPOINT_DTYPE], descriptors[N,32] uint8); -1 for an empty image (cpp:1536)."""
        if image is None or image.size == 0:
            return -1, np.zeros(0, KEYPOINT_DTYPE), np.zeros((0, 32), np.uint8)
        if image.dtype != np.uint8 or image.ndim != 2:
            raise OrbxError(E_BADARG, "image must be a 2-D uint8 array (CV_8UC1, cpp:1541)")
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        kps = np.empty(self.capacity, KEYPOINT_DTYPE)   # (the first n entries are written, only those are returned)
        desc = np.empty((self.capacity, 32), np.uint8)
        n = ctypes.c_int(0)
        r = self._L.orbx_extract(self._h, _ptr(image), w, h, image.strides[0], int(vLappingArea[0]), int(vLappingArea[1]),
                                 _ptr(kps), _ptr(desc), self.capacity, ctypes.byref(n))
        self._check(r, "orbx_extract")
        return r, kps[:n.value], desc[:n.value]  # (views of this call's own arrays)

    def extract_batch(self, images: np.ndarray, vLappingArea: Sequence[int] = (0, 0)):
        """images: [B, H, W] uint8 (host).  Returns a list of (monoIndex, keypoints, descriptors) per frame."""
        images = np.ascontiguousarray(images)
        if images.dtype != np.uint8 or images.ndim != 3:
            raise OrbxError(E_BADARG, "images must be [B,H,W] uint8")
        B, h, w = images.shape
        kps = np.zeros((B, self.capacity), KEYPOINT_DTYPE)
        desc = np.zeros((B, self.capacity, 32), np.uint8)
        n = np.zeros(B, np.int32)
        mono = np.zeros(B, np.int32)
        r = self._L.orbx_extract_batch(self._h, B, _ptr(images), w, h, w, w * h, int(vLappingArea[0]), int(vLappingArea[1]),
                                       _ptr(kps), _ptr(desc), self.capacity, _ptr(n), _ptr(mono))
        self._check(r, "orbx_extract_batch")
        return [(int(mono[f]), kps[f, :n[f]].copy(), desc[f, :n[f]].copy()) for f in range(B)]

    def extract_batch_device(self, d_imgs, n_frames: int, width: int, height: int, stride: int, frame_stride: int,
                             d_kps, d_desc, d_n, capacity: Optional[int] = None) -> None:
        """Frames resident in HBM in, keypoints/descriptors/counts resident in HBM out (device pointers or torch tensors)."""
        _need_batch(d_imgs, int(n_frames), int(width), int(height), int(stride), int(frame_stride), d_kps, d_desc, d_n,
                    int(capacity or self.capacity))
        self._order_torch(d_imgs, d_kps, d_desc, d_n)
        r = self._L.orbx_extract_batch_device(self._h, int(n_frames), _ptr(d_imgs), int(width), int(height), int(stride),
                                              int(frame_stride), _ptr(d_kps), _ptr(d_desc), int(capacity or self.capacity),
                                              _ptr(d_n))
        self._check(r, "orbx_extract_batch_device")

    def match_pairs_device(self, first: np.ndarray, second: np.ndarray, d_kps, d_desc, d_n, bounds: Tuple[int, int, int, int],
                           d_matches12, d_nmatches, d_stats=None, windowSize: int = 100, nnratio: float = 0.9,
                           checkOri: bool = True, capacity: Optional[int] = None) -> None:
        first = np.ascontiguousarray(first, np.int32)
        second = np.ascontiguousarray(second, np.int32)
        b = _Bounds(*[int(v) for v in bounds])
        if len(first):
            nfr = int(max(first.max(), second.max())) + 1
            _need_batch(None, nfr, 0, 0, 0, 0, d_kps, d_desc, d_n, int(capacity or self.capacity), first, second, d_matches12, d_nmatches, d_stats)
        self._order_torch(d_kps, d_desc, d_n, d_matches12, d_nmatches, d_stats)
        r = self._L.orbx_match_init_batch_device(self._h, len(first), _ptr(first), _ptr(second), _ptr(d_kps), _ptr(d_desc),
                                                 _ptr(d_n), int(capacity or self.capacity), ctypes.byref(b), int(windowSize),
                                                 float(nnratio), int(bool(checkOri)), _ptr(d_matches12), _ptr(d_nmatches),
                                                 _ptr(d_stats))
        self._check(r, "orbx_match_init_batch_device")

    def extract_match_batch_device(self, d_imgs, n_frames: int, width: int, height: int, stride: int, frame_stride: int,
                                   d_kps, d_desc, d_n, first: np.ndarray, second: np.ndarray,
                                   bounds: Tuple[int, int, int, int], d_matches12, d_nmatches, d_stats=None,
                                   windowSize: int = 100, nnratio: float = 0.9, checkOri: bool = True,
                                   capacity: Optional[int] = None) -> None:
        """The whole hot path of one batch (extract + match of pairs inside the batch), two half-batches on two streams."""
        first = np.ascontiguousarray(first, np.int32)
        second = np.ascontiguousarray(second, np.int32)
        b = _Bounds(*[int(v) for v in bounds])
        _need_batch(d_imgs, int(n_frames), int(width), int(height), int(stride), int(frame_stride), d_kps, d_desc, d_n,
                    int(capacity or self.capacity), first, second, d_matches12, d_nmatches, d_stats)
        self._order_torch(d_imgs, d_kps, d_desc, d_n, d_matches12, d_nmatches, d_stats)
        r = self._L.orbx_extract_match_batch_device(self._h, int(n_frames), _ptr(d_imgs), int(width), int(height), int(stride),
                                                    int(frame_stride), _ptr(d_kps), _ptr(d_desc), int(capacity or self.capacity),
                                                    _ptr(d_n), len(first), _ptr(first), _ptr(second), ctypes.byref(b),
                                                    int(windowSize), float(nnratio), int(bool(checkOri)), _ptr(d_matches12),
                                                    _ptr(d_nmatches), _ptr(d_stats))
        self._check(r, "orbx_extract_match_batch_device")

    def extract_match_batch_device_async(self, d_imgs, n_frames: int, width: int, height: int, stride: int, frame_stride: int,
                                         d_kps, d_desc, d_n, first: np.ndarray, second: np.ndarray,
                                         bounds: Tuple[int, int, int, int], d_matches12, d_nmatches, d_stats=None,
                                         windowSize: int = 100, nnratio: float = 0.9, checkOri: bool = True,
                                         capacity: Optional[int] = None) -> None:
        """Stream-ordered form: issues the batch and returns; at most two batches in flight (``wait_one`` / ``wait``).
        Batches in flight together need different output arrays."""
        first = np.ascontiguousarray(first, np.int32)
        second = np.ascontiguousarray(second, np.int32)
        b = _Bounds(*[int(v) for v in bounds])
        _need_batch(d_imgs, int(n_frames), int(width), int(height), int(stride), int(frame_stride), d_kps, d_desc, d_n,
                    int(capacity or self.capacity), first, second, d_matches12, d_nmatches, d_stats)
        self._order_torch(d_imgs, d_kps, d_desc, d_n, d_matches12, d_nmatches, d_stats)
        r = self._L.orbx_extract_match_batch_device_async(self._h, int(n_frames), _ptr(d_imgs), int(width), int(height),
                                                          int(stride), int(frame_stride), _ptr(d_kps), _ptr(d_desc),
                                                          int(capacity or self.capacity), _ptr(d_n), len(first), _ptr(first),
                                                          _ptr(second), ctypes.byref(b), int(windowSize), float(nnratio),
                                                          int(bool(checkOri)), _ptr(d_matches12), _ptr(d_nmatches), _ptr(d_stats))
        self._check(r, "orbx_extract_match_batch_device_async")

    def extract_match_batch_host_async(self, h_imgs, n_frames: int, width: int, height: int, stride: int, frame_stride: int,
                                       h_kps, h_desc, h_n, first: np.ndarray, second: np.ndarray,
                                       bounds: Tuple[int, int, int, int], h_matches12, h_nmatches, h_stats=None,
                                       windowSize: int = 100, nnratio: float = 0.9, checkOri: bool = True,
                                       capacity: Optional[int] = None) -> None:
        """Stream-ordered call for HOST frames and HOST result arrays (numpy arrays or pinned torch CPU tensors; page-locked memory
        keeps the copies asynchronous): upload, kernels and the copies back are queued on the lane the batch goes to.  Valid after
        ``wait_one`` / ``wait``; everything must stay untouched until then."""
        first = np.ascontiguousarray(first, np.int32)
        second = np.ascontiguousarray(second, np.int32)
        b = _Bounds(*[int(v) for v in bounds])
        _need_batch(h_imgs, int(n_frames), int(width), int(height), int(stride), int(frame_stride), h_kps, h_desc, h_n,
                    int(capacity or self.capacity), first, second, h_matches12, h_nmatches, h_stats)
        r = self._L.orbx_extract_match_batch_host_async(self._h, int(n_frames), _ptr(h_imgs), int(width), int(height),
                                                        int(stride), int(frame_stride), _ptr(h_kps), _ptr(h_desc),
                                                        int(capacity or self.capacity), _ptr(h_n), len(first), _ptr(first),
                                                        _ptr(second), ctypes.byref(b), int(windowSize), float(nnratio),
                                                        int(bool(checkOri)), _ptr(h_matches12), _ptr(h_nmatches), _ptr(h_stats))
        self._check(r, "orbx_extract_match_batch_host_async")

    def set_pipeline_depth(self, depth: int) -> None:
        """depth >= 1: every stream-ordered batch goes, whole, to the next of `depth` lanes (own stream, own buffers); at most `depth`
        batches in flight, which need `depth` different output arrays.  0 (default): two half batches on the context's two streams."""
        self._check(self._L.orbx_set_pipeline_depth(self._h, int(depth)), "orbx_set_pipeline_depth")

    def wait_one(self) -> None:
        """Waits for the oldest batch in flight."""
        self._check(self._L.orbx_wait_one(self._h), "orbx_wait_one")

    def wait(self) -> None:
        """Waits for every batch in flight."""
        self._check(self._L.orbx_wait(self._h), "orbx_wait")

    # -- Converter::toGray (Utils/Converter.cpp:5-19) --------------------------------------------------
    def to_gray(self, image: np.ndarray, bRGB: bool = False) -> np.ndarray:
        """(h, w) or (h, w, 1) copies; (h, w, 3) is cvtColor(RGB2GRAY if bRGB else BGR2GRAY); else OrbxError (returns false upstream)."""
        im = np.asarray(image)
        if im.dtype != np.uint8 or im.ndim not in (2, 3):
            raise OrbxError(E_BADARG, "image must be uint8 (h, w) or (h, w, c)")
        ch = 1 if im.ndim == 2 else im.shape[2]
        h, w = im.shape[:2]
        if im.strides[-1] != 1 or (im.ndim == 3 and im.strides[1] != ch):
            im = np.ascontiguousarray(im)
        out = np.zeros((h, w), np.uint8)
        self._check(self._L.orbx_to_gray(self._h, _ptr(im), w, h, im.strides[0] if h else w * ch, ch, int(bool(bRGB)), _ptr(out),
                                         w), "orbx_to_gray")
        return out

    def to_gray_batch_device(self, d_src, n_frames: int, width: int, height: int, stride: int, frame_stride: int, channels: int,
                             bRGB: bool, d_gray, gray_stride: int, gray_frame_stride: int) -> None:
        self._order_torch(d_src, d_gray)
        self._check(self._L.orbx_to_gray_batch_device(self._h, int(n_frames), _ptr(d_src), int(width), int(height), int(stride),
                                                      int(frame_stride), int(channels), int(bool(bRGB)), _ptr(d_gray),
                                                      int(gray_stride), int(gray_frame_stride)), "orbx_to_gray_batch_device")

    # -- Frame::UndistortKeyPoints / ComputeImageBounds (SlamTypes/Frame.cpp:101-161) ----------------
    def undistort_keypoints(self, kps: np.ndarray, camera: Sequence[float]) -> np.ndarray:
        """mvKeysUn from mvKeys; camera = (fx, fy, cx, cy, k1, k2, p1, p2)."""
        kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
        out = np.zeros(len(kps), KEYPOINT_DTYPE)
        cam = _Camera(*[float(v) for v in camera])
        self._check(self._L.orbx_undistort_keypoints(self._h, _ptr(kps), len(kps), ctypes.byref(cam), _ptr(out)),
                    "orbx_undistort_keypoints")
        return out

    def undistort_batch_device(self, n_frames: int, d_kps, d_n, camera: Sequence[float], d_kps_un,
                               capacity: Optional[int] = None) -> None:
        cam = _Camera(*[float(v) for v in camera])
        self._order_torch(d_kps, d_n, d_kps_un)
        self._check(self._L.orbx_undistort_batch_device(self._h, int(n_frames), _ptr(d_kps), _ptr(d_n),
                                                        int(capacity or self.capacity), ctypes.byref(cam), _ptr(d_kps_un)),
                    "orbx_undistort_batch_device")

    def image_bounds(self, camera: Sequence[float], width: int, height: int) -> Tuple[int, int, int, int]:
        cam = _Camera(*[float(v) for v in camera])
        b = _Bounds()
        self._check(self._L.orbx_image_bounds(self._h, ctypes.byref(cam), int(width), int(height), ctypes.byref(b)),
                    "orbx_image_bounds")
        return (b.min_x, b.max_x, b.min_y, b.max_y)

    # -- Initializer::CheckHomography / CheckFundamental (Initialization/Initializer.cpp:268-438) ------
    def _check_models(self, kind: int, M21, M12, keys1, keys2, matches12, sigma: float):
        M21 = np.ascontiguousarray(M21, np.float32).reshape(-1, 3, 3)
        k1 = np.ascontiguousarray(keys1, KEYPOINT_DTYPE)
        k2 = np.ascontiguousarray(keys2, KEYPOINT_DTYPE)
        m12 = np.ascontiguousarray(matches12, np.int32)
        if len(m12) != len(k1):
            raise OrbxError(E_BADARG, "matches12 must have one entry per keypoint of frame 1")
        nm = len(M21)
        scores = np.zeros(nm, np.float32)
        inl = np.zeros(max(nm * len(k1), 1), np.uint8)
        n, best = ctypes.c_int(0), ctypes.c_int(-1)
        if kind == 0:
            M12 = np.ascontiguousarray(M12, np.float32).reshape(-1, 3, 3)
            if len(M12) != nm:
                raise OrbxError(E_BADARG, "H21 and H12 must hold the same number of models")
            r = self._L.orbx_check_homography(self._h, nm, _ptr(M21), _ptr(M12), _ptr(k1), len(k1), _ptr(k2), len(k2), _ptr(m12),
                                              float(sigma), _ptr(scores), _ptr(inl), ctypes.byref(n), ctypes.byref(best))
        else:
            r = self._L.orbx_check_fundamental(self._h, nm, _ptr(M21), _ptr(k1), len(k1), _ptr(k2), len(k2), _ptr(m12),
                                               float(sigma), _ptr(scores), _ptr(inl), ctypes.byref(n), ctypes.byref(best))
        self._check(r, "orbx_check_homography" if kind == 0 else "orbx_check_fundamental")
        return scores, inl[:nm * n.value].reshape(nm, n.value).astype(bool), best.value

    def check_homography(self, H21, H12, keys1, keys2, matches12, sigma: float = 1.0):
        """CheckHomography for a stack of hypotheses -> (scores, vbMatchesInliers per model, index the RANSAC loop keeps)."""
        return self._check_models(0, H21, H12, keys1, keys2, matches12, sigma)

    def check_fundamental(self, F21, keys1, keys2, matches12, sigma: float = 1.0):
        """CheckFundamental for a stack of hypotheses -> (scores, vbMatchesInliers per model, index the RANSAC loop keeps)."""
        return self._check_models(1, F21, None, keys1, keys2, matches12, sigma)

    def check_rt(self, R21, t21, K, keys1, keys2, matches12, matches_inliers, th2: float = 4.0):
        """Initializer::CheckRT (Initializer.cpp:569-713) for a stack of (R21, t21) hypotheses ->
        (nGood[m], vbTriGood[m, n1], vP3D[m, n1, 3], parallax[m])."""
        R21 = np.ascontiguousarray(R21, np.float32).reshape(-1, 3, 3)
        t21 = np.ascontiguousarray(t21, np.float32).reshape(-1, 3)
        K = np.ascontiguousarray(K, np.float32).reshape(3, 3)
        k1 = np.ascontiguousarray(keys1, KEYPOINT_DTYPE)
        k2 = np.ascontiguousarray(keys2, KEYPOINT_DTYPE)
        m12 = np.ascontiguousarray(matches12, np.int32)
        inl = np.ascontiguousarray(matches_inliers, np.uint8)
        nm = len(R21)
        if len(t21) != nm or len(m12) != len(k1) or len(inl) != int((m12 >= 0).sum()):
            raise OrbxError(E_BADARG, "check_rt: one t21 per R21, one matches12 entry per keypoint of frame 1, one inlier flag per match")
        ngood = np.zeros(max(nm, 1), np.int32)
        par = np.zeros(max(nm, 1), np.float32)
        good = np.zeros((max(nm, 1), max(len(k1), 1)), np.uint8)
        p3d = np.zeros((max(nm, 1), max(len(k1), 1), 3), np.float32)
        self._check(self._L.orbx_check_rt(self._h, nm, _ptr(R21), _ptr(t21), _ptr(K), _ptr(k1), len(k1), _ptr(k2), len(k2), _ptr(m12),
                                          _ptr(inl), float(th2), _ptr(ngood), _ptr(good), _ptr(p3d), _ptr(par)), "orbx_check_rt")
        return ngood[:nm], good[:nm, :len(k1)].astype(bool), p3d[:nm, :len(k1)], par[:nm]

    # -- mvImagePyramid (hpp:111) ----------------------------------------------------------------
    def level_size(self, level: int) -> Tuple[int, int]:
        w, h = ctypes.c_int(0), ctypes.c_int(0)
        self._check(self._L.orbx_level_size(self._h, level, ctypes.byref(w), ctypes.byref(h)), "orbx_level_size")
        return w.value, h.value

    def image_pyramid(self, level: int, frame: int = 0, border: int = 0) -> np.ndarray:
        w, h = self.level_size(level)
        out = np.zeros((h + 2 * border, w + 2 * border), np.uint8)
        self._check(self._L.orbx_download_pyramid(self._h, frame, level, border, _ptr(out), out.strides[0]), "orbx_download_pyramid")
        return out

    # -- measurement / test hooks ------------------------------------------------------------------
    def profile_enable(self, on: bool = True):
        self._L.orbx_profile_enable(self._h, int(on))

    def profile_stages(self, stages) -> None:
        """Bracket only the given stages (names of STAGES) with events."""
        mask = 0
        for name in stages:
            mask |= 1 << STAGES.index(name)
        self._L.orbx_profile_stages(self._h, mask)

    def profile_reset(self):
        self._L.orbx_profile_reset(self._h)

    def profile_get(self):
        ms = np.zeros(len(STAGES), np.float64)
        cnt = np.zeros(len(STAGES), np.int64)
        self._L.orbx_profile_get(self._h, _ptr(ms), _ptr(cnt))
        return {s: (float(ms[i]), int(cnt[i])) for i, s in enumerate(STAGES)}

    def debug_distribute_device(self, xyr: np.ndarray, min_x: int, max_x: int, min_y: int, max_y: int, n_features: int,
                                variant: int = 0) -> np.ndarray:
        """The device selection kernels on caller-supplied, row-major ordered candidates (test hook)."""
        xyr = np.ascontiguousarray(xyr, np.float32).reshape(-1, 3)
        out = np.zeros((max(n_features, 1), 3), np.float32)
        r = self._check(self._L.orbx_debug_distribute_device(self._h, _ptr(xyr), len(xyr), min_x, max_x, min_y, max_y,
                                                             n_features, variant, _ptr(out), len(out)))
        return out[:r]

    def debug_std_sort(self, triples: np.ndarray) -> np.ndarray:
        t = np.ascontiguousarray(triples, np.int32).reshape(-1, 3).copy()
        self._check(self._L.orbx_debug_std_sort(self._h, _ptr(t), len(t)))
        return t

    def debug_sincos(self, angles_deg: np.ndarray):
        a = np.ascontiguousarray(angles_deg, np.float32)
        c, s = np.zeros(len(a), np.float32), np.zeros(len(a), np.float32)
        self._check(self._L.orbx_debug_sincos(self._h, _ptr(a), len(a), _ptr(c), _ptr(s)))
        return c, s

    def debug_last_launch(self) -> dict:
        """How the last extraction batch was issued (include/orbx.h: orbx_debug_last_launch)."""
        v = np.zeros(8, np.int32)
        self._check(self._L.orbx_debug_last_launch(self._h, _ptr(v)), "orbx_debug_last_launch")
        return dict(pyramid_banded=int(v[0]), pyramid_bands=int(v[1]), fast_wave=int(v[2]), octree_instance=int(v[3]),
                    split=int(v[4]) & 1, staged_lists=(int(v[4]) >> 1) & 1, frames_per_launch=int(v[5]), wide_with_batch=int(v[6]),
                    lane=int(v[7]))

    def debug_match_counters(self) -> dict:
        """Which matcher kernels did the work, cumulative for this context (include/orbx.h: orbx_debug_match_counters)."""
        v = np.zeros(4, np.uint32)
        self._check(self._L.orbx_debug_match_counters(self._h, _ptr(v)), "orbx_debug_match_counters")
        return dict(bf_mfma_blocks=int(v[0]))

    def debug_selection_units(self, frame: int = 0):
        """(counts, redone) per pyramid level of one frame of the last batch (include/orbx.h: orbx_debug_selection_units)."""
        c, r = np.zeros(self.GetLevels(), np.int32), np.zeros(self.GetLevels(), np.int32)
        self._check(self._L.orbx_debug_selection_units(self._h, int(frame), _ptr(c), _ptr(r)), "orbx_debug_selection_units")
        return c, r

    def debug_candidates(self, frame: int, level: int) -> np.ndarray:
        n = self._check(self._L.orbx_debug_candidates(self._h, frame, level, None, 0))
        out = np.zeros((max(n, 1), 3), np.float32)
        self._check(self._L.orbx_debug_candidates(self._h, frame, level, _ptr(out), n))
        return out[:n]


def camera_from(K, distCoef) -> Tuple[float, ...]:
    """(fx, fy, cx, cy, k1, k2, p1, p2) from mK (3x3) and mDistCoef (4 entries), both float32 as in Settings.hpp:28-39."""
    K = np.asarray(K, np.float32).reshape(3, 3)
    d = np.asarray(distCoef, np.float32).reshape(-1)
    if d.size != 4:
        raise OrbxError(E_BADARG, "distCoef must hold k1 k2 p1 p2")
    return (float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), float(d[0]), float(d[1]), float(d[2]), float(d[3]))


class Frame:
    """The slice of SlamTypes/Frame.{hpp,cpp} on this path: runs the extractor (Frame.cpp:58-60), undistorts the
    keypoints (Frame.cpp:136-161) and keeps mvKeys / mvKeysUn / mDescriptors / N and the image bounds
    (Frame.cpp:101-134).  K / distCoef = None means no distortion (mvKeysUn = mvKeys, bounds = the image)."""

    def __init__(self, im: np.ndarray, timestamp: float, extractor: ORBextractor, K=None, distCoef=None):
        self.mTimestamp = timestamp
        self.mpORBextractor = extractor
        h, w = im.shape
        self.camera = camera_from(K, distCoef) if K is not None and distCoef is not None else None
        if self.camera is not None:
            self.bounds = extractor.image_bounds(self.camera, w, h)  # Frame.cpp:44
        else:
            self.bounds = (0, w, 0, h)  # mnMinX, mnMaxX, mnMinY, mnMaxY (Frame.cpp:127-131)
        _, self.mvKeys, self.mDescriptors = extractor(im, None, (0, 0))
        if self.camera is not None and len(self.mvKeys):
            self.mvKeysUn = extractor.undistort_keypoints(self.mvKeys, self.camera)  # Frame.cpp:64
        else:
            self.mvKeysUn = self.mvKeys  # Frame.cpp:137-140
        self.N = len(self.mvKeysUn)

    @classmethod
    def from_arrays(cls, keys: np.ndarray, descriptors: np.ndarray, bounds: Tuple[int, int, int, int]) -> "Frame":
        f = cls.__new__(cls)
        f.mTimestamp, f.mpORBextractor = 0.0, None
        f.mvKeys = f.mvKeysUn = np.ascontiguousarray(keys, KEYPOINT_DTYPE)
        f.mDescriptors = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        f.N = len(f.mvKeysUn)
        f.bounds = tuple(int(v) for v in bounds)
        return f


class ORBmatcher:
    """Features/ORBmatcher.hpp:13-61."""

    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30

    def __init__(self, nnratio: float = 0.6, checkOri: bool = True, extractor: Optional[ORBextractor] = None):
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)
        self._ext = extractor
        self.last_stats = None

    def SearchForInitialization(self, F1: Frame, F2: Frame, windowSize: int = 100):
        """Returns (nmatches, vnMatches12).  nmatches keeps the reference's double-decrement quirk (cpp:95-98,130-138)."""
        ext = self._ext or F1.mpORBextractor or F2.mpORBextractor
        if ext is None:
            raise OrbxError(E_BADARG, "ORBmatcher needs an ORBextractor (device context); pass extractor=")
        k1 = np.ascontiguousarray(F1.mvKeysUn, KEYPOINT_DTYPE)
        k2 = np.ascontiguousarray(F2.mvKeysUn, KEYPOINT_DTYPE)
        d1 = np.ascontiguousarray(F1.mDescriptors, np.uint8)
        d2 = np.ascontiguousarray(F2.mDescriptors, np.uint8)
        m12 = np.full(max(len(k1), 1), -1, np.int32)
        st = _Stats()
        b = _Bounds(*[int(v) for v in F2.bounds])
        nm = ctypes.c_int32(0)
        r = ext._L.orbx_match_init(ext._h, _ptr(k1), _ptr(d1), len(k1), _ptr(k2), _ptr(d2), len(k2), ctypes.byref(b),
                                   int(windowSize), self.mfNNratio, int(self.mbCheckOrientation), _ptr(m12), ctypes.byref(nm),
                                   ctypes.byref(st))
        ext._check(r, "orbx_match_init")
        self.last_stats = (st.invalid_by_distance, st.invalid_by_ratio, st.invalid_by_orientation)
        return int(nm.value), m12[:len(k1)].copy()
