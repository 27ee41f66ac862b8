"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol include/orbx.h
declares and fails loudly without a GPU; the path-code formulation the device quadtree uses (host prototype in
tests/cpp/host_quadtree.cpp, test infrastructure, not part of liborbx.so) equals the oracle's DistributeOctTree."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_abi_exports_every_declared_symbol(orbx):
    hdr = open(os.path.join(ROOT, "include", "orbx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(orbx_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 18
    L = ctypes.CDLL(orbx.lib_path())
    for n in names:
        assert hasattr(L, n), "liborbx.so does not export %s" % n


def test_library_reads_no_environment(orbx):
    """The shipped library holds no getenv and no ORBX_* variable name: kernel choices are the library's own, and what the tests
    and tools want to steer goes through orbx_debug_set's named knobs (csrc/orbx_knobs.h), which the Python loader mirrors."""
    src = os.path.join(ROOT, "orb_slam_tracking_amd", "csrc")
    for fn in os.listdir(src):
        if fn.endswith((".cpp", ".hip", ".inc", ".h")):
            assert "getenv" not in open(os.path.join(src, fn), errors="replace").read(), fn
    blob = open(orbx.lib_path(), "rb").read()
    assert not re.findall(rb"\x00ORBX_[A-Z0-9_]+\x00", blob)
    names = re.findall(r'X\(KNOB_[A-Z0-9_]+, "([a-z0-9_]+)"\)', open(os.path.join(src, "orbx_knobs.h")).read())
    assert tuple(names) == orbx.KNOBS and len(names) >= 20
    L = ctypes.CDLL(orbx.lib_path())
    L.orbx_debug_set.argtypes = [ctypes.c_char_p, ctypes.c_longlong]
    for n in names:
        assert L.orbx_debug_set(n.encode(), orbx.KNOB_UNSET) == 0, n
    assert L.orbx_debug_set(b"no_such_knob", 1) == orbx.E_BADARG and L.orbx_debug_set(None, 1) == orbx.E_BADARG


def test_no_cpu_fallback(orbx):
    """Without a usable HIP device the product must refuse to compute (no CPU path exists)."""
    if _has_gpu():
        pytest.skip("GPU present")
    with pytest.raises(orbx.OrbxError) as e:
        orbx.ORBextractor(1000, 1.2, 8, 20, 7)
    assert e.value.code == orbx.E_HIP


def test_product_does_not_reference_oracle():
    pkg = os.path.join(ROOT, "orb_slam_tracking_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".hpp")) or fn == "Makefile":
                txt = open(os.path.join(dp, fn), errors="replace").read()
                assert "oracle_lib" not in txt and "liborbx_oracle" not in txt and "orbo_" not in txt, fn
    for fn in os.listdir(os.path.join(ROOT, "include")):
        assert "orbo_" not in open(os.path.join(ROOT, "include", fn)).read()


_HQ = None


def host_distribute(xyr, min_x, max_x, min_y, max_y, n_features):
    """DistributeOctTree through the host prototype of the device formulation (tests/cpp/libhostquadtree.so)."""
    global _HQ
    if _HQ is None:
        import subprocess
        d = os.path.join(ROOT, "tests", "cpp")
        p = subprocess.run(["make", "-C", d], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert p.returncode == 0, p.stdout
        _HQ = ctypes.CDLL(os.path.join(d, "libhostquadtree.so"))
        vp, i32 = ctypes.c_void_p, ctypes.c_int
        _HQ.hostquadtree_distribute.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, i32]
    xyr = np.ascontiguousarray(xyr, np.float32).reshape(-1, 3)
    out = np.zeros((len(xyr) + 8, 3), np.float32)
    r = _HQ.hostquadtree_distribute(xyr.ctypes.data, len(xyr), min_x, max_x, min_y, max_y, n_features, out.ctypes.data, len(out))
    assert r >= 0, r
    return out[:r]


def test_liborbx_holds_no_cpu_quadtree(orbx):
    """The product library is the HIP path only: the host prototype lives under tests/."""
    L = ctypes.CDLL(orbx.lib_path())
    assert not hasattr(L, "orbx_debug_distribute") and not hasattr(L, "hostquadtree_distribute")
    assert not os.path.exists(os.path.join(ROOT, "orb_slam_tracking_amd", "csrc", "orbx_octree.cpp"))


def _rand_cands(rng, W, H, n, sort):
    pos = rng.choice(W * H, size=n, replace=False)
    if sort:
        pos = np.sort(pos)
    x = (pos % W).astype(np.float32)
    y = (pos // W).astype(np.float32)
    r = rng.integers(1, int(rng.integers(2, 200)), size=n).astype(np.float32)
    return np.stack([x, y, r], 1)


def test_octree_matches_oracle_random(orbx, oracle):
    """DistributeOctTree (Features/ORBextractor.cpp:698-1011): same selected keys in the same list order."""
    rng = np.random.default_rng(0)
    done = 0
    for it in range(600):
        W, H = int(rng.integers(40, 900)), int(rng.integers(40, 700))
        if round(float(np.float32(W) / np.float32(H))) < 1:
            continue
        n = int(rng.integers(0, min(W * H // 4, 2500)))
        xyr = _rand_cands(rng, W, H, n, rng.random() < 0.5)
        N = int(rng.integers(0, max(2, 2 * n // 3 + 2)))
        a = oracle.distribute(xyr, 16, 16 + W, 16, 16 + H, N)
        b = host_distribute(xyr, 16, 16 + W, 16, 16 + H, N)
        assert a.shape == b.shape and np.array_equal(a, b), (it, W, H, n, N)
        done += 1
    assert done > 400


@pytest.mark.parametrize("shape", [(608, 448, 217), (720, 448, 434), (1888, 1048, 869), (3808, 2128, 1737), (147, 102, 60)])
def test_octree_level_geometries(orbx, oracle, shape):
    """Sizes of real pyramid levels (incl. nIni == 2 for 16:9 / 752x480) with clustered candidates and many ties."""
    W, H, N = shape
    rng = np.random.default_rng(W)
    for dens in (0.002, 0.01, 0.05):
        n = int(W * H * dens)
        xyr = _rand_cands(rng, W, H, n, True)
        xyr[:, 2] = rng.integers(6, 12, n)  # heavy response ties -> first-in-order rule decides
        cx, cy = W * 0.3, H * 0.6  # cluster half of the points
        half = n // 2
        xyr[:half, 0] = np.clip(np.round(cx + rng.normal(0, W * 0.05, half)), 0, W - 1)
        xyr[:half, 1] = np.clip(np.round(cy + rng.normal(0, H * 0.05, half)), 0, H - 1)
        _, uniq = np.unique(xyr[:, 1] * 4096 + xyr[:, 0], return_index=True)
        xyr = xyr[np.sort(uniq)]
        a = oracle.distribute(xyr, 16, 16 + W, 16, 16 + H, N)
        b = host_distribute(xyr, 16, 16 + W, 16, 16 + H, N)
        assert np.array_equal(a, b)


def test_octree_edge_cases(orbx, oracle):
    for xyr, N in [(np.zeros((0, 3), np.float32), 10), (np.array([[5, 5, 9]], np.float32), 10),
                   (np.array([[5, 5, 9], [6, 5, 9]], np.float32), 1), (np.array([[5, 5, 9], [300, 5, 9]], np.float32), 0)]:
        a = oracle.distribute(xyr, 16, 16 + 600, 16, 16 + 400, N)
        b = host_distribute(xyr, 16, 16 + 600, 16, 16 + 400, N)
        assert np.array_equal(a, b)


def test_path_code_tables_equal_the_divide_node_walk(orbx):
    """The selection kernels take a candidate's quadtree path code from two per-level tables (x digits + root by x, y digits by y:
    DivideNode routes the two coordinates independently, Features/ORBextractor.cpp:656-668).  Host-only hook, no device: for
    real level geometries (nIni 1 and 2, non-integral hX), thin strips (nIni up to 10) and small regions the table code equals
    the code of the 16-split walk for EVERY pixel, incl. the columns at the roots' boundaries (cpp:715-716, 747), and points
    that differ anywhere differ in their codes (the code identifies the pixel)."""
    L = orbx.lib()
    rng = np.random.default_rng(3)
    shapes = [(608, 448), (720, 448), (1888, 1048), (3808, 2128), (147, 102), (501, 167), (1000, 99), (97, 97), (35, 31),
              (767, 256), (1535, 512), (4096, 409)]
    for (W, H) in shapes:
        if W * H <= 400000:
            ys, xs = np.divmod(np.arange(W * H, dtype=np.int32), W)
        else:  # large regions: every pixel of a random third of the rows
            rows = np.sort(rng.choice(H, H // 3, replace=False)).astype(np.int32)
            ys, xs = np.repeat(rows, W), np.tile(np.arange(W, dtype=np.int32), len(rows))
        xs, ys = np.ascontiguousarray(xs, np.int32), np.ascontiguousarray(ys, np.int32)
        a, b = np.zeros(len(xs), np.uint64), np.zeros(len(xs), np.uint64)
        r = L.orbx_debug_path_codes(W, H, len(xs), xs.ctypes.data, ys.ctypes.data, a.ctypes.data, b.ctypes.data)
        assert r == 0, (W, H, r)
        assert np.array_equal(a, b), (W, H, np.nonzero(a != b)[0][:5])
        assert len(np.unique(a)) == len(a), (W, H)  # 16 splits separate every pixel of a region up to 4096 wide
    assert L.orbx_debug_path_codes(0, 10, 0, None, None, None, None) < 0


def test_synth_is_deterministic():
    from orb_slam_tracking_amd import synth
    a, b = synth.synth_pair(160, 120, 5)
    a2, b2 = synth.synth_pair(160, 120, 5)
    assert np.array_equal(a, a2) and np.array_equal(b, b2) and not np.array_equal(a, b)
    import hashlib
    assert hashlib.sha256(a.tobytes()).hexdigest()[:16] == hashlib.sha256(synth.synth(160, 120, 5).tobytes()).hexdigest()[:16]


def build_shim_demo(orbx, out_dir):
    """Compiles tests/cpp/shim_demo.cpp (the reference's demo call sequence over include/orbx_shim.hpp) with plain g++."""
    import subprocess
    exe = os.path.join(out_dir, "shim_demo")
    libdir = os.path.dirname(orbx.lib_path())
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "shim_demo.cpp"),
           "-L", libdir, "-lorbx", "-Wl,-rpath," + libdir, "-o", exe]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout
    return exe


def build_shim_latency(lib_path, out_dir):
    """Compiles tests/cpp/shim_latency.cpp: the reference's one-frame-per-call use of the C++ drop-in classes, timed (bench.py's
    single_frame.cpp_shim)."""
    import subprocess
    exe = os.path.join(out_dir, "shim_latency")
    libdir = os.path.dirname(lib_path)
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "shim_latency.cpp"),
           "-L", libdir, "-lorbx", "-Wl,-rpath," + libdir, "-o", exe]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout
    return exe


def build_shim_opencv_frame(orbx, out_dir):
    """Compiles tests/cpp/shim_opencv_frame.cpp: the -DORBX_WITH_OPENCV branch of the shim (the reference's real cv::
    signatures) against the compile-check mock of six cv:: types in tests/cpp/mock_opencv (which pins nothing)."""
    import subprocess
    exe = os.path.join(out_dir, "shim_opencv_frame")
    libdir = os.path.dirname(orbx.lib_path())
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-DORBX_WITH_OPENCV", "-I", os.path.join(ROOT, "tests", "cpp", "mock_opencv"),
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "shim_opencv_frame.cpp"),
           "-L", libdir, "-lorbx", "-Wl,-rpath," + libdir, "-o", exe]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout
    return exe


def test_cpp_shim_opencv_branch_compiles(orbx, tmp_path):
    """The branch that carries the reference's own signatures (cv::InputArray / cv::OutputArray / std::vector<cv::KeyPoint>&,
    and SearchForInitialization(Frame&, Frame&, ...) with the context taken from Frame::mpORBextractor) is compiled here on
    every CPU run; the GPU suite runs it (tests/test_gpu_boundary.py)."""
    exe = build_shim_opencv_frame(orbx, str(tmp_path))
    assert os.path.exists(exe)
    if not _has_gpu():
        import subprocess
        raw = tmp_path / "z.raw"
        raw.write_bytes(bytes(640 * 480))
        p = subprocess.run([exe, "640", "480", str(raw), str(raw), "1000", "20", "7"], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        assert p.returncode != 0 and "RESULT" not in p.stdout


def test_cpp_shim_compiles_and_links(orbx, tmp_path):
    """The C++ drop-in classes (ORB_SLAM_Tracking::ORBextractor / ORBmatcher) build against the C ABI alone."""
    exe = build_shim_demo(orbx, str(tmp_path))
    assert os.path.exists(exe)
    if not _has_gpu():  # without a device the shim must fail loudly, not compute on the CPU
        import subprocess
        raw = tmp_path / "z.raw"
        raw.write_bytes(bytes(640 * 480))
        p = subprocess.run([exe, "640", "480", str(raw), str(raw), "1000", "20", "7"], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        assert p.returncode != 0 and "RESULT" not in p.stdout


def test_bench_gpus_n_launches_its_ranks():
    """`python bench.py --gpus 2` outside a launcher starts two ranks (torch.distributed.run children) instead of asking for one;
    without a GPU each rank stops at bench.py's own "needs a GPU" exit -- which proves the ranks were started -- and the exit code
    of the children comes back."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-side check of the launcher path")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode != 0
    assert "bench.py needs a GPU" in p.stdout and "launch with torch.distributed.run" not in p.stdout, p.stdout[-2000:]


def test_libm_overload_probe():
    """which libm function the reference's unqualified cos(angle) / pow(factor, float) call is decided by the
    headers in the translation unit.  tools/pin_opencv/libm_probe.cpp lets the compiler say it (the type of cos(1.0f)); with a real
    OpenCV it names the orbx_set_libm_variant to use.  Here, on this image's libstdc++: the reference's standard headers + <cmath>
    give the DOUBLE reading, + <math.h> the FLOAT reading."""
    import subprocess
    src = os.path.join(ROOT, "tools", "pin_opencv", "libm_probe.cpp")
    for flag, want in (("-DORBX_PROBE_CMATH", "ORBX_LIBM_DOUBLE"), ("-DORBX_PROBE_MATH_H", "ORBX_LIBM_FLOAT")):
        exe = "/tmp/orbx_libm_probe%s" % flag[-6:]
        p = subprocess.run(["g++", "-std=c++17", "-DORBX_PROBE_NO_OPENCV", flag, src, "-o", exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert p.returncode == 0, p.stdout
        out = subprocess.run([exe], stdout=subprocess.PIPE, text=True).stdout
        assert out.strip().splitlines()[-1] == "orbx_set_libm_variant: " + want, out


def test_pin_kit_compiles():
    """tools/pin_opencv (the one-command diff of the oracle against a real OpenCV, for whoever has one) at least compiles: its only
    possible check in this image is the compile-check mock of the cv:: declarations it uses (tests/cpp/mock_opencv pins nothing)."""
    import subprocess
    p = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "tests", "cpp", "mock_opencv"),
                        os.path.join(ROOT, "tools", "pin_opencv", "pin_opencv.cpp")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout
    p = subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "pin_opencv", "export_fixtures.py"), "/tmp/orbx_pin_fixtures"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0 and os.path.exists("/tmp/orbx_pin_fixtures/manifest.txt"), p.stdout


def test_one_hip_runtime_whatever_the_import_order():
    """liborbx.so loaded BEFORE torch must not leave the process with two HIP runtimes (torch loads its own libamdhip64.so by path; the
    second runtime finds no device and orbx_create fails): the loader takes torch's copy first when torch is installed."""
    import subprocess
    import sys
    code = ("import orb_slam_tracking_amd as o\no.lib()\nimport torch\n"
            "print(o.hip_runtimes_mapped())")  # (the helper the loader's error text uses when orbx_create finds no device)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-500:]
    libs = eval(r.stdout.strip().splitlines()[-1])
    assert len(libs) == 1, libs


def test_bench_side_measurements_fail_loudly():
    import sys
    """VERDICT r05 item 8: a figure behind the headline (host_pipeline, other_configs, single_frame) that RAISES must not read like
    a success.  bench.py runs every one of them through run_side(): the line is still printed, the error is in it, `all_checked`
    becomes false and the exit code non-zero.  Here: a failing bench_config.measure injected into the helper bench.py uses, no GPU."""
    import importlib
    bench = importlib.import_module("bench")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_config as BC

    def failing_measure(*a, **k):
        raise RuntimeError("injected: hipErrorLaunchFailure")
    saved = BC.measure
    BC.measure = failing_measure
    try:
        out, failures = {}, []
        r = bench.run_side(out, "other_configs", lambda: {"c3": BC.measure("c3", steps=20, depth=3, device=0)}, failures)
    finally:
        BC.measure = saved
    assert failures == ["other_configs"] and "injected" in out["other_configs"]["error"] and r is out["other_configs"]
    assert bench.final_status(True, failures) == (False, 3)      # the headline batch matched, a side figure crashed: still a failure
    assert bench.final_status(True, []) == (True, 0)
    assert bench.final_status(False, []) == (False, 3)            # a mismatch against the oracle
    assert bench.final_status(None, []) == (False, 0)             # --no-check: nothing claimed, nothing failed
    # a figure that was not asked for is recorded as skipped and is no failure
    out, failures = {}, []

    def skipped():
        raise bench.Skipped("--no-other-configs")
    bench.run_side(out, "other_configs", skipped, failures)
    assert failures == [] and out["other_configs"] == {"skipped": "--no-other-configs"}
    # and main() really routes the three figures through the helper and takes its exit code from final_status
    src = open(os.path.join(ROOT, "bench.py")).read()
    for name in ("single_frame", "host_pipeline", "other_configs"):
        assert 'run_side(out, "%s"' % name in src, name
    assert "final_status(check_ok, failures)" in src and "never let them break the line" not in src


def test_binding_refuses_short_buffers(orbx):
    """A batch call whose arrays are shorter than its arguments imply must raise in the binding: behind the C ABI the same call is a
    GPU memory fault (what bench.py's first config4_as_written did on the GPU box: 64 frames' worth of arguments, 16-frame tensors).
    No device needed: the check comes before anything is issued."""
    from orb_slam_tracking_amd import _need_batch
    W, H, cap = 640, 480, 1000
    imgs = np.zeros((16, H, W), np.uint8)
    k, d, n = np.zeros(64 * cap * 28, np.uint8), np.zeros(64 * cap * 32, np.uint8), np.zeros(64, np.int32)
    first = np.arange(0, 64, 2, dtype=np.int32)
    m, nm = np.zeros(32 * cap, np.int32), np.zeros(32, np.int32)
    _need_batch(imgs, 16, W, H, W, W * H, k, d, n, cap, first[:8], first[:8] + 1, m, nm, None)  # fits
    with pytest.raises(ValueError, match="the frames"):
        _need_batch(imgs, 64, W, H, W, W * H, k, d, n, cap, first, first + 1, m, nm, None)
    with pytest.raises(ValueError, match="outside the batch"):
        _need_batch(imgs, 16, W, H, W, W * H, k, d, n, cap, first, first + 1, m, nm, None)
    with pytest.raises(ValueError, match="keypoint array"):
        _need_batch(imgs, 16, W, H, W, W * H, k[:100], d, n, cap)
    with pytest.raises(ValueError, match="matches12"):
        _need_batch(imgs, 16, W, H, W, W * H, k, d, n, cap, first[:8], first[:8] + 1, m[:10], nm, None)
    _need_batch(123456, 64, W, H, W, W * H, 1, 2, 3, cap)  # raw pointers: sizes are the caller's business
