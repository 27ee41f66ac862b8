"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU side (sanitizers on the host build; the GPU
pool offers none): the oracle (`make -C oracle asan`) and the host prototype of the path-code quadtree (`make -C tests/cpp
asan`) run a representative workload in a child process with libasan preloaded; any report fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKLOAD = r'''
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, %(root)r)
import oracle_lib as O
from orb_slam_tracking_amd import synth
a, b = synth.synth_pair(320, 240, 3)
oe = O.Extractor(500, 1.2, 6, 20, 7)
ra, ka, da = oe(a); rb, kb, db = oe(b)
assert ra == len(ka) > 50
nm, m12, st = O.match_init(ka, da, kb, db, (0, 320, 0, 240), 100, 0.9, True)
ra2, k2, d2 = oe(a, lap=(50, 200))
oe0 = O.Extractor(800, 1.2, 4, 0, 0)   # thresholds 0/0: the dense-candidate path of DistributeOctTree
r0, k0, d0 = oe0(synth.synth(200, 160, 9))
ku = O.undistort_keypoints(ka, O.SETTINGS_CAMERA); bb = O.image_bounds(O.SETTINGS_CAMERA, 320, 240)
O.match_init(ku, da, O.undistort_keypoints(kb, O.SETTINGS_CAMERA), db, bb, 100, 0.9, True)
rng = np.random.default_rng(1)
xyr = np.stack([rng.integers(0, 300, 400), rng.integers(0, 200, 400), rng.integers(1, 200, 400)], 1).astype(np.float32)
_, u = np.unique(xyr[:, 1] * 4096 + xyr[:, 0], return_index=True); xyr = xyr[np.sort(u)]
sel = O.distribute(xyr, 16, 316, 16, 216, 150)
HQ = ctypes.CDLL(os.path.join(%(root)r, "tests", "cpp", "libhostquadtree_asan.so"))
HQ.hostquadtree_distribute.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
out = np.zeros((len(xyr) + 8, 3), np.float32)
n = HQ.hostquadtree_distribute(xyr.ctypes.data, len(xyr), 16, 316, 16, 216, 150, out.ctypes.data, len(out))
assert n == len(sel) and np.array_equal(out[:n], sel)
r = O.bench_protocol((500, 1.2, 6, 20, 7), np.stack([a, b]), 100, 0.9, 2, 1, 2)
assert r.shape == (2, 2, 3)
print("SANITIZED-OK", len(ka), nm, len(k0), n)
'''


def test_oracle_and_host_quadtree_under_asan_ubsan(tmp_path):
    for d in ("oracle", os.path.join("tests", "cpp")):
        p = subprocess.run(["make", "-C", os.path.join(ROOT, d), "asan"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert p.returncode == 0, p.stdout
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan.so not found next to gcc")
    env = dict(os.environ)
    env.update({"LD_PRELOAD": asan, "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1",
                "ORBX_ORACLE_SO": os.path.join(ROOT, "oracle", "liborbx_oracle_asan.so")})
    p = subprocess.run([sys.executable, "-c", WORKLOAD % {"root": ROOT}], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       env=env, timeout=600)
    assert p.returncode == 0 and "SANITIZED-OK" in p.stdout, p.stdout[-4000:]
    assert "runtime error" not in p.stdout and "AddressSanitizer" not in p.stdout, p.stdout[-4000:]
