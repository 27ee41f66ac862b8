#!/usr/bin/env python3
"""EVERY f32 keypoint angle in [0, 360]: the (cos, sin) pair k_describe_patch computes (orbx_debug_sincos: f64 evaluation rounded to
f32) against the oracle's (float)cos((double)(angle * factorPI)) / (float)sin(...) (Features/ORBextractor.cpp:172-174 with glibc's
libm).  1,135,869,953 angles in chunks of 2^24; prints the number of differing values (must be 0).  Run on the GPU box.
usage: sincos_exhaustive.py [libm variant: 0 = through double (default), 1 = glibc cosf / sinf] [number of angles]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import orb_slam_tracking_amd as orbx  # noqa: E402
import oracle_lib as O  # noqa: E402

top = int(np.float32(360.0).view(np.uint32))  # bit pattern of 360.0f: every non-negative float up to it
chunk = 1 << 24
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lim = int(sys.argv[2]) if len(sys.argv) > 2 else top + 1
e = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=1)
e.set_libm_variant(variant)
O.set_libm_variant(variant)
from concurrent.futures import ThreadPoolExecutor  # noqa: E402
ex = ThreadPoolExecutor(16)
bad = 0
t0 = time.time()
for lo in range(0, lim, chunk):
    bits = np.arange(lo, min(lo + chunk, lim), dtype=np.uint32)
    ang = bits.view(np.float32)
    c, s = e.debug_sincos(ang)
    parts = list(ex.map(O.sincos_deg_batch, np.array_split(ang, 16)))  # (ctypes releases the GIL: 16 host threads)
    co, so = np.concatenate([p_[0] for p_ in parts]), np.concatenate([p_[1] for p_ in parts])
    d = (c.view(np.uint32) != co.view(np.uint32)) | (s.view(np.uint32) != so.view(np.uint32))
    if d.any():
        i = np.nonzero(d)[0]
        bad += len(i)
        print("chunk %#x: %d differ, first angle %r: device (%r, %r) oracle (%r, %r)" % (lo, len(i), ang[i[0]], c[i[0]], s[i[0]], co[i[0]], so[i[0]]), flush=True)
    if (lo // chunk) % 8 == 0:
        print("... %#010x of %#010x, %d differing so far, %.0f s" % (lo, lim, bad, time.time() - t0), flush=True)
print("EXHAUSTIVE SINCOS (libm variant %d): %d angles, %d differing values, %.0f s" % (variant, lim, bad, time.time() - t0))
sys.exit(1 if bad else 0)
