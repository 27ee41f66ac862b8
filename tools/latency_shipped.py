#!/usr/bin/env python3
"""Single-frame latency of the reference demo's own call: ORBextractor(2 * nFeatures = 2000, 1.2, 8, 0, 0) on the shipped
752x480 init images (tests/golden/images.npz), host buffers in and out, and SearchForInitialization on the pair."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import orb_slam_tracking_amd as orbx  # noqa: E402

im = np.load(os.path.join(ROOT, "tests", "golden", "images.npz"))
a, b = im["init0"], im["init1"]
h, w = a.shape
ext = orbx.ORBextractor(2000, 1.2, 8, 0, 0, max_width=w, max_height=h, max_batch=1)
fa, fb = orbx.Frame(a, 0.0, ext), orbx.Frame(b, 1.0, ext)
m = orbx.ORBmatcher(0.9, True)
for _ in range(5):
    ext(a)
    m.SearchForInitialization(fa, fb, 100)
n = 100
t0 = time.perf_counter()
for _ in range(n):
    ext(a)
t1 = time.perf_counter()
for _ in range(n):
    nm, _ = m.SearchForInitialization(fa, fb, 100)
t2 = time.perf_counter()
print(json.dumps({"extract_ms_per_frame": (t1 - t0) / n * 1e3, "match_ms_per_pair": (t2 - t1) / n * 1e3, "nmatches": int(nm),
                  "note": "%dx%d, 2000 features, FAST 0/0, host buffers in/out, one frame per call" % (w, h)}))
