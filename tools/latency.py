#!/usr/bin/env python3
"""Single-frame latency of the host-buffer API (what a tracker that hands over one cv::Mat per call sees):
orbx_extract on one 640x480 frame, and orbx_match_init on two frames' results.  Wall-clock, synchronous calls."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import orb_slam_tracking_amd as orbx  # noqa: E402
from orb_slam_tracking_amd import synth  # noqa: E402

W, H = 640, 480
a, b = synth.synth_pair(W, H, 5)
ext = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=1)
fa, fb = orbx.Frame(a, 0.0, ext), orbx.Frame(b, 1.0, ext)
m = orbx.ORBmatcher(0.9, True)
for _ in range(5):
    ext(a)
    m.SearchForInitialization(fa, fb, 100)
n = 200
t0 = time.perf_counter()
for _ in range(n):
    ext(a)
t1 = time.perf_counter()
for _ in range(n):
    m.SearchForInitialization(fa, fb, 100)
t2 = time.perf_counter()
print(json.dumps({"extract_ms_per_frame": (t1 - t0) / n * 1e3, "match_ms_per_pair": (t2 - t1) / n * 1e3,
                  "note": "640x480, 1000 features, host buffers in/out, one frame per call"}))
