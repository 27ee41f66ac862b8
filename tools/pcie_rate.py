#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer API (orbx_extract_batch: frames from host memory, keypoints/descriptors back
to host memory), for the note in DESIGN.md.  Never bench.py's `value` (that is measured with inputs resident in HBM)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import orb_slam_tracking_amd as orbx  # noqa: E402
from orb_slam_tracking_amd import synth  # noqa: E402

B, W, H = 256, 640, 480
frames = synth.synth_frames(B, W, H, seed0=1000)
ext = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
for _ in range(2):
    ext.extract_batch(frames)
steps = 10
t0 = time.perf_counter()
for _ in range(steps):
    ext.extract_batch(frames)
dt = time.perf_counter() - t0
print(json.dumps({"api": "orbx_extract_batch (pageable host buffers in/out, extraction only)", "frames_per_s": B * steps / dt,
                  "ms_per_batch": dt / steps * 1e3, "batch": B}))
