#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer API (orbx_extract_batch: frames from host memory, keypoints/descriptors back
to host memory), for the note in DESIGN.md.  Never bench.py's `value` (that is measured with inputs resident in HBM)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402  (before liborbx touches HIP: torch's own device discovery must come first)
torch.cuda.init()
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

# the same call with page-locked host buffers (what a capture pipeline that owns its buffers would pass)
import ctypes  # noqa: E402
L = orbx.lib()
pin_in = torch.empty(frames.shape, dtype=torch.uint8).pin_memory()
pin_in.numpy()[:] = frames
cap = ext.capacity
pin_k = torch.empty(B * cap * 28, dtype=torch.uint8).pin_memory()
pin_d = torch.empty(B * cap * 32, dtype=torch.uint8).pin_memory()
n_out = np.zeros(B, np.int32)
mono = np.zeros(B, np.int32)


def call():
    r = L.orbx_extract_batch(ext._h, B, ctypes.c_void_p(pin_in.data_ptr()), W, H, W, W * H, 0, 0, ctypes.c_void_p(pin_k.data_ptr()),
                             ctypes.c_void_p(pin_d.data_ptr()), cap, ctypes.c_void_p(n_out.ctypes.data), ctypes.c_void_p(mono.ctypes.data))
    assert r == 0, r


for _ in range(2):
    call()
t0 = time.perf_counter()
for _ in range(steps):
    call()
dt = time.perf_counter() - t0
print(json.dumps({"api": "orbx_extract_batch (page-locked host buffers in/out, extraction only)", "frames_per_s": B * steps / dt,
                  "ms_per_batch": dt / steps * 1e3, "batch": B}))
