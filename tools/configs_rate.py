#!/usr/bin/env python3
"""Stage times of the other BASELINE.json configurations (they are parity-test cases, not bench lines): frames resident
in HBM, extract_match_batch_device on B frames + B/2 consecutive pairs, per-stage device time from the ctx profile."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import orb_slam_tracking_amd as orbx  # noqa: E402
from orb_slam_tracking_amd import synth  # noqa: E402

CASES = [  # (name, w, h, nfeatures, iniTh, minTh, batch)
    ("640x480/1000 (bench)", 640, 480, 1000, 20, 7, 256),
    ("752x480/2000 FAST 0/0 (the demo's constructor call)", 752, 480, 2000, 0, 0, 64),
    ("1920x1080/4000", 1920, 1080, 4000, 20, 7, 32),
    ("3840x2160/8000", 3840, 2160, 8000, 20, 7, 8),
]
only = sys.argv[1:]
for name, w, h, nf, ini, mn, B in CASES:
    if only and not any(o in name for o in only):
        continue
    frames = synth.synth_frames(B, w, h, seed0=77)
    d_img = torch.from_numpy(frames).cuda()
    cap = nf
    d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
    d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    d_m = torch.zeros((B // 2) * cap, dtype=torch.int32, device="cuda")
    d_nm = torch.zeros(B // 2, dtype=torch.int32, device="cuda")
    first = np.arange(0, B, 2, dtype=np.int32)
    second = first + 1
    ext = orbx.ORBextractor(nf, 1.2, 8, ini, mn, max_width=w, max_height=h, max_batch=B)

    def step():
        ext.extract_match_batch_device(d_img, B, w, h, w, w * h, d_k, d_d, d_n, first, second, (0, w, 0, h), d_m, d_nm,
                                       None, 100, 0.9, True, cap)
    for _ in range(2):
        step()
    ext.profile_enable(True)
    ext.profile_reset()
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    prof = {k: round(v[0] / n, 3) for k, v in ext.profile_get().items()}
    print(json.dumps({"case": name, "batch": B, "ms_per_batch": round(dt * 1e3, 3), "frames_per_s": round(B / dt, 1),
                      "mean_keypoints": float(d_n.float().mean().item()), "mean_nmatches": float(d_nm.float().mean().item()),
                      "stage_ms": prof}), flush=True)
    del ext, d_img, d_k, d_d, d_m
