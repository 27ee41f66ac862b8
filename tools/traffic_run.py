#!/usr/bin/env python3
"""A few calls of the hot path on B resident 640x480 frames (+ B/2 consecutive pairs), for the rocprofv3 --pmc FETCH_SIZE /
WRITE_SIZE passes of tools/prof_traffic.sh.  usage: traffic_run.py B [calls]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orb_slam_tracking_amd as orbx
from orb_slam_tracking_amd import synth
B = int(sys.argv[1]); calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W, H, cap = 640, 480, 1000
frames = torch.from_numpy(synth.synth_frames(B, W, H, 1000)).cuda()
e = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"); d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
n = torch.zeros(B, dtype=torch.int32, device="cuda")
P = B // 2
first = np.arange(0, 2 * P, 2, dtype=np.int32); second = first + 1
m = torch.zeros(max(P, 1) * cap, dtype=torch.int32, device="cuda"); nm = torch.zeros(max(P, 1), dtype=torch.int32, device="cuda")
for _ in range(calls):
    if P:
        e.extract_match_batch_device(frames, B, W, H, W, W * H, k, d, n, first, second, (0, W, 0, H), m, nm, None, 100, 0.9, True, cap)
    else:
        e.extract_batch_device(frames, B, W, H, W, W * H, k, d, n, cap)
torch.cuda.synchronize()
print("B=%d calls=%d mean keypoints %.1f" % (B, calls, float(n.float().mean())))
