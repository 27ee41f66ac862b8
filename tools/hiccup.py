#!/usr/bin/env python3
"""Looks for host-side stalls in a stream of stream-ordered bench batches: prints every call that took longer than 1 ms."""
import os, sys, time, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orb_slam_tracking_amd as orbx
from orb_slam_tracking_amd import synth
B, W, H, cap = 256, 640, 480, 1000
depth = int(os.environ.get("DEPTH", "4"))
if os.environ.get("NOGC"):
    gc.disable()
frames = synth.synth_frames(B, W, H, 1000)
d_imgs = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in (frames, frames[:, ::-1, :], frames[:, :, ::-1], frames[:, ::-1, ::-1])]
first = np.arange(0, B, 2, dtype=np.int32); second = first + 1
nout = max(2, depth)
outs = [dict(k=torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"), d=torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda"),
             n=torch.zeros(B, dtype=torch.int32, device="cuda"), m=torch.zeros((B // 2) * cap, dtype=torch.int32, device="cuda"),
             nm=torch.zeros(B // 2, dtype=torch.int32, device="cuda")) for _ in range(nout)]
e = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
if depth:
    e.set_pipeline_depth(depth)
def step(k):
    o = outs[k % nout]
    e.extract_match_batch_device_async(d_imgs[k & 3], B, W, H, W, W * H, o["k"], o["d"], o["n"], first, second, (0, W, 0, H), o["m"], o["nm"], None, 100, 0.9, True, cap)
for k in range(40): step(k)
e.wait(); torch.cuda.synchronize()
N = 1200
ts = np.zeros(N + 1)
ts[0] = time.perf_counter()
for k in range(N):
    step(k)
    ts[k + 1] = time.perf_counter()
e.wait()
tend = time.perf_counter()
dt = np.diff(ts) * 1e3
print("total %.1f ms for %d batches = %.0f frames/s; call time median %.3f ms, mean %.3f" % ((tend - ts[0]) * 1e3, N, N * B / (tend - ts[0]), np.median(dt), dt.mean()))
big = np.nonzero(dt > 2.5)[0]
print("calls longer than 2.5 ms:", [(int(i), round(float(dt[i]), 2)) for i in big][:40])
if len(big) > 1:
    print("spacing between them (batches):", np.diff(big).tolist()[:40])
