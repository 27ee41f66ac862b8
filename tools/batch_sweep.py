#!/usr/bin/env python3
"""the latency-bound and the throughput regime side by side.  Frames resident in HBM, 640x480, 1000 features;
B frames + B // 2 consecutive pairs per call: synchronous calls, stream-ordered calls (two batches in flight, each as two half batches
on the context's two streams) and stream-ordered calls on lanes (orbx_set_pipeline_depth: whole batches, LANES in flight)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import orb_slam_tracking_amd as orbx  # noqa: E402
from orb_slam_tracking_amd import synth  # noqa: E402

W, H, cap = 640, 480, 1000
LANES = int(os.environ.get("SWEEP_LANES", "4"))
BYTES_PER_FRAME = 5742474  # DESIGN.md section 5 (SURVEY 8(d) stage-streaming model)
frames = synth.synth_frames(256, W, H, seed0=1000)
d_all = torch.from_numpy(frames).cuda()
for B in [int(x) for x in os.environ.get("SWEEP_B", "1,2,8,32,128,256").split(",")]:
    ext = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
    first = np.arange(0, B - 1, 2, dtype=np.int32)
    npairs = len(first)
    outs = [dict(k=torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"), d=torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda"),
                 n=torch.zeros(B, dtype=torch.int32, device="cuda"), m=torch.zeros(max(npairs, 1) * cap, dtype=torch.int32, device="cuda"),
                 nm=torch.zeros(max(npairs, 1), dtype=torch.int32, device="cuda")) for _ in range(LANES)]
    d_img = d_all[:B]
    res = {"batch": B, "pairs": npairs}
    for mode in ("sync", "stream-ordered", "lanes"):
        ext.set_pipeline_depth(LANES if mode == "lanes" else 0)
        nout = LANES if mode == "lanes" else 2

        def call(k):
            o = outs[k % nout]
            f = ext.extract_match_batch_device_async if mode != "sync" else ext.extract_match_batch_device
            f(d_img, B, W, H, W, W * H, o["k"], o["d"], o["n"], first, first + 1, (0, W, 0, H), o["m"], o["nm"], None, 100, 0.9, True, cap)
        for k in range(5):
            call(k)
        ext.wait()
        K = max(20, min(400, 4096 // B))
        t0 = time.perf_counter()
        for k in range(K):
            call(k)
        ext.wait()
        dt = (time.perf_counter() - t0) / K
        res[mode] = {"ms_per_call": round(dt * 1e3, 4), "frames_per_s": round(B / dt, 1),
                     "algorithmic_GBs": round(B * BYTES_PER_FRAME / dt / 1e9, 2),
                     "frac_of_8TBs": round(B * BYTES_PER_FRAME / dt / 8e12, 5)}
    print(json.dumps(res), flush=True)
    ext.close()
