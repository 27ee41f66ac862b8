import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import orb_slam_tracking_amd as orbx
from orb_slam_tracking_amd import synth
W, H, cap = 640, 480, 1000
for B, depth in ((32, 3), (32, 4), (256, 4), (8, 3), (1, 3)):
    frames = torch.from_numpy(synth.synth_frames(max(B, 2), W, H, 1000)).cuda()
    e = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=max(B, 2))
    e.set_pipeline_depth(depth)
    outs = [dict(k=torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"), d=torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda"),
                 n=torch.zeros(B, dtype=torch.int32, device="cuda"), m=torch.zeros(max(B // 2, 1) * cap, dtype=torch.int32, device="cuda"),
                 nm=torch.zeros(max(B // 2, 1), dtype=torch.int32, device="cuda")) for _ in range(depth)]
    first = np.arange(0, B - 1, 2, dtype=np.int32)
    def call(i):
        o = outs[i % depth]
        e.extract_match_batch_device_async(frames, B, W, H, W, W * H, o["k"], o["d"], o["n"], first, first + 1, (0, W, 0, H), o["m"], o["nm"], None, 100, 0.9, True, cap)
    for i in range(30): call(i)
    e.wait()
    # host cost of issuing alone: issue `depth` batches into an idle pipeline (no waiting inside), timed
    ts = []
    for rep in range(20):
        e.wait(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(depth): call(i)
        ts.append((time.perf_counter() - t0) / depth)
    e.wait()
    n = 300
    t0 = time.perf_counter()
    for i in range(n): call(i)
    e.wait()
    dt = (time.perf_counter() - t0) / n
    print("B=%d depth %d: host issue %.1f us per batch (median), steady state %.1f us per batch = %.0f k frames/s" % (B, depth, 1e6 * sorted(ts)[len(ts) // 2], dt * 1e6, B / dt / 1e3))
    e.close()
