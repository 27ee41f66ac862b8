#!/usr/bin/env python3
"""The host-frame pipeline alone (orbx_extract_match_batch_host_async, what bench.py reports as `host_pipeline`): page-locked input
sets, page-locked results, `depth` batches in flight.  usage: host_pipeline.py [depth] [batches] [batch size]
Under `rocprofv3 --kernel-trace --memory-copy-trace` it gives the timeline of uploads, kernels and copies back
(tools/host_pipeline_timeline.py reads the two csv files)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import orb_slam_tracking_amd as orbx  # noqa: E402
from orb_slam_tracking_amd import synth  # noqa: E402

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
W, H, cap = 640, 480, 1000
sets = [torch.from_numpy(s).pin_memory() for s in synth.bench_input_sets(B, W, H, 1000, 4)]
outs = [dict(k=torch.zeros(B * cap * 28, dtype=torch.uint8).pin_memory(), d=torch.zeros(B * cap * 32, dtype=torch.uint8).pin_memory(),
             n=torch.zeros(B, dtype=torch.int32).pin_memory(), m=torch.zeros((B // 2) * cap, dtype=torch.int32).pin_memory(),
             nm=torch.zeros(B // 2, dtype=torch.int32).pin_memory()) for _ in range(max(depth, 1))]
first = np.arange(0, B, 2, dtype=np.int32)
ext = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
ext.set_pipeline_depth(depth)
dst = torch.empty(B * W * H, dtype=torch.uint8, device="cuda")
for _ in range(3):
    dst.copy_(sets[0].view(-1), non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    dst.copy_(sets[i % 4].view(-1), non_blocking=True)
torch.cuda.synchronize()
peak = 20 * B * W * H / (time.perf_counter() - t0) / 1e9


def step(i):
    o = outs[i % max(depth, 1)]
    ext.extract_match_batch_host_async(sets[i % 4], B, W, H, W, W * H, o["k"], o["d"], o["n"], first, first + 1, (0, W, 0, H), o["m"], o["nm"],
                                       None, 100, 0.9, True, cap)


for i in range(2 * max(depth, 1)):
    step(i)
ext.wait()
t0 = time.perf_counter()
for i in range(nb):
    step(i)
ext.wait()
dt = time.perf_counter() - t0
fps = nb * B / dt
print("depth %d, %d frames per batch: %.1f k frames/s, %.3f ms per batch, H2D %.1f GB/s = %.2f of the %.1f GB/s hipMemcpyAsync alone reaches"
      % (depth, B, fps / 1e3, dt / nb * 1e3, fps * W * H / 1e9, fps * W * H / 1e9 / peak, peak))
