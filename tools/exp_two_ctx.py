#!/usr/bin/env python3
"""Experiment: does more kernel-level concurrency raise the whole-path throughput?  The bench workload (256 frames + 128 pair
matches per batch, stream-ordered) issued to ONE context (two streams, half a batch each) or alternately to TWO / THREE
contexts (four / six streams).  usage: exp_two_ctx.py [batch]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orb_slam_tracking_amd as orbx
from orb_slam_tracking_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H, cap = 640, 480, 1000
frames = synth.synth_frames(B, W, H, 1000)
d_imgs = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in (frames, frames[:, ::-1, :], frames[:, :, ::-1], frames[:, ::-1, ::-1])]
first = np.arange(0, B, 2, dtype=np.int32); second = first + 1
def outs():
    return dict(k=torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"), d=torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda"),
                n=torch.zeros(B, dtype=torch.int32, device="cuda"), m=torch.zeros((B // 2) * cap, dtype=torch.int32, device="cuda"),
                nm=torch.zeros(B // 2, dtype=torch.int32, device="cuda"))
for nctx in [int(x) for x in os.environ.get("EXP_NCTX", "1,2,3").split(",")]:
    ctxs = [orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B) for _ in range(nctx)]
    O = [[outs(), outs()] for _ in range(nctx)]
    cnt = [0] * nctx
    def step(k):
        c = k % nctx
        o = O[c][cnt[c] & 1]; cnt[c] += 1
        ctxs[c].extract_match_batch_device_async(d_imgs[k & 3], B, W, H, W, W * H, o["k"], o["d"], o["n"], first, second, (0, W, 0, H),
                                                 o["m"], o["nm"], None, 100, 0.9, True, cap)
    def barrier():
        for c in ctxs: c.wait()
        torch.cuda.synchronize()
    for k in range(6): step(k)
    barrier()
    steps = 200
    t0 = time.perf_counter()
    for k in range(steps): step(k)
    barrier()
    dt = time.perf_counter() - t0
    print("contexts %d (streams %d), batch %d: %.0f frames/s, %.3f ms per batch" % (nctx, 2 * nctx, B, steps * B / dt, dt / steps * 1e3), flush=True)
    for c in ctxs: c.close()
