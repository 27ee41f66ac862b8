#!/usr/bin/env python3
"""One frame per call (orbx_extract through the binding): k_fast (four waves per cell, the default up to 5000 cells per launch) against
k_fast_wave (knob fast_wg_max_cells = 0) -- medians of 400 calls, alternating."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import orb_slam_tracking_amd as orbx
from orb_slam_tracking_amd import synth
for (w, h, nf) in ((640, 480, 1000), (752, 480, 2000), (1920, 1080, 4000)):
    fr = synth.synth_frames(2, w, h, 77)
    e = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
    for rep in range(2):
        for knob in (None, 0):
            orbx.debug_set("fast_wg_max_cells", knob)
            for _ in range(30):
                e(fr[0])
            ts = []
            for _ in range(400):
                t0 = time.perf_counter(); e(fr[0]); ts.append(time.perf_counter() - t0)
            print("%dx%d/%d fast_wg_max_cells=%s: median %.4f ms p10 %.4f  (fast_wave %d)" % (w, h, nf, knob, np.median(ts) * 1e3, np.percentile(ts, 10) * 1e3, e.debug_last_launch()["fast_wave"]), flush=True)
    orbx.debug_set("fast_wg_max_cells", None)
    e.close()
