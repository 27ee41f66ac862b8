#!/usr/bin/env python3
"""Where k_match_bf_mfma loses: BASELINE config 3's frames (a synthetic scene that repeats its corners: about a hundred candidates per
query) in batches of 128, i.e. 64 brute-force pairs per call -- enough blocks for the matrix-core kernel -- with and without it
(knob match_no_mfma): matching stage per batch and frames/s (docs/history.md, round 5)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import orb_slam_tracking_amd as orbx
import bench_config as BC
for knob in (None, 1, None, 1):
    orbx.debug_set("match_no_mfma", knob)
    r = BC.measure("c3", steps=10, depth=3, batch=128, modes=("sync",), device=0)
    print("match_no_mfma=%s c3 batch 128: sync %.0f frames/s, match stage %.4f ms" % (knob, r["sync"]["frames_per_s"], r["sync"]["stage_ms"].get("match", -1)), flush=True)
orbx.debug_set("match_no_mfma", None)
print("checked", BC.check("c3", device=0))
