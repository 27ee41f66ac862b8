#!/usr/bin/env python3
"""Config 4's per-GPU share (32 frames 640x480 + 16 pairs per call) under launch-shape knobs: which pyramid kernel, one stream or two
halves.  usage: exp_c4_small.py [batch]   (prints synchronous / four-lane frames/s and the synchronous stage times per variant)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa
import orb_slam_tracking_amd as orbx
import bench_config as BC
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
BM = int(sys.argv[2]) if len(sys.argv) > 2 else 16
variants = [("default", {}), ("bands from %d frames" % BM, {"bands_min_frames": BM}), ("bands from %d, 2 strips" % BM, {"bands_min_frames": BM, "pyr_strips": 2}),
            ("one stream (no split)", {"no_split": 1}), ("one stream + bands from %d" % BM, {"no_split": 1, "bands_min_frames": BM}), ("default again", {})]
if len(sys.argv) > 3:
    variants = variants[:2]
for name, knobs in variants:
    for k, v in knobs.items():
        orbx.debug_set(k, v)
    r = BC.measure("c2", steps=300, depth=4, batch=B)
    for k in knobs:
        orbx.debug_set(k, None)
    st = r["sync"]["stage_ms"]
    print("B %3d %-30s sync %7.0f  lanes %7.0f frames/s | sync stage ms: pyr %.3f fast %.3f sel %.3f desc %.3f match %.3f | launch %s" % (
        B, name, r["sync"]["frames_per_s"], r["lanes"]["frames_per_s"], st["pyramid"], st["fast"], st["select"], st["describe"], st["match"],
        r["sync"]["launch"]), flush=True)
