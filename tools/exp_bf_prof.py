#!/usr/bin/env python3
"""Kernel averages of the 2000 x 2000 brute-force match (rocprofv3 --kernel-trace --stats -- python3 tools/exp_bf_prof.py [0|1]):
argument 1 sets the knob match_no_mfma (the xor / bcnt form)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orb_slam_tracking_amd as orbx
import bench_config as BC
if len(sys.argv) > 1 and sys.argv[1] == "1":
    orbx.debug_set("match_no_mfma", 1)
bf, data = BC.measure_bf(steps=20, device=0)
print("%.3f us per 2000x2000" % (bf["ms_per_2000x2000"] * 1e3))
