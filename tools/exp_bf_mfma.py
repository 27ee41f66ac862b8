#!/usr/bin/env python3
"""The 2000 x 2000 brute-force match (BASELINE config 5's matcher) on the matrix cores against the xor / bcnt form of the same
kernel (knob match_no_mfma), both checked against the CPU oracle; then BASELINE config 3 (its 869 x 869 all-pairs match inside)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orb_slam_tracking_amd as orbx
import bench_config as BC
for knob in (None, 1, None, 1):
    orbx.debug_set("match_no_mfma", knob)
    bf, data = BC.measure_bf(steps=20, device=0)
    print("match_no_mfma=%s: %.3f us per 2000x2000, %.3g pairs/s, nmatches %s, checked %s" % (
        knob, bf["ms_per_2000x2000"] * 1e3, bf["descriptor_pairs_per_s"], bf["nmatches"], BC.check_bf(data)), flush=True)
orbx.debug_set("match_no_mfma", None)
for knob in (None, 1):
    orbx.debug_set("match_no_mfma", knob)
    r = BC.measure("c3", steps=20, depth=3, device=0)
    print("match_no_mfma=%s c3: sync %.0f lanes %.0f frames/s, match stage %.4f ms, checked %s" % (
        knob, r["sync"]["frames_per_s"], r["lanes"]["frames_per_s"], r["sync"]["stage_ms"].get("match", -1), BC.check("c3", device=0)), flush=True)
