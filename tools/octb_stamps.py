#!/usr/bin/env python3
"""Diagnostic: where a bucket wave of k_octree_buckets spends its time (needs liborbx built with -DORBX_OCT_STAMPS: ORBX_LIB)."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orb_slam_tracking_amd as orbx
from orb_slam_tracking_amd import synth
a = [int(x) for x in sys.argv[1:7]] + [3840, 2160, 8000, 20, 7, 4][len(sys.argv) - 1:]
w, h, cap, ini, mn, B = a
frames = synth.synth_frames(B, w, h, 1000)
e = orbx.ORBextractor(cap, 1.2, 8, ini, mn, max_width=w, max_height=h, max_batch=B)
d_img = torch.from_numpy(frames).cuda()
d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"); d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
os.environ["ORBX_NO_SPLIT"] = "1"
for _ in range(3):
    e.extract_batch_device(d_img, B, w, h, w, w * h, d_k, d_d, d_n, cap)
L = orbx.lib()
nw = 65536
st = np.zeros((nw, 8), np.uint64)
L.orbx_diag_octb_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.orbx_diag_octb_stamps(st.ctypes.data, nw) == 0
st = st.astype(np.int64)
ok = st[:, 6] > 0  # waves that had a bucket and got through the gather
s = st[ok]
n = s[:, 6] - 1
print("%d bucket waves; keys per bucket mean %.0f, median %.0f, max %d; raw candidates listed per bucket mean %.0f" % (len(s), n.mean(), np.median(n), n.max(), s[:, 7].mean()))
names = ["setup + cell prefix", "gather", "sort", "output"]
for i, nm in enumerate(names):
    d = s[:, i + 1] - s[:, i]
    print("  %-20s mean %7.0f cycles, median %7.0f, p90 %7.0f" % (nm, d.mean(), np.median(d), np.percentile(d, 90)))
life = s[:, 4] - s[:, 0]
print("  wave lifetime        mean %7.0f cycles, median %7.0f, p90 %7.0f" % (life.mean(), np.median(life), np.percentile(life, 90)))
# s_memtime counts per XCD: waves of workgroup g run on XCD g % 8; span of the launch on each XCD
wg = np.arange(nw)[ok] // 4
for x in range(8):
    r = (wg % 8) == x
    if r.any():
        t0 = s[r, 0].min()
        print("  XCD %d: %5d waves, launch span %7d cycles, first start .. last start %7d, waves in flight at once (mean) %.1f" % (
            x, r.sum(), s[r, 4].max() - t0, s[r, 0].max() - t0, life[r].sum() / max(1, s[r, 4].max() - t0)))
