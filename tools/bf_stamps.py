#!/usr/bin/env python3
"""Phase shares of k_match_bf_mfma from a -DORBX_BF_STAMPS build (make -C orb_slam_tracking_amd/csrc VARIANT=bfstamps EXTRA=-DORBX_BF_STAMPS;
ORBX_LIB=.../liborbx_bfstamps.so): s_memtime deltas per wave and tile phase, summed over the wave's tiles; 2000 x 2000 descriptor sets,
[sets] pairs of sets per call (default 64: two workgroups per CU)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import orb_slam_tracking_amd as orbx
import bench_config as BC
sets = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = orbx.lib()
bf, data = BC.measure_bf(steps=3, device=0, sets=sets)
torch.cuda.synchronize()
nw = 4096
buf = np.zeros((nw, 8), np.uint32)
L.orbx_diag_bf_stamps(ctypes.c_void_p(buf.ctypes.data), nw)
ok = buf[:, 5] > 0
v = buf[ok, :5].astype(np.float64) / buf[ok, 5:6]
names = ["LDS reads + MFMA issue", "drain + reduction", "appends", "staging (expand, loads)", "barrier"]
print("%d sets per call: %.3f us per pair of sets; waves %d, tiles %d; ticks per tile and wave: %.0f" % (sets, bf["ms_per_2000x2000"] * 1e3, ok.sum(), int(buf[ok, 5].max()), v.sum(1).mean()))
for i, nm in enumerate(names):
    print("  %-28s mean %7.0f  median %7.0f  %5.1f %%" % (nm, v[:, i].mean(), np.median(v[:, i]), 100 * v[:, i].sum() / v.sum()))
