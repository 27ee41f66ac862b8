#!/usr/bin/env python3
"""Phase shares of k_describe_patch per wave from a -DORBX_DESC_STAMPS build
(make -C orb_slam_tracking_amd/csrc EXTRA=-DORBX_DESC_STAMPS after touching orbx_kernels.hip): 256 frames 640x480, one stream."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orb_slam_tracking_amd as orbx
from orb_slam_tracking_amd import synth
os.environ.setdefault("ORBX_NO_SPLIT", "1")
B, W, H, cap = 256, 640, 480, 1000
frames = torch.from_numpy(synth.synth_frames(B, W, H, 1000)).cuda()
e = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"); d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
n = torch.zeros(B, dtype=torch.int32, device="cuda")
L = orbx.lib()
nw = 1 << 18
buf = np.zeros((nw, 8), np.uint32)
for it in range(3):
    if it == 2:
        torch.cuda.synchronize()
        L.orbx_diag_desc_stamps(None, -1)
    e.extract_batch_device(frames, B, W, H, W, W * H, k, d, n, cap)
torch.cuda.synchronize()
L.orbx_diag_desc_stamps(ctypes.c_void_p(buf.ctypes.data), nw)
ok = buf[:, 5] != 0
t = buf[ok, :6].astype(np.int64)
dt = (t[:, 1:] - t[:, :-1]) & 0xffffffff
names = ["window fetch (loads landed)", "IC_Angle + fastAtan2", "horizontal blur", "vertical blur", "BRIEF + output"]
tot = dt.sum(1)
print("waves %d, cycles per wave: mean %.0f median %.0f" % (ok.sum(), tot.mean(), np.median(tot)))
for i, nm in enumerate(names):
    print("  %-32s mean %8.0f median %8.0f cycles  %5.1f %%" % (nm, dt[:, i].mean(), np.median(dt[:, i]), 100 * dt[:, i].sum() / tot.sum()))
