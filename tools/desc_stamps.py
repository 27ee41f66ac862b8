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
if os.environ.get("ORBX_DESC_LOOP"):  # k_describe_loop: stamps of the wave's second keypoint (steady state of the prefetch loop)
    ok = (buf[:, 5] != 0) & (buf[:, 3] != 0)
    t = buf[ok].astype(np.int64)
    d = lambda a, b: (t[:, a] - t[:, b]) & 0xffffffff
    print("waves %d (4 keypoints each); wave lifetime mean %.0f cycles" % (ok.sum(), d(5, 6).mean()))
    for nm, v in (("prologue (tables, first window issue, keypoint 0)", d(0, 6)), ("wait for the window (vmcnt)", d(1, 0)),
                  ("edge fix-up + stores + DMA issue of the next window", d(2, 1)), ("compute (IC, blur, BRIEF)", d(3, 2))):
        print("  %-52s mean %8.0f median %8.0f cycles" % (nm, v.mean(), np.median(v)))
    sys.exit(0)
ok = buf[:, 5] != 0
t = buf[ok, :6].astype(np.int64)
dt = (t[:, 1:] - t[:, :-1]) & 0xffffffff
names = ["window fetch (loads landed)", "IC_Angle + fastAtan2", "horizontal blur", "vertical blur", "BRIEF + output"]
tot = dt.sum(1)
print("waves %d, cycles per wave: mean %.0f median %.0f" % (ok.sum(), tot.mean(), np.median(tot)))
for i, nm in enumerate(names):
    print("  %-32s mean %8.0f median %8.0f cycles  %5.1f %%" % (nm, dt[:, i].mean(), np.median(dt[:, i]), 100 * dt[:, i].sum() / tot.sum()))
