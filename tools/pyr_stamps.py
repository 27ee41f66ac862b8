#!/usr/bin/env python3
"""Where a workgroup of k_pyramid_bands spends its time, from a -DORBX_PYR_STAMPS build
(make -C orb_slam_tracking_amd/csrc clean && make EXTRA=-DORBX_PYR_STAMPS): s_memtime per level (rows, barrier) per
workgroup, s_memrealtime span; 256 frames 640x480 on one stream."""
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
K = int(os.environ.get("ORBX_PYR_BANDS", "8"))
nw = B * K
buf = np.zeros((nw, 40), np.uint32)
for it in range(3):
    if it == 2:
        torch.cuda.synchronize()
        L.orbx_diag_pyr_stamps(None, -1)
    e.extract_batch_device(frames, B, W, H, W, W * H, k, d, n, cap)
torch.cuda.synchronize()
L.orbx_diag_pyr_stamps(ctypes.c_void_p(buf.ctypes.data), nw)
t = buf[:, :16].astype(np.int64)
dt = (t[:, 1:] - t[:, :-1]) & 0xffffffff
names = ["prologue (tables of level 1 + barrier)"]
for l in range(1, 8):
    names += ["level %d rows" % l, "level %d store + barrier" % l]
tot = dt[:, :15].sum(1)
print("workgroups %d, cycles per workgroup: mean %.0f median %.0f (%.1f us at 2.4 GHz)" % (nw, tot.mean(), np.median(tot), tot.mean() / 2400))
for i, nm in enumerate(names):
    print("  %-42s mean %8.0f median %8.0f cycles  %5.1f %%" % (nm, dt[:, i].mean(), np.median(dt[:, i]), 100 * dt[:, i].sum() / tot.sum()))
r0, r1 = buf[:, 36].astype(np.int64), buf[:, 37].astype(np.int64)
ref = r0[0]
a0 = ((r0 - ref + (1 << 31)) & 0xffffffff) - (1 << 31)
a1 = a0 + ((r1 - r0) & 0xffffffff)
span = a1.max() - a0.min()
print("realtime: kernel span %.1f us; workgroup mean %.1f us; mean workgroups in flight %.0f (%.1f per CU)" %
      (span / 100.0, (a1 - a0).mean() / 100.0, (a1 - a0).sum() / span, (a1 - a0).sum() / span / 256))
# timeline of starts
edges = np.linspace(a0.min(), a1.max(), 11)
print("starts per tenth of the span:", np.histogram(a0, edges)[0].tolist())
print("ends per tenth of the span:  ", np.histogram(a1, edges)[0].tolist())
hw, xcc = buf[:, 38], buf[:, 39] & 0xf
cu = ((hw >> 8) & 0xf).astype(np.int64); sh = ((hw >> 12) & 1).astype(np.int64); se = ((hw >> 13) & 7).astype(np.int64)
key = ((xcc.astype(np.int64) * 8 + se) * 2 + sh) * 16 + cu
cnt = np.bincount(key)
cnt = cnt[cnt > 0]
print("CUs seen %d; workgroups per CU min %d mean %.1f max %d; per XCC %s" % (len(cnt), cnt.min(), cnt.mean(), cnt.max(), np.bincount(xcc.astype(np.int64)).tolist()))
