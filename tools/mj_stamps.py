#!/usr/bin/env python3
"""Phase shares of k_match_jacobi per workgroup from a -DORBX_MJ_STAMPS build (make -C orb_slam_tracking_amd/csrc EXTRA=-DORBX_MJ_STAMPS
after touching orbx_kernels.hip): B frames 640x480 (default 256 -> 128 pairs), one synchronous call."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orb_slam_tracking_amd as orbx
from orb_slam_tracking_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H, cap = 640, 480, 1000
os.environ.setdefault("ORBX_NO_SPLIT", "1")
frames = torch.from_numpy(synth.synth_frames(B, W, H, 1000)).cuda()
e = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"); d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
n = torch.zeros(B, dtype=torch.int32, device="cuda"); m = torch.zeros((B // 2) * cap, dtype=torch.int32, device="cuda")
nm = torch.zeros(B // 2, dtype=torch.int32, device="cuda")
first = np.arange(0, B, 2, dtype=np.int32)
for _ in range(3):
    e.extract_match_batch_device(frames, B, W, H, W, W * H, k, d, n, first, first + 1, (0, W, 0, H), m, nm, None, 100, 0.9, True, cap)
torch.cuda.synchronize()
L = orbx.lib()
nw = B // 2
st = np.zeros((nw, 16), np.uint64)
L.orbx_diag_mj_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.orbx_diag_mj_stamps(st.ctypes.data, nw) == 0
st = st.astype(np.int64)
names = ["stage trains", "order queries", "load queries", "windows + lists", "sweeps", "bookkeeping"]
dt = st[:, 1:7] - st[:, 0:6]
tot = st[:, 6] - st[:, 0]
print("workgroups %d, cycles per workgroup: mean %.0f; sweeps mean %.1f max %d" % (nw, tot.mean(), st[:, 8].mean(), st[:, 8].max()))
print("  of the lists phase: window tests %.0f cycles, distances of the masked trains %.0f" % ((st[:, 7] - st[:, 3]).mean(), (st[:, 4] - st[:, 7]).mean()))
for i, nmn in enumerate(names):
    print("  %-18s mean %8.0f cycles  %5.1f %%" % (nmn, dt[:, i].mean(), 100 * dt[:, i].sum() / tot.sum()))
