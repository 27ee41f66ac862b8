#!/usr/bin/env python3
"""Phase shares of k_match_wide_resolve per workgroup (-DORBX_MJ_STAMPS build): BASELINE config 3 (16 frames 1920x1080, 4000 features,
window 4096 = brute force), one synchronous call on one stream."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orb_slam_tracking_amd as orbx
from orb_slam_tracking_amd import synth
os.environ.setdefault("ORBX_NO_SPLIT", "1")
B, W, H, cap, window = 16, 1920, 1080, 4000, 4096
if len(sys.argv) > 1 and sys.argv[1] == "c5":
    B, W, H, cap, window = 4, 3840, 2160, 8000, 100
frames = torch.from_numpy(synth.synth_frames(B, W, H, 77)).cuda()
e = orbx.ORBextractor(cap, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"); d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
n = torch.zeros(B, dtype=torch.int32, device="cuda"); m = torch.zeros((B // 2) * cap, dtype=torch.int32, device="cuda")
nm = torch.zeros(B // 2, dtype=torch.int32, device="cuda")
first = np.arange(0, B, 2, dtype=np.int32)
for _ in range(3):
    e.extract_match_batch_device(frames, B, W, H, W, W * H, k, d, n, first, first + 1, (0, W, 0, H), m, nm, None, window, 0.9, True, cap)
torch.cuda.synchronize()
L = orbx.lib()
nw = B // 2
st = np.zeros((nw, 16), np.uint64)
L.orbx_diag_mj_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.orbx_diag_mj_stamps(st.ctypes.data, nw) == 0
st = st.astype(np.int64)
for w in range(nw):
    s = st[w]
    print("pair %d: first sweep %d cycles, later sweeps %d (%d sweeps in all), bookkeeping %d; nmatches %d" % (
        w, s[1] - s[0] if s[8] > 1 else s[2] - s[0], s[2] - s[1] if s[8] > 1 else 0, s[8], s[3] - s[2], int(nm[w])))
