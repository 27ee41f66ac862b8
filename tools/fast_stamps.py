#!/usr/bin/env python3
"""Phase shares of k_fast_wave from a -DORBX_FAST_STAMPS build (make -C orb_slam_tracking_amd/csrc clean && make EXTRA=-DORBX_FAST_STAMPS):
s_memtime deltas per wave, summed by the kernel; 256 frames 640x480 on one stream."""
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
nw = 256 * 640
buf = np.zeros((nw, 12), np.uint32)
for it in range(3):
    if it == 2:
        torch.cuda.synchronize()
        L.orbx_diag_fast_stamps(None, -1)
    e.extract_batch_device(frames, B, W, H, W, W * H, k, d, n, cap)
L.orbx_diag_fast_stamps(ctypes.c_void_p(buf.ctypes.data), nw)
ok = buf[:, :4].sum(1) > 0
live = buf[ok, :4].astype(np.float64)
names = ["prologue + staging (loads landed)", "quick reject + full-wave evaluations", "last evaluation", "NMS + output"]
tot = live.sum()
print("waves %d (4 cells each), cycles per wave: mean %.0f, median %.0f" % (len(live), live.sum(1).mean(), np.median(live.sum(1))))
for i, nm in enumerate(names):
    print("  %-40s mean %8.0f  median %8.0f cycles  %5.1f %%" % (nm, live[:, i].mean(), np.median(live[:, i]), 100 * live[:, i].sum() / tot))
# residency: waves per CU over the kernel's span, from HW_ID (wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13) and XCC_ID
hw, xcc = buf[ok, 4], buf[ok, 5] & 0xf
start, end = buf[ok, 6].astype(np.int64), buf[ok, 7].astype(np.int64)
dur = (end - start) & 0xffffffff
cu = ((hw >> 8) & 0xf).astype(np.int64); sh = ((hw >> 12) & 1).astype(np.int64); se = ((hw >> 13) & 7).astype(np.int64)
key = ((xcc.astype(np.int64) * 8 + se) * 2 + sh) * 16 + cu
res = []
for kk in np.unique(key):
    m = key == kk
    s0 = start[m]; ref = s0[0]
    rs = ((s0 - ref + (1 << 31)) & 0xffffffff) - (1 << 31)   # starts relative to one wave of the same CU (same counter)
    re_ = rs + dur[m]
    span = re_.max() - rs.min()
    ev = np.concatenate([np.stack([rs, np.ones_like(rs)], 1), np.stack([re_, -np.ones_like(rs)], 1)])
    ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
    res.append((dur[m].sum() / span, span, m.sum(), np.cumsum(ev[:, 1]).max()))
res = np.array(res)
print("CUs seen: %d; per CU: resident waves mean %.1f (min %.1f, max %.1f), peak concurrency mean %.1f max %d; span mean %.0f ticks; waves per CU mean %.0f"
      % (len(res), res[:, 0].mean(), res[:, 0].min(), res[:, 0].max(), res[:, 3].mean(), res[:, 3].max(), res[:, 1].mean(), res[:, 2].mean()))
print("simd ids:", np.bincount(((hw >> 4) & 3).astype(np.int64)), " xcc ids:", np.bincount(xcc.astype(np.int64)))

# chip-wide timeline from s_memrealtime (constant 100 MHz)
r0, r1 = buf[ok, 8].astype(np.int64), buf[ok, 9].astype(np.int64)
ref = r0[0]
a0 = ((r0 - ref + (1 << 31)) & 0xffffffff) - (1 << 31)
a1 = a0 + ((r1 - r0) & 0xffffffff)
t0 = a0.min()
spanr = a1.max() - t0
print("realtime: kernel span %.1f us; mean wave %.1f us; shader clock from ticks %.2f GHz" % (spanr / 100.0, (a1 - a0).mean() / 100.0, dur.mean() / ((a1 - a0).mean() / 100.0) / 1000.0))
bins = 20
edges = np.linspace(0, spanr, bins + 1)
occ = []
for b in range(bins):
    lo, hi = edges[b], edges[b + 1]
    ov = np.clip(np.minimum(a1 - t0, hi) - np.maximum(a0 - t0, lo), 0, None).sum() / (hi - lo)
    occ.append(ov / 256.0)
print("resident waves per CU over the kernel (20 slices):", " ".join("%.1f" % v for v in occ))
for x in range(8):
    m = xcc == x
    if m.any():
        print("xcc %d: waves %d, mean wave %.1f us (ticks %.0f), first start %.1f us, last end %.1f us, phases mean ticks %s"
              % (x, m.sum(), (a1 - a0)[m].mean() / 100.0, dur[m].mean(), (a0[m].min() - t0) / 100.0, (a1[m].max() - t0) / 100.0,
                 np.round(live[m].mean(0)).astype(int).tolist()))
