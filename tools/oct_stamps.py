#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the selection kernel (needs liborbx.so built with -DORBX_OCT_STAMPS)."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import orb_slam_tracking_amd as orbx
from orb_slam_tracking_amd import synth
# usage: oct_stamps.py [width height nfeatures iniThFAST minThFAST batch]
a = [int(x) for x in sys.argv[1:7]] + [640, 480, 1000, 20, 7, 32][len(sys.argv) - 1:]
w, h, cap, ini, mn, B = a
frames = synth.synth_frames(B, w, h, 1000)
e = orbx.ORBextractor(cap, 1.2, 8, ini, mn, max_width=w, max_height=h, max_batch=B)
d_img = torch.from_numpy(frames).cuda()
d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"); d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
os.environ["ORBX_NO_SPLIT"] = "1"
for _ in range(2):
    e.extract_batch_device(d_img, B, w, h, w, w * h, d_k, d_d, d_n, cap)
L = orbx.lib()
nb = B * 8
st = np.zeros((nb, 16), np.uint64)
L.orbx_diag_oct_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.orbx_diag_oct_stamps(st.ctypes.data, nb) == 0
names = ["codes", "keysort", "div/alone", "passloop", "nodelist", "nodesort", "noderec", "(a)sized", "(b)introsort", "(b2)ranksort", "(c)children", "(d)cut", "(e)create", "emit"]
st = st.astype(np.int64)
big = bool(os.environ.get("OCT_BIG"))  # the units ran on k_octree_big (levels with large units): its stamps 0, 3 .. 15
if big:
    names = ["records:load+scan", "records:neighbours", "records:histogram", "passloop", "nodelist", "nodesort", "noderec", "(a)sized", "(b)std::sort", "-", "(c)children", "(d)cut", "(e)create", "emit"]
for lvl in range(8):
    s = st[lvl * B:(lvl + 1) * B]  # workgroups are dispatched level-major: block = level * nFrames + frame
    d = [s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3], s[:, 5] - s[:, 4], s[:, 6] - s[:, 5], s[:, 7] - s[:, 6],
         s[:, 8], s[:, 9], s[:, 10], s[:, 11], s[:, 12], s[:, 13], s[:, 15] - s[:, 14]]
    if big:
        d = d[0:3] + d[3:7] + d[7:13] + [d[13]]
    tot = s[:, 15] - s[:, 0]
    print("level %d total %.0f cyc:" % (lvl, tot.mean()), " ".join("%s=%.0f" % (n, x.mean()) for n, x in zip(names, d)))
if os.environ.get("OCT_TIMELINE"):  # when the units' first phase starts and ends and when the units end, from the launch's first stamp
    # (s_memtime counts per XCD: workgroup b runs on XCD b % 8, and each XCD's stamps are taken relative to its own first one)
    rel = st[:nb].copy()
    for x in range(8):
        rows = np.arange(nb) % 8 == x
        t0 = rel[rows, 0][rel[rows, 0] > 0].min()
        rel[rows] -= t0
    for lvl in range(8):
        s = rel[lvl * B:(lvl + 1) * B]
        print("level %d: step 1 starts at %6.0f .. %6.0f (mean %6.0f), ends at %6.0f .. %6.0f (mean %6.0f); unit ends at %6.0f .. %6.0f (mean %6.0f)" % (
            lvl, s[:, 0].min(), s[:, 0].max(), s[:, 0].mean(), s[:, 1].min(), s[:, 1].max(), s[:, 1].mean(), s[:, 15].min(), s[:, 15].max(), s[:, 15].mean()))

if big:  # where the buckets' waves of a unit ran (counts per XCD) and where its k_octree_big workgroup did
    xc = np.zeros((nb, 9), np.uint32)
    L.orbx_diag_oct_xcc.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert L.orbx_diag_oct_xcc(xc.ctypes.data, nb) == 0
    same = sum(int(r[:8].argmax() == r[8]) for r in xc)
    print("XCD placement: %d of %d units have their bucket waves on the XCD of their k_octree_big workgroup; first units: %s" % (same, nb, xc[:6].tolist()))
if big:  # the std::sort replay's phases (summed over its recursion levels), level-0 units
    rp = np.zeros((nb, 8), np.uint64)
    L.orbx_diag_oct_replay.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert L.orbx_diag_oct_replay(rp.ctypes.data, nb) == 0
    rp = rp.astype(np.int64)[:B]
    print("std::sort replay, level-0 units: %d recursion levels; cycles per unit: median of three %.0f, stop flags + scan %.0f, prefix arrays %.0f, scatter %.0f, swaps + cut %.0f, tail %.0f, in-range ranks %.0f" % (
        rp[:, 7].mean(), rp[:, 0].mean(), rp[:, 1].mean(), rp[:, 2].mean(), rp[:, 3].mean(), rp[:, 4].mean(), rp[:, 5].mean(), rp[:, 6].mean()))
