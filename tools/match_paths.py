#!/usr/bin/env python3
"""Diagnostics: how many pairs of a batch the parallel matcher kernels complete themselves (the diagnostic knob match_no_general leaves
the pairs they hand on to the general kernel at INT_MIN), per configuration of tools/configs_rate.py."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import orb_slam_tracking_amd as orbx  # noqa: E402
from orb_slam_tracking_amd import synth  # noqa: E402

CASES = [("640x480/1000", 640, 480, 1000, 20, 7, 64), ("752x480/2000 FAST 0/0", 752, 480, 2000, 0, 0, 64),
         ("1920x1080/4000", 1920, 1080, 4000, 20, 7, 32), ("3840x2160/8000", 3840, 2160, 8000, 20, 7, 8)]
for name, w, h, nf, ini, mn, B in CASES:
    frames = synth.synth_frames(B, w, h, seed0=77)
    d_img = torch.from_numpy(frames).cuda()
    d_k = torch.zeros(B * nf * 28, dtype=torch.uint8, device="cuda")
    d_d = torch.zeros(B * nf * 32, dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    d_m = torch.zeros((B // 2) * nf, dtype=torch.int32, device="cuda")
    d_nm = torch.zeros(B // 2, dtype=torch.int32, device="cuda")
    first = np.arange(0, B, 2, dtype=np.int32)
    ext = orbx.ORBextractor(nf, 1.2, 8, ini, mn, max_width=w, max_height=h, max_batch=B)
    out = {"case": name, "pairs": B // 2}
    for label, env in (("all kernels", {}), ("without the sequential loop", {"match_no_general": 1})):
        for k, v in env.items():
            orbx.debug_set(k, v)
        ext.extract_match_batch_device(d_img, B, w, h, w, w * h, d_k, d_d, d_n, first, first + 1, (0, w, 0, h), d_m, d_nm, None,
                                       100, 0.9, True, nf)
        for k in env:
            orbx.debug_set(k, None)
        nm = d_nm.cpu().numpy()
        out[label] = {"completed": int((nm != -2**31).sum()), "nmatches_sum": int(nm[nm != -2**31].sum())}
    kk = d_k.cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(B, nf)
    out["octave0_per_frame"] = float(np.mean([(kk[f, :int(d_n[f])]["octave"] == 0).sum() for f in range(B)]))
    print(json.dumps(out), flush=True)
    ext.close()
