#!/usr/bin/env python3
"""Rates of the BASELINE.json configurations that are not the bench line (they are parity-test cases; bench.py measures config 4's
shape): frames resident in HBM, extraction of B frames + SearchForInitialization of the B/2 consecutive pairs per call.

  c3   1920x1080, 4000 features, BF match against the previous frame: window 4096 (covers the frame), ratio 0.9, orientation check
  c5   3840x2160, 8000 features, windowed match (100) of consecutive frames -- and, timed on its own (`--bf`), the 2000 x 2000
       descriptor brute-force match (synth_desc(2000, 5): ratio + rotation-histogram filter), pairs per second against
       SURVEY 8(d)'s ops_match = 16 lane-ops per 256-bit pair (peak 4.9 T pairs/s)
  c2   640x480, 1000 features (the bench shape, for comparison)

Prints one JSON line: the synchronous call, the stream-ordered call on `--depth` lanes, per-stage device times, algorithmic bytes
(5 sum(P) - P0 - P7 + 1321 N per frame) against 8 TB/s.  Under rocprofv3 use --mode sync / lanes / bf to keep every
launch of a kernel the same size (tools/prof_config.sh)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import orb_slam_tracking_amd as orbx  # noqa: E402
from orb_slam_tracking_amd import synth  # noqa: E402
from bench import level_sizes  # noqa: E402

CFG = {"c2": (640, 480, 1000, 256, 100), "c3": (1920, 1080, 4000, 32, 4096), "c5": (3840, 2160, 8000, 8, 100)}


def alg_bytes(w, h, n_kp):
    P = [a * b for a, b in level_sizes(w, h)]
    return 5 * sum(P) - P[0] - P[-1] + 1321 * n_kp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c5", choices=sorted(CFG))
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--mode", default="all", choices=("all", "sync", "lanes", "bf"))
    ap.add_argument("--bf", action="store_true", help="also time the 2000 x 2000 descriptor brute-force match (config 5)")
    a = ap.parse_args()
    w, h, nf, B, window = CFG[a.config]
    B = a.batch or B
    out = {"config": a.config, "frame": [w, h], "nfeatures": nf, "batch": B, "window": window}
    if a.mode in ("all", "sync", "lanes"):
        frames = synth.synth_frames(B, w, h, seed0=77)
        d_img = torch.from_numpy(frames).cuda()
        cap = nf
        nout = max(2, a.depth)
        outs = [dict(k=torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"), d=torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda"),
                     n=torch.zeros(B, dtype=torch.int32, device="cuda"), m=torch.zeros((B // 2) * cap, dtype=torch.int32, device="cuda"),
                     nm=torch.zeros(B // 2, dtype=torch.int32, device="cuda")) for _ in range(nout)]
        first = np.arange(0, B - 1, 2, dtype=np.int32)
        ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)

        def call(o, async_):
            f = ext.extract_match_batch_device_async if async_ else ext.extract_match_batch_device
            f(d_img, B, w, h, w, w * h, o["k"], o["d"], o["n"], first, first + 1, (0, w, 0, h), o["m"], o["nm"], None, window, 0.9, True, cap)
        if a.mode in ("all", "sync"):
            for _ in range(3):
                call(outs[0], False)
            ext.profile_enable(True)
            ext.profile_reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                call(outs[0], False)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.steps
            ext.profile_enable(False)
            n_kp = float(outs[0]["n"].float().mean().item())
            ab = alg_bytes(w, h, n_kp)
            out["sync"] = {"ms_per_batch": dt * 1e3, "frames_per_s": B / dt, "stage_ms": {k: v[0] / a.steps for k, v in ext.profile_get().items()},
                           "mean_keypoints": n_kp, "mean_nmatches": float(outs[0]["nm"].float().mean().item()),
                           "algorithmic_bytes_per_frame": ab, "algorithmic_GBs": ab * B / dt / 1e9, "algorithmic_frac_of_8TBs": ab * B / dt / 8e12,
                           "launch": ext.debug_last_launch()}
        if a.mode in ("all", "lanes") and a.depth > 0:
            ext.set_pipeline_depth(a.depth)
            for k in range(3 * a.depth):
                call(outs[k % nout], True)
            ext.wait()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(a.steps):
                call(outs[k % nout], True)
            ext.wait()
            dt = (time.perf_counter() - t0) / a.steps
            n_kp = float(outs[0]["n"].float().mean().item())
            ab = alg_bytes(w, h, n_kp)
            out["lanes"] = {"depth": a.depth, "ms_per_batch": dt * 1e3, "frames_per_s": B / dt, "algorithmic_GBs": ab * B / dt / 1e9,
                            "algorithmic_frac_of_8TBs": ab * B / dt / 8e12}
        ext.close()
    if a.mode == "bf" or (a.bf and a.mode == "all"):
        # BASELINE config 5's matcher: 2000 x 2000 descriptors, window covering the frame -> every (query, train) pair is a candidate
        n = 2000
        kA, dA, kB, dB = synth.synth_desc(n, 5)
        P = 64  # pairs of descriptor sets per call (the same two sets: what is timed is the matcher, not the data)
        k_all = np.concatenate([kA, kB])
        d_all = np.concatenate([dA, dB])
        d_k = torch.from_numpy(np.tile(k_all.view(np.uint8).reshape(2, n * 28), (P, 1)).reshape(-1).copy()).cuda()
        d_d = torch.from_numpy(np.tile(d_all.reshape(2, n * 32), (P, 1)).reshape(-1).copy()).cuda()
        d_n = torch.full((2 * P,), n, dtype=torch.int32, device="cuda")
        d_m = torch.zeros(P * n, dtype=torch.int32, device="cuda")
        d_nm = torch.zeros(P, dtype=torch.int32, device="cuda")
        first = np.arange(0, 2 * P, 2, dtype=np.int32)
        ext = orbx.ORBextractor(2000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=1)

        def match():
            ext.match_pairs_device(first, first + 1, d_k, d_d, d_n, (0, 3840, 0, 2160), d_m, d_nm, None, 8192, 0.9, True, n)
        for _ in range(3):
            match()
        ext.profile_enable(True)
        ext.profile_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            match()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        pairs = P * n * n
        out["bf_match"] = {"sets_per_call": P, "descriptors": [n, n], "ms_per_call": dt * 1e3, "ms_per_2000x2000": dt * 1e3 / P,
                           "descriptor_pairs_per_s": pairs / dt, "frac_of_4.9T_pairs_per_s": pairs / dt / 4.9e12,
                           "device_ms_per_call": ext.profile_get()["match"][0] / a.steps, "nmatches": int(d_nm[0].item()),
                           "algorithmic_bytes_per_2000x2000": 48 * 2 * n + 4 * n}
        ext.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
