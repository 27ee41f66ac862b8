#!/usr/bin/env python3
"""Rates of the BASELINE.json configurations that are not the bench line (they are parity-test cases; bench.py measures config 4's
shape): frames resident in HBM, extraction of B frames + SearchForInitialization of the B/2 consecutive pairs per call.

  c3   1920x1080, 4000 features, BF match against the previous frame: window 4096 (covers the frame), ratio 0.9, orientation check
  c5   3840x2160, 8000 features, windowed match (100) of consecutive frames -- and, timed on its own (`--bf`), the 2000 x 2000
       descriptor brute-force match (synth_desc(2000, 5): ratio + rotation-histogram filter), pairs per second against
       SURVEY 8(d)'s ops_match = 16 lane-ops per 256-bit pair (peak 4.9 T pairs/s; 3.3 T with v_bcnt's 4-cycle issue)
  c2   640x480, 1000 features (the bench shape, for comparison)

Prints one JSON line: the synchronous call, the stream-ordered call on `--depth` lanes, per-stage device times, algorithmic bytes
(5 sum(P) - P0 - P7 + 1321 N per frame) against 8 TB/s.  Under rocprofv3 use --mode sync / lanes / bf to keep every
launch of a kernel the same size (tools/prof_config.sh).  bench.py imports measure() / measure_bf() / check() for its
`other_configs` object (outside its timed regions)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

CFG = {"c2": (640, 480, 1000, 256, 100), "c3": (1920, 1080, 4000, 32, 4096), "c5": (3840, 2160, 8000, 8, 100)}


def level_sizes(w, h, nlevels=8, sf=1.2):
    out, s = [], np.float32(1.0)
    for _ in range(nlevels):
        inv = np.float32(1.0) / s
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
        s = np.float32(np.float64(s) * np.float64(np.float32(sf)))
    return out


def alg_bytes(w, h, n_kp):
    P = [a * b for a, b in level_sizes(w, h)]
    return 5 * sum(P) - P[0] - P[-1] + 1321 * n_kp


def measure(config, steps=20, depth=3, batch=0, modes=("sync", "lanes"), device=0):
    """Synchronous and stream-ordered rates of one configuration: extraction of B resident frames + SearchForInitialization of the
    B / 2 consecutive pairs per call."""
    import torch
    import orb_slam_tracking_amd as orbx
    from orb_slam_tracking_amd import synth
    w, h, nf, B, window = CFG[config]
    B = batch or B
    out = {"frame": [w, h], "nfeatures": nf, "batch": B, "window": window}
    dev = torch.device("cuda", device)
    frames = synth.synth_frames(B, w, h, seed0=77)
    d_img = torch.from_numpy(frames).to(dev)
    cap = nf
    nout = max(2, depth)
    outs = [dict(k=torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev), d=torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev),
                 n=torch.zeros(B, dtype=torch.int32, device=dev), m=torch.zeros((B // 2) * cap, dtype=torch.int32, device=dev),
                 nm=torch.zeros(B // 2, dtype=torch.int32, device=dev)) for _ in range(nout)]
    first = np.arange(0, B - 1, 2, dtype=np.int32)
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B, device=device)

    def call(o, async_):
        f = ext.extract_match_batch_device_async if async_ else ext.extract_match_batch_device
        f(d_img, B, w, h, w, w * h, o["k"], o["d"], o["n"], first, first + 1, (0, w, 0, h), o["m"], o["nm"], None, window, 0.9, True, cap)
    if "sync" in modes:
        for _ in range(3):
            call(outs[0], False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            call(outs[0], False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        # the stage times come from a second, untimed pass: the stage events cost the call 5-8 % (round 4: 10.4 k against 11.4 k
        # frames/s at 3840x2160 with them in the timed loop)
        ext.profile_enable(True)
        ext.profile_reset()
        for _ in range(steps):
            call(outs[0], False)
        torch.cuda.synchronize()
        ext.profile_enable(False)
        n_kp = float(outs[0]["n"].float().mean().item())
        ab = alg_bytes(w, h, n_kp)
        stage = {k: v[0] / steps for k, v in ext.profile_get().items()}
        out["sync"] = {"ms_per_batch": dt * 1e3, "frames_per_s": B / dt, "stage_ms": stage, "dominant_stage": max(stage, key=stage.get),
                       "mean_keypoints": n_kp, "mean_nmatches": float(outs[0]["nm"].float().mean().item()),
                       "algorithmic_bytes_per_frame": ab, "algorithmic_GBs": ab * B / dt / 1e9, "algorithmic_frac_of_8TBs": ab * B / dt / 8e12,
                       "launch": ext.debug_last_launch()}
    if "lanes" in modes and depth > 0:
        ext.set_pipeline_depth(depth)
        for k in range(3 * depth):
            call(outs[k % nout], True)
        ext.wait()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            call(outs[k % nout], True)
        ext.wait()
        dt = (time.perf_counter() - t0) / steps
        n_kp = float(outs[0]["n"].float().mean().item())
        ab = alg_bytes(w, h, n_kp)
        out["lanes"] = {"depth": depth, "ms_per_batch": dt * 1e3, "frames_per_s": B / dt, "algorithmic_GBs": ab * B / dt / 1e9,
                        "algorithmic_frac_of_8TBs": ab * B / dt / 8e12}
    ext.close()
    del d_img, outs
    torch.cuda.empty_cache()
    return out


def measure_bf(steps=20, device=0, sets=64, sync_each=False):
    """BASELINE config 5's matcher: 2000 x 2000 descriptors, window covering the frame -> every (query, train) pair is a candidate.
    sync_each: the caller waits for every call's result before the next (the reference's own use: one SearchForInitialization, its
    result, the next) instead of queueing the calls back to back."""
    import torch
    import orb_slam_tracking_amd as orbx
    from orb_slam_tracking_amd import synth
    dev = torch.device("cuda", device)
    n = 2000
    kA, dA, kB, dB = synth.synth_desc(n, 5)
    P = sets  # pairs of descriptor sets per call (the same two sets: what is timed is the matcher, not the data)
    k_all = np.concatenate([kA, kB])
    d_all = np.concatenate([dA, dB])
    d_k = torch.from_numpy(np.tile(k_all.view(np.uint8).reshape(2, n * 28), (P, 1)).reshape(-1).copy()).to(dev)
    d_d = torch.from_numpy(np.tile(d_all.reshape(2, n * 32), (P, 1)).reshape(-1).copy()).to(dev)
    d_n = torch.full((2 * P,), n, dtype=torch.int32, device=dev)
    d_m = torch.zeros(P * n, dtype=torch.int32, device=dev)
    d_nm = torch.zeros(P, dtype=torch.int32, device=dev)
    first = np.arange(0, 2 * P, 2, dtype=np.int32)
    ext = orbx.ORBextractor(2000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=1, device=device)

    def match():
        ext.match_pairs_device(first, first + 1, d_k, d_d, d_n, (0, 3840, 0, 2160), d_m, d_nm, None, 8192, 0.9, True, n)
    for _ in range(3):
        match()
    ext.profile_enable(True)
    ext.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        match()
        if sync_each:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    pairs = P * n * n
    out = {"sets_per_call": P, "descriptors": [n, n], "ms_per_call": dt * 1e3, "ms_per_2000x2000": dt * 1e3 / P,
           "descriptor_pairs_per_s": pairs / dt, "frac_of_4.9T_pairs_per_s": pairs / dt / 4.9e12, "frac_of_3.3T_pairs_per_s": pairs / dt / 3.3e12,
           # round 5: the all-pairs distances run on the matrix cores (k_match_bf_mfma): 16 v_mfma_i32_32x32x32_i8 of 32 cycles per
           # 2048 pairs = 0.25 cycles of a SIMD per pair -> 1024 SIMDs x 2.4 GHz / 0.25 = 9.8 T pairs/s
           "frac_of_9.8T_mfma_i8_pairs_per_s": pairs / dt / 9.8e12,
           "device_ms_per_call": ext.profile_get()["match"][0] / steps, "nmatches": int(d_nm[0].item()),
           "algorithmic_bytes_per_2000x2000": 48 * 2 * n + 4 * n}
    got_m = d_m[:n].cpu().numpy()
    ext.close()
    return out, (kA, dA, kB, dB, got_m, out["nmatches"])


def check(config, device=0):
    """Not timed: one pair of the configuration's frames through the HIP path against the CPU oracle (keypoint bytes, descriptor bytes,
    matches12, nmatches)."""
    import orb_slam_tracking_amd as orbx
    from orb_slam_tracking_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    w, h, nf, _, window = CFG[config]
    a, b = synth.synth_frames(2, w, h, seed0=77)
    ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2, device=device)
    fa, fb = orbx.Frame(a, 0.0, ext), orbx.Frame(b, 1.0, ext)
    nm, m12 = orbx.ORBmatcher(0.9, True).SearchForInitialization(fa, fb, window)
    oe = O.Extractor(nf, 1.2, 8, 20, 7)
    _, ka, da = oe(a)
    _, kb, db = oe(b)
    onm, om12, _ = O.match_init(ka, da, kb, db, (0, w, 0, h), window, 0.9, True)
    ok = (fa.mvKeys.tobytes() == ka.tobytes() and np.array_equal(fa.mDescriptors, da) and fb.mvKeys.tobytes() == kb.tobytes() and
          np.array_equal(fb.mDescriptors, db) and nm == onm and np.array_equal(m12, om12))
    ext.close()
    return bool(ok)


def check_bf(bf_data):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    kA, dA, kB, dB, got_m, got_nm = bf_data
    onm, om12, _ = O.match_init(kA, dA, kB, dB, (0, 3840, 0, 2160), 8192, 0.9, True)
    return bool(onm == got_nm and np.array_equal(om12, got_m[:len(om12)]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c5", choices=sorted(CFG))
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--mode", default="all", choices=("all", "sync", "lanes", "bf"))
    ap.add_argument("--bf", action="store_true", help="also time the 2000 x 2000 descriptor brute-force match (config 5)")
    ap.add_argument("--check", action="store_true", help="compare one pair of the configuration with the CPU oracle (not timed)")
    a = ap.parse_args()
    import torch  # noqa: F401  (load order: torch's HIP runtime first)
    out = {"config": a.config}
    if a.mode in ("all", "sync", "lanes"):
        out.update(measure(a.config, a.steps, a.depth, a.batch, ("sync", "lanes") if a.mode == "all" else (a.mode,)))
    if a.mode == "bf" or (a.bf and a.mode == "all"):
        out["bf_match"], bf_data = measure_bf(a.steps)
        if a.check:
            out["bf_match"]["checked"] = check_bf(bf_data)
    if a.check and a.mode in ("all", "sync", "lanes"):
        out["checked"] = check(a.config)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
