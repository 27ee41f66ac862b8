#!/usr/bin/env python3
"""Randomised parity campaign (longer than the test suite allows): random frame sizes, extractor parameters, image
content (synthetic scenes, uniform noise, smooth gradients with sparse corners), batch sizes and matcher settings; the
device path (host API, batched device API, fused extract + match, stream-ordered call) against the CPU oracle, bit for
bit.  usage: fuzz_parity.py [trials] [seed] [mixed|big|batched|stateful]  (stateful: one context, many different calls; big: up to 4000 x 2200 and 10000 features; batched:
33 .. 80 frames of 4-aligned width, i.e. the banded pyramid and the two stream pipelines).  Prints one line per trial and a
summary; exits non-zero on a mismatch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import orb_slam_tracking_amd as orbx  # noqa: E402
from orb_slam_tracking_amd import synth  # noqa: E402
import oracle_lib as O  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
mode = sys.argv[3] if len(sys.argv) > 3 else "mixed"
KP = orbx.KEYPOINT_DTYPE


def images(kind, B, w, h, seed):
    if kind == "synth":
        return synth.synth_frames(B, w, h, seed)
    r = np.random.default_rng(seed)
    if kind == "noise":  # every pixel a candidate: the largest selection units
        return r.integers(0, 256, (B, h, w), dtype=np.uint8)
    if kind == "pairs_noise":  # frame 2k + 1 = frame 2k shifted by (3, 2) + small noise: many matches, dense windows
        a = r.integers(0, 256, ((B + 1) // 2, h, w), dtype=np.uint8)
        out = np.empty((B, h, w), np.uint8)
        out[0::2] = a[:len(out[0::2])]
        sh = np.roll(a, (2, 3), axis=(1, 2)).astype(np.int16) + r.integers(-2, 3, a.shape)
        out[1::2] = np.clip(sh, 0, 255).astype(np.uint8)[:len(out[1::2])]
        return out
    g = (np.add.outer(np.arange(h), np.arange(w)) * 96 // (w + h)).astype(np.uint8)  # "sparse": few corners
    out = np.repeat(g[None], B, 0).copy()
    for f in range(B):
        for _ in range(int(r.integers(0, 12))):
            y, x = int(r.integers(0, h - 8)), int(r.integers(0, w - 8))
            out[f, y:y + 8, x:x + 8] = int(r.integers(0, 256))
    return out


def same(k, d, ko, do):
    return len(k) == len(ko) and k.tobytes() == np.ascontiguousarray(ko, KP).tobytes() and np.array_equal(d, do)


def stateful(trials):
    """One context, many calls: frame size, batch size, image content, pair list and call form change from call to call
    (geometry switches, selection-instance hint, cached pair list, wide-matcher expectation, batches in flight)."""
    bad = 0
    t0 = time.time()
    for c in range(trials):
        nlev = int(rng.integers(1, 7))
        params = (int(rng.choice([300, 1000, 2500])), float(rng.choice([1.2, 1.3])) if nlev > 1 else 1.0, nlev,
                  int(rng.integers(5, 30)), int(rng.integers(0, 6)))
        MW, MH, MB = 900, 700, 36
        e = orbx.ORBextractor(*params, max_width=MW, max_height=MH, max_batch=MB)
        depth = int(rng.choice([0, 0, 2, 3]))   # pipeline lanes for the stream-ordered calls of this context (two sets in flight at most)
        if depth:
            e.set_pipeline_depth(depth)
        libm = int(rng.integers(0, 2))           # the libm reading of this context (round 5), the oracle set to the same
        e.set_libm_variant(libm)
        O.set_libm_variant(libm)
        oe = O.Extractor(*params)
        cap = params[0] + 64
        sets = [dict(k=torch.zeros(MB * cap * 28, dtype=torch.uint8, device="cuda"), d=torch.zeros(MB * cap * 32, dtype=torch.uint8, device="cuda"),
                     n=torch.zeros(MB, dtype=torch.int32, device="cuda"), m=torch.zeros((MB // 2) * cap, dtype=torch.int32, device="cuda"),
                     nm=torch.zeros(MB // 2, dtype=torch.int32, device="cuda"), st=torch.zeros(MB // 2 * 3, dtype=torch.int32, device="cuda"))
                for _ in range(2)]
        # the host-frame call (round 5) writes into page-locked host arrays of the same layout
        hsets = [dict(k=torch.zeros(MB * cap * 28, dtype=torch.uint8).pin_memory(), d=torch.zeros(MB * cap * 32, dtype=torch.uint8).pin_memory(),
                      n=torch.zeros(MB, dtype=torch.int32).pin_memory(), m=torch.zeros((MB // 2) * cap, dtype=torch.int32).pin_memory(),
                      nm=torch.zeros(MB // 2, dtype=torch.int32).pin_memory(), st=torch.zeros(MB // 2 * 3, dtype=torch.int32).pin_memory())
                 for _ in range(2)]
        pend = []  # (set index, frames, w, h, B, first, second, win, ratio, ori)

        def verify(item):
            si, fr, w, h, B, first, second, win, ratio, ori = item[:10]
            o = hsets[si] if len(item) > 10 and item[10] else sets[si]
            n = o["n"].cpu().numpy()
            kk = o["k"].cpu().numpy().view(KP).reshape(-1, cap)
            dd = o["d"].cpu().numpy().reshape(-1, cap, 32)
            mm = o["m"].cpu().numpy().reshape(-1, cap)
            nm = o["nm"].cpu().numpy()
            st = o["st"].cpu().numpy().reshape(-1, 3)
            ora = [oe(f, cap=cap) for f in fr]
            ok = True
            for f_ in range(B):
                ok &= n[f_] == len(ora[f_][1]) and same(kk[f_, :n[f_]], dd[f_, :n[f_]], ora[f_][1], ora[f_][2])
            for p in range(len(first)):
                a, b = ora[first[p]], ora[second[p]]
                onm, om12, ost = O.match_init(a[1], a[2], b[1], b[2], (0, w, 0, h), win, ratio, ori)
                ok &= nm[p] == onm and np.array_equal(mm[p, :len(om12)], om12) and st[p].tolist() == ost.tolist()
            return ok
        ok = True
        ncalls = 0
        for call in range(14):
            w = int(rng.integers(60, MW // 4)) * 4 if rng.uniform() < 0.6 else int(rng.integers(240, MW))
            h = int(rng.integers(200, MH))
            while min(w, h) / params[1] ** (params[2] - 1) < 80:
                w, h = w + 40, h + 40
            w, h = min(w, MW), min(h, MH)
            B = int(rng.choice([2, 4, 16, 33, 36]))
            kind = str(rng.choice(["synth", "synth", "pairs_noise", "sparse"]))
            fr = images(kind, B, w, h, 7000 + 100 * c + call)
            form = int(rng.integers(0, 3))
            first = np.arange(0, B - 1, 2, dtype=np.int32)
            second = first + 1
            if form == 1:
                first, second = second.copy(), first.copy()
            elif form == 2:
                first, second = first[: max(1, len(first) // 2)], second[: max(1, len(first) // 2)]
            win = int(rng.choice([50, 100, 4096]))
            ratio = float(rng.choice([0.9, 0.7]))
            ori = bool(rng.integers(0, 2))
            use_async = bool(rng.integers(0, 2))
            use_host = bool(rng.integers(0, 4) == 0)  # every fourth call: frames from and results to host memory, stream-ordered
            if use_host:
                use_async = True
            si = call & 1
            if len(pend) == 2 or (pend and not use_async) or any(it[0] == si for it in pend):  # check before the set is reused
                e.wait()
                for it in pend:
                    ok &= verify(it)
                pend = []
            d_img = torch.from_numpy(fr).pin_memory() if use_host else torch.from_numpy(fr).cuda()
            o = hsets[si] if use_host else sets[si]
            f = e.extract_match_batch_host_async if use_host else (e.extract_match_batch_device_async if use_async else e.extract_match_batch_device)
            try:
                f(d_img, B, w, h, w, w * h, o["k"], o["d"], o["n"], first, second, (0, w, 0, h), o["m"], o["nm"], o["st"], win, ratio, ori, cap)
            except orbx.OrbxError as err:
                if err.code == orbx.E_TOOSMALL:
                    continue
                raise
            ncalls += 1
            item = (si, fr, w, h, B, first, second, win, ratio, ori, use_host)
            if use_async:
                pend.append(item)
                item[1].flags.writeable = False
                keep = d_img  # noqa: F841  (the frames must outlive the batch)
                sets[si]["img"] = d_img
            else:
                ok &= verify(item)
        e.wait()
        for it in pend:
            ok &= verify(it)
        print("context %d: params=%r, depth %d, libm %d, %d calls ->" % (c, params, depth, libm, ncalls), "ok" if ok else "MISMATCH", flush=True)
        bad += not ok
        e.close()
        O.set_libm_variant(O.LIBM_DEFAULT)
    print("FUZZ %s: %d contexts (stateful), %d mismatching, %.0f s" % ("OK" if bad == 0 else "FAILED", trials, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if mode == "stateful":
    stateful(trials)

bad = skipped = 0
t_start = time.time()
for t in range(trials):
    w = int(rng.integers(80, 1300))
    h = int(rng.integers(80, 900))
    if rng.uniform() < 0.4:
        w &= ~3
    nlev = int(rng.integers(1, 9))
    sf = float(rng.choice([1.1, 1.2, 1.25, 1.5, 2.0])) if nlev > 1 else float(rng.choice([1.0, 1.2]))
    while nlev > 1 and min(w, h) / sf ** (nlev - 1) < 75:  # (mostly) keep the last level wider than one FAST cell
        nlev -= 1
    nf = int(rng.choice([30, 200, 500, 1000, 2000, 5000]))
    ini = int(rng.integers(0, 40))
    mn = int(rng.integers(0, ini + 1))
    B = int(rng.choice([1, 2, 3, 6, 17, 34]))
    kind = str(rng.choice(["synth", "synth", "noise", "pairs_noise", "sparse"]))
    if mode == "big":
        w, h = int(rng.integers(1300, 4000)), int(rng.integers(700, 2200))
        nf = int(rng.choice([2000, 4000, 8000, 10000]))
        B = int(rng.choice([1, 2, 4]))
        kind = str(rng.choice(["synth", "synth", "synth", "pairs_noise"]))
    elif mode == "batched":
        w, h = int(rng.integers(40, 260)) * 4, int(rng.integers(100, 600))
        B = int(rng.integers(33, 81))
        nf = int(rng.choice([200, 500, 1000, 2000]))
        while nlev > 1 and min(w, h) / sf ** (nlev - 1) < 75:
            nlev -= 1
    if kind in ("noise", "pairs_noise") and w * h > 500000 and mode != "big":
        w, h = w // 2, h // 2
    if kind == "pairs_noise" and mode == "big":
        w, h = w // 2, h // 2
    params = (nf, sf, nlev, ini, mn)
    # small launches take k_fast (a workgroup per cell), large ones k_fast_wave; every other trial forces the latter (diagnostic knob)
    orbx.debug_set("fast_wg_max_cells", 0 if t & 1 else None)
    libm = 0 if t % 3 == 0 else 1  # every third trial with the DOUBLE libm reading (round 5), the oracle set to the same
    O.set_libm_variant(libm)
    tag = "trial %d: %dx%d B=%d %s params=%r%s%s" % (t, w, h, B, kind, params, " wave-per-cell" if t & 1 else "", " libm-double" if libm == 0 else "")
    try:
        e = orbx.ORBextractor(*params, max_width=w, max_height=h, max_batch=B)
        e.set_libm_variant(libm)
    except orbx.OrbxError as err:
        if err.code in (orbx.E_TOOSMALL, orbx.E_BADARG):
            skipped += 1
            print(tag, "-> skipped (%s)" % err)
            continue
        raise
    oe = O.Extractor(*params)
    fr = images(kind, B, w, h, 5000 + t)
    cap = nf + 64
    try:
        res = e.extract_batch(fr)
    except orbx.OrbxError as err:
        if err.code == orbx.E_TOOSMALL:
            skipped += 1
            print(tag, "-> skipped (%s)" % err)
            e.close()
            continue
        raise
    ora = [oe(f, cap=cap) for f in fr]
    ok = all(r[0] == o[0] and same(r[1], r[2], o[1], o[2]) for r, o in zip(res, ora))
    # device-resident fused call (sync and stream-ordered) incl. matching of consecutive pairs
    win = int(rng.choice([30, 100, 100, 400, 4096]))
    ratio = float(rng.choice([0.9, 0.9, 0.6, 0.75, 1.0]))
    ori = bool(rng.integers(0, 2))
    if B >= 2:
        d_img = torch.from_numpy(fr).cuda()
        first = np.arange(0, B - 1, 2, dtype=np.int32)
        npairs = len(first)
        sets = []
        for use_async in range(2):
            o = dict(k=torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"), d=torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda"),
                     n=torch.zeros(B, dtype=torch.int32, device="cuda"), m=torch.zeros(npairs * cap, dtype=torch.int32, device="cuda"),
                     nm=torch.zeros(npairs, dtype=torch.int32, device="cuda"), st=torch.zeros(npairs * 3, dtype=torch.int32, device="cuda"))
            f = e.extract_match_batch_device_async if use_async else e.extract_match_batch_device
            f(d_img, B, w, h, w, w * h, o["k"], o["d"], o["n"], first, first + 1, (0, w, 0, h), o["m"], o["nm"], o["st"], win, ratio, ori, cap)
            sets.append(o)
        e.wait()
        for o in sets:
            n = o["n"].cpu().numpy()
            kk = o["k"].cpu().numpy().view(KP).reshape(B, cap)
            dd = o["d"].cpu().numpy().reshape(B, cap, 32)
            mm = o["m"].cpu().numpy().reshape(npairs, cap)
            nm = o["nm"].cpu().numpy()
            st = o["st"].cpu().numpy().reshape(npairs, 3)
            for f_ in range(B):
                ok &= n[f_] == len(ora[f_][1]) and same(kk[f_, :n[f_]], dd[f_, :n[f_]], ora[f_][1], ora[f_][2])
            for p in range(npairs):
                a, b = ora[2 * p], ora[2 * p + 1]
                onm, om12, ost = O.match_init(a[1], a[2], b[1], b[2], (0, w, 0, h), win, ratio, ori)
                ok &= nm[p] == onm and np.array_equal(mm[p, :len(om12)], om12) and st[p].tolist() == ost.tolist()
    nk = [len(o[1]) for o in ora]
    print(tag, "win=%d ratio=%.2f ori=%d keypoints=%d..%d ->" % (win, ratio, ori, min(nk), max(nk)), "ok" if ok else "MISMATCH", flush=True)
    bad += not ok
    e.close()
print("FUZZ %s: %d trials, %d skipped (too small), %d mismatching, %.0f s" % ("OK" if bad == 0 else "FAILED", trials, skipped, bad, time.time() - t_start))
sys.exit(1 if bad else 0)
