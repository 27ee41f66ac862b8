"""Parity at the shape the headline number is measured on (bench.py's default step): one context, 256 resident
640x480 frames + 128 SearchForInitialization pairs per stream-ordered call, the four mirrored input sets of bench.py in rotation,
the output sets in rotation -- whole batches on four lanes (orbx_set_pipeline_depth(4): k_pyramid_bands with three fat bands per
frame in one 768-workgroup launch, k_fast_wave, 2048 selection units per launch) and the two-half-batches mode (depth 0).
EVERY frame (keypoint bytes, descriptor bytes, count) and EVERY pair (matches12, nmatches) of all eight batches is compared with
the CPU oracle (reference: Features/ORBextractor.cpp:1531-1653, Features/ORBmatcher.cpp:11-150); orbx_debug_last_launch asserts
that the batches really took the kernels named above."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CANON = (1000, 1.2, 8, 20, 7)
B, CAP, W, H = 256, 1000, 640, 480
NSETS, NBATCH = 4, 8


def oracle_of_sets(oracle, sets, params=CANON, window=100, nnratio=0.9, threads=8):
    """Per input set: (keypoints[f], descriptors[f]) of every frame and (nmatches, matches12) of every consecutive pair."""
    out = []
    for frames in sets:
        n = len(frames)
        h, w = frames.shape[1:]

        def work(t):
            oe = oracle.Extractor(*params)  # one oracle instance per thread (operator() mutates the instance, cpp:1669)
            res = {}
            for p_ in range(t, n // 2, threads):
                a, b = oe(frames[2 * p_]), oe(frames[2 * p_ + 1])
                m = oracle.match_init(a[1], a[2], b[1], b[2], (0, w, 0, h), window, nnratio, True)
                res[p_] = (a, b, m)
            return res
        with ThreadPoolExecutor(threads) as ex:
            parts = list(ex.map(work, range(threads)))
        res = {}
        for r in parts:
            res.update(r)
        out.append(res)
    return out


@pytest.fixture(scope="module")
def headline(orbx, oracle):
    from orb_slam_tracking_amd import synth
    sets = synth.bench_input_sets(B, W, H, 1000, NSETS)  # exactly what bench.py builds for rank 0
    return sets, oracle_of_sets(oracle, sets)


def compare_batch(got, exp, nframes=B, cap=CAP):
    """got: host copies of one output set (k, d, n, m, nm); exp: the oracle's results of the input set, by pair."""
    n = got["n"]
    kk = got["k"].reshape(nframes, cap * 28)
    dd = got["d"].reshape(nframes, cap * 32)
    mm = got["m"].reshape(nframes // 2, cap)
    for p_ in range(nframes // 2):
        a, b, (nm, m12, _) = exp[p_]
        for f, (_, ko, do) in ((2 * p_, a), (2 * p_ + 1, b)):
            assert n[f] == len(ko), (f, n[f], len(ko))
            assert kk[f, :n[f] * 28].tobytes() == ko.tobytes(), ("keypoints", f)
            assert dd[f, :n[f] * 32].tobytes() == do.tobytes(), ("descriptors", f)
        assert got["nm"][p_] == nm, (p_, got["nm"][p_], nm)
        assert np.array_equal(mm[p_, :len(m12)], m12), ("matches12", p_)


@pytest.mark.parametrize("depth", [4, 0])
def test_headline_shape_equals_oracle(orbx, headline, depth):
    import torch
    sets, exp = headline
    d_imgs = [torch.from_numpy(s).cuda() for s in sets]
    e = orbx.ORBextractor(*CANON, max_width=W, max_height=H, max_batch=B)
    if depth:
        e.set_pipeline_depth(depth)
    nout = max(2, depth)
    outs = [dict(k=torch.zeros(B * CAP * 28, dtype=torch.uint8, device="cuda"), d=torch.zeros(B * CAP * 32, dtype=torch.uint8, device="cuda"),
                 n=torch.zeros(B, dtype=torch.int32, device="cuda"), m=torch.full(((B // 2) * CAP,), -7, dtype=torch.int32, device="cuda"),
                 nm=torch.zeros(B // 2, dtype=torch.int32, device="cuda")) for _ in range(nout)]
    first = np.arange(0, B, 2, dtype=np.int32)
    snaps = {}

    def snap(k):
        snaps[k] = {key: v.cpu().numpy().copy() for key, v in outs[k % nout].items()}
    for k in range(NBATCH):
        if k >= nout:      # the output set is about to be reused: its batch is the oldest in flight
            e.wait_one()
            snap(k - nout)
        o = outs[k % nout]
        e.extract_match_batch_device_async(d_imgs[k % NSETS], B, W, H, W, W * H, o["k"], o["d"], o["n"], first, first + 1, (0, W, 0, H),
                                           o["m"], o["nm"], None, 100, 0.9, True, CAP)
        info = e.debug_last_launch()
        # the launch shape the bench times: banded pyramid with three fat bands, one wave per FAST cell
        assert info["pyramid_banded"] == 1 and info["pyramid_bands"] == 3 and info["fast_wave"] == 1, info
        if depth:
            assert info["split"] == 0 and info["frames_per_launch"] == B and info["lane"] == k % depth + 1, info
        else:
            assert info["split"] == 1 and info["frames_per_launch"] == B // 2 and info["lane"] == 0, info
        assert info["wide_with_batch"] == 0, info  # no bench pair leaves k_match_jacobi
        assert info["octree_instance"] == 2048, info  # level 0 of these frames has more than 1024 candidates: 40 KB selection units
    e.wait()
    for k in range(NBATCH - nout, NBATCH):
        snap(k)
    e.close()
    assert sorted(snaps) == list(range(NBATCH))
    for k in range(NBATCH):
        compare_batch(snaps[k], exp[k % NSETS])
    # the figures bench.py prints in its config: all 1000 features in every frame
    assert all(int(snaps[k]["n"].min()) == 1000 for k in range(NBATCH))
