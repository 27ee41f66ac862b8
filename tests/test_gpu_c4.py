"""BASELINE.json config 4 ("batch of 256 synthetic 640x480 frames sharded per-frame across 8 MI355X, RCCL gather of
keypoints") exercised at its size: 8 contiguous blocks of 32 frames (sharding.shard_range), every block through the fused
device call, the counts through sharding.gather_counts -- on one GPU (all eight blocks), with two ranks sharing the GPU
(gloo), and over RCCL when the box has more than one GPU.  All 256 frames and all 128 pairs are compared with the oracle."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PARAMS = (1000, 1.2, 8, 20, 7)
W, H, N_FRAMES, CAP, B = 640, 480, 256, 1000, 32


@pytest.fixture(scope="module")
def c4_oracle(oracle):
    """The oracle's results for the 256 frames and the 128 consecutive pairs (a few seconds of CPU)."""
    from orb_slam_tracking_amd import synth
    oe = oracle.Extractor(*PARAMS)
    frames = [synth.synth(W, H, 1000 + i) for i in range(N_FRAMES)]
    ext = [oe(f) for f in frames]
    pairs = [oracle.match_init(ext[2 * p][1], ext[2 * p][2], ext[2 * p + 1][1], ext[2 * p + 1][2], (0, W, 0, H), 100, 0.9, True)
             for p in range(N_FRAMES // 2)]
    return ext, pairs


def _check(files, c4_oracle, orbx):
    ext, pairs = c4_oracle
    counts_exp = np.array([len(e[1]) for e in ext], np.int32)
    seen = set()
    for fn in files:
        z = np.load(fn)
        assert np.array_equal(z["counts_all"], counts_exp), "gathered keypoint counts differ from the oracle (%s)" % fn
        for b in z["blocks"]:
            seen.add(int(b))
            n = z["n%d" % b]
            kk = z["k%d" % b].view(orbx.KEYPOINT_DTYPE).reshape(B, CAP)
            dd = z["d%d" % b].reshape(B, CAP, 32)
            mm = z["m%d" % b].reshape(B // 2, CAP)
            nm = z["nm%d" % b]
            for f in range(B):
                g = b * B + f
                _, ko, do = ext[g]
                assert n[f] == len(ko)
                assert kk[f, :n[f]].tobytes() == ko.tobytes() and np.array_equal(dd[f, :n[f]], do), ("frame", g)
            for p in range(B // 2):
                onm, om12, _ = pairs[b * (B // 2) + p]
                assert nm[p] == onm and np.array_equal(mm[p, :len(om12)], om12), ("pair", b * (B // 2) + p)
    assert seen == set(range(8))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=e, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-4000:]
    return p.stdout


def test_config4_one_gpu_all_blocks(orbx, c4_oracle, tmp_path):
    out = str(tmp_path / "c4.npz")
    _run([sys.executable, os.path.join(ROOT, "tests", "c4_worker.py"), "--backend", "none", "--out", out])
    _check([out], c4_oracle, orbx)


def test_config4_two_ranks_sharing_the_gpu_gloo(orbx, c4_oracle, tmp_path):
    """N > 1 on a one-GPU box: two processes (fresh children, started by the launcher before anything touches the GPU), each
    with its own context on cuda:0 and four of the eight blocks; the collective is gloo here."""
    out = str(tmp_path / "c4g.npz")
    _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
          "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "c4_worker.py"), "--backend", "gloo", "--one-device",
          "--out", out])
    _check([out, out + ".rank1.npz"], c4_oracle, orbx)


def test_config4_rccl(orbx, c4_oracle, tmp_path):
    """The configuration as BASELINE.json states it: one rank per GPU, RCCL all_gather of the counts over xGMI.  Needs more
    than one GPU; uses all of them up to 8 (a power of two)."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("RCCL needs more than one GPU: this box has %d (the driver's multi-GPU run covers N = 2, 4, 8)" % n)
    world = 8 if n >= 8 else 4 if n >= 4 else 2
    out = str(tmp_path / "c4r.npz")
    _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
          "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "c4_worker.py"), "--backend", "nccl", "--out", out],
         env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    _check([out] + ["%s.rank%d.npz" % (out, r) for r in range(1, world)], c4_oracle, orbx)
