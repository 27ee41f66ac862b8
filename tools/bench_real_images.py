#!/usr/bin/env python3
"""The headline call sequence of bench.py on REAL image content instead of the synthetic scenes: 256 frames of 640 x 480 made from
the four DBoW2 demo images of tests/golden/images.npz (each frame one of them, circularly shifted by a frame-specific offset so
that the cells differ), 128 consecutive-pair matches, four pipeline lanes, inputs resident in HBM.  Real images have what the
synthetic scenes lack: large flat or faintly textured regions, where every second FAST cell finds nothing at iniThFAST and is
swept again at minThFAST (k_fast_wave's flat-cell rule; docs/history.md, round 5).  Frames 0, 1, 101 and 255 of the last batch
are compared with the CPU oracle.
usage: bench_real_images.py [steps] [regions] [lanes] [synthetic]     (ORBX_LIB selects the library build; lanes = 1 for kernel
profiles: nothing overlaps; a fourth argument takes bench.py's synthetic input sets instead, for a comparison under one harness)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import orb_slam_tracking_amd as orbx  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
regions = int(sys.argv[2]) if len(sys.argv) > 2 else 3
W, H, B, cap = 640, 480, 256, 1000
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 4
z = np.load(os.path.join(ROOT, "tests", "golden", "images.npz"))
src = [z["dbow%d" % i] for i in range(4)]
sets = []
for s in range(4):
    fr = np.empty((B, H, W), np.uint8)
    for i in range(B):
        # a pair (2k, 2k + 1) is one image at two nearby offsets, like two frames of one camera
        k = i // 2
        fr[i] = np.roll(src[(k + s) % 4], ((k * 5 + s * 11) % 48 + (i & 1) * 2, (k * 7 + s * 13) % 64 + (i & 1) * 3), axis=(0, 1))
    sets.append(fr)
if len(sys.argv) > 4:
    from orb_slam_tracking_amd import synth
    sets = synth.bench_input_sets(B, W, H, 1000, 4)
dev = torch.device("cuda", 0)
d_imgs = [torch.from_numpy(s).to(dev) for s in sets]
outs = [dict(k=torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev), d=torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev),
             n=torch.zeros(B, dtype=torch.int32, device=dev), m=torch.zeros((B // 2) * cap, dtype=torch.int32, device=dev),
             nm=torch.zeros(B // 2, dtype=torch.int32, device=dev)) for _ in range(depth)]
first = np.arange(0, B, 2, dtype=np.int32)
second = first + 1
ext = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
ext.set_pipeline_depth(depth)
nstep = [0]


def step():
    k = nstep[0]
    o = outs[k % depth]
    ext.extract_match_batch_device_async(d_imgs[k % 4], B, W, H, W, W * H, o["k"], o["d"], o["n"], first, second, (0, W, 0, H),
                                         o["m"], o["nm"], None, 100, 0.9, True, cap)
    nstep[0] = k + 1


def barrier():
    ext.wait()
    torch.cuda.synchronize()


for _ in range(16):
    step()
barrier()
vals = []
for _ in range(regions):
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    vals.append(B * steps / (time.perf_counter() - t0))
k = nstep[0] - 1
o = outs[k % depth]
n = o["n"].cpu().numpy()
kp = o["k"].cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(B, cap)
ds = o["d"].cpu().numpy().reshape(B, cap, 32)
ok = True
try:
    import oracle_lib as O
    oe = O.Extractor(1000, 1.2, 8, 20, 7)
    for f in (0, 1, 101, 255):
        ro, ko, do = oe(sets[k % 4][f])
        ok = ok and n[f] == len(ko) and kp[f, :n[f]].tobytes() == np.ascontiguousarray(ko, orbx.KEYPOINT_DTYPE).tobytes() \
            and np.array_equal(ds[f, :n[f]], do)
except ImportError:
    ok = None
print(("synthetic scenes" if len(sys.argv) > 4 else "real images (DBoW2 demo, shifted)") + ": %d frames/s median of %s, mean keypoints %.1f, checked %s, lib %s" %
      (round(float(np.median(vals))), [round(v) for v in vals], float(n.mean()), ok, os.path.basename(orbx.LIB_PATH) if hasattr(orbx, "LIB_PATH") else os.environ.get("ORBX_LIB", "liborbx.so")))
sys.exit(0 if ok in (True, None) else 1)
