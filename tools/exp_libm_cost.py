import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import orb_slam_tracking_amd as orbx
from orb_slam_tracking_amd import synth
W, H, cap, B, depth = 640, 480, 1000, 256, 4
sets = [torch.from_numpy(s).cuda() for s in synth.bench_input_sets(B, W, H, 1000, 4)]
e = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
e.set_pipeline_depth(depth)
outs = [dict(k=torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"), d=torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda"),
             n=torch.zeros(B, dtype=torch.int32, device="cuda"), m=torch.zeros((B // 2) * cap, dtype=torch.int32, device="cuda"),
             nm=torch.zeros(B // 2, dtype=torch.int32, device="cuda")) for _ in range(depth)]
first = np.arange(0, B, 2, dtype=np.int32)
def call(i):
    o = outs[i % depth]
    e.extract_match_batch_device_async(sets[i % 4], B, W, H, W, W * H, o["k"], o["d"], o["n"], first, first + 1, (0, W, 0, H), o["m"], o["nm"], None, 100, 0.9, True, cap)
for v in (0, 1, 0, 1):
    e.wait(); e.set_libm_variant(v)
    for i in range(40): call(i)
    e.wait()
    t0 = time.perf_counter()
    for i in range(400): call(i)
    e.wait()
    dt = (time.perf_counter() - t0) / 400
    print("libm variant %d: %.1f k frames/s" % (v, B / dt / 1e3))
