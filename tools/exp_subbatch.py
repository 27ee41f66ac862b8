"""Round-4 experiment: a synchronous call of a large-frame batch issued as S stream-ordered sub-batches on S lanes and waited for,
against the two-halves form.  usage: python tools/exp_subbatch.py --config c5"""
import argparse, json, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_config import CFG


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c5")
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    import torch
    import orb_slam_tracking_amd as orbx
    from orb_slam_tracking_amd import synth
    w, h, nf, B, window = CFG[a.config]
    dev = torch.device("cuda", 0)
    d_img = torch.from_numpy(synth.synth_frames(B, w, h, seed0=77)).to(dev)
    cap = nf
    res = {}
    for S in (1, 2, 4):
        sb = B // S
        if sb < 2:
            continue
        ext = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=sb, device=0)
        outs = [dict(k=torch.zeros(sb * cap * 28, dtype=torch.uint8, device=dev), d=torch.zeros(sb * cap * 32, dtype=torch.uint8, device=dev),
                     n=torch.zeros(sb, dtype=torch.int32, device=dev), m=torch.zeros((sb // 2) * cap, dtype=torch.int32, device=dev),
                     nm=torch.zeros(sb // 2, dtype=torch.int32, device=dev)) for _ in range(S)]
        first = np.arange(0, sb - 1, 2, dtype=np.int32)
        if S > 1:
            ext.set_pipeline_depth(S)

        def step():
            if S == 1:
                o = outs[0]
                ext.extract_match_batch_device(d_img, sb, w, h, w, w * h, o["k"], o["d"], o["n"], first, first + 1, (0, w, 0, h), o["m"], o["nm"], None, window, 0.9, True, cap)
                return
            for s in range(S):
                o = outs[s]
                ext.extract_match_batch_device_async(d_img[s * sb:], sb, w, h, w, w * h, o["k"], o["d"], o["n"], first, first + 1, (0, w, 0, h), o["m"], o["nm"], None, window, 0.9, True, cap)
            ext.wait()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        res[f"S{S}"] = {"ms": round(dt * 1e3, 4), "fps": round(B / dt)}
        ext.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
