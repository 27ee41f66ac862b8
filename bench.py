#!/usr/bin/env python3
"""bench.py — frames/sec of the ORB extract + initialization-match hot path on MI355X.

One step = one pass of the hot path over one batch of synthetic 640x480 frames that are already resident in HBM:
  orbx_extract_batch_device(B frames)  ->  orbx_match_init_batch_device(B/2 consecutive pairs, window 100, ratio 0.9)
  ->  (N > 1) RCCL all_gather of the per-frame keypoint counts.
Frames are sharded per rank (weak scaling: B frames per GPU), one process per GPU.
Prints ONE JSON line on rank 0 (see the task contract): metric/value/roofline/cpu_baseline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PARAMS = (1000, 1.2, 8, 20, 7)  # BASELINE "canonical" preset: 1000 features, 8 levels
W, H = 640, 480
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def level_sizes(w, h, nlevels=8, sf=1.2):
    out, s = [], np.float32(1.0)
    for _ in range(nlevels):
        inv = np.float32(1.0) / s
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
        s = np.float32(np.float64(s) * np.float64(np.float32(sf)))
    return out


def algorithmic_bytes(w, h, n_kp):
    """Per-frame algorithmic bytes of each stage (SURVEY.md 8(d) stage-streaming model)."""
    P = [a * b for a, b in level_sizes(w, h)]
    sp = sum(P)
    return {
        "pyramid": (sp - P[-1]) + (sp - P[0]),          # level reads + level writes, 7 launches
        "fast": sp,                                      # every level pixel read once
        "describe": 2 * sp + n_kp * (749 + 512) + n_kp * 60,  # blur read+write of the model, disc + samples, outputs
        "total": 5 * sp - P[0] - P[-1] + 1321 * n_kp,
    }


def cpu_share():
    """Host cores this process may really use: cgroup quota if set, else the affinity mask; capped at 64."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, 64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline leg")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only to rehearse the N > 1 path)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0 (with --backend gloo)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import orb_slam_tracking_amd as orbx
    from orb_slam_tracking_amd import sharding, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where the collectives' tensors live

    B = args.batch
    cap = 1000
    # this rank's shard of the global batch (frame i -> rank i // B, contiguous blocks; pairs never straddle ranks)
    lo, hi = sharding.shard_range(B * world, world, rank)
    frames = synth.synth_frames(hi - lo, W, H, seed0=1000 + lo // 2)
    d_img = torch.from_numpy(frames).to(dev)
    d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev)
    d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    d_m = torch.zeros((B // 2) * cap, dtype=torch.int32, device=dev)
    d_nm = torch.zeros(B // 2, dtype=torch.int32, device=dev)
    first = np.arange(0, B, 2, dtype=np.int32)
    second = first + 1
    counts_all = torch.zeros(B * world, dtype=torch.int32, device=cdev)

    ext = orbx.ORBextractor(*PARAMS, max_width=W, max_height=H, max_batch=B, device=local_rank)

    # N > 1: the all_gather of step k's counts (RCCL over xGMI) is issued asynchronously from a snapshot of the counts and
    # runs under the kernels of step k + 1; it is waited for before the next one is issued and before the clock stops
    snaps = [torch.zeros(B, dtype=torch.int32, device=cdev) for _ in range(2)]
    pending = [None]
    nstep = [0]

    def finish_gather():
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None

    def step():
        # one call = the whole hot path of the batch (orbx_extract_match_batch_device): extraction of B frames and
        # SearchForInitialization of the B/2 consecutive pairs
        ext.extract_match_batch_device(d_img, B, W, H, W, W * H, d_k, d_d, d_n, first, second, (0, W, 0, H), d_m, d_nm, None,
                                       100, 0.9, True, cap)
        if world > 1:
            finish_gather()
            snap = snaps[nstep[0] & 1]
            snap.copy_(d_n)  # the call above has completed: d_n is final
            pending[0] = dist.all_gather_into_tensor(counts_all, snap, async_op=True)
            nstep[0] += 1

    def barrier():
        finish_gather()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ext.profile_enable(True)
    ext.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = ext.profile_get()
    ext.profile_enable(False)

    if rank == 0:
        n_kp = float(d_n.float().mean().item())
        nm_mean = float(d_nm.float().mean().item())
        ab = algorithmic_bytes(W, H, n_kp)
        # dominant kernel = stage with the largest device time per step; all three are HBM-streaming / gather bound
        dev_ms = {s: prof[s][0] / args.steps for s in ("pyramid", "fast", "describe", "match")}
        kern = max(("pyramid", "fast", "describe"), key=lambda s: dev_ms[s])
        launches = max(prof[kern][1], 1)
        avg_launch_ms = prof[kern][0] / launches
        bytes_per_launch = ab[kern] * B * args.steps / launches
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        kname = {"pyramid": "k_pyramid_bands", "fast": "k_fast", "describe": "k_describe_patch"}[kern]
        # HBM bytes per launch from the committed PMC pass (rocprofv3 FETCH_SIZE + WRITE_SIZE, profiles/r01_traffic.json);
        # null when that profile has no entry for the dominant kernel
        traffic = None
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))["bytes_per_frame"].get(kname)
            if tr:
                traffic = (tr["fetch"] + tr["write"]) * B * args.steps / launches
        except Exception:
            traffic = None
        out = {
            "metric": "frames/sec (extract+match, 1000 feat, 640x480)",
            "value": B * world * args.steps / dt,
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "640x480 gray frames, 1000 features, 8 levels, FAST 20/7; %d frames per GPU per step "
                                   "resident in HBM, %d consecutive-pair SearchForInitialization (window 100, ratio 0.9)"
                                   % (B, B // 2),
                       "frames_per_gpu": B, "pairs_per_gpu": B // 2, "mean_keypoints": n_kp, "mean_nmatches": nm_mean,
                       "parallelism": "frames sharded per GPU (%d ranks), RCCL all_gather of keypoint counts" % world},
            "roofline": {"bound": "hbm", "kernel": kname,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "avg_launch_ms": avg_launch_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "whole_path_GBs": ab["total"] * B * world * args.steps / dt / 1e9},
            "stage_ms_per_step": {s: prof[s][0] / args.steps for s in prof},
        }
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            cores = cpu_share()
            sample = frames[:8]
            sec1, fr1, _ = O.bench_pairs(PARAMS, sample, 100, 0.9, 1, 1)  # calibrate: one pass, one core
            reps = max(1, min(50, int(args.cpu_seconds / max(sec1, 1e-3))))
            sec, fr, _ = O.bench_pairs(PARAMS, sample, 100, 0.9, cores, reps)
            out["cpu_baseline"] = {"value": fr / sec, "unit": "frames/s", "cores": cores, "kind": "port",
                                   "sample": "oracle restatement (scalar C++, -O3, no SIMD): %d threads x %d reps of 8 of the same "
                                             "640x480 frames (extract A + extract B + match per pair); 1-core rate %.1f frames/s"
                                             % (cores, reps, fr1 / sec1),
                                   "one_core_value": fr1 / sec1}
            out["speedup_vs_cpu_all_cores"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
