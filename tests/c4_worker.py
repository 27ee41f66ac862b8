"""BASELINE.json config 4 at its full size: 256 synthetic 640x480 frames, synth(640, 480, 1000 + i), cut into contiguous
blocks of 32 frames (sharding.shard_range(256, 8, b)), every block through the fused device call
(orbx_extract_match_batch_device: extraction + SearchForInitialization of its 16 consecutive pairs), the per-frame
keypoint counts gathered with sharding.gather_counts (RCCL when --backend nccl).

One process per rank (RANK / WORLD_SIZE / LOCAL_RANK from the environment, as torch.distributed.run sets them); rank r owns
the contiguous blocks [r * 8 / world, (r + 1) * 8 / world).
With world == 8 that is one block per GPU (the configuration as BASELINE.json states it); with world == 1 one GPU works
through all eight blocks.  Rank 0 writes everything it needs for the oracle check to --out (npz): the gathered counts of
all 256 frames and, for its own blocks, keypoints / descriptors / matches.  Every rank also writes its own blocks to
--out.rank<r>.npz so that the test can check all of them."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PARAMS = (1000, 1.2, 8, 20, 7)
W, H, N_FRAMES, N_BLOCKS, CAP = 640, 480, 256, 8, 1000


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="none", choices=("none", "gloo", "nccl"))
    ap.add_argument("--one-device", action="store_true", help="every rank uses cuda:0 (rehearsal of N > 1 on a one-GPU box)")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    assert N_BLOCKS % world == 0

    import torch
    import torch.distributed as dist
    import orb_slam_tracking_amd as orbx
    from orb_slam_tracking_amd import sharding, synth

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    elif args.backend == "gloo":
        dist.init_process_group("gloo")
    cdev = dev if args.backend != "gloo" else torch.device("cpu")

    B = N_FRAMES // N_BLOCKS
    ext = orbx.ORBextractor(*PARAMS, max_width=W, max_height=H, max_batch=B, device=local_rank)
    first = np.arange(0, B, 2, dtype=np.int32)
    my_blocks = range(rank * N_BLOCKS // world, (rank + 1) * N_BLOCKS // world)
    local_counts = []
    res = {}
    for b in my_blocks:
        lo, hi = sharding.shard_range(N_FRAMES, N_BLOCKS, b)
        assert hi - lo == B
        frames = np.stack([synth.synth(W, H, 1000 + i) for i in range(lo, hi)])
        d_img = torch.from_numpy(frames).to(dev)
        d_k = torch.zeros(B * CAP * 28, dtype=torch.uint8, device=dev)
        d_d = torch.zeros(B * CAP * 32, dtype=torch.uint8, device=dev)
        d_n = torch.zeros(B, dtype=torch.int32, device=dev)
        d_m = torch.zeros((B // 2) * CAP, dtype=torch.int32, device=dev)
        d_nm = torch.zeros(B // 2, dtype=torch.int32, device=dev)
        ext.extract_match_batch_device(d_img, B, W, H, W, W * H, d_k, d_d, d_n, first, first + 1, (0, W, 0, H), d_m, d_nm, None,
                                       100, 0.9, True, CAP)
        local_counts.append(d_n.clone())
        res["k%d" % b] = d_k.cpu().numpy()
        res["d%d" % b] = d_d.cpu().numpy()
        res["n%d" % b] = d_n.cpu().numpy()
        res["m%d" % b] = d_m.cpu().numpy()
        res["nm%d" % b] = d_nm.cpu().numpy()
    local = torch.cat(local_counts).to(cdev)
    counts = sharding.gather_counts(local)  # all_gather over RCCL / gloo; identity for one rank
    res["counts_all"] = counts.cpu().numpy()
    res["blocks"] = np.array(list(my_blocks), np.int32)
    np.savez(args.out if rank == 0 else "%s.rank%d.npz" % (args.out, rank), **res)
    if args.backend != "none":
        dist.barrier()
        dist.destroy_process_group()
    ext.close()
    print("c4_worker rank %d/%d ok (%s)" % (rank, world, args.backend))


if __name__ == "__main__":
    main()
