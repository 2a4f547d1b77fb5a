#!/usr/bin/env python3
"""Counterpart of the reference's ``train.py`` on the MI355X path.

    python examples/train.py --path /data/tandt/train [--epochs 100] [--resize 1.0]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        examples/train.py --path /data/tandt/train            # one camera view per GPU and step

Same schedule as train.py:44-83 -- shuffled views, densification every 5th and alpha reset every 15th
epoch in (1, 50], a checkpoint every 10th epoch and ``final.npy`` in the reference's record layout -- on
``GSplatDataset`` (COLMAP model + images), ``Trainer`` (fused render / loss / Adam kernels, RCCL
all-reduce of the gradients when launched with several ranks).  No matplotlib preview.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--path", required=True, help="dataset directory (sparse/0/*.bin + images/)")
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--resize", type=float, default=1.0)
    ap.add_argument("--out", default="data")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from easygaussiansplatting_amd import scene as S
    from easygaussiansplatting_amd.dataset import GSplatDataset
    from easygaussiansplatting_amd.trainer import Trainer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    rank = dist.get_rank() if world > 1 else 0
    print("Try to training %s ..." % a.path) if rank == 0 else None
    ds = GSplatDataset(a.path, resize_rate=a.resize)
    gs = ds.gs
    start = S.Scene(gs["pw"].copy(), gs["rot"].copy(), gs["scale"].copy(), gs["alpha"].copy(), gs["sh"].copy(),
                    None)
    views_per_step = world
    steps = (len(ds) // views_per_step) * a.epochs
    tr = Trainer(start, ds.cameras, ds.images, max_steps=steps, scene_size=ds.sence_size)
    os.makedirs(a.out, exist_ok=True)
    for epoch in range(a.epochs):
        loss = tr.fit(1, views_per_step=views_per_step, rng_seed=epoch, densify_until=-1)[0]
        if rank == 0:
            print("epoch:%d avg_loss:%f gaussians:%d" % (epoch, loss, tr.params["pws"].shape[0]))
        if 1 < epoch <= 50:                                   # train.py:70-76
            if epoch % 5 == 0:
                tr.densify(verbose=rank == 0)
            if epoch % 15 == 0:
                tr.reset_alpha()
        if epoch % 10 == 0 and rank == 0:
            tr.save(os.path.join(a.out, "epoch%04d.npy" % epoch))
    if rank == 0:
        tr.save(os.path.join(a.out, "final.npy"))
        print("Training is finished.")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
