"""End-to-end counterpart of Train_Onet_on_simclutter_20250407.py on synthetic K-distributed clutter:
same step order (TS:209-219), Adam(lr=5e-6) with the /2-every-100-epochs schedule (TS:181-182,248-249),
eval every 50 epochs with the reference's metrics (TS:98-172), checkpoint {'net','epoch'} at the end.

    python examples/train_simclutter_synthetic.py --frames 64 --size 128 --batch 16 --epochs 3
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_simclutter_synthetic.py
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import Onet_vanilla_20240606 as onet_vanilla_model   # noqa: E402  (the drop-in module name)
from onet_amd import data, io, metrics               # noqa: E402
from onet_amd.trainer import fit, init_distributed   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--out", default="/tmp/onet_out")
    args = ap.parse_args()
    rank, world, local = init_distributed()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    os.makedirs(args.out, exist_ok=True)

    if rank == 0:                 # one writer; the other ranks read the finished file
        X, lab = data.make_clutter_batch(args.frames, args.size, args.size, seed=1981, with_labels=True)
        io.save_simclutter_pt(os.path.join(args.out, "synthetic_kdist.pt"), X, lab, [0] * args.frames)
    if world > 1:
        torch.distributed.barrier()
    imgs, labels, snrs = io.load_simclutter_pt(os.path.join(args.out, "synthetic_kdist.pt"))
    tr, te = io.split_train_test(args.frames)
    imgs = metrics.tensor_normal_per_frame(imgs.to(dev))            # DS:110, on the GPU
    train = [(imgs[tr[i:i + args.batch]],) for i in range(0, len(tr) - args.batch + 1, args.batch)]
    Xte, Lte = imgs[te], labels[te].to(dev)

    onet = onet_vanilla_model.Onet(in_chns=1, binit=True, bshare=True).to(dev)

    def evaluate(net, epoch):
        return metrics.evaluate(net, Xte, Lte)

    hist = fit(onet, train, dev, args.epochs, schedule="sim", eval_fn=evaluate, eval_every=max(1, args.epochs - 1),
               out_root=args.out if rank == 0 else None, rank=rank, world=world)
    if rank == 0:
        print("final:", hist[-1])


if __name__ == "__main__":
    main()
