"""Wire formats of the reference so real reference artefacts interoperate.

* checkpoints: ``{'net': <232-key state_dict>, 'epoch': e}`` (TS:264-266) or ``{'net', 'save_epoch'}``
  (TZ:145-149); resume = ``load_state_dict(torch.load(f)['net'])`` (TZ:77-82).
* sim-clutter data file (``rayleigh_2sigma.pt``; RG:317-324, read at DS:106-112): a dict
  ``{'rayleigh_imgs': float32 [N,1,H,W], 'rayleigh_labels': float32 [N,H,W], 'psnr': list[int]}``.
* ZY-3 thumbnails (DZ:92-106): ``{id: {'true_color': [3,H,W], 'mask': [H,W]}}``."""
from __future__ import annotations

import numpy as np
import torch

from .modules import invalidate_packed


def save_checkpoint(onet, path, epoch, zy3: bool = False):
    torch.save({"net": onet.state_dict(), ("save_epoch" if zy3 else "epoch"): epoch}, path)


def load_checkpoint(onet, path, map_location=None):
    ck = torch.load(path, map_location=map_location)
    onet.load_state_dict(ck["net"])
    invalidate_packed(onet)
    return ck.get("epoch", ck.get("save_epoch"))


def save_simclutter_pt(path, imgs, labels, psnr):
    """imgs [N,1,H,W] float32, labels [N,H,W] float32, psnr list[int] -- the schema DS:106-112 reads."""
    imgs = torch.as_tensor(np.asarray(imgs), dtype=torch.float32)
    labels = torch.as_tensor(np.asarray(labels), dtype=torch.float32)
    assert imgs.dim() == 4 and imgs.shape[1] == 1 and labels.shape == (imgs.shape[0],) + tuple(imgs.shape[2:])
    torch.save({"rayleigh_imgs": imgs, "rayleigh_labels": labels, "psnr": [int(v) for v in psnr]}, path)


def load_simclutter_pt(path):
    """-> (imgs [N,1,H,W] f32, labels [N,H,W] f32, snrs int64 [N]) exactly as DS:106-112 unpacks it (no
    normalisation here: apply onet_amd.metrics.tensor_normal_per_frame on the GPU)."""
    d = torch.load(path, map_location="cpu")
    return d["rayleigh_imgs"], d["rayleigh_labels"], torch.tensor(d["psnr"])


def split_train_test(n, seed=1981, frac=0.9):
    """90/10 split with NumPy's global-RNG shuffle semantics of DS:118-124."""
    rng = np.random.RandomState(seed)
    ids = np.arange(n)
    rng.shuffle(ids)
    k = int(n * frac)
    return ids[:k], ids[k:]


def load_zy3_dict(path):
    """{id: {'true_color': [3,H,W], 'mask': [H,W]}} -> (ids, X [N,3,H,W] f32, masks [N,H,W] f32), the items in the file's own
    (insertion) order: the reference's datasets index `list(data_dict.keys())` (DZ:139-141, test branch DZ:163-169 returns the
    stored `true_color` / `mask` tensors as they are).  Reads what DZ:92-122 reads: a torch-saved dict."""
    d = torch.load(path, map_location="cpu")
    ids = list(d.keys())
    X = torch.stack([torch.as_tensor(d[k]["true_color"], dtype=torch.float32) for k in ids])
    M = torch.stack([torch.as_tensor(d[k]["mask"], dtype=torch.float32) for k in ids])
    return ids, X, M
