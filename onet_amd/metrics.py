"""Evaluation step either side of the hot path, mirroring the reference's helpers in
utils_20231218.py ("UT"): same names, argument meaning and edge-case behaviour, with the O(B*H*W)
work done by HIP kernels (per-frame min/max, per-image confusion counts) and only 4 integers per
image brought to the host."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .ops import _p, _stream, require_gpu

_EPS = float(np.spacing(1))


def tensor_normal_per_frame(input: torch.Tensor) -> torch.Tensor:
    """Scale every (b, c) frame to [0,1]: (x - min) / (max - min + np.spacing(1))  (UT:673-689)."""
    assert input.dim() == 4
    require_gpu(input)
    x = input.contiguous()
    nb, nc, h, w = x.shape
    y = torch.empty_like(x)
    _lib.call("onet_normalise_per_frame", _p(x), _p(y), nb * nc, h * w, _stream())
    return y


def confusion_counts(preds: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """[B, 4] int64 = (TP, FP, FN, TN) per image, positive class 1.  preds/targets: [B, H, W] (any integer
    or float dtype holding 0/1, as the reference's label tensors do)."""
    require_gpu(preds, targets)
    assert preds.shape == targets.shape
    B = preds.shape[0]
    p = preds.reshape(B, -1).to(torch.int64).contiguous()
    t = targets.reshape(B, -1).to(torch.int64).contiguous()
    counts = torch.empty((B, 4), dtype=torch.int64, device=p.device)
    _lib.call("onet_confusion2", _p(p), _p(t), _p(counts), B, p.shape[1], _stream())
    return counts


def _tot(counts):
    c = counts.sum(dim=0).tolist() if counts.dim() == 2 else counts.tolist()
    return [int(v) for v in c]


def _acc(counts) -> float:
    """(TP + TN) / all  (UT:100-117)."""
    tp, fp, fn, tn = _tot(counts)
    return (tp + tn) / float(tp + fp + fn + tn)


def _miou(counts) -> float:
    """mean IoU over the 2 classes with the reference's empty-class rules (UT:119-155): a class absent from
    both prediction and ground truth scores 1, absent from exactly one scores 0.  The reference accumulates
    `torch.sum(inter) / torch.sum(union)` (a float32 tensor) and ends with `miou.item() / nums`: when NEITHER class
    takes the tensor branch -- prediction and ground truth each a single, different class -- `miou` is still a python
    float there and UT:152 raises AttributeError; mirrored, since "same error behaviour" is the drop-in rule."""
    tp, fp, fn, tn = _tot(counts)
    miou, is_tensor = np.float32(0.0), False
    for inter, gt, pd in ((tn, tn + fp, tn + fn), (tp, tp + fn, tp + fp)):      # class 0, class 1
        if gt == 0 and pd == 0:
            miou = np.float32(miou + np.float32(1.0))
        elif gt == 0 or pd == 0:
            pass
        else:
            miou = np.float32(miou + np.float32(inter) / np.float32(gt + pd - inter))   # int64 / int64 -> float32 tensor
            is_tensor = True
    if not is_tensor:
        raise AttributeError("'float' object has no attribute 'item'")
    return float(miou) / 2


def _ratio32(num: int, den: int) -> float:
    """`torch.sum(a) / (torch.sum(b) + np.spacing(1))` as the reference evaluates it: int64 tensor + python float ->
    float32 tensor, int64 / float32 -> float32."""
    return float(np.float32(num) / (np.float32(den) + np.float32(_EPS)))


def _target_iou(counts) -> float:
    tp, fp, fn, tn = _tot(counts)
    return _ratio32(tp, tp + fp + fn)                                            # UT:157-173


def _detection_rate(counts) -> float:
    tp, fp, fn, tn = _tot(counts)
    return _ratio32(tp, tp + fn)                                                 # UT:175-186


def _false_alarm_rate(counts) -> float:
    tp, fp, fn, tn = _tot(counts)
    return _ratio32(fp, fp + tn)                                                 # UT:188-192


def flipped(counts):
    """confusion counts of 1 - pred: TP<->FN, FP<->TN."""
    return counts[..., [2, 3, 0, 1]]


def _hungarian_match(counts, num_k: int = 2):
    """UT:258-285 on the confusion counts: votes[c1][c2] = #(pred == c1 and gt == c2), assignment minimising
    N - votes (scipy's linear_sum_assignment, the reference's own dependency, so ties resolve identically).
    -> [(pred class, gt class), ...]."""
    assert num_k == 2
    from scipy.optimize import linear_sum_assignment
    tp, fp, fn, tn = _tot(counts)
    votes = np.array([[tn, fn], [fp, tp]], dtype=np.float64)
    rows, cols = linear_sum_assignment(float(tp + fp + fn + tn) - votes)
    return [(int(r), int(c)) for r, c in zip(rows, cols)]


def reorder_segmentation(predict_label: torch.Tensor, gt_label: torch.Tensor) -> torch.Tensor:
    """UT:360-375: relabel the prediction by the Hungarian match between predicted and ground-truth classes (2 classes:
    keep, or swap 0 <-> 1); returns a new tensor of gt_label's shape."""
    require_gpu(predict_label, gt_label)
    assert predict_label.numel() == gt_label.numel()
    c = confusion_counts(predict_label.reshape(1, -1), gt_label.reshape(1, -1))
    match = dict(_hungarian_match(c, 2))
    p = predict_label.reshape(gt_label.shape).to(torch.int64).contiguous()
    if match.get(0, 0) == 1:                                                     # swap
        out = torch.empty_like(p)
        _lib.call("onet_flip_labels", _p(p), _p(out), p.numel(), _stream())
    else:
        out = p.clone()
    return out.to(predict_label.dtype)


def re_assign_label(predict_label: torch.Tensor, gt_label: torch.Tensor, gt_k: int = 2) -> torch.Tensor:
    """Hard re-assignment of UT:410-453: return 1 - pred if that has the higher pixel accuracy."""
    require_gpu(predict_label, gt_label)
    c = confusion_counts(predict_label.reshape(1, -1), gt_label.reshape(1, -1))
    if _acc(c) < _acc(flipped(c)):
        p = predict_label.to(torch.int64).contiguous()
        out = torch.empty_like(p)
        _lib.call("onet_flip_labels", _p(p), _p(out), p.numel(), _stream())
        return out.to(predict_label.dtype)
    return predict_label


def evaluate(onet, X, labels):
    """One eval step like TS:98-172 (forward in eval mode -> predict_label -> re_assign_label -> metrics)."""
    with torch.no_grad():
        Lt, Vt, Ld, Vd, S = onet(X)
        pred = re_assign_label(onet.predict_label(S), labels)
        c = confusion_counts(pred, labels)
    return {"acc": _acc(c), "miou": _miou(c), "target_iou": _target_iou(c), "dr": _detection_rate(c),
            "far": _false_alarm_rate(c)}
