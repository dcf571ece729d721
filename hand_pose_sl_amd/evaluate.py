"""The reference's evaluation loop around the two text-free models (SURVEY.md 8f N4 + the callers of
the hot path): `validate(model, val_loader, criterion, device, args)`, body2hand/src/steps/traintest.py:168-213.

Per batch the reference computes  prediction = model(batch["body_kp"])  (traintest.py:183,194),
mask_output(prediction, n_frames)  (:203, steps/utils.py:309-312),  loss = criterion(prediction,
batch["target_kp"], n_frames[, batch["target_conf"]])  (:207-210)  and averages `loss.item()` over the
batches with AverageMeter (steps/utils.py:11-25).  Here the forward is the HIP model, the criterion one
of the two HIP reductions of `metrics.py`; the tail mask is skipped because neither criterion reads a
frame at or beyond `n_frames[i]` (it could not change the value).  One host synchronisation per call,
not per batch: the per-batch losses stay on the device until the end.
"""
import torch

from .metrics import l1_to_pixels, masked_pose_l1, weighted_pose_l1

LOSSES = ("L1", "confL1")  # the two `--loss` choices validate() can evaluate (traintest.py:207-210)


def validate(model, val_loader, loss="L1", return_pixels=False):
    """model: hand_pose_sl_amd.ConvModel or TransformerEnc on a GPU; val_loader: iterable of batches
    (dicts with "body_kp" (B,T,12,2), "target_kp" (B,T,21,2), "n_frames", and "target_conf" (B,T,21)
    for loss="confL1"), tensors on the host or the device as the reference's loader yields them.
    Returns the mean over batches of the batch losses (a float), like the reference; with
    return_pixels also L12Pixels(21, 1280) of it (traintest.py:27-28,139)."""
    if loss not in LOSSES:
        # MSE / huber make the reference's validate() fail with an unbound `loss` (traintest.py:207-211)
        raise ValueError(f"validate() evaluates --loss L1 or confL1, not {loss!r}")
    dev = next(model.parameters()).device
    model.eval()
    losses = []
    with torch.no_grad():
        for batch in val_loader:
            prediction = model(batch["body_kp"])
            target = batch["target_kp"].to(dev)
            if loss == "L1":
                losses.append(masked_pose_l1(prediction, target, batch["n_frames"]))
            else:
                losses.append(weighted_pose_l1(prediction, target, batch["n_frames"], batch["target_conf"]))
    if not losses:
        raise ZeroDivisionError("empty loader")  # AverageMeter.get_average() divides by its count
    value = float(torch.stack(losses).double().mean())
    return (value, l1_to_pixels(value)) if return_pixels else value
