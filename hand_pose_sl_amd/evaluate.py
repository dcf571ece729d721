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


def _loss_name(criterion, args, loss):
    """Which of the reference's two evaluation losses is meant.  The reference's loop dispatches on
    `args.loss` (traintest.py:207-210), so that wins; then an explicit name; then the criterion's class."""
    if args is not None and getattr(args, "loss", None) is not None:
        return args.loss
    if loss is not None:
        return loss
    if isinstance(criterion, str):
        return criterion
    if criterion is None:
        return "L1"
    name = type(criterion).__name__
    return {"maskedPoseL1": "L1", "poderatedPoseL1": "confL1"}.get(name, name)


def validate(model, val_loader, criterion=None, device=None, args=None, *, loss=None, return_pixels=False):
    """`validate(model, val_loader, criterion, device, args)` exactly as steps/traintest.py:136,168 calls it
    -- `criterion` a maskedPoseL1 / poderatedPoseL1 instance (the reference's or `hand_pose_sl_amd`'s: only
    its class name is read, the HIP reduction of that name computes it), `device` ignored (the model's device
    is used; the reference moves the batch there itself), `args.loss` in {"L1", "confL1"} and, when present,
    `args.model` in {"Conv", "TransformerEnc"} (anything else raises ValueError as traintest.py:199-200 does)
    -- or the short keyword form `validate(model, val_loader, loss="confL1")` / `validate(model, loader, "L1")`.

    model: hand_pose_sl_amd.ConvModel or TransformerEnc on a GPU; val_loader: iterable of batches
    (dicts with "body_kp" (B,T,12,2), "target_kp" (B,T,21,2), "n_frames", and "target_conf" (B,T,21)
    for confL1), tensors on the host or the device as the reference's loader yields them.
    Returns the mean over batches of the batch losses (a float), like the reference; with
    return_pixels also L12Pixels(21, 1280) of it (traintest.py:27-28,139)."""
    loss = _loss_name(criterion, args, loss)
    if loss not in LOSSES:
        # MSE / huber make the reference's validate() fail with an unbound `loss` (traintest.py:207-211)
        raise ValueError(f"validate() evaluates --loss L1 or confL1, not {loss!r}")
    if args is not None and getattr(args, "model", None) not in (None, "Conv", "TransformerEnc"):
        raise ValueError(f"validate() runs the text-free models Conv and TransformerEnc, not {args.model!r}")
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
