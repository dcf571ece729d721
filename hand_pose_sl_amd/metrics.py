"""Evaluation metrics of the reference on the GPU.

`masked_pose_l1` mirrors `maskedPoseL1.forward(prediction, target, lengths)`
(body2hand/src/steps/utils.py:413-428): per utterance the mean absolute error over its first
`lengths[i]` frames, averaged over the batch.  `weighted_pose_l1` mirrors `poderatedPoseL1`
(`--loss confL1`, steps/utils.py:431-452): the same with prediction and target multiplied by the
target joints' confidences, SUMMED over the batch (the class does not divide).  `l1_to_pixels`
mirrors `L12Pixels(21, 1280)` (steps/utils.py:291-299), the "pixel distance" the training loop
prints (traintest.py:27-28,139).
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib


def _l1(prediction, target, lengths, scores, return_per_sequence, what):
    if prediction.device.type != "cuda":
        raise RuntimeError(f"{what} runs on the GPU only")
    if prediction.shape != target.shape or prediction.dim() != 4 or prediction.shape[2:] != (21, 2):
        raise RuntimeError(f"expected two (B, T, 21, 2) tensors, got {tuple(prediction.shape)} and {tuple(target.shape)}")
    p = prediction.to(torch.float32).contiguous()
    t = target.to(device=p.device, dtype=torch.float32).contiguous()
    B, T = p.shape[0], p.shape[1]
    nf = None
    if lengths is not None:
        nf = torch.as_tensor(lengths).to(device=p.device, dtype=torch.int64).contiguous()
        if nf.shape != (B,):
            raise RuntimeError(f"lengths must have shape ({B},)")
    sc = None
    if scores is not None:
        sc = scores.to(device=p.device, dtype=torch.float32).contiguous()   # the reference moves them too (utils.py:439)
        if sc.shape != (B, T, 21):
            raise RuntimeError(f"scores must have shape ({B}, {T}, 21), got {tuple(sc.shape)}")
    per_seq = torch.empty((B,), dtype=torch.float32, device=p.device)
    loss = torch.empty((), dtype=torch.float32, device=p.device)
    lib = _lib.load()
    with _lib.on_device(p.device):
        st = ctypes.c_void_p(torch.cuda.current_stream(p.device).cuda_stream)
        ptr = lambda x: ctypes.c_void_p(x.data_ptr()) if x is not None else None
        if sc is None:
            _lib.check(lib.b2h_masked_l1(ptr(p), ptr(t), ptr(nf), B, T, ptr(per_seq), ptr(loss), st))
        else:
            _lib.check(lib.b2h_weighted_l1(ptr(p), ptr(t), ptr(sc), ptr(nf), B, T, ptr(per_seq), ptr(loss), st))
    return (loss, per_seq) if return_per_sequence else loss


def masked_pose_l1(prediction, target, lengths=None, return_per_sequence=False):
    """prediction, target: (B, T, 21, 2) float32 CUDA tensors; lengths: (B,) ints or None (= T).
    Returns a 0-dim CUDA tensor (and the (B,) per-utterance means when asked)."""
    return _l1(prediction, target, lengths, None, return_per_sequence, "masked_pose_l1")


def weighted_pose_l1(prediction, target, lengths, scores, return_per_sequence=False):
    """poderatedPoseL1.forward(prediction, target, lengths, scores) (steps/utils.py:437-452): scores
    (B, T, 21) = confidences of the target joints.  Returns the SUM over the batch of the per-utterance
    means, as the reference does."""
    if scores is None:
        raise RuntimeError("weighted_pose_l1 needs the (B, T, 21) scores")
    return _l1(prediction, target, lengths, scores, return_per_sequence, "weighted_pose_l1")


def l1_to_pixels(loss, num_joints=21, upsample=1280):
    """L12Pixels (steps/utils.py:291-299): loss / num_joints * upsample."""
    return loss / num_joints * upsample


class maskedPoseL1(nn.Module):  # noqa: N801 -- the reference's class name (steps/utils.py:413-428)
    """`criterion = maskedPoseL1()` as traintest.py:36-38 builds it; `criterion(prediction, target, lengths)`
    runs the HIP reduction.  Inference / evaluation only (no autograd)."""

    def forward(self, prediction, target, lengths):
        return masked_pose_l1(prediction, target, lengths)


class poderatedPoseL1(nn.Module):  # noqa: N801 -- the reference's class name (steps/utils.py:431-452)
    """`criterion = poderatedPoseL1()` (`--loss confL1`, traintest.py:39-40);
    `criterion(prediction, target, lengths, scores)`."""

    def forward(self, prediction, target, lengths, scores):
        return weighted_pose_l1(prediction, target, lengths, scores)
