"""Evaluation metric of the reference on the GPU.

`masked_pose_l1` mirrors `maskedPoseL1.forward(prediction, target, lengths)`
(body2hand/src/steps/utils.py:413-428): per utterance the mean absolute error over its first
`lengths[i]` frames, averaged over the batch.  `l1_to_pixels` mirrors `L12Pixels(21, 1280)`
(steps/utils.py:291-299), the "pixel distance" the training loop prints (traintest.py:27-28,139).
"""
import ctypes

import torch

from . import _lib


def masked_pose_l1(prediction, target, lengths=None, return_per_sequence=False):
    """prediction, target: (B, T, 21, 2) float32 CUDA tensors; lengths: (B,) ints or None (= T).
    Returns a 0-dim CUDA tensor (and the (B,) per-utterance means when asked)."""
    if prediction.device.type != "cuda":
        raise RuntimeError("masked_pose_l1 runs on the GPU only")
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
    per_seq = torch.empty((B,), dtype=torch.float32, device=p.device)
    loss = torch.empty((), dtype=torch.float32, device=p.device)
    lib = _lib.load()
    with _lib.on_device(p.device):
        st = torch.cuda.current_stream(p.device).cuda_stream
        _lib.check(lib.b2h_masked_l1(ctypes.c_void_p(p.data_ptr()), ctypes.c_void_p(t.data_ptr()),
                                     ctypes.c_void_p(nf.data_ptr()) if nf is not None else None, B, T,
                                     ctypes.c_void_p(per_seq.data_ptr()), ctypes.c_void_p(loss.data_ptr()),
                                     ctypes.c_void_p(st)))
    return (loss, per_seq) if return_per_sequence else loss


def l1_to_pixels(loss, num_joints=21, upsample=1280):
    """L12Pixels (steps/utils.py:291-299): loss / num_joints * upsample."""
    return loss / num_joints * upsample
