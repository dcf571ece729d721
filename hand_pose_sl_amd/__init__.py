"""hand_pose_sl_amd -- MI355X-native body->hand keypoint inference path.

A from-scratch gfx950 implementation of the one convolutional model of
benoriol/hand_pose_sl (`ConvModel`, body2hand/src/models/HandPoseModels.py:17-64)
behind the reference's own call surface.  See DESIGN.md and include/b2h.h.
"""
from .conv_model import ConvModel, LinearPositionalEmbedding, target_transform  # noqa: F401
from .evaluate import validate  # noqa: F401
from .metrics import l1_to_pixels, masked_pose_l1, maskedPoseL1, poderatedPoseL1, weighted_pose_l1  # noqa: F401
from .transformer_enc import PositionalEncoding, TransformerEnc  # noqa: F401

__all__ = ["ConvModel", "LinearPositionalEmbedding", "target_transform", "masked_pose_l1", "weighted_pose_l1",
           "l1_to_pixels", "maskedPoseL1", "poderatedPoseL1", "validate", "TransformerEnc", "PositionalEncoding"]
