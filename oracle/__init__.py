"""ORACLE -- test infrastructure, not product code.

CPU restatements of the reference hot path (benoriol/hand_pose_sl
`ConvModel.forward`, body2hand/src/models/HandPoseModels.py:40-64, plus the
pre/post transforms around it).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package, and only as the checker or
the reported-only CPU baseline.  hand_pose_sl_amd never imports it.

Pinned by tests/golden/*.npz (vectors produced by the reference's own classes,
tests/golden/make_golden.py) through tests/test_oracle.py.
"""
from .c_oracle import (build_oracle, forward, forward_from_state, masked_l1, weighted_l1, postprocess,  # noqa: F401
                       preprocess)
from .torch_port import TorchPort, torch_forward  # noqa: F401
from .transformer_oracle import transformer_forward  # noqa: F401
from .tenc_torch_port import TencTorchPort  # noqa: F401
