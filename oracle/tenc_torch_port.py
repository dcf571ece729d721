"""ORACLE (test infrastructure): PyTorch-CPU port of the reference's TransformerEnc.forward
(HandPoseModels.py:152-178) built from torch.nn containers and a state dict -- what the
reference costs on CPU for this model; used only as the reported CPU baseline of
tools/bench_tenc.py and as a second checker."""
import torch
import torch.nn as nn


class TencTorchPort(nn.Module):
    def __init__(self, state, nhead=4, nlayers=4):
        super().__init__()
        d = state["pose2hidden_projection.weight"].shape[0]
        layer = nn.TransformerEncoderLayer(d, nhead, d, 0.0)
        self.transformer_encoder = nn.TransformerEncoder(layer, nlayers, enable_nested_tensor=False)
        self.hidden2pose_projection = nn.Linear(d, state["hidden2pose_projection.weight"].shape[0])
        self.pose2hidden_projection = nn.Linear(state["pose2hidden_projection.weight"].shape[1], d)
        self.register_buffer("pe", torch.as_tensor(state["pos_encoder.pe"]).clone())
        own = {k: torch.as_tensor(v) for k, v in state.items() if k != "pos_encoder.pe"}
        self.load_state_dict({**own, "pe": self.pe})
        self.eval()

    @torch.no_grad()
    def forward(self, src):
        bs, T = src.shape[0], src.shape[1]
        h = src.reshape(bs, T, -1).permute(1, 0, 2)                 # :153-166
        h = h + self.pe[:T]                                         # :167
        h = self.pose2hidden_projection(h)                          # :169
        h = self.transformer_encoder(h)                             # :170 (no mask)
        h = self.hidden2pose_projection(h).permute(1, 0, 2)         # :171-172
        return h.reshape(bs, T, 21, 2)
