"""ORACLE (test infrastructure): PyTorch-CPU port of the reference hot path.

The reference executes `ConvModel.forward` (HandPoseModels.py:40-64) as four
torch.nn.Conv1d calls; on CPU that is ATen/oneDNN.  This port issues the same
four convolutions through `torch.nn.functional.conv1d` on the (B,24,T) view of
the (B,T,12,2) input, so it is what the reference's CPU path costs, without
carrying any reference code.  It is the `cpu_baseline` ("kind": "port") that
bench.py times on the GPU box's host cores, and a second checker in tests/.
"""
import torch
import torch.nn.functional as F


def torch_forward(x, state, pos_emb=False):
    """x (B,T,12,2) float32 CPU tensor -> (B,T,21,2).  `state`: conv{1..4}.{weight,bias}."""
    B, T = x.shape[0], x.shape[1]
    h = x.reshape(B, T, 24).transpose(1, 2)                    # :43-46, stride change only
    if pos_emb:                                                # :48-53, :66-84
        if T != 100:
            raise RuntimeError("pos_emb requires T == 100 (HandPoseModels.py:23,80-82)")
        pe = (torch.arange(100, dtype=torch.float32) / 100).view(1, 1, 100).expand(B, 1, 100)
        h = torch.cat([pe, h], dim=1)
    h = F.relu(F.conv1d(h, state["conv1.weight"], state["conv1.bias"], padding=2))   # :55
    h = F.relu(F.conv1d(h, state["conv2.weight"], state["conv2.bias"], padding=2))   # :56
    h = F.relu(F.conv1d(h, state["conv3.weight"], state["conv3.bias"], padding=2))   # :57
    h = F.conv1d(h, state["conv4.weight"], state["conv4.bias"], padding=2)           # :58
    return h.view(B, 21, 2, T).permute(0, 3, 1, 2)                                   # :60-62


class TorchPort:
    """Holds a state dict; callable like the reference module (inference only)."""

    def __init__(self, state, pos_emb=False):
        self.state = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in state.items()}
        self.pos_emb = pos_emb

    @torch.no_grad()
    def __call__(self, x):
        return torch_forward(x, self.state, self.pos_emb)
