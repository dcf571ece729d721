"""ORACLE (test infrastructure): numpy restatement of the reference's `TransformerEnc`
(body2hand/src/models/HandPoseModels.py:118-178), the second body->hand model that runs
without text, as constructed by the CLIs
(`TransformerEnc(ninp=24, nhead=4, nhid=128, nout=42, nlayers=4)`, infer_utterance.py:99-101).

forward (HandPoseModels.py:152-178), eval mode (dropout = identity):
    src (B,T,12,2) -> (B,T,24) -> (T,B,24)                       :153-166
    src + pe[:T]            PositionalEncoding, d_model = 24        :86-103,167  (T <= 100)
    Linear(24 -> 128)       pose2hidden_projection                 :169
    4 x torch.nn.TransformerEncoderLayer(d=128, nhead=4, ff=128), post-norm, ReLU, NO mask
        (the causal mask built at :158-162 is never passed to the encoder, :170)
    Linear(128 -> 42)       hidden2pose_projection                 :171
    -> (B,T,21,2)                                                    :172-176
The attention / layer-norm arithmetic lives in torch.nn (MultiheadAttention: q scaled by
head_dim**-0.5, softmax over all T keys, out_proj; LayerNorm eps 1e-5, biased variance).

Pinned by tests/golden/tenc_*.npz (vectors from the reference class, make_golden.py).
"""
import numpy as np


def _ln(x, g, b, eps=1e-5):
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def transformer_forward(x, state, nhead=4, dtype=np.float32):
    """x (B,T,12,2) -> (B,T,21,2).  `state`: the reference's state_dict as numpy arrays."""
    st = {k: np.asarray(v, dtype=dtype) for k, v in state.items()}
    x = np.asarray(x, dtype=dtype)
    B, T = x.shape[:2]
    pe = st["pos_encoder.pe"]                     # (max_len, 1, 24)
    if T > pe.shape[0]:
        raise RuntimeError(f"TransformerEnc: T = {T} exceeds the positional encoding's max_len {pe.shape[0]} "
                           "(HandPoseModels.py:125,101)")
    h = x.reshape(B, T, 24) + pe[:T, 0][None]     # (B,T,24)
    h = h @ st["pose2hidden_projection.weight"].T + st["pose2hidden_projection.bias"]
    d = h.shape[-1]
    hd = d // nhead
    nl = 1 + max(int(k.split(".")[2]) for k in st if k.startswith("transformer_encoder.layers."))
    for i in range(nl):
        p = f"transformer_encoder.layers.{i}."
        qkv = h @ st[p + "self_attn.in_proj_weight"].T + st[p + "self_attn.in_proj_bias"]
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        q = q.reshape(B, T, nhead, hd).transpose(0, 2, 1, 3) * dtype(hd ** -0.5)
        k = k.reshape(B, T, nhead, hd).transpose(0, 2, 1, 3)
        v = v.reshape(B, T, nhead, hd).transpose(0, 2, 1, 3)
        s = q @ k.transpose(0, 1, 3, 2)           # (B,H,T,T)
        s = s - s.max(axis=-1, keepdims=True)
        pr = np.exp(s)
        pr = pr / pr.sum(axis=-1, keepdims=True)
        o = (pr @ v).transpose(0, 2, 1, 3).reshape(B, T, d)
        o = o @ st[p + "self_attn.out_proj.weight"].T + st[p + "self_attn.out_proj.bias"]
        h = _ln(h + o, st[p + "norm1.weight"], st[p + "norm1.bias"])
        f = np.maximum(h @ st[p + "linear1.weight"].T + st[p + "linear1.bias"], 0)
        f = f @ st[p + "linear2.weight"].T + st[p + "linear2.bias"]
        h = _ln(h + f, st[p + "norm2.weight"], st[p + "norm2.bias"])
    y = h @ st["hidden2pose_projection.weight"].T + st["hidden2pose_projection.bias"]
    return y.reshape(B, T, 21, 2).astype(np.float32)
