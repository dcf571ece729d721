"""Drop-in mirror of the reference's `TransformerEnc` (body2hand/src/models/HandPoseModels.py:
118-178), the second text-free body->hand model, on libb2h's gfx950 kernels.

Same constructor `TransformerEnc(ninp, nhead, nhid, nout, nlayers, dropout=0.5)` and the same
`state_dict` (the torch.nn containers are built in the reference's order, so a seeded default
init is identical and its checkpoints load as they are); `model(src)` with src float32
(B, T, 12, 2) -> float32 (B, T, 21, 2).  The torch containers only hold parameters: the forward
runs through the C ABI (`b2h_tenc_forward`).  Inference only; only the geometry the reference's
CLIs construct (ninp=24, nhead=4, nhid=128, nout=42; infer_utterance.py:99-101) is implemented.
"""
import ctypes
import math
import warnings

import torch
import torch.nn as nn

from . import _lib


class PositionalEncoding(nn.Module):
    """Sinusoidal table added to the (T, B, d_model) input; registered as buffer `pe` of shape
    (max_len, 1, d_model) so that it is part of the state_dict like the reference's
    (HandPoseModels.py:86-103).  pe[p, 2i] = sin(p w_i), pe[p, 2i+1] = cos(p w_i),
    w_i = 10000^(-2i/d_model)."""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        freq = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        angle = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1) * freq
        table = torch.zeros(max_len, d_model)
        table[:, 0::2] = torch.sin(angle)
        table[:, 1::2] = torch.cos(angle)
        self.register_buffer("pe", table.unsqueeze(0).transpose(0, 1))


TENC_KERNELS = {"fp32": 0, "f16x3": 1}   # b2h_tenc_kernel (include/b2h.h)


class TransformerEnc(nn.Module):
    """`precision` (not in the reference; keyword only): "fp32" = fp32 operands on the matrix cores
    (default); "f16x3" = every Linear operand split into f16 hi + lo, three f16 MFMAs per product
    with fp32 accumulation: fp32-grade error at 3/16 of the matrix cycles, for activations and
    weights inside the f16 range (|x| < 65504)."""

    def __init__(self, ninp, nhead, nhid, nout, nlayers, dropout=0.5, *, precision="fp32"):
        super().__init__()
        if precision not in TENC_KERNELS:
            raise ValueError(f"precision must be one of {sorted(TENC_KERNELS)}, got {precision!r}")
        self.precision = precision
        self.model_type = "Transformer"
        self.src_mask = None
        self.pos_encoder = PositionalEncoding(ninp, dropout, max_len=100)
        encoder_layers = nn.TransformerEncoderLayer(nhid, nhead, nhid, dropout)
        with warnings.catch_warnings():  # torch notes that seq-first layers skip its nested-tensor path
            warnings.simplefilter("ignore", UserWarning)
            self.transformer_encoder = nn.TransformerEncoder(encoder_layers, nlayers)
        self.ninp = ninp
        self.hidden2pose_projection = nn.Linear(nhid, nout)
        self.pose2hidden_projection = nn.Linear(ninp, nhid)
        self._geom = (int(ninp), int(nhead), int(nhid), int(nout), int(nlayers))
        self._handle = None
        self._packed_key = None
        self._workspace = None

    def _tensors(self):
        # through the module dictionaries (53 tensors; nn.Module.__getattr__ would cost ~50 us per forward)
        M = self._modules
        p2h = M["pose2hidden_projection"]._parameters
        t = [M["pos_encoder"]._buffers["pe"], p2h["weight"], p2h["bias"]]
        for layer in M["transformer_encoder"]._modules["layers"]._modules.values():
            lm = layer._modules
            sa = lm["self_attn"]
            for owner, names in ((sa._parameters, ("in_proj_weight", "in_proj_bias")),
                                 (sa._modules["out_proj"]._parameters, ("weight", "bias")),
                                 (lm["linear1"]._parameters, ("weight", "bias")), (lm["linear2"]._parameters, ("weight", "bias")),
                                 (lm["norm1"]._parameters, ("weight", "bias")), (lm["norm2"]._parameters, ("weight", "bias"))):
                t.append(owner[names[0]])
                t.append(owner[names[1]])
        h2p = M["hidden2pose_projection"]._parameters
        t.append(h2p["weight"])
        t.append(h2p["bias"])
        return t

    def _ensure_handle(self):
        dev = self._modules["pose2hidden_projection"]._parameters["weight"].device
        if dev.type != "cuda":
            raise RuntimeError("hand_pose_sl_amd.TransformerEnc runs on an MI355X only: call model.to('cuda') "
                               "first (there is no CPU path in the product)")
        lib = _lib.load()
        tensors = self._tensors()
        key = (dev.index,) + tuple((p.data_ptr(), p._version) for p in tensors)
        if self._handle is not None and key == self._packed_key:
            return lib
        with torch.cuda.device(dev):
            if self._handle is None or self._packed_key[0] != dev.index:
                self._free()
                h = ctypes.c_void_p()
                _lib.check(lib.b2h_tenc_create(*self._geom, int(self.pos_encoder.pe.shape[0]), ctypes.byref(h)))
                self.__dict__["_handle"] = h
            ps = [p.detach().to(torch.float32).contiguous() for p in tensors]
            torch.cuda.current_stream(dev).synchronize()
            arr = (ctypes.c_void_p * len(ps))(*[p.data_ptr() for p in ps])
            _lib.check(lib.b2h_tenc_load_weights(self._handle, arr, len(ps), 1))
        self.__dict__["_packed_key"] = key
        return lib

    def _free(self):
        if self.__dict__.get("_handle") is not None:
            try:
                _lib.load().b2h_tenc_destroy(self._handle)
            except Exception:
                pass
            self.__dict__["_handle"] = None
            self.__dict__["_packed_key"] = None

    def __del__(self):
        self._free()

    def _run(self, src, flags, factor, n_frames):
        lib = self._ensure_handle()
        if src.dim() != 4 or src.shape[2] * src.shape[3] != self.ninp:
            raise RuntimeError(f"expected input of shape (B, T, {self.ninp // 2}, 2), got {tuple(src.shape)}")
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("hand_pose_sl_amd.TransformerEnc is inference-only: call model.eval() and wrap "
                               "the call in torch.no_grad()")
        dev = self._modules["pose2hidden_projection"]._parameters["weight"].device
        x = src.to(device=dev, dtype=torch.float32, non_blocking=True).contiguous()
        B, T = x.shape[0], x.shape[1]
        nf = None
        if flags & _lib.POST_MASK_TAIL:
            if n_frames is None:
                raise ValueError("mask_tail needs n_frames")
            nf = torch.as_tensor(n_frames).to(device=dev, dtype=torch.int64).contiguous()
            if nf.shape != (B,):
                raise RuntimeError(f"n_frames must have shape ({B},)")
        y = torch.empty((B, T, 21, 2), dtype=torch.float32, device=dev)
        need = lib.b2h_tenc_workspace_bytes(self._handle, B, T)
        ws = self.__dict__.get("_workspace")
        if ws is None or ws.numel() < need or ws.device != dev:
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
            self.__dict__["_workspace"] = ws
        with _lib.on_device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.b2h_tenc_set_kernel(self._handle, TENC_KERNELS[self.precision]))
            if flags == 0:
                _lib.check(lib.b2h_tenc_forward(self._handle, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                                                B, T, ctypes.c_void_p(ws.data_ptr()), ws.numel(), ctypes.c_void_p(st)))
            else:
                _lib.check(lib.b2h_tenc_forward_fused(self._handle, ctypes.c_void_p(x.data_ptr()),
                                                      ctypes.c_void_p(y.data_ptr()), B, T, flags, float(factor),
                                                      ctypes.c_void_p(nf.data_ptr()) if nf is not None else None,
                                                      ctypes.c_void_p(ws.data_ptr()), ws.numel(), ctypes.c_void_p(st)))
        return y

    def forward(self, src):
        return self._run(src, 0, 1.0, None)

    def forward_fused(self, body, n_frames=None, dif_encoding=True, normalize=True, denormalize=True,
                      mask_tail=False, factor=1280.0):
        """Raw-pixel body keypoints in, pixel-space hand keypoints out, with the item transforms inside
        the model's own first and last kernel: ChestDifference + /factor (steps/utils.py:180-210) on the
        rows as they enter (before the positional encoding, HandPoseModels.py:167) -> the encoder ->
        x factor (traintest.py:270-271) and the optional tail mask (utils.py:309-312) in the store of
        hidden2pose_projection's output.  Same flags as ConvModel.forward_fused."""
        flags = ((_lib.PRE_CHEST_DIFF if dif_encoding else 0) | (_lib.PRE_NORMALIZE if normalize else 0) |
                 (_lib.POST_DENORMALIZE if denormalize else 0) | (_lib.POST_MASK_TAIL if mask_tail else 0))
        return self._run(body, flags, factor, n_frames)
