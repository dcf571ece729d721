#!/usr/bin/env python3
"""Generate golden input/weight/output vectors from the *reference* model.

Runs ONLY in the build container (needs /root/reference).  It imports the
reference's `ConvModel` (body2hand/src/models/HandPoseModels.py:17-64) and the
item transforms (body2hand/src/steps/utils.py:180-277,309-312) by file path,
with an in-memory stub for the absent `fairseq` package (none of its names is
used by the classes exercised here), runs them on seeded
random inputs and stores inputs, weights and outputs as small .npz files.

Nothing of the reference (source or bytecode) is copied: the .npz files hold
data only.  The committed fixtures are what pins the oracle (oracle/) and,
through it, the HIP path.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/body2hand/src"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub_fairseq():
    names = {
        "fairseq": [],
        "fairseq.utils": [],
        "fairseq.models": [],
        "fairseq.models.fairseq_encoder": ["EncoderOut"],
        "fairseq.modules": ["FairseqDropout", "LayerDropModuleList", "LayerNorm",
                            "PositionalEmbedding", "SinusoidalPositionalEmbedding",
                            "TransformerEncoderLayer"],
    }
    for mod, attrs in names.items():
        m = types.ModuleType(mod)
        for a in attrs:
            setattr(m, a, type(a, (), {}))
        sys.modules[mod] = m
    sys.modules["fairseq"].utils = sys.modules["fairseq.utils"]


def _load(path, name):
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _inputs(kind, shape, gen):
    if kind == "randn":
        return torch.randn(shape, generator=gen)
    if kind == "u01":
        return torch.rand(shape, generator=gen)
    if kind == "u55":
        return torch.rand(shape, generator=gen) - 0.5
    raise ValueError(kind)


def conv_case(hpm, name, B, T, C, pos_emb, kind, seed, keep=None):
    """One ConvModel forward.  keep=(n_head_seq) stores only some sequences of
    the output for big batches (the input is regenerated from the seed by the
    test, weights are always stored in full)."""
    torch.manual_seed(seed)
    model = hpm.ConvModel(C, "ReLU", pos_emb)
    model.eval()
    gen = torch.Generator().manual_seed(seed + 1)
    x = _inputs(kind, (B, T, 12, 2), gen)
    with torch.no_grad():
        y = model(x).contiguous()
    sd = {k.replace(".", "_"): v.numpy() for k, v in model.state_dict().items()}
    rec = dict(sd)
    rec["meta"] = np.array([B, T, C, int(pos_emb), seed], dtype=np.int64)
    rec["kind"] = np.array(kind)
    if keep is None:
        rec["x"] = x.numpy()
        rec["y"] = y.numpy()
    else:
        idx = np.array(keep, dtype=np.int64)
        rec["x"] = x.numpy()                      # inputs are small (96 B/frame)
        rec["y_idx"] = idx
        rec["y"] = y.numpy()[idx]
        # whole-output checksums in float64 so a test can check every element
        rec["y_sum"] = np.array(y.double().sum().item())
        rec["y_abs_sum"] = np.array(y.double().abs().sum().item())
        rec["y_row_sum"] = y.double().sum(dim=(1, 2, 3)).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: x{tuple(x.shape)} -> y{tuple(y.shape)}  |y|max={y.abs().max():.4f}")


def transform_case(utils, hpm, name, T, n_frames, seed):
    """Pre/post-processing around the model exactly as run.py:83-107 orders it:
    WristDifference, ChestDifference, NormalizeFixedFactor(1280),
    BuildRightHandItem -> ConvModel -> x1280 (traintest.py:387-388) ->
    mask_output (steps/utils.py:309-312)."""
    gen = torch.Generator().manual_seed(seed)
    B = len(n_frames)
    body = torch.rand((B, T, 12, 2), generator=gen) * torch.tensor([1280.0, 720.0])
    rhand = torch.rand((B, T, 21, 2), generator=gen) * torch.tensor([1280.0, 720.0])
    lhand = torch.rand((B, T, 21, 2), generator=gen) * torch.tensor([1280.0, 720.0])
    tf = [utils.WristDifference(), utils.ChestDifference(),
          utils.NormalizeFixedFactor(1280), utils.BuildRightHandItem()]
    items = []
    for b in range(B):
        item = {"body_kp": body[b].clone(), "right_hand_kp": rhand[b].clone(),
                "left_hand_kp": lhand[b].clone(),
                "body_conf": torch.ones(T, 12), "right_hand_conf": torch.ones(T, 21)}
        for t in tf:
            item = t(item)
        items.append(item)
    inp = torch.stack([it["input_kp"] for it in items])
    tgt = torch.stack([it["target_kp"] for it in items])
    torch.manual_seed(seed)
    model = hpm.ConvModel(30, "ReLU", False).eval()
    with torch.no_grad():
        pred = model(inp).contiguous()
        pred_px = pred * 1280
        masked = utils.mask_output(pred_px.clone(), n_frames)
    rec = {k.replace(".", "_"): v.numpy() for k, v in model.state_dict().items()}
    rec.update(body=body.numpy(), right_hand=rhand.numpy(), n_frames=np.array(n_frames),
               input_kp=inp.numpy(), target_kp=tgt.numpy(), pred=pred.numpy(),
               pred_px=pred_px.numpy(), pred_px_masked=masked.numpy(),
               meta=np.array([B, T, 30, 0, seed], dtype=np.int64))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: body{tuple(body.shape)} -> pred_px{tuple(masked.shape)}")


def openpose_case(tpd, utils, hpm, name, n_frames, max_frames, seed):
    """The JSON inference flow of infer_utterance.py:52-111 + traintest.py:214-300 on synthetic
    OpenPose BODY_25 frames: load_keypoints -> PoseDataset.pad/clip/to_tensor ->
    NormalizeFixedFactor, BuildRightHandItem -> ConvModel -> x1280 -> array2open_pose."""
    import json
    import types
    rng = np.random.default_rng(seed)

    def flat(n):
        kp = np.concatenate([rng.uniform(0, 1280, (n, 1)), rng.uniform(0, 720, (n, 1)),
                             rng.uniform(0, 1, (n, 1))], axis=1)
        return [float(round(v, 3)) for v in kp.reshape(-1)]          # OpenPose writes 3 decimals

    frames = [{"version": 1.3, "people": [{"person_id": [-1], "pose_keypoints_2d": flat(25),
                                           "face_keypoints_2d": [], "hand_left_keypoints_2d": flat(21),
                                           "hand_right_keypoints_2d": flat(21)}]} for _ in range(n_frames)]
    fake = types.SimpleNamespace(max_frames=max_frames)
    item = tpd.PoseDataset.load_jsons(fake, frames[:max_frames])
    item = tpd.PoseDataset.pad(fake, item)
    item = tpd.PoseDataset.clip(fake, item)
    item = tpd.PoseDataset.to_tensor(fake, item)
    staged = {k: item[k].numpy().copy() for k in ("body_kp", "body_conf", "right_hand_kp", "right_hand_conf",
                                                  "left_hand_kp", "left_hand_conf")}
    for t in (utils.NormalizeFixedFactor(1280), utils.BuildRightHandItem()):
        item = t(item)
    torch.manual_seed(seed)
    model = hpm.ConvModel(30, "ReLU", False).eval()
    with torch.no_grad():
        pred = model(item["input_kp"].unsqueeze(0))[0].contiguous()
    pred = pred * 1280
    pred = pred.numpy()
    out_hands = [utils.array2open_pose(pred[i]) for i in range(min(n_frames, max_frames))]
    rec = {k.replace(".", "_"): v.numpy() for k, v in model.state_dict().items()}
    rec.update({"staged_" + k: v for k, v in staged.items()})
    rec.update(pred_px=pred, meta=np.array([1, max_frames, 30, 0, seed], dtype=np.int64),
               n_frames=np.array(n_frames), frames_json=np.array(json.dumps(frames)),
               out_hands_json=np.array(json.dumps(out_hands)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: {n_frames} frames -> max_frames {max_frames}, {len(out_hands)} hands written")


def metric_case(utils, name, B, T, lengths, seed):
    """maskedPoseL1 + L12Pixels (steps/utils.py:413-428,291-299)."""
    gen = torch.Generator().manual_seed(seed)
    pred = torch.rand((B, T, 21, 2), generator=gen) - 0.5
    tgt = torch.rand((B, T, 21, 2), generator=gen) - 0.5
    loss = utils.maskedPoseL1()(pred, tgt, lengths)
    per_seq = torch.stack([torch.nn.functional.l1_loss(pred[i, :n], tgt[i, :n]) for i, n in enumerate(lengths)])
    pix = utils.L12Pixels(21, 1280)(loss)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), pred=pred.numpy(), target=tgt.numpy(),
                        lengths=np.array(lengths, dtype=np.int64), loss=loss.numpy(), per_seq=per_seq.numpy(),
                        pixels=pix.numpy(), meta=np.array([B, T, 0, 0, seed], dtype=np.int64))
    print(f"{name}: loss {float(loss):.6f}  pixels {float(pix):.4f}")


def weighted_metric_case(utils, name, B, T, lengths, seed):
    """poderatedPoseL1 (`--loss confL1`, steps/utils.py:431-452): per utterance the mean of
    |pred * score - target * score| over its first n frames, SUMMED over the batch (the class
    does not divide by the batch size)."""
    gen = torch.Generator().manual_seed(seed)
    pred = torch.rand((B, T, 21, 2), generator=gen) - 0.5
    tgt = torch.rand((B, T, 21, 2), generator=gen) - 0.5
    scores = torch.rand((B, T, 21), generator=gen)
    loss = utils.poderatedPoseL1()(pred, tgt, lengths, scores)
    per_seq = torch.stack([torch.nn.functional.l1_loss(pred[i, :n] * scores[i, :n].unsqueeze(2),
                                                       tgt[i, :n] * scores[i, :n].unsqueeze(2))
                           for i, n in enumerate(lengths)])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), pred=pred.numpy(), target=tgt.numpy(), scores=scores.numpy(),
                        lengths=np.array(lengths, dtype=np.int64), loss=loss.numpy(), per_seq=per_seq.numpy(),
                        meta=np.array([B, T, 0, 0, seed], dtype=np.int64))
    print(f"{name}: confL1 loss {float(loss):.6f}")


def validate_case(utils, tt, hpm, name, seed):
    """The evaluation loop itself, `validate(model, val_loader, criterion, device, args)`
    (steps/traintest.py:168-213), on a three-batch synthetic loader, for both text-free models and both
    losses the loop supports (`--loss L1` / `confL1`).  Batches as the reference's collate produces them
    after the item transforms: body_kp = input_kp (B,T,12,2), target_kp (B,T,21,2), n_frames, target_conf."""
    import types as _types
    gen = torch.Generator().manual_seed(seed)
    shapes = ((4, 100, [100, 37, 1, 64]), (3, 100, [99, 100, 12]), (2, 100, [50, 77]))
    batches = []
    for B, T, nf in shapes:
        body = torch.rand((B, T, 12, 2), generator=gen) - 0.5
        batches.append({"body_kp": body, "input_kp": body.clone(), "target_kp": torch.rand((B, T, 21, 2), generator=gen) - 0.5,
                        "n_frames": list(nf), "target_conf": torch.rand((B, T, 21), generator=gen)})
    torch.manual_seed(seed)
    conv = hpm.ConvModel(30, "ReLU", False).eval()
    torch.manual_seed(41)
    tenc = hpm.TransformerEnc(ninp=12 * 2, nhead=4, nhid=128, nout=21 * 2, nlayers=4, dropout=0.5).eval()  # = tenc_weights.npz
    rec = {k.replace(".", "_"): v.numpy() for k, v in conv.state_dict().items()}
    for i, b in enumerate(batches):
        rec.update({f"b{i}_body_kp": b["body_kp"].numpy(), f"b{i}_target_kp": b["target_kp"].numpy(),
                    f"b{i}_n_frames": np.array(b["n_frames"], dtype=np.int64), f"b{i}_target_conf": b["target_conf"].numpy()})
    for mname, model in (("Conv", conv), ("TransformerEnc", tenc)):
        for lname, crit in (("L1", utils.maskedPoseL1()), ("confL1", utils.poderatedPoseL1())):
            args = _types.SimpleNamespace(model=mname, loss=lname)
            loader = [{k: (v.clone() if torch.is_tensor(v) else list(v)) for k, v in b.items()} for b in batches]
            val = tt.validate(model, loader, crit, torch.device("cpu"), args)
            rec[f"loss_{mname}_{lname}"] = np.array(val, dtype=np.float64)
            print(f"{name}: validate({mname}, {lname}) = {val:.6f}")
    rec["meta"] = np.array([len(batches), 100, 30, 0, seed], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)


def tenc_cases(hpm):
    """TransformerEnc (HandPoseModels.py:118-178) as the CLIs build it (infer_utterance.py:99-101).
    One seeded model (weights stored once in tenc_weights.npz), several inputs."""
    torch.manual_seed(41)
    model = hpm.TransformerEnc(ninp=12 * 2, nhead=4, nhid=128, nout=21 * 2, nlayers=4, dropout=0.5).eval()
    np.savez_compressed(os.path.join(OUT, "tenc_weights.npz"),
                        **{"sd__" + k: v.numpy() for k, v in model.state_dict().items()})
    rec = {}
    for name, B, T, kind, seed in (("b2_t100", 2, 100, "u55", 1), ("b3_t37", 3, 37, "u01", 2),
                                   ("b1_t1", 1, 1, "randn", 3), ("b5_t16", 5, 16, "randn", 4),
                                   ("b2_t17", 2, 17, "u55", 5)):
        x = _inputs(kind, (B, T, 12, 2), torch.Generator().manual_seed(seed))
        with torch.no_grad():
            y = model(x).contiguous()
        rec["x_" + name], rec["y_" + name] = x.numpy(), y.numpy()
        print(f"tenc {name}: x{tuple(x.shape)} -> y{tuple(y.shape)} |y|max={y.abs().max():.4f}")
    np.savez_compressed(os.path.join(OUT, "tenc_cases.npz"), **rec)
    print("tenc params", sum(v.numel() for v in model.parameters()))


def tenc_transform_case(utils, hpm, name, T, n_frames, seed):
    """The item transforms around TransformerEnc exactly as run.py:83-107 / infer_utterance.py order
    them for `--model TransformerEnc --dif-encoding`: WristDifference, ChestDifference,
    NormalizeFixedFactor(1280), BuildRightHandItem (steps/utils.py:180-277) -> TransformerEnc
    (HandPoseModels.py:152-178; the model of tenc_weights.npz, seed 41) -> x1280
    (traintest.py:270-271) -> mask_output (steps/utils.py:309-312)."""
    gen = torch.Generator().manual_seed(seed)
    B = len(n_frames)
    body = torch.rand((B, T, 12, 2), generator=gen) * torch.tensor([1280.0, 720.0])
    rhand = torch.rand((B, T, 21, 2), generator=gen) * torch.tensor([1280.0, 720.0])
    lhand = torch.rand((B, T, 21, 2), generator=gen) * torch.tensor([1280.0, 720.0])
    torch.manual_seed(41)
    model = hpm.TransformerEnc(ninp=12 * 2, nhead=4, nhid=128, nout=21 * 2, nlayers=4, dropout=0.5).eval()
    rec = {"body": body.numpy(), "n_frames": np.array(n_frames)}
    for tag, tf in (("dif", [utils.WristDifference(), utils.ChestDifference(), utils.NormalizeFixedFactor(1280),
                             utils.BuildRightHandItem()]),
                    ("nodif", [utils.NormalizeFixedFactor(1280), utils.BuildRightHandItem()])):
        items = []
        for b in range(B):
            item = {"body_kp": body[b].clone(), "right_hand_kp": rhand[b].clone(), "left_hand_kp": lhand[b].clone(),
                    "body_conf": torch.ones(T, 12), "right_hand_conf": torch.ones(T, 21)}
            for t in tf:
                item = t(item)
            items.append(item)
        inp = torch.stack([it["input_kp"] for it in items])
        with torch.no_grad():
            pred = model(inp).contiguous()
            pred_px = pred * 1280
            masked = utils.mask_output(pred_px.clone(), n_frames)
        rec.update({tag + "_input_kp": inp.numpy(), tag + "_pred": pred.numpy(), tag + "_pred_px": pred_px.numpy(),
                    tag + "_pred_px_masked": masked.numpy()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: body{tuple(body.shape)} -> pred_px{tuple(masked.shape)} |px|max={masked.abs().max():.2f}")


def wire_formats_case(tpd, tt, name, seed):
    """The other two wire formats of SURVEY 8f N2, from the reference's own functions:
    merged JSON (frames as {"json_path", "json_data"} entries: How2Sign/util_scripts/build_dataset.py:66-72,
    consumed by select_jsons + FastTextPoseDataset.load_jsons, text_pose_dataset.py:52-68,478-505) and the HDF5
    row (n_frames, 150) = [x*50 | y*50 | c*50], joints 8 body | 21 left | 21 right
    (TextPoseH5Dataset.array2item / pad / clip, text_pose_dataset.py:587-632; writer-side
    order_and_reshape_toh5, steps/traintest.py:302-317)."""
    import json
    import types
    rng = np.random.default_rng(seed)

    def flat(n):
        kp = np.concatenate([rng.uniform(0, 1280, (n, 1)), rng.uniform(0, 720, (n, 1)),
                             rng.uniform(0, 1, (n, 1))], axis=1)
        return [float(round(v, 3)) for v in kp.reshape(-1)]

    frames = [{"version": 1.3, "people": [{"person_id": [-1], "pose_keypoints_2d": flat(25),
                                           "face_keypoints_2d": [], "hand_left_keypoints_2d": flat(21),
                                           "hand_right_keypoints_2d": flat(21)}]} for _ in range(9)]
    merged = [{"json_path": f"/data/utt_0/frame_{i:012d}_keypoints.json", "json_data": fr} for i, fr in enumerate(frames)]
    rec = {"merged_json": np.array(json.dumps(merged))}
    for n, sel in ((20, None), (9, "first"), (5, "first")):
        chosen, start = tpd.select_jsons(merged, n, selection_type=sel)
        item = tpd.FastTextPoseDataset.load_jsons(None, chosen)   # the variant that reads "json_data" entries (:478-505)
        tag = f"sel{n}"
        rec[tag + "_start"] = np.array(start)
        rec[tag + "_paths"] = np.array(json.dumps(item["json_paths"]))
        for k in ("body_kp", "body_conf", "right_hand_kp", "right_hand_conf", "left_hand_kp", "left_hand_conf"):
            rec[tag + "_" + k] = np.asarray(item[k], dtype=np.float64)
    # HDF5 row
    row = rng.uniform(0, 1280, (7, 150)).astype(np.float32)
    rec["h5_row"] = row
    for mf in (12, 5):
        fake = types.SimpleNamespace(max_frames=mf)
        item = tpd.TextPoseH5Dataset.array2item(fake, row.copy())
        item = tpd.TextPoseH5Dataset.pad(fake, item)
        item = tpd.TextPoseH5Dataset.clip(fake, item)
        for k, v in item.items():
            rec[f"h5_m{mf}_{k}"] = np.asarray(v)
    hand = torch.from_numpy(rng.uniform(0, 1280, (6, 21, 2)).astype(np.float32))
    rec["h5w_in"] = hand.numpy()
    rec["h5w_out"] = np.asarray(tt.order_and_reshape_toh5(hand))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: merged {len(merged)} frames, h5 row {row.shape}, h5 writer {rec['h5w_out'].shape}")


def predict_variants_case(utils, name, seed):
    """The other two `--predict` choices (run.py:56-60): the reference's item builders
    BuildIndexItem / Build3fingerItem (steps/utils.py:215-259) and output writers
    array2open_pose_index / array2open_pose_3finger (steps/utils.py:341-353, 366-381) on synthetic data."""
    import json
    gen = torch.Generator().manual_seed(seed)
    T = 7
    body = torch.rand((T, 12, 2), generator=gen) * torch.tensor([1280.0, 720.0])
    rhand = torch.rand((T, 21, 2), generator=gen) * torch.tensor([1280.0, 720.0])
    rec = {"body": body.numpy(), "right_hand": rhand.numpy()}
    for tag, cls in (("right_index", utils.BuildIndexItem), ("right_3fingers", utils.Build3fingerItem)):
        item = cls()({"body_kp": body.clone(), "right_hand_kp": rhand.clone()})
        rec[tag + "_input_kp"] = item["input_kp"].numpy()
        rec[tag + "_target_kp"] = item["target_kp"].numpy()
    rng = np.random.default_rng(seed)
    hand_list = [float(round(v, 3)) for v in rng.uniform(0, 1280, 63)]
    pred4 = rng.uniform(0, 1280, (4, 2)).astype(np.float32)
    pred12 = rng.uniform(0, 1280, (12, 2)).astype(np.float32)
    rec.update(hand_list=np.array(hand_list), pred4=pred4, pred12=pred12,
               out_index=np.array(utils.array2open_pose_index(list(hand_list), pred4)),
               out_3finger=np.array(utils.array2open_pose_3finger(list(hand_list), pred12)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: index input {rec['right_index_input_kp'].shape}, 3finger input {rec['right_3fingers_input_kp'].shape}")


def _load_traintest(utils):
    """steps/traintest.py imports its siblings relatively (`from .utils import ...`): give it an
    in-memory parent package whose `utils` is the module already loaded from steps/utils.py."""
    pkg = types.ModuleType("ref_steps")
    pkg.__path__ = [os.path.join(REF, "steps")]
    sys.modules["ref_steps"] = pkg
    sys.modules["ref_steps.utils"] = utils
    spec = importlib.util.spec_from_file_location("ref_steps.traintest", os.path.join(REF, "steps", "traintest.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_steps.traintest"] = mod
    spec.loader.exec_module(mod)
    return mod


def _stub_io_deps():
    """text_pose_dataset.py imports h5py at module top (absent here, unused by PoseDataset)."""
    if "h5py" not in sys.modules:
        sys.modules["h5py"] = types.ModuleType("h5py")


def main():
    _stub_fairseq()
    hpm = _load(os.path.join(REF, "models", "HandPoseModels.py"), "ref_HandPoseModels")
    utils = _load(os.path.join(REF, "steps", "utils.py"), "ref_steps_utils")

    # BASELINE.json config 1: single sequence / single frame, CPU plumbing
    conv_case(hpm, "cfg1_b1_t200", 1, 200, 30, False, "randn", 0)
    conv_case(hpm, "cfg1_b1_t1", 1, 1, 30, False, "randn", 1)
    # edge lengths around the 5-tap kernel and the 17-frame receptive field
    for T in (2, 3, 5, 8, 16, 17, 18, 33, 199, 201, 257, 600):
        conv_case(hpm, f"edge_b3_t{T}", 3, T, 30, False, "u55", 100 + T)
    # positional-embedding branch (requires T == 100)
    conv_case(hpm, "posemb_b2_t100", 2, 100, 30, True, "u01", 7)
    # other widths
    for C in (8, 16, 32, 64, 128):
        conv_case(hpm, f"width_c{C}_b2_t50", 2, 50, C, False, "randn", 200 + C)
    # BASELINE.json config 2: batch=64, three input distributions
    for kind in ("randn", "u01", "u55"):
        conv_case(hpm, f"cfg2_b64_t200_{kind}", 64, 200, 30, False, kind, 0,
                  keep=[0, 1, 31, 62, 63])
    # pre/post-processing
    transform_case(utils, hpm, "transforms_b6_t40", 40, [40, 1, 17, 39, 25, 8], 11)
    # TransformerEnc
    tenc_cases(hpm)
    tenc_transform_case(utils, hpm, "tenc_transforms_b6_t40", 40, [40, 1, 17, 39, 25, 8], 13)
    # evaluation metric
    metric_case(utils, "metric_b5_t60", 5, 60, [60, 1, 33, 59, 17], 31)
    weighted_metric_case(utils, "metric_conf_b5_t60", 5, 60, [60, 1, 33, 59, 17], 37)
    # OpenPose JSON wire format + utterance staging
    _stub_io_deps()
    tpd = _load(os.path.join(REF, "dataloaders", "text_pose_dataset.py"), "ref_text_pose_dataset")
    openpose_case(tpd, utils, hpm, "openpose_short_n7_m12", 7, 12, 21)
    openpose_case(tpd, utils, hpm, "openpose_long_n30_m20", 30, 20, 22)
    # merged JSON + HDF5 row
    tt = _load_traintest(utils)
    wire_formats_case(tpd, tt, "wire_formats", 23)
    # the evaluation loop around the models (traintest.py:168-213), both losses it supports
    validate_case(utils, tt, hpm, "validate_loop", 43)
    # --predict right_index / right_3fingers item builders and writers
    predict_variants_case(utils, "predict_variants", 29)


if __name__ == "__main__":
    main()
