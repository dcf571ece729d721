import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU is a hard error, not a silent skip
    return


def golden_names(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    rec = {k: d[k] for k in d.files}
    B, T, C, pe, seed = [int(v) for v in rec["meta"]]
    rec.update(B=B, T=T, C=C, pos_emb=bool(pe), seed=seed)
    rec["state"] = {k.replace("_", ".", 1): rec[k] for k in list(rec) if k.startswith("conv")}
    return rec


CONV_CASES = [n for n in golden_names() if not n.startswith(("transforms", "openpose", "metric", "tenc", "wire_formats", "predict_variants", "validate"))]


@pytest.fixture(scope="session")
def cuda_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu but no GPU is visible")
    return torch.device("cuda:0")
