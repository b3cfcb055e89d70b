import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the GPU box advertises every host core but grants a 16-core share; oversubscribing torch's
# intra-op pool makes the CPU oracle ~100x slower
torch.set_num_threads(min(8, os.cpu_count() or 1))

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def base_kwargs(**over):
    kw = dict(src_pad_idx=0, n_enc_exits=6, enc_voc_size=256, dec_voc_size=256, d_model=256, n_head=8, max_len=2000,
              d_feed_forward=2048, n_enc_layers=2, features_length=80, drop_prob=0.1, depthwise_kernel_size=31,
              device="cpu")
    kw.update(over)
    return kw


def load_golden(name):
    import numpy as np
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    kw = base_kwargs(**eval(str(z["kwargs"])))  # the dict literal written by make_golden.py
    return z, kw
