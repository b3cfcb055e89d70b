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


def ref_decoder_logits(model, trg, enc, idx):
    """TEST-SIDE reference of the AED decoder: the reference's arithmetic (early_exit.py:739-762, 776-790) through the torch
    modules that hold the product class's parameters -- embedding, positional encoding, nn.TransformerDecoder (norm_first, causal
    + target-padding masks, shared final LayerNorm), linears_2[idx] -- on whatever device they are (the CPU in the tests).  Raw
    logits.  The product itself never runs these modules (it has no CPU path)."""
    sz = trg.size(1)
    tgt_mask = torch.triu(torch.full((sz, sz), float("-inf"), device=trg.device), diagonal=1)
    t = model.positional_encoder_2(model.emb(trg))
    return model.linears_2[idx](model.decoders[idx](t, enc, tgt_mask=tgt_mask, tgt_key_padding_mask=trg == model.trg_pad_idx))


def ref_decoder_logprobs(model, trg, enc, layer_n):
    """``_decoder_`` of the reference (log-probs; layer_n counts from 1, out-of-range values select the last exit)."""
    n = model.n_enc_exits
    idx = (int(layer_n) if 1 <= int(layer_n) <= n else n) - 1
    return torch.log_softmax(ref_decoder_logits(model, trg, enc, idx), dim=2)
