"""Helpers shared by the AED fixtures' generator (make_golden.py, build container only) and the tests that replay them (GPU box):
the portable weights of a whole full_conformer and the reference's decode length.  No reference import, no restated algorithm:
the generator stays out of every test that runs on the GPU box."""
from early_exit_transformer_amd import synth


def aed_state_dict(model, seed):
    """Portable weights for the WHOLE full_conformer (encoder + AED decoder).  The decoders' final norm is ONE shared
    LayerNorm instance (early_exit.py:666,701-717): its aliases get the same values.  ``linears_2`` x 8: peaky decoder
    outputs, so that greedy tokens are stable under a 1e-3 log-prob tolerance."""
    sd = synth.synth_state_dict(model.state_dict(), seed=seed, style="trained")
    for k in list(sd):
        if ".norm." in k and k.startswith("decoders."):
            sd[k] = sd["layer_norm." + k.rsplit(".", 1)[1]].clone()
        if k.startswith("linears_2.") and k.endswith("weight"):
            sd[k] = sd[k] * 8.0
    return sd


def aed_max_length(T):
    """inference.py:31-39 (p = 30, m = 5/200)."""
    return int(30 - T * 5 / 200) if T < 200 else int(T / 12)
