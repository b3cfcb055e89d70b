"""CPU restatement of the CTC prefix beam search the reference's CTC inference runs (the oracle for csrc/ctc_beam.hip).

TEST INFRASTRUCTURE -- see oracle/__init__.py for who may import this.

Reference call site: util/beam_infer.py:79-80,102-112 (``ctc_cuda_predict``):
    cuda_ctc_decoder(tokens, nbest=1, beam_size=args.beam_size (10), blank_skip_threshold=0.95)(log_probs [B, T', V], lengths = T')
``torchaudio.models.decoder.cuda_ctc_decoder`` is third-party CUDA code that is neither in the reference tree nor installed
(version unpinned): PARITY UNPINNED.  What is restated is the published algorithm -- CTC prefix beam search without a
language model (Graves 2012 / Hannun 2014): every prefix carries log p(ending in blank) and log p(ending in a label);
per frame each prefix stays (blank, or repeat of its last label) or is extended by a label; extensions that spell an
existing prefix merge into it; the ``beam`` best prefixes by total probability survive; blank is label 0.
``blank_skip_threshold``: a frame whose blank probability exceeds the threshold is not expanded; it is taken as a blank
frame (all mass moves to "ending in blank", so a label repeated across the skipped frame stays a repeat).  Ties between
equal scores resolve to the lower candidate id (stays before extensions, then beam index, then label)."""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np

NEG = -float("inf")


def _lae(a: float, b: float) -> float:
    if a == NEG:
        return b
    if b == NEG:
        return a
    m = max(a, b)
    return m + math.log1p(math.exp(-abs(a - b)))


def ctc_prefix_beam_search(logp: np.ndarray, beam: int = 10, blank: int = 0, blank_skip_threshold: float = 0.95, return_beams: bool = False,
                           skip_drops_frame: bool = False):
    """logp [T', V] natural-log probabilities -> (best prefix, its total log-probability); ``return_beams``: additionally the
    final beam as [(prefix, total log-probability)] sorted best first.  ``skip_drops_frame``: the other reading of the skip
    rule -- a frame above the threshold is removed from the sequence (a label repeated across it collapses) instead of being
    taken as a blank frame."""
    T, V = logp.shape
    log_thr = math.log(blank_skip_threshold) if 0.0 < blank_skip_threshold < 1.0 else 0.0
    beams = [((), 0.0, NEG)]  # (prefix, log p_blank, log p_nonblank)
    for t in range(T):
        lp = logp[t].astype(np.float64)
        lpb = float(lp[blank])
        if blank_skip_threshold < 1.0 and lpb > log_thr:
            if not skip_drops_frame:
                beams = [(p, _lae(pb, pnb) + lpb, NEG) for p, pb, pnb in beams]
            continue
        index = {p: i for i, (p, _, _) in enumerate(beams)}
        stay_pb = [_lae(pb, pnb) + lpb for _, pb, pnb in beams]
        stay_pnb = [(pnb + float(lp[p[-1]])) if p else NEG for p, _, pnb in beams]
        cands = []  # (score, id, prefix, pb, pnb)
        for i, (p, pb, pnb) in enumerate(beams):
            tot = _lae(pb, pnb)
            for c in range(V):
                if c == blank:
                    continue
                v = (pb if (p and p[-1] == c) else tot) + float(lp[c])
                q = p + (c,)
                j = index.get(q)
                if j is not None:
                    stay_pnb[j] = _lae(stay_pnb[j], v)  # the extension spells an existing prefix
                else:
                    cands.append((v, 16 + c * 16 + i, q, NEG, v))
        for i, (p, _, _) in enumerate(beams):
            cands.append((_lae(stay_pb[i], stay_pnb[i]), i, p, stay_pb[i], stay_pnb[i]))
        cands = [c for c in cands if c[0] > NEG]
        cands.sort(key=lambda c: (-c[0], c[1]))
        beams = [(c[2], c[3], c[4]) for c in cands[:beam]]
    best = max(range(len(beams)), key=lambda i: (_lae(beams[i][1], beams[i][2]), -i))
    if return_beams:
        final = sorted(((list(p), _lae(pb, pnb)) for p, pb, pnb in beams), key=lambda e: -e[1])
        return list(beams[best][0]), _lae(beams[best][1], beams[best][2]), final
    return list(beams[best][0]), _lae(beams[best][1], beams[best][2])
