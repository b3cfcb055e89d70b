"""Decoders on top of the HIP encoder: mirrors of the reference's ``util/beam_infer.py`` pieces that this package can
serve without torchaudio.

* ``GreedyCTCDecoder``            util/beam_infer.py:9-24, on the batched HIP kernel (``eec_greedy_ctc``).
* ``BeamInference.beam_search``   util/beam_infer.py:198-307: the AED beam search that ``inference.py:44-51`` drives per
  utterance and exit.  Same signature, same arithmetic and the same quirks (the length penalty DIVIDES the step's
  log-probs; ``min_length`` defaults to 300, so EOS never finalises a beam inside ``max_length`` steps; the best beam
  is the one with the highest accumulated score), but the candidate bookkeeping runs as tensor ops on the device
  instead of a Python loop over beams, and ``decode_all_exits`` runs the encoder ONCE for all exits (one
  ``eec_encoder_forward`` with taps) where the reference re-runs the first n exit groups for every n
  (inference.py:44-46: O(E^2) groups per utterance), and the decoder advances step-wise over a key / value cache
  (``model.decoder_session``, csrc/decoder_step.hip) where the reference re-runs it on the whole prefix per step.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from .model import beam_select, ctc_beam_decode, greedy_ctc


class GreedyCTCDecoder(torch.nn.Module):
    """``forward(emission[T', V]) -> List[int]``: argmax, collapse repeats, drop blank (util/beam_infer.py:9-24)."""

    def __init__(self, blank: int = 0):
        super().__init__()
        self.blank = blank

    def forward(self, emission: Tensor) -> List[int]:
        tokens, counts = greedy_ctc(emission.unsqueeze(0), self.blank)
        return tokens[0, : int(counts[0])].tolist()


def sequence_length_penalty(length: int, alpha: float = 0.6) -> float:
    """util/beam_infer.py:194-195."""
    return ((5 + length) / (5 + 1)) ** alpha


class CTCHypothesis:
    """One CTC beam-search hypothesis, with the attribute names of torchaudio's ``CUCTCHypothesis`` (tokens: List[int] without
    blanks / repeats, words: List[str] -- empty here, as for a lexicon-free decoder --, score: float)."""
    __slots__ = ("tokens", "words", "score")

    def __init__(self, tokens, words, score):
        self.tokens, self.words, self.score = tokens, words, score

    def __repr__(self):
        return f"CTCHypothesis(tokens={self.tokens}, words={self.words}, score={self.score:.4f})"


class BeamInference:
    """``args`` needs ``dec_voc_size, trg_sos_idx, trg_eos_idx, trg_pad_idx, beam_size, pen_alpha, device`` (the fields
    util/conf.py:455-486 injects); every one of them can also be given per call, as in the reference."""

    def __init__(self, args=None):
        self.args = args

    sequence_length_penalty = staticmethod(sequence_length_penalty)

    def ctc_cuda_predict(self, emission: Tensor, tokens=None, beam_size: Optional[int] = None) -> List[List["CTCHypothesis"]]:
        """util/beam_infer.py:102-112: the nbest (= 1) beam-search hypotheses of one exit's log-probs ``emission`` [B, T', V],
        input length T' for every utterance, beam ``args.beam_size``, blank_skip_threshold 0.95 -- per utterance a list of
        hypothesis objects with ``.tokens`` / ``.words`` / ``.score`` like torchaudio's, so the reference's call sites
        (``best[0][0].tokens`` train.py:82, ``best_[0].tokens`` inference.py:70) work unchanged.  ``tokens`` (the token file
        the torchaudio decoder takes) is accepted and unused: ids are returned, blank = 0."""
        beam = self._arg(beam_size, "beam_size")
        tok, cnt, score = ctc_beam_decode(emission, beam_size=beam, blank=0, blank_skip_threshold=0.95)
        tok, cnt, score = tok.cpu(), cnt.cpu(), score.cpu()
        return [[CTCHypothesis(tok[b, : int(cnt[b])].tolist(), [], float(score[b]))] for b in range(tok.size(0))]

    def _arg(self, value, name):
        if value is not None:
            return value
        if self.args is None or not hasattr(self.args, name):
            raise ValueError(f"beam_search: {name} was not given and there is no args.{name}")
        return getattr(self.args, name)

    @torch.no_grad()
    def beam_search(self, model, encoder_output: Tensor, layer_n: int, vocab_size: Optional[int] = None, max_length: int = 500,
                    min_length: int = 300, SOS_token: Optional[int] = None, EOS_token: Optional[int] = None,
                    PAD_token: Optional[int] = None, beam_size: Optional[int] = None, pen_alpha: Optional[float] = None,
                    return_best_beam: bool = True, kv_cache: bool = True):
        """Returns ``(final_tokens, final_scores, best_tokens)`` like the reference: lists of 1-D token tensors / 0-D score
        tensors, and the best beam as a Python list (SOS included).  ``kv_cache`` (not in the reference): decode step-wise
        through ``model.decoder_session`` (only the new position of every beam is computed; same log-probs to fp32
        rounding) when the model offers one for this geometry; False re-runs ``_decoder_`` on the whole prefix per step."""
        V = self._arg(vocab_size, "dec_voc_size")
        sos, eos = self._arg(SOS_token, "trg_sos_idx"), self._arg(EOS_token, "trg_eos_idx")
        self._arg(PAD_token, "trg_pad_idx")  # accepted and unused, as in the reference
        beam = self._arg(beam_size, "beam_size")
        alpha = self._arg(pen_alpha, "pen_alpha")
        dev = encoder_output.device
        count = beam
        tokens = torch.tensor([[sos]], dtype=torch.long, device=dev)  # [live beams, s]
        scores = torch.zeros(1, dtype=torch.float32, device=dev)
        final_tokens: List[Tensor] = []
        final_scores: List[Tensor] = []
        if kv_cache and return_best_beam and encoder_output.is_cuda and encoder_output.size(0) == 1:
            # no beam can finish inside max_length (the reference's defaults): the lockstep search with one member, whose
            # per-step bookkeeping is one launch
            one = self.beam_search_exits(model, [encoder_output], [layer_n], vocab_size=V, max_length=max_length, min_length=min_length,
                                         SOS_token=sos, EOS_token=eos, PAD_token=self._arg(PAD_token, "trg_pad_idx"), beam_size=beam,
                                         pen_alpha=alpha)
            if one is not None:
                return one[0]
        session = None
        if kv_cache and max_length >= 1 and hasattr(model, "decoder_session") and encoder_output.size(0) == 1:
            session = model.decoder_session(encoder_output, layer_n, max_length)
            if session is not None and beam > session.max_beams:
                session = None
        parent: Optional[Tensor] = None
        i = -1
        for i in range(max_length):
            if session is not None:
                logp = session.step(tokens[:, -1], parent)
            else:
                enc = encoder_output if i == 0 else encoder_output.expand(tokens.size(0), *encoder_output.shape[1:])
                logp = model._decoder_(tokens, enc, layer_n)[:, -1]
            logp = logp / sequence_length_penalty(i + 1, alpha)
            cand, idx = torch.topk((scores.unsqueeze(1) + logp).reshape(-1), count)
            beam_idx = torch.div(idx, V, rounding_mode="floor")
            tok_idx = torch.remainder(idx, V)
            grown = torch.cat([tokens[beam_idx], tok_idx.unsqueeze(1)], dim=1)
            parent = beam_idx
            if i > min_length:  # never while max_length <= min_length (the reference's defaults): no host sync per step then
                done = tok_idx == eos
                if bool(done.any()):
                    for j in torch.nonzero(done).flatten().tolist():
                        final_tokens.append(grown[j])
                        final_scores.append(cand[j])
                        count -= 1
                    grown, cand, parent = grown[~done], cand[~done], beam_idx[~done]
            scores = cand
            if len(final_scores) == beam:
                break
            tokens = grown
        if i == max_length - 1:  # ran out of steps: every live beam is final
            for t, s in zip(tokens, scores):
                final_tokens.append(t)
                final_scores.append(s)
            assert len(final_tokens) == beam and len(final_scores) == beam, \
                "Final_tokens and final_scores lists do not match beam_size size!"
        best = None
        if return_best_beam:
            best = final_tokens[int(torch.stack(final_scores).argmax())].tolist()
        return final_tokens, final_scores, best

    @torch.no_grad()
    def decode_all_exits(self, model, spec: Tensor, valid_len: Tensor, max_length: Optional[int] = None, beam_size: int = 10,
                         **kw) -> List[List[int]]:
        """What inference.py:31-51 does for ONE utterance: the best beam of every exit.  ``spec`` [n_mels, T],
        ``valid_len`` 0-D / [1].  The encoder runs once (taps of all exits)."""
        T = spec.size(1)
        if max_length is None:  # inference.py:31-39 (p = 30, m = 5 / 200)
            max_length = int(30 - T * 5 / 200) if T < 200 else int(T / 12)
        taps = model._run_encoder(spec.unsqueeze(0), valid_len.reshape(1), want_out=False, want_taps=True,
                                  n_groups=model._cfg.n_exits)[1]
        exits = list(range(1, model._cfg.n_exits + 1))
        if kw.get("kv_cache", True):
            together = self.beam_search_exits(model, [taps[n - 1] for n in exits], exits, max_length=max_length, beam_size=beam_size,
                                              **{k: v for k, v in kw.items() if k != "kv_cache"})
            if together is not None:
                return [best for _, _, best in together]
        return [self.beam_search(model, taps[n - 1], n, max_length=max_length, beam_size=beam_size, **kw)[2] for n in exits]

    @torch.no_grad()
    def beam_search_exits(self, model, encoder_outputs: Sequence[Tensor], layer_ns: Sequence[int], vocab_size: Optional[int] = None,
                          max_length: int = 500, min_length: int = 300, SOS_token: Optional[int] = None, EOS_token: Optional[int] = None,
                          PAD_token: Optional[int] = None, beam_size: Optional[int] = None, pen_alpha: Optional[float] = None):
        """``beam_search`` for several exits of one utterance in lockstep: the searches are independent, so every decoder
        launch and every bookkeeping op covers all of them (``model.decoder_session_group``).  Returns the list of
        ``(final_tokens, final_scores, best_tokens)`` per exit, the same values as ``beam_search`` exit by exit -- or None
        when the lockstep does not apply: no session group for this model / geometry, or EOS could finalise beams
        (``max_length - 1 > min_length``), which would let the exits' beam counts diverge."""
        V = self._arg(vocab_size, "dec_voc_size")
        sos = self._arg(SOS_token, "trg_sos_idx")
        self._arg(EOS_token, "trg_eos_idx"), self._arg(PAD_token, "trg_pad_idx")
        beam = self._arg(beam_size, "beam_size")
        alpha = self._arg(pen_alpha, "pen_alpha")
        if max_length < 1 or max_length - 1 > min_length or not hasattr(model, "decoder_session_group"):
            return None
        if any(e.size(0) != 1 for e in encoder_outputs):
            return None
        group = model.decoder_session_group(encoder_outputs, layer_ns, max_length)
        if group is None or beam > group.max_beams:
            return None
        n, dev = len(layer_ns), encoder_outputs[0].device
        scores = torch.zeros((n, 1), dtype=torch.float32, device=dev)
        parent: Optional[Tensor] = None
        if dev.type == "cuda":  # top-k, parent / token split and the token gather of a step in one launch (eec_beam_select)
            rows = max(beam, 1)
            bufs = [torch.zeros((n, rows, max_length + 1), dtype=torch.long, device=dev) for _ in range(2)]
            bufs[0][:, 0, 0] = sos
            last = bufs[0][:, :1, 0].contiguous()
            for i in range(max_length):
                logp = group.step(last, parent)
                scores, parent, last = beam_select(logp, scores, sequence_length_penalty(i + 1, alpha), beam, bufs[i & 1], bufs[(i + 1) & 1], i + 1)
            tokens = bufs[max_length & 1][:, :beam]
        else:
            tokens = torch.full((n, 1, 1), sos, dtype=torch.long, device=dev)  # [exits, live beams, s]
            for i in range(max_length):
                logp = group.step(tokens[:, :, -1], parent) / sequence_length_penalty(i + 1, alpha)
                scores, idx = torch.topk((scores.unsqueeze(2) + logp).reshape(n, -1), beam, dim=1)
                parent = torch.div(idx, V, rounding_mode="floor")
                tok_idx = torch.remainder(idx, V)
                tokens = torch.cat([torch.gather(tokens, 1, parent.unsqueeze(2).expand(-1, -1, tokens.size(2))), tok_idx.unsqueeze(2)], dim=2)
        best = scores.argmax(dim=1).tolist()
        tokens_h = tokens.cpu()
        return [(list(tokens[e]), list(scores[e]), tokens_h[e, best[e]].tolist()) for e in range(n)]
