"""Surface of the reference's evaluation.py: ConformerCriterion (CTC loss, evaluation.py:8-16) on the gfx950 lattice
kernels, ConformerMetric (WER / CER, evaluation.py:18-27) as a plain host-side edit distance (the reference delegates to
torchmetrics' WordErrorRate / CharErrorRate: total edit distance / total reference length over the batch)."""
from __future__ import annotations

from typing import List, Sequence, Union

import torch

from . import autograd as ag


class ConformerCriterion:
    def __init__(self, blank_id: int = 0) -> None:
        self.blank_id = int(blank_id)

    def ctc_loss(self, outputs: torch.Tensor, targets: torch.Tensor, input_lengths: torch.Tensor,
                 target_lengths: torch.Tensor) -> torch.Tensor:
        """outputs: logits (B,T',V) as Conformer.forward returns them (any float dtype; computed in fp32 as
        evaluation.py:13); targets (B,Lmax) padded or 1-D concatenated (float targets are accepted as the reference
        passes them, evaluation.py:14); lengths (B,).  Mean over utterances of nll / target_length, infinite terms zeroed."""
        if not outputs.is_cuda:
            raise RuntimeError("ConformerCriterion.ctc_loss: logits must be on the HIP device (no CPU fallback)")
        return ag.CtcLossFn.apply(outputs.float(), targets.long(), input_lengths, target_lengths, self.blank_id)


def _edit_distance(ref: Sequence, hyp: Sequence) -> int:
    prev = list(range(len(hyp) + 1))
    for i, r in enumerate(ref, 1):
        cur = [i] + [0] * len(hyp)
        for j, h in enumerate(hyp, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (r != h))
        prev = cur
    return prev[-1]


def _error_rate(prediction: Union[str, List[str]], target: Union[str, List[str]], split) -> torch.Tensor:
    preds = [prediction] if isinstance(prediction, str) else list(prediction)
    refs = [target] if isinstance(target, str) else list(target)
    errors = total = 0
    for p, r in zip(preds, refs):
        rt = split(r)
        errors += _edit_distance(rt, split(p))
        total += len(rt)
    return torch.tensor(errors / total if total else float("nan"))


class ConformerMetric:
    def wer_score(self, prediction: Union[str, List[str]], target: Union[str, List[str]]) -> torch.Tensor:
        return _error_rate(prediction, target, str.split)

    def cer_score(self, prediction: Union[str, List[str]], target: Union[str, List[str]]) -> torch.Tensor:
        return _error_rate(prediction, target, list)
