"""Greedy CTC decoding on the device (reference: ConformerProcessor.greedy_decode /
batch_greedy_decode, processing/processor.py:301-334): one kernel does the per-frame argmax ("CTC alignment indices")
and the pad/unk filtering + repeat collapse, replacing the reference's per-frame `.item()` host loop (train.py:61)."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib, ops


def greedy_ctc_decode(logits: torch.Tensor, pad_id: int, unk_id: int, lengths: Optional[torch.Tensor] = None
                      ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """logits (B,T,V) fp32 on the HIP device -> (frame_ids (B,T) int64, tokens (B,T) int64 padded with -1, counts (B)).
    `lengths=None` decodes every frame, as the reference does."""
    x = ops._req(logits, "logits")
    B, T, V = x.shape
    frame_ids = torch.empty(B, T, dtype=torch.int64, device=x.device)
    tokens = torch.empty(B, T, dtype=torch.int64, device=x.device)
    counts = torch.empty(B, dtype=torch.int64, device=x.device)
    if lengths is not None:
        lengths = ops._req(lengths, "lengths", torch.int64)
    st = _lib.load().cfm_greedy_ctc_decode_f32(x.data_ptr(), ops._p(lengths), frame_ids.data_ptr(), tokens.data_ptr(),
                                               counts.data_ptr(), B, T, V, int(pad_id), int(unk_id), ops._stream())
    _lib.check(st, "cfm_greedy_ctc_decode_f32")
    return frame_ids, tokens, counts


def tokens_to_text(tokens: torch.Tensor, counts: torch.Tensor, vocab: Sequence[str], delim_token: str = "|") -> List[str]:
    """Host-side join of the decoded ids (processor.py:319): ''.join(vocab[id]).replace(delim, ' ')."""
    tk, ct = tokens.cpu().tolist(), counts.cpu().tolist()
    return ["".join(vocab[i] for i in row[:n]).replace(delim_token, " ") for row, n in zip(tk, ct)]
