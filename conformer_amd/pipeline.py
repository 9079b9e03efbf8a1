"""Front-end pipeline overlap (reference: ConformerCollate.__call__, dataset.py:88-108, which runs
pad -> H2D -> log-mel serially in the main thread before every step).

`FrontendPipeline` prepares batch n+1 on a side HIP stream -- pinned host staging of the zero-padded waveforms, async
H2D copy, log-mel (+ SpecAugment) kernels, the length sort of dataset.py:97 -- while the caller's stream runs the model on
batch n.  The hand-over is an event wait on the consumer's stream; nothing synchronises the host."""
from __future__ import annotations

from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import torch

from .frontend import ConformerAudioFrontend, ConformerAugment


class FrontendPipeline:
    """Iterate `(mels (B,n_mels,T), lengths (B,) int64, order (B,) int64)`: utterances sorted by length, descending
    (`order` = indices into the source batch, for permuting targets as dataset.py:99-101 does).

    source: iterable of batches, each a sequence of 1-D CPU float waveforms.
    """

    def __init__(self, source: Iterable[Sequence[torch.Tensor]], frontend: ConformerAudioFrontend,
                 augment: Optional[ConformerAugment] = None, depth: int = 2) -> None:
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.source, self.frontend, self.augment, self.depth = source, frontend, augment, depth
        self.stream = torch.cuda.Stream(device=frontend.device)
        self.timeline: Optional[List[Tuple[torch.cuda.Event, torch.cuda.Event]]] = None   # set to [] to record (start, done) events per batch

    def _stage(self, audios: Sequence[torch.Tensor]):
        fe = self.frontend
        lengths = [int(a.numel()) for a in audios]
        host = torch.zeros(len(audios), max(lengths), dtype=torch.float32).pin_memory()
        for i, a in enumerate(audios):
            host[i, :lengths[i]] = a
        host_len = torch.tensor(lengths, dtype=torch.int64).pin_memory()
        with torch.cuda.stream(self.stream):
            if self.timeline is not None:
                ev0 = torch.cuda.Event(enable_timing=True)
                ev0.record(self.stream)
            wave = host.to(fe.device, non_blocking=True)
            n = host_len.to(fe.device, non_blocking=True)
            mels = fe.mel_spectrogram(wave)
            if self.augment is not None:
                mels = self.augment(mels)
            frames = n // fe.hop_length + 1                                  # processor.py:392
            frames, order = torch.sort(frames, descending=True)              # dataset.py:97
            mels = mels.index_select(0, order)
            ev = torch.cuda.Event(enable_timing=self.timeline is not None)
            ev.record(self.stream)
            if self.timeline is not None:
                self.timeline.append((ev0, ev))
        return mels, frames, order, ev, (host, host_len)                     # host buffers stay alive until consumed

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        queue: List[tuple] = []
        it = iter(self.source)
        done = False
        while True:
            while not done and len(queue) < self.depth:
                try:
                    queue.append(self._stage(next(it)))
                except StopIteration:
                    done = True
            if not queue:
                return
            mels, frames, order, ev, _keep = queue.pop(0)
            cur = torch.cuda.current_stream(self.frontend.device)
            cur.wait_event(ev)
            for t in (mels, frames, order):
                t.record_stream(cur)                                         # allocated on the side stream, used on `cur`
            yield mels, frames, order
