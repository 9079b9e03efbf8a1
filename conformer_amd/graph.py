"""hipGraph capture of the encoder forward (static shapes): the ~260 kernel launches of one Encoder.forward are recorded
once and replayed with a single host call, so small configurations stop being launch-bound and large ones lose the
host-side gaps.  Uses torch.cuda.CUDAGraph (= hipGraph on ROCm): every kernel of libconformer_hip.so is enqueued on
torch's current stream and never allocates or synchronises, which is exactly what stream capture requires; outputs
and intermediates live in the graph's private memory pool.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def _weights_fingerprint(module: torch.nn.Module, tensors=None):
    from conformer_amd.model.utils._guard import _EPOCH, _ver
    if tensors is None:
        tensors = list(module.parameters()) + list(module.buffers())
    return (_EPOCH[0],) + tuple((t.data_ptr(), _ver(t)) for t in tensors)


class GraphedEncoder:
    """Wraps an eval-mode Encoder for fixed (B, n_mel, T) inputs.  `lengths` is part of the static input (its VALUES may
    change between replays: the attention kernel reads them from device memory).

    Precondition: FROZEN WEIGHTS.  The captured graph holds the addresses of the parameters AND of the derived weight
    copies built during warm-up (fused QKV matrix, packed conv / linear weights, projected position tables, 16-bit / split
    planes).  Every call compares the (address, in-place version) of each parameter and buffer with the capture-time
    record; after an optimizer step, `load_state_dict` or any in-place update the graph is captured again (the old one
    is dropped first) instead of replaying stale packs.  Writes that bypass the version counter (`param.data.copy_`)
    need `conformer_amd.model.utils._guard.invalidate_weight_caches()`, which this check also sees.
    check_weights=False (frozen-weight serving) skips that per-call walk over the ~470 tensors of a Conformer-L (the list itself is
    cached at capture time either way); a re-capture replaces `static_y`: outputs handed out earlier are invalid after it
    (`captures` counts them)."""

    def __init__(self, encoder: torch.nn.Module, example_x: torch.Tensor, example_lengths: Optional[torch.Tensor],
                 warmup: int = 2, autocast_dtype: Optional[torch.dtype] = None, check_weights: bool = True) -> None:
        if encoder.training:
            raise ValueError("GraphedEncoder captures the inference path: call encoder.eval() first")
        self.encoder = encoder
        self.static_x = example_x.clone()
        self.static_len = None if example_lengths is None else example_lengths.clone()
        self.warmup = warmup
        self.autocast_dtype = autocast_dtype             # bfloat16 / float16: capture the 16-bit matrix-pipe path
        self.captures = 0
        self.check_weights = bool(check_weights)
        self._capture()

    def _capture(self) -> None:
        encoder, warmup = self.encoder, self.warmup
        self.graph = None                                # free the previous graph's pool before building the new one
        self.static_y = self.static_out_len = None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        amp = lambda: torch.autocast("cuda", dtype=self.autocast_dtype or torch.bfloat16, enabled=self.autocast_dtype is not None)
        with torch.cuda.stream(side), torch.no_grad(), amp():
            for _ in range(warmup):                      # builds every cached pack / table / 16-bit weight copy outside the capture
                encoder(self.static_x, self.static_len)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph), amp():
            self.static_y, self.static_out_len = encoder(self.static_x, self.static_len)
        self._tensors = list(encoder.parameters()) + list(encoder.buffers())     # (walked once per capture, not per replay)
        self._fingerprint = _weights_fingerprint(encoder, self._tensors)
        self.captures += 1

    def __call__(self, x: torch.Tensor, lengths: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        if x.shape != self.static_x.shape:
            raise ValueError(f"graph captured for input {tuple(self.static_x.shape)}, got {tuple(x.shape)}")
        if self.encoder.training:
            raise ValueError("GraphedEncoder replays the inference path: the wrapped encoder was switched to .train()")
        if self.check_weights and _weights_fingerprint(self.encoder, self._tensors) != self._fingerprint:
            self._capture()                              # weights changed since the capture: never replay stale packs
        if x.data_ptr() != self.static_x.data_ptr():
            self.static_x.copy_(x)
        if self.static_len is not None and lengths is not None and lengths.data_ptr() != self.static_len.data_ptr():
            self.static_len.copy_(lengths)
        self.graph.replay()
        return self.static_y, self.static_out_len
