"""Chunk-by-chunk (streaming) evaluation of the Encoder with cached K/V and depthwise-convolution state -- BASELINE cfg-5
("T=20000 in 640-frame chunks with cached K/V + depthwise state").

The reference has no streaming code at all, so there is nothing to be identical to chunk by chunk; the semantics chosen are
the ones its own operators already have at the END of an utterance, applied at the end of every chunk (the prefix rule):

    an encoder frame that belongs to chunk c is computed, in every layer, from the frames of chunks <= c only --
    self-attention sees the keys received so far (the same key limit `lengths` gives Encoder.forward), the depthwise
    convolution treats the not-yet-received frames as the zero padding of convolution.py:14.

Consequences that the tests pin: one chunk holding the whole utterance IS Encoder.forward; the frames of the first chunk
equal Encoder.forward of that prefix alone; any chunking equals the masked whole-sequence restatement
`oracle.encoder_forward_chunked` (float64).  The conv-subsampling stem, the input Linear, LayerNorm and the feed-forward
modules are local in time, so they are exact under chunking; relative positions need no absolute frame index.

State per layer: the fused Q|K|V projections of every frame so far (B, T'max, 3d) -- new rows are appended in place and
`cfm_relpos_attention_rows_f32` computes the new query rows only, against the whole cache -- and the last (K-1)/2 GLU
outputs feeding the depthwise convolution.  Per utterance: the un-consumed tail (3..6 frames) of the mel stream and the
positional table projected ONCE for T'max.  fp32 inference only.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch

from . import ops
from .model.modules.encoder import Encoder


class StreamingEncoder:
    def __init__(self, encoder: Encoder, batch: int, max_mel_frames: int) -> None:
        if encoder.training:
            raise RuntimeError("StreamingEncoder: put the encoder in eval() mode (running BatchNorm statistics, no dropout)")
        p = next(encoder.parameters())
        if not p.is_cuda or p.dtype != torch.float32:
            raise RuntimeError("StreamingEncoder: the encoder must live on the HIP device in fp32 (no CPU fallback)")
        self.enc = encoder
        self.B = int(batch)
        self.d = encoder.linear.out_features
        self.t_max = ((int(max_mel_frames) - 1) // 2 - 1) // 2
        if self.t_max < 1:
            raise ValueError("max_mel_frames must give at least one encoder frame (>= 7)")
        dev = p.device
        layers = list(encoder.layers)
        self.half = [(l.conv.deepwise_conv.kernel_size[0] - 1) // 2 for l in layers]
        with torch.no_grad():
            self.table = encoder.rel_pe.table(self.t_max)
            self.pos_all = encoder._projected_positions(self.table)       # (2T'max-1, L*d): every layer's pos_proj, once
        self.qkv = [torch.zeros(self.B, self.t_max, 3 * self.d, device=dev) for _ in layers]
        self.ctx = torch.zeros(self.B, self.t_max, self.d, device=dev)     # scratch shared by the layers
        self.conv_state = [torch.zeros(self.B, h, self.d, device=dev) for h in self.half]
        self.lengths = torch.zeros(self.B, dtype=torch.int64, device=dev)
        self.mel_tail: Optional[torch.Tensor] = None
        self.frames = 0                                                    # encoder frames produced so far

    def reset(self) -> None:
        for t in self.conv_state:
            t.zero_()
        self.mel_tail = None
        self.frames = 0

    @torch.no_grad()
    def step(self, mel_chunk: torch.Tensor) -> torch.Tensor:
        """mel_chunk (B, n_mel, Tc): the next Tc log-mel frames of every utterance (any Tc >= 1).  Returns the encoder
        frames that became computable, (B, k, d) with k = ((buffered-1)//2-1)//2 >= 0."""
        if torch.is_autocast_enabled("cuda"):
            raise RuntimeError("StreamingEncoder runs in fp32 only (the incremental attention kernel has no 16-bit form)")
        enc, d = self.enc, self.d
        x = mel_chunk if self.mel_tail is None else torch.cat([self.mel_tail, mel_chunk], dim=2)
        k = ((x.shape[2] - 1) // 2 - 1) // 2
        if k <= 0:
            self.mel_tail = x
            return x.new_empty(self.B, 0, d)
        n0 = self.frames
        if n0 + k > self.t_max:
            raise RuntimeError(f"StreamingEncoder: stream longer than max_mel_frames (T'max = {self.t_max})")
        # encoder frame t covers mel frames 4t .. 4t+6: the buffer starts at mel frame 4*n0, keep what frame n0+k needs
        self.mel_tail = x[:, :, 4 * k:].contiguous()
        h = enc.downsampling_conv.channel_last(x.contiguous())             # (B, k, F'*C): the stem is local in time
        wlp = enc._packs.get("wlp", (enc.linear.weight,), lambda: ops.pack_linear_weight(enc.linear.weight, d, enc.n_freq_out))
        h = ops.linear(h, wlp, enc.linear.bias)
        self.lengths.fill_(n0 + k)
        for i, blk in enumerate(enc.layers):
            h = self._block(i, blk, h, n0, k)
        self.frames = n0 + k
        return h

    def _block(self, i: int, blk, x: torch.Tensor, n0: int, k: int) -> torch.Tensor:
        d = self.d
        y = blk.ffn_1.fused(x, residual=x, alpha=0.5)
        # ---- self-attention of the new rows against the whole cache
        att, a = blk.attention, blk.attention.attention
        xn = ops.layernorm(y, att.layer_norm.weight, att.layer_norm.bias, att.layer_norm.eps)
        w, b = a._qkv_params()
        self.qkv[i][:, n0:n0 + k].copy_(ops.linear(xn, w, b))
        ops.relpos_attention_rows(self.qkv[i], self.pos_all[:, i * d:(i + 1) * d], a.content_bias, a.position_bias,
                                  self.lengths, a.n_heads, n0, k, self.ctx, keys_hint=n0 + k)
        y = ops.linear_residual(self.ctx[:, n0:n0 + k].contiguous(), a.out_proj.weight, a.out_proj.bias, y, 1.0)
        # ---- convolution module: the depthwise window reaches (K-1)/2 frames back into the cached GLU outputs
        cv, half = blk.conv, self.half[i]
        bn = cv.batch_norm
        hn = ops.layernorm(y, cv.layer_norm.weight, cv.layer_norm.bias, cv.layer_norm.eps)
        g = ops.linear_glu(hn, cv.pointwise_conv_1.weight, cv.pointwise_conv_1.bias)
        buf = torch.cat([self.conv_state[i], g], dim=1)                    # (B, half + k, C)
        s = ops.dwconv_bn_swish(buf, cv.deepwise_conv.weight, cv.deepwise_conv.bias, bn.weight, bn.bias, bn.running_mean,
                                bn.running_var, bn.eps)
        self.conv_state[i] = buf[:, buf.shape[1] - half:].contiguous()
        y = ops.linear_residual(s[:, half:].contiguous(), cv.pointwise_conv_2.weight, cv.pointwise_conv_2.bias, y, 1.0)
        y = blk.ffn_2.fused(y, residual=y, alpha=0.5)
        return ops.layernorm(y, blk.layer_norm.weight, blk.layer_norm.bias, blk.layer_norm.eps)

    def run(self, mel: torch.Tensor, chunk_frames: int = 640) -> torch.Tensor:
        """Feeds mel (B, n_mel, T) in chunks of `chunk_frames` and returns the concatenated (B, T', d) output."""
        outs: List[torch.Tensor] = [self.step(mel[:, :, t:t + chunk_frames]) for t in range(0, mel.shape[2], chunk_frames)]
        return torch.cat(outs, dim=1)


def chunk_ends(total_mel_frames: int, chunk_frames: Iterable[int]) -> List[int]:
    """Encoder-frame boundaries produced by feeding chunks of the given sizes (for comparing with the masked restatement)."""
    ends, got = [], 0
    for c in chunk_frames:
        got += c
        n = ((got - 1) // 2 - 1) // 2
        if n > 0 and (not ends or n > ends[-1]):
            ends.append(n)
    return ends
