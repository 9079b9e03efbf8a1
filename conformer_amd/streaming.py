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
    """graphs=True: every distinct chunk step -- keyed on (frames so far, chunk length, buffered tail) -- is captured ONCE as a
    hipGraph and replayed from then on (the next utterance batch of a streaming service walks through the same keys): one host
    call per chunk instead of ~420 launches, whose issue cost (not their device time) was what a 7-11 ms chunk step consisted
    of (profiles/r02_streaming_bench.json vs r03).  All graphs share one memory pool; the state (K/V caches, depthwise state,
    mel tail) lives in fixed buffers outside it.  The returned frames are a copy (the graph's output buffer is reused)."""

    def __init__(self, encoder: Encoder, batch: int, max_mel_frames: int, graphs: bool = False) -> None:
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
        self.qkv = [torch.zeros(self.B, self.t_max, 3 * self.d, device=dev, dtype=torch.float32) for _ in layers]
        self.ctx = torch.zeros(self.B, self.t_max, self.d, device=dev, dtype=torch.float32)     # scratch shared by the layers
        self.conv_state = [torch.zeros(self.B, h, self.d, device=dev, dtype=torch.float32) for h in self.half]
        self.lengths = torch.zeros(self.B, dtype=torch.int64, device=dev)
        self.mel_tail_buf: Optional[torch.Tensor] = None                   # (B, n_mel, 6): the un-consumed 0..6 mel frames
        self.tail_len = 0
        self.frames = 0                                                    # encoder frames produced so far
        self.use_graphs = bool(graphs)
        self._graphs = {}                                                  # key -> (graph, static input, static output, k, new tail)
        self._pool = None

    def reset(self) -> None:
        for t in self.conv_state:
            t.zero_()
        self.tail_len = 0
        self.frames = 0

    @property
    def mel_tail(self) -> Optional[torch.Tensor]:
        return None if self.tail_len == 0 or self.mel_tail_buf is None else self.mel_tail_buf[:, :, :self.tail_len]

    @torch.no_grad()
    def step(self, mel_chunk: torch.Tensor) -> torch.Tensor:
        """mel_chunk (B, n_mel, Tc): the next Tc log-mel frames of every utterance (any Tc >= 1).  Returns the encoder
        frames that became computable, (B, k, d) with k = ((buffered-1)//2-1)//2 >= 0."""
        x_in = ops._req(mel_chunk, "mel_chunk")
        if self.mel_tail_buf is None:
            self.mel_tail_buf = torch.zeros(self.B, x_in.shape[1], 8, device=x_in.device, dtype=torch.float32)
        n0, tail = self.frames, self.tail_len
        total = tail + x_in.shape[2]
        k = max(0, ((total - 1) // 2 - 1) // 2)
        if k <= 0:
            # not enough frames for one encoder frame yet: just buffer (<= 6 frames; tiny copy, never graphed)
            self.mel_tail_buf[:, :, tail:total].copy_(x_in)
            self.tail_len = total
            return x_in.new_empty(self.B, 0, self.d)
        if n0 + k > self.t_max:
            raise RuntimeError(f"StreamingEncoder: stream longer than max_mel_frames (T'max = {self.t_max})")
        if not self.use_graphs:
            out = self._step_core(x_in, n0, tail, k)
        else:
            key = (n0, x_in.shape[2], tail)
            ent = self._graphs.get(key)
            if ent is None:
                if self._pool is None:
                    self._pool = torch.cuda.graph_pool_handle()
                    self._step_warm(x_in, n0, tail, k)                    # derived weights / tables are built outside any capture
                static_in = x_in.clone()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self._pool):
                    static_out = self._step_core(static_in, n0, tail, k)
                ent = self._graphs[key] = (g, static_in, static_out)
            else:
                ent[1].copy_(x_in)
            ent[0].replay()
            out = ent[2].clone()
        self.tail_len = total - 4 * k
        self.frames = n0 + k
        return out

    def _step_warm(self, x_in: torch.Tensor, n0: int, tail: int, k: int) -> None:
        """One eager step whose effects on the state are undone (it overwrites cache rows the real step rewrites; the mel tail and
        the depthwise state are restored)."""
        keep_tail = self.mel_tail_buf.clone()
        keep_conv = [t.clone() for t in self.conv_state]
        self._step_core(x_in, n0, tail, k)
        self.mel_tail_buf.copy_(keep_tail)
        for t, kt in zip(self.conv_state, keep_conv):
            t.copy_(kt)
        torch.cuda.synchronize()

    def _step_core(self, x_in: torch.Tensor, n0: int, tail: int, k: int) -> torch.Tensor:
        """The device work of one chunk: capture-safe (fixed state buffers, no host synchronisation, no data-dependent shapes)."""
        enc, d = self.enc, self.d
        x = x_in if tail == 0 else torch.cat([self.mel_tail_buf[:, :, :tail], x_in], dim=2)
        # encoder frame t covers mel frames 4t .. 4t+6: the buffer starts at mel frame 4*n0, keep what frame n0+k needs
        rest = x.shape[2] - 4 * k
        self.mel_tail_buf[:, :, :rest].copy_(x[:, :, 4 * k:])
        h = enc.downsampling_conv.channel_last(x.contiguous())             # (B, k, F'*C): the stem is local in time
        wlp = enc._packs.get("wlp", (enc.linear.weight,), lambda: ops.pack_linear_weight(enc.linear.weight, d, enc.n_freq_out))
        h = ops.linear(h, wlp, enc.linear.bias)
        self.lengths.fill_(n0 + k)
        st = None
        for i, blk in enumerate(enc.layers):
            h, st = self._block(i, blk, h, n0, k, st, want_stats=i + 1 < len(enc.layers))
        return h

    def _block(self, i: int, blk, x: torch.Tensor, n0: int, k: int, x_stats=None, want_stats: bool = False):
        """One Conformer block on the chunk's rows.  Returns (y, LayerNorm statistics of y's rows or None).  On the folded-LayerNorm
        path (ops.ln_fold_ok) the residual GEMMs emit the statistics the next sub-layer's LayerNorm needs, as in
        ConformerBlock.fused_chain; a sub-layer whose producer took the split-K form instead (FFN out at chunk sizes) runs its
        LayerNorm kernel."""
        d = self.d
        fold = blk._ln_fold(x)
        y, st = blk.ffn_1.fused(x, residual=x, alpha=0.5, stats=x_stats, emit_stats=True) if fold else \
            (blk.ffn_1.fused(x, residual=x, alpha=0.5), None)
        # ---- self-attention of the new rows against the whole cache
        att, a = blk.attention, blk.attention.attention
        w, b = a._qkv_params()
        if st is not None:
            ln = att.layer_norm
            wf, bf, cs = a._packs.get("qkv_ln_fold", (w, b, ln.weight, ln.bias), lambda: ops.fold_layernorm(w, b, ln.weight, ln.bias))
            qkv_new = ops.linear_lnfold(y, st, wf, bf, cs, ln.eps)
        else:
            qkv_new = ops.linear(ops.layernorm(y, att.layer_norm.weight, att.layer_norm.bias, att.layer_norm.eps), w, b)
        self.qkv[i][:, n0:n0 + k].copy_(qkv_new)
        ops.relpos_attention_rows(self.qkv[i], self.pos_all[:, i * d:(i + 1) * d], a.content_bias, a.position_bias,
                                  self.lengths, a.n_heads, n0, k, self.ctx, keys_hint=n0 + k)
        rows = self.ctx[:, n0:n0 + k].contiguous()
        y, st = ops.linear_residual(rows, a.out_proj.weight, a.out_proj.bias, y, 1.0, emit_stats=True) if fold else \
            (ops.linear_residual(rows, a.out_proj.weight, a.out_proj.bias, y, 1.0), None)
        # ---- convolution module: the depthwise window reaches (K-1)/2 frames back into the cached GLU outputs
        cv, half = blk.conv, self.half[i]
        bn = cv.batch_norm
        if st is not None:
            ln, pw1 = cv.layer_norm, cv.pointwise_conv_1
            wf, bf, cs = cv._packs.get("ln_fold", (pw1.weight, pw1.bias, ln.weight, ln.bias),
                                       lambda: ops.fold_layernorm(pw1.weight, pw1.bias, ln.weight, ln.bias))
            g = ops.linear_lnfold(y, st, wf, bf, cs, ln.eps, glu=True)
        else:
            hn = ops.layernorm(y, cv.layer_norm.weight, cv.layer_norm.bias, cv.layer_norm.eps)
            g = ops.linear_glu(hn, cv.pointwise_conv_1.weight, cv.pointwise_conv_1.bias)
        buf = torch.cat([self.conv_state[i], g], dim=1)                    # (B, half + k, C)
        s = ops.dwconv_bn_swish(buf, cv.deepwise_conv.weight, cv.deepwise_conv.bias, bn.weight, bn.bias, bn.running_mean,
                                bn.running_var, bn.eps)
        self.conv_state[i].copy_(buf[:, buf.shape[1] - half:])            # fixed buffer (the graph replays write the same address)
        rows = s[:, half:].contiguous()
        y, st = ops.linear_residual(rows, cv.pointwise_conv_2.weight, cv.pointwise_conv_2.bias, y, 1.0, emit_stats=True) if fold else \
            (ops.linear_residual(rows, cv.pointwise_conv_2.weight, cv.pointwise_conv_2.bias, y, 1.0), None)
        y = blk.ffn_2.fused(y, residual=y, alpha=0.5, stats=st)
        ln = blk.layer_norm
        if fold and want_stats:
            return ops.layernorm(y, ln.weight, ln.bias, ln.eps, emit_stats=True)
        return ops.layernorm(y, ln.weight, ln.bias, ln.eps), None

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
