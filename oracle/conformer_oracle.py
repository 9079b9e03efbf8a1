"""CPU oracle for the Conformer encoder hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional restatement (plain PyTorch CPU ops, any float
dtype) of the arithmetic the reference performs on the hot path.  It is NOT part of the
product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the reported CPU baseline.  The product
path (``conformer_amd``) never routes through it and fails loudly without its HIP library.

Parity status: PINNED.  ``tests/golden/*.npz`` were produced by importing the reference's
own ``model`` package in the build container (``tests/golden/make_golden.py``); every
function here is checked against those vectors in ``tests/test_oracle_golden.py``.
The log-mel / SpecAugment front end lives in torchaudio (absent here and in the
reference tree) and is therefore "parity unpinned" -- see DESIGN.md.

Each function cites the reference lines it follows (paths relative to /root/reference).
Parameters are passed as a flat ``dict[str, Tensor]`` using the reference's state_dict
keys, addressed with a key prefix.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

LN_EPS = 1e-5   # nn.LayerNorm default, model/utils/ffn.py:8
BN_EPS = 1e-5   # nn.BatchNorm1d default, model/utils/convolution.py:16
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------- primitives
def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = LN_EPS) -> torch.Tensor:
    """nn.LayerNorm over the last dim (ffn.py:8,16; attention.py:10,15; convolution.py:12,22; block.py:15,27)."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * w + b


def swish(x: torch.Tensor) -> torch.Tensor:
    """activation.py:7-8."""
    return x * torch.sigmoid(x)


def subsampled_lengths(lengths: torch.Tensor) -> torch.Tensor:
    """convolution.py:55."""
    return ((lengths - 1) // 2 - 1) // 2


def subsampled_frames(t: int) -> int:
    return ((t - 1) // 2 - 1) // 2


# --------------------------------------------------------------------------- FFN module
def ffn_module(x: torch.Tensor, p: Params, pre: str) -> torch.Tensor:
    """FeedForwardModule.forward, ffn.py:15-23 (dropout p=0)."""
    h = layer_norm(x, p[pre + "layer_norm.weight"], p[pre + "layer_norm.bias"])
    h = h @ p[pre + "hidden_linear.weight"].t() + p[pre + "hidden_linear.bias"]
    h = swish(h)
    return h @ p[pre + "out_linear.weight"].t() + p[pre + "out_linear.bias"]


# --------------------------------------------------------------------------- rel-pos MHSA
def relpos_table(t: int, div_term: torch.Tensor) -> torch.Tensor:
    """RelativePositionalEncoding.forward without the batch repeat, position.py:11-27.

    Row j (0 <= j <= 2t-2) encodes relative position r = t-1-j:
    pe[j, 2c] = sin(r*w_c), pe[j, 2c+1] = cos(r*w_c).
    ``div_term`` is the (1, d/2) parameter stored in the state_dict (position.py:9).
    """
    dt = div_term.reshape(-1)
    r = torch.arange(t - 1, -t, -1, dtype=dt.dtype)
    ang = r.abs()[:, None] * dt[None, :]           # the reference forms |r|*w then negates
    sgn = torch.sign(r)[:, None]
    pe = torch.empty(2 * t - 1, 2 * dt.numel(), dtype=dt.dtype)
    pe[:, 0::2] = torch.sin(sgn * ang)
    pe[:, 1::2] = torch.cos(sgn * ang)
    return pe


def relpos_attention_core(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, pp: torch.Tensor,
                          u: torch.Tensor, vb: torch.Tensor, lengths: Optional[torch.Tensor],
                          visible_end: Optional[torch.Tensor] = None) -> torch.Tensor:
    """scaled_dot_product_relative_attention + _relative_shift, attention.py:47-72,94-102.

    q,k,v: (B,T,H,dh) projected; pp: (2T-1,H,dh) projected positions (row j <-> r=T-1-j);
    u,vb: (H,dh) content/position biases.  Returns context (B,T,H*dh), heads h-major.
    The shift is restated as an explicit gather: score_pos[i,k] = (q_i+v).p_{r=i-k}.
    """
    B, T, H, dh = q.shape
    content = torch.einsum("bihc,bkhc->bhik", q + u, k)
    full = torch.einsum("bihc,jhc->bhij", q + vb, pp)                   # (B,H,T,2T-1)
    i = torch.arange(T)[:, None]
    kk = torch.arange(T)[None, :]
    j = (T - 1) - (i - kk)                                                # row of pp for r=i-k
    pos = full.gather(-1, j.expand(B, H, T, T))
    s = (content + pos) / math.sqrt(dh)
    if lengths is not None:
        pad = torch.arange(T)[None, :] >= lengths[:, None]                # True at padded keys
        s = s.masked_fill(pad[:, None, None, :], torch.finfo(s.dtype).min)
    if visible_end is not None:                                           # chunked evaluation: row i sees keys < visible_end[i]
        hidden = torch.arange(T)[None, :] >= visible_end[:, None]         # (T,T)
        s = s.masked_fill(hidden[None, None], torch.finfo(s.dtype).min)
    a = torch.softmax(s, dim=-1)
    ctx = torch.einsum("bhik,bkhc->bihc", a, v)
    return ctx.reshape(B, T, H * dh)


def mhsa_module(x: torch.Tensor, pe: torch.Tensor, lengths: Optional[torch.Tensor], p: Params, pre: str,
                n_heads: int, visible_end: Optional[torch.Tensor] = None) -> torch.Tensor:
    """MultiHeadSelfAttentionModule.forward, attention.py:14-18, 74-92 (dropout p=0).

    ``pe`` is the un-repeated (2T-1, d) table.
    """
    B, T, d = x.shape
    dh = d // n_heads
    a = pre + "attention."
    xn = layer_norm(x, p[pre + "layer_norm.weight"], p[pre + "layer_norm.bias"])
    q = (xn @ p[a + "query_proj.weight"].t() + p[a + "query_proj.bias"]).view(B, T, n_heads, dh)
    k = (xn @ p[a + "key_proj.weight"].t() + p[a + "key_proj.bias"]).view(B, T, n_heads, dh)
    v = (xn @ p[a + "value_proj.weight"].t() + p[a + "value_proj.bias"]).view(B, T, n_heads, dh)
    pp = (pe @ p[a + "pos_proj.weight"].t() + p[a + "pos_proj.bias"]).view(2 * T - 1, n_heads, dh)
    ctx = relpos_attention_core(q, k, v, pp, p[a + "content_bias"], p[a + "position_bias"], lengths, visible_end)
    return ctx @ p[a + "out_proj.weight"].t() + p[a + "out_proj.bias"]


# --------------------------------------------------------------------------- conv module
def conv_module(x: torch.Tensor, p: Params, pre: str, training: bool = False,
                bn_state: Optional[Dict[str, torch.Tensor]] = None,
                visible_end: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ConvolutionModule.forward, convolution.py:21-32 (dropout p=0).

    Channel-last restatement: the pointwise convs are row GEMMs on (B,T,C).
    eval: BatchNorm uses running stats; train: batch mean / biased variance over all
    B*T positions (padded frames included) and, if ``bn_state`` is given,
    the running-stat update (momentum 0.1, unbiased variance) is written into it.
    """
    B, T, C = x.shape
    xn = layer_norm(x, p[pre + "layer_norm.weight"], p[pre + "layer_norm.bias"])
    z = xn @ p[pre + "pointwise_conv_1.weight"][:, :, 0].t() + p[pre + "pointwise_conv_1.bias"]
    g = z[..., :C] * torch.sigmoid(z[..., C:])                            # GLU(dim=1): value first, gate second
    wdw = p[pre + "deepwise_conv.weight"][:, 0, :]                        # (C,K)
    K = wdw.shape[1]
    half = (K - 1) // 2
    gp = F.pad(g, (0, 0, half, half))                                     # zero pad in time, per utterance
    c = p[pre + "deepwise_conv.bias"].expand(B, T, C).clone()
    for jtap in range(K):
        tap = gp[:, jtap:jtap + T, :] * wdw[:, jtap]
        if visible_end is not None:                                       # chunked: frames >= visible_end[i] count as padding
            seen = (torch.arange(T) + jtap - half) < visible_end
            tap = tap * seen[None, :, None].to(tap.dtype)
        c = c + tap
    if training:
        flat = c.reshape(-1, C)
        mean = flat.mean(0)
        var = ((flat - mean) ** 2).mean(0)
        if bn_state is not None:
            n = flat.shape[0]
            bn_state["running_mean"] = (1 - BN_MOMENTUM) * p[pre + "batch_norm.running_mean"] + BN_MOMENTUM * mean
            bn_state["running_var"] = (1 - BN_MOMENTUM) * p[pre + "batch_norm.running_var"] + BN_MOMENTUM * var * n / max(n - 1, 1)
    else:
        mean = p[pre + "batch_norm.running_mean"]
        var = p[pre + "batch_norm.running_var"]
    bn = (c - mean) / torch.sqrt(var + BN_EPS) * p[pre + "batch_norm.weight"] + p[pre + "batch_norm.bias"]
    s = swish(bn)
    return s @ p[pre + "pointwise_conv_2.weight"][:, :, 0].t() + p[pre + "pointwise_conv_2.bias"]


# --------------------------------------------------------------------------- block / stem / encoder
def conformer_block(x: torch.Tensor, pe: torch.Tensor, lengths: Optional[torch.Tensor], p: Params, pre: str,
                    n_heads: int, training: bool = False, visible_end: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ConformerBlock.forward, block.py:17-29."""
    y = 0.5 * ffn_module(x, p, pre + "ffn_1.") + x
    y = mhsa_module(y, pe, lengths, p, pre + "attention.", n_heads, visible_end) + y
    y = conv_module(y, p, pre + "conv.", training, visible_end=visible_end) + y
    y = 0.5 * ffn_module(y, p, pre + "ffn_2.") + y
    return layer_norm(y, p[pre + "layer_norm.weight"], p[pre + "layer_norm.bias"])


def conv_subsampling(x: torch.Tensor, p: Params, pre: str) -> torch.Tensor:
    """ConvolutionSubsampling.forward, convolution.py:42-57.  x: (B, n_mel, T) -> (B, T', C*F'), feature = c*F'+f."""
    h = F.relu(F.conv2d(x[:, None], p[pre + "conv_1.weight"], p[pre + "conv_1.bias"], stride=2))
    h = F.relu(F.conv2d(h, p[pre + "conv_2.weight"], p[pre + "conv_2.bias"], stride=2))
    B, C, Fp, Tp = h.shape
    return h.permute(0, 3, 1, 2).reshape(B, Tp, C * Fp)


def encoder_forward(x: torch.Tensor, lengths: Optional[torch.Tensor], p: Params, n_blocks: int, n_heads: int,
                    pre: str = "encoder.", training: bool = False,
                    return_block_outputs: bool = False):
    """Encoder.forward, encoder.py:18-37.  Returns (y (B,T',d), lengths')."""
    h = conv_subsampling(x, p, pre + "downsampling_conv.")
    out_len = subsampled_lengths(lengths) if lengths is not None else None
    h = h @ p[pre + "linear.weight"].t() + p[pre + "linear.bias"]
    T = h.shape[1]
    if out_len is not None and int(out_len.max()) != T:
        # masking.py:9-12 + encoder.py:30 raise a broadcast error in this case
        raise RuntimeError(f"lengths.max() after subsampling ({int(out_len.max())}) must equal T'={T}")
    pe = relpos_table(T, p[pre + "rel_pe.div_term"])
    outs = []
    for li in range(n_blocks):
        h = conformer_block(h, pe, out_len, p, f"{pre}layers.{li}.", n_heads, training)
        if return_block_outputs:
            outs.append(h)
    if return_block_outputs:
        return h, out_len, outs
    return h, out_len


def encoder_forward_chunked(x: torch.Tensor, p: Params, n_blocks: int, n_heads: int, chunk_ends, pre: str = "encoder."):
    """Chunk-by-chunk (streaming) evaluation restated on the WHOLE sequence with masks.  The reference has no streaming
    code; the semantics are the prefix rule: an encoder frame of chunk c is computed, in every layer, from the frames of
    chunks <= c only -- self-attention sees keys < end(c), the depthwise convolution treats frames >= end(c) as zero
    padding (exactly what Encoder.forward does at the end of an utterance).  chunk_ends: increasing frame indices, the last
    one == T'.  With a single chunk this IS encoder_forward."""
    h = conv_subsampling(x, p, pre + "downsampling_conv.")
    h = h @ p[pre + "linear.weight"].t() + p[pre + "linear.bias"]
    T = h.shape[1]
    ends = [int(e) for e in chunk_ends]
    if ends[-1] != T or any(b <= a for a, b in zip(ends, ends[1:])):
        raise ValueError(f"chunk_ends must increase and finish at T'={T}")
    visible_end = torch.empty(T, dtype=torch.long)
    start = 0
    for e in ends:
        visible_end[start:e] = e
        start = e
    pe = relpos_table(T, p[pre + "rel_pe.div_term"])
    for li in range(n_blocks):
        h = conformer_block(h, pe, None, p, f"{pre}layers.{li}.", n_heads, visible_end=visible_end)
    return h


def lstm_layer(x: torch.Tensor, lengths: Optional[torch.Tensor], w_ih, w_hh, b_ih, b_hh) -> torch.Tensor:
    """One nn.LSTM(batch_first=True) layer over a packed batch (decoder.py:10,17-22), written out: gate order i|f|g|o,
    c_t = f*c + i*g, h_t = o*tanh(c_t); utterance b is advanced for lengths[b] frames only and later outputs are 0
    (pack_padded_sequence / pad_packed_sequence).  Checked against torch.nn.LSTM in tests/test_oracle_golden.py."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gx = x @ w_ih.t() + b_ih + b_hh
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    out = x.new_zeros(B, T, H)
    for t in range(T):
        g = gx[:, t] + h @ w_hh.t()
        i, f, gg, o = torch.sigmoid(g[:, :H]), torch.sigmoid(g[:, H:2 * H]), torch.tanh(g[:, 2 * H:3 * H]), torch.sigmoid(g[:, 3 * H:])
        cn = f * c + i * gg
        hn = o * torch.tanh(cn)
        live = torch.ones(B, 1, dtype=torch.bool) if lengths is None else (t < lengths)[:, None]
        c = torch.where(live, cn, c)
        h = torch.where(live, hn, h)
        out[:, t] = torch.where(live, hn, torch.zeros_like(hn))
    return out


def decoder_forward(x: torch.Tensor, lengths: Optional[torch.Tensor], p: Params, pre: str = "decoder.") -> torch.Tensor:
    """Decoder.forward in eval mode, decoder.py:15-27: packed LSTM -> Swish -> BatchNorm1d(running stats) -> Linear."""
    y = x
    k = 0
    while pre + f"lstm.weight_ih_l{k}" in p:
        y = lstm_layer(y, None if lengths is None else lengths.cpu(), p[pre + f"lstm.weight_ih_l{k}"], p[pre + f"lstm.weight_hh_l{k}"],
                       p[pre + f"lstm.bias_ih_l{k}"], p[pre + f"lstm.bias_hh_l{k}"])
        k += 1
    y = swish(y)
    y = (y - p[pre + "norm.running_mean"]) / torch.sqrt(p[pre + "norm.running_var"] + BN_EPS) \
        * p[pre + "norm.weight"] + p[pre + "norm.bias"]
    return y @ p[pre + "linear.weight"].t() + p[pre + "linear.bias"]


def conformer_forward(x, lengths, p: Params, n_blocks: int, n_heads: int):
    """Conformer.forward (eval), conformer.py:24-27."""
    h, out_len = encoder_forward(x, lengths, p, n_blocks, n_heads)
    return decoder_forward(h, out_len, p), out_len


def ctc_loss(logits: torch.Tensor, targets: torch.Tensor, in_len: torch.Tensor, tgt_len: torch.Tensor,
             blank: int = 0) -> torch.Tensor:
    """ConformerCriterion.ctc_loss, evaluation.py:12-16 (integer targets give the same value)."""
    lp = logits.float().log_softmax(-1).transpose(0, 1)
    return F.ctc_loss(lp, targets, in_len, tgt_len, blank=blank, zero_infinity=True)


def ctc_lattice(logits: torch.Tensor, targets: torch.Tensor, in_len, tgt_len, blank: int = 0):
    """The same loss written out (Graves et al. 2006 forward-backward over the blank-extended label sequence, which is
    what nn.CTCLoss of evaluation.py:10 computes), float64: returns (loss, nll (B,), dloss/dlogits (B,T,V)) with
    reduction='mean' (mean over utterances of nll / max(target_length, 1)) and zero_infinity=True.  Pinned against
    F.ctc_loss and its autograd in tests/test_oracle_golden.py."""
    import numpy as np
    x = logits.detach().double().cpu().numpy()
    B, T, V = x.shape
    tg = targets.detach().cpu().numpy().astype(np.int64)
    il = [int(v) for v in torch.as_tensor(in_len).tolist()]
    tl = [int(v) for v in torch.as_tensor(tgt_len).tolist()]
    offs = np.cumsum([0] + tl[:-1]) if tg.ndim == 1 else None
    mx = x.max(-1, keepdims=True)
    lp = x - (mx + np.log(np.exp(x - mx).sum(-1, keepdims=True)))
    nll = np.zeros(B)
    grad = np.zeros_like(x)

    def lse(*a):
        m = np.maximum.reduce(a)
        ms = np.where(np.isfinite(m), m, 0.0)
        with np.errstate(divide="ignore"):
            return np.where(np.isfinite(m), ms + np.log(sum(np.exp(v - ms) for v in a)), -np.inf)

    for b in range(B):
        Tb, L = il[b], tl[b]
        lab = tg[b, :L] if tg.ndim == 2 else tg[offs[b]:offs[b] + L]
        ext = np.full(2 * L + 1, blank, dtype=np.int64)
        ext[1::2] = lab
        S = ext.size
        if Tb == 0:
            nll[b] = 0.0 if L == 0 else np.inf
            continue
        hop = np.zeros(S, dtype=bool)                  # state s reachable from s-2
        hop[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])
        em = lp[b, :Tb][:, ext]                        # (Tb, S)
        al = np.full((Tb, S), -np.inf)
        al[0, 0] = em[0, 0]
        if S > 1:
            al[0, 1] = em[0, 1]
        ninf1, ninf2 = np.array([-np.inf]), np.array([-np.inf, -np.inf])
        for t in range(1, Tb):
            p = al[t - 1]
            s1 = np.concatenate([ninf1, p[:-1]])
            s2 = np.where(hop, np.concatenate([ninf2, p[:-2]])[:S], -np.inf)
            al[t] = lse(p, s1, s2) + em[t]
        be = np.full((Tb, S), -np.inf)
        be[Tb - 1, S - 1] = em[Tb - 1, S - 1]
        if S > 1:
            be[Tb - 1, S - 2] = em[Tb - 1, S - 2]
        hop_f = np.zeros(S, dtype=bool)                # state s may go to s+2
        hop_f[:-2] = hop[2:]
        for t in range(Tb - 2, -1, -1):
            n = be[t + 1]
            s1 = np.concatenate([n[1:], ninf1])
            s2 = np.where(hop_f, np.concatenate([n[2:], ninf2])[:S], -np.inf)
            be[t] = lse(n, s1, s2) + em[t]
        ll = lse(al[Tb - 1, S - 1:S], al[Tb - 1, S - 2:S - 1] if S > 1 else ninf1)[0]
        nll[b] = -ll
        if not np.isfinite(ll):
            continue
        occ = np.exp(al + be - ll - em)                # posterior state occupancy (Tb, S)
        g = np.exp(lp[b, :Tb])                         # softmax
        for s in range(S):
            g[:, ext[s]] -= occ[:, s]
        grad[b, :Tb] = g / (B * max(L, 1))
    fin = np.isfinite(nll)
    loss = float(np.sum(np.where(fin, nll, 0.0) / np.maximum(np.array(tl), 1)) / B)
    return loss, torch.from_numpy(nll), torch.from_numpy(grad)


def greedy_indices(logits: torch.Tensor) -> torch.Tensor:
    """Per-frame argmax of processor.py:302-303 (the 'CTC alignment indices')."""
    return logits.argmax(-1)


# --------------------------------------------------------------------------- deterministic weights
def make_params(vocab: int, n_mel: int, n_blocks: int, d: int, n_heads: int, ksize: int, lstm_hidden: int,
                seed: int, dtype=torch.float32, with_decoder: bool = True) -> Params:
    """Deterministic random weights with the reference's state_dict keys/shapes.

    numpy RandomState stream => identical on any machine, so large configs never need
    to be stored.  Scales are fan-in style so activations stay O(1) through 16 blocks;
    BN running stats are non-trivial so eval-mode BN is actually exercised.
    """
    import numpy as np
    rs = np.random.RandomState(seed)
    P: Params = {}

    def rnd(name, shape, scale):
        P[name] = torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float64)).to(dtype)

    def lin(name, out_f, in_f, extra=()):
        rnd(name + ".weight", (out_f, in_f) + tuple(extra), 1.0 / math.sqrt(in_f * max(1, int(np.prod(extra)) if extra else 1)))
        rnd(name + ".bias", (out_f,), 0.05)

    def ln(name, n):
        P[name + ".weight"] = torch.from_numpy(1.0 + 0.1 * rs.standard_normal(n)).to(dtype)
        P[name + ".bias"] = torch.from_numpy(0.05 * rs.standard_normal(n)).to(dtype)

    def bn(name, n):
        ln(name, n)
        P[name + ".running_mean"] = torch.from_numpy(0.1 * rs.standard_normal(n)).to(dtype)
        P[name + ".running_var"] = torch.from_numpy(0.5 + rs.uniform(0, 1, n)).to(dtype)
        P[name + ".num_batches_tracked"] = torch.tensor(3, dtype=torch.int64)

    fp = subsampled_frames(n_mel)
    e = "encoder."
    lin(e + "downsampling_conv.conv_1", d, 1, (3, 3))
    lin(e + "downsampling_conv.conv_2", d, d, (3, 3))
    lin(e + "linear", d, d * fp)
    P[e + "rel_pe.div_term"] = torch.exp(torch.arange(0, d, 2) * -(math.log(10000.0) / d)).unsqueeze(0).to(dtype)
    dh = d // n_heads
    for i in range(n_blocks):
        b = f"{e}layers.{i}."
        for f in ("ffn_1.", "ffn_2."):
            ln(b + f + "layer_norm", d)
            lin(b + f + "hidden_linear", 4 * d, d)
            lin(b + f + "out_linear", d, 4 * d)
        ln(b + "attention.layer_norm", d)
        a = b + "attention.attention."
        rnd(a + "content_bias", (n_heads, dh), 0.2)
        rnd(a + "position_bias", (n_heads, dh), 0.2)
        for nm in ("query_proj", "key_proj", "value_proj", "pos_proj", "out_proj"):
            lin(a + nm, d, d)
        ln(b + "conv.layer_norm", d)
        lin(b + "conv.pointwise_conv_1", 2 * d, d, (1,))
        lin(b + "conv.deepwise_conv", d, 1, (ksize,))
        bn(b + "conv.batch_norm", d)
        lin(b + "conv.pointwise_conv_2", d, d, (1,))
        ln(b + "layer_norm", d)
    if with_decoder:
        dd = "decoder."
        rnd(dd + "lstm.weight_ih_l0", (4 * lstm_hidden, d), 1.0 / math.sqrt(d))
        rnd(dd + "lstm.weight_hh_l0", (4 * lstm_hidden, lstm_hidden), 1.0 / math.sqrt(lstm_hidden))
        rnd(dd + "lstm.bias_ih_l0", (4 * lstm_hidden,), 0.05)
        rnd(dd + "lstm.bias_hh_l0", (4 * lstm_hidden,), 0.05)
        bn(dd + "norm", lstm_hidden)
        lin(dd + "linear", vocab, lstm_hidden)
    return P


# --------------------------------------------------------------------------- "next" rows: N2 Adam, N4 greedy decode
def greedy_decode_ids(logits: torch.Tensor, pad_id: int, unk_id: int):
    """Token-id level restatement of ConformerProcessor.greedy_decode (processor.py:301-318) for one (T,V) utterance:
    argmax per frame, skip pad/unk (WITHOUT resetting the repeat filter), drop consecutive repeats."""
    ids = logits.argmax(-1).tolist()
    out, prev = [], None
    for t in ids:
        if t == pad_id or t == unk_id:
            continue
        if prev is None or prev != t:
            prev = t
            out.append(t)
    return ids, out


def adam_reference(params, grads_per_step, lr=2e-5, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam (train.py:188) on CPU float64 tensors: returns the parameters after len(grads_per_step) steps."""
    ps = [p.clone().double().requires_grad_(True) for p in params]
    opt = torch.optim.Adam(ps, lr=lr, betas=betas, eps=eps)
    for grads in grads_per_step:
        for p, g in zip(ps, grads):
            p.grad = g.double().clone()
        opt.step()
    return [p.detach() for p in ps]
