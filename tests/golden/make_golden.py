#!/usr/bin/env python3
"""Generate the golden vectors in this directory by running the REFERENCE itself on CPU.

Run once in the build container (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports ``/root/reference/model`` read-only, builds the reference nn.Modules, loads the
deterministic weights of ``oracle.conformer_oracle.make_params`` into them (strict
``load_state_dict`` -- this also pins the state_dict key/shape contract), runs seeded fp32
CPU forward (and autograd backward) and stores inputs + outputs as small ``.npz`` files.
Weights are NOT stored: tests regenerate them from the same seed.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
# The repo's own ``model/`` (a regular package) would shadow the reference's ``model/`` (a namespace package) if the repo
# root were importable, so only /root/reference goes on sys.path and the oracle is loaded by file path.
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
sys.path.insert(0, "/root/reference")
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("conformer_oracle", os.path.join(ROOT, "oracle", "conformer_oracle.py"))
O = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(O)

from model.conformer import Conformer  # noqa: E402  (reference)
from model.modules.encoder import Encoder  # noqa: E402
from model.utils.attention import MultiHeadSelfAttentionModule  # noqa: E402
from model.utils.block import ConformerBlock  # noqa: E402
from model.utils.convolution import ConvolutionModule, ConvolutionSubsampling  # noqa: E402
from model.utils.ffn import FeedForwardModule  # noqa: E402
from model.utils.masking import generate_padding_mask  # noqa: E402
from model.utils.position import RelativePositionalEncoding  # noqa: E402

torch.set_num_threads(8)


def sub(P, prefix):
    return {k[len(prefix):]: v.clone() for k, v in P.items() if k.startswith(prefix)}


def npy(t):
    return t.detach().cpu().numpy()


def save(name, meta, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def grads_of(out, tensors):
    g = torch.autograd.grad(out, tensors, allow_unused=True)
    return [torch.zeros_like(t) if gi is None else gi for gi, t in zip(g, tensors)]


def module_cases(tag, d, H, K, B, T, lengths, seed, param_grads=True):
    """Per-module goldens at one (d,H,K,B,T') point; lengths are encoder-frame lengths."""
    cfg = dict(vocab=11, n_mel=80, n_blocks=1, d=d, n_heads=H, ksize=K, lstm_hidden=16, seed=seed)
    P = O.make_params(**cfg)
    blk = "encoder.layers.0."
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, T, d, generator=g)
    w = torch.randn(B, T, d, generator=g)          # cotangent for the backward goldens
    L = torch.tensor(lengths, dtype=torch.int64)
    mask = (~generate_padding_mask(L))[:, None, None, :]
    rel = RelativePositionalEncoding(d)
    rel.load_state_dict({"div_term": P["encoder.rel_pe.div_term"]})
    pe = rel(x)
    out = dict(x=npy(x), w=npy(w), lengths=npy(L), pe=npy(pe[0]))

    def run(mod, prefix, fn, key):
        mod.load_state_dict(sub(P, prefix))
        xi = x.clone().requires_grad_(True)
        y = fn(mod, xi)
        names = [n for n, p_ in mod.named_parameters() if p_.requires_grad]
        gs = grads_of((y * w).sum(), [xi] + [dict(mod.named_parameters())[n] for n in names])
        out[key + "_y"] = npy(y)
        out[key + "_dx"] = npy(gs[0])
        for n, gi in zip(names, gs[1:]):
            if param_grads or gi.numel() <= 4 * d:      # big configs keep only vector-sized grads
                out[f"{key}_d.{n}"] = npy(gi)
        return mod

    run(FeedForwardModule(d).eval(), blk + "ffn_1.", lambda m, xi: m(xi), "ffn")
    run(MultiHeadSelfAttentionModule(d, H).eval(), blk + "attention.", lambda m, xi: m(xi, pe, mask), "mhsa")
    run(ConvolutionModule(d, K).eval(), blk + "conv.", lambda m, xi: m(xi), "conv_eval")
    cm = run(ConvolutionModule(d, K).train(), blk + "conv.", lambda m, xi: m(xi), "conv_train")
    out["conv_train_running_mean"] = npy(cm.batch_norm.running_mean)
    out["conv_train_running_var"] = npy(cm.batch_norm.running_var)
    run(ConformerBlock(d, H, K).eval(), blk, lambda m, xi: m(xi, pe, mask), "block")
    # no-mask variant of attention (mask=None path, encoder.py:28-30)
    m = MultiHeadSelfAttentionModule(d, H).eval()
    m.load_state_dict(sub(P, blk + "attention."))
    out["mhsa_nomask_y"] = npy(m(x, pe, None))
    save(f"modules_{tag}", dict(cfg, B=B, T=T), **out)


def stream_mask_case(tag, d, H, B, T, chunk_ends, seed):
    """The attention half of the streaming PREFIX RULE pinned to the reference: MultiHeadSelfAttentionModule accepts any
    broadcastable bool mask (attention.py:60-62, True = hidden), so a block-triangular (1, 1, T', T') mask -- row i hides keys
    >= the end of its own chunk -- gives the reference's answer for 'chunk c attends to chunks <= c'.  (The depthwise-conv half
    of the rule has no reference counterpart: nn.Conv1d cannot be masked per output row.)"""
    cfg = dict(vocab=11, n_mel=80, n_blocks=1, d=d, n_heads=H, ksize=15, lstm_hidden=16, seed=seed)
    P = O.make_params(**cfg)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, T, d, generator=g)
    rel = RelativePositionalEncoding(d)
    rel.load_state_dict({"div_term": P["encoder.rel_pe.div_term"]})
    pe = rel(x)
    visible_end = torch.empty(T, dtype=torch.long)
    start = 0
    for e in chunk_ends:
        visible_end[start:e] = e
        start = e
    assert start == T
    mask = (torch.arange(T)[None, :] >= visible_end[:, None])[None, None]     # (1, 1, T, T), True = hidden
    m = MultiHeadSelfAttentionModule(d, H).eval()
    m.load_state_dict(sub(P, "encoder.layers.0.attention."))
    with torch.no_grad():
        y = m(x, pe, mask)
    save(f"stream_mhsa_{tag}", dict(cfg, B=B, T=T), x=npy(x), chunk_ends=np.array(chunk_ends, dtype=np.int64), mhsa_y=npy(y))


def stem_case(tag, d, B, T, seed):
    cfg = dict(vocab=11, n_mel=80, n_blocks=0, d=d, n_heads=1, ksize=3, lstm_hidden=8, seed=seed)
    P = O.make_params(**cfg)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, 80, T, generator=g)
    m = ConvolutionSubsampling(d).eval()
    m.load_state_dict(sub(P, "encoder.downsampling_conv."))
    L = torch.tensor([T] + [max(7, T - 13 * i) for i in range(1, B)], dtype=torch.int64)
    y, L2 = m(x, L)
    save(f"stem_{tag}", dict(cfg, B=B, T=T), x=npy(x), y=npy(y), lengths=npy(L), out_lengths=npy(L2))


def model_case(tag, vocab, n_blocks, d, H, K, hid, B, T, lengths, seed, tgt_len, with_grads):
    cfg = dict(vocab=vocab, n_mel=80, n_blocks=n_blocks, d=d, n_heads=H, ksize=K, lstm_hidden=hid, seed=seed)
    P = O.make_params(**cfg)
    model = Conformer(vocab, 80, n_blocks, d, H, K, hid, 1, 0.0).eval()
    model.load_state_dict(P, strict=True)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, 80, T, generator=g)
    L = torch.tensor(lengths, dtype=torch.int64)
    out = dict(x=npy(x), lengths=npy(L))
    # per-block encoder outputs via forward hooks
    blocks = []
    hooks = [l.register_forward_hook(lambda m_, i_, o_: blocks.append(npy(o_))) for l in model.encoder.layers]
    with torch.no_grad():
        enc, L2 = model.encoder(x, L)
        logits, _ = model(x, L)
    for h in hooks:
        h.remove()
    blocks = blocks[:n_blocks]
    out.update(enc=npy(enc), out_lengths=npy(L2), logits=npy(logits), argmax=npy(logits.argmax(-1)))
    for i in (0, n_blocks - 1):
        out[f"block{i}"] = blocks[i]
    tg = torch.randint(1, vocab, (B, max(tgt_len)), generator=g)
    TL = torch.tensor(tgt_len, dtype=torch.int64)
    lp = logits.float().log_softmax(-1).transpose(0, 1)
    loss = torch.nn.CTCLoss(blank=0, zero_infinity=True)(lp, tg.float(), L2, TL)   # evaluation.py:10-16
    out.update(targets=npy(tg), target_lengths=npy(TL), ctc=npy(loss))
    # logits with mask=None path
    with torch.no_grad():
        enc_nm, _ = model.encoder(x, None)
    out["enc_nomask"] = npy(enc_nm)
    if with_grads:
        # encoder-only backward in eval mode (BN running stats): d(sum(enc*w))/d(param)
        w = torch.randn(enc.shape, generator=g)
        model.zero_grad()
        e2, _ = model.encoder(x, L)
        (e2 * w).sum().backward()
        out["w"] = npy(w)
        for n, p_ in model.encoder.named_parameters():
            if p_.grad is not None:
                out["grad." + n] = npy(p_.grad)
    save(f"model_{tag}", dict(cfg, B=B, T=T), **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    if sys.argv[1:] == ["stream"]:             # (added in round 3: only the streaming-mask goldens, the others stay as committed)
        stream_mask_case("d64_t70", d=64, H=4, B=2, T=70, chunk_ends=[16, 32, 37, 70], seed=51)
        stream_mask_case("d144_t49", d=144, H=4, B=1, T=49, chunk_ends=[1, 2, 40, 49], seed=52)
        sys.exit(0)
    # T' in {1, 7, 48, 49}; ragged lengths; odd head dims; K=31 and a small K
    module_cases("d32_t7", d=32, H=4, K=31, B=3, T=7, lengths=[7, 5, 1], seed=11)
    module_cases("d32_t48", d=32, H=4, K=31, B=2, T=48, lengths=[48, 33], seed=12)
    module_cases("d32_t1", d=32, H=4, K=7, B=2, T=1, lengths=[1, 1], seed=13)
    module_cases("d144_t49", d=144, H=4, K=31, B=2, T=49, lengths=[49, 39], seed=14, param_grads=False)
    module_cases("d64_t70", d=64, H=1, K=15, B=2, T=70, lengths=[70, 2], seed=15, param_grads=False)
    stem_case("d32", d=32, B=3, T=57, seed=21)
    stem_case("d144", d=144, B=1, T=31, seed=22)
    # tiny end-to-end with grads; T=200 -> T'=49; T=31 -> T'=7
    model_case("tiny", vocab=17, n_blocks=2, d=32, H=4, K=31, hid=24, B=3, T=103, lengths=[103, 80, 31],
               seed=31, tgt_len=[5, 4, 2], with_grads=True)
    # BASELINE cfg-1: Conformer-S, B=2, T=200, lengths [200,160]
    model_case("cfg1_S", vocab=370, n_blocks=4, d=144, H=4, K=31, hid=320, B=2, T=200, lengths=[200, 160],
               seed=41, tgt_len=[12, 9], with_grads=False)
    stream_mask_case("d64_t70", d=64, H=4, B=2, T=70, chunk_ends=[16, 32, 37, 70], seed=51)
    stream_mask_case("d144_t49", d=144, H=4, B=1, T=49, chunk_ends=[1, 2, 40, 49], seed=52)
