#!/usr/bin/env python3
"""Goldens of the REFERENCE under ``torch.autocast('cpu', bfloat16)`` next to its fp32 results on the same inputs
("also the reference's CPU-bf16-autocast outputs"; the reference trains under autocast, train.py:232-240).

Run once in the build container (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_autocast.py

Writes ``autocast_*.npz``:
* ``autocast_modules_<tag>``: per module (FFN, MHSA, conv eval / train-BN, whole block) -- output, input gradient and every
  parameter gradient, once in fp32 and once under bf16 autocast, from the reference's own autograd;
* ``autocast_model_tiny`` / ``autocast_model_cfg1_S``: full model, eval-mode encoder output + logits under autocast, and one
  TRAINING step as train.py:232-240 writes it (train-mode BatchNorm, dropout 0, model under autocast, CTC loss in fp32
  outside autocast, backward): loss + every parameter gradient, fp32 and bf16.

Tensors with more than ``SAMPLE_ABOVE`` elements are stored as a strided sample ``flat[3::p]`` of at most ~``SAMPLE_ABOVE``
elements (p prime, key suffix ``@s<p>``); the tests take the same sample of the HIP result (``tests/util.py::golden_pick``).  Weights are never stored (regenerated from the seed).
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
from model.utils.attention import MultiHeadSelfAttentionModule  # noqa: E402
from model.utils.block import ConformerBlock  # noqa: E402
from model.utils.convolution import ConvolutionModule  # noqa: E402
from model.utils.ffn import FeedForwardModule  # noqa: E402
from model.utils.masking import generate_padding_mask  # noqa: E402
from model.utils.position import RelativePositionalEncoding  # noqa: E402

torch.set_num_threads(8)
SAMPLE_ABOVE = 8000
PRIMES = (3, 7, 13, 29, 61, 127, 251, 509, 1021, 2039, 4093, 8191, 16381)


def sub(P, prefix):
    return {k[len(prefix):]: v.clone() for k, v in P.items() if k.startswith(prefix)}


def put(out, key, t, cap=None):
    t = t.detach().float().cpu()
    cap = cap or SAMPLE_ABOVE
    if t.numel() > cap:
        p = next(q for q in PRIMES if t.numel() / q <= cap)
        out[f"{key}@s{p}"] = t.flatten()[3::p].contiguous().numpy()
    else:
        out[key] = t.numpy()


def save(name, meta, arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def amp(on):
    return torch.autocast("cpu", dtype=torch.bfloat16, enabled=on)


def module_cases(tag, d, H, K, B, T, lengths, seed):
    cfg = dict(vocab=11, n_mel=80, n_blocks=1, d=d, n_heads=H, ksize=K, lstm_hidden=16, seed=seed)
    P = O.make_params(**cfg)
    blk = "encoder.layers.0."
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, T, d, generator=g)
    w = torch.randn(B, T, d, generator=g)
    L = torch.tensor(lengths, dtype=torch.int64)
    mask = (~generate_padding_mask(L))[:, None, None, :]
    rel = RelativePositionalEncoding(d)
    rel.load_state_dict({"div_term": P["encoder.rel_pe.div_term"]})
    pe = rel(x)
    out = dict(x=x.numpy(), w=w.numpy(), lengths=L.numpy())

    def run(make, prefix, fn, key):
        for prec in ("f32", "bf16"):
            mod = make()
            mod.load_state_dict(sub(P, prefix))
            xi = x.clone().requires_grad_(True)
            with amp(prec == "bf16"):
                y = fn(mod, xi)
            names = [n for n, p_ in mod.named_parameters() if p_.requires_grad]
            ps = dict(mod.named_parameters())
            gs = torch.autograd.grad((y.float() * w).sum(), [xi] + [ps[n] for n in names], allow_unused=True)
            put(out, f"{key}.{prec}.y", y)
            put(out, f"{key}.{prec}.dx", gs[0])
            for n, gi in zip(names, gs[1:]):
                put(out, f"{key}.{prec}.d.{n}", torch.zeros_like(ps[n]) if gi is None else gi)
            if key == "conv_train":
                put(out, f"{key}.{prec}.running_mean", mod.batch_norm.running_mean)
                put(out, f"{key}.{prec}.running_var", mod.batch_norm.running_var)

    run(lambda: FeedForwardModule(d).eval(), blk + "ffn_1.", lambda m, xi: m(xi), "ffn")
    run(lambda: MultiHeadSelfAttentionModule(d, H).eval(), blk + "attention.", lambda m, xi: m(xi, pe, mask), "mhsa")
    run(lambda: ConvolutionModule(d, K).eval(), blk + "conv.", lambda m, xi: m(xi), "conv_eval")
    run(lambda: ConvolutionModule(d, K).train(), blk + "conv.", lambda m, xi: m(xi), "conv_train")
    run(lambda: ConformerBlock(d, H, K).eval(), blk, lambda m, xi: m(xi, pe, mask), "block")
    save(f"autocast_modules_{tag}", dict(cfg, B=B, T=T), out)


def model_case(tag, vocab, n_blocks, d, H, K, hid, B, T, lengths, seed, tgt_len, cap=None):
    cfg = dict(vocab=vocab, n_mel=80, n_blocks=n_blocks, d=d, n_heads=H, ksize=K, lstm_hidden=hid, seed=seed)
    P = O.make_params(**cfg)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, 80, T, generator=g)
    L = torch.tensor(lengths, dtype=torch.int64)
    tg = torch.randint(1, vocab, (B, max(tgt_len)), generator=g)
    TL = torch.tensor(tgt_len, dtype=torch.int64)
    out = dict(lengths=L.numpy(), targets=tg.numpy(), target_lengths=TL.numpy())
    if cap is None:
        out["x"] = x.numpy()          # (the big case regenerates x from the seed: same torch build, CPU generator)
    else:
        out["x_check"] = x.flatten()[::997].numpy()
    ctc = torch.nn.CTCLoss(blank=0, zero_infinity=True)                      # evaluation.py:10
    for prec in ("f32", "bf16"):
        model = Conformer(vocab, 80, n_blocks, d, H, K, hid, 1, 0.0).eval()
        model.load_state_dict(P, strict=True)
        with torch.no_grad(), amp(prec == "bf16"):
            enc, L2 = model.encoder(x, L)
            logits, _ = model(x, L)
        put(out, f"eval.{prec}.enc", enc, cap)
        put(out, f"eval.{prec}.logits", logits, cap)
        out[f"eval.{prec}.argmax"] = logits.float().argmax(-1).numpy()
        out["out_lengths"] = L2.numpy()
        # one training step, train.py:225,232-240 (GradScaler is a no-op for bf16; dropout 0)
        model = Conformer(vocab, 80, n_blocks, d, H, K, hid, 1, 0.0).train()
        model.load_state_dict(P, strict=True)
        with amp(prec == "bf16"):
            outputs, xl = model(x, L)
            with torch.autocast("cpu", enabled=False):
                loss = ctc(outputs.float().log_softmax(dim=-1).transpose(0, 1), tg.float(), xl, TL)   # evaluation.py:12-16
        loss.backward()
        out[f"train.{prec}.loss"] = np.array(float(loss))
        put(out, f"train.{prec}.logits", outputs, cap)
        for n, p_ in model.named_parameters():
            if p_.grad is not None:
                put(out, f"train.{prec}.grad.{n}", p_.grad, cap)
        for n, b_ in model.named_buffers():
            if n.endswith("running_mean") or n.endswith("running_var"):
                put(out, f"train.{prec}.buf.{n}", b_, cap)
    save(f"autocast_model_{tag}", dict(cfg, B=B, T=T), out)


if __name__ == "__main__":
    torch.manual_seed(0)
    if len(sys.argv) > 1 and sys.argv[1] == "L":
        model_case("L_b4", vocab=370, n_blocks=16, d=512, H=8, K=31, hid=640, B=4, T=1000, lengths=[1000, 1000, 870, 640],
                   seed=51, tgt_len=[40, 40, 33, 21], cap=1500)
        sys.exit(0)
    module_cases("d32_t48", d=32, H=4, K=31, B=2, T=48, lengths=[48, 33], seed=12)
    module_cases("d144_t49", d=144, H=4, K=31, B=2, T=49, lengths=[49, 39], seed=14)
    # Conformer-L block geometry (cfg-2/3: d=512, H=8, K=31, T'=249)
    module_cases("d512_t249", d=512, H=8, K=31, B=2, T=249, lengths=[249, 131], seed=16)
    model_case("tiny", vocab=17, n_blocks=2, d=32, H=4, K=31, hid=24, B=3, T=103, lengths=[103, 80, 31], seed=31,
               tgt_len=[5, 4, 2])
    model_case("cfg1_S", vocab=370, n_blocks=4, d=144, H=4, K=31, hid=320, B=2, T=200, lengths=[200, 160], seed=41,
               tgt_len=[12, 9])
    # BASELINE cfg-3 geometry (Conformer-L: 16 blocks, d=512, H=8, lstm 640, vocab 370, T=1000 -> T'=249) at B=4
    model_case("L_b4", vocab=370, n_blocks=16, d=512, H=8, K=31, hid=640, B=4, T=1000, lengths=[1000, 1000, 870, 640], seed=51,
               tgt_len=[40, 40, 33, 21], cap=1500)
