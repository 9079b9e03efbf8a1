"""Shared by tests/test_autocast_golden_gpu.py and tools/autocast_report.py: runs the gfx950 modules under
``torch.autocast('cuda', bfloat16)`` on the inputs of the ``autocast_*`` goldens and returns, per tensor, three numbers:

* ``ours``  = rel-L2(HIP bf16-autocast result, reference fp32 result)      -- the north_star's "1e-2 bf16 per tensor"
* ``ref``   = rel-L2(reference bf16-autocast result, reference fp32 result) -- the reference's own autocast drift
* ``cross`` = rel-L2(HIP bf16-autocast result, reference bf16-autocast result)

The reference results come from ``tests/golden/make_golden_autocast.py`` (the reference's own modules and autograd on CPU)."""
import torch

from tests.util import cfg_params, golden_find, golden_pick, load_golden, rel_l2

ZERO_GRAD = 1e-4          # |reference fp32 gradient| below this: numerically zero
# Gradients that are zero by construction; what the kernels return for them is pure rounding noise, and so is what the
# reference returns (under autocast its "relative error" on them is 1e3..1e4):
#  * key_proj.bias / pos_proj.bias: a bias on the keys (or on the projected positions) adds one constant per query row to
#    every score of that row, and softmax is invariant to that (attention.py:49-66);
#  * deepwise_conv.bias in front of a TRAIN-mode BatchNorm: the batch mean removes any per-channel constant
#    (convolution.py:26-27).
# They are compared in absolute terms against the scale of a sibling gradient that is not zero.
MATH_ZERO = (("key_proj.bias", "query_proj.bias", None), ("pos_proj.bias", "query_proj.bias", None),
             ("deepwise_conv.bias", "batch_norm.bias", "train"))


def _math_zero_sibling(name):
    for suffix, sibling, only in MATH_ZERO:
        if name.endswith(suffix) and (only is None or only in name):
            return name[: -len(suffix)] + sibling
    return None


def _entry(rows, name, ours_t, g, stem):
    k32, k16 = golden_find(g, stem.replace("{p}", "f32")), golden_find(g, stem.replace("{p}", "bf16"))
    if k32 is None:
        return
    ref32, ref16 = g[k32], g[k16]
    got = golden_pick(ours_t, k32).float().cpu()
    sib = _math_zero_sibling(name)
    if sib is not None or float(ref32.norm()) < ZERO_GRAD:
        rms = lambda t: float(t.double().pow(2).mean().sqrt())
        rows.append(dict(tensor=name, zero=True, ours_abs=float(got.abs().max()), ref_abs=float(ref16.abs().max()),
                         ours_rms=rms(got), ref_rms=rms(ref16), sibling_scale=_sibling_scale(g, stem, name, sib),
                         sibling_rms=_sibling_scale(g, stem, name, sib, rms=True)))
        return
    rows.append(dict(tensor=name, zero=False, ours=rel_l2(got, ref32), ref=rel_l2(ref16, ref32), cross=rel_l2(got, ref16)))


def _sibling_scale(g, stem, name, sib, rms=False):
    """max (or rms) |reference fp32 gradient| of the non-zero sibling parameter of a mathematically-zero gradient (or None)."""
    if sib is None:
        return None
    suffix = next(s for s, _, _ in MATH_ZERO if name.endswith(s))
    sib_suffix = sib[len(name) - len(suffix):]
    k = golden_find(g, stem.replace("{p}", "f32")[: -len(suffix)] + sib_suffix)
    if k is None:
        return None
    return float(g[k].double().pow(2).mean().sqrt()) if rms else float(g[k].abs().max())


def module_rows(case, dev, dtype=torch.bfloat16):
    from model.utils.attention import MultiHeadSelfAttentionModule
    from model.utils.block import ConformerBlock
    from model.utils.convolution import ConvolutionModule
    from model.utils.ffn import FeedForwardModule
    from model.utils.position import RelativePositionalEncoding
    meta, g = load_golden(case)
    P = cfg_params(meta)
    d, H, K = meta["d"], meta["n_heads"], meta["ksize"]
    blk = "encoder.layers.0."
    L = g["lengths"].to(dev)
    mask = (torch.arange(meta["T"], device=dev)[None, :] >= L[:, None])[:, None, None, :]
    w = g["w"].to(dev)
    rel = RelativePositionalEncoding(d).to(dev)
    rel.load_state_dict({"div_term": P["encoder.rel_pe.div_term"]})
    with torch.no_grad():
        pe = rel(g["x"].to(dev))
    rows = []

    def run(mod, prefix, fn, key, train=False):
        mod.load_state_dict({k[len(prefix):]: v for k, v in P.items() if k.startswith(prefix)})
        mod = mod.to(dev)
        mod.train() if train else mod.eval()
        x = g["x"].to(dev).clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=dtype):
            y = fn(mod, x)
        (y.float() * w).sum().backward()
        _entry(rows, f"{key}.y", y, g, f"{key}.{{p}}.y")
        _entry(rows, f"{key}.dx", x.grad, g, f"{key}.{{p}}.dx")
        for n, p in mod.named_parameters():
            if p.requires_grad and p.grad is not None:
                _entry(rows, f"{key}.d.{n}", p.grad, g, f"{key}.{{p}}.d.{n}")
        if train:
            _entry(rows, f"{key}.running_mean", mod.batch_norm.running_mean, g, f"{key}.{{p}}.running_mean")
            _entry(rows, f"{key}.running_var", mod.batch_norm.running_var, g, f"{key}.{{p}}.running_var")

    run(FeedForwardModule(d), blk + "ffn_1.", lambda m, x: m(x), "ffn")
    run(MultiHeadSelfAttentionModule(d, H), blk + "attention.", lambda m, x: m(x, pe, mask), "mhsa")
    run(ConvolutionModule(d, K), blk + "conv.", lambda m, x: m(x), "conv_eval")
    run(ConvolutionModule(d, K), blk + "conv.", lambda m, x: m(x), "conv_train", train=True)
    run(ConformerBlock(d, H, K), blk, lambda m, x: m(x, pe, mask), "block")
    return rows


def model_rows(case, dev, dtype=torch.bfloat16):
    """Eval-mode encoder / logits under autocast, and one training step as train.py:225,232-240 writes it.
    ``dtype=None`` runs the fp32 path instead (``ours`` is then the fp32 kernels' distance to the reference's fp32 results)."""
    ac = lambda: torch.autocast("cuda", dtype=dtype or torch.bfloat16, enabled=dtype is not None)
    from conformer_amd.evaluation import ConformerCriterion
    from model.conformer import Conformer
    meta, g = load_golden(case)
    P = cfg_params(meta)
    mk = lambda: Conformer(meta["vocab"], 80, meta["n_blocks"], meta["d"], meta["n_heads"], meta["ksize"], meta["lstm_hidden"], 1, 0.0)
    if "x" in g:
        x = g["x"]
    else:                      # big case: the input is regenerated from the seed (CPU generator) and spot-checked
        gen = torch.Generator().manual_seed(meta["seed"] + 1)
        x = torch.randn(meta["B"], 80, meta["T"], generator=gen)
        assert torch.equal(x.flatten()[::997], g["x_check"]), "torch.randn stream differs from the golden's build"
    x, L = x.to(dev), g["lengths"].to(dev)
    rows = []
    m = mk(); m.load_state_dict(P, strict=True); m = m.to(dev).eval()
    with torch.no_grad(), ac():
        enc, L2 = m.encoder(x, L)
        logits, _ = m(x, L)
    _entry(rows, "eval.enc", enc, g, "eval.{p}.enc")
    _entry(rows, "eval.logits", logits, g, "eval.{p}.logits")
    am = logits.float().argmax(-1).cpu()
    rows.append(dict(tensor="eval.argmax_mismatch", zero=False,
                     ours=float((am != g["eval.f32.argmax"]).float().mean()),
                     ref=float((g["eval.bf16.argmax"] != g["eval.f32.argmax"]).float().mean()),
                     cross=float((am != g["eval.bf16.argmax"]).float().mean())))
    m = mk(); m.load_state_dict(P, strict=True); m = m.to(dev).train()
    crit = ConformerCriterion(blank_id=0)
    with ac():
        out, xl = m(x, L)
        with torch.autocast("cuda", enabled=False):
            loss = crit.ctc_loss(out, g["targets"].to(dev), xl, g["target_lengths"].to(dev))
    loss.backward()
    l32, l16 = float(g["train.f32.loss"]), float(g["train.bf16.loss"])
    rows.append(dict(tensor="train.loss", zero=False, ours=abs(float(loss) - l32) / abs(l32), ref=abs(l16 - l32) / abs(l32),
                     cross=abs(float(loss) - l16) / abs(l16)))
    _entry(rows, "train.logits", out, g, "train.{p}.logits")
    for n, p in m.named_parameters():
        if p.grad is not None:
            _entry(rows, f"train.grad.{n}", p.grad, g, f"train.{{p}}.grad.{n}")
    for n, b in m.named_buffers():
        if n.endswith("running_mean") or n.endswith("running_var"):
            _entry(rows, f"train.buf.{n}", b, g, f"train.{{p}}.buf.{n}")
    return rows
