"""Dropout on the training path: the counter-based mask is (a) statistically a Bernoulli(1-p) keep mask scaled by 1/(1-p),
(b) identical in the fused GEMM epilogues, the stand-alone kernel and the backward, so gradients equal those of an
fp64 torch model that uses the SAME (extracted) masks.  (The stream is not torch's Philox stream; reference parity runs
use p = 0.)"""
import math

import pytest
import torch

from oracle import conformer_oracle as O
from tests.util import rel_l2

pytestmark = pytest.mark.gpu
P_DROP = 0.25


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def mask_of(shape, p, seed, dev):
    from conformer_amd import ops
    return ops.dropout_apply(torch.ones(*shape, device=dev), p, seed).cpu().double()


def test_mask_statistics_and_determinism(dev):
    m = mask_of((1000, 512), P_DROP, 12345, dev)
    keep = (m > 0).double().mean().item()
    assert abs(keep - (1 - P_DROP)) < 0.005
    assert torch.allclose(m[m > 0], torch.tensor(1 / (1 - P_DROP), dtype=torch.float64), rtol=1e-6)
    assert torch.equal(m, mask_of((1000, 512), P_DROP, 12345, dev))
    assert not torch.equal(m, mask_of((1000, 512), P_DROP, 12346, dev))
    assert abs((m[:, 0] > 0).double().mean().item() - (1 - P_DROP)) < 0.06          # no structure along rows/cols
    assert abs((m[3] > 0).double().mean().item() - (1 - P_DROP)) < 0.08


def test_gemm_epilogue_masks_match_standalone(dev):
    from conformer_amd import ops
    M, N, K = 300, 192, 64
    a, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2) / 8, rnd(N, seed=3), rnd(M, N, seed=4)
    y = (a.double() @ w.double().t() + b.double())
    m = mask_of((M, N), P_DROP, 77, dev)
    got = ops.linear_train("bias", a.to(dev), w.to(dev), b.to(dev), drop_p=P_DROP, seed=77)
    assert rel_l2(got, y * m) < 2e-5
    got = ops.linear_train("residual", a.to(dev), w.to(dev), b.to(dev), residual=r.to(dev), alpha=0.5, drop_p=P_DROP, seed=77)
    assert rel_l2(got, 0.5 * y * m + r.double()) < 2e-5
    got, z = ops.linear_train("swish", a.to(dev), w.to(dev), b.to(dev), drop_p=P_DROP, seed=77, save_z=True)
    assert rel_l2(z, y) < 2e-5 and rel_l2(got, O.swish(y) * m) < 2e-5


def test_ffn_gradients_with_dropout(dev):
    from conformer_amd import ops
    from model.utils.ffn import FeedForwardModule
    d, B, T = 64, 3, 20
    P = O.make_params(vocab=5, n_mel=80, n_blocks=1, d=d, n_heads=4, ksize=7, lstm_hidden=4, seed=3, with_decoder=False)
    pre = "encoder.layers.0.ffn_1."
    mod = FeedForwardModule(d, dropout_rate=P_DROP)
    mod.load_state_dict({k[len(pre):]: v for k, v in P.items() if k.startswith(pre)})
    mod = mod.to(dev).train()
    x = rnd(B, T, d, seed=5)
    w = rnd(B, T, d, seed=6)
    torch.manual_seed(99)
    s1, s2 = ops.new_seeds(2)                       # the seeds the Function will draw after the same manual_seed
    torch.manual_seed(99)
    xg = x.to(dev).requires_grad_(True)
    y = mod.fused(xg, residual=xg, alpha=0.5)
    (y * w.to(dev)).sum().backward()
    # fp64 reference with the same masks
    m1, m2 = mask_of((B * T, 4 * d), P_DROP, s1, dev).view(B, T, 4 * d), mask_of((B * T, d), P_DROP, s2, dev).view(B, T, d)
    Pd = {k[len(pre):]: v.double().requires_grad_(True) for k, v in P.items() if k.startswith(pre)}
    xd = x.double().requires_grad_(True)
    h = O.layer_norm(xd, Pd["layer_norm.weight"], Pd["layer_norm.bias"])
    h = O.swish(h @ Pd["hidden_linear.weight"].t() + Pd["hidden_linear.bias"]) * m1
    ref = 0.5 * ((h @ Pd["out_linear.weight"].t() + Pd["out_linear.bias"]) * m2) + xd
    (ref * w.double()).sum().backward()
    assert rel_l2(y, ref) < 2e-5
    assert rel_l2(xg.grad, xd.grad) < 5e-5
    for n, p_ in mod.named_parameters():
        assert rel_l2(p_.grad, Pd[n].grad) < 5e-5, n


def test_attention_gradients_with_dropout(dev):
    from conformer_amd import ops
    from model.utils.attention import MultiHeadSelfAttentionModule
    d, H, B, T = 32, 4, 2, 12
    dh = d // H
    P = O.make_params(vocab=5, n_mel=80, n_blocks=1, d=d, n_heads=H, ksize=7, lstm_hidden=4, seed=4, with_decoder=False)
    pre = "encoder.layers.0.attention."
    mod = MultiHeadSelfAttentionModule(d, H, dropout_rate=P_DROP)
    mod.load_state_dict({k[len(pre):]: v for k, v in P.items() if k.startswith(pre)})
    mod = mod.to(dev).train()
    x, w = rnd(B, T, d, seed=7), rnd(B, T, d, seed=8)
    L = torch.tensor([12, 9])
    pe = O.relpos_table(T, P["encoder.rel_pe.div_term"])
    torch.manual_seed(5)
    s_att, s_out = ops.new_seeds(2)
    torch.manual_seed(5)
    xg = x.to(dev).requires_grad_(True)
    y = mod.fused(xg, pe.to(dev), L.to(dev), residual=xg)
    (y * w.to(dev)).sum().backward()
    ma = mask_of((B, H, T, T), P_DROP, s_att, dev)
    mo = mask_of((B * T, d), P_DROP, s_out, dev).view(B, T, d)
    Pd = {k[len(pre):]: v.double().requires_grad_(True) for k, v in P.items() if k.startswith(pre)}
    xd = x.double().requires_grad_(True)
    a = "attention."
    xn = O.layer_norm(xd, Pd["layer_norm.weight"], Pd["layer_norm.bias"])
    q = (xn @ Pd[a + "query_proj.weight"].t() + Pd[a + "query_proj.bias"]).view(B, T, H, dh)
    k = (xn @ Pd[a + "key_proj.weight"].t() + Pd[a + "key_proj.bias"]).view(B, T, H, dh)
    v = (xn @ Pd[a + "value_proj.weight"].t() + Pd[a + "value_proj.bias"]).view(B, T, H, dh)
    pp = (pe.double() @ Pd[a + "pos_proj.weight"].t() + Pd[a + "pos_proj.bias"]).view(2 * T - 1, H, dh)
    content = torch.einsum("bihc,bkhc->bhik", q + Pd[a + "content_bias"], k)
    full = torch.einsum("bihc,jhc->bhij", q + Pd[a + "position_bias"], pp)
    idx = ((T - 1) - (torch.arange(T)[:, None] - torch.arange(T)[None, :])).expand(B, H, T, T)
    s = (content + full.gather(-1, idx)) / math.sqrt(dh)
    s = s.masked_fill((torch.arange(T)[None, :] >= L[:, None])[:, None, None, :], float("-inf"))
    att = torch.softmax(s, -1) * ma
    ctxr = torch.einsum("bhik,bkhc->bihc", att, v).reshape(B, T, d)
    ref = (ctxr @ Pd[a + "out_proj.weight"].t() + Pd[a + "out_proj.bias"]) * mo + xd
    (ref * w.double()).sum().backward()
    assert rel_l2(y, ref) < 2e-5
    assert rel_l2(xg.grad, xd.grad) < 5e-5
    for n, p_ in mod.named_parameters():
        gr = Pd[n].grad
        if float(gr.norm()) < 1e-6:
            assert float(p_.grad.abs().max()) < 1e-4, n
        else:
            assert rel_l2(p_.grad, gr) < 5e-5, n


def test_full_model_trains_with_reference_default_dropout(dev):
    """train.py's default dropout_rate=0.1 (train.py:126,330): one optimisation step runs and changes the loss."""
    from model.conformer import Conformer
    torch.manual_seed(0)
    m = Conformer(17, 80, 2, 32, 4, 31, 24, 1, 0.1).to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 80, 103, generator=g).to(dev)
    L = torch.full((4,), 103, device=dev)
    tg = torch.randint(1, 17, (4, 5), generator=g).to(dev)
    tl = torch.full((4,), 5, device=dev)
    crit = torch.nn.CTCLoss(blank=0, zero_infinity=True)
    losses = []
    for _ in range(3):
        logits, ol = m(x, L)
        loss = crit(logits.float().log_softmax(-1).transpose(0, 1), tg, ol, tl)
        opt.zero_grad()
        loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters() if p.requires_grad)
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0]
