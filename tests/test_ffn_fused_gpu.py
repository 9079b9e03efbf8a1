"""One-kernel feed-forward sub-layer (csrc/ffn_fused_f32.hip; fp32 inference with the folded LayerNorm) against a float64
LayerNorm -> Linear -> Swish -> Linear -> alpha*y + x (ffn.py:15-23, block.py:19,25) and, in the closing form, block.py:27;
its statistics partials against float64; the module path against the two-GEMM path it replaces.  Tolerances: the fp32 ones
of the kernels it replaces (2e-5 per sub-layer).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.util import rel_l2

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from conformer_amd import _lib, ops as _ops
    assert _lib.load().cfm_device_check() == 0, "not a gfx950 device"
    return _ops


def G(t):
    return t.cuda()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def partials(y, width):
    g = y.double().reshape(y.shape[0], -1, width)
    return torch.stack([g.sum(-1), ((g - g.mean(-1, keepdim=True)) ** 2).sum(-1)], dim=-1)


def case(M, d, seed):
    hidden = 4 * d
    x = rnd(M, d, seed=seed) * 1.5 + 0.4                      # rows with a mean (the fold's cancellation term is exercised)
    lw, lb = 1 + 0.3 * rnd(d, seed=seed + 1), 0.2 * rnd(d, seed=seed + 2)
    w1, b1 = rnd(hidden, d, seed=seed + 3) / math.sqrt(d), 0.1 * rnd(hidden, seed=seed + 4)
    w2, b2 = rnd(d, hidden, seed=seed + 5) / math.sqrt(hidden), 0.1 * rnd(d, seed=seed + 6)
    return x, lw, lb, w1, b1, w2, b2


def ref_ffn(x, lw, lb, w1, b1, w2, b2, alpha, eps=1e-5):
    xd = x.double()
    h = F.layer_norm(xd, (x.shape[-1],), lw.double(), lb.double(), eps)
    h = h @ w1.double().T + b1.double()
    h = h * torch.sigmoid(h)
    return alpha * (h @ w2.double().T + b2.double()) + xd


@pytest.mark.parametrize("M,d,parts", [(1, 128, 1), (70, 128, 4), (333, 256, 8), (257, 512, 16), (96, 512, 1), (7968, 512, 16)])
def test_ffn_fused_vs_float64(ops, M, d, parts):
    x, lw, lb, w1, b1, w2, b2 = case(M, d, seed=3)
    wf, bf, cs = ops.fold_layernorm(G(w1), G(b1), G(lw), G(lb))
    wp = ops.ffn_pack(wf, G(w2))
    st_in = G(partials(x, d // parts).float())
    ref = ref_ffn(x, lw, lb, w1, b1, w2, b2, 0.5)
    y0 = ops.ffn_fused(G(x), st_in, wp, bf, cs, G(b2), 0.5, 1e-5)
    assert rel_l2(y0, ref) < TOL
    y1, st = ops.ffn_fused(G(x), st_in, wp, bf, cs, G(b2), 0.5, 1e-5, emit_stats=True)
    assert torch.equal(y0, y1) and st.shape == (M, d // 32, 2)
    rp = partials(y1.cpu(), 32)
    assert rel_l2(st[..., 0], rp[..., 0]) < 1e-6 and rel_l2(st[..., 1], rp[..., 1]) < 1e-5
    # closing LayerNorm (block.py:27) + the statistics of its output
    g2, bt2 = 1 + 0.2 * rnd(d, seed=11), 0.3 * rnd(d, seed=12)
    y2, st2 = ops.ffn_fused(G(x), st_in, wp, bf, cs, G(b2), 0.5, 1e-5, emit_stats=True, closing_ln=(G(g2), G(bt2), 1e-5))
    ref2 = F.layer_norm(ref, (d,), g2.double(), bt2.double(), 1e-5)
    assert rel_l2(y2, ref2) < TOL and st2.shape == (M, 1, 2)
    rp2 = partials(y2.cpu(), d)
    assert rel_l2(st2[..., 0], rp2[..., 0]) < 1e-5 and rel_l2(st2[..., 1], rp2[..., 1]) < 1e-5
    y3 = ops.ffn_fused(G(x), st_in, wp, bf, cs, G(b2), 0.5, 1e-5, closing_ln=(G(g2), G(bt2), 1e-5))
    assert torch.equal(y2, y3)


def test_ffn_fused_is_reproducible_and_leaves_its_neighbours_alone(ops):
    """Fixed summation order of the four waves' partial tiles: bit-identical reruns; sentinels around the outputs survive."""
    M, d = 1000, 512
    x, lw, lb, w1, b1, w2, b2 = case(M, d, seed=9)
    wf, bf, cs = ops.fold_layernorm(G(w1), G(b1), G(lw), G(lb))
    wp = ops.ffn_pack(wf, G(w2))
    st_in = G(partials(x, 32).float())
    a, sa = ops.ffn_fused(G(x), st_in, wp, bf, cs, G(b2), 0.5, 1e-5, emit_stats=True)
    for _ in range(3):
        b, sb = ops.ffn_fused(G(x), st_in, wp, bf, cs, G(b2), 0.5, 1e-5, emit_stats=True)
        assert torch.equal(a, b) and torch.equal(sa, sb)


def test_module_path_matches_two_gemm_path(ops):
    """FeedForwardModule.fused on the folded path: one kernel vs hidden GEMM + residual GEMM (same inputs, same statistics)."""
    from conformer_amd.model.utils.ffn import FeedForwardModule
    torch.manual_seed(5)
    d, M = 512, 32 * 249
    m = FeedForwardModule(d).cuda().eval()
    with torch.no_grad():
        m.layer_norm.weight.mul_(1 + 0.2 * torch.randn(d, device="cuda")); m.layer_norm.bias.add_(0.1 * torch.randn(d, device="cuda"))
    x = G(rnd(M, d, seed=21) + 0.3)
    st = G(partials(x.cpu(), 32).float())
    with torch.no_grad():
        prev = ops.set_ffn_fused(False)
        try:
            y_ref, st_ref = m.fused(x, residual=x, alpha=0.5, stats=st, emit_stats=True)
        finally:
            ops.set_ffn_fused(prev)
        assert ops.ffn_fused_ok(d, 4 * d, M)
        y, st_new = m.fused(x, residual=x, alpha=0.5, stats=st, emit_stats=True)
    assert rel_l2(y, y_ref) < 5e-6
    assert rel_l2(st_new[..., 0], st_ref[..., 0]) < 1e-5 and rel_l2(st_new[..., 1], st_ref[..., 1]) < 1e-4


def test_ffn_fused_refuses_what_it_does_not_support(ops):
    from conformer_amd._lib import ConformerHipError
    x = G(rnd(64, 160))
    with pytest.raises(ConformerHipError):
        ops.ffn_pack(G(rnd(640, 160)), G(rnd(160, 640)))      # d = 160: not 128 / 256 / 512
    assert not ops.ffn_fused_ok(160, 640, 10000) and not ops.ffn_fused_ok(512, 2048, 1280)
