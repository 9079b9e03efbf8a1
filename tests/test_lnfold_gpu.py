"""Folded LayerNorm (fp32 inference): the producers' statistics partials, the consumer GEMM against a float64 LayerNorm + Linear,
and the whole block / encoder against the un-folded kernels and the float64 oracle.

The fold changes WHERE the LayerNorm arithmetic happens (rstd * (x.W'^T - mean * colsum) in the GEMM epilogue), not what is
computed (ffn.py:16-17, attention.py:15 + 78-80, convolution.py:22-25, block.py:27): tolerances are the fp32 ones of the
un-folded kernels (2e-5 per op, 1e-4 per encoder).
"""
import math

import pytest
import torch

from oracle import conformer_oracle as O
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


def ref_partials(y, width):
    """(sum, M2 about the group's own mean) of every `width` consecutive values of each row, float64."""
    g = y.double().reshape(y.shape[0], -1, width)
    s = g.sum(-1)
    m2 = ((g - g.mean(-1, keepdim=True)) ** 2).sum(-1)
    return torch.stack([s, m2], dim=-1)


def merged(stats, d):
    """(mean, var) of each row from its partials (the merge the consumer kernel performs), float64."""
    st = stats.double().cpu()
    n = d // st.shape[1]
    mean = st[..., 0].sum(-1) / d
    m2 = st[..., 1].sum(-1) + (n * (st[..., 0] / n - mean[:, None]) ** 2).sum(-1)
    return mean, m2 / d


@pytest.mark.parametrize("M,N,K", [(1, 32, 16), (100, 64, 144), (257, 512, 512), (7968, 512, 2048), (130, 256, 20), (333, 128, 64)])
def test_producer_statistics(ops, M, N, K):
    """Residual / plain GEMM with emit_stats: same C as without, partials == float64 statistics of the stored rows."""
    a, w, b, r = rnd(M, K, seed=4), rnd(N, K, seed=5) / math.sqrt(K), rnd(N, seed=6), rnd(M, N, seed=7) * 2 + 0.7
    c0 = ops.linear_residual(G(a), G(w), G(b), G(r), 0.5)
    c1, st = ops.linear_residual(G(a), G(w), G(b), G(r), 0.5, emit_stats=True)
    assert torch.equal(c0, c1)
    assert st.shape == (M, N // 32, 2)
    ref = ref_partials(c1.cpu(), 32)
    assert rel_l2(st[..., 0], ref[..., 0]) < 1e-6 and rel_l2(st[..., 1], ref[..., 1]) < 1e-5
    p0 = ops.linear(G(a), G(w), G(b))
    p1, st = ops.linear(G(a), G(w), G(b), emit_stats=True)
    assert torch.equal(p0, p1)
    ref = ref_partials(p1.cpu(), 32)
    assert rel_l2(st[..., 0], ref[..., 0]) < 1e-6 and rel_l2(st[..., 1], ref[..., 1]) < 1e-5
    mean, var = merged(st, N)
    assert rel_l2(var, p1.double().cpu().var(-1, unbiased=False)) < 1e-5
    assert float((mean - p1.double().cpu().mean(-1)).abs().max()) < 1e-5


@pytest.mark.parametrize("rows,d", [(1, 32), (1000, 160), (7968, 512)])
def test_layernorm_emits_statistics_of_its_output(ops, rows, d):
    x, w, b = rnd(rows, d, seed=1) * 3 + 1, rnd(d, seed=2), rnd(d, seed=3)
    y0 = ops.layernorm(G(x), G(w), G(b))
    y1, st = ops.layernorm(G(x), G(w), G(b), emit_stats=True)
    assert torch.equal(y0, y1) and st.shape == (rows, 1, 2)
    ref = ref_partials(y1.cpu(), d)
    assert rel_l2(st[..., 0], ref[..., 0]) < 1e-5 + 1e-6 and rel_l2(st[..., 1], ref[..., 1]) < 1e-5


@pytest.mark.parametrize("M,N,d,mean_shift", [(1, 16, 32, 0.0), (100, 576, 64, 0.3), (65, 144, 128, -0.4), (257, 2048, 512, 0.0), (7968, 2048, 512, 0.5),
                                              (7968, 1536, 512, -1.0), (300, 128, 256, 2.0)])
def test_consumer_vs_float64_layernorm_linear(ops, M, N, d, mean_shift):
    """linear_lnfold(x, stats(x)) == act(LN(x).W^T + b) in float64, for statistics written by a GEMM (d/32 partials) and by the
    LayerNorm kernel (1 partial); rows with a mean of the order of their spread included (the a - mean * colsum cancellation)."""
    x = rnd(M, d, seed=11) * 1.7 + mean_shift
    gam, bet = rnd(d, seed=12) * 0.3 + 1.0, rnd(d, seed=13) * 0.2
    w, b = rnd(N, d, seed=14) / math.sqrt(d), rnd(N, seed=15)
    ref = O.layer_norm(x.double(), gam.double(), bet.double()) @ w.double().t() + b.double()
    wf, bf, cs = ops.fold_layernorm(G(w), G(b), G(gam), G(bet))
    # statistics from a producing GEMM: x = 1.0 * (0 . W0^T + 0) + x  (identity through the residual epilogue)
    z = torch.zeros(M, 16)
    xg, st16 = ops.linear_residual(G(z), G(torch.zeros(d, 16)), G(torch.zeros(d)), G(x), 1.0, emit_stats=True)
    assert torch.equal(xg.cpu(), x)
    st1 = torch.from_numpy(ref_partials(x, d).float().numpy()).cuda().contiguous()          # one partial per row
    for st in (st16, st1):
        assert rel_l2(ops.linear_lnfold(xg, st, wf, bf, cs, 1e-5), ref) < TOL
        assert rel_l2(ops.linear_lnfold(xg, st, wf, bf, cs, 1e-5, act="swish"), O.swish(ref)) < TOL
        if N % 2 == 0:
            n = N // 2
            assert rel_l2(ops.linear_lnfold(xg, st, wf, bf, cs, 1e-5, glu=True), ref[:, :n] * torch.sigmoid(ref[:, n:])) < TOL


def test_fold_refused_where_it_does_not_apply(ops):
    from conformer_amd._lib import ConformerHipError
    assert not ops.ln_fold_ok(144) and not ops.ln_fold_ok(1024) and not ops.ln_fold_ok(96) and ops.ln_fold_ok(512) and ops.ln_fold_ok(32)
    a, w, b = G(rnd(8, 16, seed=1)), G(rnd(144, 16, seed=2)), G(rnd(144, seed=3))
    with pytest.raises(ConformerHipError):
        ops.linear(a, w, b, emit_stats=True)                      # N % 32 != 0
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert not ops.ln_fold_ok(512)                            # the 16-bit path keeps its LayerNorm kernels
    prev = ops.set_ln_fold(False)
    try:
        assert not ops.ln_fold_ok(512)
    finally:
        ops.set_ln_fold(prev)


def _encoder(n_blocks, d, heads, seed):
    from model.modules.encoder import Encoder
    P = O.make_params(vocab=11, n_mel=80, n_blocks=n_blocks, d=d, n_heads=heads, ksize=31, lstm_hidden=16, seed=seed)
    enc = Encoder(80, n_blocks, d, heads, 31, 0.0)
    enc.load_state_dict({k[len("encoder."):]: v for k, v in P.items() if k.startswith("encoder.")}, strict=True)
    return enc.cuda().eval(), P


@pytest.mark.parametrize("n_blocks,d,heads,B,T,lens", [(2, 32, 4, 3, 103, [103, 80, 31]), (3, 64, 4, 2, 200, [200, 160]),
                                                        (2, 512, 8, 2, 400, [400, 333])])
def test_encoder_folded_vs_unfolded_and_oracle(ops, n_blocks, d, heads, B, T, lens):
    """Whole encoder: folded path (default) against the un-folded kernels (launch-for-launch rounds 1-2 path) and the float64
    oracle (Encoder.forward, encoder.py:18-37)."""
    enc, P = _encoder(n_blocks, d, heads, seed=41)
    x = rnd(B, 80, T, seed=42)
    L = torch.tensor(lens)
    with torch.no_grad():
        assert ops.ln_fold_ok(d)
        y_fold, _ = enc(x.cuda(), L.cuda())
        prev = ops.set_ln_fold(False)
        try:
            y_plain, _ = enc(x.cuda(), L.cuda())
        finally:
            ops.set_ln_fold(prev)
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    ref, _ = O.encoder_forward(x.double(), L, P64, n_blocks, heads)
    e_fold, e_plain = rel_l2(y_fold, ref), rel_l2(y_plain, ref)
    assert e_plain < 1e-4 and e_fold < 1e-4, (e_fold, e_plain)
    assert e_fold < 3 * e_plain + 2e-6, (e_fold, e_plain)            # the fold costs no accuracy worth naming
    assert rel_l2(y_fold, y_plain) < 2e-5


def test_fold_follows_weight_updates(ops):
    """The folded parameters are cached per weight version: an in-place update of gamma / W must show in the next forward."""
    enc, _ = _encoder(1, 32, 4, seed=43)
    x = rnd(2, 80, 120, seed=44).cuda()
    with torch.no_grad():
        y0, _ = enc(x, None)
        enc.layers[0].ffn_1.layer_norm.weight.mul_(1.5)
        enc.layers[0].conv.layer_norm.bias.add_(0.25)
        enc.layers[0].attention.attention.query_proj.weight.mul_(0.5)
        y1, _ = enc(x, None)
        prev = ops.set_ln_fold(False)
        try:
            y2, _ = enc(x, None)
        finally:
            ops.set_ln_fold(prev)
    assert rel_l2(y1, y0) > 1e-3
    assert rel_l2(y1, y2) < 2e-5
