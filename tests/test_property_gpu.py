"""Property-based shape / length sweeps ("hypothesis: T' odd/even/1, ragged lengths, chunking equivalences").

Hand-picked shapes live in test_ops_gpu.py; here hypothesis draws them (derandomised: the same examples every run, one
process, a few dozen small launches per test) for the kernels whose edge handling depends on the geometry: LayerNorm rows,
the GEMM epilogues (ragged M / N / K, every block tile), the depthwise-conv window at both sequence ends, the rel-pos attention
(T = 1, odd / even, lengths 0..T, every head size) and the streaming encoder's chunking equivalence.  Oracle: float64 restatement
(oracle/conformer_oracle.py, pinned to the reference's goldens by tests/test_oracle_golden.py).
"""
import math

import pytest
import torch

hypothesis = pytest.importorskip("hypothesis")
from hypothesis import given, settings, strategies as st, HealthCheck  # noqa: E402

from oracle import conformer_oracle as O  # noqa: E402
from tests.util import rel_l2  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 2e-5
SET = dict(deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from conformer_amd import ops as _ops
    return _ops


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


@settings(max_examples=30, **SET)
@given(rows=st.integers(1, 300), d4=st.integers(1, 300), seed=st.integers(0, 10 ** 6))
def test_layernorm_any_shape(ops, rows, d4, seed):
    d = 4 * d4
    x, w, b = rnd(rows, d, seed=seed) * 2 + 0.5, rnd(d, seed=seed + 1), rnd(d, seed=seed + 2)
    y = ops.layernorm(x.cuda(), w.cuda(), b.cuda())
    assert rel_l2(y, O.layer_norm(x.double(), w.double(), b.double())) < TOL


@settings(max_examples=30, **SET)
@given(M=st.integers(1, 400), N=st.integers(1, 300), K4=st.integers(1, 80), seed=st.integers(0, 10 ** 6))
def test_gemm_epilogues_any_shape(ops, M, N, K4, seed):
    K = 4 * K4
    a, w, b, r = rnd(M, K, seed=seed), rnd(N, K, seed=seed + 1) / math.sqrt(K), rnd(N, seed=seed + 2), rnd(M, N, seed=seed + 3)
    ref = a.double() @ w.double().t() + b.double()
    A, W, Bv = a.cuda(), w.cuda(), b.cuda()
    assert rel_l2(ops.linear(A, W, Bv), ref) < TOL
    assert rel_l2(ops.linear(A, W, Bv, act="swish"), O.swish(ref)) < TOL
    assert rel_l2(ops.linear_residual(A, W, Bv, r.cuda(), 0.5), 0.5 * ref + r.double()) < TOL
    if N % 2 == 0:
        assert rel_l2(ops.linear_glu(A, W, Bv), ref[:, :N // 2] * torch.sigmoid(ref[:, N // 2:])) < TOL
    with torch.autocast("cuda", dtype=torch.bfloat16):
        r16 = a.to(torch.bfloat16).double() @ w.to(torch.bfloat16).double().t() + b.double()
        assert rel_l2(ops.linear(A, W, Bv), r16) < TOL


@settings(max_examples=25, **SET)
@given(B=st.integers(1, 3), T=st.integers(1, 70), C4=st.integers(1, 40), K=st.sampled_from([3, 7, 15, 31]), seed=st.integers(0, 10 ** 6))
def test_dwconv_bn_swish_any_length(ops, B, T, C4, K, seed):
    C = 4 * C4
    g, wd, bd = rnd(B, T, C, seed=seed), rnd(C, 1, K, seed=seed + 1) / 3, rnd(C, seed=seed + 2) * 0.1
    bw, bb, bm, bv = rnd(C, seed=seed + 3) * 0.2 + 1, rnd(C, seed=seed + 4) * 0.1, rnd(C, seed=seed + 5) * 0.1, rnd(C, seed=seed + 6).abs() + 0.5
    y = ops.dwconv_bn_swish(*(t.cuda() for t in (g, wd, bd, bw, bb, bm, bv)))
    c = torch.nn.functional.conv1d(g.double().transpose(1, 2), wd.double(), bd.double(), padding=K // 2, groups=C)
    bn = (c - bm.double()[:, None]) / torch.sqrt(bv.double()[:, None] + 1e-5) * bw.double()[:, None] + bb.double()[:, None]
    assert rel_l2(y, O.swish(bn).transpose(1, 2)) < TOL


@settings(max_examples=30, **SET)
@given(B=st.integers(1, 3), T=st.integers(1, 300), H=st.sampled_from([1, 2, 4]), dh4=st.integers(1, 16), data=st.data())
def test_relpos_attention_any_geometry(ops, B, T, H, dh4, data):
    """T = 1, odd, even, beyond 128 / 256 query rows (both workgroup shapes); key lengths from 0 (uniform weights) to T."""
    dh = 4 * dh4
    d = H * dh
    lengths = data.draw(st.one_of(st.none(), st.lists(st.integers(0, T), min_size=B, max_size=B)))
    seed = data.draw(st.integers(0, 10 ** 6))
    qkv, pos = rnd(B, T, 3 * d, seed=seed) * 0.5, rnd(2 * T - 1, d, seed=seed + 1) * 0.5
    u, v = rnd(H, dh, seed=seed + 2) * 0.3, rnd(H, dh, seed=seed + 3) * 0.3
    L = None if lengths is None else torch.tensor(lengths)
    ctx = ops.relpos_attention(qkv.cuda(), pos.cuda(), u.cuda(), v.cuda(), None if L is None else L.cuda(), H)
    q, k, vv = (t.reshape(B, T, H, dh).double() for t in qkv.split(d, dim=-1))
    ref = O.relpos_attention_core(q, k, vv, pos.double().view(2 * T - 1, H, dh), u.double(), v.double(), L)
    assert torch.isfinite(ctx).all()
    assert rel_l2(ctx, ref.reshape(B, T, d)) < TOL


@settings(max_examples=6, **SET)
@given(chunks=st.lists(st.integers(1, 90), min_size=1, max_size=6), seed=st.integers(0, 10 ** 6))
def test_streaming_chunking_equivalence(ops, chunks, seed):
    """Any chunking of the mel stream == the masked whole-sequence restatement of the prefix rule (oracle.encoder_forward_chunked)."""
    from conformer_amd.streaming import StreamingEncoder, chunk_ends
    from model.modules.encoder import Encoder
    T = sum(chunks)
    if ((T - 1) // 2 - 1) // 2 < 1:
        return
    P = O.make_params(vocab=8, n_mel=80, n_blocks=2, d=32, n_heads=4, ksize=31, lstm_hidden=8, seed=3, with_decoder=False)
    enc = Encoder(80, 2, 32, 4, 31, 0.0)
    enc.load_state_dict({k[len("encoder."):]: v for k, v in P.items() if k.startswith("encoder.")}, strict=True)
    enc = enc.cuda().eval()
    x = rnd(2, 80, T, seed=seed)
    se = StreamingEncoder(enc, 2, T)
    outs, t0 = [], 0
    for c in chunks:
        outs.append(se.step(x[:, :, t0:t0 + c].cuda()))
        t0 += c
    y = torch.cat(outs, dim=1)
    ends = chunk_ends(T, chunks)
    Pd = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    ref = O.encoder_forward_chunked(x.double(), Pd, 2, 4, ends)
    assert y.shape == ref.shape
    assert rel_l2(y, ref) < 5e-5


@settings(max_examples=20, **SET)
@given(n=st.integers(1, 70), data=st.data())
def test_subsampled_lengths_floor_division(ops, n, data):
    vals = data.draw(st.lists(st.integers(-5, 30000), min_size=n, max_size=n))
    L = torch.tensor(vals, dtype=torch.int64)
    want = torch.div(torch.div(L - 1, 2, rounding_mode="floor") - 1, 2, rounding_mode="floor")
    assert torch.equal(ops.subsampled_lengths(L.cuda()).cpu(), want)


def _partials(y, parts):
    g = y.double().reshape(y.shape[0], parts, -1)
    return torch.stack([g.sum(-1), ((g - g.mean(-1, keepdim=True)) ** 2).sum(-1)], dim=-1).float()


@settings(max_examples=20, **SET)
@given(M=st.integers(1, 200), d=st.sampled_from([128, 256, 512]), hid128=st.integers(1, 6), lp=st.integers(0, 4),
       mean=st.floats(-1.5, 1.5), scale=st.floats(0.3, 3.0), seed=st.integers(0, 10 ** 6))
def test_ffn_fused_any_rows_hidden_statistics(ops, M, d, hid128, lp, mean, scale, seed):
    """One-kernel feed-forward: any row count (ragged last workgroup), hidden = 128 .. 768 (1 .. 6 slices per wave), 1 .. 16 incoming
    statistics partials, rows with a mean up to 1.5 standard deviations (the fold's cancellation term), all three finish modes."""
    import torch.nn.functional as F
    hidden, parts = 128 * hid128, 1 << lp
    x = rnd(M, d, seed=seed) * scale + mean * scale
    lw, lb = 1 + 0.3 * rnd(d, seed=seed + 1), 0.2 * rnd(d, seed=seed + 2)
    w1, b1 = rnd(hidden, d, seed=seed + 3) / math.sqrt(d), 0.1 * rnd(hidden, seed=seed + 4)
    w2, b2 = rnd(d, hidden, seed=seed + 5) / math.sqrt(hidden), 0.1 * rnd(d, seed=seed + 6)
    wf, bf, cs = ops.fold_layernorm(w1.cuda(), b1.cuda(), lw.cuda(), lb.cuda())
    wp = ops.ffn_pack(wf, w2.cuda())
    st_in = _partials(x, parts).cuda()
    h = F.layer_norm(x.double(), (d,), lw.double(), lb.double(), 1e-5) @ w1.double().T + b1.double()
    ref = 0.5 * ((h * torch.sigmoid(h)) @ w2.double().T + b2.double()) + x.double()
    y, st = ops.ffn_fused(x.cuda(), st_in, wp, bf, cs, b2.cuda(), 0.5, 1e-5, emit_stats=True)
    tol = TOL * max(1.0, math.sqrt(1 + mean * mean))                  # (the documented amplification of the folded LayerNorm)
    assert rel_l2(y, ref) < tol
    rp = _partials(y.cpu(), d // 32)
    assert rel_l2(st[..., 0], rp[..., 0]) < 1e-5 and rel_l2(st[..., 1], rp[..., 1]) < 1e-4
    g2, bt2 = 1 + 0.2 * rnd(d, seed=seed + 7), 0.3 * rnd(d, seed=seed + 8)
    y2 = ops.ffn_fused(x.cuda(), st_in, wp, bf, cs, b2.cuda(), 0.5, 1e-5, closing_ln=(g2.cuda(), bt2.cuda(), 1e-5))
    assert rel_l2(y2, F.layer_norm(ref, (d,), g2.double(), bt2.double(), 1e-5)) < 2 * tol


@settings(max_examples=12, **SET)
@given(M=st.integers(1, 150), d=st.sampled_from([128, 256, 512]), seed=st.integers(0, 10 ** 6))
def test_row_chain_out_glu_any_rows(ops, M, d, seed):
    """K2 of the row chains (PRE + POST without the feed-forward core: in-kernel row statistics, GLU store staging) at any row count."""
    import torch.nn.functional as F
    ctx, res = rnd(M, d, seed=seed), rnd(M, d, seed=seed + 1) * 1.5 + 0.4
    wo, bo = rnd(d, d, seed=seed + 2) / math.sqrt(d), 0.1 * rnd(d, seed=seed + 3)
    wg, bg = rnd(2 * d, d, seed=seed + 4) / math.sqrt(d), 0.1 * rnd(2 * d, seed=seed + 5)
    lw, lb = 1 + 0.3 * rnd(d, seed=seed + 6), 0.2 * rnd(d, seed=seed + 7)
    wgf, bgf, csg = ops.fold_layernorm(wg.cuda(), bg.cuda(), lw.cuda(), lb.cuda())
    y2, g = ops.rowchain_out_glu(ctx.cuda(), ops.rowgemm_pack(wo.cuda()), bo.cuda(), res.cuda(), ops.rowgemm_pack(wgf, glu=True), bgf, csg, 1e-5)
    y_ref = ctx.double() @ wo.double().T + bo.double() + res.double()
    hh = F.layer_norm(y_ref, (d,), lw.double(), lb.double(), 1e-5) @ wg.double().T + bg.double()
    assert rel_l2(y2, y_ref) < TOL and rel_l2(g, hh[:, :d] * torch.sigmoid(hh[:, d:])) < TOL
