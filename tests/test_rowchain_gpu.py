"""Row-local chains of a block (csrc/rowchain_f32.hip; fp32 inference): K1 = FFN1 + q|k|v projection, K2 = out_proj + residual +
pointwise_conv_1 + GLU, K3 = pointwise_conv_2 + residual + FFN2 + closing LayerNorm -- each against the float64 composition of
the reference's operators (block.py:17-29, ffn.py:15-23, attention.py:15,78-80,90, convolution.py:22-25,29), and the block built
from them against the one-kernel-per-GEMM path.  Tolerances: the fp32 ones of the kernels they replace (2e-5 per sub-layer)."""
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


def lin(n, k, seed):
    return rnd(n, k, seed=seed) / math.sqrt(k), 0.1 * rnd(n, seed=seed + 1)


def lnp(d, seed):
    return 1 + 0.3 * rnd(d, seed=seed), 0.2 * rnd(d, seed=seed + 1)


def ref_ffn(x, lw, lb, w1, b1, w2, b2, alpha):
    h = F.layer_norm(x, (x.shape[-1],), lw.double(), lb.double(), 1e-5) @ w1.double().T + b1.double()
    h = h * torch.sigmoid(h)
    return alpha * (h @ w2.double().T + b2.double()) + x


def ffn_params(ops, d, seed):
    lw, lb = lnp(d, seed)
    (w1, b1), (w2, b2) = lin(4 * d, d, seed + 2), lin(d, 4 * d, seed + 4)
    wf, bf, cs = ops.fold_layernorm(G(w1), G(b1), G(lw), G(lb))
    return (lw, lb, w1, b1, w2, b2), (ops.ffn_pack(wf, G(w2)), bf, cs)


CASES = [(1, 128), (70, 128), (333, 256), (257, 512), (7968, 512)]


@pytest.mark.parametrize("M,d", CASES)
def test_k1_ffn_then_qkv(ops, M, d):
    x = rnd(M, d, seed=1) * 1.5 + 0.4
    raw, ffn = ffn_params(ops, d, 10)
    law, lab = lnp(d, 20)
    wq, bq = lin(3 * d, d, 22)
    wqf, bqf, csq = ops.fold_layernorm(G(wq), G(bq), G(law), G(lab))
    y, qkv = ops.rowchain_ffn_qkv(G(x), G(partials(x, 32).float()), ffn, G(raw[5]), 0.5, 1e-5, ops.rowgemm_pack(wqf), bqf, csq, 1e-5)
    y_ref = ref_ffn(x.double(), *raw, 0.5)
    q_ref = F.layer_norm(y_ref, (d,), law.double(), lab.double(), 1e-5) @ wq.double().T + bq.double()
    assert rel_l2(y, y_ref) < TOL and rel_l2(qkv, q_ref) < TOL


@pytest.mark.parametrize("M,d", CASES)
def test_k2_out_proj_then_glu(ops, M, d):
    ctx, res = rnd(M, d, seed=2), rnd(M, d, seed=3) * 1.5 + 0.4
    (wo, bo), (wg, bg) = lin(d, d, 30), lin(2 * d, d, 32)
    lcw, lcb = lnp(d, 34)
    wgf, bgf, csg = ops.fold_layernorm(G(wg), G(bg), G(lcw), G(lcb))
    y2, g = ops.rowchain_out_glu(G(ctx), ops.rowgemm_pack(G(wo)), G(bo), G(res), ops.rowgemm_pack(wgf, glu=True), bgf, csg, 1e-5)
    y_ref = ctx.double() @ wo.double().T + bo.double() + res.double()
    h = F.layer_norm(y_ref, (d,), lcw.double(), lcb.double(), 1e-5) @ wg.double().T + bg.double()
    g_ref = h[:, :d] * torch.sigmoid(h[:, d:])
    assert rel_l2(y2, y_ref) < TOL and rel_l2(g, g_ref) < TOL


@pytest.mark.parametrize("M,d", CASES)
def test_k3_pw2_then_ffn_then_layernorm(ops, M, d):
    c, res = rnd(M, d, seed=4), rnd(M, d, seed=5) * 1.5 + 0.4
    w2, b2 = lin(d, d, 40)
    raw, ffn = ffn_params(ops, d, 42)
    g2, bt2 = lnp(d, 50)
    out, st = ops.rowchain_pw2_ffn_ln(G(c), ops.rowgemm_pack(G(w2)), G(b2), G(res), ffn, G(raw[5]), 0.5, 1e-5, (G(g2), G(bt2), 1e-5),
                                      want_stats=True)
    y3 = c.double() @ w2.double().T + b2.double() + res.double()
    ref = F.layer_norm(ref_ffn(y3, *raw, 0.5), (d,), g2.double(), bt2.double(), 1e-5)
    assert rel_l2(out, ref) < TOL and st.shape == (M, 1, 2)
    rp = partials(out.cpu(), d)
    assert rel_l2(st[..., 0], rp[..., 0]) < 1e-5 and rel_l2(st[..., 1], rp[..., 1]) < 1e-5
    out2, none = ops.rowchain_pw2_ffn_ln(G(c), ops.rowgemm_pack(G(w2)), G(b2), G(res), ffn, G(raw[5]), 0.5, 1e-5, (G(g2), G(bt2), 1e-5))
    assert none is None and torch.equal(out, out2)                   # fixed summation order: bit-identical reruns


def test_block_of_row_chains_matches_kernel_per_gemm_block(ops):
    """ConformerBlock.fused_chain at cfg-2 geometry (B=32, T'=249, d=512): 5 launches vs the folded-LayerNorm path it replaces."""
    from conformer_amd.model.utils.block import ConformerBlock
    torch.manual_seed(7)
    d, B, T, H = 512, 32, 249, 8
    blk = ConformerBlock(d, H, 31).cuda().eval()
    with torch.no_grad():
        for p in blk.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
        blk.conv.batch_norm.running_mean.normal_(0, 0.2); blk.conv.batch_norm.running_var.uniform_(0.5, 1.5)
        x = (torch.randn(B, T, d, device="cuda") + 0.2)
        st = G(partials(x.reshape(-1, d).cpu(), 32).float())
        table = ops.relpos_table(torch.exp(torch.arange(0, d, 2, device="cuda") * -(math.log(10000.0) / d))[None], T)
        L = torch.randint(100, T + 1, (B,), device="cuda"); L[0] = T
        prev = ops.set_rowchain(False)
        try:
            ref, st_ref = blk.fused_chain(x, table, L, x_stats=st, want_stats=True)
            ops.set_rowchain(True)                                        # (opt-in: default off, see ops.set_rowchain)
            assert ops.rowchain_ok(d, 4 * d, B * T)
            out, st_out = blk.fused_chain(x, table, L, x_stats=st, want_stats=True)
        finally:
            ops.set_rowchain(prev)
    assert rel_l2(out, ref) < 1e-5
    assert rel_l2(st_out[..., 0], st_ref[..., 0]) < 1e-4 and rel_l2(st_out[..., 1], st_ref[..., 1]) < 1e-4
