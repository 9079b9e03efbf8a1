"""GPU parity of the backward kernels: (a) the general backward GEMM against fp64 matmuls for every operand layout,
(b) each autograd.Function against gradients produced by the reference itself (tests/golden, autograd on CPU).

Tolerance: fp32 kernels with atomics-based reductions: 5e-5 rel-L2 per gradient tensor (north_star budget 1e-3)."""
import pytest
import torch

from oracle import conformer_oracle as O
from tests.util import cfg_params, load_golden, rel_l2

pytestmark = pytest.mark.gpu
TOL = 5e-5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("I,J,Kc", [(64, 64, 64), (100, 36, 52), (249, 64, 249), (512, 2048, 7968), (7968, 512, 2048),
                                    (37, 128, 1000), (130, 12, 8)])
def test_gemm_bwd_layouts(dev, I, J, Kc):
    from conformer_amd import ops
    for a_col in (False, True):
        for b_col in (False, True):
            lda = (I if a_col else Kc) + 3 & ~3
            ldb = (J if b_col else Kc) + 3 & ~3
            A = rnd(Kc if a_col else I, lda, seed=1)
            Bm = rnd(Kc if b_col else J, ldb, seed=2)
            Aop = (A[:, :I].t() if a_col else A[:, :Kc]).double()          # (I, Kc)
            Bop = (Bm[:, :J].t() if b_col else Bm[:, :Kc]).double()        # (J, Kc)
            ref = 0.7 * Aop @ Bop.t()
            ldc = J + 3 & ~3
            out = torch.zeros(I, ldc, device=dev)
            ops.gemm_bwd(A.to(dev), a_col, Bm.to(dev), b_col, I, J, Kc, alpha=0.7, out=out, lda=lda, ldb=ldb, ldc=ldc)
            assert rel_l2(out[:, :J], ref) < 2e-5, (a_col, b_col)
            out2 = torch.zeros(I, ldc, device=dev)
            ops.gemm_bwd(A.to(dev), a_col, Bm.to(dev), b_col, I, J, Kc, alpha=0.7, out=out2, lda=lda, ldb=ldb, ldc=ldc,
                         allow_split=True)
            assert rel_l2(out2[:, :J], ref) < 2e-5, ("split", a_col, b_col)
            ops.gemm_bwd(A.to(dev), a_col, Bm.to(dev), b_col, I, J, Kc, alpha=0.7, out=out, lda=lda, ldb=ldb, ldc=ldc,
                         accumulate=True)
            assert rel_l2(out[:, :J], 2 * ref) < 2e-5, ("accumulate", a_col, b_col)


def test_gemm_bwd_batched_and_dswish(dev):
    from conformer_amd import ops
    nb0, nb1, I, J, Kc = 3, 4, 50, 36, 44
    A = rnd(nb0, nb1, I, Kc, seed=3); Bm = rnd(nb0, nb1, J, Kc, seed=4)
    out = torch.empty(nb0, nb1, I, J, device=dev)
    ops.gemm_bwd(A.to(dev), False, Bm.to(dev), False, I, J, Kc, out=out, lda=Kc, ldb=Kc, ldc=J, nbatch=nb0 * nb1, nb1=nb1,
                 sa=(nb1 * I * Kc, I * Kc), sb=(nb1 * J * Kc, J * Kc), sc=(nb1 * I * J, I * J))
    assert rel_l2(out, A.double() @ Bm.double().transpose(-1, -2)) < 2e-5
    M, N, K = 300, 64, 128
    dy, w, z = rnd(M, N, seed=5), rnd(N, K, seed=6), rnd(M, K, seed=7)
    dx = ops.gemm_bwd(dy.to(dev), False, w.to(dev), True, M, K, N, alpha=0.5, Z=z.to(dev))
    zz = z.double(); sg = torch.sigmoid(zz)
    assert rel_l2(dx, 0.5 * (dy.double() @ w.double()) * (sg * (1 + zz * (1 - sg)))) < 2e-5


@pytest.mark.parametrize("rows,d", [(7, 32), (1000, 144), (7968, 512)])
def test_layernorm_bwd(dev, rows, d):
    from conformer_amd import ops
    x, w, b, dy, dres = rnd(rows, d, seed=8) * 2 + 0.5, rnd(d, seed=9), rnd(d, seed=10), rnd(rows, d, seed=11), rnd(rows, d, seed=12)
    xd = x.double().requires_grad_(True); wd = w.double().requires_grad_(True); bd = b.double().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xd, (d,), wd, bd, 1e-5)
    y.backward(dy.double())
    yg, mean, rstd = ops.layernorm_train(x.to(dev), w.to(dev), b.to(dev))
    assert rel_l2(yg, y) < 2e-5
    dx, dw, db = ops.layernorm_bwd(x.to(dev), w.to(dev), dy.to(dev), mean, rstd, dres=dres.to(dev))
    assert rel_l2(dx, xd.grad + dres.double()) < TOL
    assert rel_l2(dw, wd.grad) < TOL and rel_l2(db, bd.grad) < TOL


def _load_sub(mod, P, prefix, dev):
    mod.load_state_dict({k[len(prefix):]: v for k, v in P.items() if k.startswith(prefix)})
    return mod.to(dev).eval()            # eval: BatchNorm running statistics, as in the goldens


def _check_grads(mod, y, x, w, g, key, names_prefix=""):
    (y * w).sum().backward()
    assert rel_l2(x.grad, g[key + "_dx"]) < TOL, key + "_dx"
    for n, p in mod.named_parameters():
        gk = f"{key}_d.{n}"
        if gk not in g:
            continue
        ref = g[gk]
        if float(ref.norm()) < 1e-4:                     # mathematically-zero grads (key/pos projection biases)
            assert float(p.grad.abs().max()) < 1e-4, gk
        else:
            assert rel_l2(p.grad, ref) < TOL, gk


@pytest.mark.parametrize("case", ["modules_d32_t7", "modules_d32_t48", "modules_d32_t1", "modules_d144_t49",
                                  "modules_d64_t70"])
def test_module_grads_vs_reference_golden(dev, case):
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

    def fresh_x():
        return g["x"].to(dev).clone().requires_grad_(True)

    ffn = _load_sub(FeedForwardModule(d), P, blk + "ffn_1.", dev)
    x = fresh_x(); y = ffn(x)
    assert rel_l2(y, g["ffn_y"]) < 2e-5
    _check_grads(ffn, y, x, w, g, "ffn")

    att = _load_sub(MultiHeadSelfAttentionModule(d, H), P, blk + "attention.", dev)
    x = fresh_x(); y = att(x, pe, mask)
    assert rel_l2(y, g["mhsa_y"]) < 2e-5
    _check_grads(att, y, x, w, g, "mhsa")

    conv = _load_sub(ConvolutionModule(d, K), P, blk + "conv.", dev)
    x = fresh_x(); y = conv(x)
    assert rel_l2(y, g["conv_eval_y"]) < 2e-5
    _check_grads(conv, y, x, w, g, "conv_eval")

    # train-mode BatchNorm: batch statistics, running-stat update, coupled backward (convolution.py:27 under .train())
    convt = _load_sub(ConvolutionModule(d, K), P, blk + "conv.", dev).train()
    x = fresh_x(); y = convt(x)
    assert rel_l2(y, g["conv_train_y"]) < 2e-5
    assert rel_l2(convt.batch_norm.running_mean, g["conv_train_running_mean"]) < 2e-5
    assert rel_l2(convt.batch_norm.running_var, g["conv_train_running_var"]) < 2e-5
    assert int(convt.batch_norm.num_batches_tracked) == 4
    if meta["T"] * g["x"].shape[0] > 2:          # B*T = 2 (d32_t1): rstd ~ 1/sqrt(eps) amplifies fp32 noise 300x
        _check_grads(convt, y, x, w, g, "conv_train")

    block = _load_sub(ConformerBlock(d, H, K), P, blk, dev)
    x = fresh_x(); y = block(x, pe, mask)
    assert rel_l2(y, g["block_y"]) < 2e-5
    _check_grads(block, y, x, w, g, "block")


def test_encoder_grads_vs_reference_golden(dev):
    """Tiny 2-block encoder: every parameter gradient of sum(enc*w), stem included."""
    from model.modules.encoder import Encoder
    meta, g = load_golden("model_tiny")
    P = cfg_params(meta)
    enc = Encoder(80, meta["n_blocks"], meta["d"], meta["n_heads"], meta["ksize"], 0.0)
    enc.load_state_dict({k[len("encoder."):]: v for k, v in P.items() if k.startswith("encoder.")}, strict=True)
    enc = enc.to(dev).eval()
    y, _ = enc(g["x"].to(dev), g["lengths"].to(dev))
    assert rel_l2(y, g["enc"]) < 1e-4
    (y * g["w"].to(dev)).sum().backward()
    checked = 0
    for n, p in enc.named_parameters():
        gk = "grad." + n
        if gk not in g or not p.requires_grad:
            continue
        ref = g[gk]
        if float(ref.norm()) < 1e-4:
            assert float(p.grad.abs().max()) < 1e-4, n
        else:
            assert rel_l2(p.grad, ref) < 2e-4, n
        checked += 1
    assert checked > 60


@pytest.mark.parametrize("d,H,K,B,T,train_bn", [(256, 4, 15, 3, 61, False), (96, 8, 7, 2, 33, True), (512, 8, 31, 2, 100, True),
                                                (64, 1, 3, 2, 5, False), (144, 4, 31, 1, 17, False)])
def test_block_forward_backward_other_geometries_vs_oracle_autograd(dev, d, H, K, B, T, train_bn):
    """Shapes the goldens do not cover (Conformer-M width, head sizes 12 / 64 / 36, kernel sizes 3 / 7 / 15, one head,
    B = 1, train-mode BatchNorm): the block's output, input gradient and every parameter gradient against torch autograd
    through the float64 oracle (itself pinned by the reference's goldens at the other shapes)."""
    from model.utils.block import ConformerBlock
    from model.utils.position import RelativePositionalEncoding
    P = O.make_params(vocab=8, n_mel=80, n_blocks=1, d=d, n_heads=H, ksize=K, lstm_hidden=8, seed=d + K, with_decoder=False)
    blk = "encoder.layers.0."
    m = ConformerBlock(d, H, K).to(dev)
    m.load_state_dict({k[len(blk):]: v for k, v in P.items() if k.startswith(blk)})
    m.train()
    if not train_bn:
        m.conv.batch_norm.eval()
    g = torch.Generator().manual_seed(T)
    x = torch.randn(B, T, d, generator=g)
    w = torch.randn(B, T, d, generator=g)
    L = torch.sort(torch.randint(1, T + 1, (B,), generator=g), descending=True).values
    L[0] = T
    Pd = {k: (v.double().requires_grad_(True) if v.is_floating_point() and k.startswith(blk) and "running" not in k
              and "div_term" not in k else (v.double() if v.is_floating_point() else v)) for k, v in P.items()}
    xr = x.double().requires_grad_(True)
    pe = O.relpos_table(T, Pd["encoder.rel_pe.div_term"])
    ref = O.conformer_block(xr, pe, L, Pd, blk, H, training=train_bn)
    (ref * w.double()).sum().backward()
    xd = x.to(dev).requires_grad_(True)
    rel = RelativePositionalEncoding(d).to(dev)
    y = m.fused(xd, rel.table(T), L.to(dev))
    (y * w.to(dev)).sum().backward()
    assert rel_l2(y, ref) < 2e-5
    assert rel_l2(xd.grad, xr.grad) < 1e-4
    for n, p in m.named_parameters():
        r = Pd[blk + n].grad
        if r is None or float(r.norm()) < 1e-6 * max(1.0, float(Pd[blk + n].detach().norm())):
            assert p.grad is None or float(p.grad.abs().max()) < 1e-3, n      # mathematically-zero gradients
        else:
            assert rel_l2(p.grad, r) < 2e-4, n


def test_bare_relative_attention_module_is_differentiable(dev):
    """RelativeMultiHeadAttention.forward(q, k, v, pos, mask) called directly in training (attention.py:74-92: projections,
    attention core, out_proj; the blocks use the fused MultiHeadSelfAttentionModule path instead): output and every gradient
    vs torch autograd through the float64 restatement of the same lines."""
    from model.utils.attention import RelativeMultiHeadAttention
    d, H, B, T = 64, 4, 3, 37
    g = torch.Generator().manual_seed(11)
    mod = RelativeMultiHeadAttention(d, H)
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.2 if p.dim() > 1 else 0.1))
    x = torch.randn(B, T, d, generator=g)
    pe = torch.randn(2 * T - 1, d, generator=g)
    L = torch.tensor([T, 20, 5])
    w = torch.randn(B, T, d, generator=g)
    # float64 restatement (attention.py:76-91)
    P = {n: p.detach().double().requires_grad_(True) for n, p in mod.named_parameters()}
    xr, per = x.double().requires_grad_(True), pe.double().requires_grad_(True)
    dh = d // H
    lin = lambda t, n: t @ P[n + ".weight"].t() + P[n + ".bias"]
    ctx = O.relpos_attention_core(lin(xr, "query_proj").view(B, T, H, dh), lin(xr, "key_proj").view(B, T, H, dh),
                                  lin(xr, "value_proj").view(B, T, H, dh), lin(per, "pos_proj").view(2 * T - 1, H, dh),
                                  P["content_bias"], P["position_bias"], L)
    ref = lin(ctx, "out_proj")
    (ref * w.double()).sum().backward()
    mod = mod.to(dev).train()
    xd, ped = x.to(dev).requires_grad_(True), pe.to(dev).requires_grad_(True)
    mask = (torch.arange(T)[None, :] >= L[:, None])[:, None, None, :].to(dev)
    y = mod(xd, xd, xd, ped, mask)
    (y * w.to(dev)).sum().backward()
    assert rel_l2(y, ref.detach()) < 2e-5
    assert rel_l2(xd.grad, xr.grad) < 5e-5 and rel_l2(ped.grad, per.grad) < 5e-5
    for n, p in mod.named_parameters():
        if float(P[n].grad.norm()) < 1e-9:                     # key / position-projection biases: zero by construction
            assert float(p.grad.abs().max()) < 1e-4, n
        else:
            assert rel_l2(p.grad, P[n].grad) < 5e-5, n
