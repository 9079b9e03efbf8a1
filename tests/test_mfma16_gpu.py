"""16-bit-MFMA path (torch.autocast with bfloat16, or float16 as the reference's --fp16 1): (a) forward and backward GEMMs
against an fp64 product of the ROUNDED operands (tight: only fp32 accumulation error remains), (b) per-block output
against the fp32 oracle within the north_star's 1e-2 rel for bf16 (each block is fed the oracle's fp32 input),
(c) the fp16 (reference AMP dtype) encoder gradients against the fp32 golden gradients.  The bf16 end-to-end and
per-tensor gradient checks against the reference's OWN autocast goldens live in tests/test_autocast_golden_gpu.py."""
import math

import pytest
import torch

from oracle import conformer_oracle as O
from tests.util import cfg_params, load_golden, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


DT = [torch.bfloat16, torch.float16]


@pytest.mark.parametrize("dt16", DT)
@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (100, 144, 144), (257, 576, 144), (7968, 512, 512), (300, 2048, 512),
                                   (300, 512, 2048), (129, 130, 20)])
def test_mfma16_gemm_epilogues(dev, M, N, K, dt16):
    from conformer_amd import ops
    bf = lambda t: t.to(dt16).double()
    a, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2) / math.sqrt(K), rnd(N, seed=3), rnd(M, N, seed=4)
    ref = bf(a) @ bf(w).t() + b.double()
    G = lambda t: t.to(dev)
    with torch.autocast("cuda", dtype=dt16):
        assert rel_l2(ops.linear(G(a), G(w), G(b)), ref) < 2e-5
        assert rel_l2(ops.linear(G(a), G(w), G(b), act="swish"), O.swish(ref)) < 2e-5
        assert rel_l2(ops.linear(G(a), G(w), G(b), act="relu"), torch.relu(ref)) < 2e-5
        assert rel_l2(ops.linear_residual(G(a), G(w), G(b), G(r), 0.5), 0.5 * ref + r.double()) < 2e-5
        if N % 2 == 0:
            n = N // 2
            assert rel_l2(ops.linear_glu(G(a), G(w), G(b)), ref[:, :n] * torch.sigmoid(ref[:, n:])) < 2e-5
    # and against the un-rounded product: bf16 operand rounding only (2^-9 per operand, averaged over K)
    assert rel_l2(ops.linear(G(a), G(w), G(b)), a.double() @ w.double().t() + b.double()) < 2e-5   # fp32 path outside autocast


@pytest.mark.parametrize("dt16", DT)
@pytest.mark.parametrize("I,J,Kc", [(64, 64, 64), (100, 144, 70), (257, 130, 144), (7968, 512, 2048), (512, 2048, 7968),
                                    (33, 36, 1000), (144, 576, 98)])
def test_mfma16_gemm_bwd_layouts(dev, I, J, Kc, dt16):
    """All four operand layouts, the swish' epilogue, split-K atomics and accumulate, vs fp64 products of rounded operands."""
    from conformer_amd import ops
    prec = {torch.bfloat16: ops.PREC_BF16, torch.float16: ops.PREC_FP16}[dt16]
    r16 = lambda t: t.to(dt16).double()
    Kp = (Kc + 3) // 4 * 4
    Ip, Jp = (I + 3) // 4 * 4, (J + 3) // 4 * 4
    A = rnd(I, Kc, seed=1) / math.sqrt(Kc)
    Bm = rnd(J, Kc, seed=2)
    ref = r16(A) @ r16(Bm).t()
    def lay(x, col):                       # device tensor in the requested layout (padded leading dimension)
        if col:
            buf = torch.zeros(x.shape[1], (x.shape[0] + 3) // 4 * 4)
            buf[:, : x.shape[0]] = x.t()
        else:
            buf = torch.zeros(x.shape[0], Kp)
            buf[:, : x.shape[1]] = x
        return buf.to(dev)
    for a_col in (False, True):
        for b_col in (False, True):
            Ad, Bd = lay(A, a_col), lay(Bm, b_col)
            out = torch.full((I, Jp), 7.0, device=dev)
            ops.gemm_bwd(Ad, a_col, Bd, b_col, I, J, Kc, alpha=0.5, out=out, prec=prec)
            assert rel_l2(out[:, :J], 0.5 * ref) < 2e-5, (a_col, b_col)
            assert (out[:, J:] == 7.0).all()
            out2 = torch.zeros(I, Jp, device=dev)
            ops.gemm_bwd(Ad, a_col, Bd, b_col, I, J, Kc, out=out2, allow_split=True, prec=prec)
            assert rel_l2(out2[:, :J], ref) < 2e-5, ("split", a_col, b_col)
            ops.gemm_bwd(Ad, a_col, Bd, b_col, I, J, Kc, out=out2, accumulate=True, prec=prec)
            assert rel_l2(out2[:, :J], 2 * ref) < 2e-5, ("acc", a_col, b_col)
    # B already stored in the 16-bit type (cached weight cast), contraction-major, padded leading dimension
    J8 = (J + 7) // 8 * 8
    B16 = torch.zeros(Kc, J8, dtype=dt16)
    B16[:, :J] = Bm.t().to(dt16)
    for a_col in (False, True):
        out = torch.full((I, Jp), 7.0, device=dev)
        ops.gemm_bwd(lay(A, a_col), a_col, B16.to(dev), True, I, J, Kc, alpha=0.5, out=out, prec=prec, b16=True)
        assert rel_l2(out[:, :J], 0.5 * ref) < 2e-5, ("b16", a_col)
        assert (out[:, J:] == 7.0).all()
        out2 = torch.zeros(I, Jp, device=dev)
        ops.gemm_bwd(lay(A, a_col), a_col, B16.to(dev), True, I, J, Kc, out=out2, allow_split=True, prec=prec, b16=True)
        assert rel_l2(out2[:, :J], ref) < 2e-5, ("b16 split", a_col)
    Z = rnd(I, Jp, seed=3)
    dz = ops.gemm_bwd(lay(A, False), False, lay(Bm, True), True, I, J, Kc, Z=Z.to(dev), out=torch.empty(I, Jp, device=dev),
                      prec=prec)
    zd = Z[:, :J].double()
    sg = torch.sigmoid(zd)
    assert rel_l2(dz[:, :J], ref * (sg * (1 + zd * (1 - sg)))) < 2e-5


@pytest.mark.parametrize("dt16", DT)
def test_mfma16_stem_backward_vs_fp32_path(dev, dt16):
    """Stem (C=64) forward + backward under autocast vs the fp32 HIP path: operand rounding only."""
    from conformer_amd import autograd as A
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 80, 57, generator=g).to(dev)
    C = 64
    ps = [torch.randn(C, 1, 3, 3, generator=g) / 3, torch.randn(C, generator=g) * 0.1,
          torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C), torch.randn(C, generator=g) * 0.1]
    outs = {}
    for mode in ("f32", "lowp"):
        p = [t.clone().to(dev).requires_grad_(True) for t in ps]
        if mode == "lowp":
            with torch.autocast("cuda", dtype=dt16):
                h = A.SubsampleStemFn.apply(x, *p)
        else:
            h = A.SubsampleStemFn.apply(x, *p)
        (h * (1.0 + 0.5 * torch.cos(torch.arange(h.numel(), device=dev).view_as(h) * 0.37))).sum().backward()
        outs[mode] = [h.detach()] + [q.grad for q in p]
    # (a positive cotangent keeps the gradient sums free of cancellation, so the relative error stays a rounding measure)
    tols = [1e-2] * 5 if dt16 == torch.bfloat16 else [4e-3] * 5
    errs = [rel_l2(a, b.double()) for a, b in zip(outs["lowp"], outs["f32"])]
    assert all(e < tol for e, tol in zip(errs, tols)), errs
    assert rel_l2(outs["lowp"][3], outs["f32"][3].double()) > 1e-6          # really took the 16-bit path


def test_fp16_encoder_grads_vs_fp32_golden(dev):
    """model_tiny encoder under float16 autocast (the reference's own AMP dtype, train.py:6,217,232-240), backward of a
    loss-scaled scalar as GradScaler does: every parameter gradient vs the reference's fp32 gradients.  fp16 carries three
    more mantissa bits than bf16; its budget here is 1e-2 per tensor end to end.  (bfloat16 is graded per tensor against
    the reference's own autocast goldens in tests/test_autocast_golden_gpu.py.)"""
    from model.modules.encoder import Encoder
    meta, g = load_golden("model_tiny")
    P = cfg_params(meta)
    enc = Encoder(80, meta["n_blocks"], meta["d"], meta["n_heads"], meta["ksize"], 0.0)
    enc.load_state_dict({k[len("encoder."):]: v for k, v in P.items() if k.startswith("encoder.")}, strict=True)
    enc = enc.to(dev).eval()
    with torch.autocast("cuda", dtype=torch.float16):
        y, _ = enc(g["x"].to(dev), g["lengths"].to(dev))
    assert y.dtype == torch.float32
    assert 1e-6 < rel_l2(y, g["enc"]) < 3e-3
    scale = 1024.0
    ((y * g["w"].to(dev)).sum() * scale).backward()
    worst, checked = 0.0, 0
    for n, p in enc.named_parameters():
        gk = "grad." + n
        if gk not in g or not p.requires_grad or float(g[gk].norm()) < 1e-4:
            continue
        worst = max(worst, rel_l2(p.grad / scale, g[gk]))
        checked += 1
    assert checked > 50 and worst < 1e-2, worst


def test_block_bf16_within_1e2_of_fp32_oracle(dev):
    """Conformer-L block geometry (d=512, H=8, T'=249): bf16-MFMA block output vs the fp32/fp64 oracle."""
    from model.utils.block import ConformerBlock
    from model.utils.position import RelativePositionalEncoding
    P = O.make_params(vocab=8, n_mel=80, n_blocks=1, d=512, n_heads=8, ksize=31, lstm_hidden=8, seed=5, with_decoder=False)
    blk = "encoder.layers.0."
    m = ConformerBlock(512, 8, 31).to(dev).eval()
    m.load_state_dict({k[len(blk):]: v for k, v in P.items() if k.startswith(blk)})
    x = rnd(2, 249, 512, seed=3)
    L = torch.tensor([249, 131])
    rel = RelativePositionalEncoding(512).to(dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y = m.fused(x.to(dev), rel.table(249), L.to(dev))
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        y16 = m.fused(x.to(dev), rel.table(249), L.to(dev))
    Pd = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    ref = O.conformer_block(x.double(), O.relpos_table(249, Pd["encoder.rel_pe.div_term"]), L, Pd, blk, 8)
    err = rel_l2(y, ref)
    assert 1e-5 < err < 1e-2, err            # really on the bf16 path, and inside the bf16 budget
    err16 = rel_l2(y16, ref)
    assert 1e-6 < err16 < 2e-3, err16        # fp16 operands carry 3 more mantissa bits


@pytest.mark.parametrize("dt16", DT)
@pytest.mark.parametrize("B,T,H,dh,lengths", [(2, 1, 4, 8, [1, 1]), (3, 7, 4, 8, [7, 5, 1]), (2, 49, 4, 36, [49, 39]),
                                             (2, 130, 2, 32, [130, 64]), (2, 249, 8, 64, [249, 131]), (2, 40, 2, 8, [40, 0]),
                                             (1, 300, 2, 64, None)])
def test_mfma16_attention_forward(dev, B, T, H, dh, lengths, dt16):
    """16-bit-MFMA attention vs the float64 oracle fed the ROUNDED q+u, q+v, k, v and positions (the probabilities are
    rounded once more inside the kernel for P.V: tolerance 2^-8 / 2^-11) and vs the unrounded oracle (16-bit budget)."""
    from conformer_amd import ops
    d = H * dh
    qkv = rnd(B, T, 3 * d, seed=11) * 0.5
    pos = rnd(2 * T - 1, d, seed=12) * 0.5
    u, v = rnd(H, dh, seed=13) * 0.3, rnd(H, dh, seed=14) * 0.3
    L = None if lengths is None else torch.tensor(lengths, dtype=torch.int64)
    G = lambda t: t.to(dev)
    with torch.autocast("cuda", dtype=dt16):
        ctx = ops.relpos_attention(G(qkv), G(pos), G(u), G(v), None if L is None else G(L), H)
        ctx2, lse = ops.relpos_attention_train(G(qkv), G(pos), G(u), G(v), None if L is None else G(L), H)
    assert torch.equal(ctx, ctx2) and torch.isfinite(lse).all()
    r16 = lambda t: t.to(dt16).double()
    q, k, vv = (t.reshape(B, T, H, dh) for t in qkv.split(d, dim=-1))
    zero = torch.zeros(H, dh, dtype=torch.float64)
    # oracle on rounded operands: scores from r16(q+u).r16(k) + r16(q+v).r16(p); the biases are folded into two query sets,
    # which relpos_attention_core cannot express, so compose it from its parts
    import math
    qu, qv = r16(q + u), r16(q + v)
    kk, vr, pp = r16(k), r16(vv), r16(pos).view(2 * T - 1, H, dh)
    content = torch.einsum("bihc,bkhc->bhik", qu, kk)
    full = torch.einsum("bihc,jhc->bhij", qv, pp)
    i = torch.arange(T)[:, None]; kx = torch.arange(T)[None, :]
    posb = full.gather(-1, ((T - 1) - (i - kx)).expand(B, H, T, T))
    s = (content + posb) / math.sqrt(dh)
    if L is not None:
        pad = torch.arange(T)[None, :] >= L[:, None]
        s = s.masked_fill(pad[:, None, None, :], torch.finfo(torch.float32).min)
    ref16 = torch.einsum("bhik,bkhc->bihc", torch.softmax(s, -1), vr).reshape(B, T, d)
    ref = O.relpos_attention_core(q.double(), k.double(), vv.double(), pos.double().view(2 * T - 1, H, dh), u.double(),
                                  v.double(), L)
    tol_p = 6e-3 if dt16 == torch.bfloat16 else 8e-4
    assert rel_l2(ctx, ref16) < tol_p
    assert rel_l2(ctx, ref) < (1e-2 if dt16 == torch.bfloat16 else 2e-3)


@pytest.mark.parametrize("dt16", DT)
def test_mfma16_stem_inference_keeps_16bit_activations(dev, dt16):
    """Inference under autocast: conv1 writes h1 in the 16-bit type, conv2 consumes it and writes a 16-bit h2 that the
    input Linear consumes -- against the fp32 path (operand + one output rounding)."""
    from conformer_amd import ops
    g = torch.Generator().manual_seed(3)
    C = 64
    x = torch.randn(2, 80, 103, generator=g).to(dev)
    w1, b1 = (torch.randn(C, 1, 3, 3, generator=g) / 3).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
    w2, b2 = (torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C)).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
    wl, bl = (torch.randn(48, 19 * C, generator=g) / math.sqrt(19 * C)).to(dev), torch.zeros(48, device=dev)
    w2p = ops.pack_conv2_weight(w2)
    ref_h2 = ops.subsample_stem(x, w1, b1, w2p, b2)
    ref = ops.linear(ref_h2, wl, bl)
    with torch.autocast("cuda", dtype=dt16):
        h2 = ops.subsample_stem(x, w1, b1, w2p, b2)
        y = ops.linear(h2, wl, bl)
    assert h2.dtype == dt16 and y.dtype == torch.float32
    tol = 1e-2 if dt16 == torch.bfloat16 else 2e-3
    assert rel_l2(h2.float(), ref_h2) < tol and rel_l2(y, ref) < tol


def test_weight16_refresh_after_fused_adam(dev):
    """FusedAdam re-casts the cached 16-bit weight copies (plain and transposed) in one batched launch, in place: the next
    forward / backward must see exactly the copies a fresh cast of the updated weights gives."""
    from conformer_amd import ops
    from conformer_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(5)
    ws = [torch.nn.Parameter(torch.randn(n, k, generator=g).to(dev)) for n, k in ((64, 32), (128, 96), (40, 72))]
    conv = torch.nn.Parameter(torch.randn(48, 24, 1, generator=g).to(dev))            # pointwise-conv weight: cached through a 2-D view
    opt = FusedAdam(ws + [conv], lr=1e-2)
    prec = ops.PREC_BF16
    before = [ops.weight16(w, prec) for w in ws] + [ops.weight16(w, prec, transposed=True) for w in ws[:2]]
    before.append(ops.weight16(conv.reshape(48, -1), prec))
    ptrs = [t.data_ptr() for t in before]
    for p in ws + [conv]:
        p.grad = torch.randn(p.shape, generator=g).to(dev)
    opt.step()
    after = [ops.weight16(w, prec) for w in ws] + [ops.weight16(w, prec, transposed=True) for w in ws[:2]]
    after.append(ops.weight16(conv.reshape(48, -1), prec))
    assert [t.data_ptr() for t in after] == ptrs                                      # refreshed in place, no new cast
    for t, w in zip(after[:3], ws):
        assert torch.equal(t, w.detach().to(torch.bfloat16))
    for t, w in zip(after[3:5], ws[:2]):
        assert torch.equal(t, w.detach().t().to(torch.bfloat16))
    assert torch.equal(after[5], conv.detach().reshape(48, -1).to(torch.bfloat16))


@pytest.mark.parametrize("dt16", DT)
def test_attention_16bit_projections_and_context_in_inference(dev, dt16):
    """Inference under autocast: the fused q|k|v projection is written in the 16-bit type (what autocast's nn.Linear returns) and
    so is the attention context.  (a) A 16-bit context changes nothing: its only consumer, the out-projection GEMM, rounds an
    fp32 context to that type anyway -- EQUAL outputs.  (b) 16-bit projections: k and v are rounded exactly as the kernel's staging
    rounds them, q is rounded before the bias add as well as after (the reference's arithmetic) -- within 16-bit distance."""
    from conformer_amd import ops
    from model.utils.attention import MultiHeadSelfAttentionModule
    torch.manual_seed(3)
    B, T, d, H = 3, 49, 144, 4
    m = MultiHeadSelfAttentionModule(d, H).to(dev).eval()
    x = torch.randn(B, T, d, device=dev)
    L = torch.tensor([49, 30, 7], device=dev)
    table = ops.relpos_table(torch.exp(torch.arange(0, d, 2, device=dev).float() * (-math.log(10000.0) / d)), T)
    with torch.no_grad(), torch.autocast("cuda", dtype=dt16):
        a = m.attention
        got = m.fused(x, table, L, residual=x)
        xn = ops.layernorm(x, m.layer_norm.weight, m.layer_norm.bias, m.layer_norm.eps, for_gemm=True)
        w, b = a._qkv_params()
        pos = ops.linear(table, a.pos_proj.weight, a.pos_proj.bias)
        qkv32, qkv16 = ops.linear(xn, w, b), ops.linear(xn, w, b, for_gemm=True)
        assert qkv16.dtype == dt16 and torch.equal(qkv16, qkv32.to(dt16))
        c_a = ops.relpos_attention(qkv16, pos, a.content_bias, a.position_bias, L, H)                  # fp32 context
        c_b = ops.relpos_attention(qkv16, pos, a.content_bias, a.position_bias, L, H, for_gemm=True)   # 16-bit context
        c_0 = ops.relpos_attention(qkv32, pos, a.content_bias, a.position_bias, L, H)                  # fp32 projections
        want = ops.linear_residual(c_a, a.out_proj.weight, a.out_proj.bias, x, 1.0)
    assert c_a.dtype == torch.float32 and c_b.dtype == dt16
    assert torch.equal(c_b, c_a.to(dt16))
    assert torch.isfinite(got).all() and torch.equal(got, want)
    assert rel_l2(c_a, c_0) < (6e-3 if dt16 == torch.bfloat16 else 1e-3)


@pytest.mark.parametrize("dt16", DT)
@pytest.mark.parametrize("d,H", [(100, 5), (36, 1), (20, 1)])
def test_autocast_backward_at_d_not_multiple_of_8(dev, d, H, dt16):
    """d % 8 == 4: producers must not hand a 16-bit gradient to a consumer whose aligned 16-bit kernels do
    not cover its shape.  A block's autocast training step runs and its gradients track the fp32 path's (3e-2: bf16 drift)."""
    from model.utils.block import ConformerBlock
    from model.utils.position import RelativePositionalEncoding
    P = O.make_params(vocab=8, n_mel=80, n_blocks=1, d=d, n_heads=H, ksize=7, lstm_hidden=8, seed=9, with_decoder=False)
    blk = "encoder.layers.0."
    m = ConformerBlock(d, H, 7).to(dev).eval()
    m.load_state_dict({k[len(blk):]: v for k, v in P.items() if k.startswith(blk)})
    x = rnd(2, 33, d, seed=4).to(dev)
    L = torch.tensor([33, 20], device=dev)
    table = RelativePositionalEncoding(d).to(dev).table(33)
    grads = {}
    for name, amp in (("f32", None), ("amp", dt16)):
        m.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        import contextlib
        with (torch.autocast("cuda", dtype=amp) if amp else contextlib.nullcontext()):
            y = m.fused(xi, table, L)
        (y.float() * torch.cos(torch.arange(y.numel(), device=dev).view_as(y) * 0.1)).sum().backward()
        grads[name] = {"x": xi.grad.clone(), **{n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}}
    assert len(grads["amp"]) == len(grads["f32"]) > 30
    for n, gref in grads["f32"].items():
        if float(gref.norm()) < 1e-3:
            continue                                   # (zero-by-construction gradients: graded elsewhere in absolute terms)
        assert rel_l2(grads["amp"][n], gref) < (3e-2 if dt16 == torch.bfloat16 else 8e-3), n


def test_training_attention_refuses_16bit_projections(dev):
    """relpos_attention_train reads fp32 q|k|v (a 16-bit tensor used to slip through to fp32 kernels)."""
    from conformer_amd import ops
    from conformer_amd._lib import ConformerHipError
    qkv = rnd(2, 9, 3 * 32, seed=1).to(dev)
    pos, u, v = rnd(17, 32, seed=2).to(dev), rnd(4, 8, seed=3).to(dev), rnd(4, 8, seed=4).to(dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ops.relpos_attention_train(qkv, pos, u, v, None, 4)
        with pytest.raises(ConformerHipError):
            ops.relpos_attention_train(qkv.to(torch.bfloat16), pos, u, v, None, 4)
