"""Silent out-of-bounds guard (round-1 fault 5377ffc: the attention-backward score GEMMs read up to three K rows past
the end of qkv for the last batch element -- visible only when the page behind the buffer happened to be unmapped).

Every operand of the batched / gathering / strided entry points is placed INSIDE a larger NaN-filled allocation, so the
elements immediately before and after each operand -- in particular behind the last batch element and the last head --
are poison: an over-read that reaches a result turns it non-finite and different from the run on ordinary tensors, and
an over-WRITE destroys the poison next to an output.  One pass, no repetition."""
import math

import pytest
import torch

from oracle import conformer_oracle as O
from tests.util import rel_l2

pytestmark = pytest.mark.gpu
GUARD = 1 << 15            # poisoned elements on each side of an operand (128 KiB of fp32)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


class Poisoned:
    def __init__(self, dev):
        self.dev, self.bufs = dev, []

    def place(self, t: torch.Tensor) -> torch.Tensor:
        """A copy of `t` whose storage is surrounded by NaN on both sides (16-byte alignment kept)."""
        n = t.numel()
        buf = torch.full((n + 2 * GUARD,), float("nan"), device=self.dev, dtype=t.dtype)
        v = buf[GUARD:GUARD + n].view(t.shape)
        v.copy_(t)
        self.bufs.append((buf, n))
        return v

    def like(self, shape, dtype=torch.float32, fill=None) -> torch.Tensor:
        t = torch.zeros(shape, dtype=dtype) if fill is None else torch.full(shape, fill, dtype=dtype)
        return self.place(t.to(self.dev))

    def intact(self) -> bool:
        return all(bool(torch.isnan(b[:GUARD]).all()) and bool(torch.isnan(b[GUARD + n:]).all()) for b, n in self.bufs)


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("amp", [None, torch.bfloat16])
def test_attention_forward_backward_next_to_poison(dev, amp):
    from conformer_amd import ops
    B, T, H, dh = 3, 49, 4, 36                      # T not a multiple of 4 or 32; Conformer-S head size
    d = H * dh
    qkv, pos = rnd(B, T, 3 * d, seed=1) * 0.5, rnd(2 * T - 1, d, seed=2) * 0.5
    u, v, dctx = rnd(H, dh, seed=3) * 0.3, rnd(H, dh, seed=4) * 0.3, rnd(B, T, d, seed=5)
    L = torch.tensor([T, 33, 7])
    P = Poisoned(dev)
    G = lambda t: t.to(dev)

    def run(place):
        import contextlib
        with (torch.autocast("cuda", dtype=amp) if amp else contextlib.nullcontext()):
            a = [place(G(t)) for t in (qkv, pos, u, v)]
            ctx, lse = ops.relpos_attention_train(*a, G(L), H)
            ctx_p, lse_p, dctx_p = place(ctx), place(lse), place(G(dctx))
            return (ctx,) + tuple(ops.relpos_attention_bwd(*a, G(L), H, ctx_p, lse_p, dctx_p))

    plain = run(lambda t: t.clone())
    guarded = run(P.place)
    assert P.intact()
    for a, b in zip(guarded, plain):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b) or rel_l2(a, b) < 1e-5      # (dpos / du / dv are atomic sums: order-dependent rounding)
    if amp is None:
        q, k, vv = (t.reshape(B, T, H, dh).double().requires_grad_(True) for t in qkv.split(d, dim=-1))
        pp = pos.double().view(2 * T - 1, H, dh).requires_grad_(True)
        ud, vd = u.double().requires_grad_(True), v.double().requires_grad_(True)
        ref = O.relpos_attention_core(q, k, vv, pp, ud, vd, L)
        (ref.reshape(B, T, d) * dctx.double()).sum().backward()
        assert rel_l2(guarded[0], ref.reshape(B, T, d)) < 2e-5
        dqkv_ref = torch.cat([q.grad.reshape(B, T, d), k.grad.reshape(B, T, d), vv.grad.reshape(B, T, d)], dim=-1)
        assert rel_l2(guarded[1], dqkv_ref) < 5e-5
        assert rel_l2(guarded[2], pp.grad.reshape(2 * T - 1, d)) < 5e-5
        assert rel_l2(guarded[3], ud.grad) < 5e-5 and rel_l2(guarded[4], vd.grad) < 5e-5


@pytest.mark.parametrize("prec_name", ["f32", "bf16"])
def test_batched_gemm_bwd_layouts_next_to_poison(dev, prec_name):
    """The general backward GEMM: all four operand layouts, batched over (nb0, nb1) with strides, ragged I / J / Kc, output
    written into a poisoned pool (pad columns of ldc must stay untouched)."""
    from conformer_amd import ops
    prec = {"f32": ops.PREC_F32, "bf16": ops.PREC_BF16}[prec_name]
    nb0, nb1, I, J, Kc = 2, 3, 37, 50, 45
    r = (lambda t: t.to(torch.bfloat16).double()) if prec_name == "bf16" else (lambda t: t.double())
    P = Poisoned(dev)
    for a_col in (False, True):
        for b_col in (False, True):
            lda = ((I if a_col else Kc) + 3) // 4 * 4
            ldb = ((J if b_col else Kc) + 3) // 4 * 4
            ldc = (J + 3) // 4 * 4
            ra, rb = (Kc if a_col else I), (Kc if b_col else J)
            A = torch.zeros(nb0, nb1, ra, lda); A[..., : (I if a_col else Kc)] = rnd(nb0, nb1, ra, I if a_col else Kc, seed=1)
            Bm = torch.zeros(nb0, nb1, rb, ldb); Bm[..., : (J if b_col else Kc)] = rnd(nb0, nb1, rb, J if b_col else Kc, seed=2)
            Aop = (A[..., :I].transpose(-1, -2) if a_col else A[..., :Kc])
            Bop = (Bm[..., :J].transpose(-1, -2) if b_col else Bm[..., :Kc])
            ref = r(Aop) @ r(Bop).transpose(-1, -2)
            out = P.like((nb0, nb1, I, ldc), fill=7.0)
            ops.gemm_bwd(P.place(A.to(dev)), a_col, P.place(Bm.to(dev)), b_col, I, J, Kc, out=out, lda=lda, ldb=ldb, ldc=ldc,
                         nbatch=nb0 * nb1, nb1=nb1, sa=(nb1 * ra * lda, ra * lda), sb=(nb1 * rb * ldb, rb * ldb),
                         sc=(nb1 * I * ldc, I * ldc), prec=prec)
            assert torch.isfinite(out).all() and rel_l2(out[..., :J], ref) < 2e-5, (a_col, b_col)
            assert (out[..., J:] == 7.0).all(), (a_col, b_col)
    assert P.intact()


def test_stem_gathers_next_to_poison(dev):
    """The conv-subsampling stem: implicit-GEMM gather of h1 (forward), im2col and transposed-conv gathers (backward)."""
    from conformer_amd import ops
    g = torch.Generator().manual_seed(0)
    B, T, C = 2, 57, 64
    x = torch.randn(B, 80, T, generator=g)
    w1, b1 = torch.randn(C, 1, 3, 3, generator=g) / 3, torch.randn(C, generator=g) * 0.1
    w2, b2 = torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C), torch.randn(C, generator=g) * 0.1
    G = lambda t: t.to(dev)

    def run(place):
        xs, w1s, b1s, w2s, b2s = (place(G(t)) for t in (x, w1, b1, w2, b2))
        w2p = place(ops.pack_conv2_weight(w2s))
        h2, h1 = ops.subsample_stem_train(xs, w1s, b1s, w2p, b2s)
        dh2 = torch.cos(torch.arange(h2.numel(), device=dev, dtype=torch.float32).view_as(h2) * 0.37)
        return (h2,) + tuple(ops.subsample_stem_bwd(xs, w1s, b1s, w2s, place(h1), place(h2), place(dh2)))

    P = Poisoned(dev)
    plain, guarded = run(lambda t: t.clone()), run(P.place)
    assert P.intact()
    for a, b in zip(guarded, plain):
        assert torch.isfinite(a).all() and rel_l2(a, b) < 1e-5
    ref = torch.relu(torch.nn.functional.conv2d(torch.relu(torch.nn.functional.conv2d(
        x.double()[:, None], w1.double(), b1.double(), stride=2)), w2.double(), b2.double(), stride=2))
    ref = ref.permute(0, 3, 1, 2).reshape(B, ref.shape[3], -1)                      # convolution.py:51-52 feature order c*19+f
    got = guarded[0].view(B, -1, 19, C).permute(0, 1, 3, 2).reshape(B, -1, 19 * C)    # (b,t,f,c) -> (b,t,c*19+f)
    assert rel_l2(got, ref) < 2e-5


def test_forward_gemms_dwconv_layernorm_next_to_poison(dev):
    """Forward GEMM epilogues with ragged M / N / K, the depthwise conv window (frames -15..+15 around both sequence ends
    of the last utterance) and LayerNorm rows, all operands inside poison."""
    from conformer_amd import ops
    P = Poisoned(dev)
    G = lambda t: P.place(t.to(dev))
    M, N, K = 249 * 3 + 1, 130, 52
    a, w, b, res = rnd(M, K, seed=1), rnd(N, K, seed=2) / math.sqrt(K), rnd(N, seed=3), rnd(M, N, seed=4)
    ref = a.double() @ w.double().t() + b.double()
    assert rel_l2(ops.linear(G(a), G(w), G(b)), ref) < 2e-5
    assert rel_l2(ops.linear(G(a), G(w), G(b), act="swish"), O.swish(ref)) < 2e-5
    assert rel_l2(ops.linear_residual(G(a), G(w), G(b), G(res), 0.5), 0.5 * ref + res.double()) < 2e-5
    assert rel_l2(ops.linear_glu(G(a), G(w), G(b)), ref[:, :N // 2] * torch.sigmoid(ref[:, N // 2:])) < 2e-5
    with torch.autocast("cuda", dtype=torch.bfloat16):
        r16 = a.to(torch.bfloat16).double() @ w.to(torch.bfloat16).double().t() + b.double()
        assert rel_l2(ops.linear(G(a), G(w), G(b)), r16) < 2e-5
    B, T, C, Kd = 2, 41, 96, 31
    g_, wd, bd = rnd(B, T, C, seed=5), rnd(C, 1, Kd, seed=6) / 5, rnd(C, seed=7) * 0.1
    bw, bb, bm, bv = rnd(C, seed=8) * 0.2 + 1, rnd(C, seed=9) * 0.1, rnd(C, seed=10) * 0.1, rnd(C, seed=11).abs() + 0.5
    y = ops.dwconv_bn_swish(G(g_), G(wd), G(bd), G(bw), G(bb), G(bm), G(bv))
    c = torch.nn.functional.conv1d(g_.double().transpose(1, 2), wd.double(), bd.double(), padding=Kd // 2, groups=C)
    bn = (c - bm.double()[:, None]) / torch.sqrt(bv.double()[:, None] + 1e-5) * bw.double()[:, None] + bb.double()[:, None]
    assert rel_l2(y, O.swish(bn).transpose(1, 2)) < 2e-5
    dy = rnd(B, T, C, seed=12)
    outs = ops.dwconv_bn_swish_bwd(G(g_), G(dy), G(wd), G(bd), G(bw), G(bb), G(bm), G(bv), 1e-5, True)
    assert all(torch.isfinite(o).all() for o in outs)
    x, lw, lb = rnd(77, 144, seed=13), rnd(144, seed=14), rnd(144, seed=15)
    assert rel_l2(ops.layernorm(G(x), G(lw), G(lb)), torch.nn.functional.layer_norm(x.double(), (144,), lw.double(), lb.double())) < 2e-5
    assert P.intact()


def test_round2_training_kernels_next_to_poison(dev):
    """The kernels added in round 2 load with clamped addresses + selects instead of branches: the weight(+bias)-gradient kernel
    (ragged contraction length: the last K-tile of the last split is partial), the fused Swish-backward epilogue with 16-bit Z
    and output, the one-pass LayerNorm backward (ragged row count per wave) and the 16-bit LSTM recurrence."""
    from conformer_amd import ops
    P = Poisoned(dev)
    G = lambda t: t.to(dev)
    m, n, k = 7968 // 8 + 3 * 8, 256, 128                 # m = 1020: not a multiple of 64 (contraction tile) nor of the split
    x, w, dy, z = rnd(m, k, seed=1), rnd(n, k, seed=2) * 0.1, rnd(m, n, seed=3), rnd(m, n, seed=4)
    w2 = rnd(k, n, seed=5) * 0.1

    def run(place):
        out = []
        with torch.autocast("cuda", dtype=torch.bfloat16):
            xs = place(G(x).to(torch.bfloat16))                                   # activation stored by its producer in bf16
            dx, dw, db = ops.linear_bwd(xs, place(G(w)), place(G(dy)))             # dX (general kernel), dW+db (gemm_dw16: A fp32, B bf16)
            out += [dx, dw, db]
            zs = place(G(z).to(torch.bfloat16))
            dz, dw2, db2 = ops.linear_bwd(place(G(rnd(m, n, seed=6)).to(torch.bfloat16)), place(G(w2)), place(G(rnd(m, k, seed=7))),
                                          alpha=0.5, Z=zs, dx16=True)              # EPI 5: swish'(Z) epilogue, 16-bit Z and result
            out += [dz.float(), dw2, db2]
            dx3, dw3, db3 = ops.linear_bwd(xs, place(G(w)), place(G(dy).to(torch.bfloat16)))   # 16-bit dY: forward-kernel dX + all-16-bit dW
            out += [dx3, dw3, db3]
        return out

    plain = run(lambda t: t.clone())
    guarded = run(P.place)
    assert P.intact()
    for a, b in zip(guarded, plain):
        assert torch.isfinite(a).all()
        assert rel_l2(a, b) < 1e-3                      # (split-K atomics: not bit-identical)
    # LayerNorm backward, fused: 1003 rows (ragged rows per wave), d = 144
    rows, d = 1003, 144
    xl, gl, dyl, dr = rnd(rows, d, seed=8), rnd(d, seed=9), rnd(rows, d, seed=10), rnd(rows, d, seed=11)
    mean, rstd = xl.mean(-1), 1.0 / torch.sqrt(xl.var(-1, unbiased=False) + 1e-5)
    P2 = Poisoned(dev)
    ref = ops.layernorm_bwd(G(xl), G(gl), G(dyl), G(mean), G(rstd), dres=G(dr))
    got = ops.layernorm_bwd(P2.place(G(xl)), P2.place(G(gl)), P2.place(G(dyl)), P2.place(G(mean)), P2.place(G(rstd)), dres=P2.place(G(dr)))
    assert P2.intact()
    for a, b in zip(got, ref):
        assert torch.isfinite(a).all() and torch.equal(a, b)     # deterministic by construction
    # 16-bit LSTM forward: B = 37 (ragged utterance block), H = 48 (three MFMA steps: waves 3 of 4 idle), ragged lengths
    B, T, D, H = 37, 9, 32, 48
    xx, wih, whh, bias = rnd(B, T, D, seed=12), rnd(4 * H, D, seed=13) * 0.2, rnd(4 * H, H, seed=14) * 0.2, rnd(4 * H, seed=15) * 0.1
    L = torch.sort(torch.randint(1, T + 1, (B,), generator=torch.Generator().manual_seed(3)), descending=True).values
    P3 = Poisoned(dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ref = ops.lstm_forward(G(xx), G(wih), G(whh), G(bias), G(L), save=True)
        got = ops.lstm_forward(P3.place(G(xx)), P3.place(G(wih)), P3.place(G(whh)), P3.place(G(bias)), G(L), save=True)
    assert P3.intact()
    for a, b in zip(got, ref):
        assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.parametrize("C", [64, 256])
def test_stem_backward_bf16_class_gather_next_to_poison(dev, C):
    """Under autocast the stem's input-gradient product runs on the forward 16-bit GEMM kernel as four parity-class implicit GEMMs
    (per-row tap validity, scattered output rows; 128x128 tiles at C = 64, the 8-wave 256x256 tiles at C = 256): operands inside
    poison, odd T1 / F1 extents; every dh1 element written exactly once; equal to the general backward kernel's result (same
    operand rounding, different summation order) and within bf16 distance of the float64 transposed convolution."""
    from conformer_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(1)
    B, T1, F1 = 2, 29, 39                                   # h1 extents (odd); T2 = 14, F2 = 19
    T2, F2 = (T1 - 1) // 2, (F1 - 1) // 2
    dz2 = torch.randn(B, T2, F2, C, generator=g)
    w2 = torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C)
    P = Poisoned(dev)
    st = torch.cuda.current_stream().cuda_stream
    w2d = w2.to(dev)
    w2c = torch.empty(9 * C * C, device=dev)
    assert lib.cfm_pack_conv2_weight_t_f32(w2d.data_ptr(), w2c.data_ptr(), C, st) == 0
    w2c16 = P.place(w2c.to(torch.bfloat16))
    dzp = P.place(dz2.to(dev))
    zb = P.place(torch.zeros(C, device=dev))
    dh1 = P.like((B, T1, F1, C), fill=float("inf"))       # every element must be overwritten
    assert lib.cfm_subsample_conv2_bwd_input_fwdkernel_mfma16_f32(1, dzp.data_ptr(), 0, w2c16.data_ptr(), zb.data_ptr(), dh1.data_ptr(),
                                                                  B, F1, T1, C, st) == 0
    old = torch.empty(B, T1, F1, C, device=dev)
    assert lib.cfm_subsample_conv2_bwd_input_mfma16_f32(1, dzp.data_ptr(), w2c.data_ptr(), old.data_ptr(), B, F1, T1, C, st) == 0
    torch.cuda.synchronize()
    assert P.intact() and torch.isfinite(dh1).all()
    assert rel_l2(dh1, old) < 1e-5
    # float64: h2[b,co,f2,t2] = conv2d(h1[b,ci,f1,t1], w2, stride 2)  ->  dh1 = conv_transpose2d(dz2)
    d = dz2.double().permute(0, 3, 2, 1)                    # (B, C, F2, T2)
    ref = torch.nn.functional.conv_transpose2d(d, w2.double(), stride=2, output_padding=(F1 - (2 * F2 + 1), T1 - (2 * T2 + 1)))
    ref = ref.permute(0, 3, 2, 1)                           # (B, T1, F1, C)
    assert rel_l2(dh1, ref) < 1e-2
    # ---- the weight gradient from a 16-bit h1: im2col gather through the position table inside the weight-gradient kernel
    h1 = torch.relu(torch.randn(B, T1, F1, C, generator=g)).to(torch.bfloat16)
    h1p = P.place(h1.to(dev))
    n_tab = int(lib.cfm_subsample_conv2_rowtab_elems(B, F1, T1))
    tab = torch.empty(n_tab, device=dev, dtype=torch.int32)
    dw_new = P.like((C, 9 * C))
    assert lib.cfm_subsample_conv2_bwd_weight_h16_mfma16_f32(1, dzp.data_ptr(), 0, h1p.data_ptr(), tab.data_ptr(), dw_new.data_ptr(),
                                                             None, B, F1, T1, C, st) == 0
    dw_old = torch.zeros(C, 9 * C, device=dev)
    h1f = h1.to(dev).float()                                # exactly representable: the general kernel's rounding is the identity
    assert lib.cfm_subsample_conv2_bwd_weight_mfma16_f32(1, dzp.data_ptr(), h1f.data_ptr(), dw_old.data_ptr(), B, F1, T1, C, st) == 0
    torch.cuda.synchronize()
    assert P.intact() and torch.isfinite(dw_new).all() and rel_l2(dw_new, dw_old) < 1e-4
    # float64: dw2[co][ci][kf][kt] = sum dz2 * h1 patches  (conv2d weight gradient), packed (co, kf, kt, ci)
    hh = h1.double().permute(0, 3, 2, 1).requires_grad_(False)          # (B, C, F1, T1)
    wz = torch.zeros(C, C, 3, 3, dtype=torch.float64, requires_grad=True)
    out = torch.nn.functional.conv2d(hh, wz, stride=2)                   # (B, C, F2, T2)
    (out * dz2.double().permute(0, 3, 2, 1)).sum().backward()
    ref_dw = wz.grad.permute(0, 2, 3, 1).reshape(C, 9 * C)               # (co, kf, kt, ci)
    assert rel_l2(dw_new, ref_dw) < 1e-2
    # ---- 16-bit dz2 (cfm_relu_bwd_out16_f32's output type): both kernels with a 16-bit A operand, bias gradient fused
    dz16 = P.place(dz2.to(dev).to(torch.bfloat16))
    dzr = dz16.float()                                      # the same values as fp32: the fp32-A variants round them identically
    dh1_b, dh1_a = P.like((B, T1, F1, C), fill=float("inf")), torch.empty(B, T1, F1, C, device=dev)
    assert lib.cfm_subsample_conv2_bwd_input_fwdkernel_mfma16_f32(1, dz16.data_ptr(), 1, w2c16.data_ptr(), zb.data_ptr(),
                                                                  dh1_b.data_ptr(), B, F1, T1, C, st) == 0
    assert lib.cfm_subsample_conv2_bwd_input_fwdkernel_mfma16_f32(1, dzr.data_ptr(), 0, w2c16.data_ptr(), zb.data_ptr(),
                                                                  dh1_a.data_ptr(), B, F1, T1, C, st) == 0
    dw_b, db_b = P.like((C, 9 * C)), P.like((C,))
    assert lib.cfm_subsample_conv2_bwd_weight_h16_mfma16_f32(1, dz16.data_ptr(), 1, h1p.data_ptr(), tab.data_ptr(), dw_b.data_ptr(),
                                                             db_b.data_ptr(), B, F1, T1, C, st) == 0
    dw_a = torch.zeros(C, 9 * C, device=dev)
    assert lib.cfm_subsample_conv2_bwd_weight_h16_mfma16_f32(1, dzr.data_ptr(), 0, h1p.data_ptr(), tab.data_ptr(), dw_a.data_ptr(),
                                                             None, B, F1, T1, C, st) == 0
    torch.cuda.synchronize()
    assert P.intact() and torch.isfinite(dh1_b).all() and torch.isfinite(dw_b).all()
    assert rel_l2(dh1_b, dh1_a) < 1e-5 and rel_l2(dw_b, dw_a) < 1e-4
    assert rel_l2(db_b, dzr.double().sum(dim=(0, 1, 2)).float()) < 1e-5
    # ---- dh1 stored in the 16-bit type (its only consumer: the conv1 parameter gradients): the same accumulators rounded once at
    #      the store, every element written, poison intact; the conv1 reduction from it = the fp32 reduction of the same values
    dh1_16 = torch.full((B, T1, F1, C), float("inf"), device=dev).to(torch.bfloat16)
    dh1_16 = P.place(dh1_16)
    assert lib.cfm_subsample_conv2_bwd_input_fwdkernel_out16_mfma16_f32(1, dz16.data_ptr(), 1, w2c16.data_ptr(), zb.data_ptr(),
                                                                        dh1_16.data_ptr(), B, F1, T1, C, st) == 0
    torch.cuda.synchronize()
    assert P.intact() and torch.isfinite(dh1_16.float()).all()
    assert torch.equal(dh1_16, dh1_b.to(torch.bfloat16))
    F, T = 2 * F1 + 1, 2 * T1 + 2                           # an input extent that gives (F1, T1)
    x = P.place(torch.randn(B, F, T, generator=g).to(dev))
    w1 = P.place((torch.randn(C, 1, 3, 3, generator=g) / 3).to(dev))
    b1 = P.place((torch.randn(C, generator=g) * 0.1).to(dev))
    dw_16, db_16 = P.like((C, 9), fill=0.0), P.like((C,), fill=0.0)
    dw_32, db_32 = torch.zeros(C, 9, device=dev), torch.zeros(C, device=dev)
    assert lib.cfm_subsample_conv1_bwd_d16_f32(1, x.data_ptr(), w1.data_ptr(), b1.data_ptr(), dh1_16.data_ptr(), dw_16.data_ptr(),
                                               db_16.data_ptr(), B, F, T, C, st) == 0
    up = dh1_16.float()
    assert lib.cfm_subsample_conv1_bwd_f32(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), up.data_ptr(), dw_32.data_ptr(), db_32.data_ptr(),
                                           B, F, T, C, st) == 0
    torch.cuda.synchronize()
    assert P.intact() and rel_l2(dw_16, dw_32) < 1e-5 and rel_l2(db_16, db_32) < 1e-5
    # float64: gradient of sum(relu(conv1(x)) * dh1) w.r.t. (w1, b1)
    xd = x.double().cpu().unsqueeze(1)                       # (B, 1, F, T)
    w1d, b1d = w1.double().cpu().requires_grad_(True), b1.double().cpu().requires_grad_(True)
    h = torch.relu(torch.nn.functional.conv2d(xd, w1d, b1d, stride=2))   # (B, C, F1, T1)
    (h * up.double().cpu().permute(0, 3, 2, 1)).sum().backward()
    assert rel_l2(dw_16, w1d.grad.reshape(C, 9)) < 1e-4 and rel_l2(db_16, b1d.grad) < 1e-4



@pytest.mark.parametrize("M,N,K", [(300, 264, 128), (300, 260, 128), (77, 72, 64), (7968, 2048, 64), (4100, 512, 128)])
@pytest.mark.parametrize("epi", ["bias", "swish", "resid", "dswish"])
def test_gemm16_epilogue_paths_next_to_poison(dev, M, N, K, epi):
    """The 16-bit GEMM's row-major epilogue has a 4-column and an 8-column (16-byte stores for 16-bit outputs) form, fetches
    its bias / residual / Z operands ahead of the stores, and loops over row groups: every form is checked through the C ABI
    against float64 on the rounded operands, with every output inside a NaN-poisoned slab (nothing outside it may change)
    -- N % 8 == 0 takes the wide form for 16-bit outputs, N % 8 == 4 the narrow one, 7968 x 2048 the 256x256 tile."""
    from conformer_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    bf = torch.bfloat16
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    zin = torch.randn(M, N, generator=g).to(dev).to(bf)
    a16, w16 = a.to(bf).contiguous(), w.to(bf).contiguous()
    acc = a16.double() @ w16.double().t()
    st = torch.cuda.current_stream().cuda_stream

    def slab(dtype):                                                   # output of M x N inside a poisoned allocation
        full = torch.full((M + 16, N), float("nan"), device=dev, dtype=dtype)
        return full, full[8:8 + M]

    for c16 in ((False, True) if epi != "resid" else (False,)):
        cfull, c = slab(bf if c16 else torch.float32)
        z16 = N % 8 == 0                                               # (the entry point takes 16-bit Z tensors at N % 8 == 0 only)
        zfull, z = slab(bf if z16 else torch.float32)
        if epi == "bias":
            rc = lib.cfm_gemm_mfma16_f32(1, 0, a16.data_ptr(), 1, w16.data_ptr(), 1, bias.data_ptr(), None, 1.0, c.data_ptr(), int(c16),
                                         None, 0, M, N, K, K, N, N, 0.0, 0, st)
            want = acc + bias.double()
        elif epi == "swish":
            rc = lib.cfm_gemm_mfma16_f32(1, 1, a16.data_ptr(), 1, w16.data_ptr(), 1, bias.data_ptr(), None, 1.0, c.data_ptr(), int(c16),
                                         z.data_ptr(), int(z16), M, N, K, K, N, N, 0.0, 0, st)
            zz = acc + bias.double()
            want = zz * torch.sigmoid(zz)
        elif epi == "resid":
            rc = lib.cfm_gemm_mfma16_f32(1, 4, a16.data_ptr(), 1, w16.data_ptr(), 1, bias.data_ptr(), res.data_ptr(), 0.5, c.data_ptr(), 0,
                                         None, 0, M, N, K, K, N, N, 0.0, 0, st)
            want = 0.5 * (acc + bias.double()) + res.double()
        else:
            rc = lib.cfm_gemm_mfma16_f32(1, 5, a16.data_ptr(), 1, w16.data_ptr(), 1, None, None, 0.7, c.data_ptr(), int(c16),
                                         zin.data_ptr(), 1, M, N, K, K, N, N, 0.0, 0, st)
            if N % 8:                                                  # refused, not mis-indexed
                assert rc != 0
                continue
            zd = zin.double()
            sg = torch.sigmoid(zd)
            want = 0.7 * acc * (sg * (1 + zd * (1 - sg)))
        assert rc == 0, (epi, c16, rc)
        torch.cuda.synchronize()
        tol = 6e-3 if c16 else 2e-5                                   # a bf16 result carries its own rounding (2^-9 relative)
        err = (c.double() - want).abs().max() / want.abs().max()
        assert torch.isfinite(c.float()).all() and err < tol, (epi, c16, float(err))
        assert torch.isnan(cfull[:8].float()).all() and torch.isnan(cfull[8 + M:].float()).all(), (epi, c16, "C neighbours")
        if epi == "swish":
            zerr = (z.double() - (acc + bias.double())).abs().max() / (acc + bias.double()).abs().max()
            assert zerr < 6e-3, float(zerr)
            assert torch.isnan(zfull[:8].float()).all() and torch.isnan(zfull[8 + M:].float()).all(), "Z neighbours"
