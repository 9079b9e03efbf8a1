"""GPU parity of every C-ABI entry point against the CPU oracle on identical seeded inputs.

Tolerance (north_star): 1e-3 rel per tensor for fp32; these kernels accumulate in fp32 with accurate
exp/sin/cos/div, so the tests hold them to 2e-5 (H10: argmax stability wants <~1e-5 per op).
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


@pytest.mark.parametrize("rows,d", [(1, 32), (1000, 144), (7968, 512), (5, 2048), (3, 36)])
def test_layernorm(ops, rows, d):
    x, w, b = rnd(rows, d, seed=1) * 3 + 1, rnd(d, seed=2), rnd(d, seed=3)
    y = ops.layernorm(G(x), G(w), G(b))
    assert rel_l2(y, O.layer_norm(x.double(), w.double(), b.double())) < TOL


GEMM_SHAPES = [(1, 16, 16), (100, 144, 144), (257, 576, 144), (130, 144, 576), (7968, 512, 512), (300, 2048, 512),
               (300, 512, 2048), (64, 32, 2736), (129, 130, 20)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_epilogues(ops, M, N, K):
    a, w, b = rnd(M, K, seed=4), rnd(N, K, seed=5) / math.sqrt(K), rnd(N, seed=6)
    ref = a.double() @ w.double().t() + b.double()
    assert rel_l2(ops.linear(G(a), G(w), G(b)), ref) < TOL
    assert rel_l2(ops.linear(G(a), G(w), G(b), act="swish"), O.swish(ref)) < TOL
    assert rel_l2(ops.linear(G(a), G(w), G(b), act="relu"), torch.relu(ref)) < TOL
    r = rnd(M, N, seed=7)
    assert rel_l2(ops.linear_residual(G(a), G(w), G(b), G(r), 0.5), 0.5 * ref + r.double()) < TOL
    if N % 2 == 0:
        n = N // 2
        assert rel_l2(ops.linear_glu(G(a), G(w), G(b)), ref[:, :n] * torch.sigmoid(ref[:, n:])) < TOL


def test_gemm_batched_leading_dims(ops):
    a, w, b = rnd(3, 50, 64, seed=8), rnd(96, 64, seed=9), rnd(96, seed=10)
    y = ops.linear(G(a), G(w), G(b))
    assert y.shape == (3, 50, 96)
    assert rel_l2(y, a.double() @ w.double().t() + b.double()) < TOL


@pytest.mark.parametrize("T,d", [(1, 32), (7, 32), (49, 144), (249, 512)])
def test_relpos_table(ops, T, d):
    dt = torch.exp(torch.arange(0, d, 2) * -(math.log(10000.0) / d)).unsqueeze(0)
    pe = ops.relpos_table(G(dt), T)
    ref = O.relpos_table(T, dt.double())      # fp64 trig of the fp32 div_term
    # angles up to T rad in fp32: |d sin| <= ulp(angle) ~ 1.5e-5 at 249 rad -> compare against the fp32 oracle too
    assert float((pe.cpu() - O.relpos_table(T, dt)).abs().max()) < 4e-6
    assert float((pe.cpu().double() - ref).abs().max()) < 3e-5


ATT_CASES = [  # B, T, H, dh, lengths
    (2, 1, 4, 8, [1, 1]), (3, 7, 4, 8, [7, 5, 1]), (2, 48, 4, 8, [48, 33]), (2, 49, 4, 36, [49, 39]),
    (2, 70, 1, 64, [70, 2]), (2, 33, 2, 16, None), (2, 130, 2, 32, [130, 64]), (2, 249, 8, 64, [249, 131]),
    (1, 300, 2, 64, [300]), (2, 40, 2, 8, [40, 0]),          # lengths 0: the reference's uniform-softmax degenerate case
    (1, 4999, 1, 36, [4999]),                                # BASELINE cfg-5 utterance length (T = 20000 mel frames)
]


@pytest.mark.parametrize("B,T,H,dh,lengths", ATT_CASES)
def test_relpos_attention(ops, B, T, H, dh, lengths):
    d = H * dh
    qkv = rnd(B, T, 3 * d, seed=11)
    pos = rnd(2 * T - 1, d, seed=12)
    u, v = rnd(H, dh, seed=13, scale=0.3), rnd(H, dh, seed=14, scale=0.3)
    L = None if lengths is None else torch.tensor(lengths, dtype=torch.int64)
    ctx = ops.relpos_attention(G(qkv), G(pos), G(u), G(v), None if L is None else G(L), H)
    q, k, vv = (t.reshape(B, T, H, dh).double() for t in qkv.split(d, dim=-1))
    ref = O.relpos_attention_core(q, k, vv, pos.double().view(2 * T - 1, H, dh), u.double(), v.double(), L)
    assert rel_l2(ctx, ref) < TOL


def test_relpos_attention_sharp_softmax(ops):
    """Forces big running-max jumps between key tiles (online-softmax rescale path)."""
    B, T, H, dh = 1, 200, 2, 64
    d = H * dh
    qkv = rnd(B, T, 3 * d, seed=15)
    qkv[:, :, :2 * d] *= 6.0                       # |scores| ~ 36*8 -> near one-hot rows, maxima in late tiles
    pos = rnd(2 * T - 1, d, seed=16)
    u, v = rnd(H, dh, seed=17), rnd(H, dh, seed=18)
    ctx = ops.relpos_attention(G(qkv), G(pos), G(u), G(v), None, H)
    q, k, vv = (t.reshape(B, T, H, dh).double() for t in qkv.split(d, dim=-1))
    ref = O.relpos_attention_core(q, k, vv, pos.double().view(2 * T - 1, H, dh), u.double(), v.double(), None)
    assert rel_l2(ctx, ref) < 1e-4


@pytest.mark.parametrize("B,T,C,K", [(2, 1, 32, 7), (3, 7, 32, 31), (2, 49, 144, 31), (2, 249, 512, 31),
                                     (2, 70, 64, 15), (2, 20, 40, 9), (1, 100, 32, 3)])
def test_dwconv_bn_swish(ops, B, T, C, K):
    g = rnd(B, T, C, seed=19)
    w, b = rnd(C, 1, K, seed=20) / math.sqrt(K), rnd(C, seed=21, scale=0.1)
    bw, bb, bm = 1 + 0.1 * rnd(C, seed=22), 0.1 * rnd(C, seed=23), 0.1 * rnd(C, seed=24)
    bv = 0.5 + torch.rand(C, generator=torch.Generator().manual_seed(25))
    y = ops.dwconv_bn_swish(G(g), G(w), G(b), G(bw), G(bb), G(bm), G(bv))
    gd = torch.nn.functional.pad(g.double(), (0, 0, (K - 1) // 2, (K - 1) // 2))
    c = b.double().expand(B, T, C).clone()
    for j in range(K):
        c = c + gd[:, j:j + T] * w.double()[:, 0, j]
    ref = O.swish((c - bm.double()) / torch.sqrt(bv.double() + 1e-5) * bw.double() + bb.double())
    assert rel_l2(y, ref) < TOL


@pytest.mark.parametrize("B,T,C", [(3, 57, 32), (1, 31, 144), (2, 200, 144), (2, 103, 64)])
def test_subsample_stem(ops, B, T, C):
    P = O.make_params(vocab=5, n_mel=80, n_blocks=0, d=C, n_heads=1, ksize=3, lstm_hidden=4, seed=77)
    pre = "encoder.downsampling_conv."
    x = rnd(B, 80, T, seed=26)
    w2p = ops.pack_conv2_weight(G(P[pre + "conv_2.weight"]))
    h2 = ops.subsample_stem(G(x), G(P[pre + "conv_1.weight"]), G(P[pre + "conv_1.bias"]), w2p, G(P[pre + "conv_2.bias"]))
    ref = O.conv_subsampling(x.double(), {k: v.double() for k, v in P.items() if v.is_floating_point()}, pre)
    T2, F2 = ref.shape[1], 19
    # ours is (B,T2,[f][c]); the reference flattens [c][f]
    ours = h2.cpu().view(B, T2, F2, C).permute(0, 1, 3, 2).reshape(B, T2, C * F2)
    assert rel_l2(ours, ref) < TOL
    # packed input-linear weight: x_ours @ wlp.T == x_ref @ wl.T
    wl = P["encoder.linear.weight"]
    wlp = ops.pack_linear_weight(G(wl), C, F2)
    y = ops.linear(h2, wlp, G(P["encoder.linear.bias"]))
    assert rel_l2(y, ref @ wl.double().t() + P["encoder.linear.bias"].double()) < TOL


def test_errors_are_loud(ops):
    from conformer_amd._lib import ConformerHipError
    with pytest.raises(ConformerHipError):
        ops.layernorm(torch.zeros(4, 32), torch.ones(32), torch.zeros(32))          # CPU tensor: no fallback
    with pytest.raises(ConformerHipError):
        ops.linear(G(torch.zeros(4, 30)), G(torch.zeros(8, 30)), G(torch.zeros(8)))   # K % 4 != 0
    with pytest.raises(ConformerHipError):
        ops.relpos_attention(G(torch.zeros(1, 4, 3 * 128)), G(torch.zeros(7, 128)), G(torch.zeros(1, 128)),
                             G(torch.zeros(1, 128)), None, 1)                        # dh = 128 unsupported


def test_relpos_attention_forms_agree(ops):
    """The three forms of the fp32 attention forward (4 waves; 8 waves half a key tile apart = the default above 128 query rows;
    8 waves software-pipelined) on one input with ragged lengths: the pipelined form shares the 4-wave arithmetic bit for bit, the
    staggered form sums its scores in two chains (<= 1e-6)."""
    from conformer_amd import _lib
    lib = _lib.load()
    B, T, H, dh = 3, 249, 4, 64
    d = H * dh
    qkv, pos = rnd(B, T, 3 * d, seed=21), rnd(2 * T - 1, d, seed=22)
    u, v = rnd(H, dh, seed=23, scale=0.3), rnd(H, dh, seed=24, scale=0.3)
    L = torch.tensor([249, 131, 17], dtype=torch.int64)
    outs = {}
    try:
        for nw in (4, 8, 9):
            lib.cfm_debug_set_attention_waves(nw)
            outs[nw] = ops.relpos_attention(G(qkv), G(pos), G(u), G(v), G(L), H).clone()
    finally:
        lib.cfm_debug_set_attention_waves(0)
    assert torch.equal(outs[9], outs[4])
    assert rel_l2(outs[8], outs[4].double()) < 1e-6
    q, k, vv = (t.reshape(B, T, H, dh).double() for t in qkv.split(d, dim=-1))
    ref = O.relpos_attention_core(q, k, vv, pos.double().view(2 * T - 1, H, dh), u.double(), v.double(), L)
    assert rel_l2(outs[8], ref) < TOL
