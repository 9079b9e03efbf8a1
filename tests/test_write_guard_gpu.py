"""Over-WRITE guard (the round-2 fault: an fp32 kernel wrote into a buffer allocated with a 16-bit
input's dtype).  The C ABI takes raw pointers without byte sizes, so an output buffer that is too small is invisible at the
boundary and the NaN guard bands of test_guard_gpu.py (over-READS) do not see it.

Here EVERY buffer the host layer allocates while the product code runs -- activations, gradients, workspaces, caches -- is
carved out of a larger allocation whose bytes on both sides are a sentinel pattern; after the step the sentinels must be
intact.  The hook is test-only: the modules' global name `torch` is replaced by a proxy whose empty / zeros / full /
empty_like / zeros_like allocate guarded storage (no product code changes, nothing of it runs in production).
One pass per configuration, as after any change.
"""
import contextlib
import math

import pytest
import torch

from oracle import conformer_oracle as O
from tests.util import rel_l2

pytestmark = pytest.mark.gpu
PAD = 1 << 16               # sentinel bytes on each side of every allocation
SENT = 0x5A


class GuardedTorch:
    """Stands in for the `torch` module inside the patched modules: allocation functions hand out guarded storage."""

    def __init__(self):
        self.allocs = []

    def __getattr__(self, name):
        return getattr(torch, name)

    def _carve(self, shape, dtype, device, fill):
        dtype = dtype or torch.get_default_dtype()
        device = torch.device(device) if device is not None else torch.device("cpu")
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        shape = tuple(int(s) for s in shape)
        if device.type != "cuda":
            return torch.full(shape, 0 if fill is None else fill, dtype=dtype, device=device)
        nbytes = int(math.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        body = (nbytes + 255) // 256 * 256
        raw = torch.full((2 * PAD + body,), SENT, dtype=torch.uint8, device=device)
        view = raw[PAD:PAD + nbytes].view(dtype).view(shape)
        if fill is not None:
            view.fill_(fill)
        self.allocs.append((raw, nbytes))
        return view

    def empty(self, *shape, dtype=None, device=None, **kw):
        return self._carve(shape, dtype, device, None)

    def zeros(self, *shape, dtype=None, device=None, **kw):
        return self._carve(shape, dtype, device, 0)

    def full(self, shape, value, dtype=None, device=None, **kw):
        return self._carve((shape,), dtype, device, value)

    def empty_like(self, t, dtype=None, **kw):
        return self._carve(tuple(t.shape), dtype or t.dtype, t.device, None)

    def zeros_like(self, t, dtype=None, **kw):
        return self._carve(tuple(t.shape), dtype or t.dtype, t.device, 0)

    def check(self):
        torch.cuda.synchronize()
        bad = []
        for i, (raw, nbytes) in enumerate(self.allocs):
            head_ok = bool((raw[:PAD] == SENT).all())
            tail_ok = bool((raw[PAD + nbytes:] == SENT).all())
            if not (head_ok and tail_ok):
                bad.append((i, nbytes, head_ok, tail_ok))
        return bad


@contextlib.contextmanager
def guarded_allocations():
    import conformer_amd.autograd as ag
    import conformer_amd.decode as decode
    import conformer_amd.evaluation as evaluation
    import conformer_amd.frontend as frontend
    import conformer_amd.ops as ops
    import conformer_amd.optim as optim
    import conformer_amd.streaming as streaming
    mods = [ops, ag, decode, evaluation, frontend, optim, streaming]
    proxy = GuardedTorch()
    saved = [(m, m.torch) for m in mods if hasattr(m, "torch")]
    chunk, arena = ops._ZERO_CHUNK, dict(ops._ZERO_ARENA)
    ops._ZERO_CHUNK = 0                      # every accumulate-with-atomics output gets its own (guarded) fill
    ops._ZERO_ARENA.clear()
    split = ops._zeros_split
    ops._zeros_split = lambda device, dtype, *shapes: [proxy.zeros(tuple(torch.Size(sh)), dtype=dtype, device=device) for sh in shapes]
    try:
        for m, _ in saved:
            m.torch = proxy
        yield proxy
    finally:
        for m, t in saved:
            m.torch = t
        ops._ZERO_CHUNK, ops._zeros_split = chunk, split
        ops._ZERO_ARENA.clear()
        ops._ZERO_ARENA.update(arena)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _model(d, heads, n_blocks, dev, dropout=0.0):
    from model.conformer import Conformer
    P = O.make_params(vocab=29, n_mel=80, n_blocks=n_blocks, d=d, n_heads=heads, ksize=31, lstm_hidden=48, seed=7)
    m = Conformer(29, 80, n_blocks, d, heads, 31, 48, 1, dropout)
    m.load_state_dict(P, strict=True)
    return m.to(dev)


@pytest.mark.parametrize("amp", [None, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d,heads", [(64, 4), (144, 4)])
def test_training_step_writes_stay_inside_their_buffers(dev, amp, d, heads):
    """Forward + CTC + backward + FusedAdam of a small Conformer (train-mode BatchNorm, dropout 0.1, ragged lengths): stem
    backward, both flash attention backwards, the weight-gradient kernels, LayerNorm / depthwise-conv / GLU backward, LSTM,
    CTC lattices, the optimiser -- every buffer they write is guarded.  d=64: the all-16-bit forms; d=144: the mixed ones."""
    from conformer_amd.evaluation import ConformerCriterion
    from conformer_amd.optim import FusedAdam
    model = _model(d, heads, 2, dev, dropout=0.1).train()
    g = torch.Generator().manual_seed(3)
    B, T = 3, 135
    x = torch.randn(B, 80, T, generator=g).to(dev)
    lengths = torch.tensor([135, 120, 77], device=dev)
    targets = torch.randint(1, 29, (B, 6), generator=g).to(dev)
    tlen = torch.tensor([6, 5, 3], device=dev)
    crit = ConformerCriterion(blank_id=0)
    with guarded_allocations() as gt:
        opt = FusedAdam(model.parameters(), lr=1e-4)
        for _ in range(2):
            with (torch.autocast("cuda", dtype=amp) if amp else contextlib.nullcontext()):
                logits, out_len = model(x, lengths)
                with torch.autocast("cuda", enabled=False):
                    loss = crit.ctc_loss(logits.float(), targets, out_len, tlen)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        bad = gt.check()
        n = len(gt.allocs)
    assert n > 100, f"the hook saw only {n} allocations: the proxy is not in the path"
    assert not bad, f"{len(bad)} of {n} buffers were written outside their bounds: {bad[:5]}"
    assert math.isfinite(float(loss))


@pytest.mark.parametrize("amp", [None, torch.bfloat16])
@pytest.mark.parametrize("d,heads,T", [(64, 4, 135), (512, 8, 203), (144, 4, 90)])
def test_inference_writes_stay_inside_their_buffers(dev, amp, d, heads, T):
    """Eval forward (folded-LayerNorm path at d = 64 / 512, the 16-bit q|k|v / context / stem activations under autocast),
    greedy decode, and the streaming encoder's chunk step with its caches."""
    from conformer_amd.decode import greedy_ctc_decode
    model = _model(d, heads, 2, dev).eval()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 80, T, generator=g).to(dev)
    lengths = torch.tensor([T, T - 29], device=dev)
    with torch.no_grad():
        with (torch.autocast("cuda", dtype=amp) if amp else contextlib.nullcontext()):
            ref, _ = model(x, lengths)
        with guarded_allocations() as gt:
            with (torch.autocast("cuda", dtype=amp) if amp else contextlib.nullcontext()):
                logits, out_len = model(x, lengths)
            greedy_ctc_decode(logits.float(), 0, 1, out_len)
            from conformer_amd.streaming import StreamingEncoder
            se = StreamingEncoder(model.encoder, 2, T)
            for a in range(0, T, 64):                                  # (under autocast: the 16-bit incremental attention)
                with (torch.autocast("cuda", dtype=amp) if amp else contextlib.nullcontext()):
                    se.step(x[:, :, a:min(a + 64, T)])
            bad = gt.check()
            n = len(gt.allocs)
    assert n > 30 and not bad, f"{len(bad)} of {n} buffers were written outside their bounds: {bad[:5]}"
    assert rel_l2(logits, ref) < (1e-6 if amp is None else 2e-2)


@pytest.mark.parametrize("M,d", [(333, 256), (97, 512), (1, 128)])
def test_row_block_kernels_write_inside_their_buffers(dev, M, d):
    """The one-kernel feed-forward and the three row chains at ragged row counts (the last workgroup holds 13 / 1 / 1 live rows of 32):
    outputs, statistics partials, q|k|v and the stored intermediate rows are guarded."""
    from conformer_amd import ops
    g = torch.Generator().manual_seed(8)
    R = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    x, ctx, c = R(M, d) + 0.3, R(M, d), R(M, d)
    lw, lb = 1 + 0.1 * R(d), 0.1 * R(d)
    w1, b1, w2, b2 = R(4 * d, d) / math.sqrt(d), 0.1 * R(4 * d), R(d, 4 * d) / math.sqrt(4 * d), 0.1 * R(d)
    wq, bq, wo, bo, wg, bg = R(3 * d, d) / math.sqrt(d), 0.1 * R(3 * d), R(d, d) / math.sqrt(d), 0.1 * R(d), R(2 * d, d) / math.sqrt(d), 0.1 * R(2 * d)
    xs = x.view(M, d // 32, 32)
    st = torch.stack([xs.sum(-1), ((xs - xs.mean(-1, keepdim=True)) ** 2).sum(-1)], dim=-1).contiguous()
    with guarded_allocations() as gt:
        wf, bf, cs = ops.fold_layernorm(w1, b1, lw, lb)
        ffn = (ops.ffn_pack(wf, w2), bf, cs)
        ops.ffn_fused(x, st, ffn[0], bf, cs, b2, 0.5, 1e-5)
        ops.ffn_fused(x, st, ffn[0], bf, cs, b2, 0.5, 1e-5, emit_stats=True)
        ops.ffn_fused(x, st, ffn[0], bf, cs, b2, 0.5, 1e-5, emit_stats=True, closing_ln=(lw, lb, 1e-5))
        wqf, bqf, csq = ops.fold_layernorm(wq, bq, lw, lb)
        ops.rowchain_ffn_qkv(x, st, ffn, b2, 0.5, 1e-5, ops.rowgemm_pack(wqf), bqf, csq, 1e-5)
        wgf, bgf, csg = ops.fold_layernorm(wg, bg, lw, lb)
        ops.rowchain_out_glu(ctx, ops.rowgemm_pack(wo), bo, x, ops.rowgemm_pack(wgf, glu=True), bgf, csg, 1e-5)
        ops.rowchain_pw2_ffn_ln(c, ops.rowgemm_pack(wo), bo, x, ffn, b2, 0.5, 1e-5, (lw, lb, 1e-5), want_stats=True)
        bad = gt.check()
        n = len(gt.allocs)
    assert n >= 15 and not bad, f"{len(bad)} of {n} buffers were written outside their bounds: {bad[:5]}"


def test_big_batch_block_writes_stay_inside_their_buffers(dev):
    """One Conformer-L block at cfg-2 geometry (7968 rows: the one-kernel feed-forward path of ConformerBlock.fused_chain, then the
    five-launch row-chain form): every buffer guarded."""
    from conformer_amd import ops
    from conformer_amd.model.utils.block import ConformerBlock
    torch.manual_seed(3)
    d, B, T = 512, 32, 249
    blk = ConformerBlock(d, 8, 31).to(dev).eval()
    x = torch.randn(B, T, d, device=dev) + 0.2
    xs = x.view(B * T, d // 32, 32)
    st = torch.stack([xs.sum(-1), ((xs - xs.mean(-1, keepdim=True)) ** 2).sum(-1)], dim=-1).contiguous()
    table = ops.relpos_table(torch.exp(torch.arange(0, d, 2, device=dev) * -(math.log(10000.0) / d))[None], T)
    L = torch.full((B,), T, dtype=torch.int64, device=dev)
    with torch.no_grad(), guarded_allocations() as gt:
        import conformer_amd.model.utils.block as blockmod                 # (the block allocates nothing itself: ops does)
        assert ops.ffn_fused_ok(d, 4 * d, B * T)
        a, _ = blk.fused_chain(x, table, L, x_stats=st, want_stats=True)
        prev = ops.set_rowchain(True)
        try:
            b, _ = blk.fused_chain(x, table, L, x_stats=st, want_stats=True)
        finally:
            ops.set_rowchain(prev)
        bad = gt.check()
        n = len(gt.allocs)
    assert n >= 20 and not bad, f"{len(bad)} of {n} buffers were written outside their bounds: {bad[:5]}"
    assert rel_l2(b, a) < 1e-5
