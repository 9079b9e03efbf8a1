"""N2 fused Adam and N4 greedy CTC decode on the device vs their oracle restatements."""
import pytest
import torch

from oracle import conformer_oracle as O
from tests.util import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def test_fused_adam_matches_torch_adam(dev):
    from conformer_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(512, 2048), (17,), (3, 5, 7), (4099,), (1,), (64, 1, 31)]
    params = [torch.randn(*s, generator=g) for s in shapes]
    grads = [[torch.randn(*s, generator=g) * (0.1 + k) for s in shapes] for k in range(4)]
    ref = O.adam_reference(params, grads, lr=1e-2)
    ps = [torch.nn.Parameter(p.to(dev)) for p in params]
    opt = FusedAdam(ps, lr=1e-2)
    for k in range(4):
        for p, gr in zip(ps, grads[k]):
            p.grad = gr.to(dev)
        versions = [p._version for p in ps]
        opt.step()
        assert all(p._version > v for p, v in zip(ps, versions))        # caches keyed on _version must see the update
    for p, r in zip(ps, ref):
        assert rel_l2(p, r) < 2e-6
    # state layout interchanges with torch.optim.Adam (manager.py:34-40 saves optimizer.state_dict())
    sd = opt.state_dict()
    t_opt = torch.optim.Adam(ps, lr=1e-2)
    t_opt.load_state_dict(sd)
    assert int(t_opt.state[ps[0]]["step"]) == 4
    for p, gr in zip(ps, grads[0]):
        p.grad = gr.to(dev)
    before = [p.detach().clone() for p in ps]
    t_opt.step()                                           # stock Adam continues from the fused optimizer's state
    assert all(torch.isfinite(p).all() and not torch.equal(p, b) for p, b in zip(ps, before))


@pytest.mark.parametrize("B,T,V", [(1, 1, 5), (3, 49, 370), (2, 249, 370), (4, 25, 17)])
def test_greedy_decode_bit_exact(dev, B, T, V):
    from conformer_amd.decode import greedy_ctc_decode, tokens_to_text
    g = torch.Generator().manual_seed(B * 1000 + T)
    logits = torch.randn(B, T, V, generator=g)
    # long runs, pads between repeats, exact ties
    logits[:, : T // 2, 3 % V] += 6.0
    logits[:, ::5, 0] += 9.0
    if T > 4:
        logits[0, 4, :] = 0.0
        logits[0, 4, [V - 1, 2 % V]] = 5.0                   # tie -> lowest index
    pad_id, unk_id = 0, 1 % V
    frame_ids, tokens, counts = greedy_ctc_decode(logits.to(dev), pad_id, unk_id)
    for b in range(B):
        ids, out = O.greedy_decode_ids(logits[b], pad_id, unk_id)
        assert frame_ids[b].tolist() == ids
        n = int(counts[b])
        assert tokens[b, :n].tolist() == out and (tokens[b, n:] == -1).all()
    L = torch.tensor([max(1, T - 3 * b) for b in range(B)])
    _, tok2, cnt2 = greedy_ctc_decode(logits.to(dev), pad_id, unk_id, L.to(dev))
    for b in range(B):
        _, out = O.greedy_decode_ids(logits[b, : int(L[b])], pad_id, unk_id)
        assert tok2[b, : int(cnt2[b])].tolist() == out
    vocab = [f"<{i}>" for i in range(V)]
    assert len(tokens_to_text(tokens, counts, vocab)) == B


@pytest.mark.parametrize("B,T,D,H,ragged", [(1, 1, 16, 8, False), (3, 25, 32, 8, True), (2, 49, 144, 320, True),
                                            (32, 60, 512, 640, True), (40, 12, 64, 36, True), (70, 7, 32, 16, False)])
def test_lstm_recurrence_vs_oracle(dev, B, T, D, H, ragged):
    """N1: the gfx950 LSTM layer (GEMM + one launch per frame) vs the written-out float64 oracle, packed semantics."""
    from conformer_amd import ops
    g = torch.Generator().manual_seed(B * 100 + T)
    x = torch.randn(B, T, D, generator=g)
    k = 1.0 / H ** 0.5
    w_ih, w_hh = (torch.rand(4 * H, D, generator=g) * 2 - 1) * k, (torch.rand(4 * H, H, generator=g) * 2 - 1) * k
    b_ih, b_hh = (torch.rand(4 * H, generator=g) * 2 - 1) * k, (torch.rand(4 * H, generator=g) * 2 - 1) * k
    L = None
    if ragged:
        L = torch.sort(torch.randint(1, T + 1, (B,), generator=g), descending=True).values
        L[0] = T
    ref = O.lstm_layer(x.double(), L, w_ih.double(), w_hh.double(), b_ih.double(), b_hh.double())
    y, gates, cells = ops.lstm_forward(x.to(dev), w_ih.to(dev), w_hh.to(dev), (b_ih + b_hh).to(dev),
                                       None if L is None else L.to(dev), save=True)
    assert rel_l2(y, ref) < 5e-6
    if L is not None:
        for b in range(B):
            assert (y[b, int(L[b]):] == 0).all()
    assert torch.isfinite(gates).all() and torch.isfinite(cells).all()


def test_decoder_eval_runs_on_hip_and_matches_oracle(dev):
    """Decoder.forward in eval / no_grad: LSTM + Swish + BatchNorm(eval) + Linear on the gfx950 kernels vs the oracle."""
    from model.modules.decoder import Decoder
    P = O.make_params(vocab=370, n_mel=80, n_blocks=0, d=144, n_heads=4, ksize=31, lstm_hidden=320, seed=3)
    dec = Decoder(370, 144, 320, 1)
    dec.load_state_dict({k[len("decoder."):]: v for k, v in P.items() if k.startswith("decoder.")}, strict=True)
    dec = dec.to(dev).eval()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 49, 144, generator=g)
    L = torch.tensor([49, 30, 5])
    Pd = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    ref = O.decoder_forward(x.double(), L, Pd)
    with torch.no_grad():
        assert dec._hip_eligible(x.to(dev))
        y = dec(x.to(dev), L.to(dev))
    assert rel_l2(y, ref) < 2e-5
    assert torch.equal(y.argmax(-1).cpu(), ref.argmax(-1))
    # the stock-module path (CPU tensors) computes the same function
    cpu = Decoder(370, 144, 320, 1).eval()
    cpu.load_state_dict(dec.state_dict())
    with torch.no_grad():
        assert rel_l2(cpu(x, L), ref) < 2e-5


@pytest.mark.parametrize("B,T,D,H,ragged", [(1, 1, 16, 8, False), (3, 25, 32, 8, True), (2, 49, 144, 320, True),
                                            (32, 30, 64, 640, True), (40, 12, 64, 36, True), (70, 7, 32, 16, False)])
def test_lstm_backward_vs_oracle_autograd(dev, B, T, D, H, ragged):
    """N1 training: BPTT kernels + GEMMs vs torch autograd through the float64 written-out LSTM."""
    from conformer_amd.autograd import LstmFn
    g = torch.Generator().manual_seed(B * 100 + T + 1)
    x = torch.randn(B, T, D, generator=g)
    k = 1.0 / H ** 0.5
    ps = [(torch.rand(4 * H, D, generator=g) * 2 - 1) * k, (torch.rand(4 * H, H, generator=g) * 2 - 1) * k,
          (torch.rand(4 * H, generator=g) * 2 - 1) * k, (torch.rand(4 * H, generator=g) * 2 - 1) * k]
    L = None
    if ragged:
        L = torch.sort(torch.randint(1, T + 1, (B,), generator=g), descending=True).values
        L[0] = T
    w = torch.randn(B, T, H, generator=g)
    xr = x.double().requires_grad_(True)
    pr = [p.double().requires_grad_(True) for p in ps]
    (O.lstm_layer(xr, L, *pr) * w.double()).sum().backward()
    xd = x.to(dev).requires_grad_(True)
    pd = [p.to(dev).requires_grad_(True) for p in ps]
    y = LstmFn.apply(xd, *pd, None if L is None else L.to(dev))
    (y * w.to(dev)).sum().backward()
    assert rel_l2(xd.grad, xr.grad) < 2e-5
    for a, b, name in zip(pd, pr, ("w_ih", "w_hh", "b_ih", "b_hh")):
        assert rel_l2(a.grad, b.grad) < 2e-5, name


@pytest.mark.parametrize("B,T,H", [(40, 9, 48), (3, 25, 16), (64, 6, 640)])
def test_lstm_fragment_order_kernels_are_bit_identical_to_row_major(dev, B, T, H):
    """The fragment-order fp32 recurrence / BPTT kernels do the same products in the same order as the row-major ones:
    outputs, saved activations and gate gradients must be EQUAL, ragged lengths included, next to poisoned scratch."""
    from conformer_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(B + T + H)
    gx = torch.randn(B, T, 4 * H, generator=g).to(dev)
    w_hh = ((torch.rand(4 * H, H, generator=g) * 2 - 1) / H ** 0.5).to(dev)
    L = torch.sort(torch.randint(1, T + 1, (B,), generator=g), descending=True).values
    L[0] = T
    L = L.to(dev)
    st = torch.cuda.current_stream().cuda_stream
    nb = (B + 15) // 16 * 16

    def fwd(frag):
        y, c = torch.full((B, T, H), float("nan"), device=dev), torch.full((B, H), float("nan"), device=dev)
        gates, cells = torch.full((B, T, 4 * H), float("nan"), device=dev), torch.full((B, T, H), float("nan"), device=dev)
        if frag:
            wf = w_hh.view(4, H // 4, 4, H // 16, 4, 4).permute(1, 3, 4, 0, 2, 5).contiguous()
            hf = torch.full((2 * nb * H,), float("nan"), device=dev)
            rc = lib.cfm_lstm_fwd_frag_f32(gx.data_ptr(), wf.data_ptr(), L.data_ptr(), y.data_ptr(), c.data_ptr(), hf.data_ptr(),
                                           gates.data_ptr(), cells.data_ptr(), B, T, H, st)
        else:
            rc = lib.cfm_lstm_fwd_f32(gx.data_ptr(), w_hh.data_ptr(), L.data_ptr(), y.data_ptr(), c.data_ptr(), gates.data_ptr(),
                                      cells.data_ptr(), B, T, H, st)
        assert rc == 0
        return y, gates, cells

    y0, g0, c0 = fwd(False)
    y1, g1, c1 = fwd(True)
    assert torch.isfinite(y0).all()
    assert torch.equal(y0, y1) and torch.equal(g0, g1) and torch.equal(c0, c1)
    dy = torch.randn(B, T, H, generator=g).to(dev)

    def bwd(frag):
        dG, dc = torch.full((B, T, 4 * H), float("nan"), device=dev), torch.full((B, H), float("nan"), device=dev)
        if frag:
            wtf = w_hh.t().reshape(H // 16, 16, 4 * H // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous()
            dgf = torch.full((2 * nb * 4 * H,), float("nan"), device=dev)
            rc = lib.cfm_lstm_bwd_frag_f32(dy.data_ptr(), g0.data_ptr(), c0.data_ptr(), wtf.data_ptr(), L.data_ptr(), dG.data_ptr(),
                                           dc.data_ptr(), dgf.data_ptr(), B, T, H, st)
        else:
            wt = w_hh.t().contiguous()
            rc = lib.cfm_lstm_bwd_f32(dy.data_ptr(), g0.data_ptr(), c0.data_ptr(), wt.data_ptr(), L.data_ptr(), dG.data_ptr(),
                                      dc.data_ptr(), B, T, H, st)
        assert rc == 0
        return dG

    d0, d1 = bwd(False), bwd(True)
    assert torch.isfinite(d0).all() and torch.equal(d0, d1)
    # shapes the fragment kernels cannot take are refused, not mis-indexed
    assert lib.cfm_lstm_fwd_frag_f32(gx.data_ptr(), w_hh.data_ptr(), None, y0.data_ptr(), c0.data_ptr(), y1.data_ptr(), None, None,
                                     B, T, 8, st) != 0


@pytest.mark.parametrize("B,T,D,H,ragged", [(3, 25, 32, 16, True), (2, 49, 144, 320, True), (64, 30, 64, 640, True), (40, 12, 64, 48, True)])
def test_lstm_bf16_recurrence_and_bptt_vs_oracle(dev, B, T, D, H, ragged):
    """Under autocast the forward recurrent product runs on the 16-bit matrix pipe (lstm_mfma16.hip: 32-utterance x 8-unit
    workgroups, 16-bit W_hh and h exchange; BPTT on the fp32 kernel): forward and every gradient within the north_star's
    1e-2 of the float64 oracle."""
    from conformer_amd.autograd import LstmFn
    g = torch.Generator().manual_seed(B * 100 + T + 7)
    x = torch.randn(B, T, D, generator=g)
    k = 1.0 / H ** 0.5
    ps = [(torch.rand(4 * H, D, generator=g) * 2 - 1) * k, (torch.rand(4 * H, H, generator=g) * 2 - 1) * k,
          (torch.rand(4 * H, generator=g) * 2 - 1) * k, (torch.rand(4 * H, generator=g) * 2 - 1) * k]
    L = None
    if ragged:
        L = torch.sort(torch.randint(1, T + 1, (B,), generator=g), descending=True).values
        L[0] = T
    w = torch.randn(B, T, H, generator=g)
    xr = x.double().requires_grad_(True)
    pr = [p.double().requires_grad_(True) for p in ps]
    ref = O.lstm_layer(xr, L, *pr)
    (ref * w.double()).sum().backward()
    xd = x.to(dev).requires_grad_(True)
    pd = [p.to(dev).requires_grad_(True) for p in ps]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = LstmFn.apply(xd, *pd, None if L is None else L.to(dev))
    (y.float() * w.to(dev)).sum().backward()
    assert y.dtype == torch.float32 and rel_l2(y, ref.detach()) < 1e-2
    if L is not None:
        for b in range(B):
            assert (y[b, int(L[b]):] == 0).all()
    assert rel_l2(xd.grad, xr.grad) < 1e-2
    for a, b, name in zip(pd, pr, ("w_ih", "w_hh", "b_ih", "b_hh")):
        assert rel_l2(a.grad, b.grad) < 1e-2, name


@pytest.mark.parametrize("train_bn", [False, True])
def test_decoder_training_grads_vs_oracle_autograd(dev, train_bn):
    """N1 training: the whole decoder (LSTM -> Swish -> BatchNorm -> Linear(370)) forward + backward on the gfx950 kernels
    vs torch autograd through the float64 oracle (BatchNorm with running or with batch statistics)."""
    from model.modules.decoder import Decoder
    P = O.make_params(vocab=370, n_mel=80, n_blocks=0, d=64, n_heads=4, ksize=31, lstm_hidden=96, seed=4)
    dec = Decoder(370, 64, 96, 1)
    dec.load_state_dict({k[len("decoder."):]: v for k, v in P.items() if k.startswith("decoder.")}, strict=True)
    dec = dec.to(dev).train()
    if not train_bn:
        dec.norm.eval()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(5, 21, 64, generator=g)
    L = torch.tensor([21, 21, 13, 8, 2])
    w = torch.randn(5, 21, 370, generator=g)
    # float64 oracle with autograd (train-mode BatchNorm written out: batch statistics over all B*T rows)
    Pd = {k: (v.double().requires_grad_(True) if v.is_floating_point() else v) for k, v in P.items() if k.startswith("decoder.")}
    xr = x.double().requires_grad_(True)
    pre = "decoder."
    y = O.lstm_layer(xr, L, Pd[pre + "lstm.weight_ih_l0"], Pd[pre + "lstm.weight_hh_l0"], Pd[pre + "lstm.bias_ih_l0"],
                     Pd[pre + "lstm.bias_hh_l0"])
    s = O.swish(y)
    if train_bn:
        flat = s.reshape(-1, s.shape[-1])
        mean, var = flat.mean(0), flat.var(0, unbiased=False)
    else:
        mean, var = Pd[pre + "norm.running_mean"], Pd[pre + "norm.running_var"]
    z = (s - mean) / torch.sqrt(var + 1e-5) * Pd[pre + "norm.weight"] + Pd[pre + "norm.bias"]
    ref = z @ Pd[pre + "linear.weight"].t() + Pd[pre + "linear.bias"]
    (ref * w.double()).sum().backward()
    rm0 = dec.norm.running_mean.clone()
    nbt0 = int(dec.norm.num_batches_tracked)
    xd = x.to(dev).requires_grad_(True)
    out = dec(xd, L.to(dev))
    (out * w.to(dev)).sum().backward()
    assert rel_l2(out, ref) < 2e-5
    assert rel_l2(xd.grad, xr.grad) < 5e-5
    for name, p in dec.named_parameters():
        assert rel_l2(p.grad, Pd[pre + name].grad) < 5e-5, name
    if train_bn:                                             # running statistics moved (momentum 0.1, unbiased variance)
        n = 5 * 21
        assert rel_l2(dec.norm.running_mean, 0.9 * rm0.cpu().double() + 0.1 * mean.detach()) < 1e-5
        assert rel_l2(dec.norm.running_var, 0.9 * P[pre + "norm.running_var"].double() + 0.1 * var.detach() * n / (n - 1)) < 1e-5
        assert int(dec.norm.num_batches_tracked) == nbt0 + 1
    else:
        assert torch.equal(dec.norm.running_mean, rm0)


def test_weight_caches_follow_the_fused_optimizer(dev):
    """The cached 16-bit weight copy (ops.weight16) and the modules' packed weights are keyed on the parameter version:
    a FusedAdam step (raw-pointer kernel) must invalidate them."""
    from conformer_amd import ops
    from conformer_amd.optim import FusedAdam
    w = torch.nn.Parameter(torch.randn(64, 64, device=dev))
    a, b = torch.randn(8, 64, device=dev), torch.zeros(64, device=dev)
    opt = FusedAdam([w], lr=0.5)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y0 = ops.linear(a, w.detach(), b)
        w.grad = torch.ones_like(w)
        opt.step()
        y1 = ops.linear(a, w.detach(), b)
    ref = a.double().cpu() @ w.detach().double().cpu().t()
    assert rel_l2(y1, ref) < 1e-2 and rel_l2(y0, ref) > 1e-1
    # a temporary weight whose address is recycled by the allocator must not hit the stale entry of its predecessor
    with torch.autocast("cuda", dtype=torch.bfloat16):
        for k in range(4):
            tmp = torch.full((64, 64), float(k + 1), device=dev)
            y = ops.linear(a, tmp, b)
            assert rel_l2(y, (k + 1.0) * a.double().cpu().sum(-1, keepdim=True).expand(8, 64)) < 1e-2, k
            del tmp, y


@pytest.mark.parametrize("amp", [None, torch.bfloat16])
def test_training_trajectory_fused_adam_equals_torch_adam(dev, amp):
    """Five full training steps of a tiny Conformer (fwd + CTC + bwd + optimizer) with FusedAdam vs torch.optim.Adam: the
    loss trajectories agree (weights are not compared: parameters with mathematically zero gradients -- key/pos projection
    biases, biases in front of a train-mode BatchNorm -- get Adam-amplified summation noise) -- guards every weight-derived cache (fused QKV, packed conv / linear
    weights, 16-bit weight copies) against going stale when the optimizer writes parameters through raw pointers."""
    from conformer_amd.optim import FusedAdam
    from model.conformer import Conformer
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 80, 120, generator=g).to(dev)
    L = torch.tensor([120, 101, 77], device=dev)
    tg = torch.randint(1, 30, (3, 6), generator=g).to(dev)
    tl = torch.tensor([6, 5, 3], device=dev)
    crit = torch.nn.CTCLoss(blank=0, zero_infinity=True)
    runs = {}
    for name in ("torch", "fused"):
        torch.manual_seed(1)
        m = Conformer(30, 80, 2, 64, 4, 7, 32, 1, 0.0).to(dev).train()
        opt = (FusedAdam if name == "fused" else torch.optim.Adam)(m.parameters(), lr=2e-3)
        losses = []
        for _ in range(5):
            with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
                logits, out_len = m(x, L)
            loss = crit(logits.float().log_softmax(-1).transpose(0, 1), tg, out_len, tl)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        runs[name] = losses
    # (run-to-run: split-K atomics reorder sums; Adam turns that noise into small trajectory differences)
    tol = 1e-3 if amp is None else 5e-2
    assert runs["torch"][4] < runs["torch"][0]                             # it trains
    for a, b in zip(runs["torch"], runs["fused"]):
        assert abs(a - b) < tol * abs(a), (runs["torch"], runs["fused"])


# ---- N1 loss: CTC lattice kernels vs the written-out float64 lattice (itself pinned against F.ctc_loss on the CPU) ----------
def _ctc_inputs(B, T, V, L, in_len, tgt_len, seed, scale=2.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, V, generator=g) * scale
    tg = torch.randint(1, V, (B, L), generator=g)
    tg[0, :min(4, L)] = torch.tensor([2, 2, 2, 3])[:min(4, L)]          # repeated labels in every case
    return x, tg, torch.tensor(in_len), torch.tensor(tgt_len)


CTC_GPU_CASES = {
    "cfg1": (2, 49, 370, 12, [49, 40], [12, 9]),
    "ragged_empty_target": (3, 20, 11, 6, [20, 7, 20], [6, 6, 0]),
    "infeasible": (2, 8, 7, 4, [8, 3], [4, 4]),
    "one_frame": (2, 1, 9, 1, [1, 1], [1, 0]),
    "zero_frames": (2, 6, 9, 2, [0, 6], [2, 2]),
    "two_pairs_per_lane": (2, 260, 40, 100, [260, 150], [100, 64]),
    "eight_pairs_per_lane": (2, 700, 30, 300, [700, 650], [300, 257]),
    "sixteen_pairs_per_lane": (1, 1400, 30, 600, [1400], [600]),
}


@pytest.mark.parametrize("name", list(CTC_GPU_CASES))
def test_ctc_loss_matches_lattice_oracle(dev, name):
    from conformer_amd.evaluation import ConformerCriterion
    B, T, V, L, il, tl = CTC_GPU_CASES[name]
    x, tg, ilt, tlt = _ctc_inputs(B, T, V, L, il, tl, seed=3)
    loss_o, nll_o, grad_o = O.ctc_lattice(x, tg, ilt, tlt)
    xd = x.to(dev).requires_grad_()
    crit = ConformerCriterion(blank_id=0)
    loss = crit.ctc_loss(xd, tg.float().to(dev), ilt.to(dev), tlt.to(dev))     # float targets, as evaluation.py:14
    (loss * 3.0).backward()
    err = rel_l2(xd.grad, 3.0 * grad_o) if grad_o.abs().sum() > 0 else float(xd.grad.abs().max())
    print(f"ctc[{name}]: loss {float(loss):.6f} vs {loss_o:.6f}, grad rel-L2 {err:.2e}")
    assert abs(float(loss) - loss_o) <= 1e-5 * max(1.0, abs(loss_o))
    assert err < 1e-4                # re-centred fp32 lattice: independent of the utterance length (1e-3 is the bar)
    assert torch.isfinite(xd.grad).all()
    # frames beyond an utterance's length and utterances without a valid alignment get exactly zero gradient
    for b in range(B):
        assert not xd.grad[b, il[b]:].any()
        if not torch.isfinite(nll_o[b]):
            assert not xd.grad[b].any()
    # 1-D concatenated targets give the same numbers
    cat = torch.cat([tg[b, :tl[b]] for b in range(B)])
    if cat.numel():
        x2 = x.to(dev).requires_grad_()
        loss2 = crit.ctc_loss(x2, cat.to(dev), ilt.to(dev), tlt.to(dev))
        loss2.backward()
        assert float(loss2) == float(loss)
        assert torch.equal(x2.grad * 3.0, xd.grad) or rel_l2(x2.grad * 3.0, xd.grad) < 1e-6


def test_ctc_loss_training_geometry_matches_torch(dev):
    """cfg-3 geometry (B=64, T'=249, V=370, 40 labels): against the lattice oracle AND torch's own CTC on the same device."""
    from conformer_amd.evaluation import ConformerCriterion
    B, T, V, L = 64, 249, 370, 40
    il = sorted([249 - 3 * (i % 40) for i in range(B)], reverse=True)
    tl = [40 - (i % 7) for i in range(B)]
    x, tg, ilt, tlt = _ctc_inputs(B, T, V, L, il, tl, seed=5, scale=1.0)
    loss_o, _, grad_o = O.ctc_lattice(x, tg, ilt, tlt)
    xd = x.to(dev).requires_grad_()
    loss = ConformerCriterion(0).ctc_loss(xd, tg.to(dev), ilt.to(dev), tlt.to(dev))
    loss.backward()
    assert abs(float(loss) - loss_o) <= 1e-5 * abs(loss_o)
    assert rel_l2(xd.grad, grad_o) < 1e-4
    xt = x.to(dev).requires_grad_()
    lt = torch.nn.functional.ctc_loss(xt.log_softmax(-1).transpose(0, 1), tg.to(dev), ilt.to(dev), tlt.to(dev), blank=0,
                                      zero_infinity=True)
    lt.backward()
    assert abs(float(loss) - float(lt)) <= 1e-5 * abs(float(lt))
    assert rel_l2(xd.grad, xt.grad) < 1e-3      # torch's un-centred fp32 lattice is the looser side (2e-4 from the oracle)


def test_ctc_loss_rejects_cpu_and_long_targets(dev):
    from conformer_amd.evaluation import ConformerCriterion
    crit = ConformerCriterion(0)
    with pytest.raises(RuntimeError):
        crit.ctc_loss(torch.randn(1, 4, 5), torch.ones(1, 2), torch.tensor([4]), torch.tensor([2]))
    with pytest.raises(NotImplementedError):
        crit.ctc_loss(torch.randn(1, 4, 5, device=dev), torch.ones(1, 1024, device=dev), torch.tensor([4]), torch.tensor([2]))


def test_error_rates_follow_torchmetrics_definition():
    from conformer_amd.evaluation import ConformerMetric
    m = ConformerMetric()
    assert abs(float(m.wer_score(["a b c", "d e"], ["a x c", "d e f"])) - 2 / 6) < 1e-7      # 1 sub + 1 del over 6 words
    assert abs(float(m.cer_score("kitten", "sitting")) - 3 / 7) < 1e-7


@pytest.mark.parametrize("amp", [None, torch.float16, torch.bfloat16])
def test_reference_training_loop_with_all_drop_ins(dev, amp):
    """The loop of train.py:225-245 assembled from this repo's drop-ins only: Conformer, ConformerCriterion (lattice kernels),
    FusedAdam, GradScaler under fp16 (`--fp16 1`).  First-step loss and gradients equal the torch.nn.CTCLoss route on a
    twin model; the loss goes down."""
    from conformer_amd.evaluation import ConformerCriterion
    from conformer_amd.optim import FusedAdam
    from model.conformer import Conformer
    torch.manual_seed(0)
    m = Conformer(23, 80, 2, 32, 4, 31, 24, 1, 0.0).to(dev).train()
    twin = Conformer(23, 80, 2, 32, 4, 31, 24, 1, 0.0).to(dev).train()
    twin.load_state_dict(m.state_dict())
    g = torch.Generator().manual_seed(2)
    x = torch.randn(4, 80, 131, generator=g).to(dev)
    L = torch.tensor([131, 120, 99, 64], device=dev)                    # sorted descending, as dataset.py:97 guarantees
    tg = torch.randint(1, 23, (4, 6), generator=g).to(dev)
    tl = torch.tensor([6, 5, 6, 3], device=dev)
    crit = ConformerCriterion(blank_id=0)
    opt = FusedAdam(m.parameters(), lr=2e-3)
    scaler = torch.amp.GradScaler("cuda", enabled=amp == torch.float16)

    def forward(model):
        with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
            return model(x, L)

    # twin: torch's CTC on the same logits
    lt, ol = forward(twin)
    ref = torch.nn.functional.ctc_loss(lt.float().log_softmax(-1).transpose(0, 1), tg, ol, tl, blank=0, zero_infinity=True)
    ref.backward()
    losses = []
    for it in range(6):
        logits, ol = forward(m)
        loss = crit.ctc_loss(logits, tg, ol, tl)
        assert not torch.isnan(loss)                                     # train.py:236
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        if it == 0:
            scaler.unscale_(opt)
            assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))
            gm = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
            gt = torch.cat([p.grad.flatten() for p in twin.parameters() if p.grad is not None])
            assert rel_l2(gm, gt) < (2e-4 if amp is None else 5e-2)      # 16-bit: atomics order + lattice precision
        scaler.step(opt)
        scaler.update()
        losses.append(float(loss))
    assert losses[-1] < losses[0]
