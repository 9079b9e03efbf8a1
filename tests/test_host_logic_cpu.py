"""Host-side caches and seed plumbing (no GPU): PackCache keys, cache invalidation, per-rank dropout seeds."""
import os

import torch

from conformer_amd.model.utils import _guard


def test_packcache_handles_inference_tensors_and_in_place_updates():
    """Inference tensors carry no version counter (`_version` raises): the reference's infer.py / test.py run under
    torch.inference_mode(), where cached packs and position tables are such tensors (round-1 advisor finding)."""
    cache = _guard.PackCache()
    with torch.inference_mode():
        w = torch.ones(4, 4)
        b = torch.zeros(4)
    calls = []

    def make():
        calls.append(1)
        return (w.sum() + b.sum()).clone()

    v1 = cache.get("k", (w, b), make)
    v2 = cache.get("k", (w, b), make)
    assert v1 is v2 and len(calls) == 1
    p = torch.nn.Parameter(torch.ones(3))
    mk = lambda: p.detach().clone()
    a = cache.get("p", (p,), mk)
    assert cache.get("p", (p,), mk) is a
    with torch.no_grad():
        p.mul_(2.0)                                  # optimizer-step-like update: version bump -> rebuilt
    a2 = cache.get("p", (p,), mk)
    assert a2 is not a and float(a2[0]) == 2.0
    p.data.copy_(torch.full((3,), 5.0))              # bypasses the version counter: invisible ...
    assert cache.get("p", (p,), mk) is a2
    _guard.invalidate_weight_caches()                # ... until the documented invalidation call
    assert float(cache.get("p", (p,), mk)[0]) == 5.0


def test_packcache_identity_not_only_address():
    cache = _guard.PackCache()
    t1 = torch.ones(8)
    v1 = cache.get("k", (t1,), lambda: t1 * 2)
    t2 = t1.detach()                                 # same storage address and version, different tensor object
    v2 = cache.get("k", (t2,), lambda: t2 * 3)
    assert v2 is not v1 and float(v2[0]) == 3.0


def test_dropout_seeds_are_per_rank_and_leave_the_default_generator_alone(monkeypatch):
    from conformer_amd import ops
    torch.manual_seed(1234)
    before = torch.get_rng_state().clone()
    monkeypatch.setenv("RANK", "0")
    s0 = ops.new_seeds(4)
    assert torch.equal(torch.get_rng_state(), before)          # SpecAugment's band draws are not perturbed
    monkeypatch.setenv("RANK", "1")
    s1 = ops.new_seeds(4)
    assert s0 != s1 and len(set(s0 + s1)) == 8                  # replicas seeded alike draw different masks
    torch.manual_seed(1234)
    monkeypatch.setenv("RANK", "0")
    ops._SEED_GEN.clear()
    assert ops.new_seeds(4) == s0                               # torch.manual_seed still reproduces a run
    assert ops.new_seeds(4) != s0                               # and the stream advances from step to step


def test_kernel_outputs_state_their_dtype():
    """no kernel-output buffer inherits an input's dtype (`empty_like` / `zeros_like`): the C ABI takes
    raw pointers, so a 16-bit input would silently halve an fp32 output's allocation."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r"torch\.(empty|zeros|ones)_like\(")
    for rel in ("ops.py", "autograd.py", "streaming.py", "frontend.py", "decode.py", "optim.py", "evaluation.py", "pipeline.py",
                "graph.py"):
        src = open(os.path.join(root, "conformer_amd", rel)).read()
        code = "\n".join(line.split("#", 1)[0] for line in src.splitlines())
        assert not pat.search(code), f"conformer_amd/{rel} allocates an output with *_like"


def test_fused_feed_forward_gating_follows_cu_rounds():
    """ops.ffn_fused_ok: one workgroup per 32 rows and per CU -- taken when the row blocks fill whole rounds of 256 CUs to >= 90 %."""
    from conformer_amd import ops
    prev_fold, prev_ffn = ops.set_ln_fold(True), ops.set_ffn_fused(True)
    try:
        assert ops.ffn_fused_ok(512, 2048, 7968)            # cfg-2: 249 blocks
        assert ops.ffn_fused_ok(512, 2048, 15936)           # cfg-3 geometry: 498 blocks = two rounds
        assert ops.ffn_fused_ok(256, 1024, 32 * 240) and not ops.ffn_fused_ok(256, 1024, 32 * 192)   # 0.94 / 0.75 of a round
        assert not ops.ffn_fused_ok(512, 2048, 1280)        # a streaming chunk
        assert not ops.ffn_fused_ok(512, 2048, 32 * 257)    # one full round + one block
        assert not ops.ffn_fused_ok(144, 576, 7968) and not ops.ffn_fused_ok(512, 2000, 7968)
        assert not ops.rowchain_ok(512, 2048, 7968)         # the row chains are opt-in
        ops.set_ffn_fused(False)
        assert not ops.ffn_fused_ok(512, 2048, 7968)
    finally:
        ops.set_ln_fold(prev_fold); ops.set_ffn_fused(prev_ffn)
