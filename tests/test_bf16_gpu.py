"""bf16-MFMA path (torch.autocast(bfloat16)): (a) the GEMM against an fp64 product of the bf16-ROUNDED operands (tight:
only fp32 accumulation error remains), (b) per-block output against the fp32 oracle within the north_star's 1e-2 rel for
bf16 (each block is fed the oracle's fp32 input), (c) argmax indices of the cfg-1 model vs the fp32 golden."""
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


def bf(t):
    return t.to(torch.bfloat16).double()


@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (100, 144, 144), (257, 576, 144), (7968, 512, 512), (300, 2048, 512),
                                   (300, 512, 2048), (129, 130, 20)])
def test_bf16_gemm_epilogues(dev, M, N, K):
    from conformer_amd import ops
    a, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2) / math.sqrt(K), rnd(N, seed=3), rnd(M, N, seed=4)
    ref = bf(a) @ bf(w).t() + b.double()
    G = lambda t: t.to(dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert rel_l2(ops.linear(G(a), G(w), G(b)), ref) < 2e-5
        assert rel_l2(ops.linear(G(a), G(w), G(b), act="swish"), O.swish(ref)) < 2e-5
        assert rel_l2(ops.linear(G(a), G(w), G(b), act="relu"), torch.relu(ref)) < 2e-5
        assert rel_l2(ops.linear_residual(G(a), G(w), G(b), G(r), 0.5), 0.5 * ref + r.double()) < 2e-5
        if N % 2 == 0:
            n = N // 2
            assert rel_l2(ops.linear_glu(G(a), G(w), G(b)), ref[:, :n] * torch.sigmoid(ref[:, n:])) < 2e-5
    # and against the un-rounded product: bf16 operand rounding only (2^-9 per operand, averaged over K)
    assert rel_l2(ops.linear(G(a), G(w), G(b)), a.double() @ w.double().t() + b.double()) < 2e-5   # fp32 path outside autocast


def test_fp16_autocast_is_refused(dev):
    from conformer_amd import ops
    from conformer_amd._lib import ConformerHipError
    with torch.autocast("cuda", dtype=torch.float16), pytest.raises(ConformerHipError):
        ops.linear(torch.zeros(4, 16, device=dev), torch.zeros(8, 16, device=dev), torch.zeros(8, device=dev))


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
    Pd = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    ref = O.conformer_block(x.double(), O.relpos_table(249, Pd["encoder.rel_pe.div_term"]), L, Pd, blk, 8)
    err = rel_l2(y, ref)
    assert 1e-5 < err < 1e-2, err            # really on the bf16 path, and inside the bf16 budget


def test_cfg1_model_bf16_argmax_and_drift(dev):
    """BASELINE cfg-1 (Conformer-S): under bf16 autocast the encoder drifts < 2e-2 end to end (the reference's own CPU
    bf16 autocast drifts 0.9e-2 after 4 blocks) and the CTC argmax indices agree with the fp32 reference
    except at near-ties (reported, must be < 2 % of the frames)."""
    from model.conformer import Conformer
    meta, g = load_golden("model_cfg1_S")
    P = cfg_params(meta)
    m = Conformer(meta["vocab"], 80, meta["n_blocks"], meta["d"], meta["n_heads"], meta["ksize"], meta["lstm_hidden"], 1, 0.0)
    m.load_state_dict(P, strict=True)
    m = m.to(dev).eval()
    x, L = g["x"].to(dev), g["lengths"].to(dev)
    with torch.no_grad():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            enc, L2 = m.encoder(x, L)
        logits = m.decoder(enc, L2)
    assert rel_l2(enc, g["enc"]) < 2e-2
    mism = (logits.argmax(-1).cpu() != g["argmax"]).float().mean().item()
    assert mism < 0.02, mism
