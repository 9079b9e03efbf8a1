"""Opt-in fp32 matmul modes (csrc/gemm_split.hip): fp32 GEMMs computed on the bf16 matrix pipe from exact bf16 expansions
of the fp32 operands.  "bf16x6" must be as accurate as the native fp32 MFMA kernel (both against float64), "bf16x3" within
2^-15-class error; both far inside the 1e-3 parity bar."""
import pytest
import torch

from tests.util import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _native_after():
    from conformer_amd import ops
    yield
    ops.set_fp32_matmul("native")


def _data(m, n, k, dev, seed=0, spread=False):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    if spread:                                   # wide dynamic range: exponents from 2^-20 to 2^20
        a = a * torch.exp2(torch.randint(-20, 21, (m, k), generator=g).float())
        w = w * torch.exp2(torch.randint(-20, 21, (n, k), generator=g).float())
    b = torch.randn(n, generator=g)
    return a.to(dev), w.to(dev), b.to(dev)


def test_split_planes_reconstruct_fp32_exactly(dev):
    from conformer_amd import ops
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(64, 128, generator=g) * torch.exp2(torch.randint(-60, 60, (64, 128), generator=g).float())).to(dev)
    w[0, :4] = torch.tensor([0.0, -0.0, 1.0, 3.0e38])                 # (|x| >= 3.39e38 rounds to bf16 inf, as under autocast)
    packed = ops.weight_split(w, 3)
    planes = ops.unpack_weight_split(packed, 3, 64, 128)
    assert planes.shape == (3, 64, 128) and planes.dtype == torch.bfloat16
    bad = (planes.double().sum(0) != w.double()).sum()
    assert int(bad) == 0, f"{int(bad)} of {w.numel()} values are not reproduced exactly"   # x0 + x1 + x2 == x
    assert ops.weight_split(w, 3) is packed                            # cached
    w.add_(1.0)
    assert ops.weight_split(w, 3) is not packed                        # invalidated by the version bump
    w40 = torch.randn(40, 48, device=dev)                              # rows beyond N in the last block are zero
    p40 = ops.weight_split(w40, 2)
    assert p40.numel() == 2 * 64 * 48
    assert torch.equal(ops.unpack_weight_split(p40, 2, 40, 48).float().sum(0), w40.bfloat16().float() + (w40 - w40.bfloat16().float()).bfloat16().float())


@pytest.mark.parametrize("shape", [(7968, 2048, 512), (7968, 512, 2048), (300, 520, 80), (129, 72, 48), (5, 8, 16)])
@pytest.mark.parametrize("mode,bound", [("bf16x6", 1.0), ("bf16x3", 300.0)])
def test_split_gemm_error_vs_native_fp32(dev, shape, mode, bound):
    from conformer_amd import ops
    m, n, k = shape
    for act, spread in (("none", False), ("swish", False), ("none", True)):
        a, w, b = _data(m, n, k, dev, seed=2, spread=spread)
        ref = torch.nn.functional.linear(a.double(), w.double(), b.double())
        if act == "swish":
            ref = ref * torch.sigmoid(ref)
        ops.set_fp32_matmul("native")
        e_native = rel_l2(ops.linear(a, w, b, act), ref)
        ops.set_fp32_matmul(mode)
        e_split = rel_l2(ops.linear(a, w, b, act), ref)
        print(f"{mode} {shape} {act} spread={spread}: native {e_native:.2e} split {e_split:.2e}")
        assert e_split < 1e-4
        assert e_split <= bound * max(e_native, 3e-8) * 2.0              # x6: the native kernel's error class


def test_split_glu_residual_and_conv2(dev):
    from conformer_amd import ops
    a, w, b = _data(1000, 1024, 512, dev, seed=3)
    res = torch.randn(1000, 512, device=dev)
    z = torch.nn.functional.linear(a.double(), w.double(), b.double())
    glu_ref = z[:, :512] * torch.sigmoid(z[:, 512:])
    a2, w2, b2 = _data(1000, 512, 512, dev, seed=4)
    res_ref = 0.5 * torch.nn.functional.linear(a2.double(), w2.double(), b2.double()) + res.double()
    x = torch.randn(2, 80, 200, device=dev)
    c1w, c1b = torch.randn(64, 1, 3, 3, device=dev) / 3, torch.randn(64, device=dev)
    c2w, c2b = torch.randn(64, 64, 3, 3, device=dev) / 24, torch.randn(64, device=dev)
    h = torch.relu(torch.nn.functional.conv2d(x.double().unsqueeze(1), c1w.double(), c1b.double(), stride=2))
    h = torch.relu(torch.nn.functional.conv2d(h, c2w.double(), c2b.double(), stride=2))                    # (B,C,F2,T2)
    stem_ref = h.permute(0, 3, 2, 1).reshape(2, h.shape[3], -1)                                            # [t][f][c]
    for mode in ("native", "bf16x6", "bf16x3"):
        ops.set_fp32_matmul(mode)
        tol = 2e-6 if mode != "bf16x3" else 1e-4
        assert rel_l2(ops.linear_glu(a, w, b), glu_ref) < tol
        assert rel_l2(ops.linear_residual(a2, w2, b2, res, 0.5), res_ref) < tol
        w2p = ops.pack_conv2_weight(c2w)
        assert rel_l2(ops.subsample_stem(x, c1w, c1b, w2p, c2b), stem_ref) < tol


@pytest.mark.parametrize("mode,tol", [("bf16x6", 1e-4), ("bf16x3", 1e-4)])
def test_full_model_in_split_mode_vs_reference_golden(dev, mode, tol):
    """BASELINE cfg-1 model against the golden the reference produced: same bars as the native fp32 path (encoder / logits
    rel-L2 < 1e-4, per-frame argmax bit-exact) with every eligible GEMM computed from split operands."""
    from conformer_amd import ops
    from tests.test_model_gpu import build_model
    from tests.util import cfg_params, load_golden
    meta, g = load_golden("model_cfg1_S")
    m = build_model(meta, cfg_params(meta), dev)
    x, L = g["x"].to(dev), g["lengths"].to(dev)
    ops.set_fp32_matmul("native")
    with torch.no_grad():
        enc_native, _ = m.encoder(x, L)
    ops.set_fp32_matmul(mode)
    with torch.no_grad():
        enc, _ = m.encoder(x, L)
        logits, _ = m(x, L)
    e_native, e_split = rel_l2(enc_native, g["enc"]), rel_l2(enc, g["enc"])
    print(f"{mode}: encoder rel-L2 vs reference golden {e_split:.2e} (native fp32 MFMA: {e_native:.2e})")
    assert e_split < tol and rel_l2(logits, g["logits"]) < tol
    assert torch.equal(logits.argmax(-1).cpu(), g["argmax"])
    if mode == "bf16x6":
        assert e_split < 2.0 * e_native + 1e-7


def test_cfg2_encoder_split_vs_native(dev):
    """Full-size encoder (BASELINE cfg-2 shapes, B=4 to keep it quick): split modes against the native fp32 path."""
    from conformer_amd import ops
    from model.modules.encoder import Encoder
    torch.manual_seed(0)
    enc = Encoder(80, 16, 512, 8, 31, 0.0).to(dev).eval()
    x = torch.randn(4, 80, 1000, device=dev)
    L = torch.tensor([1000, 900, 640, 333], device=dev)
    with torch.no_grad():
        ref, _ = enc(x, L)
        for mode, tol in (("bf16x6", 2e-6), ("bf16x3", 5e-5)):
            ops.set_fp32_matmul(mode)
            y, _ = enc(x, L)
            err = rel_l2(y, ref)
            print(f"cfg-2 encoder {mode} vs native: rel-L2 {err:.2e}")
            assert err < tol


@pytest.mark.parametrize("mode", ["bf16x6", "autocast"])
def test_weight_caches_under_inference_mode(dev, mode):
    """Packs built inside torch.inference_mode() are inference tensors (no version counter): the 16-bit / split weight caches
    must accept them (fused QKV matrix, packed conv / linear weights)."""
    from conformer_amd import ops
    from model.modules.encoder import Encoder
    torch.manual_seed(0)
    enc = Encoder(80, 2, 64, 4, 31, 0.0).to(dev).eval()
    x = torch.randn(2, 80, 120, device=dev)
    L = torch.tensor([120, 77], device=dev)
    with torch.no_grad():
        ref, _ = enc(x, L)
    enc2 = Encoder(80, 2, 64, 4, 31, 0.0).to(dev).eval()
    enc2.load_state_dict(enc.state_dict())
    if mode != "autocast":
        ops.set_fp32_matmul(mode)
    with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode == "autocast"):
        y1, _ = enc2(x, L)                              # first call builds every pack inside inference mode
        y2, _ = enc2(x, L)
    assert torch.equal(y1, y2)
    assert rel_l2(y1, ref) < (2e-2 if mode == "autocast" else 2e-6)


def test_split_gemm_random_shapes_all_epilogues(dev):
    """Ragged M / N / K (K % 16 == 0, GLU widths % 32 == 0) through every epilogue, both split modes, against float64."""
    from conformer_amd import ops
    import random
    rnd = random.Random(7)
    g = torch.Generator().manual_seed(7)
    for it in range(36):
        m = rnd.choice([1, 2, 31, 33, 64, 65, 127, 129, 200, 257, 700])
        k = 16 * rnd.randint(1, 12)
        epi = ("none", "swish", "relu", "resid", "glu")[it % 5]
        n = 32 * rnd.randint(1, 6) if epi == "glu" else rnd.choice([1, 3, 8, 31, 32, 33, 64, 100, 129, 370])
        a = torch.randn(m, k, generator=g).to(dev)
        w = (torch.randn(2 * n if epi == "glu" else n, k, generator=g) / k ** 0.5).to(dev)
        b = torch.randn(w.shape[0], generator=g).to(dev)
        r = torch.randn(m, n, generator=g).to(dev)
        z = torch.nn.functional.linear(a.double(), w.double(), b.double())
        ref = {"none": lambda: z, "swish": lambda: z * torch.sigmoid(z), "relu": lambda: torch.relu(z),
               "resid": lambda: 0.25 * z + r.double(), "glu": lambda: z[:, :n] * torch.sigmoid(z[:, n:])}[epi]()
        for mode, tol in (("bf16x6", 2e-6), ("bf16x3", 1e-4)):
            ops.set_fp32_matmul(mode)
            y = {"none": lambda: ops.linear(a, w, b), "swish": lambda: ops.linear(a, w, b, "swish"),
                 "relu": lambda: ops.linear(a, w, b, "relu"), "resid": lambda: ops.linear_residual(a, w, b, r, 0.25),
                 "glu": lambda: ops.linear_glu(a, w, b)}[epi]()
            assert y.shape == ref.shape
            err = rel_l2(y, ref)
            assert err < tol, f"{mode} {epi} M={m} N={n} K={k}: {err:.2e}"
