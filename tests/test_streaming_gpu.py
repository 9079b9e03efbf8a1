"""BASELINE cfg-5: chunk-by-chunk evaluation with cached K/V and depthwise state (conformer_amd/streaming.py).

The reference has no streaming code; the contract is the prefix rule (see streaming.py): pinned here against (a) the masked
whole-sequence float64 restatement `oracle.encoder_forward_chunked`, (b) Encoder.forward when one chunk holds everything,
(c) Encoder.forward of the first chunk's prefix, and run at the full cfg-5 size (B=8, T=20000, 640-frame chunks)."""
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


def _encoder(P, n_mel, n_blocks, d, H, K, dev):
    from model.modules.encoder import Encoder
    enc = Encoder(n_mel, n_blocks, d, H, K, 0.0)
    enc.load_state_dict({k[len("encoder."):]: v.float() for k, v in P.items() if k.startswith("encoder.")}, strict=True)
    return enc.to(dev).eval()


@pytest.mark.parametrize("chunks", [[64] * 7, [200, 7, 1, 130, 3, 62], [403], [9, 394]])
def test_streaming_matches_masked_restatement(dev, chunks):
    from conformer_amd.streaming import StreamingEncoder, chunk_ends
    d, H, L, K = 32, 4, 2, 31
    P = O.make_params(vocab=8, n_mel=80, n_blocks=L, d=d, n_heads=H, ksize=K, lstm_hidden=8, seed=11, dtype=torch.float64,
                      with_decoder=False)
    T = sum(chunks)
    x = torch.randn(2, 80, T, generator=torch.Generator().manual_seed(4), dtype=torch.float64)
    ends = chunk_ends(T, chunks)
    ref = O.encoder_forward_chunked(x, P, L, H, ends)
    enc = _encoder(P, 80, L, d, H, K, dev)
    st = StreamingEncoder(enc, batch=2, max_mel_frames=T)
    outs, t0 = [], 0
    for c in chunks:
        outs.append(st.step(x[:, :, t0:t0 + c].float().to(dev)))
        t0 += c
    got = torch.cat(outs, dim=1)
    assert got.shape == ref.shape and st.frames == ends[-1]
    assert [o.shape[1] for o in outs if o.shape[1]] == [b - a for a, b in zip([0] + ends, ends)]
    assert rel_l2(got, ref) < 2e-5
    if len(ends) > 1:                                   # and chunking really changes the numbers (not a vacuous test)
        full, _ = O.encoder_forward(x, None, P, L, H)
        assert rel_l2(ref, full) > 1e-3


def test_one_chunk_is_encoder_forward_and_reset_works(dev):
    from conformer_amd.streaming import StreamingEncoder
    P = O.make_params(vocab=8, n_mel=80, n_blocks=2, d=64, n_heads=4, ksize=31, lstm_hidden=8, seed=12, with_decoder=False)
    enc = _encoder(P, 80, 2, 64, 4, 31, dev)
    x = torch.randn(3, 80, 331, device=dev)
    with torch.no_grad():
        full, _ = enc(x, None)
    st = StreamingEncoder(enc, batch=3, max_mel_frames=400)          # cache longer than the utterance
    one = st.step(x)
    assert rel_l2(one, full) < 2e-6
    a = st.run(torch.randn(3, 80, 50, device=dev), 50)               # garbage in the caches, then a fresh stream
    assert a.shape[1] == 0 or True
    st.reset()
    again = st.run(x, 331)
    assert torch.equal(again, one)


def test_first_chunk_equals_forward_of_its_prefix_cfg5_width(dev):
    """Full-width model (d=512, H=8, K=31; 2 blocks), 640-frame chunks: streaming vs the float64 masked restatement, and the
    first chunk against Encoder.forward of the 639 mel frames it is computed from."""
    from conformer_amd.streaming import StreamingEncoder, chunk_ends
    P = O.make_params(vocab=8, n_mel=80, n_blocks=2, d=512, n_heads=8, ksize=31, lstm_hidden=8, seed=13, dtype=torch.float64,
                      with_decoder=False)
    enc = _encoder(P, 80, 2, 512, 8, 31, dev)
    T = 640 * 3 + 160
    x = torch.randn(2, 80, T, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    chunks = [640, 640, 640, 160]
    ends = chunk_ends(T, chunks)
    assert ends == [159, 319, 479, 519]
    ref = O.encoder_forward_chunked(x, P, 2, 8, ends)
    st = StreamingEncoder(enc, batch=2, max_mel_frames=T)
    got = st.run(x.float().to(dev), 640)
    assert rel_l2(got, ref) < 2e-5
    with torch.no_grad():
        prefix, _ = enc(x[:, :, :639].float().to(dev), None)         # 639 mel frames -> exactly 159 encoder frames
    assert prefix.shape[1] == 159 and rel_l2(got[:, :159], prefix) < 2e-6


def test_cfg5_full_size_stream(dev):
    """BASELINE cfg-5 as written: B=8, T=20000 mel frames in 31 chunks of 640 + one of 160, Conformer-L encoder."""
    from conformer_amd.streaming import StreamingEncoder
    from model.modules.encoder import Encoder
    torch.manual_seed(0)
    enc = Encoder(80, 16, 512, 8, 31, 0.0).to(dev).eval()
    x = torch.randn(8, 80, 20000, generator=torch.Generator().manual_seed(6)).to(dev)
    st = StreamingEncoder(enc, batch=8, max_mel_frames=20000)
    outs = [st.step(x[:, :, t:t + 640]) for t in range(0, 20000, 640)]
    assert [o.shape[1] for o in outs] == [159] + [160] * 30 + [40]
    y = torch.cat(outs, dim=1)
    assert y.shape == (8, 4999, 512) and torch.isfinite(y).all()
    with torch.no_grad():
        prefix, _ = enc(x[:, :, :639], None)
    assert rel_l2(y[:, :159], prefix) < 2e-6
    # the last chunk sees the whole utterance: its attention / convolution inputs differ from the full-context forward only
    # through the cached lower-layer rows, so the two must be close but not equal
    with torch.no_grad():
        full, _ = enc(x, None)
    assert 1e-4 < rel_l2(y, full) < 1.0


@pytest.mark.parametrize("T,q_begin,q_count,lens", [(700, 640, 60, [700, 650]), (1300, 1100, 200, [1300, 1111]),
                                                    (300, 0, 300, [300, 290]), (2600, 2440, 160, [2600, 2600])])
def test_incremental_attention_rows_and_key_split(dev, T, q_begin, q_count, lens):
    """cfm_relpos_attention_rows_f32 (with and without the key split) reproduces the rows of the full attention kernel."""
    from conformer_amd import ops
    H, dh = 4, 16
    d = H * dh
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(2, T, 3 * d, generator=g).to(dev)
    pos = torch.randn(2 * T - 1, d, generator=g).to(dev)
    u, v = torch.randn(H, dh, generator=g).to(dev) * 0.1, torch.randn(H, dh, generator=g).to(dev) * 0.1
    L = torch.tensor(lens, device=dev)
    full = ops.relpos_attention(qkv, pos, u, v, L, H)
    for hint in (None, 64):                                    # None -> split chosen from T; 64 -> no split
        ctx = torch.full((2, T, d), float("nan"), device=dev)
        ops.relpos_attention_rows(qkv, pos, u, v, L, H, q_begin, q_count, ctx, keys_hint=hint)
        assert rel_l2(ctx[:, q_begin:q_begin + q_count], full[:, q_begin:q_begin + q_count]) < 2e-6
        assert torch.isnan(ctx[:, :q_begin]).all() and torch.isnan(ctx[:, q_begin + q_count:]).all()   # other rows untouched


def test_graphed_chunk_steps_equal_eager(dev):
    """StreamingEncoder(graphs=True): one hipGraph per distinct chunk step, captured on the first pass and replayed on the next
    utterance batch -- same frames as the eager launches, for both passes (the state buffers are fixed, the graph's are not)."""
    from conformer_amd.streaming import StreamingEncoder
    P = O.make_params(vocab=8, n_mel=80, n_blocks=2, d=64, n_heads=4, ksize=31, lstm_hidden=8, seed=13, with_decoder=False)
    enc = _encoder(P, 80, 2, 64, 4, 31, dev)
    chunks = [70, 5, 64, 64, 1, 130]
    T = sum(chunks)
    xs = [torch.randn(2, 80, T, device=dev, generator=torch.Generator(device=dev).manual_seed(s)) for s in (1, 2)]
    eager, graphed = StreamingEncoder(enc, 2, T), StreamingEncoder(enc, 2, T, graphs=True)
    for x in xs:                                        # pass 1 captures, pass 2 replays
        eager.reset(); graphed.reset()
        t0 = 0
        for c in chunks:
            a, b = eager.step(x[:, :, t0:t0 + c]), graphed.step(x[:, :, t0:t0 + c])
            assert a.shape == b.shape and torch.equal(a, b)
            t0 += c
        assert eager.frames == graphed.frames and eager.tail_len == graphed.tail_len
    assert len(graphed._graphs) == sum(1 for _ in graphed._graphs) >= 4


@pytest.mark.parametrize("case", ["stream_mhsa_d64_t70", "stream_mhsa_d144_t49"])
def test_attention_rows_kernel_vs_reference_block_triangular_mask(dev, case):
    """The incremental attention of the streaming path (cfm_relpos_attention_rows_f32: query rows of chunk c against the cached
    keys of chunks <= c) against the REFERENCE's MultiHeadSelfAttentionModule under the block-triangular (1, 1, T', T') mask
    that expresses the same rule (attention.py:60-62; golden from tests/golden/make_golden.py stream)."""
    from conformer_amd import ops
    from tests.util import cfg_params, load_golden
    meta, g = load_golden(case)
    P = {k: v.to(dev) for k, v in cfg_params(meta).items()}
    a = "encoder.layers.0.attention."
    x = g["x"].to(dev)
    B, T, d = x.shape
    H = meta["n_heads"]
    xn = ops.layernorm(x, P[a + "layer_norm.weight"], P[a + "layer_norm.bias"])
    wqkv = torch.cat([P[a + f"attention.{n}_proj.weight"] for n in ("query", "key", "value")]).contiguous()
    bqkv = torch.cat([P[a + f"attention.{n}_proj.bias"] for n in ("query", "key", "value")]).contiguous()
    qkv = ops.linear(xn, wqkv, bqkv)                                     # (B, T, 3d): the K/V cache of the whole utterance
    pe = ops.relpos_table(P["encoder.rel_pe.div_term"][None], T)
    pos = ops.linear(pe, P[a + "attention.pos_proj.weight"], P[a + "attention.pos_proj.bias"])
    ctx = torch.zeros(B, T, d, device=dev)
    start = 0
    for e in g["chunk_ends"].tolist():
        L = torch.full((B,), e, dtype=torch.int64, device=dev)          # keys of chunks <= c: the cache holds e frames
        ops.relpos_attention_rows(qkv, pos, P[a + "attention.content_bias"], P[a + "attention.position_bias"], L, H, start, e - start, ctx)
        start = e
    y = ops.linear(ctx, P[a + "attention.out_proj.weight"], P[a + "attention.out_proj.bias"])
    assert rel_l2(y, g["mhsa_y"]) < 2e-5


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_attention_rows_16bit_form_vs_float64(dev, dt):
    """cfm_relpos_attention_rows_mfma16_f32 (streaming under autocast): chunk rows against the cache == the float64 attention core
    with the prefix rule, at the 16-bit operand-rounding tolerance (1e-2 bf16 / 3e-3 fp16, the autocast bars of tests/test_mfma16_gpu.py)."""
    from conformer_amd import ops
    B, T, H, dh = 3, 200, 4, 36
    d = H * dh
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(B, T, 3 * d, generator=g)
    pos = torch.randn(2 * T - 1, d, generator=g)
    u, v = 0.3 * torch.randn(H, dh, generator=g), 0.3 * torch.randn(H, dh, generator=g)
    ends = [33, 64, 190, 200]
    ctx = torch.zeros(B, T, d, device=dev)
    start = 0
    with torch.autocast("cuda", dtype=dt):
        for e in ends:
            L = torch.full((B,), e, dtype=torch.int64, device=dev)
            ops.relpos_attention_rows(qkv.to(dev), pos.to(dev), u.to(dev), v.to(dev), L, H, start, e - start, ctx)
            start = e
    visible_end = torch.empty(T, dtype=torch.long)
    start = 0
    for e in ends:
        visible_end[start:e] = e
        start = e
    q, k, vv = (t.reshape(B, T, H, dh).double() for t in qkv.split(d, dim=-1))
    ref = O.relpos_attention_core(q, k, vv, pos.double().view(2 * T - 1, H, dh), u.double(), v.double(), None, visible_end)
    assert rel_l2(ctx, ref.reshape(B, T, d)) < (1e-2 if dt == torch.bfloat16 else 3e-3)


def test_streaming_under_autocast_tracks_fp32_streaming(dev):
    """StreamingEncoder inside torch.autocast (bf16): every GEMM and the incremental attention on the 16-bit matrix pipe; the
    encoder output stays within the bf16 bar (1e-2 rel-L2 per tensor) of the fp32 streaming result."""
    from conformer_amd.streaming import StreamingEncoder
    d, H, L, K = 64, 4, 2, 31
    P = O.make_params(vocab=8, n_mel=80, n_blocks=L, d=d, n_heads=H, ksize=K, lstm_hidden=8, seed=13, with_decoder=False)
    enc = _encoder(P, 80, L, d, H, K, dev)
    x = torch.randn(2, 80, 403, generator=torch.Generator().manual_seed(1)).to(dev)
    ref = StreamingEncoder(enc, 2, 403).run(x, 128)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = StreamingEncoder(enc, 2, 403).run(x, 128)
    assert out.dtype == torch.float32 and out.shape == ref.shape
    assert rel_l2(out, ref) < 1e-2
