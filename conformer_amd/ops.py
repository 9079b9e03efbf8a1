"""Tensor-level wrappers over the C ABI: torch is used only for device memory and the current stream.

Every function takes CUDA(HIP) fp32 tensors, allocates its output with torch.empty and enqueues the HIP
kernels on torch's current stream.  Inputs on the wrong device/dtype raise; there is no eager fallback.
"""
from __future__ import annotations

import threading
import ctypes
import weakref
from typing import Optional, Tuple

import torch

from . import _lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.ConformerHipError(
            f"{name}: expected a tensor on a HIP device, got {getattr(t, 'device', type(t))}; "
            "the Conformer hot path only exists as gfx950 kernels (no CPU fallback)")
    if t.dtype != dtype:
        raise _lib.ConformerHipError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _f32_like(t: torch.Tensor) -> torch.Tensor:
    """An fp32 kernel-output buffer with t's shape.  Every kernel output in this file states its dtype: an output that inherits
    the dtype of an input (`empty_like`) is half the size its kernel writes as soon as that input is stored in a 16-bit type --
    the ABI takes raw pointers, so nothing else would notice (the stem-backward fault of round 2)."""
    return torch.empty(t.shape, device=t.device, dtype=torch.float32)


PREC_F32, PREC_BF16, PREC_FP16 = 0, 1, 2          # CFM_PREC_* of include/conformer_hip.h (0 = the fp32 MFMA path)
_DT16 = {PREC_BF16: torch.bfloat16, PREC_FP16: torch.float16}
_tls = threading.local()       # .forced: precision pinned by `precision(...)` for the current thread (autograd runs the
                               # backward on its own device thread: a process-global would race between threads)


def mfma16_prec() -> int:
    """Matrix-pipe precision of the dense GEMMs: inside `torch.autocast("cuda")` they round their operands to the autocast
    dtype (bf16, or fp16 as the reference's `--fp16 1` does, train.py:6,217,232) and run on the 16-bit matrix pipe with
    fp32 accumulation and fp32 tensors -- the arithmetic autocast gives nn.Linear / Conv.  Backward passes run under
    `precision(...)` with the value their forward saw (autocast is not active inside autograd's backward)."""
    forced = getattr(_tls, "forced", None)
    if forced is not None:
        return forced
    if not torch.is_autocast_enabled("cuda"):
        return PREC_F32
    dt = torch.get_autocast_dtype("cuda")
    if dt == torch.bfloat16:
        return PREC_BF16
    if dt == torch.float16:
        return PREC_FP16
    raise _lib.ConformerHipError(f"autocast dtype {dt} is not supported by the gfx950 path")


class precision:
    """Context manager pinning mfma16_prec() (used by the autograd Functions around their backward)."""

    def __init__(self, prec: int) -> None:
        self.prec, self.prev = prec, None

    def __enter__(self):
        self.prev = getattr(_tls, "forced", None)
        _tls.forced = self.prec
        return self

    def __exit__(self, *exc):
        _tls.forced = self.prev
        return False


def bf16_mfma_active() -> bool:
    return mfma16_prec() == PREC_BF16


def _ver(t: torch.Tensor) -> int:
    """`_version` of a weight, 0 for inference tensors (packs built under torch.inference_mode() carry no version counter;
    they are never modified in place -- PackCache rebuilds them as new tensors when a source parameter changes)."""
    return 0 if t.is_inference() else t._version


_W16_CACHE = {}        # data_ptr -> (weakref(base tensor), version, prec, shape, 16-bit tensor): per-optimizer-step weight casts
def weight16(w: torch.Tensor, prec: int, transposed: bool = False) -> Optional[torch.Tensor]:
    """The weight matrix rounded to the 16-bit matrix-pipe type, cached until the parameter changes in place (optimizer
    step / load_state_dict bump `_version`).  None when the 16-bit operand path does not apply (K % 8 != 0).
    transposed: the (K, N) transpose of the (N, K) weight (the backward's dX = dY.W as a row-major-A forward GEMM)."""
    if w.shape[-1] % 8 or w.numel() % 4 or (transposed and w.shape[0] % 8):
        return None
    key = (w.data_ptr(), transposed)
    base = w._base if w._base is not None else w     # views are re-created per call; the owning tensor identifies the weight
    hit = _W16_CACHE.get(key)
    # (a freed temporary -- e.g. last step's fused QKV matrix -- can hand its address to a new tensor: the weak reference
    # to the owner tells the two apart; `_version` catches in-place updates of a live parameter)
    if hit is not None and hit[0]() is base and hit[1] == _ver(w) and hit[2] == prec and hit[3] == w.shape:
        return hit[4]
    if transposed:
        out = torch.empty(w.shape[1], w.shape[0], device=w.device, dtype=_DT16[prec])
        item = (_lib.CastItem * 1)()
        item[0].src, item[0].dst, item[0].rows, item[0].cols, item[0].transpose = w.data_ptr(), out.data_ptr(), w.shape[0], w.shape[1], 1
        _lib.check(_lib.load().cfm_cast16_multi_f32(prec, ctypes.addressof(item), 1, _stream()), "cfm_cast16_multi_f32")
    else:
        out = torch.empty(w.shape, device=w.device, dtype=_DT16[prec])
        _lib.check(_lib.load().cfm_cast16_f32(prec, w.data_ptr(), out.data_ptr(), w.numel(), _stream()), "cfm_cast16_f32")
    if len(_W16_CACHE) > 4096:
        for k in [k for k, v in _W16_CACHE.items() if v[0]() is None]:
            del _W16_CACHE[k]
    _W16_CACHE[key] = (weakref.ref(base), _ver(w), prec, w.shape, out)
    return out


def refresh_weight16(params) -> int:
    """Re-cast, in ONE batched launch per 48 weights, every cached 16-bit copy (plain and transposed) of the given parameters
    that went stale -- called by FusedAdam right after its update, so the next forward finds every copy current instead of
    issuing ~130 five-microsecond cast launches (+ ~130 ATen kernels for the transposed copies) one by one.  The copies are
    rewritten IN PLACE (same addresses: graph-friendly; the previous step's kernels that read them precede the optimizer
    step on the stream).  Returns the number of copies refreshed."""
    by_prec = {}
    for p in params:
        for transposed in (False, True):
            key = (p.data_ptr(), transposed)
            hit = _W16_CACHE.get(key)
            if hit is None or hit[0]() is not (p._base if p._base is not None else p) or hit[1] == _ver(p):
                continue
            shape2 = tuple(hit[3])                     # the 2-D view the copy was made from (conv weights arrive reshaped)
            if len(shape2) != 2 or shape2[0] * shape2[1] != p.numel() or not p.is_contiguous():
                continue
            w2 = p.detach().reshape(shape2)
            by_prec.setdefault(hit[2], []).append((w2, hit[4], transposed))
            _W16_CACHE[key] = (hit[0], _ver(p), hit[2], hit[3], hit[4])
    n = 0
    for prec, items in by_prec.items():
        arr = (_lib.CastItem * len(items))()
        for i, (w2, out, transposed) in enumerate(items):
            arr[i].src, arr[i].dst = w2.data_ptr(), out.data_ptr()
            arr[i].rows, arr[i].cols, arr[i].transpose = w2.shape[0], w2.shape[1], int(transposed)
        _lib.check(_lib.load().cfm_cast16_multi_f32(prec, ctypes.addressof(arr), len(items), _stream()), "cfm_cast16_multi_f32")
        n += len(items)
    return n


_ZERO_BIAS = {}


def _zero_bias(n: int, device) -> torch.Tensor:
    """A cached all-zero bias vector (read-only) for GEMM entry points that always add one."""
    t = _ZERO_BIAS.get((n, device))
    if t is None:
        t = _ZERO_BIAS[(n, device)] = torch.zeros(n, device=device, dtype=torch.float32)
    return t


# ---- opt-in fp32 matmul mode: exact bf16 operand splitting on the 16x faster bf16 matrix pipe (csrc/gemm_split.hip) -----------
_FP32_MATMUL = {"native": 0, "bf16x6": 3, "bf16x3": 2}       # value = bf16 planes per operand
_fp32_planes = _FP32_MATMUL[__import__("os").environ.get("CONFORMER_AMD_FP32_MATMUL", "native")]


def set_fp32_matmul(mode: str) -> str:
    """How the forward computes its fp32 GEMMs outside autocast (the plain / GLU / residual linear layers and the stem's
    conv2; GEMMs with fused dropout or saved pre-activations and every backward GEMM stay "native"):
      "native"  v_mfma_f32_32x32x2_f32 (default);
      "bf16x6"  every fp32 operand split exactly into three bf16 terms, six bf16 MFMAs per K-step: fp32 inputs, fp32
                accumulation, fp32 outputs, per-product error <= 2^-23 (fp32 class), 0.375x the matrix-pipe time;
      "bf16x3"  two terms, three MFMAs: error <= 2^-15 (60x inside the 1e-3 parity bar).
    Returns the previous mode.  Also settable with CONFORMER_AMD_FP32_MATMUL."""
    global _fp32_planes
    if mode not in _FP32_MATMUL:
        raise ValueError(f"fp32 matmul mode must be one of {sorted(_FP32_MATMUL)}, got {mode!r}")
    prev = fp32_matmul()
    _fp32_planes = _FP32_MATMUL[mode]
    return prev


def fp32_matmul() -> str:
    return {v: k for k, v in _FP32_MATMUL.items()}[_fp32_planes]


_WSPLIT_CACHE = {}     # data_ptr -> (weakref(base), version, planes, shape, [planes][N][K] bf16)
def weight_split(w: torch.Tensor, planes: int) -> Optional[torch.Tensor]:
    """Exact bf16 expansion of a (N,K) weight matrix in MFMA fragment order ([row block of 32][K-step of 16][plane][lane][8]),
    cached like weight16.  None when the split path does not apply (K % 16 != 0)."""
    if w.dim() != 2 or w.shape[1] % 16:
        return None
    key = w.data_ptr()
    base = w._base if w._base is not None else w
    hit = _WSPLIT_CACHE.get(key)
    if hit is not None and hit[0]() is base and hit[1] == _ver(w) and hit[2] == planes and hit[3] == w.shape:
        return hit[4]
    lib = _lib.load()
    n, k = w.shape
    out = torch.empty(int(lib.cfm_split_pack_elems(planes, n, k)), device=w.device, dtype=torch.bfloat16)
    _lib.check(lib.cfm_split_pack_bf16_f32(planes, w.data_ptr(), out.data_ptr(), n, k, _stream()), "cfm_split_pack_bf16_f32")
    if len(_WSPLIT_CACHE) > 4096:
        for kk in [kk for kk, v in _WSPLIT_CACHE.items() if v[0]() is None]:
            del _WSPLIT_CACHE[kk]
    _WSPLIT_CACHE[key] = (weakref.ref(base), _ver(w), planes, w.shape, out)
    return out


def unpack_weight_split(packed: torch.Tensor, planes: int, n: int, k: int) -> torch.Tensor:
    """(planes, n, k) bf16 view of weight_split's buffer in natural order (tests / inspection)."""
    nb, ks = (n + 31) // 32, k // 16
    p = packed.view(nb, ks, planes, 2, 32, 8)                      # [block][kstep][plane][hf][li][e]
    return p.permute(2, 0, 4, 1, 3, 5).reshape(planes, nb * 32, k)[:, :n]


def _split_gemm(epi: int, a, w2, b, c, m, n, k, res=None, alpha: float = 1.0):
    """Returns c, or None when the split path does not apply to this call (the caller then runs the native kernel)."""
    if not _fp32_planes or a.dtype != torch.float32 or c.dtype != torch.float32:
        return None
    ws = weight_split(w2, _fp32_planes) if (epi != 3 or n % 32 == 0) else None
    if ws is None:
        return None
    _lib.check(_lib.load().cfm_gemm_split_bf16_f32(_fp32_planes, epi, a.data_ptr(), ws.data_ptr(), b.data_ptr(), _p(res), alpha,
                                                   c.data_ptr(), m, n, k, k, n, n, _stream()), "cfm_gemm_split_bf16_f32")
    return c


def out16_ok(k: int) -> int:
    """Precision in which a producer may write a tensor that only feeds GEMM operands with contraction length k (0: keep fp32)."""
    prec = mfma16_prec()
    return prec if prec and k % 8 == 0 else 0


def _mfma16_gemm(prec: int, epi: int, a, w2, b, c, m, n, k, res=None, alpha: float = 1.0, z=None, drop_p: float = 0.0,
                 seed: int = 0):
    w16 = weight16(w2, prec)
    a16 = a.dtype != torch.float32
    if a16 and (w16 is None or a.dtype != _DT16[prec]):
        raise _lib.ConformerHipError(f"16-bit A operand ({a.dtype}) does not match the precision mode / weight shape")
    st = _lib.load().cfm_gemm_mfma16_f32(prec, epi, a.data_ptr(), int(a16), (w2 if w16 is None else w16).data_ptr(),
                                         int(w16 is not None), b.data_ptr(), _p(res), alpha, c.data_ptr(),
                                         int(c.dtype != torch.float32), _p(z), int(z is not None and z.dtype != torch.float32),
                                         m, n, k, k, n, n, float(drop_p), int(seed), _stream())
    _lib.check(st, "cfm_gemm_mfma16_f32")
    return c


def subsampled_length(n: int) -> int:
    return int(_lib.load().cfm_subsampled_length(int(n)))


def subsampled_lengths(lengths: torch.Tensor) -> torch.Tensor:
    """((L-1)//2-1)//2 of an int64 device array (convolution.py:55) in one launch."""
    lengths = _req(lengths, "lengths", torch.int64)
    out = torch.empty(lengths.shape, device=lengths.device, dtype=torch.int64)
    if lengths.numel():
        _lib.check(_lib.load().cfm_subsampled_lengths_i64(lengths.data_ptr(), out.data_ptr(), lengths.numel(), _stream()),
                   "cfm_subsampled_lengths_i64")
    return out


# ---- LayerNorm folded into the GEMMs either side of it (fp32 inference; csrc/gemm_f32.hip "LN fold") -------------------------
_LN_FOLD = __import__("os").environ.get("CONFORMER_AMD_LN_FOLD", "1") != "0"


def set_ln_fold(on: bool) -> bool:
    """Enable / disable the folded-LayerNorm inference path (default on; CONFORMER_AMD_LN_FOLD=0 disables).  Returns the
    previous setting.  Off = every LayerNorm is its own kernel launch, as in rounds 1-2."""
    global _LN_FOLD
    prev, _LN_FOLD = _LN_FOLD, bool(on)
    return prev


def ln_fold_ok(d: int) -> bool:
    """The fold applies to the native fp32 matrix-pipe path (no autocast, no split mode) for rows of d = 32, 64, 128, 256 or 512
    values (the statistics travel as one partial per 32 columns; a workgroup merges the 1..16 partials of a row in a binary tree)."""
    return bool(_LN_FOLD and not _fp32_planes and mfma16_prec() == PREC_F32 and d in (32, 64, 128, 256, 512))


def fold_layernorm(w: torch.Tensor, b: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor):
    """One-time re-parametrisation (cached per weight version by the callers): LN(x).W^T + b with LN's affine part moved
    into the Linear: Wf = W.diag(gamma), bias_f = b + W.beta, colsum[n] = sum_k Wf[n,k] (the sum of the fp32 values the
    kernel multiplies, accumulated in float64).  Returns (Wf, bias_f, colsum)."""
    w2 = w.detach().reshape(w.shape[0], -1)
    wf = (w2 * gamma.detach()[None, :]).contiguous()
    bf = (b.detach().double() + w2.double() @ beta.detach().double()).float().contiguous()
    cs = wf.double().sum(dim=1).float().contiguous()
    return wf, bf, cs


def linear_lnfold(a, stats, wf, bf, cs, eps: float, act: str = "none", glu: bool = False) -> torch.Tensor:
    """epi(LN(a) @ W.T + b) from the un-normalised rows `a` (..., K), their statistics partials `stats` (rows, parts, 2)
    and the folded parameters of fold_layernorm.  act: none | swish; glu: W has 2N rows (values, gates)."""
    a = _req(a, "A"); wf = _req(wf, "Wf"); bf = _req(bf, "bias_f"); cs = _req(cs, "colsum"); stats = _req(stats, "ln_stats")
    k = a.shape[-1]
    m = a.numel() // k
    n = wf.shape[0] // 2 if glu else wf.shape[0]
    if wf.shape[1] != k or stats.dim() != 3 or stats.shape[0] != m or stats.shape[2] != 2:
        raise _lib.ConformerHipError(f"linear_lnfold: A(...,{k}), Wf{tuple(wf.shape)}, stats{tuple(stats.shape)} do not match")
    c = torch.empty(*a.shape[:-1], n, device=a.device, dtype=torch.float32)
    epi = 3 if glu else {"none": 0, "swish": 1}[act]
    _lib.check(_lib.load().cfm_gemm_lnfold_f32(epi, a.data_ptr(), stats.data_ptr(), stats.shape[1], float(eps), wf.data_ptr(),
                                               bf.data_ptr(), cs.data_ptr(), c.data_ptr(), m, n, k, k, n, _stream()),
               "cfm_gemm_lnfold_f32")
    return c


# ---- fused feed-forward sub-layer (fp32 inference; csrc/ffn_fused_f32.hip) ---------------------------------------------------
_FFN_FUSED = __import__("os").environ.get("CONFORMER_AMD_FFN_FUSED", "1") != "0"


def set_ffn_fused(on: bool) -> bool:
    """Enable / disable the one-kernel feed-forward sub-layer of the folded-LayerNorm inference path (default on;
    CONFORMER_AMD_FFN_FUSED=0 disables).  Returns the previous setting.  Off = hidden GEMM + residual GEMM (two launches)."""
    global _FFN_FUSED
    prev, _FFN_FUSED = _FFN_FUSED, bool(on)
    return prev


def ffn_fused_ok(d: int, hidden: int, rows: int) -> bool:
    """One workgroup per 32 rows, one workgroup per CU, ~264 us per round whatever the fill: worth it when the row blocks fill whole
    rounds of the chip's 256 CUs to >= 90 % (measured: 7968 rows = 249 blocks 264 vs 287 us for the two GEMMs, 15936 rows = 498
    blocks 527 vs 560 us, but 6144 rows = 192 blocks 262 vs 226 us: the two-GEMM path scales with the rows, ~36 ns per row)."""
    if not (_FFN_FUSED and ln_fold_ok(d) and d in (128, 256, 512) and hidden % 128 == 0):
        return False
    blocks = (rows + 31) // 32
    rounds = (blocks + 255) // 256
    return blocks >= 0.9 * 256 * rounds


_FFN_LAYOUT_ENV = __import__("os").environ.get("CONFORMER_AMD_FFN_ROTATE")    # diagnostics: "0" = every workgroup walks the slices in order


def ffn_pack(w1f: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """Both Linear weights of a FeedForwardModule in MFMA fragment order, slice by slice (once per weight version).
    w1f: (hidden, d) = the LayerNorm-folded hidden weight (fold_layernorm); w2: (d, hidden)."""
    w1f = _req(w1f, "W1f"); w2 = _req(w2, "W2")
    hidden, d = w1f.shape
    if tuple(w2.shape) != (d, hidden):
        raise _lib.ConformerHipError(f"ffn_pack: W1f{tuple(w1f.shape)} and W2{tuple(w2.shape)} do not match")
    lib = _lib.load()
    if _FFN_LAYOUT_ENV is not None:
        lib.cfm_debug_ffn_layout(-1, int(_FFN_LAYOUT_ENV))
    wp = torch.zeros(int(lib.cfm_ffn_pack_elems(d, hidden)), device=w1f.device, dtype=torch.float32)   # (pads between tiles: zero)
    _lib.check(lib.cfm_ffn_pack_f32(w1f.data_ptr(), w2.data_ptr(), wp.data_ptr(), d, hidden, _stream()), "cfm_ffn_pack_f32")
    return wp


def ffn_fused(x: torch.Tensor, stats: torch.Tensor, wp: torch.Tensor, b1f: torch.Tensor, cs: torch.Tensor, b2: torch.Tensor,
              alpha: float, eps: float, emit_stats: bool = False, closing_ln=None):
    """alpha * (swish(LN(x) @ W1.T + b1) @ W2.T + b2) + x in one kernel, LN folded (stats = the statistics partials of x's rows).
    emit_stats: also return the statistics partials of the result's rows.  closing_ln = (weight, bias, eps): the result goes
    through that LayerNorm (block.py:27) before it is stored; emit_stats then returns the (rows, 1, 2) statistics of ITS output."""
    x = _req(x, "x"); stats = _req(stats, "ln_stats"); wp = _req(wp, "Wp"); b1f = _req(b1f, "b1f"); cs = _req(cs, "colsum")
    b2 = _req(b2, "b2")
    d = x.shape[-1]
    rows = x.numel() // d
    hidden = b1f.numel()
    if stats.dim() != 3 or stats.shape[0] != rows or stats.shape[2] != 2 or wp.numel() != int(_lib.load().cfm_ffn_pack_elems(d, hidden)) or b2.numel() != d:
        raise _lib.ConformerHipError(f"ffn_fused: x(...,{d}), stats{tuple(stats.shape)}, Wp({wp.numel()}), hidden {hidden} do not match")
    y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    st_out, g2, bt2, eps2, mode = None, None, None, 0.0, 0
    if closing_ln is not None:
        g2 = _req(closing_ln[0], "ln.weight"); bt2 = _req(closing_ln[1], "ln.bias"); eps2 = float(closing_ln[2]); mode = 2
        if emit_stats:
            st_out = torch.empty(rows, 1, 2, device=x.device, dtype=torch.float32)
    elif emit_stats:
        mode = 1
        st_out = torch.empty(rows, d // 32, 2, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().cfm_ffn_fused_f32(x.data_ptr(), d, stats.data_ptr(), stats.shape[1], float(eps), wp.data_ptr(),
                                             b1f.data_ptr(), cs.data_ptr(), b2.data_ptr(), float(alpha), y.data_ptr(), d, mode,
                                             None if st_out is None else st_out.data_ptr(),
                                             None if g2 is None else g2.data_ptr(), None if bt2 is None else bt2.data_ptr(),
                                             eps2, rows, d, hidden, _stream()), "cfm_ffn_fused_f32")
    return (y, st_out) if emit_stats else y


# ---- row-local chains of a block (fp32 inference; csrc/rowchain_f32.hip) --------------------------------------------------------
_ROWCHAIN = __import__("os").environ.get("CONFORMER_AMD_ROWCHAIN", "0") == "1"


def set_rowchain(on: bool) -> bool:
    """Enable / disable the three-kernel row-chain form of a block's row-local operators.  Default OFF (CONFORMER_AMD_ROWCHAIN=1
    enables): measured inside the cfg-2 forward the chains are a wash against the kernels they merge (K1 378 vs 276 + 110 us,
    K2 126 vs 43 + 77 us, K3 314 vs 43 + 272 us; forward 20.6 vs 20.5 ms -- DESIGN.md section 5, round 3).  Returns the previous
    setting."""
    global _ROWCHAIN
    prev, _ROWCHAIN = _ROWCHAIN, bool(on)
    return prev


def rowchain_ok(d: int, hidden: int, rows: int) -> bool:
    return bool(_ROWCHAIN and ffn_fused_ok(d, hidden, rows))


def rowgemm_pack(w: torch.Tensor, glu: bool = False) -> torch.Tensor:
    """A (N, d) Linear weight in MFMA fragment order per wave for the row chains (once per weight version); glu: (2 d, d), value
    rows then gate rows."""
    w = _req(w.reshape(w.shape[0], -1), "W")
    n, d = w.shape
    wp = torch.empty(n * d, device=w.device, dtype=torch.float32)
    _lib.check(_lib.load().cfm_rowgemm_pack_f32(w.data_ptr(), wp.data_ptr(), n, d, 1 if glu else 0, _stream()), "cfm_rowgemm_pack_f32")
    return wp


def _rowchain(pre, core, post, mode, x, d, *, wpre=None, bpre=None, res=None, y1=None, stats=None, ln_eps=0.0, ffn=None, b2=None,
              alpha=0.0, y=None, stats_out=None, ln2=None, wpost=None, bpost=None, cspost=None, post_eps=0.0, z=None):
    rows = x.numel() // d
    ptr = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    wp, b1f, cs1 = ffn if ffn is not None else (None, None, None)
    hidden = 0 if b1f is None else b1f.numel()
    g2, bt2, eps2 = ln2 if ln2 is not None else (None, None, 0.0)
    st = _lib.load().cfm_rowchain_f32(pre, core, post, mode, x.data_ptr(), d, ptr(wpre), ptr(bpre), ptr(res), d, ptr(y1), d,
                                      ptr(stats), 0 if stats is None else stats.shape[1], float(ln_eps), ptr(wp), ptr(b1f), ptr(cs1),
                                      ptr(b2), float(alpha), hidden, ptr(y), d, ptr(stats_out), ptr(g2), ptr(bt2), float(eps2),
                                      ptr(wpost), ptr(bpost), ptr(cspost), float(post_eps), ptr(z), 0 if z is None else z.shape[-1],
                                      rows, d, _stream())
    _lib.check(st, "cfm_rowchain_f32")


def rowchain_ffn_qkv(x, stats, ffn, b2, alpha: float, ffn_eps: float, wqkv_p, bqkv_f, csqkv, att_eps: float):
    """K1 (block.py:19-21): y = x + alpha FFN(x) [LN folded, `stats` of x's rows; ffn = (Wp, b1f, colsum1) of ffn_pack / fold_layernorm]
    and qkv = LN_att(y) @ Wqkv.T + b (LN folded: wqkv_p = rowgemm_pack of the folded (3d, d) weight).  Returns (y, qkv)."""
    x = _req(x, "x"); stats = _req(stats, "ln_stats")
    d = x.shape[-1]
    y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    qkv = torch.empty(*x.shape[:-1], 3 * d, device=x.device, dtype=torch.float32)
    _rowchain(0, 1, 1, 0, x, d, stats=stats, ln_eps=ffn_eps, ffn=ffn, b2=_req(b2, "b2"), alpha=alpha, y=y, wpost=wqkv_p,
              bpost=bqkv_f, cspost=csqkv, post_eps=att_eps, z=qkv)
    return y, qkv


def rowchain_out_glu(ctx, wout_p, bout, res, wglu_p, bglu_f, csglu, conv_eps: float):
    """K2 (block.py:21-23): y2 = ctx @ Wout.T + b + res and g = GLU(LN_conv(y2) @ Wpw1.T + b) (LN folded).  Returns (y2, g)."""
    ctx = _req(ctx, "ctx"); res = _req(res, "residual")
    d = ctx.shape[-1]
    y2 = torch.empty(ctx.shape, device=ctx.device, dtype=torch.float32)
    g = torch.empty(ctx.shape, device=ctx.device, dtype=torch.float32)
    _rowchain(1, 0, 2, 0, ctx, d, wpre=wout_p, bpre=_req(bout, "bias"), res=res, y1=y2, ln_eps=conv_eps, wpost=wglu_p, bpost=bglu_f,
              cspost=csglu, z=g)
    return y2, g


def rowchain_pw2_ffn_ln(c, wpw2_p, bpw2, res, ffn, b2, alpha: float, ffn_eps: float, closing_ln, want_stats: bool = False):
    """K3 (block.py:23-27): y3 = c @ Wpw2.T + b + res; out = LayerNorm(y3 + alpha FFN(y3)) (closing_ln = (weight, bias, eps)).
    Returns (out, statistics (rows, 1, 2) of out or None)."""
    c = _req(c, "c"); res = _req(res, "residual")
    d = c.shape[-1]
    out = torch.empty(c.shape, device=c.device, dtype=torch.float32)
    st = torch.empty(c.numel() // d, 1, 2, device=c.device, dtype=torch.float32) if want_stats else None
    _rowchain(1, 1, 0, 2, c, d, wpre=wpw2_p, bpre=_req(bpw2, "bias"), res=res, ln_eps=ffn_eps, ffn=ffn, b2=_req(b2, "b2"), alpha=alpha,
              y=out, stats_out=st, ln2=(_req(closing_ln[0], "ln.weight"), _req(closing_ln[1], "ln.bias"), float(closing_ln[2])))
    return out, st


def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5,
              out: Optional[torch.Tensor] = None, for_gemm: bool = False, emit_stats: bool = False):
    """for_gemm: the result only feeds GEMM operands -- under autocast it is written in the 16-bit matrix-pipe type (the
    GEMM would round it to that type anyway: identical results, half the bytes).
    emit_stats (fp32): returns (y, stats (rows, 1, 2)) -- the statistics partial of every OUTPUT row for a folded LayerNorm."""
    x = _req(x, "x"); weight = _req(weight, "weight"); bias = _req(bias, "bias")
    d = x.shape[-1]
    rows = x.numel() // d
    if emit_stats:
        y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
        stats = torch.empty(rows, 1, 2, device=x.device, dtype=torch.float32)
        _lib.check(_lib.load().cfm_layernorm_fwd_stats_f32(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                                           stats.data_ptr(), rows, d, eps, _stream()), "cfm_layernorm_fwd_stats_f32")
        return y, stats
    p16 = out16_ok(d) if for_gemm and out is None else 0
    if p16:
        y = torch.empty(x.shape, device=x.device, dtype=_DT16[p16])
        _lib.check(_lib.load().cfm_layernorm_fwd_out16_f32(p16, x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                                           None, None, rows, d, eps, _stream()), "cfm_layernorm_fwd_out16_f32")
        return y
    y = _f32_like(x) if out is None else _req(out, "out")
    if y.shape != x.shape:
        raise _lib.ConformerHipError(f"layernorm: out{tuple(y.shape)} does not match x{tuple(x.shape)}")
    st = _lib.load().cfm_layernorm_fwd_f32(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                           None, None, rows, d, eps, _stream())
    _lib.check(st, "cfm_layernorm_fwd_f32")
    return y


def _gemm_common(a: torch.Tensor, w: torch.Tensor, b: torch.Tensor):
    a = _req(a, "A", a.dtype if isinstance(a, torch.Tensor) and a.dtype in _DT16.values() else torch.float32)
    w = _req(w, "W"); b = _req(b, "bias")
    k = a.shape[-1]
    m = a.numel() // k
    w2 = w.reshape(w.shape[0], -1)
    if w2.shape[1] != k:
        raise _lib.ConformerHipError(f"GEMM shape mismatch: A(...,{k}) vs W{tuple(w.shape)}")
    return a, w2, b, m, w2.shape[0], k


def _splitk(m: int, n: int, k: int) -> int:
    """Contraction slices for the native fp32 GEMM at SMALL M (0 = the ordinary kernel): when the product has fewer 64x64 tiles
    than the chip has CUs and K is long, every wave would run K/2 dependent MFMAs on a mostly idle chip (FFN out of a streaming
    chunk, M = 1280: 160 tiles, 63 us); the split form gives ~2-4 workgroups per CU and sums the partial tiles in a fixed order."""
    tiles = ((m + 63) // 64) * ((n + 63) // 64)
    if tiles > 192 or k < 1024 or n % 4 or m < 1:
        return 0
    return max(2, min(8, k // 512, -(-768 // tiles)))


def _splitk_gemm(epi: int, a, w2, b, c, m, n, k, res=None, alpha: float = 1.0):
    sp = _splitk(m, n, k)
    if not sp or a.dtype != torch.float32 or c.dtype != torch.float32 or _fp32_planes or mfma16_prec():
        return None
    ws = torch.empty(sp * m * n, device=a.device, dtype=torch.float32)
    _lib.check(_lib.load().cfm_gemm_splitk_f32(epi, a.data_ptr(), w2.data_ptr(), b.data_ptr(), _p(res), alpha, c.data_ptr(),
                                               ws.data_ptr(), sp, m, n, k, k, n, n, _stream()), "cfm_gemm_splitk_f32")
    return c


def linear(a, w, b, act: str = "none", for_gemm: bool = False, emit_stats: bool = False):
    """y = act(a @ w.T + b); act in {none, swish, relu}.  for_gemm: see layernorm (y feeds a GEMM of contraction length n).
    emit_stats (native fp32, act none, ln_fold_ok(n)): returns (y, stats (rows, n/32, 2)) for a folded LayerNorm of y."""
    a, w2, b, m, n, k = _gemm_common(a, w, b)
    if emit_stats:
        if act != "none" or not ln_fold_ok(n) or a.dtype != torch.float32:
            raise _lib.ConformerHipError("linear(emit_stats=True) needs the native fp32 path, act='none' and ln_fold_ok(N)")
        c = torch.empty(*a.shape[:-1], n, device=a.device, dtype=torch.float32)
        stats = torch.empty(m, n // 32, 2, device=a.device, dtype=torch.float32)
        _lib.check(_lib.load().cfm_gemm_bias_stats_f32(a.data_ptr(), w2.data_ptr(), b.data_ptr(), c.data_ptr(), stats.data_ptr(),
                                                       m, n, k, k, n, _stream()), "cfm_gemm_bias_stats_f32")
        return c, stats
    prec = mfma16_prec()
    p16 = out16_ok(n) if for_gemm else 0
    c = torch.empty(*a.shape[:-1], n, device=a.device, dtype=_DT16[p16] if p16 else torch.float32)
    if prec:
        return _mfma16_gemm(prec, {"none": 0, "swish": 1, "relu": 2}[act], a, w2, b, c, m, n, k)
    if _split_gemm({"none": 0, "swish": 1, "relu": 2}[act], a, w2, b, c, m, n, k) is not None:
        return c
    if act != "relu" and _splitk_gemm({"none": 0, "swish": 1}[act], a, w2, b, c, m, n, k) is not None:
        return c
    fn = {"none": "cfm_gemm_bias_f32", "swish": "cfm_gemm_bias_swish_f32", "relu": "cfm_gemm_bias_relu_f32"}[act]
    st = getattr(_lib.load(), fn)(a.data_ptr(), w2.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, k, n, _stream())
    _lib.check(st, fn)
    return c


def linear_glu(a, w, b) -> torch.Tensor:
    """(a @ Wv.T + bv) * sigmoid(a @ Wg.T + bg) with W = [Wv; Wg] (pointwise Conv1d(d->2d) + GLU)."""
    a, w2, b, m, n2, k = _gemm_common(a, w, b)
    n = n2 // 2
    c = torch.empty(*a.shape[:-1], n, device=a.device, dtype=torch.float32)
    prec = mfma16_prec()
    if prec:
        return _mfma16_gemm(prec, 3, a, w2, b, c, m, n, k)
    if _split_gemm(3, a, w2, b, c, m, n, k) is not None:
        return c
    st = _lib.load().cfm_gemm_bias_glu_f32(a.data_ptr(), w2.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, k, n,
                                           _stream())
    _lib.check(st, "cfm_gemm_bias_glu_f32")
    return c


def linear_residual(a, w, b, res: torch.Tensor, alpha: float = 1.0, emit_stats: bool = False):
    """alpha * (a @ w.T + b) + res.  emit_stats: see linear."""
    a, w2, b, m, n, k = _gemm_common(a, w, b)
    res = _req(res, "residual")
    if emit_stats:
        if not ln_fold_ok(n) or a.dtype != torch.float32:
            raise _lib.ConformerHipError("linear_residual(emit_stats=True) needs the native fp32 path and ln_fold_ok(N)")
        c = torch.empty(*a.shape[:-1], n, device=a.device, dtype=torch.float32)
        stats = torch.empty(m, n // 32, 2, device=a.device, dtype=torch.float32)
        _lib.check(_lib.load().cfm_gemm_bias_residual_stats_f32(a.data_ptr(), w2.data_ptr(), b.data_ptr(), res.data_ptr(), alpha,
                                                                c.data_ptr(), stats.data_ptr(), m, n, k, k, n, n, _stream()),
                   "cfm_gemm_bias_residual_stats_f32")
        return c, stats
    c = torch.empty(*a.shape[:-1], n, device=a.device, dtype=torch.float32)
    prec = mfma16_prec()
    if prec:
        return _mfma16_gemm(prec, 4, a, w2, b, c, m, n, k, res, alpha)
    if _split_gemm(4, a, w2, b, c, m, n, k, res, alpha) is not None:
        return c
    if _splitk_gemm(4, a, w2, b, c, m, n, k, res, alpha) is not None:
        return c
    st = _lib.load().cfm_gemm_bias_residual_f32(a.data_ptr(), w2.data_ptr(), b.data_ptr(), res.data_ptr(), alpha,
                                                c.data_ptr(), m, n, k, k, n, n, _stream())
    _lib.check(st, "cfm_gemm_bias_residual_f32")
    return c


def relpos_table(div_term: torch.Tensor, t: int) -> torch.Tensor:
    dt = _req(div_term, "div_term").reshape(-1)
    d = 2 * dt.numel()
    pe = torch.empty(2 * t - 1, d, device=dt.device, dtype=torch.float32)
    st = _lib.load().cfm_relpos_table_f32(dt.data_ptr(), pe.data_ptr(), t, d, _stream())
    _lib.check(st, "cfm_relpos_table_f32")
    return pe


def relpos_attention(qkv: torch.Tensor, pos: torch.Tensor, u: torch.Tensor, v: torch.Tensor,
                     lengths: Optional[torch.Tensor], n_heads: int, for_gemm: bool = False) -> torch.Tensor:
    """qkv: (B,T,3d) fused projections [q|k|v]; pos: (2T-1,d) projected table; returns ctx (B,T,d).
    for_gemm: the context only feeds the out-projection GEMM: under autocast it is written in the 16-bit type (that GEMM rounds
    an fp32 context to it anyway: bit-identical layer output, half the bytes, the GEMM's 16-bit-A staging)."""
    u = _req(u, "content_bias"); v = _req(v, "position_bias")
    prec = mfma16_prec()
    q16 = bool(prec) and isinstance(qkv, torch.Tensor) and qkv.dtype == _DT16[prec]      # autocast inference: 16-bit projections
    if q16:
        if not (qkv.is_cuda and qkv.is_contiguous() and qkv.shape[-1] % 24 == 0):
            raise _lib.ConformerHipError("16-bit qkv: expected a contiguous HIP tensor with d % 8 == 0")
    else:
        qkv = _req(qkv, "qkv")
    B, T, d3 = qkv.shape
    d = d3 // 3
    dh = d // n_heads
    if not (isinstance(pos, torch.Tensor) and pos.is_cuda and pos.dtype == torch.float32 and pos.dim() == 2
            and pos.stride(1) == 1 and pos.shape == (2 * T - 1, d)):
        raise _lib.ConformerHipError(f"pos: expected a ({2 * T - 1},{d}) fp32 HIP tensor with unit column stride")
    ldp = pos.stride(0)
    if lengths is not None:
        lengths = _req(lengths, "lengths", torch.int64)
    base = qkv.data_ptr()
    c16 = bool(prec and for_gemm and d % 8 == 0)
    if q16 or c16:
        esz = 2 if q16 else 4
        ctx = torch.empty(B, T, d, device=qkv.device, dtype=_DT16[prec] if c16 else torch.float32)
        st = _lib.load().cfm_relpos_attention_io16_mfma16_f32(prec, base, base + esz * d, base + 2 * esz * d, int(q16), d3,
                                                              pos.data_ptr(), ldp, u.data_ptr(), v.data_ptr(), _p(lengths),
                                                              ctx.data_ptr(), int(c16), d, B, T, n_heads, dh, _stream())
        _lib.check(st, "cfm_relpos_attention_io16_mfma16_f32")
        return ctx
    ctx = torch.empty(B, T, d, device=qkv.device, dtype=torch.float32)
    if prec:
        st = _lib.load().cfm_relpos_attention_mfma16_f32(prec, base, base + 4 * d, base + 8 * d, d3, pos.data_ptr(), ldp,
                                                         u.data_ptr(), v.data_ptr(), _p(lengths), ctx.data_ptr(), d, None,
                                                         B, T, n_heads, dh, 0.0, 0, _stream())
        _lib.check(st, "cfm_relpos_attention_mfma16_f32")
        return ctx
    st = _lib.load().cfm_relpos_attention_fwd_f32(base, base + 4 * d, base + 8 * d, d3, pos.data_ptr(), ldp,
                                                  u.data_ptr(), v.data_ptr(), _p(lengths), ctx.data_ptr(), d, None,
                                                  B, T, n_heads, dh, _stream())
    _lib.check(st, "cfm_relpos_attention_fwd_f32")
    return ctx


def relpos_attention_rows(qkv: torch.Tensor, pos: torch.Tensor, u: torch.Tensor, v: torch.Tensor, lengths: torch.Tensor,
                          n_heads: int, q_begin: int, q_count: int, ctx: torch.Tensor, keys_hint: Optional[int] = None
                          ) -> torch.Tensor:
    """Incremental attention over a K/V cache: qkv (B,Tmax,3d) holds the projections of every frame seen so far (rows
    beyond `lengths` are ignored, every length >= 1), pos the projected (2Tmax-1,d) table; only query rows
    [q_begin, q_begin+q_count) are computed, into the same rows of `ctx` (B,Tmax,d).  keys_hint (host-side upper bound of
    `lengths`, default Tmax) picks the key split: few query rows against a long cache would otherwise occupy
    B*H*ceil(q_count/128) workgroups only.  Under torch.autocast the 16-bit form runs (cfm_relpos_attention_rows_mfma16_f32)."""
    qkv = _req(qkv, "qkv"); u = _req(u, "content_bias"); v = _req(v, "position_bias"); ctx = _req(ctx, "ctx")
    B, T, d3 = qkv.shape
    d = d3 // 3
    if not (pos.is_cuda and pos.dtype == torch.float32 and pos.dim() == 2 and pos.stride(1) == 1 and pos.shape == (2 * T - 1, d)):
        raise _lib.ConformerHipError(f"pos: expected a ({2 * T - 1},{d}) fp32 HIP tensor with unit column stride")
    if ctx.shape != (B, T, d) or not qkv.is_contiguous() or not ctx.is_contiguous():
        raise _lib.ConformerHipError("relpos_attention_rows: qkv (B,T,3d) and ctx (B,T,d) must be contiguous cache buffers")
    lengths = _req(lengths, "lengths", torch.int64)
    prec = mfma16_prec()
    if prec != PREC_F32:
        # under autocast: the 16-bit matrix-pipe form (fp32 cache, operands rounded where they enter a product; no key split)
        base = qkv.data_ptr()
        st = _lib.load().cfm_relpos_attention_rows_mfma16_f32(prec, base, base + 4 * d, base + 8 * d, d3, pos.data_ptr(), pos.stride(0),
                                                              u.data_ptr(), v.data_ptr(), lengths.data_ptr(), ctx.data_ptr(), d, B, T,
                                                              n_heads, d // n_heads, int(q_begin), int(q_count), _stream())
        _lib.check(st, "cfm_relpos_attention_rows_mfma16_f32")
        return ctx
    tiles = ((T if keys_hint is None else min(int(keys_hint), T)) + 31) // 32
    blocks = B * n_heads * ((q_count + 127) // 128)
    nsplit = max(1, min(16, tiles // 8, -(-1024 // blocks)))          # >= 8 key tiles per split, aim at ~1024 workgroups
    ws = torch.empty(nsplit * B * q_count * (d + n_heads), device=qkv.device, dtype=torch.float32) if nsplit > 1 else None
    base = qkv.data_ptr()
    st = _lib.load().cfm_relpos_attention_rows_f32(base, base + 4 * d, base + 8 * d, d3, pos.data_ptr(), pos.stride(0),
                                                   u.data_ptr(), v.data_ptr(), lengths.data_ptr(), ctx.data_ptr(), d, B, T,
                                                   n_heads, d // n_heads, int(q_begin), int(q_count), nsplit, _p(ws), _stream())
    _lib.check(st, "cfm_relpos_attention_rows_f32")
    return ctx


def dwconv_bn_swish(g, w, b, bn_w, bn_b, bn_mean, bn_var, eps: float = 1e-5, for_gemm: bool = False) -> torch.Tensor:
    """for_gemm: see layernorm (the result feeds the pointwise_conv_2 GEMM of contraction length C)."""
    g = _req(g, "g"); w = _req(w, "dw weight"); b = _req(b, "dw bias")
    B, T, C = g.shape
    K = w.shape[-1]
    p16 = out16_ok(C) if for_gemm and K in (3, 7, 15, 31) else 0
    if p16:
        y = torch.empty(g.shape, device=g.device, dtype=_DT16[p16])
        st = _lib.load().cfm_dwconv_bn_swish_fwd_out16_f32(p16, g.data_ptr(), w.data_ptr(), b.data_ptr(), _req(bn_w, "bn_w").data_ptr(),
                                                           _req(bn_b, "bn_b").data_ptr(), _req(bn_mean, "bn_mean").data_ptr(),
                                                           _req(bn_var, "bn_var").data_ptr(), eps, y.data_ptr(), B, T, C, K,
                                                           _stream())
        _lib.check(st, "cfm_dwconv_bn_swish_fwd_out16_f32")
        return y
    y = _f32_like(g)
    st = _lib.load().cfm_dwconv_bn_swish_fwd_f32(g.data_ptr(), w.data_ptr(), b.data_ptr(), _req(bn_w, "bn_w").data_ptr(),
                                                 _req(bn_b, "bn_b").data_ptr(), _req(bn_mean, "bn_mean").data_ptr(),
                                                 _req(bn_var, "bn_var").data_ptr(), eps, y.data_ptr(), B, T, C, K,
                                                 _stream())
    _lib.check(st, "cfm_dwconv_bn_swish_fwd_f32")
    return y


def pack_conv2_weight(w2: torch.Tensor) -> torch.Tensor:
    w2 = _req(w2, "conv_2.weight")
    C = w2.shape[0]
    out = torch.empty(C, 9 * C, device=w2.device, dtype=torch.float32)
    _lib.check(_lib.load().cfm_pack_conv2_weight_f32(w2.data_ptr(), out.data_ptr(), C, _stream()), "cfm_pack_conv2_weight_f32")
    return out


def pack_linear_weight(wl: torch.Tensor, C: int, F2: int) -> torch.Tensor:
    wl = _req(wl, "linear.weight")
    out = _f32_like(wl)
    _lib.check(_lib.load().cfm_pack_linear_weight_f32(wl.data_ptr(), out.data_ptr(), wl.shape[0], C, F2, _stream()),
               "cfm_pack_linear_weight_f32")
    return out


def subsample_stem(x: torch.Tensor, w1, b1, w2p, b2) -> torch.Tensor:
    """x (B,F,T) -> h2 (B, T2, F2*C) channel-last flattening [f][c] (pair with pack_linear_weight)."""
    x = _req(x, "x"); w1 = _req(w1, "conv_1.weight"); b1 = _req(b1, "conv_1.bias")
    w2p = _req(w2p, "packed conv_2.weight"); b2 = _req(b2, "conv_2.bias")
    B, F, T = x.shape
    C = w1.shape[0]
    F1, T1 = (F - 1) // 2, (T - 1) // 2
    F2, T2 = (F1 - 1) // 2, (T1 - 1) // 2
    lib = _lib.load()
    p16 = out16_ok(C) if C % 64 == 0 else 0       # inference under autocast: h1 and h2 only feed 16-bit GEMM operands
    if p16:
        h1 = torch.empty(B, T1, F1, C, device=x.device, dtype=_DT16[p16])
        _lib.check(lib.cfm_subsample_conv1_relu_out16_f32(p16, x.data_ptr(), w1.data_ptr(), b1.data_ptr(), h1.data_ptr(), B, F,
                                                          T, C, _stream()), "cfm_subsample_conv1_relu_out16_f32")
        h2 = torch.empty(B, T2, F2 * C, device=x.device, dtype=_DT16[p16])
    else:
        h1 = torch.empty(B, T1, F1, C, device=x.device, dtype=torch.float32)
        _lib.check(lib.cfm_subsample_conv1_relu_f32(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), h1.data_ptr(), B, F, T, C,
                                                    _stream()), "cfm_subsample_conv1_relu_f32")
        h2 = torch.empty(B, T2, F2 * C, device=x.device, dtype=torch.float32)
    _conv2_relu(lib, h1, w2p, b2, h2, B, F1, T1, C)
    return h2


def _conv2_relu(lib, h1, w2p, b2, h2, B, F1, T1, C):
    prec = mfma16_prec()
    if prec and C % 64 == 0:
        w16 = weight16(w2p, prec)
        _lib.check(lib.cfm_subsample_conv2_relu_mfma16_f32(prec, h1.data_ptr(), int(h1.dtype != torch.float32),
                                                           (w2p if w16 is None else w16).data_ptr(), int(w16 is not None),
                                                           b2.data_ptr(), h2.data_ptr(), int(h2.dtype != torch.float32), B, F1,
                                                           T1, C, _stream()), "cfm_subsample_conv2_relu_mfma16_f32")
    elif _fp32_planes and C % 64 == 0:
        ws = weight_split(w2p.view(C, 9 * C), _fp32_planes)
        _lib.check(lib.cfm_subsample_conv2_relu_split_bf16_f32(_fp32_planes, h1.data_ptr(), ws.data_ptr(), b2.data_ptr(),
                                                               h2.data_ptr(), B, F1, T1, C, _stream()),
                   "cfm_subsample_conv2_relu_split_bf16_f32")
    else:
        _lib.check(lib.cfm_subsample_conv2_relu_f32(h1.data_ptr(), w2p.data_ptr(), b2.data_ptr(), h2.data_ptr(), B, F1, T1,
                                                    C, _stream()), "cfm_subsample_conv2_relu_f32")


# ======================================================================================================
# training forward variants + backward ops (fp32).  Same rules: HIP tensors only, no eager fallback.
# ======================================================================================================
_ZERO_ARENA = {}           # (device, dtype) -> [zero-filled chunk, elements already handed out]
_ZERO_CHUNK = 16 << 20     # elements per chunk (64 MiB of fp32): one fill kernel instead of one per gradient tensor


def _zeros_split(device, dtype, *shapes):
    """Zero-filled, 16-byte-aligned tensors for the accumulate-with-atomics outputs (weight / bias gradients) of one
    backward call.  They are carved out of a zero-filled CHUNK that is shared by consecutive calls: a training step used to
    issue ~400 tiny ATen fill kernels (one per call: 2 % of the bf16 step); now it issues one 64 MiB fill per ~16 M gradient
    elements.  Every slice is handed out exactly once, so nobody else ever writes to it; a chunk's memory goes back to the
    caching allocator when the last gradient living in it is freed (zero_grad(set_to_none=True) / the optimizer's step)."""
    sizes = [int(torch.Size(sh).numel()) for sh in shapes]
    padded = [(n + 3) // 4 * 4 for n in sizes]
    total = sum(padded)
    if total >= _ZERO_CHUNK // 4 or torch.cuda.is_current_stream_capturing():
        buf, off = torch.zeros(total, device=device, dtype=dtype), 0       # big requests (dqkv, ...) get their own fill
    else:
        key = (device, dtype, torch.cuda.current_stream(device).cuda_stream)   # (a chunk is zero-filled on the stream that made it:
        ent = _ZERO_ARENA.get(key)                                                #  slices are only handed to work on that stream)
        if ent is None or ent[1] + total > ent[0].numel():
            ent = [torch.zeros(_ZERO_CHUNK, device=device, dtype=dtype), 0]
            _ZERO_ARENA[key] = ent
        buf, off = ent[0], ent[1]
        ent[1] = off + total
    outs = []
    for sh, n, pn in zip(shapes, sizes, padded):
        outs.append(buf[off:off + n].view(sh))
        off += pn
    return outs


def layernorm_train(x, weight, bias, eps: float = 1e-5, for_gemm: bool = False):
    """LayerNorm forward that also returns the per-row mean / rstd the backward needs (for_gemm: see layernorm)."""
    x = _req(x, "x"); weight = _req(weight, "weight"); bias = _req(bias, "bias")
    d = x.shape[-1]
    rows = x.numel() // d
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    p16 = out16_ok(d) if for_gemm else 0
    if p16:
        y = torch.empty(x.shape, device=x.device, dtype=_DT16[p16])
        _lib.check(_lib.load().cfm_layernorm_fwd_out16_f32(p16, x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                                           mean.data_ptr(), rstd.data_ptr(), rows, d, eps, _stream()),
                   "cfm_layernorm_fwd_out16_f32")
        return y, mean, rstd
    y = _f32_like(x)
    st = _lib.load().cfm_layernorm_fwd_f32(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                           mean.data_ptr(), rstd.data_ptr(), rows, d, eps, _stream())
    _lib.check(st, "cfm_layernorm_fwd_f32")
    return y, mean, rstd


def layernorm_bwd(x, weight, dy, mean, rstd, dres=None):
    """Returns (dx [+ dres], dweight, dbias)."""
    x = _req(x, "x"); dy = _req(dy, "dy"); weight = _req(weight, "weight")
    d = x.shape[-1]
    rows = x.numel() // d
    lib = _lib.load()
    dx = _f32_like(x)
    if dres is not None:
        dres = _req(dres, "dres")
    dw, db = _zeros_split(x.device, x.dtype, (d,), (d,))
    if d <= 2048:                                  # one pass over x and dy: input and parameter gradients together
        nws = int(lib.cfm_layernorm_bwd_workspace_bytes(rows, d))
        ws = torch.empty(nws // 4, device=x.device, dtype=torch.float32)
        _lib.check(lib.cfm_layernorm_bwd_f32(x.data_ptr(), weight.data_ptr(), dy.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                             _p(dres), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), rows, d, ws.data_ptr(), nws,
                                             _stream()),
                   "cfm_layernorm_bwd_f32")
        return dx, dw, db
    _lib.check(lib.cfm_layernorm_bwd_dx_f32(x.data_ptr(), weight.data_ptr(), dy.data_ptr(), mean.data_ptr(),
                                            rstd.data_ptr(), _p(dres), dx.data_ptr(), rows, d, _stream()),
               "cfm_layernorm_bwd_dx_f32")
    _lib.check(lib.cfm_layernorm_bwd_params_f32(x.data_ptr(), dy.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                dw.data_ptr(), db.data_ptr(), rows, d, _stream()),
               "cfm_layernorm_bwd_params_f32")
    return dx, dw, db


def linear_swish_save(a, w, b):
    """Returns (swish(z), z) with z = a @ w.T + b."""
    a, w2, b, m, n, k = _gemm_common(a, w, b)
    c = torch.empty(*a.shape[:-1], n, device=a.device, dtype=torch.float32)
    z = _f32_like(c)
    st = _lib.load().cfm_gemm_bias_swish_save_f32(a.data_ptr(), w2.data_ptr(), b.data_ptr(), c.data_ptr(), z.data_ptr(),
                                                  m, n, k, k, n, _stream())
    _lib.check(st, "cfm_gemm_bias_swish_save_f32")
    return c, z


_SEED_GEN = {}


def _mix64(*vals: int) -> int:
    z = 0x243F6A8885A308D3
    for v in vals:
        z = ((z ^ (v & 0xFFFFFFFFFFFFFFFF)) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z ^= z >> 29
        z = (z * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z ^= z >> 32
    return z & 0x3FFFFFFFFFFFFFFF


def _rank() -> int:
    import os
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_rank()
    return int(os.environ.get("RANK", "0"))


def new_seeds(n: int):
    """n dropout seeds (one per fused-dropout site and step).  They come from the (seed, offset) state of the current
    device's default generator -- the state stock dropout kernels consume -- mixed with the data-parallel rank: the offset
    advances by 4 per seed, torch.manual_seed() rewinds it (a re-seeded run replays its masks), replicas seeded alike draw
    DIFFERENT masks, and the default CPU generator (SpecAugment's band draws) is neither consumed nor consulted."""
    if torch.cuda.is_available():
        g = torch.cuda.default_generators[torch.cuda.current_device()]
        seed, off = g.initial_seed(), g.get_offset()
        g.set_offset(off + 4 * n)
        return [_mix64(seed, _rank(), off + 4 * i) for i in range(n)]
    key = (torch.initial_seed(), _rank())                     # GPU-less hosts (host-logic tests only): a private counter
    if _SEED_GEN.get("key") != key:
        _SEED_GEN["key"], _SEED_GEN["off"] = key, 0
    off = _SEED_GEN["off"]
    _SEED_GEN["off"] = off + 4 * n
    return [_mix64(key[0], key[1], off + 4 * i) for i in range(n)]


def linear_train(epi: str, a, w, b, *, residual=None, alpha: float = 1.0, drop_p: float = 0.0, seed: int = 0,
                 save_z: bool = False, for_gemm: bool = False):
    """Training forward GEMM with dropout fused in the epilogue.  epi: 'bias' | 'swish' | 'residual'.
    Returns C (and Z, the pre-activation, when save_z).  for_gemm: see layernorm."""
    a, w2, b, m, n, k = _gemm_common(a, w, b)
    p16 = out16_ok(n) if for_gemm else 0
    c = torch.empty(*a.shape[:-1], n, device=a.device, dtype=_DT16[p16] if p16 else torch.float32)
    # (the saved pre-activation is kept in the 16-bit type together with the activation: what autocast keeps for silu's backward)
    z = torch.empty(*a.shape[:-1], n, device=a.device, dtype=_DT16[p16] if p16 and epi == "swish" else torch.float32) if save_z else None
    code = {"bias": 0, "swish": 1, "residual": 4}[epi]
    if residual is not None:
        residual = _req(residual, "residual")
    prec = mfma16_prec()
    if prec:
        _mfma16_gemm(prec, code, a, w2, b, c, m, n, k, residual, alpha, z, drop_p, seed)
        return (c, z) if save_z else c
    st = _lib.load().cfm_gemm_train_f32(code, a.data_ptr(), w2.data_ptr(), b.data_ptr(), _p(residual), alpha, c.data_ptr(),
                                        _p(z), m, n, k, k, n, n, float(drop_p), int(seed), _stream())
    _lib.check(st, "cfm_gemm_train_f32")
    return (c, z) if save_z else c


def dropout_apply(x, drop_p: float, seed: int, for_gemm: bool = False):
    """x * mask(seed, flat index) -- the same mask cfm_gemm_train_f32 applied to a contiguous (M,N) result.
    for_gemm: the result only feeds GEMM operands (a masked gradient going into dX / dW products): under a 16-bit precision
    mode it is written in that type -- what those kernels would round it to anyway; with drop_p = 0 this is the plain cast that
    lets them run their all-16-bit forms."""
    x = _req(x, "x")
    prec = mfma16_prec() if for_gemm else 0
    if prec and x.numel() % 8 == 0 and x.shape[-1] % 8 == 0:
        y = torch.empty(x.shape, device=x.device, dtype=_DT16[prec])
        _lib.check(_lib.load().cfm_dropout_out16_f32(prec, x.data_ptr(), y.data_ptr(), x.numel(), float(max(drop_p, 0.0)), int(seed),
                                                     _stream()), "cfm_dropout_out16_f32")
        return y
    if drop_p <= 0.0:
        return x
    y = _f32_like(x)
    _lib.check(_lib.load().cfm_dropout_f32(x.data_ptr(), y.data_ptr(), x.numel(), float(drop_p), int(seed), _stream()),
               "cfm_dropout_f32")
    return y


def glu_fwd(z):
    z = _req(z, "z")
    n = z.shape[-1] // 2
    rows = z.numel() // (2 * n)
    y = torch.empty(*z.shape[:-1], n, device=z.device, dtype=torch.float32)
    _lib.check(_lib.load().cfm_glu_fwd_f32(z.data_ptr(), y.data_ptr(), rows, n, _stream()), "cfm_glu_fwd_f32")
    return y


def glu_bwd(z, dy, for_gemm: bool = False):
    """for_gemm: dz only feeds the pointwise_conv_1 gradient GEMMs: under autocast it is written in the 16-bit type."""
    z = _req(z, "z"); dy = _req(dy, "dy")
    n = dy.shape[-1]
    rows = dy.numel() // n
    # (the consumer is linear_bwd(h0, w1 (2n, n), dz): its all-16-bit kernels need BOTH dimensions of w1 to be multiples of 8)
    p16 = out16_ok(2 * n) if for_gemm and n % 8 == 0 else 0
    if p16:
        dz = torch.empty(z.shape, device=z.device, dtype=_DT16[p16])
        _lib.check(_lib.load().cfm_glu_bwd_out16_f32(p16, z.data_ptr(), dy.data_ptr(), dz.data_ptr(), rows, n, _stream()),
                   "cfm_glu_bwd_out16_f32")
        return dz
    dz = _f32_like(z)
    _lib.check(_lib.load().cfm_glu_bwd_f32(z.data_ptr(), dy.data_ptr(), dz.data_ptr(), rows, n, _stream()),
               "cfm_glu_bwd_f32")
    return dz


def colsum(x2d, alpha: float = 1.0, rows=None, cols=None, ld=None, out=None):
    """alpha * column sums of a (rows, cols) fp32 matrix with row stride ld (bias gradients); `out` must be zeroed."""
    rows = x2d.shape[0] if rows is None else rows
    cols = x2d.shape[1] if cols is None else cols
    ld = x2d.stride(0) if ld is None else ld
    if out is None:
        out = torch.zeros(cols, device=x2d.device, dtype=x2d.dtype)
    _lib.check(_lib.load().cfm_colsum_f32(x2d.data_ptr(), ld, rows, cols, alpha, out.data_ptr(), _stream()),
               "cfm_colsum_f32")
    return out


def gemm_bwd(A, a_col: bool, B, b_col: bool, I: int, J: int, Kc: int, *, alpha: float = 1.0, Z=None, out=None,
             lda=None, ldb=None, ldc=None, ldz=None, allow_split: bool = False, accumulate: bool = False,
             nbatch: int = 1, nb1: int = 1, sa=(0, 0), sb=(0, 0), sc=(0, 0), a_ptr=None, b_ptr=None, c_ptr=None,
             drop_p: float = 0.0, drop_seed: int = 0, prec: int = 0, b16: bool = False, pad4: bool = False,
             c16: bool = False):
    """C (I,J) (+)= alpha * sum_k A(i,k) B(j,k) [* swish'(Z)]; *_col selects the contraction-major layout.
    Pointers default to the tensors' data_ptr(); explicit *_ptr / ld* let callers address sub-blocks (head slices).
    prec: PREC_F32 (fp32 MFMA) | PREC_BF16 | PREC_FP16 (operands rounded while staged, fp32 accumulate).
    b16: B is a tensor already stored in the 16-bit type of `prec` (contraction-major only; ldb in elements).
    pad4 (16-bit kernels): ragged Kc / I / J are physically padded to a multiple of 4 with zeros (fast load path).
    c16 (16-bit kernels, with Z): C is written in the 16-bit type of `prec` (a gradient that only feeds GEMM operands)."""
    lda = A.stride(-2) if lda is None else lda
    ldb = B.stride(-2) if ldb is None else ldb
    if out is None:
        out = (torch.zeros if allow_split else torch.empty)(I, J, device=A.device, dtype=_DT16[prec] if c16 else A.dtype)
    ldc = out.stride(-2) if ldc is None else ldc
    ldz = 0 if Z is None else (Z.stride(-2) if ldz is None else ldz)
    head = (A.data_ptr() if a_ptr is None else a_ptr, int(a_col), lda, B.data_ptr() if b_ptr is None else b_ptr, int(b_col))
    z16 = Z is not None and Z.dtype != torch.float32
    tail = (ldb, _p(Z), int(z16), ldz, alpha, out.data_ptr() if c_ptr is None else c_ptr, ldc, int(c16), I, J, Kc, int(allow_split),
            int(accumulate), nbatch, nb1, sa[0], sa[1], sb[0], sb[1], sc[0], sc[1], float(drop_p), int(drop_seed))
    if prec:
        _lib.check(_lib.load().cfm_gemm_bwd_batched_mfma16_f32(prec, *head, int(b16), *tail, int(pad4), _stream()),
                   "cfm_gemm_bwd_batched_mfma16_f32")
    else:
        if b16 or c16 or z16:
            raise _lib.ConformerHipError("16-bit operands / results need a 16-bit precision mode")
        _lib.check(_lib.load().cfm_gemm_bwd_batched_f32(*head, *tail[:2], *tail[3:7], *tail[8:], _stream()), "cfm_gemm_bwd_batched_f32")
    return out


def linear_bwd(x2d, w, dy2d, *, alpha: float = 1.0, Z=None, need_dx: bool = True, drop_p: float = 0.0, drop_seed: int = 0,
               dx16: bool = False):
    """Backward of y = x @ w.T + b for 2-D views: returns (dx or None, dw, db), all scaled by alpha;
    dx is additionally multiplied by swish'(Z) when Z is given (then it is d/d(pre-activation)).
    dx16 (with Z, under a 16-bit precision mode): dx only feeds GEMM operands (the next layer's dX / dW products round it to
    the 16-bit type anyway) and is written in that type -- identical results, half the bytes on three passes over it.
    dy2d may itself be such a 16-bit gradient."""
    m, k = x2d.shape
    n = w.shape[0]
    w2 = w.reshape(n, -1)
    dx = None
    prec = mfma16_prec()
    dy16 = dy2d.dtype != torch.float32
    z_ok = Z is not None and n % 8 == 0 and k % 8 == 0 and Z.stride(0) % 8 == 0      # the swish' product on the forward kernel
    if dy16 and not (prec and dy2d.dtype == _DT16[prec]):
        raise _lib.ConformerHipError("a 16-bit dY needs the matching 16-bit precision mode")
    if dy16 and not (n % 8 == 0 and k % 8 == 0 and (z_ok or (Z is None and alpha == 1.0))):
        # a producer wrote its gradient in the 16-bit type for a consumer whose shape the aligned 16-bit kernels do not cover
        # (d % 8 == 4): widen it and take the general kernels rather than refuse (off the tuned shapes; one stock cast)
        dy2d, dy16 = dy2d.float(), False
    if dy2d.stride(0) & 3:                        # e.g. the vocabulary projection (N = 370): rows must start 16-byte aligned
        padded = torch.zeros(m, (n + 3) // 4 * 4, device=dy2d.device, dtype=dy2d.dtype)
        padded[:, :n] = dy2d
        dy2d = padded[:, :n]
    if need_dx and dy16 and Z is None:
        # dX = dY.W with a 16-bit dY: the forward kernel (row-major 16-bit A operand) on the cached transposed 16-bit weight
        wt16 = weight16(w2, prec, transposed=True)
        dx = torch.empty(m, k, device=dy2d.device, dtype=torch.float32)
        _lib.check(_lib.load().cfm_gemm_mfma16_f32(prec, 0, dy2d.data_ptr(), 1, wt16.data_ptr(), 1, _zero_bias(k, dy2d.device).data_ptr(),
                                                   None, 1.0, dx.data_ptr(), 0, None, 0, m, k, n, dy2d.stride(0), k, k, 0.0, 0, _stream()),
                   "cfm_gemm_mfma16_f32")
    elif need_dx and prec and z_ok:
        # d(pre-activation) = alpha * (dY.W) * swish'(Z) [* dropout mask]: forward kernel (big tiles, deep prefetch) on the
        # cached transposed 16-bit weight, swish' fused in its row-major epilogue
        wt16 = weight16(w2, prec, transposed=True)
        dx = torch.empty(m, k, device=dy2d.device, dtype=_DT16[prec] if dx16 else torch.float32)
        _lib.check(_lib.load().cfm_gemm_mfma16_f32(prec, 5, dy2d.data_ptr(), int(dy16), wt16.data_ptr(), 1, None, None, alpha, dx.data_ptr(),
                                                   int(dx16), Z.data_ptr(), int(Z.dtype != torch.float32), m, k, n, dy2d.stride(0),
                                                   Z.stride(0), k, float(drop_p), int(drop_seed), _stream()), "cfm_gemm_mfma16_f32")
    elif need_dx:
        w16 = weight16(w2, prec) if prec else None              # the cast the forward made (same parameter version)
        # (a 16-bit dx feeds linear_bwd of the layer below, whose weight is (k, n'): it takes a 16-bit dY only when both k and
        # its own contraction length are multiples of 8 -- for the FFN that length is this layer's n)
        c16 = bool(dx16 and prec and Z is not None and k % 8 == 0 and n % 8 == 0)
        dx = gemm_bwd(dy2d, False, w2 if w16 is None else w16, True, m, k, n, alpha=alpha, Z=Z, drop_p=drop_p,
                      drop_seed=drop_seed, prec=prec, b16=w16 is not None, c16=c16)
    dw, db = _zeros_split(dy2d.device, torch.float32, (n, k), (n,))
    x16 = x2d.dtype != torch.float32                  # stored in the 16-bit type by its producer (for_gemm)
    if (prec and (x16 or dy16) and n % 8 == 0 and k % 8 == 0 and x2d.stride(1) == 1 and dy2d.stride(1) == 1
            and x2d.stride(0) % (8 if x16 else 4) == 0 and dy2d.stride(0) % (8 if dy16 else 4) == 0
            and x2d.data_ptr() % 16 == 0 and dy2d.data_ptr() % 16 == 0):
        # weight and bias gradient in one kernel (gemm_dw16_impl.h): dY is read once
        _lib.check(_lib.load().cfm_linear_bwd_weight_mfma16_f32(prec, dy2d.data_ptr(), int(dy16), dy2d.stride(0), x2d.data_ptr(),
                                                                int(x16), x2d.stride(0), dw.data_ptr(), k, db.data_ptr(), n, k, m,
                                                                alpha, _stream()), "cfm_linear_bwd_weight_mfma16_f32")
    else:
        if dy16:
            raise _lib.ConformerHipError("a 16-bit dY needs the aligned 16-bit weight-gradient kernel (N % 8 == 0, K % 8 == 0)")
        gemm_bwd(dy2d, True, x2d, True, n, k, m, alpha=alpha, allow_split=True, out=dw, prec=prec, b16=x16)
        colsum(dy2d, alpha, out=db)
    return dx, dw.view_as(w), db


def dwconv_bn_batch_stats(g, w, b, running_mean, running_var, momentum: float = 0.1):
    """Train-mode BatchNorm statistics of dwconv(g)+b: returns (batch_mean, batch_var_biased) and updates the running
    buffers in place (momentum 0.1, unbiased variance), as nn.BatchNorm1d does in .train()."""
    g = _req(g, "g")
    B, T, C = g.shape
    K = w.shape[-1]
    lib = _lib.load()
    nws = int(lib.cfm_dwconv_bn_stats_workspace_bytes(B, T, C))
    buf = torch.empty(2 * C + (nws + 3) // 4, device=g.device, dtype=torch.float32)     # mean | var | per-workgroup partials
    mean, var, ws = buf[:C], buf[C:2 * C], buf[2 * C:]
    st = lib.cfm_dwconv_bn_stats_f32(g.data_ptr(), w.data_ptr(), b.data_ptr(), mean.data_ptr(), var.data_ptr(),
                                     _p(running_mean), _p(running_var), momentum, B, T, C, K, ws.data_ptr(), nws, _stream())
    _lib.check(st, "cfm_dwconv_bn_stats_f32")
    return mean, var


def dwconv_bn_swish_bwd(g, dy, w, b, bn_w, bn_b, bn_mean, bn_var, eps: float = 1e-5, train_stats: bool = False):
    """Returns (dg, dw, db, dbn_weight, dbn_bias); train_stats: bn_mean/bn_var are batch statistics."""
    g = _req(g, "g"); dy = _req(dy, "dy")
    B, T, C = g.shape
    K = w.shape[-1]
    dc = _f32_like(g)
    dg = _f32_like(g)
    dw, db, dga, dbe = _zeros_split(g.device, g.dtype, tuple(w.shape), (C,), (C,), (C,))
    st = _lib.load().cfm_dwconv_bn_swish_bwd_f32(g.data_ptr(), dy.data_ptr(), w.data_ptr(), b.data_ptr(), bn_w.data_ptr(),
                                                 bn_b.data_ptr(), bn_mean.data_ptr(), bn_var.data_ptr(), eps,
                                                 int(train_stats), dc.data_ptr(), dg.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                                 dga.data_ptr(), dbe.data_ptr(), B, T, C, K, _stream())
    _lib.check(st, "cfm_dwconv_bn_swish_bwd_f32")
    return dg, dw, db, dga, dbe


def relpos_attention_train(qkv, pos, u, v, lengths, n_heads, drop_p: float = 0.0, seed: int = 0):
    """Forward that also returns the per-row log-sum-exp (B,H,T) for the backward; optional weight dropout."""
    u = _req(u, "content_bias"); v = _req(v, "position_bias")
    qkv = _req(qkv, "qkv")       # fp32 projections only: the training kernels (forward with log-sum-exp, flash backward) read fp32 q|k|v
    B, T, d3 = qkv.shape
    d = d3 // 3
    dh = d // n_heads
    ctx = torch.empty(B, T, d, device=qkv.device, dtype=torch.float32)
    lse = torch.empty(B, n_heads, T, device=qkv.device, dtype=torch.float32)
    base = qkv.data_ptr()
    prec = mfma16_prec()
    if prec:
        st = _lib.load().cfm_relpos_attention_mfma16_f32(prec, base, base + 4 * d, base + 8 * d, d3, pos.data_ptr(),
                                                         pos.stride(0), u.data_ptr(), v.data_ptr(), _p(lengths),
                                                         ctx.data_ptr(), d, lse.data_ptr(), B, T, n_heads, dh, float(drop_p),
                                                         int(seed), _stream())
        _lib.check(st, "cfm_relpos_attention_mfma16_f32")
        return ctx, lse
    st = _lib.load().cfm_relpos_attention_train_f32(base, base + 4 * d, base + 8 * d, d3, pos.data_ptr(), pos.stride(0),
                                                    u.data_ptr(), v.data_ptr(), _p(lengths), ctx.data_ptr(), d,
                                                    lse.data_ptr(), B, T, n_heads, dh, float(drop_p), int(seed), _stream())
    _lib.check(st, "cfm_relpos_attention_train_f32")
    return ctx, lse


def relpos_attention_bwd(qkv, pos, u, v, lengths, n_heads, ctx, lse, dctx, drop_p: float = 0.0, seed: int = 0):
    """Backward of the attention core.  Returns (dqkv (B,T,3d), dpos (2T-1,d), du (H,dh), dv (H,dh)).
    ONE fused flash-style kernel (attention_bwd_flash_f32.hip) that recomputes score tiles from the forward's
    log-sum-exp: no (B,H,T,T) / (H,B,T,2T-1) tensor is ever allocated (the reference's autograd keeps both,
    attention.py:49-70).  Under autocast: the same algorithm with every product on the 16-bit matrix pipe
    (attention_bwd_flash_mfma16.hip) and the operand rounding of the 16-bit forward kernel (attention_mfma16.hip), so the
    recomputed probabilities match its log-sum-exp."""
    qkv = _req(qkv, "qkv"); pos = _req(pos, "pos"); ctx = _req(ctx, "ctx"); lse = _req(lse, "lse")
    dctx = _req(dctx, "dctx")
    B, T, d3 = qkv.shape
    d = d3 // 3
    dh = d // n_heads
    ctx, dctx = ctx.contiguous(), dctx.contiguous()
    P = 2 * T - 1
    dqkv, dpos, du, dvb = _zeros_split(qkv.device, qkv.dtype, (B, T, d3), (P, d), (n_heads, dh), (n_heads, dh))
    base, dbase = qkv.data_ptr(), dqkv.data_ptr()
    args = (base, base + 4 * d, base + 8 * d, d3, pos.data_ptr(), pos.stride(0), u.data_ptr(), v.data_ptr(), _p(lengths),
            ctx.data_ptr(), dctx.data_ptr(), d, lse.data_ptr(), dbase, dbase + 4 * d, dbase + 8 * d, d3, dpos.data_ptr(), d,
            du.data_ptr(), dvb.data_ptr(), B, T, n_heads, dh, float(drop_p), int(seed))
    prec = mfma16_prec()
    if prec and dh > 16:
        _lib.check(_lib.load().cfm_relpos_attention_bwd_mfma16_f32(prec, *args, _stream()), "cfm_relpos_attention_bwd_mfma16_f32")
    else:
        # fp32 products; under autocast with heads of <= 16 dims (too few contraction terms for 16-bit products to average
        # out, and no matrix work worth saving) still the fp32 kernel, replaying the forward's operand rounding (`prec`)
        _lib.check(_lib.load().cfm_relpos_attention_bwd_f32(*args, prec, _stream()), "cfm_relpos_attention_bwd_f32")
    return dqkv, dpos, du, dvb


# ---- conv-subsampling stem: training forward (keeps h1, h2) and backward -------------------------------------------
def subsample_stem_train(x, w1, b1, w2p, b2):
    """Like subsample_stem but also returns h1 (B,T1,F1,C), needed by the backward (h2 fp32; h1 in the 16-bit type under
    autocast: only GEMM operands read it)."""
    x = _req(x, "x")
    B, F, T = x.shape
    C = w1.shape[0]
    F1, T1 = (F - 1) // 2, (T - 1) // 2
    F2, T2 = (F1 - 1) // 2, (T1 - 1) // 2
    lib = _lib.load()
    p16 = out16_ok(C) if C % 64 == 0 and B * T1 * F1 * C < 2 ** 31 else 0
    if p16:
        # under autocast h1 only feeds the 16-bit conv2 GEMM (forward) and its weight-gradient GEMM (backward): stored in that type
        h1 = torch.empty(B, T1, F1, C, device=x.device, dtype=_DT16[p16])
        _lib.check(lib.cfm_subsample_conv1_relu_out16_f32(p16, x.data_ptr(), w1.data_ptr(), b1.data_ptr(), h1.data_ptr(), B, F,
                                                          T, C, _stream()), "cfm_subsample_conv1_relu_out16_f32")
    else:
        h1 = torch.empty(B, T1, F1, C, device=x.device, dtype=torch.float32)
        _lib.check(lib.cfm_subsample_conv1_relu_f32(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), h1.data_ptr(), B, F, T, C,
                                                    _stream()), "cfm_subsample_conv1_relu_f32")
    h2 = torch.empty(B, T2, F2 * C, device=x.device, dtype=torch.float32)
    _conv2_relu(lib, h1, w2p, b2, h2, B, F1, T1, C)
    return h2, h1


def subsample_stem_bwd(x, w1, b1, w2, h1, h2, dh2):
    """Returns (dw1, db1, dw2 [reference (Co,Ci,3,3) layout], db2)."""
    lib = _lib.load()
    B, F, T = x.shape
    C = w1.shape[0]
    F1, T1 = (F - 1) // 2, (T - 1) // 2
    F2, T2 = (F1 - 1) // 2, (T1 - 1) // 2
    dh2 = _req(dh2, "dh2")
    prec = mfma16_prec() if C % 64 == 0 else 0
    h16 = h1.dtype != torch.float32
    if h16 and (not prec or h1.dtype != _DT16[prec]):
        raise _lib.ConformerHipError("a 16-bit h1 needs the matching precision mode in the backward")
    dw2p, db2, dw1, db1 = _zeros_split(x.device, torch.float32, (C, 9 * C), (C,), tuple(w1.shape), (C,))
    if h16:
        # the all-16-bit stem backward of the autocast path: dz2 = relu'(h2) * dh2 only feeds the two conv2 gradient GEMMs (and
        # the bias gradient, which the weight-gradient kernel sums from the dz2 values it stages)
        dz2 = torch.empty(dh2.shape, device=x.device, dtype=_DT16[prec])
        _lib.check(lib.cfm_relu_bwd_out16_f32(prec, h2.data_ptr(), dh2.data_ptr(), dz2.data_ptr(), dz2.numel(), _stream()),
                   "cfm_relu_bwd_out16_f32")
        rowtab = torch.empty(int(lib.cfm_subsample_conv2_rowtab_elems(B, F1, T1)), device=x.device, dtype=torch.int32)
        _lib.check(lib.cfm_subsample_conv2_bwd_weight_h16_mfma16_f32(prec, dz2.data_ptr(), 1, h1.data_ptr(), rowtab.data_ptr(),
                                                                     dw2p.data_ptr(), db2.data_ptr(), B, F1, T1, C, _stream()),
                   "cfm_subsample_conv2_bwd_weight_h16_mfma16_f32")
    else:
        dz2 = _f32_like(dh2)
        _lib.check(lib.cfm_relu_bwd_f32(h2.data_ptr(), dh2.data_ptr(), dz2.data_ptr(), dz2.numel(), _stream()),
                   "cfm_relu_bwd_f32")
        colsum(dz2.view(-1, C), out=db2)
        if prec:
            _lib.check(lib.cfm_subsample_conv2_bwd_weight_mfma16_f32(prec, dz2.data_ptr(), h1.data_ptr(), dw2p.data_ptr(), B, F1,
                                                                     T1, C, _stream()), "cfm_subsample_conv2_bwd_weight_mfma16_f32")
        else:
            _lib.check(lib.cfm_subsample_conv2_bwd_weight_f32(dz2.data_ptr(), h1.data_ptr(), dw2p.data_ptr(), B, F1, T1, C,
                                                              _stream()), "cfm_subsample_conv2_bwd_weight_f32")
    w2c = torch.empty(9 * C * C, device=x.device, dtype=torch.float32)
    _lib.check(lib.cfm_pack_conv2_weight_t_f32(w2.data_ptr(), w2c.data_ptr(), C, _stream()), "cfm_pack_conv2_weight_t_f32")
    # dh1: fp32, or -- all-16-bit stem of the autocast path -- in the 16-bit type: its only consumer is the conv1 parameter-gradient
    # reduction, and under torch.autocast conv1's incoming gradient is a 16-bit tensor.  (Never empty_like(h1) for the fp32 case:
    # h1 itself may be stored in the 16-bit type.)
    d16 = bool(prec and h16)
    dh1 = torch.empty(h1.shape, device=h1.device, dtype=_DT16[prec] if d16 else torch.float32)
    if prec:
        # transposed conv as four parity-class implicit GEMMs on the FORWARD 16-bit kernel (two-tile prefetch, row-major epilogue)
        w2c16 = torch.empty(9 * C * C, device=x.device, dtype=_DT16[prec])
        _lib.check(lib.cfm_cast16_f32(prec, w2c.data_ptr(), w2c16.data_ptr(), w2c.numel(), _stream()), "cfm_cast16_f32")
        fn = lib.cfm_subsample_conv2_bwd_input_fwdkernel_out16_mfma16_f32 if d16 else lib.cfm_subsample_conv2_bwd_input_fwdkernel_mfma16_f32
        _lib.check(fn(prec, dz2.data_ptr(), int(dz2.dtype != torch.float32), w2c16.data_ptr(), _zero_bias(C, x.device).data_ptr(),
                      dh1.data_ptr(), B, F1, T1, C, _stream()), "cfm_subsample_conv2_bwd_input_fwdkernel_mfma16_f32")
    else:
        _lib.check(lib.cfm_subsample_conv2_bwd_input_f32(dz2.data_ptr(), w2c.data_ptr(), dh1.data_ptr(), B, F1, T1, C,
                                                         _stream()), "cfm_subsample_conv2_bwd_input_f32")
    if d16:
        _lib.check(lib.cfm_subsample_conv1_bwd_d16_f32(prec, x.data_ptr(), w1.data_ptr(), b1.data_ptr(), dh1.data_ptr(), dw1.data_ptr(),
                                                       db1.data_ptr(), B, F, T, C, _stream()), "cfm_subsample_conv1_bwd_d16_f32")
    else:
        _lib.check(lib.cfm_subsample_conv1_bwd_f32(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), dh1.data_ptr(), dw1.data_ptr(),
                                                   db1.data_ptr(), B, F, T, C, _stream()), "cfm_subsample_conv1_bwd_f32")
    dw2 = dw2p.view(C, 3, 3, C).permute(0, 3, 1, 2).contiguous()        # packed (co,kf,kt,ci) -> (co,ci,kf,kt): tiny glue
    return dw1, db1, dw2, db2


# ---- N1 decoder: LSTM over a packed batch, Swish + BatchNorm(eval), vocabulary projection -------------------------------
def lstm_forward(x, w_ih, w_hh, bias, lengths=None, save: bool = False):
    """nn.LSTM(batch_first) forward of one layer: x (B,T,D) -> y (B,T,H); bias = b_ih + b_hh (4H).  `lengths` (B) int64 on
    the device gives pack_padded_sequence semantics (outputs beyond an utterance's length are 0).  The input projection is
    one GEMM; under autocast both it and the recurrent product run on the 16-bit matrix pipe.  save=True also returns (gates, cells)."""
    x = _req(x, "x"); w_hh = _req(w_hh, "weight_hh")
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gx = linear(x, w_ih, bias)                                        # (B,T,4H)
    y = torch.empty(B, T, H, device=x.device, dtype=torch.float32)
    c = torch.empty(B, H, device=x.device, dtype=torch.float32)
    gates = torch.empty(B, T, 4 * H, device=x.device, dtype=torch.float32) if save else None
    cells = torch.empty(B, T, H, device=x.device, dtype=torch.float32) if save else None
    if lengths is not None:
        lengths = _req(lengths, "lengths", torch.int64)
    prec = mfma16_prec()
    w16 = weight16(w_hh, prec) if prec and H % 16 == 0 else None
    if w16 is not None:
        # under autocast the recurrent product runs on the 16-bit matrix pipe too (what autocast does to nn.LSTM on a GPU);
        # gate math, cell state and all stored tensors stay fp32
        h16 = torch.empty(2 * ((B + 31) // 32) * 32 * H, device=x.device, dtype=_DT16[prec])
        # W_hh in MFMA fragment order (H/8, H/16, 2, 32, 8): one contiguous 1 KB block per workgroup and contraction step
        w16 = w16.view(4, H // 8, 8, H // 16, 2, 8).permute(1, 3, 4, 0, 2, 5).contiguous()
        _lib.check(_lib.load().cfm_lstm_fwd_mfma16_f32(prec, gx.data_ptr(), w16.data_ptr(), _p(lengths), y.data_ptr(), c.data_ptr(),
                                                       h16.data_ptr(), _p(gates), _p(cells), B, T, H, _stream()),
                   "cfm_lstm_fwd_mfma16_f32")
        return (y, gates, cells) if save else y
    if H % 16 == 0:
        # fp32 recurrence with both operands in MFMA fragment order (bit-identical to the row-major kernel, fewer cache
        # lines per load): W_hh -> (H/4, H/16, kq 4, gate 4, unit 4, 4)
        wf = w_hh.view(4, H // 4, 4, H // 16, 4, 4).permute(1, 3, 4, 0, 2, 5).contiguous()
        hf = torch.empty(2 * ((B + 15) // 16) * 16 * H, device=x.device, dtype=torch.float32)
        _lib.check(_lib.load().cfm_lstm_fwd_frag_f32(gx.data_ptr(), wf.data_ptr(), _p(lengths), y.data_ptr(), c.data_ptr(),
                                                     hf.data_ptr(), _p(gates), _p(cells), B, T, H, _stream()),
                   "cfm_lstm_fwd_frag_f32")
        return (y, gates, cells) if save else y
    _lib.check(_lib.load().cfm_lstm_fwd_f32(gx.data_ptr(), w_hh.data_ptr(), _p(lengths), y.data_ptr(), c.data_ptr(), _p(gates),
                                            _p(cells), B, T, H, _stream()), "cfm_lstm_fwd_f32")
    return (y, gates, cells) if save else y


def swish_bn_eval(h, bn_mean, bn_var, bn_weight, bn_bias, eps: float = 1e-5):
    h = _req(h, "h")
    C = h.shape[-1]
    out = _f32_like(h)
    _lib.check(_lib.load().cfm_swish_bn_eval_f32(h.data_ptr(), _req(bn_mean, "bn_mean").data_ptr(),
                                                 _req(bn_var, "bn_var").data_ptr(), _req(bn_weight, "bn_weight").data_ptr(),
                                                 _req(bn_bias, "bn_bias").data_ptr(), eps, out.data_ptr(), h.numel() // C, C,
                                                 _stream()), "cfm_swish_bn_eval_f32")
    return out


def lstm_backward(x, w_ih, w_hh, y, gates, cells, dy, lengths=None, need_dx: bool = True):
    """Backward of lstm_forward(save=True): returns (dx or None, dw_ih, dw_hh, dbias) with dbias = d/d(b_ih) = d/d(b_hh).
    The time recursion yields dG (B,T,4H); the weight / input gradients are dense GEMMs over all frames."""
    lib = _lib.load()
    B, T, D = x.shape
    H = w_hh.shape[1]
    dy = _req(dy, "dy")
    dG = torch.empty(B, T, 4 * H, device=x.device, dtype=torch.float32)
    dc = torch.empty(B, H, device=x.device, dtype=torch.float32)
    prec = mfma16_prec()
    wt16 = weight16(w_hh, prec, transposed=True) if prec and H % 16 == 0 else None
    if wt16 is not None:
        # W_hh^T (H,4H) in MFMA fragment order (H/16, 4H/16, 2, 16, 8); dG_t exchanged between steps in the same order
        wt16 = wt16.view(H // 16, 16, 4 * H // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
        dg16 = torch.empty(2 * ((B + 15) // 16) * 16 * 4 * H, device=x.device, dtype=_DT16[prec])
        _lib.check(lib.cfm_lstm_bwd_mfma16_f32(prec, dy.data_ptr(), gates.data_ptr(), cells.data_ptr(), wt16.data_ptr(), _p(lengths),
                                               dG.data_ptr(), dc.data_ptr(), dg16.data_ptr(), B, T, H, _stream()),
                   "cfm_lstm_bwd_mfma16_f32")
    elif H % 16 == 0:
        # fp32: W_hh^T (H,4H) -> (H/16, 4H/16, kq 4, unit 16, 4); dG_t exchanged between steps in the same order
        wtf = w_hh.t().reshape(H // 16, 16, 4 * H // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous()
        dgf = torch.empty(2 * ((B + 15) // 16) * 16 * 4 * H, device=x.device, dtype=torch.float32)
        _lib.check(lib.cfm_lstm_bwd_frag_f32(dy.data_ptr(), gates.data_ptr(), cells.data_ptr(), wtf.data_ptr(), _p(lengths),
                                             dG.data_ptr(), dc.data_ptr(), dgf.data_ptr(), B, T, H, _stream()),
                   "cfm_lstm_bwd_frag_f32")
    else:
        whh_t = w_hh.t().contiguous()                                 # (H,4H): 6.5 MB of glue per step
        _lib.check(lib.cfm_lstm_bwd_f32(dy.data_ptr(), gates.data_ptr(), cells.data_ptr(), whh_t.data_ptr(), _p(lengths),
                                        dG.data_ptr(), dc.data_ptr(), B, T, H, _stream()), "cfm_lstm_bwd_f32")
    dG2, x2 = dG.view(B * T, 4 * H), x.reshape(B * T, D)
    h_prev = torch.zeros(y.shape, device=y.device, dtype=torch.float32)                                      # h_{t-1}: y shifted by one frame per utterance
    h_prev[:, 1:] = y[:, :-1]
    dx = gemm_bwd(dG2, False, w_ih, True, B * T, D, 4 * H, prec=prec).view(B, T, D) if need_dx else None
    dw_ih, dw_hh, db = _zeros_split(x.device, x.dtype, (4 * H, D), (4 * H, H), (4 * H,))
    gemm_bwd(dG2, True, x2, True, 4 * H, D, B * T, allow_split=True, out=dw_ih, prec=prec)
    gemm_bwd(dG2, True, h_prev.view(B * T, H), True, 4 * H, H, B * T, allow_split=True, out=dw_hh, prec=prec)
    colsum(dG2, out=db)
    return dx, dw_ih, dw_hh, db


def swish_bn_batch_stats(h, running_mean, running_var, momentum: float = 0.1):
    """Train-mode BatchNorm statistics of swish(h) over all rows; updates the running buffers in place."""
    h = _req(h, "h")
    C = h.shape[-1]
    mean = torch.empty(C, device=h.device, dtype=torch.float32)
    var = torch.empty(C, device=h.device, dtype=torch.float32)
    _lib.check(_lib.load().cfm_swish_bn_stats_f32(h.data_ptr(), mean.data_ptr(), var.data_ptr(), _p(running_mean),
                                                  _p(running_var), momentum, h.numel() // C, C, _stream()),
               "cfm_swish_bn_stats_f32")
    return mean, var


def swish_bn_bwd(h, dz, bn_mean, bn_var, bn_weight, eps: float = 1e-5, train_stats: bool = False):
    """Returns (dh, dgamma, dbeta)."""
    h = _req(h, "h"); dz = _req(dz, "dz")
    C = h.shape[-1]
    dh = _f32_like(h)
    dga, dbe = _zeros_split(h.device, h.dtype, (C,), (C,))
    _lib.check(_lib.load().cfm_swish_bn_bwd_f32(h.data_ptr(), dz.data_ptr(), bn_mean.data_ptr(), bn_var.data_ptr(),
                                                bn_weight.data_ptr(), eps, int(train_stats), dh.data_ptr(), dga.data_ptr(),
                                                dbe.data_ptr(), h.numel() // C, C, _stream()), "cfm_swish_bn_bwd_f32")
    return dh, dga, dbe


# ---- N1 loss: CTC over the logits (log-softmax folded in), evaluation.py:12-16 ------------------------------------------------
CTC_MAX_TARGET = 1023            # CFM_CTC_MAX_TARGET


def _ctc_geometry(logits, targets, input_lengths, target_lengths):
    logits = _req(logits, "logits")
    if logits.dim() != 3:
        raise _lib.ConformerHipError(f"ctc_loss: logits must be (B,T,V), got {tuple(logits.shape)}")
    B = logits.shape[0]
    targets = _req(targets.to(logits.device), "targets", torch.int64)
    in_len = _req(torch.as_tensor(input_lengths).to(logits.device), "input_lengths", torch.int64)
    tg_len = _req(torch.as_tensor(target_lengths).to(logits.device), "target_lengths", torch.int64)
    if in_len.numel() != B or tg_len.numel() != B:
        raise _lib.ConformerHipError("ctc_loss: input_lengths / target_lengths must hold one entry per utterance")
    if targets.dim() == 2:
        if targets.shape[0] != B:
            raise _lib.ConformerHipError("ctc_loss: 2-D targets must be (B, max_target_length)")
        off, stride, lmax = None, targets.shape[1], targets.shape[1]
    elif targets.dim() == 1:                              # concatenated targets: one host read for the lattice width
        off = (torch.cumsum(tg_len, 0) - tg_len).contiguous()
        stride, lmax = 0, int(tg_len.max())
    else:
        raise _lib.ConformerHipError("ctc_loss: targets must be 1-D (concatenated) or 2-D (padded)")
    lmax = max(lmax, 1)
    if lmax > CTC_MAX_TARGET:
        raise NotImplementedError(f"ctc_loss: targets longer than {CTC_MAX_TARGET} labels are not built")
    return logits, targets, off, stride, in_len, tg_len, lmax


def ctc_loss_forward(logits, targets, input_lengths, target_lengths, blank: int = 0):
    """mean-reduced, zero_infinity CTC loss of log_softmax(logits) -- logits (B,T,V) fp32 batch-first.  Returns
    (loss (scalar tensor), ctx) where ctx carries the lattice workspace for ctc_loss_backward."""
    logits, targets, off, stride, in_len, tg_len, lmax = _ctc_geometry(logits, targets, input_lengths, target_lengths)
    B, T, V = logits.shape
    lib = _lib.load()
    ws = torch.empty(int(lib.cfm_ctc_workspace_floats(B, T, lmax)), device=logits.device, dtype=torch.float32)
    loss = torch.empty((), device=logits.device, dtype=torch.float32)
    _lib.check(lib.cfm_ctc_loss_fwd_f32(logits.data_ptr(), targets.data_ptr(), _p(off), stride, targets.numel(),
                                        in_len.data_ptr(), tg_len.data_ptr(), B, T, V, lmax, int(blank), ws.data_ptr(),
                                        loss.data_ptr(), _stream()), "cfm_ctc_loss_fwd_f32")
    return loss, (logits, targets, off, stride, in_len, tg_len, lmax, int(blank), ws)


def ctc_loss_backward(ctx, grad_out):
    logits, targets, off, stride, in_len, tg_len, lmax, blank, ws = ctx
    B, T, V = logits.shape
    g = _req(grad_out.reshape(1), "grad_out")
    dlogits = _f32_like(logits)
    _lib.check(_lib.load().cfm_ctc_loss_bwd_f32(logits.data_ptr(), targets.data_ptr(), _p(off), stride, targets.numel(),
                                                in_len.data_ptr(), tg_len.data_ptr(), B, T, V, lmax, blank, ws.data_ptr(),
                                                g.data_ptr(), dlogits.data_ptr(), _stream()), "cfm_ctc_loss_bwd_f32")
    return dlogits


def ctc_nll(ctx) -> torch.Tensor:
    """Per-utterance negative log-likelihood (B,) left in the workspace by ctc_loss_forward (inf = no valid alignment)."""
    return ctx[-1][-ctx[0].shape[0]:]
