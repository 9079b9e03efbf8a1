"""Tensor-level wrappers over the C ABI: torch is used only for device memory and the current stream.

Every function takes CUDA(HIP) fp32 tensors, allocates its output with torch.empty and enqueues the HIP
kernels on torch's current stream.  Inputs on the wrong device/dtype raise; there is no eager fallback.
"""
from __future__ import annotations

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


def subsampled_length(n: int) -> int:
    return int(_lib.load().cfm_subsampled_length(int(n)))


def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    x = _req(x, "x"); weight = _req(weight, "weight"); bias = _req(bias, "bias")
    d = x.shape[-1]
    rows = x.numel() // d
    y = torch.empty_like(x) if out is None else out
    st = _lib.load().cfm_layernorm_fwd_f32(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                           None, None, rows, d, eps, _stream())
    _lib.check(st, "cfm_layernorm_fwd_f32")
    return y


def _gemm_common(a: torch.Tensor, w: torch.Tensor, b: torch.Tensor):
    a = _req(a, "A"); w = _req(w, "W"); b = _req(b, "bias")
    k = a.shape[-1]
    m = a.numel() // k
    w2 = w.reshape(w.shape[0], -1)
    if w2.shape[1] != k:
        raise _lib.ConformerHipError(f"GEMM shape mismatch: A(...,{k}) vs W{tuple(w.shape)}")
    return a, w2, b, m, w2.shape[0], k


def linear(a, w, b, act: str = "none") -> torch.Tensor:
    """y = act(a @ w.T + b); act in {none, swish, relu}."""
    a, w2, b, m, n, k = _gemm_common(a, w, b)
    c = torch.empty(*a.shape[:-1], n, device=a.device, dtype=a.dtype)
    fn = {"none": "cfm_gemm_bias_f32", "swish": "cfm_gemm_bias_swish_f32", "relu": "cfm_gemm_bias_relu_f32"}[act]
    st = getattr(_lib.load(), fn)(a.data_ptr(), w2.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, k, n, _stream())
    _lib.check(st, fn)
    return c


def linear_glu(a, w, b) -> torch.Tensor:
    """(a @ Wv.T + bv) * sigmoid(a @ Wg.T + bg) with W = [Wv; Wg] (pointwise Conv1d(d->2d) + GLU)."""
    a, w2, b, m, n2, k = _gemm_common(a, w, b)
    n = n2 // 2
    c = torch.empty(*a.shape[:-1], n, device=a.device, dtype=a.dtype)
    st = _lib.load().cfm_gemm_bias_glu_f32(a.data_ptr(), w2.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, k, n,
                                           _stream())
    _lib.check(st, "cfm_gemm_bias_glu_f32")
    return c


def linear_residual(a, w, b, res: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """alpha * (a @ w.T + b) + res."""
    a, w2, b, m, n, k = _gemm_common(a, w, b)
    res = _req(res, "residual")
    c = torch.empty(*a.shape[:-1], n, device=a.device, dtype=a.dtype)
    st = _lib.load().cfm_gemm_bias_residual_f32(a.data_ptr(), w2.data_ptr(), b.data_ptr(), res.data_ptr(), alpha,
                                                c.data_ptr(), m, n, k, k, n, n, _stream())
    _lib.check(st, "cfm_gemm_bias_residual_f32")
    return c


def relpos_table(div_term: torch.Tensor, t: int) -> torch.Tensor:
    dt = _req(div_term, "div_term").reshape(-1)
    d = 2 * dt.numel()
    pe = torch.empty(2 * t - 1, d, device=dt.device, dtype=dt.dtype)
    st = _lib.load().cfm_relpos_table_f32(dt.data_ptr(), pe.data_ptr(), t, d, _stream())
    _lib.check(st, "cfm_relpos_table_f32")
    return pe


def relpos_attention(qkv: torch.Tensor, pos: torch.Tensor, u: torch.Tensor, v: torch.Tensor,
                     lengths: Optional[torch.Tensor], n_heads: int) -> torch.Tensor:
    """qkv: (B,T,3d) fused projections [q|k|v]; pos: (2T-1,d) projected table; returns ctx (B,T,d)."""
    qkv = _req(qkv, "qkv"); u = _req(u, "content_bias"); v = _req(v, "position_bias")
    B, T, d3 = qkv.shape
    d = d3 // 3
    dh = d // n_heads
    if not (isinstance(pos, torch.Tensor) and pos.is_cuda and pos.dtype == torch.float32 and pos.dim() == 2
            and pos.stride(1) == 1 and pos.shape == (2 * T - 1, d)):
        raise _lib.ConformerHipError(f"pos: expected a ({2 * T - 1},{d}) fp32 HIP tensor with unit column stride")
    ldp = pos.stride(0)
    if lengths is not None:
        lengths = _req(lengths, "lengths", torch.int64)
    ctx = torch.empty(B, T, d, device=qkv.device, dtype=qkv.dtype)
    base = qkv.data_ptr()
    st = _lib.load().cfm_relpos_attention_fwd_f32(base, base + 4 * d, base + 8 * d, d3, pos.data_ptr(), ldp,
                                                  u.data_ptr(), v.data_ptr(), _p(lengths), ctx.data_ptr(), d, None,
                                                  B, T, n_heads, dh, _stream())
    _lib.check(st, "cfm_relpos_attention_fwd_f32")
    return ctx


def dwconv_bn_swish(g, w, b, bn_w, bn_b, bn_mean, bn_var, eps: float = 1e-5) -> torch.Tensor:
    g = _req(g, "g"); w = _req(w, "dw weight"); b = _req(b, "dw bias")
    B, T, C = g.shape
    K = w.shape[-1]
    y = torch.empty_like(g)
    st = _lib.load().cfm_dwconv_bn_swish_fwd_f32(g.data_ptr(), w.data_ptr(), b.data_ptr(), _req(bn_w, "bn_w").data_ptr(),
                                                 _req(bn_b, "bn_b").data_ptr(), _req(bn_mean, "bn_mean").data_ptr(),
                                                 _req(bn_var, "bn_var").data_ptr(), eps, y.data_ptr(), B, T, C, K,
                                                 _stream())
    _lib.check(st, "cfm_dwconv_bn_swish_fwd_f32")
    return y


def pack_conv2_weight(w2: torch.Tensor) -> torch.Tensor:
    w2 = _req(w2, "conv_2.weight")
    C = w2.shape[0]
    out = torch.empty(C, 9 * C, device=w2.device, dtype=w2.dtype)
    _lib.check(_lib.load().cfm_pack_conv2_weight_f32(w2.data_ptr(), out.data_ptr(), C, _stream()), "cfm_pack_conv2_weight_f32")
    return out


def pack_linear_weight(wl: torch.Tensor, C: int, F2: int) -> torch.Tensor:
    wl = _req(wl, "linear.weight")
    out = torch.empty_like(wl)
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
    h1 = torch.empty(B, T1, F1, C, device=x.device, dtype=x.dtype)
    _lib.check(lib.cfm_subsample_conv1_relu_f32(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), h1.data_ptr(), B, F, T, C,
                                                _stream()), "cfm_subsample_conv1_relu_f32")
    h2 = torch.empty(B, T2, F2 * C, device=x.device, dtype=x.dtype)
    _lib.check(lib.cfm_subsample_conv2_relu_f32(h1.data_ptr(), w2p.data_ptr(), b2.data_ptr(), h2.data_ptr(), B, F1, T1,
                                                C, _stream()), "cfm_subsample_conv2_relu_f32")
    return h2
