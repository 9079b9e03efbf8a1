"""torch.autograd.Function wrappers: one Function per residual sub-layer of the Conformer block, so the backward of
`alpha*module(LN(x)) + x` is computed by explicit gfx950 kernels end to end (no autograd-side adds, no eager ops).

The reference relies on plain autograd over ~50 eager ops per block (train.py:239); here every Function saves the
handful of activations its backward kernels need and returns the input gradient -- residual
path already folded in -- plus all parameter gradients.  fp32 only; dropout p = 0 (p > 0 is refused by the modules);
BatchNorm1d in both modes (running statistics in eval, batch statistics + running update + coupled backward in train).
"""
from __future__ import annotations

from typing import Optional

import functools

import torch
from torch.autograd import Function

from . import ops


def _flat(t: torch.Tensor) -> torch.Tensor:
    return t.reshape(-1, t.shape[-1])


def _fwd_prec(fn):
    """Remember the matrix-pipe precision (autocast state) the forward ran under ..."""
    @functools.wraps(fn)
    def wrapped(ctx, *args):
        ctx.prec = ops.mfma16_prec()
        return fn(ctx, *args)
    return wrapped


def _bwd_prec(fn):
    """... and run the backward GEMMs at the same precision (autocast is off inside autograd's backward)."""
    @functools.wraps(fn)
    def wrapped(ctx, *grads):
        with ops.precision(ctx.prec):
            return fn(ctx, *grads)
    return wrapped


class LayerNormFn(Function):
    """y = LayerNorm(x) -- the block's closing norm (block.py:27)."""

    @staticmethod
    @_fwd_prec
    def forward(ctx, x, weight, bias, eps):
        y, mean, rstd = ops.layernorm_train(x, weight, bias, eps)
        ctx.save_for_backward(x, weight, mean, rstd)
        return y

    @staticmethod
    @_bwd_prec
    def backward(ctx, dy):
        x, weight, mean, rstd = ctx.saved_tensors
        dx, dw, db = ops.layernorm_bwd(x, weight, dy.contiguous(), mean, rstd)
        return dx, dw, db, None


class LinearFn(Function):
    """y = x @ w.T + b (the batched positional projection, attention.py:81, and the encoder's input Linear, encoder.py:23).
    dx is produced only when x requires grad."""

    @staticmethod
    @_fwd_prec
    def forward(ctx, x, w, b, drop_p=0.0):
        ctx.save_for_backward(x, w)
        ctx.drop_p = float(drop_p)
        if ctx.drop_p > 0.0:
            ctx.seed, = ops.new_seeds(1)
            return ops.linear_train("bias", x, w, b, drop_p=ctx.drop_p, seed=ctx.seed)
        return ops.linear(x, w, b)

    @staticmethod
    @_bwd_prec
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.drop_p > 0.0:
            dy = ops.dropout_apply(dy, ctx.drop_p, ctx.seed)
        dx, dw, db = ops.linear_bwd(_flat(x), w, _flat(dy), need_dx=ctx.needs_input_grad[0])
        return (dx.view_as(x) if dx is not None else None), dw, db, None


class SubsampleStemFn(Function):
    """h2 (B,T',F'*C) = relu(conv2(relu(conv1(x)))) in the channel-last hot-path layout (convolution.py:42-52).
    The packed conv2 weight is rebuilt per call in training (weights change every step)."""

    @staticmethod
    @_fwd_prec
    def forward(ctx, x, w1, b1, w2, b2):
        w2p = ops.pack_conv2_weight(w2)
        h2, h1 = ops.subsample_stem_train(x, w1, b1, w2p, b2)
        ctx.save_for_backward(x, w1, b1, w2, h1, h2)
        return h2

    @staticmethod
    @_bwd_prec
    def backward(ctx, dh2):
        x, w1, b1, w2, h1, h2 = ctx.saved_tensors
        dw1, db1, dw2, db2 = ops.subsample_stem_bwd(x, w1, b1, w2, h1, h2, dh2.contiguous())
        return None, dw1, db1, dw2, db2


class FeedForwardFn(Function):
    """out = alpha * (W2 . swish(W1 . LN(x) + b1) + b2) + x        (ffn.py:15-23 + block.py:19,25)"""

    @staticmethod
    @_fwd_prec
    def forward(ctx, x, ln_w, ln_b, w1, b1, w2, b2, alpha, eps, drop_p=0.0):
        h0, mean, rstd = ops.layernorm_train(x, ln_w, ln_b, eps, for_gemm=True)
        ctx.drop_p = float(drop_p)
        ctx.seeds = ops.new_seeds(2) if ctx.drop_p > 0.0 else (0, 0)
        # h = drop1(swish(z)); out = alpha * drop2(h.W2^T + b2) + x     (ffn.py:17-21)
        h, z = ops.linear_train("swish", h0, w1, b1, drop_p=ctx.drop_p, seed=ctx.seeds[0], save_z=True, for_gemm=True)
        out = ops.linear_train("residual", h, w2, b2, residual=x, alpha=alpha, drop_p=ctx.drop_p, seed=ctx.seeds[1])
        ctx.save_for_backward(x, ln_w, mean, rstd, h0, z, h, w1, w2)
        ctx.alpha = alpha
        return out

    @staticmethod
    @_bwd_prec
    def backward(ctx, dout):
        x, ln_w, mean, rstd, h0, z, h, w1, w2 = ctx.saved_tensors
        dout = dout.contiguous()
        d2 = ops.dropout_apply(dout, ctx.drop_p, ctx.seeds[1], for_gemm=True)   # gradient w.r.t. the out-projection result (GEMM operand only)
        # d(pre-activation) = alpha * (d2 . W2) * mask1 * swish'(z); dW2 = alpha * d2^T . h (h already carries mask1)
        # (dz only feeds the two GEMMs below: under autocast it is stored in the 16-bit type they would round it to)
        dz, dw2, db2 = ops.linear_bwd(_flat(h), w2, _flat(d2), alpha=ctx.alpha, Z=_flat(z), drop_p=ctx.drop_p,
                                      drop_seed=ctx.seeds[0], dx16=True)
        dh0, dw1, db1 = ops.linear_bwd(_flat(h0), w1, dz)
        dx, dlw, dlb = ops.layernorm_bwd(x, ln_w, dh0.view_as(x), mean, rstd, dres=dout)
        return dx, dlw, dlb, dw1, db1, dw2, db2, None, None, None


class RelPosAttentionFn(Function):
    """ctx = RelPosAttention(qkv, pos): the attention core alone (attention.py:47-72,94-102) -- what RelativeMultiHeadAttention's
    own forward needs between its projection Linears when it is called directly in training."""

    @staticmethod
    @_fwd_prec
    def forward(ctx, qkv, pos, u, vb, lengths, n_heads, drop_p=0.0):
        ctx.drop_p = float(drop_p)
        ctx.seed = ops.new_seeds(1)[0] if ctx.drop_p > 0.0 else 0
        att, lse = ops.relpos_attention_train(qkv, pos, u, vb, lengths, n_heads, ctx.drop_p, ctx.seed)
        ctx.save_for_backward(qkv, pos, u, vb, att, lse, lengths if lengths is not None else torch.empty(0))
        ctx.has_len = lengths is not None
        ctx.n_heads = n_heads
        return att

    @staticmethod
    @_bwd_prec
    def backward(ctx, datt):
        qkv, pos, u, vb, att, lse, lengths = ctx.saved_tensors
        dqkv, dpos, du, dvb = ops.relpos_attention_bwd(qkv, pos, u, vb, lengths if ctx.has_len else None, ctx.n_heads, att, lse,
                                                       datt.contiguous(), ctx.drop_p, ctx.seed)
        return dqkv, dpos, du, dvb, None, None, None


class SelfAttentionFn(Function):
    """out = Wo . RelPosAttention(LN(x)) + bo + x                  (attention.py:14-18,47-102 + block.py:21)
    `pos` is this layer's (2T-1, d) slice of the projected position table (it has its own graph through LinearFn)."""

    @staticmethod
    @_fwd_prec
    def forward(ctx, x, ln_w, ln_b, wq, bq, wk, bk, wv, bv, pos, u, vb, wo, bo, lengths, n_heads, eps, drop_p=0.0):
        h0, mean, rstd = ops.layernorm_train(x, ln_w, ln_b, eps, for_gemm=True)
        wqkv = torch.cat([wq, wk, wv], dim=0)
        bqkv = torch.cat([bq, bk, bv], dim=0)
        qkv = ops.linear(h0, wqkv, bqkv)
        ctx.drop_p = float(drop_p)
        ctx.seeds = ops.new_seeds(2) if ctx.drop_p > 0.0 else (0, 0)
        att, lse = ops.relpos_attention_train(qkv, pos, u, vb, lengths, n_heads, ctx.drop_p, ctx.seeds[0])   # attention.py:67
        out = ops.linear_train("residual", att, wo, bo, residual=x, alpha=1.0, drop_p=ctx.drop_p, seed=ctx.seeds[1])
        ctx.save_for_backward(x, ln_w, mean, rstd, h0, wqkv, qkv, pos, u, vb, att, lse, wo,
                              lengths if lengths is not None else torch.empty(0))
        ctx.has_len = lengths is not None
        ctx.n_heads = n_heads
        return out

    @staticmethod
    @_bwd_prec
    def backward(ctx, dout):
        x, ln_w, mean, rstd, h0, wqkv, qkv, pos, u, vb, att, lse, wo, lengths = ctx.saved_tensors
        lengths = lengths if ctx.has_len else None
        dout = dout.contiguous()
        d = x.shape[-1]
        d2 = ops.dropout_apply(dout, ctx.drop_p, ctx.seeds[1], for_gemm=True)
        datt, dwo, dbo = ops.linear_bwd(_flat(att), wo, _flat(d2))
        dqkv, dpos, du, dvb = ops.relpos_attention_bwd(qkv, pos, u, vb, lengths, ctx.n_heads, att,
                                                       lse, datt.view_as(att), ctx.drop_p, ctx.seeds[0])
        dh0, dwqkv, dbqkv = ops.linear_bwd(_flat(h0), wqkv, _flat(dqkv))
        dx, dlw, dlb = ops.layernorm_bwd(x, ln_w, dh0.view_as(x), mean, rstd, dres=dout)
        dwq, dwk, dwv = dwqkv[:d], dwqkv[d:2 * d], dwqkv[2 * d:]
        dbq, dbk, dbv = dbqkv[:d], dbqkv[d:2 * d], dbqkv[2 * d:]
        return dx, dlw, dlb, dwq, dbq, dwk, dbk, dwv, dbv, dpos, du, dvb, dwo, dbo, None, None, None, None


class ConvModuleFn(Function):
    """out = Wpw2 . swish(BN(dwconv(GLU(Wpw1 . LN(x))))) + x       (convolution.py:21-32 + block.py:23)
    train_bn = False: BatchNorm uses the fixed (running) statistics in forward and backward (eval).
    train_bn = True : batch statistics over all B*T positions (padded frames included); the running buffers are
                      updated in place (momentum) and the backward carries the mean/variance coupling."""

    @staticmethod
    @_fwd_prec
    def forward(ctx, x, ln_w, ln_b, w1, b1, wd, bd, bn_w, bn_b, bn_mean, bn_var, w2, b2, eps_ln, eps_bn, train_bn,
                momentum, drop_p=0.0):
        h0, mean, rstd = ops.layernorm_train(x, ln_w, ln_b, eps_ln, for_gemm=True)
        z = ops.linear(h0, w1, b1)                                  # (B,T,2C) pre-activation kept for GLU'
        g = ops.glu_fwd(z)
        if train_bn:
            bn_mean, bn_var = ops.dwconv_bn_batch_stats(g, wd, bd, bn_mean, bn_var, momentum)
        s = ops.dwconv_bn_swish(g, wd, bd, bn_w, bn_b, bn_mean, bn_var, eps_bn, for_gemm=True)
        ctx.drop_p = float(drop_p)
        ctx.seed = ops.new_seeds(1)[0] if ctx.drop_p > 0.0 else 0
        out = ops.linear_train("residual", s, w2, b2, residual=x, alpha=1.0, drop_p=ctx.drop_p, seed=ctx.seed)
        ctx.save_for_backward(x, ln_w, mean, rstd, h0, z, g, s, w1, wd, bd, bn_w, bn_b, bn_mean, bn_var, w2)
        ctx.eps_bn = eps_bn
        ctx.train_bn = bool(train_bn)
        return out

    @staticmethod
    @_bwd_prec
    def backward(ctx, dout):
        x, ln_w, mean, rstd, h0, z, g, s, w1, wd, bd, bn_w, bn_b, bn_mean, bn_var, w2 = ctx.saved_tensors
        dout = dout.contiguous()
        d2 = ops.dropout_apply(dout, ctx.drop_p, ctx.seed, for_gemm=True)
        ds, dw2, db2 = ops.linear_bwd(_flat(s), w2, _flat(d2))
        dg, dwd, dbd, dbnw, dbnb = ops.dwconv_bn_swish_bwd(g, ds.view_as(g), wd, bd, bn_w, bn_b, bn_mean, bn_var,
                                                           ctx.eps_bn, ctx.train_bn)
        dz = ops.glu_bwd(z, dg, for_gemm=True)
        dh0, dw1, db1 = ops.linear_bwd(_flat(h0), w1, _flat(dz))
        dx, dlw, dlb = ops.layernorm_bwd(x, ln_w, dh0.view_as(x), mean, rstd, dres=dout)
        return dx, dlw, dlb, dw1, db1, dwd, dbd, dbnw, dbnb, None, None, dw2, db2, None, None, None, None, None


def needs_grad(module: torch.nn.Module, *tensors: Optional[torch.Tensor]) -> bool:
    """True when the call must go through the autograd Functions (grad mode on and something requires grad)."""
    if not torch.is_grad_enabled():
        return False
    return any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors) or \
        any(p.requires_grad for p in module.parameters())


class LstmFn(Function):
    """One nn.LSTM(batch_first) layer over a packed batch (decoder.py:10,17-22): GEMM + per-frame recurrence kernels
    forward, back-propagation through time + three GEMMs backward (csrc/lstm.hip)."""

    @staticmethod
    @_fwd_prec
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, lengths):
        bias = b_ih + b_hh
        y, gates, cells = ops.lstm_forward(x, w_ih, w_hh, bias, lengths, save=True)
        ctx.save_for_backward(x, w_ih, w_hh, y, gates, cells)
        ctx.lengths = lengths
        return y

    @staticmethod
    @_bwd_prec
    def backward(ctx, dy):
        x, w_ih, w_hh, y, gates, cells = ctx.saved_tensors
        dx, dw_ih, dw_hh, db = ops.lstm_backward(x, w_ih, w_hh, y, gates, cells, dy.contiguous(), ctx.lengths,
                                                 need_dx=ctx.needs_input_grad[0])
        return dx, dw_ih, dw_hh, db, db, None


class SwishBatchNormFn(Function):
    """BatchNorm1d(swish(h)) over the channel-last (B,T,C) tensor (decoder.py:23-26): eval = running statistics,
    train = batch statistics over all B*T rows + running update + coupled backward."""

    @staticmethod
    def forward(ctx, h, weight, bias, running_mean, running_var, train_bn, momentum, eps):
        if train_bn:
            mean, var = ops.swish_bn_batch_stats(h, running_mean, running_var, momentum)
        else:
            mean, var = running_mean, running_var
        z = ops.swish_bn_eval(h, mean, var, weight, bias, eps)
        ctx.save_for_backward(h, weight, mean, var)
        ctx.train_bn, ctx.eps = bool(train_bn), eps
        return z

    @staticmethod
    def backward(ctx, dz):
        h, weight, mean, var = ctx.saved_tensors
        dh, dga, dbe = ops.swish_bn_bwd(h, dz.contiguous(), mean, var, weight, ctx.eps, ctx.train_bn)
        return dh, dga, dbe, None, None, None, None, None


class CtcLossFn(Function):
    """ConformerCriterion.ctc_loss (evaluation.py:12-16): mean-reduced zero_infinity CTC of log_softmax(logits), logits
    (B,T,V) batch-first fp32.  Forward: log-sum-exp + alpha lattice; backward: beta lattice + the logits gradient
    (softmax - occupancy), csrc/ctc.hip.  The log-softmax / transpose tensors of the reference are never formed."""

    @staticmethod
    def forward(ctx, logits, targets, input_lengths, target_lengths, blank):
        loss, state = ops.ctc_loss_forward(logits, targets, input_lengths, target_lengths, blank)
        ctx.state = state
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        return ops.ctc_loss_backward(ctx.state, grad_out.float().contiguous()), None, None, None, None
