"""ConvolutionModule + ConvolutionSubsampling (surface of model/utils/convolution.py:7-93) on gfx950 kernels.

ConvolutionModule keeps activations channel-LAST (B,T,C) end to end, so the two transposes of
convolution.py:23,31 disappear:
    LN -> pointwise Conv1d(C->2C)+GLU as ONE MFMA GEMM with a GLU epilogue
       -> depthwise Conv1d(K)+BatchNorm1d(eval)+Swish as one register-sliding-window kernel
       -> pointwise Conv1d(C->C) GEMM (+ fused residual when called from ConformerBlock).
ConvolutionSubsampling: conv1+ReLU writes a channel-last activation; conv2+ReLU is an implicit GEMM on the
MFMA pipe over a re-laid-out (cached) weight; the (B,T',C*F') flattening order of convolution.py:51-52 is
folded into a cached column permutation of the encoder's input Linear, so no permute copy runs on the
hot path.  Parameter names (incl. the reference's `deepwise_conv` spelling) are the state_dict contract.
"""
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import autograd as ag
from ... import ops
from ._guard import PackCache, active_dropout, refuse_dropout
from .activation import GLU, Swish


class ConvolutionModule(nn.Module):
    def __init__(self, channels: int, kernel_size: int, dropout_rate: float = 0.0) -> None:
        super().__init__()
        self.layer_norm = nn.LayerNorm(normalized_shape=channels)
        self.pointwise_conv_1 = nn.Conv1d(channels, channels * 2, kernel_size=1)
        self.glu = GLU(dim=1)
        self.deepwise_conv = nn.Conv1d(channels, channels, kernel_size=kernel_size, padding=(kernel_size - 1) // 2,
                                       groups=channels)
        self.batch_norm = nn.BatchNorm1d(num_features=channels)
        self.swish = Swish()
        self.pointwise_conv_2 = nn.Conv1d(channels, channels, kernel_size=1)
        self.dropout = nn.Dropout(p=dropout_rate)
        self._packs = PackCache()

    def fused(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None, stats: Optional[torch.Tensor] = None,
              emit_stats: bool = False):
        """stats / emit_stats: see FeedForwardModule.fused (the LayerNorm of convolution.py:22 folds into the pointwise_conv_1 +
        GLU GEMM; eval-mode BatchNorm only)."""
        refuse_dropout(self, "ConvolutionModule")
        bn = self.batch_norm
        train_bn = bn.training or bn.running_mean is None      # nn.BatchNorm1d semantics: the BN sub-module's own flag
        if train_bn and (bn.momentum is None or not bn.track_running_stats):
            raise NotImplementedError("ConvolutionModule: only the default BatchNorm1d(momentum=0.1, "
                                      "track_running_stats=True) is built")
        if train_bn and not ag.needs_grad(self, x, residual):
            # .train() under no_grad: still batch statistics + running-stat update (nn.BatchNorm1d semantics)
            h = ops.layernorm(x, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps, for_gemm=True)
            g = ops.linear_glu(h, self.pointwise_conv_1.weight, self.pointwise_conv_1.bias)
            bm, bv = ops.dwconv_bn_batch_stats(g, self.deepwise_conv.weight, self.deepwise_conv.bias, bn.running_mean,
                                               bn.running_var, bn.momentum)
            bn.num_batches_tracked += 1
            s = ops.dwconv_bn_swish(g, self.deepwise_conv.weight, self.deepwise_conv.bias, bn.weight, bn.bias, bm, bv,
                                    bn.eps, for_gemm=True)
            if residual is None:
                return ops.linear(s, self.pointwise_conv_2.weight, self.pointwise_conv_2.bias)
            return ops.linear_residual(s, self.pointwise_conv_2.weight, self.pointwise_conv_2.bias, residual, 1.0)
        if ag.needs_grad(self, x, residual):
            if residual is not None and residual is not x:
                raise NotImplementedError("ConvolutionModule.fused: pass residual=x (block.py:23)")
            out = ag.ConvModuleFn.apply(x, self.layer_norm.weight, self.layer_norm.bias, self.pointwise_conv_1.weight,
                                        self.pointwise_conv_1.bias, self.deepwise_conv.weight, self.deepwise_conv.bias,
                                        bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                        self.pointwise_conv_2.weight, self.pointwise_conv_2.bias,
                                        self.layer_norm.eps, bn.eps, train_bn, bn.momentum if train_bn else 0.0,
                                        active_dropout(self.dropout))
            if train_bn:
                bn.num_batches_tracked += 1
            return out if residual is not None else out - x
        if stats is not None:
            ln, pw1 = self.layer_norm, self.pointwise_conv_1
            wf, bf, cs = self._packs.get("ln_fold", (pw1.weight, pw1.bias, ln.weight, ln.bias),
                                         lambda: ops.fold_layernorm(pw1.weight, pw1.bias, ln.weight, ln.bias))
            g = ops.linear_lnfold(x, stats, wf, bf, cs, ln.eps, glu=True)
        else:
            h = ops.layernorm(x, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps, for_gemm=True)
            g = ops.linear_glu(h, self.pointwise_conv_1.weight, self.pointwise_conv_1.bias)
        s = ops.dwconv_bn_swish(g, self.deepwise_conv.weight, self.deepwise_conv.bias, bn.weight, bn.bias,
                                bn.running_mean, bn.running_var, bn.eps, for_gemm=True)
        if residual is None:
            if emit_stats:
                return ops.linear(s, self.pointwise_conv_2.weight, self.pointwise_conv_2.bias, emit_stats=True)
            return ops.linear(s, self.pointwise_conv_2.weight, self.pointwise_conv_2.bias)
        return ops.linear_residual(s, self.pointwise_conv_2.weight, self.pointwise_conv_2.bias, residual, 1.0, emit_stats=emit_stats)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.fused(x)


class ConvolutionSubsampling(nn.Module):
    def __init__(self, channels: int) -> None:
        super().__init__()
        self.conv_1 = nn.Conv2d(1, channels, kernel_size=3, stride=2)
        self.act_1 = nn.ReLU()
        self.conv_2 = nn.Conv2d(channels, channels, kernel_size=3, stride=2)
        self.act_2 = nn.ReLU()
        self._packs = PackCache()

    @staticmethod
    def out_lengths(lengths: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        if lengths is None:
            return None
        if lengths.is_cuda and lengths.dtype == torch.int64:
            return ops.subsampled_lengths(lengths)          # one launch (four stock elementwise kernels otherwise)
        return torch.div(torch.div(lengths - 1, 2, rounding_mode="floor") - 1, 2, rounding_mode="floor")

    def channel_last(self, x: torch.Tensor) -> torch.Tensor:
        """(B, n_mel, T) -> (B, T', F'*C) with feature index f*C + c (the hot-path layout)."""
        if ag.needs_grad(self, x):
            if x.requires_grad:
                raise NotImplementedError("ConvolutionSubsampling: the gradient w.r.t. the input spectrogram is not "
                                          "built (the reference never needs it: the input is data)")
            return ag.SubsampleStemFn.apply(x, self.conv_1.weight, self.conv_1.bias, self.conv_2.weight, self.conv_2.bias)
        w2p = self._packs.get("w2p", (self.conv_2.weight,), lambda: ops.pack_conv2_weight(self.conv_2.weight))
        return ops.subsample_stem(x, self.conv_1.weight, self.conv_1.bias, w2p, self.conv_2.bias)

    def forward(self, x: torch.Tensor, lengths: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """Reference-compatible output layout (feature index c*F' + f); off the hot path (one permute copy)."""
        h = self.channel_last(x)
        if h.dtype != torch.float32:                # inference under autocast keeps the stem output in the 16-bit type
            h = h.float()
        B, T2, _ = h.shape
        C = self.conv_1.out_channels
        h = h.view(B, T2, -1, C).transpose(2, 3).reshape(B, T2, -1)
        return h, self.out_lengths(lengths)


class DepthWiseSeperableConvolution(nn.Module):
    """Unused by the reference model (convolution.py:59-69); stock ops, kept for surface parity."""

    def __init__(self, in_channels: int, out_channels: int) -> None:
        super().__init__()
        self.depth_wise_conv = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=2, groups=in_channels)
        self.point_wise_conv = nn.Conv2d(out_channels, out_channels, kernel_size=1)

    def forward(self, x: torch.Tensor):
        return self.point_wise_conv(self.depth_wise_conv(x))


class DownsamplingConvolution(nn.Module):
    """Unused by the reference model (convolution.py:71-93); stock ops, kept for surface parity."""

    def __init__(self, channels: int) -> None:
        super().__init__()
        self.conv_1 = DepthWiseSeperableConvolution(1, channels)
        self.conv_2 = DepthWiseSeperableConvolution(channels, channels)

    def forward(self, x: torch.Tensor, lengths: Optional[torch.Tensor]):
        h = F.relu(self.conv_2(F.relu(self.conv_1(x.unsqueeze(1)))))
        B, C, Fp, Tp = h.shape
        h = h.permute(0, 3, 1, 2).reshape(B, Tp, C * Fp)
        return h, ConvolutionSubsampling.out_lengths(lengths)
