"""Pure-torch layer builders with the reference's names, constructor keywords and
``state_dict`` layout (pointnet2_lib/pointnet2/pytorch_utils.py:7-236): ``SharedMLP``,
``Conv1d``, ``Conv2d``, ``FC``, ``BatchNorm1d/2d``.  These run on rocBLAS/MIOpen through
PyTorch -- they are not part of the hand-written hot path, but checkpoints trained with
the reference must load key-for-key (``layer{i}.conv.weight``, ``layer{i}.bn.bn.*``)."""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

FUSED_BN_TRAIN = True   # training step: BatchNorm + ReLU through ws3d_bn_relu_train_* (clear to use the library pair)


class _BnReluTrain(Function):
    """training-mode BatchNorm (+ReLU) of a channels-first tensor in 3 + 5 HBM passes (bn_relu.hip)"""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, relu, tracked):
        from . import compat as _C
        x = x.contiguous()
        y, mean, invstd = _C.bn_relu_train_fwd(x, gamma.detach(), beta.detach(), running_mean, running_var, momentum, eps, relu,
                                               tracked)
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import compat as _C
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        dx, dgamma, dbeta = _C.bn_relu_train_bwd(x, dy.contiguous(), gamma.detach(), beta.detach(), mean, invstd, ctx.relu)
        return dx, dgamma, dbeta, None, None, None, None, None, None


FUSED_CONV_WGRAD = True   # training step: weight gradient of the 1x1 convolutions through ws3d_conv1x1_wgrad


class _Conv1x1Train(Function):
    """1x1 convolution whose WEIGHT gradient runs on ws3d_conv1x1_wgrad: channels-first operands read
    where they lie, fp32 matrix cores, slices added in a fixed order -- bit-reproducible, and faster than the
    library's NHWC split-K kernels + their transposes (2.7 vs 3.8 ms per Stage-1 step).  Forward and the input
    gradient stay on the library convolution."""

    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        nd = x.dim() - 2
        return torch.ops.aten.convolution(x, w, bias, [1] * nd, [0] * nd, [1] * nd, False, [0] * nd, 1)

    @staticmethod
    def backward(ctx, gy):
        from . import compat as _C
        x, w = ctx.saved_tensors
        nd = x.dim() - 2
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.ops.aten.convolution_backward(gy, x, w, None, [1] * nd, [0] * nd, [1] * nd, False, [0] * nd, 1,
                                                     [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            gw = _C.conv1x1_wgrad(gy, x.contiguous(), w.shape)      # a fresh tensor in the weight's shape: autograd keeps it as .grad without a copy
        if ctx.needs_input_grad[2]:
            gb = gy.sum(dim=[0] + list(range(2, gy.dim())))
        return gx, gw, gb


def conv1x1_train(x: torch.Tensor, conv: nn.Module) -> torch.Tensor:
    """``conv(x)`` for a 1x1 Conv1d / Conv2d in training, weight gradient on our kernel"""
    if FUSED_CONV_WGRAD and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled():
        return _Conv1x1Train.apply(x, conv.weight, conv.bias)
    return conv(x)


def bn_relu_train(x: torch.Tensor, bn: nn.modules.batchnorm._BatchNorm, relu: bool) -> torch.Tensor:
    """``relu(bn(x))`` for a BatchNorm module in train() mode, with the module's bookkeeping
    (running statistics, num_batches_tracked, momentum=None -> cumulative average)"""
    momentum = 0.0 if bn.momentum is None else bn.momentum
    tracked = bn.num_batches_tracked if bn.track_running_stats else None
    if tracked is not None and bn.momentum is None:      # cumulative average: the count is needed on the host
        tracked.add_(1)
        momentum, tracked = 1.0 / float(tracked), None
    stats = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    return _BnReluTrain.apply(x, bn.weight, bn.bias, stats[0], stats[1], momentum, bn.eps, relu, tracked)   # counts in-kernel


class _BN(nn.Sequential):
    """wrapper that yields the reference's double ``bn.bn`` key (pytorch_utils.py:104-111)"""

    def __init__(self, norm_cls, channels: int, name: str = ""):
        super().__init__()
        self.add_module(name + "bn", norm_cls(channels))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0)


class BatchNorm1d(_BN):
    def __init__(self, in_size: int, *, name: str = ""):
        super().__init__(nn.BatchNorm1d, in_size, name)


class BatchNorm2d(_BN):
    def __init__(self, in_size: int, name: str = ""):
        super().__init__(nn.BatchNorm2d, in_size, name)


class _ConvBlock(nn.Sequential):
    """conv -> [bn] -> [activation] (or the pre-activation order), pytorch_utils.py:35-101"""

    def __init__(self, conv_cls, bn_cls, in_cls, in_size, out_size, kernel_size, stride, padding,
                 activation, bn, init, bias, preact, name, instance_norm):
        super().__init__()
        conv = conv_cls(in_size, out_size, kernel_size=kernel_size, stride=stride, padding=padding,
                        bias=bias and (not bn))
        init(conv.weight)
        if conv.bias is not None:
            nn.init.constant_(conv.bias, 0)
        norm_width = in_size if preact else out_size

        def add_norm_act():
            if bn:
                self.add_module(name + "bn", bn_cls(norm_width))
            if activation is not None:
                self.add_module(name + "activation", activation)
            if not bn and instance_norm:
                self.add_module(name + "in", in_cls(norm_width, affine=False, track_running_stats=False))

        if preact:
            add_norm_act()
        self.add_module(name + "conv", conv)
        if not preact:
            add_norm_act()
        ks = kernel_size if isinstance(kernel_size, (tuple, list)) else (kernel_size,)
        st = stride if isinstance(stride, (tuple, list)) else (stride,)
        pd = padding if isinstance(padding, (tuple, list)) else (padding,)
        self._pointwise = (not preact and not instance_norm and name == "" and all(k == 1 for k in ks)
                           and all(v == 1 for v in st) and all(v == 0 for v in pd))

    def _fold_sources(self):
        conv = self.conv
        src = [conv.weight, conv.bias]
        bn = getattr(self, "bn", None)
        if bn is not None:
            bn = bn[0]
            src += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
        return [t for t in src if t is not None]

    def _folded(self):
        """(w (O,C), shift (O,) or None, activation) with eval-mode BatchNorm folded in.  The
        folded pair is cached per block and rebuilt when any source tensor is replaced or
        modified in place (load_state_dict, optimizer step, .to()): 5 tiny launches per block
        and forward otherwise, ~250 launches in a Stage-1 forward."""
        key = tuple((t.data_ptr(), t._version) for t in self._fold_sources())
        cache = self.__dict__.get("_fold_cache")
        if cache is not None and cache[0] == key:
            return cache[1], cache[2], getattr(self, "activation", None)
        conv = self.conv
        with torch.no_grad():
            w = conv.weight.reshape(conv.weight.shape[0], -1)
            shift = conv.bias
            bn = getattr(self, "bn", None)
            if bn is not None:
                bn = bn[0]
                scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
                w = w * scale[:, None]
                shift = bn.bias - bn.running_mean * scale + (0 if shift is None else shift * scale)
            w = w.contiguous()
            shift = None if shift is None else shift.contiguous()
        if not (w.is_cuda and torch.cuda.is_current_stream_capturing()):
            self.__dict__["_fold_cache"] = (key, w, shift)
        return w, shift, getattr(self, "activation", None)

    def fast_path_ok(self, x):
        """the folded-weights path is inference only: the fold runs under no_grad, so it is declined whenever autograd
        could want a path to the input OR to the block's own parameters (eval() with grad enabled), and whenever the
        block's BatchNorm is itself in training mode (it must then see batch statistics -- stock module path)"""
        if self.training or not self._pointwise:
            return False
        bn = getattr(self, "bn", None)
        if bn is not None and bn[0].training:
            return False
        if torch.is_grad_enabled() and (x.requires_grad or any(t.requires_grad for t in self._fold_sources() if t is not None)):
            return False
        return True

    def forward_then_max(self, x):
        """y = max over the last axis of this block's output, computed as
        act(max_s(W x_s) + shift): bias add and ReLU are monotone non-decreasing and rounding is
        monotone, so they commute with the max EXACTLY -- and run on a tensor nsample times
        smaller.  x (B,C,M,S) -> (B,O,M)."""
        w, shift, act = self._folded()
        assert act is None or isinstance(act, nn.ReLU)
        B, C, M, S = x.shape
        y = torch.matmul(w, x.reshape(B, C, M * S))
        if x.is_cuda:
            from . import compat as _C       # max over S + bias + ReLU in one pass over y
            return _C.rowmax_bias_act(y.view(B, -1, M, S), shift, relu=act is not None)
        y = y.reshape(B, -1, M, S).amax(dim=3)
        if shift is not None:
            y = y + shift[None, :, None]
        return torch.relu_(y) if act is not None else y

    def forward(self, x):
        """Inference fast path for the 1x1 case: conv + eval-mode BatchNorm folded into ONE
        (O,C) x (B,C,L) GEMM (+bias, +ReLU) on rocBLAS.  MIOpen resolves several of these 1x1
        NCHW fp32 shapes to its naive direct-convolution kernel on gfx950 (56 % of a Stage-1
        forward in the round-1 profile); training keeps the stock module path."""
        if not self.fast_path_ok(x):
            bn = getattr(self, "bn", None)
            act = getattr(self, "activation", None)
            if (FUSED_BN_TRAIN and self.training and self._pointwise and bn is not None and bn[0].training and x.is_cuda
                    and x.dtype == torch.float32 and bn[0].affine and isinstance(act, (nn.ReLU, type(None)))):
                # training: conv on the library GEMMs, BatchNorm + ReLU in one forward and one backward op
                return bn_relu_train(conv1x1_train(x, self.conv), bn[0], act is not None)
            if (FUSED_CONV_WGRAD and self.training and self._pointwise and bn is None and x.is_cuda and x.dtype == torch.float32
                    and getattr(self, "in", None) is None):
                y = conv1x1_train(x, self.conv)            # head layers without a norm: same weight-gradient kernel
                return act(y) if act is not None else y
            return super().forward(x)
        w, shift, act = self._folded()
        y = torch.matmul(w, x.reshape(x.shape[0], x.shape[1], -1))
        if x.is_cuda and shift is not None and (act is None or isinstance(act, nn.ReLU)):
            from . import compat as _C       # one fused epilogue pass instead of add + clamp
            _C.bias_act_inplace(y, shift, relu=act is not None)
            return y.reshape(x.shape[0], -1, *x.shape[2:])
        if shift is not None:
            y = y + shift[None, :, None]
        if act is not None:
            y = torch.relu_(y) if isinstance(act, nn.ReLU) else act(y)
        return y.reshape(x.shape[0], -1, *x.shape[2:])


class Conv1d(_ConvBlock):
    def __init__(self, in_size: int, out_size: int, *, kernel_size: int = 1, stride: int = 1,
                 padding: int = 0, activation=nn.ReLU(inplace=True), bn: bool = False,
                 init=nn.init.kaiming_normal_, bias: bool = True, preact: bool = False,
                 name: str = "", instance_norm=False):
        super().__init__(nn.Conv1d, BatchNorm1d, nn.InstanceNorm1d, in_size, out_size, kernel_size,
                         stride, padding, activation, bn, init, bias, preact, name, instance_norm)


class Conv2d(_ConvBlock):
    def __init__(self, in_size: int, out_size: int, *, kernel_size: Tuple[int, int] = (1, 1),
                 stride: Tuple[int, int] = (1, 1), padding: Tuple[int, int] = (0, 0),
                 activation=nn.ReLU(inplace=True), bn: bool = False, init=nn.init.kaiming_normal_,
                 bias: bool = True, preact: bool = False, name: str = "", instance_norm=False):
        super().__init__(nn.Conv2d, BatchNorm2d, nn.InstanceNorm2d, in_size, out_size, kernel_size,
                         stride, padding, activation, bn, init, bias, preact, name, instance_norm)


class SharedMLP(nn.Sequential):
    """stack of 1x1 Conv2d blocks named layer0, layer1, ... (pytorch_utils.py:5-32)"""

    def __init__(self, args: List[int], *, bn: bool = False, activation=nn.ReLU(inplace=True),
                 preact: bool = False, first: bool = False, name: str = "", instance_norm: bool = False):
        super().__init__()
        for i in range(len(args) - 1):
            plain = first and preact and i == 0  # the very first pre-act layer has no bn/act
            self.add_module(name + "layer{}".format(i),
                            Conv2d(args[i], args[i + 1], bn=bn and not plain,
                                   activation=None if plain else activation, preact=preact,
                                   instance_norm=instance_norm))


class FC(nn.Sequential):
    """pytorch_utils.py:204-236"""

    def __init__(self, in_size: int, out_size: int, *, activation=nn.ReLU(inplace=True), bn: bool = False,
                 init=None, preact: bool = False, name: str = ""):
        super().__init__()
        fc = nn.Linear(in_size, out_size, bias=not bn)
        if init is not None:
            init(fc.weight)
        if not bn:
            nn.init.constant_(fc.bias, 0)

        def add_norm_act(width):
            if bn:
                self.add_module(name + "bn", BatchNorm1d(width))
            if activation is not None:
                self.add_module(name + "activation", activation)

        if preact:
            add_norm_act(in_size)
        self.add_module(name + "fc", fc)
        if not preact:
            add_norm_act(out_size)
