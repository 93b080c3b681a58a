"""DepthwiseSeparableConv -- the exported conv layer of the reference (layers/conv.py:10-64) on MI355X.

Named by the north star but never instantiated by the reference decoder, so it is a standalone op:
depthwise Conv1d(k, stride, pad k//2, groups=C, no bias) -> pointwise Conv1d(1x1, bias) -> GroupNorm(min(8, C_out)) -> erf-GELU,
channel-first [B, C, T] -> [B, C_out, (T + 2*(k//2) - k)//stride + 1].  State-dict keys match the reference: depthwise.weight
[C,1,k], pointwise.weight [Co,C,1], pointwise.bias, norm.weight, norm.bias.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import native


class _Leaf(nn.Module):
    pass


class DepthwiseSeparableConv(nn.Module):
    def __init__(self, in_ch: int, out_ch: int, kernel_size: int = 3, stride: int = 1):
        super().__init__()
        if stride < 1:
            raise ValueError("stride must be >= 1")
        self.in_ch, self.out_ch, self.kernel_size, self.stride = in_ch, out_ch, kernel_size, int(stride)
        self.groups = min(8, out_ch)
        self.depthwise, self.pointwise, self.norm = _Leaf(), _Leaf(), _Leaf()
        bd = 1.0 / math.sqrt(kernel_size)
        self.depthwise.weight = nn.Parameter(torch.empty(in_ch, 1, kernel_size).uniform_(-bd, bd), requires_grad=False)
        bp = 1.0 / math.sqrt(in_ch)
        self.pointwise.weight = nn.Parameter(torch.empty(out_ch, in_ch, 1).uniform_(-bp, bp), requires_grad=False)
        self.pointwise.bias = nn.Parameter(torch.empty(out_ch).uniform_(-bp, bp), requires_grad=False)
        self.norm.weight = nn.Parameter(torch.ones(out_ch), requires_grad=False)
        self.norm.bias = nn.Parameter(torch.zeros(out_ch), requires_grad=False)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return native.dsconv_forward(x.contiguous(), self.depthwise.weight.reshape(self.in_ch, self.kernel_size).contiguous(),
                                     self.pointwise.weight.reshape(self.out_ch, self.in_ch).contiguous(), self.pointwise.bias,
                                     self.norm.weight, self.norm.bias, self.groups, self.stride)
