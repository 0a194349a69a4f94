import collections.abc
from itertools import repeat
import torch
from torch import nn


def to_2tuple(x):
    if isinstance(x, collections.abc.Iterable) and not isinstance(x, str):
        return tuple(x)
    return tuple(repeat(x, 2))


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class DropPath(nn.Module):
    """Stochastic depth; identity in eval mode / p == 0 (the only mode the golden generator uses)."""
    def __init__(self, drop_prob=0.):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * mask / keep


class SqueezeExcite(nn.Module):
    """Only constructed by out-of-scope backbones that the reference imports unconditionally."""
    def __init__(self, channels, rd_ratio=0.25, **kw):
        super().__init__()
        rd = max(1, int(round(channels * rd_ratio)))
        self.fc1 = nn.Conv2d(channels, rd, 1)
        self.act = nn.ReLU(inplace=True)
        self.fc2 = nn.Conv2d(rd, channels, 1)
        self.gate = nn.Sigmoid()

    def forward(self, x):
        s = x.mean((2, 3), keepdim=True)
        return x * self.gate(self.fc2(self.act(self.fc1(s))))
