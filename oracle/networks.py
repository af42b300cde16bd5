"""CPU oracle (test infrastructure): the backbones as plain torch.nn modules with the reference's module trees.

  * EfficientNetV2S  — torchvision 0.14.1 `efficientnet_v2_s` + the reference's classifier (src/networks.py:301-332).
                       torchvision is absent here and not vendored by the reference: PARITY UNPINNED, restated from
                       the published architecture.
  * RepVggNet        — src/networks.py:229-290 + src/repvgg.py:35-76,165-199 (RepVGG-B1g2, train form, never re-parameterised)
  * TinyViT21M       — src/tiny_vit.py:28-591 + head of src/networks.py:407-441
The latter two are pinned against the imported reference by tests/golden/backbones.npz.

Random masks (Dropout, StochasticDepth 'row', DropPath) can be routed through `set_mask_hook` so that the HIP path
and this oracle consume identical Bernoulli draws in parity tests.
"""
import itertools
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

_mask_hook = None


def set_mask_hook(fn):
    """fn(kind, shape, keep) -> tensor of 0 / (1/keep), or None for torch's own RNG."""
    global _mask_hook
    _mask_hook = fn


def _mask(kind, shape, keep, ref):
    if _mask_hook is not None:
        m = _mask_hook(kind, tuple(shape), keep)
        if m is not None:
            return m.to(ref.dtype)
    return torch.empty(shape, dtype=ref.dtype).bernoulli_(keep) / keep


class Dropout(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        return x * _mask("dropout", x.shape, 1.0 - self.p, x)


class StochasticDepth(nn.Module):
    """torchvision.ops.StochasticDepth(p, 'row')"""

    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        m = _mask("stochastic_depth", (x.shape[0],), 1.0 - self.p, x)
        return x * m.view(-1, *([1] * (x.ndim - 1)))


def regression_head(in_dim, h1, h2, num_classes):
    head = nn.Sequential(Dropout(0.2), nn.Linear(in_dim, h1), nn.BatchNorm1d(h1), nn.ReLU6(inplace=False),
                         nn.Linear(h1, h2), nn.BatchNorm1d(h2), nn.ReLU6(inplace=False), nn.Linear(h2, num_classes))
    for m in head:
        if isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, 0, 0.01)
            nn.init.zeros_(m.bias)
    return head


# ------------------------------------------------------------------------------------------------ EfficientNetV2-S
V2_S = [("fused", 1, 3, 1, 24, 24, 2), ("fused", 4, 3, 2, 24, 48, 4), ("fused", 4, 3, 2, 48, 64, 4),
        ("mb", 4, 3, 2, 64, 128, 6), ("mb", 6, 3, 1, 128, 160, 9), ("mb", 6, 3, 2, 160, 256, 15)]


def _cna(cin, cout, k, stride, groups=1, act=True):
    layers = [nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False), nn.BatchNorm2d(cout, eps=1e-3)]
    if act:
        layers.append(nn.SiLU())
    return nn.Sequential(*layers)


class SqueezeExcitation(nn.Module):
    def __init__(self, c, csq):
        super().__init__()
        self.fc1 = nn.Conv2d(c, csq, 1)
        self.fc2 = nn.Conv2d(csq, c, 1)

    def forward(self, x):
        s = F.adaptive_avg_pool2d(x, 1)
        s = torch.sigmoid(self.fc2(F.silu(self.fc1(s))))
        return x * s


class _Block(nn.Module):
    def forward(self, x):
        y = self.block(x)
        if self.use_res_connect:
            y = self.stochastic_depth(y) + x
        return y


class FusedMBConv(_Block):
    def __init__(self, expand, k, stride, cin, cout, sd):
        super().__init__()
        self.use_res_connect = stride == 1 and cin == cout
        exp = cin * expand
        if exp != cin:
            self.block = nn.Sequential(_cna(cin, exp, k, stride), _cna(exp, cout, 1, 1, act=False))
        else:
            self.block = nn.Sequential(_cna(cin, cout, k, stride))
        self.stochastic_depth = StochasticDepth(sd)


class MBConv(_Block):
    def __init__(self, expand, k, stride, cin, cout, sd):
        super().__init__()
        self.use_res_connect = stride == 1 and cin == cout
        exp = cin * expand
        self.block = nn.Sequential(_cna(cin, exp, 1, 1), _cna(exp, exp, k, stride, groups=exp),
                                   SqueezeExcitation(exp, max(1, cin // 4)), _cna(exp, cout, 1, 1, act=False))
        self.stochastic_depth = StochasticDepth(sd)


class EfficientNetV2S(nn.Module):
    def __init__(self, num_classes=9):
        super().__init__()
        feats = [_cna(3, 24, 3, 2)]
        total, idx = sum(c[-1] for c in V2_S), 0
        for kind, expand, k, stride, cin, cout, n in V2_S:
            stage = []
            for i in range(n):
                cls = FusedMBConv if kind == "fused" else MBConv
                stage.append(cls(expand, k, stride if i == 0 else 1, cin if i == 0 else cout, cout, 0.2 * idx / total))
                idx += 1
            feats.append(nn.Sequential(*stage))
        feats.append(_cna(256, 1280, 1, 1))
        self.features = nn.Sequential(*feats)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        self.classifier = regression_head(1280, 512, 128, num_classes)

    def forward(self, x):
        return self.classifier(torch.flatten(self.avgpool(self.features(x)), 1))


# ------------------------------------------------------------------------------------------------ RepVGG-B1g2
class RepVGGBlock(nn.Module):
    """relu(bn(conv3x3(x)) + bn(conv1x1(x)) + [bn(x)])  (src/repvgg.py:35-76, train form)"""

    def __init__(self, cin, cout, stride, groups):
        super().__init__()
        self.rbr_identity = nn.BatchNorm2d(cin) if cin == cout and stride == 1 else None
        self.rbr_dense = nn.Sequential()
        self.rbr_dense.add_module("conv", nn.Conv2d(cin, cout, 3, stride, 1, groups=groups, bias=False))
        self.rbr_dense.add_module("bn", nn.BatchNorm2d(cout))
        self.rbr_1x1 = nn.Sequential()
        self.rbr_1x1.add_module("conv", nn.Conv2d(cin, cout, 1, stride, 0, groups=groups, bias=False))
        self.rbr_1x1.add_module("bn", nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = 0 if self.rbr_identity is None else self.rbr_identity(x)
        return F.relu(self.rbr_dense(x) + self.rbr_1x1(x) + idt)


def repvgg_b1g2_stages():
    """(cin, cout, stride, groups) per block: widths [2,2,2,4], blocks [4,6,16,1], groups 2 on even layer indices
    2..26 (src/repvgg.py:165-199, 213-214, 237-239)."""
    cfg = [(3, 64, 2, 1)]
    stages = [[]]
    in_planes, layer_idx = 64, 1
    for planes, nblocks in ((128, 4), (256, 6), (512, 16), (2048, 1)):
        st = []
        for i in range(nblocks):
            groups = 2 if (layer_idx % 2 == 0 and 2 <= layer_idx <= 26) else 1
            st.append((in_planes, planes, 2 if i == 0 else 1, groups))
            in_planes = planes
            layer_idx += 1
        stages.append(st)
    stages[0] = cfg
    return stages


class RepVggNet(nn.Module):
    def __init__(self, num_classes=9):
        super().__init__()
        st = repvgg_b1g2_stages()
        self.layer0 = RepVGGBlock(*st[0][0])
        for i in range(1, 5):
            setattr(self, f"layer{i}", nn.Sequential(*[RepVGGBlock(*c) for c in st[i]]))
        self.gap = nn.AdaptiveAvgPool2d(1)
        self.linear_reg = regression_head(2048, 512, 128, num_classes)

    def forward(self, x):
        x = self.layer0(x)
        for i in range(1, 5):
            x = getattr(self, f"layer{i}")(x)
        return self.linear_reg(torch.flatten(self.gap(x), 1))
