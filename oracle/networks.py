"""CPU oracle (test infrastructure): the backbones as plain torch.nn modules with the reference's module trees.

  * EfficientNetV2S  — torchvision 0.14.1 `efficientnet_v2_s` + the reference's classifier (src/networks.py:301-332).
                       torchvision is absent here and not vendored by the reference: PARITY UNPINNED against torchvision, restated
                       from the published architecture; the state-dict keys / shapes and the MBConv block (against
                       transformers' EfficientNetBlock, tests/golden/mbconv_hf.npz) are pinned, FusedMBConv and the stage table are not.
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


# ------------------------------------------------------------------------------------------------ TinyViT-21M
class ConvBN(nn.Sequential):
    """src/tiny_vit.py:28-50: children 'c' (conv, no bias) and 'bn'"""

    def __init__(self, a, b, ks=1, stride=1, pad=0, groups=1):
        super().__init__()
        self.add_module("c", nn.Conv2d(a, b, ks, stride, pad, groups=groups, bias=False))
        self.add_module("bn", nn.BatchNorm2d(b))


class DropPath(nn.Module):
    """timm DropPath: per-sample Bernoulli(keep)/keep in train mode"""

    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        m = _mask("drop_path", (x.shape[0],), 1.0 - self.p, x)
        return x * m.view(-1, *([1] * (x.ndim - 1)))


class TvMBConv(nn.Module):
    """src/tiny_vit.py:84-122"""

    def __init__(self, dim, expand, drop_path):
        super().__init__()
        hid = int(dim * expand)
        self.conv1, self.conv2, self.conv3 = ConvBN(dim, hid, 1), ConvBN(hid, hid, 3, 1, 1, groups=hid), ConvBN(hid, dim, 1)
        self.drop_path = DropPath(drop_path)

    def forward(self, x):
        y = F.gelu(self.conv1(x))
        y = F.gelu(self.conv2(y))
        y = self.drop_path(self.conv3(y))
        return F.gelu(y + x)


class TvPatchMerging(nn.Module):
    """src/tiny_vit.py:125-151: 1x1 -> GELU -> dw3x3 s2 -> GELU -> 1x1, tokens out"""

    def __init__(self, res, dim, out_dim):
        super().__init__()
        self.res = res
        self.conv1, self.conv2, self.conv3 = ConvBN(dim, out_dim, 1), ConvBN(out_dim, out_dim, 3, 2, 1, groups=out_dim), ConvBN(out_dim, out_dim, 1)

    def forward(self, x):
        if x.ndim == 3:
            B = x.shape[0]
            x = x.view(B, self.res, self.res, -1).permute(0, 3, 1, 2)
        x = self.conv3(F.gelu(self.conv2(F.gelu(self.conv1(x)))))
        return x.flatten(2).transpose(1, 2)


def attention_bias_index(ws):
    """first-seen order of (|dy|,|dx|) offsets over points x points (src/tiny_vit.py:237-252) -> (N,N) long, count"""
    pts = list(itertools.product(range(ws), range(ws)))
    seen, idx = {}, []
    for p1 in pts:
        for p2 in pts:
            off = (abs(p1[0] - p2[0]), abs(p1[1] - p2[1]))
            if off not in seen:
                seen[off] = len(seen)
            idx.append(seen[off])
    return torch.tensor(idx, dtype=torch.long).view(len(pts), len(pts)), len(seen)


class TvAttention(nn.Module):
    """src/tiny_vit.py:216-286 with attn_ratio = 1"""

    def __init__(self, dim, key_dim, heads, ws):
        super().__init__()
        self.heads, self.kd, self.scale = heads, key_dim, key_dim ** -0.5
        self.norm = nn.LayerNorm(dim)
        self.qkv = nn.Linear(dim, 3 * key_dim * heads)
        self.proj = nn.Linear(key_dim * heads, dim)
        idx, n = attention_bias_index(ws)
        self.attention_biases = nn.Parameter(torch.zeros(heads, n))
        self.register_buffer("attention_bias_idxs", idx, persistent=False)

    def forward(self, x):
        B, N, _ = x.shape
        qkv = self.qkv(self.norm(x)).view(B, N, self.heads, -1)
        q, k, v = (t.permute(0, 2, 1, 3) for t in qkv.split([self.kd, self.kd, self.kd], dim=3))
        attn = (q @ k.transpose(-2, -1)) * self.scale + self.attention_biases[:, self.attention_bias_idxs]
        x = (attn.softmax(-1) @ v).transpose(1, 2).reshape(B, N, self.heads * self.kd)
        return self.proj(x)


class TvMlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.norm, self.fc1, self.fc2 = nn.LayerNorm(dim), nn.Linear(dim, hidden), nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(self.norm(x))))


class TvBlock(nn.Module):
    """src/tiny_vit.py:289-379"""

    def __init__(self, dim, res, heads, ws, drop_path):
        super().__init__()
        self.dim, self.res, self.ws = dim, res, ws
        self.attn = TvAttention(dim, dim // heads, heads, ws)
        self.mlp = TvMlp(dim, dim * 4)
        self.local_conv = ConvBN(dim, dim, 3, 1, 1, groups=dim)
        self.drop_path = DropPath(drop_path)

    def forward(self, x):
        B, L, C = x.shape
        H = W = self.res
        res_x = x
        if H == self.ws:
            x = self.attn(x)
        else:
            assert H % self.ws == 0, "padding path not exercised at 224x224"
            n = H // self.ws
            x = x.view(B, n, self.ws, n, self.ws, C).transpose(2, 3).reshape(B * n * n, self.ws * self.ws, C)
            x = self.attn(x)
            x = x.view(B, n, n, self.ws, self.ws, C).transpose(2, 3).reshape(B, L, C)
        x = res_x + self.drop_path(x)
        x = self.local_conv(x.transpose(1, 2).reshape(B, C, H, W)).view(B, C, L).transpose(1, 2)
        return x + self.drop_path(self.mlp(x))


class _Stage(nn.Module):
    def __init__(self, blocks, downsample):
        super().__init__()
        self.blocks = nn.ModuleList(blocks)
        if downsample is not None:
            self.downsample = downsample

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return self.downsample(x) if hasattr(self, "downsample") else x


class TinyViT21M(nn.Module):
    """tiny_vit_21m_224 (src/tiny_vit.py:666-676) + the reference head (src/networks.py:419-428)"""
    DIMS, DEPTHS, HEADS, WS = (96, 192, 384, 576), (2, 2, 6, 2), (3, 6, 12, 18), (7, 7, 14, 7)

    def __init__(self, num_classes=9, drop_path_rate=0.2):
        super().__init__()
        d = self.DIMS
        self.patch_embed = nn.Module()
        self.patch_embed.seq = nn.Sequential(ConvBN(3, d[0] // 2, 3, 2, 1), nn.GELU(), ConvBN(d[0] // 2, d[0], 3, 2, 1))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(self.DEPTHS))]
        self.layers = nn.ModuleList()
        res, k = 56, 0
        for i in range(4):
            out_dim = d[min(i + 1, 3)]
            if i == 0:
                blocks = [TvMBConv(d[0], 4.0, dpr[k + j]) for j in range(self.DEPTHS[0])]
            else:
                blocks = [TvBlock(d[i], res, self.HEADS[i], self.WS[i], dpr[k + j]) for j in range(self.DEPTHS[i])]
            k += self.DEPTHS[i]
            down = TvPatchMerging(res, d[i], out_dim) if i < 3 else None
            self.layers.append(_Stage(blocks, down))
            res //= 2 if i < 3 else 1
        self.norm_head = nn.LayerNorm(d[3])
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)
        for m in self.modules():
            if isinstance(m, TvMBConv):
                nn.init.zeros_(m.conv3.bn.weight)           # bn_weight_init=0.0 (tiny_vit.py:100-101)
        self.head = regression_head(d[3], 128, 64, num_classes)

    def forward(self, x):
        x = self.patch_embed.seq(x)
        for layer in self.layers:
            x = layer(x)
        return self.head(self.norm_head(x.mean(1)))


# ------------------------------------------------------------------------------------------------ ResNet-50
class Bottleneck(nn.Module):
    """torchvision resnet.Bottleneck (v1.5: the stride sits on the 3x3 conv), expansion 4"""

    def __init__(self, cin, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return F.relu(out + (x if self.downsample is None else self.downsample(x)))


class ResNet50(nn.Module):
    """torchvision 0.14.1 `resnet50` with `fc` replaced by the reference's regression head (src/networks.py:443-500; released
    checkpoint DAD-COCOHead-ResNet50-best.pth).  torchvision is absent here: unpinned against torchvision itself; the body IS pinned against
    an independent implementation of the same network (transformers' ResNetModel, tests/golden/resnet50_hf.npz).  Restated from the published
    architecture; sanity pin: 25,557,032 parameters with the 1000-class fc."""

    def __init__(self, num_classes=9):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), start=1):
            layers = []
            for b in range(blocks):
                s = stride if b == 0 else 1
                down = None
                if s != 1 or cin != planes * 4:
                    down = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, s, bias=False), nn.BatchNorm2d(planes * 4))
                layers.append(Bottleneck(cin, planes, s, down))
                cin = planes * 4
            setattr(self, f"layer{i}", nn.Sequential(*layers))
        self.fc = regression_head(2048, 512, 128, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")

    def forward(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(x.mean((2, 3)))
