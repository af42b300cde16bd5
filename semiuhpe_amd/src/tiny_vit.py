"""TinyViT-21M (224) backbone with the reference's module tree (src/tiny_vit.py:28-591, variant :666-676; head
src/networks.py:407-441), as Layers with explicit forward/backward over the HIP kernels.

Everything stays in the (B, H, W, C) pixel-major layout: the reference's NCHW <-> token transposes
(tiny_vit.py:141-150, 370-374) and the window partition / reverse copies (:356-366) disappear — convolutions, LayerNorm,
Linear and the windowed attention kernel all address the same tensor.
"""
import itertools

import os

import torch
import torch.nn as nn

from .. import kernels as K
from ..layers import (BatchNorm2d, Conv2d, Dropout, Identity, Layer, LayerNorm, Linear, Network, Sequential, conv_bn_fwd, draw_mask,
                      grad_of)
from .efficientnet_v2 import bn_sums_from_dw_ok, dw_dgrad_expand_bwd, init_regression_head_, regression_head


class Conv2d_BN(Layer):
    """src/tiny_vit.py:28-50: children 'c' (conv, no bias) and 'bn'; the activation that follows is fused into bn."""

    def __init__(self, a, b, ks=1, stride=1, pad=0, groups=1, bn_weight_init=1.0, act="none"):
        super().__init__()
        self.c = Conv2d(a, b, ks, stride, pad, groups, bias=False)
        self.bn = BatchNorm2d(b, act=act)
        nn.init.constant_(self.bn.weight, bn_weight_init)

    def fwd(self, x, tape):
        return conv_bn_fwd(self.c, self.bn, x, tape)

    def bwd(self, dy, tape):
        return self.c.bwd(self.bn.bwd(dy, tape), tape)


class DropPath(Layer):
    """timm DropPath fused with the residual add: y = skip + branch * mask[n]/keep"""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def draw(self, n, device):
        """the keep scale of this call (or None): for callers that fold `skip + scale * branch` into the branch's last GEMM (RESIDUAL_EPILOGUE)
        and push it on the tape themselves, after the branch's own entries"""
        if self.training and self.drop_prob > 0.0:
            return draw_mask("drop_path", (n,), 1.0 - self.drop_prob, device)
        return None

    def fwd2(self, branch, skip, tape):
        rs = self.draw(branch.shape[0], branch.device)
        if tape is not None:
            tape.append(rs)
        return K.add_rowscale(skip, branch, rs)

    def bwd2(self, dy, tape):
        rs = tape.pop()
        return (dy if rs is None else K.add_rowscale(None, dy, rs)), dy


class PatchEmbed(Layer):
    def __init__(self, in_chans, embed_dim):
        super().__init__()
        self.seq = Sequential(Conv2d_BN(in_chans, embed_dim // 2, 3, 2, 1, act="gelu"), Identity(), Conv2d_BN(embed_dim // 2, embed_dim, 3, 2, 1))
        self.seq[0].c.need_input_grad = False

    def fwd(self, x, tape):
        return self.seq.fwd(x, tape)

    def bwd(self, dy, tape):
        return self.seq.bwd(dy, tape)


class MBConv(Layer):
    """src/tiny_vit.py:84-122: 1x1 -> GELU -> dw3x3 -> GELU -> 1x1 (BN weight init 0) -> drop_path -> + shortcut -> GELU"""

    def __init__(self, in_chans, out_chans, expand_ratio, drop_path):
        super().__init__()
        hid = int(in_chans * expand_ratio)
        self.conv1 = Conv2d_BN(in_chans, hid, 1, act="gelu")
        self.conv2 = Conv2d_BN(hid, hid, 3, 1, 1, groups=hid, act="gelu")
        self.conv3 = Conv2d_BN(hid, out_chans, 1, bn_weight_init=0.0)
        self.drop_path = DropPath(drop_path)

    def fwd(self, x, tape):
        y = self.conv3.fwd(self.conv2.fwd(self.conv1.fwd(x, tape), tape), tape)
        z = self.drop_path.fwd2(y, x, tape)
        if tape is not None:
            tape.append(z)
        return K.act_fwd(z, z.shape[-1], "gelu")

    def bwd(self, dy, tape):
        z = tape.pop()
        dz = K.act_bwd(z, dy, z.shape[-1], "gelu")
        d_branch, d_skip = self.drop_path.bwd2(dz, tape)
        d3 = self.conv3.bwd(d_branch, tape)
        dy2 = self.conv2.bn.bwd(d3, tape)
        if bn_sums_from_dw_ok(self.conv2.c, tape):
            # the depthwise input-gradient kernel also reduces conv1's BatchNorm backward sums (one read of conv1's raw output there
            # instead of a reduction pass over it and the gradient: the 4x-expanded 56 x 56 tensors are the largest of the network)
            dx = dw_dgrad_expand_bwd(self.conv2.c, (self.conv1.c, self.conv1.bn), dy2, tape)
        else:
            dx = self.conv1.bwd(self.conv2.c.bwd(dy2, tape), tape)
        return K.add(dx, d_skip)


class PatchMerging(Layer):
    """src/tiny_vit.py:125-151"""

    def __init__(self, dim, out_dim):
        super().__init__()
        self.conv1 = Conv2d_BN(dim, out_dim, 1, act="gelu")
        self.conv2 = Conv2d_BN(out_dim, out_dim, 3, 2, 1, groups=out_dim, act="gelu")
        self.conv3 = Conv2d_BN(out_dim, out_dim, 1)

    def fwd(self, x, tape):
        return self.conv3.fwd(self.conv2.fwd(self.conv1.fwd(x, tape), tape), tape)

    def bwd(self, dy, tape):
        return self.conv1.bwd(self.conv2.bwd(self.conv3.bwd(dy, tape), tape), tape)


def attention_bias_index(ws):
    """offsets (|dy|,|dx|) numbered in first-seen order over points x points (src/tiny_vit.py:237-252)"""
    pts = list(itertools.product(range(ws), range(ws)))
    seen, idx = {}, []
    for p1 in pts:
        for p2 in pts:
            off = (abs(p1[0] - p2[0]), abs(p1[1] - p2[1]))
            if off not in seen:
                seen[off] = len(seen)
            idx.append(seen[off])
    return torch.tensor(idx, dtype=torch.long).view(len(pts), len(pts)), len(seen)


DETERMINISTIC_DBIAS = os.environ.get("SUHPE_ATOMIC_DBIAS", "0") != "1"     # fixed-order bias-table gradient (no atomics)


# windowed attention from row statistics (the backward recomputes the probabilities) instead of saved N x N probabilities; 0 = the round-3 kernels
ATTN_RECOMPUTE = os.environ.get("SUHPE_ATTN_RECOMPUTE", "1") != "0"


class Attention(Layer):
    """src/tiny_vit.py:216-286 (attn_ratio = 1): LayerNorm -> qkv -> windowed softmax(q k^T s + bias) v -> proj"""

    def __init__(self, dim, key_dim, num_heads, window):
        super().__init__()
        self.dim, self.key_dim, self.num_heads, self.window = dim, key_dim, num_heads, window
        self.scale = key_dim ** -0.5
        self.norm = LayerNorm(dim)
        self.qkv = Linear(dim, 3 * key_dim * num_heads)
        self.proj = Linear(key_dim * num_heads, dim)
        idx, n = attention_bias_index(window)
        self.attention_biases = nn.Parameter(torch.zeros(num_heads, n))
        self.register_buffer("attention_bias_idxs", idx, persistent=False)
        self._idx32 = None
        self._inverse = None

    def _inverse_index(self, device):
        """CSR inverse of attention_bias_idxs (pairs i*N + j grouped by table offset): the fixed-order bias-gradient reduction"""
        if self._inverse is None or self._inverse[0].device != device:
            flat = self.attention_bias_idxs.reshape(-1).cpu()
            order = torch.argsort(flat, stable=True)
            counts = torch.bincount(flat, minlength=self.attention_biases.shape[1])
            ptrs = torch.zeros(counts.numel() + 1, dtype=torch.long)
            ptrs[1:] = torch.cumsum(counts, 0)
            self._inverse = (ptrs.to(device=device, dtype=torch.int32).contiguous(), order.to(device=device, dtype=torch.int32).contiguous())
        return self._inverse

    def _index(self, device):
        if self._idx32 is None or self._idx32.device != device:
            self._idx32 = self.attention_bias_idxs.to(device=device, dtype=torch.int32).contiguous()
        return self._idx32

    def fwd(self, x, tape, residual=None):
        """residual = (skip, keep scale per sample or None): x + drop_path(attn(x)) closed by the proj GEMM's epilogue (src/tiny_vit.py:366)"""
        B, H, W, C = x.shape
        xn = self.norm.fwd(x, tape)
        qkv = self.qkv.fwd(xn.view(-1, C), tape)
        if ATTN_RECOMPUTE:
            out, stats = K.window_attn_fwd_rs(qkv, self.attention_biases, self._index(x.device), B, H, W, self.window, self.num_heads,
                                              self.scale, save_stats=tape is not None)
            if tape is not None:
                tape.append((qkv, (out, stats), (B, H, W)))
        else:
            out, probs = K.window_attn_fwd(qkv, self.attention_biases, self._index(x.device), B, H, W, self.window, self.num_heads,
                                           self.scale, save_probs=tape is not None)
            if tape is not None:
                tape.append((qkv, probs, (B, H, W)))
        if residual is not None:
            return self.proj.fwd(out, tape, skip=residual[0], rowscale=residual[1], rows_per_sample=H * W).view(B, H, W, C)
        return self.proj.fwd(out, tape).view(B, H, W, C)

    def bwd(self, dy, tape, skip=None):
        """skip: the gradient of the residual connection around the block, added by the LayerNorm backward's kernel"""
        B, H, W, C = dy.shape
        d_out = self.proj.bwd(dy.view(-1, C), tape)
        qkv, probs, _ = tape.pop()
        inverse = self._inverse_index(dy.device) if DETERMINISTIC_DBIAS else None
        if isinstance(probs, tuple):                                   # (forward output, row statistics): the recomputing kernels
            dqkv = K.window_attn_bwd_rs(qkv, probs[0], probs[1], d_out, self.attention_biases, self._index(dy.device), B, H, W, self.window,
                                        self.num_heads, self.attention_biases.shape[1], self.scale, grad_of(self.attention_biases), inverse=inverse)
        else:
            dqkv = K.window_attn_bwd(qkv, probs, d_out, self._index(dy.device), B, H, W, self.window, self.num_heads,
                                     self.attention_biases.shape[1], self.scale, grad_of(self.attention_biases), inverse=inverse)
        d_xn = self.qkv.bwd(dqkv, tape)
        return self.norm.bwd(d_xn.view(B, H, W, C), tape, skip=skip)


MLP_ACT_EPILOGUE = os.environ.get("SUHPE_MLP_EPILOGUE", "1") != "0"     # GELU / GELU' in the fc1 / fc2 GEMM epilogues
RESIDUAL_EPILOGUE = os.environ.get("SUHPE_RESIDUAL_EPILOGUE", "1") != "0"   # x + drop_path(branch) in the proj / fc2 GEMM epilogues


class Mlp(Layer):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.norm = LayerNorm(in_features)
        self.fc1 = Linear(in_features, hidden_features)
        self.fc2 = Linear(hidden_features, in_features)
        self.hidden = hidden_features

    # fc1 -> GELU -> fc2 (reference src/tiny_vit.py:205-213) with the activation in the GEMM epilogues (MLP_ACT_EPILOGUE): fc1 writes the
    # pre-activation h (kept for the backward) and gelu(h) from one launch, and fc2's input gradient comes back multiplied by gelu'(h):
    # the two elementwise passes over the 4x-wide tensors (read h, write a; read h and da, write dh) and the tensor da disappear.
    def _fused(self):
        return MLP_ACT_EPILOGUE and self.fc1.act_epilogue_ok() and self.fc2.act_epilogue_ok()

    def fwd(self, x, tape, residual=None):
        """residual = (skip, keep scale per sample or None): x + drop_path(mlp(x)) closed by the fc2 GEMM's epilogue (src/tiny_vit.py:378)"""
        shp = x.shape
        xn = self.norm.fwd(x, tape).view(-1, shp[-1])
        if self._fused():
            h, a = self.fc1.fwd_act(xn, tape, "gelu")
        else:
            h = self.fc1.fwd(xn, tape)
            a = K.act_fwd(h, self.hidden, "gelu")
        if tape is not None:
            tape.append(h)
        if residual is not None:
            return self.fc2.fwd(a, tape, skip=residual[0], rowscale=residual[1], rows_per_sample=x.numel() // (shp[0] * shp[-1])).view(shp)
        return self.fc2.fwd(a, tape).view(shp)

    def bwd(self, dy, tape, skip=None):
        shp = dy.shape
        h = tape[-2]                                         # (fc2's input entry is on top of it)
        if self._fused():
            dh = self.fc2.bwd(dy.view(-1, shp[-1]), tape, pre=h, act="gelu")
            tape.pop()
        else:
            da = self.fc2.bwd(dy.view(-1, shp[-1]), tape)
            tape.pop()
            dh = K.act_bwd(h, da, self.hidden, "gelu")
        return self.norm.bwd(self.fc1.bwd(dh, tape).view(shp), tape, skip=skip)


class TinyViTBlock(Layer):
    """src/tiny_vit.py:289-379"""

    def __init__(self, dim, input_resolution, num_heads, window_size=7, mlp_ratio=4.0, drop_path=0.0):
        super().__init__()
        self.dim, self.input_resolution, self.window_size = dim, input_resolution, window_size
        self.drop_path = DropPath(drop_path)
        self.attn = Attention(dim, dim // num_heads, num_heads, window_size)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.local_conv = Conv2d_BN(dim, dim, 3, 1, 1, groups=dim)

    def fwd(self, x, tape):
        if not RESIDUAL_EPILOGUE:
            a = self.attn.fwd(x, tape)
            x1 = self.drop_path.fwd2(a, x, tape)
            x2 = self.local_conv.fwd(x1, tape)
            m = self.mlp.fwd(x2, tape)
            return self.drop_path.fwd2(m, x2, tape)
        # x + drop_path(attn(x)) and x + drop_path(mlp(x)) from the proj / fc2 GEMM epilogues: the keep scales are drawn in the same order as
        # by fwd2 and pushed on the tape where fwd2 pushes them (after the branch's entries), so the backward is unchanged
        rs = self.drop_path.draw(x.shape[0], x.device)
        x1 = self.attn.fwd(x, tape, residual=(x, rs))
        if tape is not None:
            tape.append(rs)
        x2 = self.local_conv.fwd(x1, tape)
        rs = self.drop_path.draw(x2.shape[0], x2.device)
        y = self.mlp.fwd(x2, tape, residual=(x2, rs))
        if tape is not None:
            tape.append(rs)
        return y

    def bwd(self, dy, tape):
        d_m, d_x2 = self.drop_path.bwd2(dy, tape)
        d_x2 = self.mlp.bwd(d_m, tape, skip=d_x2)            # (+ the skip gradient: in the LayerNorm backward's kernel)
        d_x1 = self.local_conv.bwd(d_x2, tape)
        d_a, d_x = self.drop_path.bwd2(d_x1, tape)
        return self.attn.bwd(d_a, tape, skip=d_x)


class _Stage(Layer):
    def __init__(self, blocks, downsample):
        super().__init__()
        self.blocks = Sequential(*blocks)
        if downsample is not None:
            self.downsample = downsample

    def fwd(self, x, tape):
        x = self.blocks.fwd(x, tape)
        return self.downsample.fwd(x, tape) if hasattr(self, "downsample") else x

    def bwd(self, dy, tape):
        if hasattr(self, "downsample"):
            dy = self.downsample.bwd(dy, tape)
        return self.blocks.bwd(dy, tape)


class TinyViT21M(Network):
    """tiny_vit_21m_224: embed_dims 96/192/384/576, depths 2/2/6/2, heads 3/6/12/18, windows 7/7/14/7, drop_path 0.2"""
    DIMS, DEPTHS, HEADS, WS = (96, 192, 384, 576), (2, 2, 6, 2), (3, 6, 12, 18), (7, 7, 14, 7)

    def __init__(self, num_classes=9, drop_path_rate=0.2):
        super().__init__()
        d = self.DIMS
        self.patch_embed = PatchEmbed(3, d[0])
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(self.DEPTHS))]
        stages, res, k = [], 56, 0
        for i in range(4):
            if i == 0:
                blocks = [MBConv(d[0], d[0], 4.0, dpr[k + j]) for j in range(self.DEPTHS[0])]
            else:
                blocks = [TinyViTBlock(d[i], (res, res), self.HEADS[i], self.WS[i], 4.0, dpr[k + j]) for j in range(self.DEPTHS[i])]
            k += self.DEPTHS[i]
            stages.append(_Stage(blocks, PatchMerging(d[i], d[min(i + 1, 3)]) if i < 3 else None))
            if i < 3:
                res //= 2
        self.layers = Sequential(*stages)
        self.norm_head = LayerNorm(d[3])
        for m in self.modules():
            if isinstance(m, Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)
        self.head = regression_head(d[3], 128, 64, num_classes)
        init_regression_head_(self.head)

    def fwd(self, x, tape):
        if self.training:                   # every DropPath mask of the pass from one launch (same values as draw-by-draw)
            from ..layers import prefetch_masks
            reqs = []
            for stage in self.layers:
                for blk in stage.blocks:
                    dp = blk.drop_path
                    if dp.training and dp.drop_prob > 0.0:
                        reqs += [("drop_path", (x.shape[0],), 1.0 - dp.drop_prob)] * (2 if isinstance(blk, TinyViTBlock) else 1)
            prefetch_masks(reqs, x.device)
        x = self.layers.fwd(self.patch_embed.fwd(x, tape), tape)
        if tape is not None:
            tape.append(x.shape)
        return self.head.fwd(self.norm_head.fwd(K.avgpool_fwd(x), tape), tape)

    def bwd(self, dy, tape):
        dy = self.norm_head.bwd(self.head.bwd(dy, tape), tape)
        dy = K.avgpool_bwd(dy, tape.pop())
        return self.patch_embed.bwd(self.layers.bwd(dy, tape), tape)


def get_TinyViT(config, pretrain=True, model_name="21m"):
    """src/networks.py:407-441 (pretrained blobs are not available offline; same init recipe)."""
    if model_name != "21m":
        raise NotImplementedError("the reference hard-codes the 21m variant (src/networks.py:29)")
    return TinyViT21M(num_classes=config.num_classes)
