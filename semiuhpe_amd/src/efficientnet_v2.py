"""EfficientNetV2-S backbone with the module tree (and therefore state_dict keys) of torchvision 0.14.1's
`efficientnet_v2_s`, which the reference instantiates at src/networks.py:301-303 and whose `classifier` it replaces
(:309-331).  torchvision itself is not part of the reference repo and is not installed here: the architecture is
restated from its published definition — parity of this backbone body is UNPINNED
(sanity-pinned by parameter count 20,177,488 and 2.849 GMAC @224 in tests/test_networks_cpu.py).

Blocks are Layers with explicit forward/backward over the HIP kernels (semiuhpe_amd/layers.py).
"""
import os

import torch
import torch.nn as nn

from .. import kernels as K
from ..layers import (conv_bn_fwd, draw_mask, grad_of, BatchNorm1d, Conv2d, ConvNormActivation, Dropout, GlobalAvgPool, Identity, Layer, Linear, Network,
                      Sequential, SqueezeExcitation, StochasticDepth)

# (block, expand_ratio, kernel, stride, in, out, layers) — torchvision `_efficientnet_conf("efficientnet_v2_s")`
V2_S = [
    ("fused", 1, 3, 1, 24, 24, 2),
    ("fused", 4, 3, 2, 24, 48, 4),
    ("fused", 4, 3, 2, 48, 64, 4),
    ("mb", 4, 3, 2, 64, 128, 6),
    ("mb", 6, 3, 1, 128, 160, 9),
    ("mb", 6, 3, 2, 160, 256, 15),
]
LAST_CHANNEL = 1280
BN_EPS = 1e-3           # torchvision: partial(nn.BatchNorm2d, eps=1e-03) for the V2 family
# Folding the expand conv's BatchNorm + SiLU into the depthwise loaders removes two passes over the largest activation.  ON by default since
# round 3: the image-tile depthwise kernels apply the transform once per element while they fill LDS (57.1 -> 56.4 ms/step on C2); with the
# sliding-window kernels of round 2 the SiLU was recomputed 1.5x (forward) to 3x (filter gradient) per element and the fold lost (84.1 vs
# 79.7 ms/step), which is why SUHPE_FUSE_EXPAND_DW=0 still exists.
FUSE_EXPAND_BN_INTO_DW = os.environ.get("SUHPE_FUSE_EXPAND_DW", "1") != "0"
BN_SUMS_FROM_PROJECT = os.environ.get("SUHPE_BN_SUMS_PROJECT", "1") != "0"     # FusedMBConv: expand BatchNorm's backward sums from the project conv's dgrad epilogue
FUSE_EXPAND_NO_TAPE = os.environ.get("SUHPE_FUSE_EXPAND_NO_TAPE", "1") != "0"      # the same fold in train-mode forwards that keep no tape (teacher pass)
# the expand BatchNorm's backward sums reduced by the depthwise input-gradient kernel (no reduction pass over the expanded tensor pair)
BN_SUMS_FROM_DW = os.environ.get("SUHPE_BN_SUMS_FROM_DW", "1") != "0"
FUSE_DW_BN_SE = os.environ.get("SUHPE_FUSE_DW_BN_SE", "1") != "0"     # depthwise BatchNorm + SiLU + SqueezeExcitation as one section (MBConv._dw_se_*)
STOCHASTIC_DEPTH_PROB = 0.2


def _make_divisible(v, divisor=8):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def bn_sums_from_dw_ok(conv_d, tape):
    """the tape's next two entries are the depthwise conv's plain input and the expand BatchNorm's train-mode record in a form the
    fused kernels take (one tensor of statistics, or the (mean[2, c], rstd[2, c]) pair of a merged two-segment pass)"""
    if not BN_SUMS_FROM_DW or len(tape) < 2:
        return False
    top = tape[-1]
    if isinstance(top, tuple) and len(top) == 2 and torch.is_tensor(top[0]):      # (raw input, input transform): folded expand BatchNorm
        top = top[0]
    if not torch.is_tensor(top):
        return False
    e = tape[-2]
    if not (isinstance(e, tuple) and len(e) == 7 and e[3] is True and e[4] is None and e[5] is None):
        return False
    if not (torch.is_tensor(e[1]) or (isinstance(e[1], tuple) and len(e[1]) == 2 and torch.is_tensor(e[1][0]))):
        return False
    return K.dwconv_dgrad_bnsums_ok(top.shape, conv_d.weight.shape, conv_d.stride, conv_d.padding)

# The depthwise conv's whole backward as ONE launch (suhpe_dwconv_bwd_fused, round 6): the filter gradient comes from the same staged tile of the
# output gradient as the input gradient and the expand BatchNorm's sums, and -- inside the fused depthwise / SqueezeExcitation section -- that output
# gradient is formed while the tile is filled, so se_bn_bwd_apply's pass and the separate filter-gradient kernel disappear (8 -> 4 passes over the
# expanded tensor, two launches fewer per MBConv block).  SUHPE_DW_BWD_FUSED=0 (read by the library): the three-kernel path.
def dw_bwd_fused_ok(conv_d, tape):
    """the folded-expand form on a shape the fused kernel takes: the tape's top is (raw expand output, its BatchNorm as input transform) and the
    entry below it that BatchNorm's train-mode record of the SAME tensor"""
    top, rec = tape[-1], tape[-2]
    return (isinstance(top, tuple) and len(top) == 2 and torch.is_tensor(top[0]) and top[1] is not None and top[1][0] == "bn" and rec[0] is top[0]
            and K.dwconv_bwd_fused_ok(top[0].shape, conv_d.weight.shape, conv_d.stride, conv_d.padding))


def dw_dgrad_expand_bwd(conv_d, expand, dy, tape, mid=None):
    """depthwise filter gradient; depthwise input gradient + the expand BatchNorm's (sum g, sum g xhat) from ONE kernel; that
    BatchNorm's apply pass; the expand conv's backward.  mid: see kernels.dwconv_bwd_fused (dy is then the section's dout)"""
    conv_e, bn_e = expand
    fused = dw_bwd_fused_ok(conv_d, tape)
    assert fused or mid is None
    a1, xf = tape.pop(), None
    if isinstance(a1, tuple):                            # folded: the filter gradient recomputes act(BN(a1)) while it stages a1
        a1, xf = a1
    if not fused:
        K.conv2d_wgrad(a1, dy, conv_d.weight.shape, conv_d.stride, conv_d.padding, conv_d.groups, out=grad_of(conv_d.weight), accumulate=True, xf=xf)
    x1, mean, rstd, _, _, _, hw = tape.pop()
    cexp = bn_e.num_features
    if fused:
        two = isinstance(mean, tuple)
        (m2, r2), seg = (mean, rstd) if two else ((mean, rstd), 0)
        d, sums = K.dwconv_bwd_fused(dy, conv_d.weight, a1.shape, conv_d.stride, conv_d.padding, x1, m2, r2, bn_e.weight, bn_e.bias, bn_e.act, seg,
                                     grad_of(bn_e.weight), grad_of(bn_e.bias), grad_of(conv_d.weight), mid=mid)
        if two:
            dz = K.bn_act_bwd_apply_seg(x1, d, seg * hw, cexp, m2, r2, bn_e.weight, bn_e.bias, bn_e.act, sums)
        else:
            dz = K.bn_act_bwd_apply(x1, d, cexp, mean, rstd, bn_e.weight, bn_e.bias, bn_e.act, sums, x1.numel() // cexp)
        return conv_e.bwd(dz, tape)
    if isinstance(mean, tuple):                          # two batch segments: ((mean[2, c], rstd[2, c]), samples of the first)
        (m2, r2), seg = mean, rstd
        d, sums = K.dwconv_dgrad_bnsums(dy, conv_d.weight, a1.shape, conv_d.stride, conv_d.padding, x1, m2, r2, bn_e.weight, bn_e.bias, bn_e.act,
                                        seg, grad_of(bn_e.weight), grad_of(bn_e.bias))
        dz = K.bn_act_bwd_apply_seg(x1, d, seg * hw, cexp, m2, r2, bn_e.weight, bn_e.bias, bn_e.act, sums)
    else:
        d, sums = K.dwconv_dgrad_bnsums(dy, conv_d.weight, a1.shape, conv_d.stride, conv_d.padding, x1, mean, rstd, bn_e.weight, bn_e.bias,
                                        bn_e.act, 0, grad_of(bn_e.weight), grad_of(bn_e.bias))
        dz = K.bn_act_bwd_apply(x1, d, cexp, mean, rstd, bn_e.weight, bn_e.bias, bn_e.act, sums, x1.numel() // cexp)
    return conv_e.bwd(dz, tape)



class _ResidualBlock(Layer):
    """Shared tail: x + StochasticDepth('row')(block(x)).  The add and the per-sample keep scale run inside the last
    BatchNorm's apply kernel (and its backward), so the skip costs no pass of its own."""

    def _front_fwd(self, x, tape):
        """every layer of the block but the last"""
        for layer in list(self.block)[:-1]:
            x = layer.fwd(x, tape)
        return x

    def _front_bwd(self, dy, tape):
        for layer in reversed(list(self.block)[:-1]):
            dy = layer.bwd(dy, tape)
        return dy

    def fwd(self, x, tape):
        y = self._front_fwd(x, tape)
        last = self.block[-1]
        xf = None
        if isinstance(y, tuple):            # (raw tensor, input transform): the last conv applies the front's BatchNorm + activation itself
            y, xf = y
        rowscale = None
        if self.use_res_connect:
            sd = self.stochastic_depth
            if sd.training and sd.p > 0.0:
                rowscale = draw_mask("stochastic_depth", (x.shape[0],), 1.0 - sd.p, x.device)
        res = x if self.use_res_connect else None
        if xf is not None:
            y1, partials = last[0].fwd_stats(y, tape, xf=xf)
            return last[1].fwd(y1, tape, residual=res, partials=partials, post=self.use_res_connect, rowscale=rowscale)
        if not self.use_res_connect:
            return last.fwd(y, tape)
        return conv_bn_fwd(last[0], last[1], y, tape, residual=x, residual_after_act=True, rowscale=rowscale)

    def bwd(self, dy, tape):
        from ..layers import park_skip
        first = self.block[0]
        from .. import layers as L
        fused = self.use_res_connect and park_skip(first[0] if isinstance(first, ConvNormActivation) else None, dy)
        try:
            d = self._front_bwd(self._last_bwd(dy, tape), tape)         # (the last BatchNorm's backward applies the keep scale)
            if fused:                       # the first conv's input-gradient epilogue added dy
                assert L._PENDING_SKIP is None, "the parked skip gradient was not consumed"
                return d
        finally:
            L._PENDING_SKIP = None          # an exception between park and take must not poison every later backward of the process
        return K.add(d, dy) if self.use_res_connect else d


    def _last_bwd(self, dy, tape):
        return self.block[-1].bwd(dy, tape)


class FusedMBConv(_ResidualBlock):
    def __init__(self, expand, kernel, stride, cin, cout, sd_prob):
        super().__init__()
        self.use_res_connect = stride == 1 and cin == cout
        exp = _make_divisible(cin * expand)
        if exp != cin:
            layers = [ConvNormActivation(cin, exp, kernel, stride, act="silu", bn_eps=BN_EPS),
                      ConvNormActivation(exp, cout, 1, 1, act="none", bn_eps=BN_EPS)]
        else:
            layers = [ConvNormActivation(cin, cout, kernel, stride, act="silu", bn_eps=BN_EPS)]
        self.block = Sequential(*layers)
        self.stochastic_depth = StochasticDepth(sd_prob, "row")

    # 3x3 expand -> BatchNorm -> SiLU -> 1x1 project: in training the project conv's loader applies the BatchNorm + SiLU of the
    # expand conv's raw output (suhpe_conv2d_fwd_stats_xf), so the activated 4x-expanded tensor -- the largest activation of the
    # network -- is neither written nor read; the tape holds the standard BatchNorm entry, so the default backward applies
    def _front_fwd(self, x, tape):
        if len(self.block) != 2:
            return super()._front_fwd(x, tape)
        (c0, b0), c1 = self.block[0], self.block[1][0]
        from .. import layers as L
        if not (b0.training and self.block[1][1].training and c1.takes_input_xform() and c0.fuses_stats() and not L.SYNC_BN):
            return super()._front_fwd(x, tape)
        y0, partials = c0.fwd_stats(x, tape)
        mean, rstd, rows_a = b0.train_stats(y0, partials)
        hw = y0.shape[1] * y0.shape[2]
        if tape is not None:                # what _BatchNorm.bwd expects (segments: statistics pair + samples of the first segment)
            tape.append((y0, (mean, rstd), rows_a // hw, True, None, None, hw) if rows_a else (y0, mean, rstd, True, None, None, hw))
        sc, sh = K.bn_affine_from_stats(mean, rstd, b0.weight, b0.bias)
        return y0, (sc, sh, None, b0.act, hw, rows_a)


    # Backward of the folded form: the project conv's input-gradient GEMM multiplies by SiLU'(BN(y0)) in its epilogue and reduces the expand
    # BatchNorm's backward sums there (suhpe_conv2d_dgrad_bnbwd), so that BatchNorm's backward is ONE apply pass over (y0, g) instead of a
    # reduction pass + an apply pass over the two widest tensors of the network.  SUHPE_BN_SUMS_PROJECT=0: the layer-by-layer backward.
    def _last_bwd(self, dy, tape):
        if not (BN_SUMS_FROM_PROJECT and len(self.block) == 2 and len(tape) >= 3):
            return super()._last_bwd(dy, tape)
        (conv0, bn0), (conv1, bn1) = self.block[0], self.block[1]
        top, rec = tape[-2], tape[-3]          # (y0, input transform) of the project conv; the expand BatchNorm's train-mode record
        if not (isinstance(top, tuple) and len(top) == 2 and torch.is_tensor(top[0]) and isinstance(rec, tuple) and len(rec) == 7 and rec[0] is top[0] and
                rec[3] is True and rec[4] is None and rec[5] is None and conv1.need_input_grad and conv1.bias is None):
            return super()._last_bwd(dy, tape)
        dt = bn1.bwd(dy, tape)
        y0, xf = tape.pop()
        K.conv2d_wgrad(y0, dt, conv1.weight.shape, conv1.stride, conv1.padding, conv1.groups, out=grad_of(conv1.weight), accumulate=True, xf=xf)
        sc, sh = xf[0], xf[1]
        _, mean, rstd, _, _, _, hw = tape.pop()
        c = bn0.num_features
        if isinstance(mean, tuple):                          # two batch segments: ((mean[2, c], rstd[2, c]), samples of the first)
            (m2, r2), seg = mean, rstd
            g, sums = K.conv2d_dgrad_bnbwd(dt, conv1.weight, y0.shape, y0, m2, r2, sc, sh, bn0.act, seg, grad_of(bn0.weight), grad_of(bn0.bias))
            dz = K.bn_act_bwd_apply_seg(y0, g, seg * hw, c, m2, r2, bn0.weight, bn0.bias, "none", sums)
        else:
            g, sums = K.conv2d_dgrad_bnbwd(dt, conv1.weight, y0.shape, y0, mean, rstd, sc, sh, bn0.act, 0, grad_of(bn0.weight), grad_of(bn0.bias))
            dz = K.bn_act_bwd_apply(y0, g, c, mean, rstd, bn0.weight, bn0.bias, "none", sums, y0.numel() // c)
        return ("expand_bn_done", dz)

    def _front_bwd(self, dy, tape):
        if isinstance(dy, tuple) and dy[0] == "expand_bn_done":
            return self.block[0][0].bwd(dy[1], tape)
        return super()._front_bwd(dy, tape)


class MBConv(_ResidualBlock):
    def __init__(self, expand, kernel, stride, cin, cout, sd_prob):
        super().__init__()
        self.use_res_connect = stride == 1 and cin == cout
        exp = _make_divisible(cin * expand)
        layers = []
        if exp != cin:
            layers.append(ConvNormActivation(cin, exp, 1, 1, act="silu", bn_eps=BN_EPS))
        layers.append(ConvNormActivation(exp, exp, kernel, stride, groups=exp, act="silu", bn_eps=BN_EPS))
        layers.append(SqueezeExcitation(exp, max(1, cin // 4), act="silu", scale_act="sigmoid"))
        layers.append(ConvNormActivation(exp, cout, 1, 1, act="none", bn_eps=BN_EPS))
        self.block = Sequential(*layers)
        self.stochastic_depth = StochasticDepth(sd_prob, "row")
        self.has_expand = exp != cin

    # The expand conv's BatchNorm + SiLU is applied by the depthwise kernels while they load (suhpe_dwconv_bnact_fwd /
    # _wgrad): the normalised, 4-6x expanded tensor -- the largest activation of the block -- is never written or re-read.
    # Its BatchNorm still gets exact statistics (from the expand conv's epilogue) and the standard backward (from the raw
    # output and the gradient the depthwise dgrad returns).  Train mode, inside the fused depthwise-SE section only.
    def _expand_folded(self, no_tape=False):
        """no_tape: a forward without a backward (the teacher pass, FUSE_EXPAND_NO_TAPE); with a tape the fold is on too since round 3
        (FUSE_EXPAND_BN_INTO_DW, default ON): the image-tile depthwise kernels apply BatchNorm + SiLU once per element while they fill LDS, so
        the filter gradient no longer pays a threefold recompute (that was the sliding-window kernels of round 2, where the fold lost)."""
        from .. import layers as L
        if not ((FUSE_EXPAND_BN_INTO_DW or (no_tape and FUSE_EXPAND_NO_TAPE)) and self.has_expand and self._dw_se_fused() and not L.SYNC_BN):
            return False
        (conv_e, bn_e), (conv_d, bn_d) = self.block[0], self.block[1]
        return bn_e.training and bn_d.training and conv_e.fuses_stats() and conv_d.fuses_stats()

    # depthwise conv -> BatchNorm -> SiLU -> SqueezeExcitation as ONE section around the BatchNorm (kernels.bn_act_pool /
    # bn_act_scale_fwd / se_bn_bwd_*): the activated tensor z = silu(BN(y)) is never stored; forward = squeeze pass (read y) +
    # scale pass (read y, write z*s); backward = one reduction pass + one apply pass over (y, dout) instead of scale-backward
    # (two kernels) + BatchNorm backward (two passes).  SyncBN keeps the layer-by-layer form (its statistics are all-reduced).
    def _dw_se_fused(self):
        from .. import layers as L
        conv_d, se = self.block[-3][0], self.block[-2]
        return FUSE_DW_BN_SE and not L.SYNC_BN and se._fused() and conv_d.in_channels % 4 == 0

    def _dw_se_fwd(self, x, tape, xf=None):
        (conv_d, bn_d), se = self.block[-3], self.block[-2]
        c = bn_d.num_features
        partials = None
        if bn_d.training and conv_d.fuses_stats():
            y, partials = conv_d.fwd_stats(x, tape, xf=xf)
        else:
            y = conv_d.fwd(x, tape)
        hw = y.shape[1] * y.shape[2]
        if bn_d.training:
            mean, rstd, rows_a = bn_d.train_stats(y, partials)
        else:
            (mean, rstd), rows_a = K.bn_eval_stats(bn_d.running_mean, bn_d.running_var, bn_d.eps), 0
        p = K.bn_act_pool(y, rows_a, c, mean, rstd, bn_d.weight, bn_d.bias, bn_d.act, hw)
        a1, s = K.se_mlp_fwd(p, se.fc1.weight, se.fc1.bias, se.fc2.weight, se.fc2.bias)
        if tape is not None:
            tape.append((y, mean, rstd, rows_a, bn_d.training, p, a1, s, hw))
        proj = self.block[-1]
        if bn_d.training and proj[1].training and proj[0].takes_input_xform(level=2):
            # the project conv's loader applies silu(BN(y)) * s itself: nothing of the expanded width is written here
            sc, sh = K.bn_affine_from_stats(mean, rstd, bn_d.weight, bn_d.bias)
            return y, (sc, sh, s, bn_d.act, hw, rows_a)
        return K.bn_act_scale_fwd(y, rows_a, c, mean, rstd, bn_d.weight, bn_d.bias, bn_d.act, s, hw)

    def _dw_se_bwd(self, dout, tape, expand=None):
        """expand: the block's (expand conv, BatchNorm) pair when its backward follows (it is run here, so that the depthwise
        input-gradient kernel can hand that BatchNorm its backward sums)"""
        (conv_d, bn_d), se = self.block[-3], self.block[-2]
        y, mean, rstd, rows_a, train_stats, p, a1, s, hw = tape.pop()
        c = bn_d.num_features
        sums = K.se_bn_bwd_reduce(y, dout, rows_a, c, mean, rstd, bn_d.weight, bn_d.bias, bn_d.act, hw)
        dp = K.se_mlp_bwd(sums[0], s, a1, p, se.fc1.weight, se.fc2.weight, grad_of(se.fc1.weight), grad_of(se.fc1.bias),
                          grad_of(se.fc2.weight), grad_of(se.fc2.bias))
        chan = K.se_bn_bwd_combine(sums, s, dp, rows_a // hw, hw, grad_of(bn_d.weight), grad_of(bn_d.bias), True)
        if expand is not None and self._bn_sums_from_dw(conv_d, tape) and dw_bwd_fused_ok(conv_d, tape):
            # the section's apply pass, the depthwise filter gradient and input gradient (+ the expand BatchNorm's sums) as one launch: dy is never stored
            return dw_dgrad_expand_bwd(conv_d, expand, dout.contiguous(), tape,
                                       mid=(y, s, dp, chan, mean, rstd, bn_d.weight, bn_d.bias, bn_d.act, train_stats))
        dy = K.se_bn_bwd_apply(y, dout, rows_a, c, mean, rstd, bn_d.weight, bn_d.bias, bn_d.act, s, dp, chan, train_stats, hw)
        if expand is not None and self._bn_sums_from_dw(conv_d, tape):
            return self._dw_dgrad_expand_bwd(conv_d, expand, dy, tape)
        d = conv_d.bwd(dy, tape)
        return d if expand is None else expand.bwd(d, tape)

    def _bn_sums_from_dw(self, conv_d, tape):
        return bn_sums_from_dw_ok(conv_d, tape)

    def _dw_dgrad_expand_bwd(self, conv_d, expand, dy, tape):
        return dw_dgrad_expand_bwd(conv_d, expand, dy, tape)

    def _front_fwd(self, x, tape):
        # expand -> depthwise conv -> BatchNorm + SiLU + squeeze -> excitation
        layers = list(self.block)[:-1]
        if self._expand_folded(no_tape=tape is None):
            conv_e, bn_e = layers[0]
            y0, partials = conv_e.fwd_stats(x, tape)
            mean, rstd, rows_a = bn_e.train_stats(y0, partials)
            hw = y0.shape[1] * y0.shape[2]
            if tape is not None:            # what _BatchNorm.bwd expects
                tape.append((y0, (mean, rstd), rows_a // hw, True, None, None, hw) if rows_a else (y0, mean, rstd, True, None, None, hw))
            return self._dw_se_fwd(y0, tape, xf=("bn", mean, rstd, bn_e.weight, bn_e.bias, bn_e.act, rows_a // hw))
        if self.has_expand:
            x = layers[0].fwd(x, tape)
        if self._dw_se_fused():
            return self._dw_se_fwd(x, tape)
        dw, se = layers[-2], layers[-1]
        z, pooled = conv_bn_fwd(dw[0], dw[1], x, tape, pool=True)
        return se.fwd(z, tape, pooled=pooled)

    def _front_bwd(self, dy, tape):
        if self._dw_se_fused():
            # (folded expand BatchNorm: the depthwise conv's tape entry is (raw expand output, transform); its input gradient is the
            #  gradient of the activated tensor the loaders recomputed, and the BatchNorm's backward follows as in the unfolded form)
            return self._dw_se_bwd(dy, tape, expand=self.block[0] if self.has_expand else None)
        return super()._front_bwd(dy, tape)


def regression_head(in_dim, hidden1, hidden2, num_classes):
    """The 3-layer MLP the reference bolts on every backbone (src/networks.py:264-273, 309-319, 419-428):
    Dropout(0.2) Linear BN1d ReLU6 Linear BN1d ReLU6 Linear; indices 0..7 so keys read '<name>.1.weight' ..."""
    return Sequential(Dropout(0.2), Linear(in_dim, hidden1), BatchNorm1d(hidden1, act="relu6"), Identity(),
                      Linear(hidden1, hidden2), BatchNorm1d(hidden2, act="relu6"), Identity(),
                      Linear(hidden2, num_classes))


def init_regression_head_(head):
    """Linear ~ N(0, 0.01), bias 0; BN ones/zeros (src/networks.py:275-285, 321-331, 430-440)."""
    for m in head:
        if isinstance(m, Linear):
            nn.init.normal_(m.weight, 0, 0.01)
            nn.init.zeros_(m.bias)
        elif isinstance(m, BatchNorm1d):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)


class EfficientNetV2S(Network):
    def __init__(self, num_classes=9):
        super().__init__()
        feats = [ConvNormActivation(3, V2_S[0][4], 3, 2, act="silu", bn_eps=BN_EPS)]
        feats[0][0].need_input_grad = False                     # nothing upstream of the image
        total = sum(c[-1] for c in V2_S)
        idx = 0
        for kind, expand, kernel, stride, cin, cout, nlayers in V2_S:
            stage = []
            for i in range(nlayers):
                sd = STOCHASTIC_DEPTH_PROB * float(idx) / total
                cls = FusedMBConv if kind == "fused" else MBConv
                stage.append(cls(expand, kernel, stride if i == 0 else 1, cin if i == 0 else cout, cout, sd))
                idx += 1
            feats.append(Sequential(*stage))
        feats.append(ConvNormActivation(V2_S[-1][5], LAST_CHANNEL, 1, 1, act="silu", bn_eps=BN_EPS))
        self.features = Sequential(*feats)
        self.avgpool = GlobalAvgPool()
        self.classifier = regression_head(LAST_CHANNEL, 512, 128, num_classes)
        self.init_like_reference_()
        init_regression_head_(self.classifier)

    def fwd(self, x, tape):
        if self.training:                   # every StochasticDepth mask of the pass from one launch (same values as block-by-block draws)
            from ..layers import prefetch_masks
            reqs = [("stochastic_depth", (x.shape[0],), 1.0 - blk.stochastic_depth.p) for stage in list(self.features)[1:-1] for blk in stage
                    if blk.use_res_connect and blk.stochastic_depth.training and blk.stochastic_depth.p > 0.0]
            prefetch_masks(reqs, x.device)
        x = self.features.fwd(x, tape)
        x = self.avgpool.fwd(x, tape)
        return self.classifier.fwd(x, tape)

    def bwd(self, dy, tape):
        dy = self.classifier.bwd(dy, tape)
        dy = self.avgpool.bwd(dy, tape)
        return self.features.bwd(dy, tape)


def get_EfficientNet_V2(config, model_name="S"):
    """src/networks.py:301-332 (pretrained ImageNet weights are not available offline: random init, same recipe)."""
    if model_name != "S":
        raise NotImplementedError("only EfficientNetV2-S is instantiated by the reference (src/networks.py:25)")
    return EfficientNetV2S(num_classes=config.num_classes)
