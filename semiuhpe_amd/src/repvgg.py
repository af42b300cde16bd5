"""RepVGG-B1g2 backbone with the reference's module tree (src/repvgg.py:35-76,165-199; wrapper src/networks.py:229-290).

The reference never re-parameterises during training or evaluation (`switch_to_deploy` is not called anywhere,
SURVEY.md F11), so the block is the three-branch training form
    out = ReLU( BN(conv3x3(x)) + BN(conv1x1(x)) + [BN(x) if cin == cout and stride == 1] )
with per-branch batch statistics.  Blocks are Layers with explicit forward/backward over the HIP kernels.
"""
import os

import torch.nn as nn

from .. import kernels as K
from .. import layers as L
from ..layers import BatchNorm2d, Conv2d, GlobalAvgPool, Layer, Network, Sequential, conv_bn_fwd, grad_of

# training: the three BatchNorm applies + ReLU of a block (and their backward) run as one kernel each (suhpe_repvgg_bn_fwd / _bwd)
FUSE_BLOCK_TAIL = True
# the tail kernel of a block whose successor has an identity branch also emits the BatchNorm statistics records of its output: the successor's
# identity BatchNorm then skips its statistics pass over that tensor (SUHPE_REPVGG_OUT_STATS=0: off)
OUT_STATS = os.environ.get("SUHPE_REPVGG_OUT_STATS", "1") != "0"
REMASK = os.environ.get("SUHPE_REPVGG_REMASK", "1") != "0"     # the tail's backward recomputes the ReLU mask instead of reading the block output
from .efficientnet_v2 import init_regression_head_, regression_head


class _ConvBN(Layer):
    """`conv_bn` of the reference (src/repvgg.py:28-32): children 'conv' and 'bn'"""

    def __init__(self, cin, cout, k, stride, pad, groups):
        super().__init__()
        self.conv = Conv2d(cin, cout, k, stride, pad, groups, bias=False)
        self.bn = BatchNorm2d(cout)


class RepVGGBlock(Layer):
    def __init__(self, in_channels, out_channels, stride=1, groups=1):
        super().__init__()
        self.rbr_identity = BatchNorm2d(in_channels) if out_channels == in_channels and stride == 1 else None
        self.rbr_dense = _ConvBN(in_channels, out_channels, 3, stride, 1, groups)
        self.rbr_1x1 = _ConvBN(in_channels, out_channels, 1, stride, 0, groups)
        self.rbr_dense.bn.act = "relu"           # the block's ReLU is fused into the last kernel of the sum
        self.need_input_grad = True
        self.feeds_identity = False              # set by the network: the next block normalises this block's output in its identity branch

    def fwd(self, x, tape):
        train = self.rbr_dense.bn.training
        if not train and tape is None and self.rbr_dense.conv.fuses_affine():
            # eval without a tape (teacher threshold scan, validation, eval.py): the reference's structural re-parameterisation
            # (repvgg.py:104-142) — both conv branches and the identity BatchNorm collapse into ONE 3x3 conv + bias + ReLU;
            # the equivalent filter is rebuilt per call (the weights move every training step), one small kernel
            c3, c1 = self.rbr_dense.conv, self.rbr_1x1.conv
            for c in (c3, c1):
                if not K.filter_is_physical(c.weight):
                    c.weight.data = K.to_filter_layout(c.weight.data)
            bn = lambda b: (b.weight, b.bias, b.running_mean, b.running_var)
            w_eq, ones, b_eq = K.repvgg_fuse(c3.weight, c1.weight, bn(self.rbr_dense.bn), bn(self.rbr_1x1.bn),
                                             bn(self.rbr_identity) if self.rbr_identity is not None else None, self.rbr_dense.bn.eps)
            return K.conv2d_fwd_affine(x, w_eq, ones, b_eq, "relu", None, False, c3.stride, c3.padding, c3.groups)
        if train and FUSE_BLOCK_TAIL and L.SEGMENT_LAUNCHES != "split" and not L.SYNC_BN:
            return self._fwd_fused_tail(x, tape)
        t3, p3 = self.rbr_dense.conv.fwd_stats(x, tape) if train else (self.rbr_dense.conv.fwd(x, tape), None)
        t1, p1 = self.rbr_1x1.conv.fwd_stats(x, tape) if train else (self.rbr_1x1.conv.fwd(x, tape), None)
        u = self.rbr_1x1.bn.fwd(t1, tape, partials=p1)
        if self.rbr_identity is not None:
            u = self.rbr_identity.fwd(x, tape, residual=u)
        return self.rbr_dense.bn.fwd(t3, tape, residual=u, partials=p3)          # relu(bn(t3) + u)

    def _fwd_fused_tail(self, x, tape):
        """training: both convs emit their BatchNorm statistics, then ONE kernel forms relu(BN(t3) + BN(t1) [+ BN(x)])"""
        bns = [self.rbr_dense.bn, self.rbr_1x1.bn] + ([self.rbr_identity] if self.rbr_identity is not None else [])
        t3, p3 = self.rbr_dense.conv.fwd_stats(x, tape)
        t1, p1 = self.rbr_1x1.conv.fwd_stats(x, tape)
        xs = [t3, t1] + ([x] if self.rbr_identity is not None else [])
        # the identity BatchNorm's statistics: from the records the PREVIOUS block's tail kernel wrote for its output (= x), when it did
        xp = getattr(x, "suhpe_out_partials", None) if OUT_STATS else None
        stats = [bns[0].train_stats(t3, p3), bns[1].train_stats(t1, p1)] + ([bns[2].train_stats(x, xp)] if len(bns) == 3 else [])
        means, rstds, rows_a = [s[0] for s in stats], [s[1] for s in stats], stats[0][2]
        if OUT_STATS and self.feeds_identity:
            y, yp = K.repvgg_bn_fwd(xs, means, rstds, [b.weight for b in bns], [b.bias for b in bns], rows_a, bns[0].num_features, emit_stats=True)
            y.suhpe_out_partials = yp          # (an attribute of this tensor object: consumed by the next block of the stage, in this pass)
        else:
            y = K.repvgg_bn_fwd(xs, means, rstds, [b.weight for b in bns], [b.bias for b in bns], rows_a, bns[0].num_features)
        if tape is not None:
            # (the parameter versions: REMASK recomputes the ReLU mask from the LIVE gamma / beta in the backward, which equals the forward's mask
            #  bit for bit only while no parameter was written in between -- an optimiser / EMA step (kernels.PARAM_EPOCH: raw-pointer writes), load_state_dict (_version); _bwd_fused_tail compares)
            tape.append(("fused_tail", xs, means, rstds, rows_a, y, (K.PARAM_EPOCH[0],) + tuple(t._version for b in bns for t in (b.weight, b.bias))))
        return y

    def _bwd_fused_tail(self, dout, tape):
        _, xs, means, rstds, rows_a, y, versions = tape.pop()
        bns = [self.rbr_dense.bn, self.rbr_1x1.bn] + ([self.rbr_identity] if self.rbr_identity is not None else [])
        # REMASK: the ReLU mask is recomputed from the three branch inputs and y is not read -- valid only while gamma / beta are the forward's:
        # a parameter written since (its _version moved) sends this block to the exact y-based mask
        remask = REMASK and versions == (K.PARAM_EPOCH[0],) + tuple(t._version for b in bns for t in (b.weight, b.bias))
        dxs = K.repvgg_bn_bwd(xs, means, rstds, [b.weight for b in bns], dout.contiguous(), y, rows_a, bns[0].num_features,
                              [grad_of(b.weight) for b in bns], [grad_of(b.bias) for b in bns], betas=[b.bias for b in bns] if remask else None)
        self.rbr_1x1.conv.need_input_grad = self.rbr_dense.conv.need_input_grad = self.need_input_grad
        return self._branch_dgrads(dxs[0], dxs[1], dxs[2] if len(dxs) == 3 else None, tape)

    def _branch_dgrads(self, d_t3, d_t1, d_xid, tape):
        """input gradients of the 1x1 and the 3x3 branch (+ the identity branch's), summed: each GEMM adds the sum so far in its
        epilogue (kernels.conv2d_dgrad(skip=...)) unless layers.SKIP_ADD_FUSION is off"""
        from .. import layers as L
        fuse = L.SKIP_ADD_FUSION and self.need_input_grad
        dx1 = self.rbr_1x1.conv.bwd(d_t1, tape, skip=d_xid if fuse else None)
        dx3 = self.rbr_dense.conv.bwd(d_t3, tape, skip=dx1 if fuse else None)
        if not self.need_input_grad:
            return None
        if fuse:
            return dx3
        return K.add3(dx3, dx1, d_xid) if d_xid is not None else K.add(dx3, dx1)

    def bwd(self, dout, tape):
        if isinstance(tape[-1], tuple) and len(tape[-1]) == 7 and tape[-1][0] == "fused_tail":
            return self._bwd_fused_tail(dout, tape)
        out = tape[-1][4]                                           # forward output kept for the ReLU mask
        d_t3 = self.rbr_dense.bn.bwd(dout, tape)
        g = K.act_bwd(out, dout, out.shape[-1], "relu")             # dL/du = dout * (out > 0)
        d_xid = self.rbr_identity.bwd(g, tape) if self.rbr_identity is not None else None
        d_t1 = self.rbr_1x1.bn.bwd(g, tape)
        self.rbr_1x1.conv.need_input_grad = self.rbr_dense.conv.need_input_grad = self.need_input_grad
        return self._branch_dgrads(d_t3, d_t1, d_xid, tape)


def b1g2_stage_configs():
    """(cin, cout, stride, groups): widths 64 | 128 256 512 2048, blocks 4/6/16/1, groups=2 on even layer indices 2..26
    (src/repvgg.py:190-199, 213-214, 237-239)."""
    stages = [[(3, 64, 2, 1)]]
    in_planes, idx = 64, 1
    for planes, n in ((128, 4), (256, 6), (512, 16), (2048, 1)):
        blocks = []
        for i in range(n):
            blocks.append((in_planes, planes, 2 if i == 0 else 1, 2 if (idx % 2 == 0 and idx <= 26) else 1))
            in_planes = planes
            idx += 1
        stages.append(blocks)
    return stages


class RepVggNet(Network):
    def __init__(self, backbone_name="RepVGG-B1g2", num_classes=9):
        super().__init__()
        if backbone_name != "RepVGG-B1g2":
            raise NotImplementedError("only RepVGG-B1g2 is instantiated by the reference (src/networks.py:231)")
        st = b1g2_stage_configs()
        self.layer0 = RepVGGBlock(*st[0][0])
        self.layer0.need_input_grad = False
        for i in range(1, 5):
            setattr(self, f"layer{i}", Sequential(*[RepVGGBlock(*c) for c in st[i]]))
        blocks = [b for st_ in self._stages() for b in ([st_] if isinstance(st_, RepVGGBlock) else list(st_))]
        for prev, nxt in zip(blocks, blocks[1:]):
            prev.feeds_identity = nxt.rbr_identity is not None
        self.gap = GlobalAvgPool()
        self.linear_reg = regression_head(2048, 512, 128, num_classes)
        self.init_like_reference_()
        init_regression_head_(self.linear_reg)

    def _stages(self):
        return [self.layer0, self.layer1, self.layer2, self.layer3, self.layer4]

    def fwd(self, x, tape):
        for s in self._stages():
            x = s.fwd(x, tape)
        return self.linear_reg.fwd(self.gap.fwd(x, tape), tape)

    def bwd(self, dy, tape):
        dy = self.gap.bwd(self.linear_reg.bwd(dy, tape), tape)
        for s in reversed(self._stages()):
            dy = s.bwd(dy, tape)
        return dy
