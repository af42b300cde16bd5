"""Layers with explicit forward/backward over the HIP kernels (no autograd inside the backbone).

Why not torch.autograd per op: the backbone is ~500 kernels per pass; an explicit tape (what the layer needs for
its own backward, pushed in forward order and popped in reverse) keeps the host cost per kernel at one ctypes call,
lets parameter gradients accumulate in place into a flat, pre-zeroed buffer (one RCCL all-reduce, one Adam
launch), and keeps every addition of gradients (residual fan-out) in our own kernels.  torch.autograd only sees
ONE node per network call (`NetFunction`), so reference-style code (`loss.backward()`) keeps working.

Layers are nn.Modules purely for parameter / buffer / state_dict bookkeeping: names and shapes equal the
reference's modules so released checkpoints load.
Activations are (N, H, W, C) contiguous fp32; filters are (Cout, Cin/g, R, S) stored [Cout][R][S][Cin/g].
"""
import math
import os

import torch
import torch.nn as nn

from . import kernels as K

# ------------------------------------------------------------------------------------------------ randomness hooks
# Parity tests feed host-made masks so that the HIP path and the CPU oracle see identical Bernoulli draws.
# hook(kind, shape, keep) -> float tensor of 0 / (1/keep) on the device, or None to use the device generator.
_mask_hook = None


def set_mask_hook(fn):
    global _mask_hook
    _mask_hook = fn


# The masks of a forward pass drawn ahead by ONE launch (a network that knows its sequence of draws calls prefetch_masks): draw_mask hands
# them out in order; a draw that does not match the head of the queue drops the queue (the generator position has moved on, the values
# stay valid random numbers).  Sequential and prefetched draws of the same sequence give the same values (same stream positions).
_mask_queue = None


def _drop_mask_queue():
    global _mask_queue
    _mask_queue = None


K._reseed_hooks.append(_drop_mask_queue)        # a reseeded generator must not hand out masks drawn from the old seed


def prefetch_masks(requests, device):
    """requests: [(kind, shape, keep)] in the order the forward pass will draw them"""
    global _mask_queue
    _mask_queue = None
    if _mask_hook is not None or len(requests) < 2:
        return
    import collections
    masks = K.dropout_masks([(shape, keep) for _, shape, keep in requests], device)
    _mask_queue = collections.deque(zip(requests, masks))


def draw_mask(kind, shape, keep, device):
    global _mask_queue
    if _mask_queue:
        (k0, s0, p0), m = _mask_queue[0]
        if _mask_hook is None and k0 == kind and tuple(s0) == tuple(shape) and p0 == keep:
            _mask_queue.popleft()
            return m
        _mask_queue = None
    if _mask_hook is not None:
        m = _mask_hook(kind, tuple(shape), keep)
        if m is not None:
            if tuple(m.shape) != tuple(shape):
                raise ValueError(f"mask hook returned shape {tuple(m.shape)} for a {kind} mask of shape {tuple(shape)}")
            return m.to(device=device, dtype=torch.float32).contiguous()
    return K.dropout_mask(tuple(shape), keep, device)


# Data-parallel gradient exchange in buckets: when set, called with every gradient buffer a layer is about to accumulate into
# (SSLAgent points it at distributed.BucketedAllReduce through the buffer's offset in the flat gradient).
_grad_watch = None


def set_grad_watch(fn):
    global _grad_watch
    _grad_watch = fn


def grad_of(p):
    """The accumulation buffer of a parameter (a view of the flat gradient buffer once the optimiser has
    flattened the model; allocated lazily otherwise)."""
    if p.grad is None:
        p.grad = torch.empty_strided(p.shape, p.stride(), dtype=p.dtype, device=p.device)
        K.fill_(p.grad, 0.0)
    if _grad_watch is not None:
        _grad_watch(p.grad)
    return p.grad


class Layer(nn.Module):
    """fwd(x, tape) pushes what bwd needs; bwd(dy, tape) pops it, accumulates parameter gradients, returns dx."""

    def fwd(self, x, tape):
        raise NotImplementedError

    def bwd(self, dy, tape):
        raise NotImplementedError


def _filter_param(cout, cg, r, s):
    w = torch.empty_strided((cout, cg, r, s), (r * s * cg, 1, s * cg, cg), dtype=torch.float32)
    return nn.Parameter(w)


# The backward of `x + block(x)`: the gradient that bypasses the block is added by the input-gradient GEMM of the block's FIRST conv
# (kernels.conv2d_dgrad(skip=...)) instead of a pass of its own.  The residual block parks the tensor here right before it runs
# its layers' backward; the conv it names picks it up.  SUHPE_SKIP_ADD_FUSION=0: separate add kernel.
SKIP_ADD_FUSION = os.environ.get("SUHPE_SKIP_ADD_FUSION", "1") != "0"
_PENDING_SKIP = None


def park_skip(conv, dy):
    """-> True when `conv` (the first layer of a residual block) will add dy in its input-gradient epilogue"""
    global _PENDING_SKIP
    if not (SKIP_ADD_FUSION and isinstance(conv, Conv2d) and not conv._depthwise() and conv.need_input_grad):
        return False
    assert _PENDING_SKIP is None
    _PENDING_SKIP = (conv, dy)
    return True


def take_pending_skip(conv):
    global _PENDING_SKIP
    if _PENDING_SKIP is not None and _PENDING_SKIP[0] is conv:
        skip, _PENDING_SKIP = _PENDING_SKIP[1], None
        return skip
    return None


class Conv2d(Layer):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, groups=1, bias=False):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding, self.groups = stride, padding, groups
        self.weight = _filter_param(out_channels, in_channels // groups, kernel_size, kernel_size)
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.need_input_grad = True
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.weight.shape[1] * self.kernel_size * self.kernel_size
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def fwd(self, x, tape):
        if not K.filter_is_physical(self.weight):
            self.weight.data = K.to_filter_layout(self.weight.data)
        y = K.conv2d_fwd(x, self.weight, self.bias, self.stride, self.padding, self.groups)
        if tape is not None:
            tape.append(x)
        return y

    def fwd_stats(self, x, tape, xf=None):
        """forward + BatchNorm partial statistics from the GEMM epilogue (dense, bias-free convs only) -> (y, (part, tiles))
        (with batch segments: (y, (part, tiles, records of the first segment, boundary record))).
        xf (pointwise convs): the conv's input is act(x * scale + shift) * rowscale, applied by the GEMM's loader -- x is the raw
        output of the producer whose BatchNorm + activation were never materialised (kernels.conv2d_fwd_stats)."""
        if not K.filter_is_physical(self.weight):
            self.weight.data = K.to_filter_layout(self.weight.data)
        if self._depthwise():
            out = K.dwconv_fwd_stats(x, self.weight, self.stride, self.padding, seg=_segment_of(x), xf=xf)
        else:
            out = K.conv2d_fwd_stats(x, self.weight, self.stride, self.padding, self.groups, seg=_segment_of(x), xf=xf)
        if tape is not None:
            tape.append(x if xf is None else (x, xf))
        return out[0], out[1:]

    def takes_input_xform(self, level=1):
        """can this conv apply its producer's BatchNorm + activation while loading?  (pointwise, 16-byte channel vectors)"""
        return (XF_FUSION >= level and self.kernel_size == 1 and self.stride == 1 and self.padding == 0 and self.groups == 1 and self.bias is None
                and self.in_channels % 4 == 0 and self.out_channels % 4 == 0)

    def _depthwise(self):
        return self.groups > 1 and self.groups == self.in_channels == self.out_channels

    def fuses_stats(self):
        """dense convs: statistics from the GEMM epilogue; depthwise 3x3 / pad 1 / stride 1-2: from the depthwise kernel's"""
        if self.bias is not None:
            return False
        if not self._depthwise():
            return True
        return DW_STATS and self.kernel_size == 3 and self.padding == 1 and self.stride in (1, 2) and self.in_channels % 4 == 0

    def fuses_affine(self):
        """eval mode without a tape: BatchNorm + activation (+ skip) folded into the GEMM epilogue (dense convs only)"""
        return self.bias is None and not self._depthwise()

    def bwd(self, dy, tape, skip=None):
        x, xf = tape.pop(), None
        if isinstance(x, tuple):                             # forward applied an input transform: the filter gradient recomputes it
            x, xf = x
        K.conv2d_wgrad(x, dy, self.weight.shape, self.stride, self.padding, self.groups, out=grad_of(self.weight), accumulate=True, xf=xf)
        if self.bias is not None:
            K.bias_grad(dy, self.out_channels, out=grad_of(self.bias), accumulate=True)
        if skip is None:
            skip = take_pending_skip(self)
        if not self.need_input_grad:
            return None
        return K.conv2d_dgrad(dy, self.weight, x.shape, self.stride, self.padding, self.groups, skip=skip)

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, k={self.kernel_size}, s={self.stride}, p={self.padding}, g={self.groups}"


class Linear(Layer):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            nn.init.uniform_(self.bias, -1 / math.sqrt(in_features), 1 / math.sqrt(in_features))

    def _w4(self):
        return self.weight.view(self.out_features, self.in_features, 1, 1)

    def fwd(self, x, tape, skip=None, rowscale=None, rows_per_sample=1):
        """skip / rowscale: the layer closes a residual connection -- y = skip + rowscale[sample] * (x W^T + b) from the GEMM epilogue
        (suhpe_conv2d_fwd_skip); rows_per_sample = rows of x per entry of rowscale"""
        shp = x.shape
        x4 = x.reshape(-1, 1, 1, self.in_features)
        if skip is not None and self.in_features % 4 == 0 and self.out_features % 4 == 0:
            y = K.conv2d_fwd_skip(x4, self._w4(), self.bias, skip, rowscale, rows_per_sample)
        else:
            y = K.conv2d_fwd(x4, self._w4(), self.bias, 1, 0, 1)
            if skip is not None:                             # (channel counts the vector tiles cannot take: separate pass)
                ns = y.shape[0] // rows_per_sample if rowscale is not None else y.shape[0]
                y = K.add_rowscale(skip.reshape(ns, -1), y.reshape(ns, -1), rowscale)
        if tape is not None:
            tape.append(x4)
        return y.view(*shp[:-1], self.out_features)

    def bwd(self, dy, tape, pre=None, act=None):
        """pre / act: this layer's INPUT was act(pre) (see fwd_act of the layer before): the input gradient comes back already
        multiplied by act'(pre) (GEMM epilogue), i.e. as the gradient of `pre`"""
        x4 = tape.pop()
        shp = dy.shape
        dy4 = dy.reshape(-1, 1, 1, self.out_features)
        gw = grad_of(self.weight).view(self.out_features, self.in_features, 1, 1)
        K.conv2d_wgrad(x4, dy4, gw.shape, 1, 0, 1, out=gw, accumulate=True)
        if self.bias is not None:
            K.bias_grad(dy4, self.out_features, out=grad_of(self.bias), accumulate=True)
        if pre is not None:
            dx = K.conv2d_dgrad_actgrad(dy4, self._w4(), x4.shape, pre.reshape(x4.shape), act)
        else:
            dx = K.conv2d_dgrad(dy4, self._w4(), x4.shape, 1, 0, 1)
        return dx.view(*shp[:-1], self.in_features)

    def act_epilogue_ok(self):
        return self.in_features % 4 == 0 and self.out_features % 4 == 0

    def fwd_act(self, x, tape, act):
        """-> (pre-activation, act(pre-activation)) from one GEMM launch (kernels.conv2d_fwd_act2)"""
        shp = x.shape
        x4 = x.reshape(-1, 1, 1, self.in_features)
        pre, out = K.conv2d_fwd_act2(x4, self._w4(), self.bias, act)
        if tape is not None:
            tape.append(x4)
        return pre.view(*shp[:-1], self.out_features), out.view(*shp[:-1], self.out_features)


# Batch segments.  A training step runs the student on the labeled batch and on the strong views of the unlabeled batch
# (reference src/agent.py: two self.net(...) calls); the weights are the same, so both can go through the network as ONE
# batch -- every GEMM gets 160 instead of 32 / 128 images -- provided BatchNorm still normalises each sub-batch with its own
# statistics and updates the running statistics twice, in the reference's order.  `with batch_segments(n):` declares that
# the first n samples of every activation are one segment and the rest the other; all other layers are per-sample.
# SyncBN (optional): with several ranks, train-mode BatchNorm normalises over the GLOBAL batch -- one small
# all-reduce of (mean, E[x^2]) per layer in the forward and one of (sum g, sum g*xhat) in the backward -- which reproduces
# single-process global-batch BatchNorm exactly; the default is per-rank statistics (standard data-parallel semantics).
# Set by SSLAgent from config.sync_bn; it implies two separate student passes (no batch segments).
SYNC_BN = False
# A producer's train-mode BatchNorm + activation (+ SqueezeExcitation scale) applied by the consuming 1x1 conv's loader instead of a
# kernel of its own: 0 = never, 1 = FusedMBConv blocks (BatchNorm + SiLU, the network's largest tensors: a measured win),
# 2 = MBConv blocks too (their per-sample excitation needs one more vector load per row and slab in the GEMM loader, which costs
# the GEMMs more than the removed pass saves on MI355X -- it remains as the memory-saving mode: the excited tensor is never stored)
XF_FUSION = int(os.environ.get("SUHPE_XF_FUSION", "1"))
DW_STATS = os.environ.get("SUHPE_DW_STATS", "1") != "0"      # depthwise convs emit their BatchNorm's statistics records themselves

_SEGMENT = None
SEGMENT_LAUNCHES = os.environ.get("SUHPE_SEGMENT_LAUNCHES", "fused")     # "fused": one launch per kernel for both segments; "split"


class batch_segments:
    def __init__(self, first):
        self.first = int(first) if first else None

    def __enter__(self):
        global _SEGMENT
        self.prev, _SEGMENT = _SEGMENT, self.first

    def __exit__(self, *exc):
        global _SEGMENT
        _SEGMENT = self.prev


def _segment_of(x):
    return _SEGMENT if (_SEGMENT is not None and 0 < _SEGMENT < x.shape[0]) else None


class _BatchNorm(Layer):
    """BatchNorm fused with the activation that follows it (and an optional pre-activation residual)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, act="none"):
        super().__init__()
        self.num_features, self.eps, self.momentum, self.act = num_features, eps, momentum, act
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def fwd(self, x, tape, residual=None, partials=None, post=False, rowscale=None, pool=False):
        """residual, post=False: act(BN(x) + residual).  post=True: residual + rowscale[n] * act(BN(x)), the skip
        connection of a residual block with StochasticDepth('row') folded in (rowscale None = 1)."""
        c = self.num_features
        if self.training and SYNC_BN:
            y = self._fwd_sync(x, tape, residual, partials, post, rowscale)
            return (y, K.avgpool_fwd(y)) if pool else y
        seg = _segment_of(x) if self.training else None
        if pool:                                            # -> (y, per-sample channel means of y), see _fwd_pool
            return self._fwd_pool(x, tape, partials, seg)
        if seg is not None:
            return self._fwd_segments(x, tape, residual, partials, post, rowscale, seg)
        if self.training and partials is not None:          # statistics already accumulated by the producing conv's epilogue
            mean, rstd = K.bn_stats_from_partials(partials[0], partials[1], x.numel() // c, c, self.eps, self.momentum,
                                                  self.running_mean, self.running_var, self.num_batches_tracked)
        elif self.training:
            mean, rstd = K.bn_stats(x, c, self.eps, self.momentum, self.running_mean, self.running_var, self.num_batches_tracked)
        else:
            mean, rstd = K.bn_eval_stats(self.running_mean, self.running_var, self.eps)
        hw = x.numel() // (c * x.shape[0])
        y = K.bn_act_fwd(x, c, mean, rstd, self.weight, self.bias, self.act, residual, post, rowscale, hw)
        if tape is not None:
            keep_y = y if (residual is not None and not post and self.act == "relu") else None
            tape.append((x, mean, rstd, self.training, keep_y, rowscale if post else None, hw))
        return y

    def _fwd_sync(self, x, tape, residual, partials, post, rowscale):
        """training forward with statistics of the global batch (all ranks hold equally many rows)"""
        from .src import distributed as D
        c = self.num_features
        rows = x.numel() // c
        hw = x.numel() // (c * x.shape[0])
        if partials is not None:
            mean, rstd = K.bn_stats_from_partials(partials[0], partials[1], rows, c, self.eps, self.momentum)
        else:
            mean, rstd = K.bn_stats(x, c, self.eps, self.momentum)
        world = D.world()[1]
        buf = K.syncbn_moments(mean, rstd, self.eps)                             # (E[x], E[x^2]) of this rank's rows, [2, c]
        D.allreduce_avg_(buf)
        n = rows * world
        # global-batch mean / rstd + the running statistics' momentum update: one HIP kernel (no torch arithmetic on the path)
        mean, rstd = K.syncbn_finish(buf, self.eps, self.momentum, n, self.running_mean, self.running_var, self.num_batches_tracked)
        y = K.bn_act_fwd(x, c, mean, rstd, self.weight, self.bias, self.act, residual, post, rowscale, hw)
        if tape is not None:
            keep_y = y if (residual is not None and not post and self.act == "relu") else None
            tape.append((x, mean, rstd, ("sync", n), keep_y, rowscale if post else None, hw))
        return y

    def train_stats(self, x, partials=None):
        """train-mode statistics of x (running statistics updated) -> (mean, rstd, rows of the first segment or 0); the
        statistics are [2, c] tensors when the batch has two segments"""
        c, n = self.num_features, x.shape[0]
        hw = x.numel() // (c * n)
        seg = _segment_of(x)
        if seg is not None:
            if partials is not None:
                part, tiles, (first_records, boundary) = partials
                mean, rstd = K.bn_stats_from_partials_seg(part, tiles, first_records, boundary, n * hw, seg * hw, c, self.eps, self.momentum,
                                                          self.running_mean, self.running_var, self.num_batches_tracked)
            else:
                mean, rstd = K.bn_stats_seg(x, seg * hw, c, self.eps, self.momentum, self.running_mean, self.running_var, self.num_batches_tracked)
            return mean, rstd, seg * hw
        if partials is not None:
            mean, rstd = K.bn_stats_from_partials(partials[0], partials[1], x.numel() // c, c, self.eps, self.momentum, self.running_mean,
                                                  self.running_var, self.num_batches_tracked)
        else:
            mean, rstd = K.bn_stats(x, c, self.eps, self.momentum, self.running_mean, self.running_var, self.num_batches_tracked)
        return mean, rstd, 0

    def _fwd_pool(self, x, tape, partials, seg):
        """act(BN(x)) together with the global average pool of it (the squeeze of the SqueezeExcitation block that follows):
        one pass, the pooled means are fixed-order sums inside one workgroup per sample."""
        c, n = self.num_features, x.shape[0]
        hw = x.numel() // (c * n)
        if seg is not None and SEGMENT_LAUNCHES == "split":
            y = self._fwd_segments(x, tape, None, partials, False, None, seg)
            return y, K.avgpool_fwd(y)
        if seg is not None:
            if partials is not None:
                part, tiles, (first_records, boundary) = partials
                mean, rstd = K.bn_stats_from_partials_seg(part, tiles, first_records, boundary, n * hw, seg * hw, c, self.eps, self.momentum,
                                                          self.running_mean, self.running_var, self.num_batches_tracked)
            else:
                mean, rstd = K.bn_stats_seg(x, seg * hw, c, self.eps, self.momentum, self.running_mean, self.running_var, self.num_batches_tracked)
            y, p = K.bn_act_pool_fwd(x, seg * hw, c, mean, rstd, self.weight, self.bias, self.act, hw)
            if tape is not None:
                tape.append((x, (mean, rstd), seg, True, None, None, hw))
            return y, p
        if self.training and partials is not None:
            mean, rstd = K.bn_stats_from_partials(partials[0], partials[1], x.numel() // c, c, self.eps, self.momentum,
                                                  self.running_mean, self.running_var, self.num_batches_tracked)
        elif self.training:
            mean, rstd = K.bn_stats(x, c, self.eps, self.momentum, self.running_mean, self.running_var, self.num_batches_tracked)
        else:
            mean, rstd = K.bn_eval_stats(self.running_mean, self.running_var, self.eps)
        y, p = K.bn_act_pool_fwd(x, 0, c, mean, rstd, self.weight, self.bias, self.act, hw)
        if tape is not None:
            tape.append((x, mean, rstd, self.training, None, None, hw))
        return y, p

    def _fwd_segments(self, x, tape, residual, partials, post, rowscale, seg):
        """training forward with per-segment batch statistics: samples [0, seg) and [seg, n) are normalised separately and
        the running statistics take two momentum updates, first segment first.  One launch per kernel covers both segments
        (statistics are [2, c]); SEGMENT_LAUNCHES = "split" runs the one-segment kernels on the two halves instead."""
        c, n = self.num_features, x.shape[0]
        hw = x.numel() // (c * n)
        if SEGMENT_LAUNCHES != "split":
            if partials is not None:
                part, tiles, (first_records, boundary) = partials
                mean, rstd = K.bn_stats_from_partials_seg(part, tiles, first_records, boundary, n * hw, seg * hw, c, self.eps, self.momentum,
                                                          self.running_mean, self.running_var, self.num_batches_tracked)
            else:
                mean, rstd = K.bn_stats_seg(x, seg * hw, c, self.eps, self.momentum, self.running_mean, self.running_var, self.num_batches_tracked)
            y = K.bn_act_fwd_seg(x, seg * hw, c, mean, rstd, self.weight, self.bias, self.act, residual, post, rowscale, hw)
            if tape is not None:
                keep_y = y if (residual is not None and not post and self.act == "relu") else None
                tape.append((x, (mean, rstd), seg, True, keep_y, rowscale if post else None, hw))
            return y
        stats = []
        for lo, hi in ((0, seg), (seg, n)):
            rows = (hi - lo) * hw
            if partials is not None:
                part, tiles, (first_records, boundary) = partials
                if lo == 0:
                    first, count, extra = 0, first_records, None
                else:
                    first, count, extra = first_records, tiles - first_records, boundary
                stats.append(K.bn_stats_from_partials_range(part, first, count, extra, rows, c, self.eps, self.momentum, self.running_mean,
                                                            self.running_var, self.num_batches_tracked))
            else:
                stats.append(K.bn_stats(x[lo:hi], c, self.eps, self.momentum, self.running_mean, self.running_var, self.num_batches_tracked))
        y = torch.empty_like(x)
        for (lo, hi), (mean, rstd) in zip(((0, seg), (seg, n)), stats):
            K.bn_act_fwd(x[lo:hi], c, mean, rstd, self.weight, self.bias, self.act, None if residual is None else residual[lo:hi], post,
                         None if rowscale is None else rowscale[lo:hi], hw, out=y[lo:hi])
        if tape is not None:
            keep_y = y if (residual is not None and not post and self.act == "relu") else None
            tape.append((x, stats, seg, True, keep_y, rowscale if post else None, hw))
        return y

    def bwd(self, dy, tape):
        x, mean, rstd, train_stats, y_out, rowscale, hw = tape.pop()
        if isinstance(train_stats, tuple) and train_stats[0] == "sync":      # global-batch statistics: all-reduce the gradient sums
            from .src import distributed as D
            c = self.num_features
            sums = K.bn_act_bwd_sums(x, dy, c, mean, rstd, self.weight, self.bias, self.act, grad_of(self.weight), grad_of(self.bias), True, y_out,
                                     rowscale, hw)
            D.allreduce_sum_(sums)
            return K.bn_act_bwd_apply(x, dy, c, mean, rstd, self.weight, self.bias, self.act, sums, train_stats[1], y_out, rowscale, hw)
        if isinstance(mean, tuple):                          # two segments, one launch: (x, (mean[2,c], rstd[2,c]), seg, ...)
            (m2, r2), seg = mean, rstd
            return K.bn_act_bwd_seg(x, dy, seg * hw, self.num_features, m2, r2, self.weight, self.bias, self.act, grad_of(self.weight),
                                    grad_of(self.bias), True, y_out, rowscale, hw)
        if isinstance(mean, list):                           # per-segment launches: (x, [(mean, rstd)] * 2, seg, ...)
            stats, seg, n = mean, rstd, x.shape[0]
            dx = torch.empty_like(x)
            for (lo, hi), (m, r) in zip(((0, seg), (seg, n)), stats):
                K.bn_act_bwd(x[lo:hi], dy[lo:hi], self.num_features, m, r, self.weight, self.bias, self.act, True, grad_of(self.weight),
                             grad_of(self.bias), True, None if y_out is None else y_out[lo:hi],
                             None if rowscale is None else rowscale[lo:hi], hw, out=dx[lo:hi])
            return dx
        return K.bn_act_bwd(x, dy, self.num_features, mean, rstd, self.weight, self.bias, self.act, train_stats,
                            grad_of(self.weight), grad_of(self.bias), True, y_out, rowscale, hw)

    def extra_repr(self):
        return f"{self.num_features}, eps={self.eps}, momentum={self.momentum}, act={self.act}"


class BatchNorm2d(_BatchNorm):
    pass


class BatchNorm1d(_BatchNorm):
    pass


class LayerNorm(Layer):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))

    def fwd(self, x, tape):
        y, mean, rstd = K.layernorm_fwd(x, self.dim, self.eps, self.weight, self.bias)
        if tape is not None:
            tape.append((x, mean, rstd))
        return y

    def bwd(self, dy, tape, skip=None):
        x, mean, rstd = tape.pop()
        return K.layernorm_bwd(x, dy, self.dim, mean, rstd, self.weight, grad_of(self.weight), grad_of(self.bias), True, skip=skip)


class Activation(Layer):
    """Stand-alone activation (kept as a module so Sequential indices match the reference's)."""

    def __init__(self, act):
        super().__init__()
        self.act = act

    def fwd(self, x, tape):
        y = K.act_fwd(x, x.shape[-1], self.act)
        if tape is not None:
            tape.append(x)
        return y

    def bwd(self, dy, tape):
        x = tape.pop()
        return K.act_bwd(x, dy, x.shape[-1], self.act)


class Identity(Layer):
    """Placeholder for an activation that has been fused into the preceding BatchNorm."""

    def fwd(self, x, tape):
        return x

    def bwd(self, dy, tape):
        return dy


class Dropout(Layer):
    def __init__(self, p=0.5):
        super().__init__()
        self.p = p

    def fwd(self, x, tape):
        if not self.training or self.p == 0.0:
            if tape is not None:
                tape.append(None)
            return x
        mask = draw_mask("dropout", x.shape, 1.0 - self.p, x.device)
        if tape is not None:
            tape.append(mask)
        return K.mul(x, mask)

    def bwd(self, dy, tape):
        mask = tape.pop()
        return dy if mask is None else K.mul(dy, mask)


class StochasticDepth(Layer):
    """torchvision.ops.StochasticDepth(p, 'row') fused with the residual add: y = skip + branch * keep_mask[n]/keep."""

    def __init__(self, p, mode="row"):
        super().__init__()
        assert mode == "row"
        self.p = p

    def fwd2(self, branch, skip, tape):
        rowscale = None
        if self.training and self.p > 0.0:
            rowscale = draw_mask("stochastic_depth", (branch.shape[0],), 1.0 - self.p, branch.device)
        if tape is not None:
            tape.append(rowscale)
        return K.add_rowscale(skip, branch, rowscale)

    def bwd2(self, dy, tape):
        """-> (d_branch, d_skip)"""
        rowscale = tape.pop()
        return (dy if rowscale is None else K.add_rowscale(None, dy, rowscale)), dy


class MaxPool2d(Layer):
    def __init__(self, kernel_size, stride, padding):
        super().__init__()
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding

    def fwd(self, x, tape):
        y, idx = K.maxpool_fwd(x, self.kernel_size, self.stride, self.padding)
        if tape is not None:
            tape.append((idx, x.shape))
        return y

    def bwd(self, dy, tape):
        idx, shape = tape.pop()
        return K.maxpool_bwd(dy, idx, shape, self.kernel_size, self.stride, self.padding)


class GlobalAvgPool(Layer):
    def fwd(self, x, tape):
        if tape is not None:
            tape.append(x.shape)
        return K.avgpool_fwd(x)

    def bwd(self, dy, tape):
        return K.avgpool_bwd(dy, tape.pop())


class SqueezeExcitation(Layer):
    """torchvision.ops.SqueezeExcitation (avgpool -> fc1 -> SiLU -> fc2 -> Sigmoid -> scale) as one layer so that the
    two gradient paths into the input (scale and squeeze) are summed inside our own kernel."""

    def __init__(self, input_channels, squeeze_channels, act="silu", scale_act="sigmoid"):
        super().__init__()
        self.fc1 = Conv2d(input_channels, squeeze_channels, 1, bias=True)
        self.fc2 = Conv2d(squeeze_channels, input_channels, 1, bias=True)
        self.act, self.scale_act = act, scale_act
        self.c, self.csq = input_channels, squeeze_channels

    def _fused(self):
        return (self.act == "silu" and self.scale_act == "sigmoid" and self.c % 4 == 0 and self.c <= 2048 and self.csq % 4 == 0
                and self.csq <= 64)

    def fwd(self, z, tape, pooled=None):
        """pooled: the global average pool of z when the producer already has it (BatchNorm's apply kernel, conv_bn_fwd(pool=True))"""
        n = z.shape[0]
        if self._fused():                       # pool -> one micro-kernel for the whole MLP -> scale
            p = pooled if pooled is not None else K.avgpool_fwd(z)
            a1, s = K.se_mlp_fwd(p, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)
            if tape is not None:
                tape.append((z, p, a1, None, None, s))
            return K.se_scale_fwd(z, s)
        p = (pooled if pooled is not None else K.avgpool_fwd(z)).view(n, 1, 1, self.c)
        a1 = K.conv2d_fwd(p, self.fc1.weight, self.fc1.bias, 1, 0, 1)
        h = K.act_fwd(a1, self.csq, self.act)
        a2 = K.conv2d_fwd(h, self.fc2.weight, self.fc2.bias, 1, 0, 1)
        s = K.act_fwd(a2, self.c, self.scale_act).view(n, self.c)
        if tape is not None:
            tape.append((z, p, a1, h, a2, s))
        return K.se_scale_fwd(z, s)

    def bwd(self, dy, tape):
        z, p, a1, h, a2, s = tape.pop()
        n = z.shape[0]
        if h is None:
            ds = K.se_scale_bwd_s(dy, z)
            dp = K.se_mlp_bwd(ds, s, a1, p, self.fc1.weight, self.fc2.weight, grad_of(self.fc1.weight), grad_of(self.fc1.bias),
                              grad_of(self.fc2.weight), grad_of(self.fc2.bias))
            return K.se_scale_bwd_x(dy, s, dp)
        ds = K.se_scale_bwd_s(dy, z).view(n, 1, 1, self.c)
        da2 = K.act_bwd(a2, ds, self.c, self.scale_act)
        K.conv2d_wgrad(h, da2, self.fc2.weight.shape, 1, 0, 1, out=grad_of(self.fc2.weight), accumulate=True)
        K.bias_grad(da2, self.c, out=grad_of(self.fc2.bias), accumulate=True)
        dh = K.conv2d_dgrad(da2, self.fc2.weight, h.shape, 1, 0, 1)
        da1 = K.act_bwd(a1, dh, self.csq, self.act)
        K.conv2d_wgrad(p, da1, self.fc1.weight.shape, 1, 0, 1, out=grad_of(self.fc1.weight), accumulate=True)
        K.bias_grad(da1, self.csq, out=grad_of(self.fc1.bias), accumulate=True)
        dp = K.conv2d_dgrad(da1, self.fc1.weight, p.shape, 1, 0, 1).view(n, self.c)
        return K.se_scale_bwd_x(dy, s, dp)


class Sequential(Layer):
    """Sequential container with numeric child names (state_dict keys '0.weight', ...)."""

    def __init__(self, *mods):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def __iter__(self):
        return iter(self._modules.values())

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, i):
        return list(self._modules.values())[i]

    def fwd(self, x, tape):
        for m in self._modules.values():
            x = m.fwd(x, tape)
        return x

    def bwd(self, dy, tape):
        for m in reversed(list(self._modules.values())):
            dy = m.bwd(dy, tape)
        return dy


def conv_bn_fwd(conv, bn, x, tape, residual=None, residual_after_act=False, rowscale=None, pool=False):
    """conv -> BatchNorm(+act): in training the statistics come out of the conv's epilogue when the conv is dense; in
    eval() without a tape (teacher pass, eval.py) BatchNorm, activation and the skip add are folded into that epilogue."""
    if pool:                                                # depthwise conv -> BatchNorm + act + squeeze: (y, pooled)
        if bn.training and conv.fuses_stats():
            y, partials = conv.fwd_stats(x, tape)
            return bn.fwd(y, tape, partials=partials, pool=True)
        return bn.fwd(conv.fwd(x, tape), tape, pool=True)
    if not bn.training and tape is None and conv.fuses_affine() and rowscale is None:
        if not K.filter_is_physical(conv.weight):
            conv.weight.data = K.to_filter_layout(conv.weight.data)
        scale, shift = K.bn_eval_affine(bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps)
        return K.conv2d_fwd_affine(x, conv.weight, scale, shift, bn.act, residual, residual_after_act, conv.stride, conv.padding, conv.groups)
    if bn.training and conv.fuses_stats():
        y, partials = conv.fwd_stats(x, tape)
        return bn.fwd(y, tape, residual=residual, partials=partials, post=residual_after_act, rowscale=rowscale)
    return bn.fwd(conv.fwd(x, tape), tape, residual=residual, post=residual_after_act, rowscale=rowscale)


class ConvNormActivation(Sequential):
    """torchvision.ops.Conv2dNormActivation: children '0' (conv, no bias) and '1' (BatchNorm); the activation that
    torchvision keeps as child '2' has no parameters and is fused into the BatchNorm kernel."""

    def __init__(self, cin, cout, kernel_size=3, stride=1, groups=1, act="silu", bn_eps=1e-5, bn_momentum=0.1):
        super().__init__(Conv2d(cin, cout, kernel_size, stride, (kernel_size - 1) // 2, groups, bias=False),
                         BatchNorm2d(cout, bn_eps, bn_momentum, act))

    def fwd(self, x, tape):
        return conv_bn_fwd(self[0], self[1], x, tape)


# ------------------------------------------------------------------------------------------------ autograd bridge
class NetFunction(torch.autograd.Function):
    """One autograd node for a whole backbone call: forward runs the explicit layer chain recording a tape, backward
    replays it in reverse; parameter gradients are accumulated in place (grad_of), the node itself returns none."""

    @staticmethod
    def forward(ctx, net, x, anchor):
        tape = []
        ctx.net, ctx.tape = net, tape
        return net.fwd(x, tape)

    @staticmethod
    def backward(ctx, dy):
        net, tape = ctx.net, ctx.tape
        ctx.tape = None
        with K.deferred_wgrads():               # the split-K slabs of every layer are reduced by one launch at the end
            net.bwd(dy.contiguous(), tape)
        assert len(tape) == 0, "backward did not consume the whole tape"
        return None, None, None


class Network(Layer):
    """Base of the backbones: forward(img NCHW) like the reference's nn.Modules; routes through NetFunction when
    gradients are needed and straight through the kernels otherwise (teacher, validation)."""

    def forward(self, img, nhwc=False):
        K._chk(img, "img")
        x = img if nhwc else K.nchw_to_nhwc(img)
        anchor = next(self.parameters())
        if torch.is_grad_enabled() and anchor.requires_grad:
            return NetFunction.apply(self, x, anchor)
        return self.fwd(x, None)

    def init_like_reference_(self):
        """conv: kaiming_normal(fan_out); BN: ones/zeros — the init torchvision / RepVGG apply before pretrained
        weights overwrite them (pretrained blobs are not available offline)."""
        for m in self.modules():
            if isinstance(m, Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, _BatchNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
