"""Batch segments: one training pass over cat[labeled; strong] with per-segment BatchNorm statistics must equal the
reference's two separate student passes (src/agent.py: self.net(img) ... self.net(img_strong)) -- outputs, parameter
gradients, BatchNorm running statistics (two momentum updates in the reference's order) -- on all three backbones, and
the agent's merged step must equal its two-pass step.

Tolerances: outputs 2e-4 relative, gradients 2e-3 relative to each tensor's max (the two paths tile and sum in different
orders; everything is fp32), running statistics 1e-5 relative."""
import copy
import os
from types import SimpleNamespace

import pytest
import torch

from parity_utils import grad_mismatches, rel_err

pytestmark = pytest.mark.gpu


class SplitMasks:
    """Records the masks of the two separate passes (own generator) and serves the merged pass their concatenation."""

    def __init__(self, seed):
        self.gen = torch.Generator().manual_seed(seed)
        self.passes = [[], []]
        self.which = 0
        self.cursor = 0

    def record(self, kind, shape, keep):
        m = (torch.rand(shape, generator=self.gen) < keep).float() / keep
        self.passes[self.which].append(m)
        return m

    def merged(self, kind, shape, keep):
        a, b = self.passes[0][self.cursor], self.passes[1][self.cursor]
        self.cursor += 1
        m = torch.cat([a, b])
        assert tuple(m.shape) == tuple(shape)
        return m


def build(name):
    cfg = SimpleNamespace(network=name, num_classes=9)
    from semiuhpe_amd.src.networks import get_network
    torch.manual_seed(0)
    net = get_network(cfg)
    g = torch.Generator().manual_seed(1)
    for m in net.modules():
        if hasattr(m, "running_mean"):
            m.running_mean.copy_((torch.randn(m.num_features, generator=g) * 0.1).cuda())
            m.running_var.copy_((torch.rand(m.num_features, generator=g) * 0.5 + 0.75).cuda())
    return net


@pytest.mark.parametrize("name,size,na,nb", [("effinetv2", 96, 3, 5), ("repvgg", 64, 2, 7), ("tinyvit", 224, 2, 3),
                                             ("effinetv2", 64, 4, 4), ("resnet50", 64, 3, 4)])
def test_merged_pass_equals_two_passes(name, size, na, nb):
    from semiuhpe_amd import layers as L
    two = build(name).train()
    one = copy.deepcopy(two).train()
    g = torch.Generator().manual_seed(5)
    xa = torch.randn(na, 3, size, size, generator=g).cuda()
    xb = (torch.randn(nb, 3, size, size, generator=g) * 1.5 + 0.3).cuda()        # a different distribution per segment
    dya, dyb = torch.randn(na, 9, generator=g).cuda(), torch.randn(nb, 9, generator=g).cuda()
    masks = SplitMasks(9)
    L.set_mask_hook(masks.record)
    try:
        masks.which = 0
        ya = two(xa)
        masks.which = 1
        yb = two(xb)
        torch.autograd.backward([ya, yb], [dya, dyb])
        L.set_mask_hook(masks.merged)
        with L.batch_segments(na):
            y = one(torch.cat([xa, xb]))
        y.backward(torch.cat([dya, dyb]))
    finally:
        L.set_mask_hook(None)
    assert masks.cursor == len(masks.passes[0]) == len(masks.passes[1])
    assert rel_err(y[:na], ya) < 2e-4 and rel_err(y[na:], yb) < 2e-4
    ref = {n: p.grad.detach().cpu() for n, p in two.named_parameters()}
    got = {n: p.grad for n, p in one.named_parameters()}
    assert not grad_mismatches(ref, got, 2e-3)
    for (n, b2), (_, b1) in zip(two.named_buffers(), one.named_buffers()):
        if b2.dtype == torch.float32:
            assert rel_err(b1, b2) < 1e-5, n
        elif n.endswith("num_batches_tracked"):
            assert int(b1) == int(b2) == 2, n                                   # two updates


def test_segmented_conv_statistics_match_per_segment_sums():
    """The conv epilogue's per-segment records, finished by the range finaliser, against BatchNorm statistics of the two halves
    computed separately -- boundary inside a row tile, on a tile edge, and a first segment smaller than one tile."""
    from semiuhpe_amd import kernels as K
    g = torch.Generator().manual_seed(3)
    for (n, h, cin, cout, k, seg) in [(6, 14, 32, 48, 1, 2), (5, 9, 16, 24, 3, 1), (8, 16, 64, 128, 1, 4), (3, 7, 24, 40, 3, 2)]:
        x = (torch.randn(n, h, h, cin, generator=g) + 0.5).cuda()
        w = K.to_filter_layout((torch.randn(cout, cin, k, k, generator=g) * 0.2).cuda())
        y, part, tiles, (first_records, boundary) = K.conv2d_fwd_stats(x, w, 1, k // 2, 1, seg=seg)
        assert rel_err(y, K.conv2d_fwd(x, w, None, 1, k // 2, 1)) == 0.0
        hw = y.shape[1] * y.shape[2]
        for lo, hi, first, count, extra in ((0, seg, 0, first_records, None), (seg, n, first_records, tiles - first_records, boundary)):
            mean, rstd = K.bn_stats_from_partials_range(part, first, count, extra, (hi - lo) * hw, cout, 1e-5, 0.1)
            ys = y[lo:hi].reshape(-1, cout).double()
            assert rel_err(mean, ys.mean(0)) < 1e-5
            assert rel_err(rstd, (ys.var(0, unbiased=False) + 1e-5).rsqrt()) < 1e-5


@pytest.mark.parametrize("network", ["effinetv2"])
def test_agent_merged_step_equals_two_pass_step(network, tmp_path, monkeypatch):
    """SSLAgent.train_func with merge_student_passes on / off: same losses, same updated weights (dropout and stochastic depth
    fed the same per-sample masks)."""
    from semiuhpe_amd import layers as L
    from semiuhpe_amd.src.agent import SSLAgent
    from semiuhpe_amd.synthetic import synthetic_batches
    monkeypatch.delenv("SUHPE_MERGE", raising=False)

    def cfg(merge):
        return SimpleNamespace(
            lr=1e-4, log_dir=str(tmp_path), model_dir=str(tmp_path), distribution="matrixFisher", is_full_range=False,
            rotate_aug=False, train_labeled="300WLP", train_unlabeled="300WLP", cutout_aug=False, cutmix_aug=False, conf_thres=1e9,
            type_unsuper="ce", SSL_lambda=1.0, is_ema=True, ema_decay=0.999, eman=False, stage1_iteration=0, network=network,
            num_classes=9, exp_name="t", date="t", left_ratio=0.95, save_feat=False, dynamic_thres=False, autotune=False,
            merge_student_passes=merge, overlap_teacher=merge)

    torch.manual_seed(11)
    a2 = SSLAgent(cfg(False))
    a1 = SSLAgent(cfg(True))
    a1.net.load_state_dict(a2.net.state_dict()); a1.ema_net.load_state_dict(a2.ema_net.state_dict())
    assert a1.merge_student_passes and not a2.merge_student_passes
    data, ulb = synthetic_batches(3, 6, size=96, seed=2, rotate_aug=False)
    to = lambda d: {k: v.cuda().contiguous() for k, v in d.items()}
    data, ulb = to(data), to(ulb)
    masks = SplitMasks(4)
    teacher, teacher_cursor = [], [0]

    def record(kind, shape, keep):                     # two-pass agent: labeled pass, teacher pass (train mode), strong pass
        if shape[0] == 6 and len(teacher) < len(masks.passes[0]):
            teacher.append((torch.rand(shape, generator=masks.gen) < keep).float() / keep)
            return teacher[-1]
        masks.which = 0 if shape[0] == 3 else 1
        return masks.record(kind, shape, keep)

    def replay(kind, shape, keep):                     # merged agent: student pass over 3 + 6 samples, teacher pass
        if shape[0] == 9:
            return masks.merged(kind, shape, keep)
        teacher_cursor[0] += 1
        return teacher[teacher_cursor[0] - 1]

    L.set_mask_hook(record)
    try:
        fd2, fu2, out2 = a2.train_func(data, ulb)
        L.set_mask_hook(replay)
        fd1, fu1, out1 = a1.train_func(data, ulb)
    finally:
        L.set_mask_hook(None)
    assert rel_err(out1["loss_all"], out2["loss_all"]) < 2e-4
    assert rel_err(fu1["unsuper_loss"], fu2["unsuper_loss"]) < 2e-4
    assert rel_err(fd1["pred"], fd2["pred"]) < 2e-4
    ref = {n: p.grad.detach().cpu() for n, p in a2.net.named_parameters()}
    assert not grad_mismatches(ref, {n: p.grad for n, p in a1.net.named_parameters()}, 2e-3)
    # Adam's first step is lr * g / (|g| + 1e-8): compare the updated weights where the gradient is neither rounding noise
    # nor down at Adam's eps (there the step is a steep function of g)
    for (n, p2), (_, p1) in zip(a2.net.named_parameters(), a1.net.named_parameters()):
        g2 = p2.grad
        big = (g2.abs() > 0.01 * g2.abs().max()) & (g2.abs() > 1e-5)
        if bool(big.any()):
            assert float((p1.data - p2.data)[big].abs().max()) < 2e-5, n


@pytest.mark.parametrize("rows_hw,c,n,seg,act,form", [
    (49, 1536, 6, 2, "silu", "plain"), (196, 96, 5, 1, "none", "skip"), (25, 24, 7, 3, "relu", "relu_res"), (9, 6, 4, 3, "relu6", "plain"),
    (784, 192, 3, 1, "gelu", "plain"), (1, 512, 8, 4, "relu6", "plain")])
def test_two_segment_batchnorm_kernels_equal_per_segment_calls(rows_hw, c, n, seg, act, form):
    """suhpe_bn_stats_seg / _act_fwd_seg / _act_bwd_seg (one launch for both segments) against the one-segment entry points on
    the two halves: statistics, running statistics (two ordered updates), outputs, dx, dgamma, dbeta."""
    from semiuhpe_amd import kernels as K
    g = torch.Generator().manual_seed(rows_hw + c)
    x = torch.randn(n, rows_hw, c, generator=g).cuda()
    x[seg:] = x[seg:] * 2.0 + 1.0
    dy = torch.randn(n, rows_hw, c, generator=g).cuda()
    gamma, beta = (torch.rand(c, generator=g) + 0.5).cuda(), torch.randn(c, generator=g).cuda()
    res = torch.randn(n, rows_hw, c, generator=g).cuda() if form != "plain" else None
    rowscale = ((torch.rand(n, generator=g) < 0.7).float() / 0.7).cuda() if form == "skip" else None
    post = form == "skip"
    rm1, rv1, nb1 = torch.zeros(c).cuda(), torch.ones(c).cuda(), torch.zeros(1, dtype=torch.long).cuda()
    rm2, rv2, nb2 = rm1.clone(), rv1.clone(), nb1.clone()
    rows_a = seg * rows_hw
    mean2, rstd2 = K.bn_stats_seg(x, rows_a, c, 1e-3, 0.1, rm1, rv1, nb1)
    y = K.bn_act_fwd_seg(x, rows_a, c, mean2, rstd2, gamma, beta, act, res, post, rowscale, rows_hw)
    y_out = y if form == "relu_res" else None
    dg1, db1 = torch.zeros(c).cuda(), torch.zeros(c).cuda()
    dx = K.bn_act_bwd_seg(x, dy, rows_a, c, mean2, rstd2, gamma, beta, act, dg1, db1, True, y_out, rowscale if post else None, rows_hw)
    dg2, db2 = torch.zeros(c).cuda(), torch.zeros(c).cuda()
    for s, (lo, hi) in enumerate(((0, seg), (seg, n))):
        m, r = K.bn_stats(x[lo:hi], c, 1e-3, 0.1, rm2, rv2, nb2)
        assert rel_err(mean2[s], m) < 5e-6 and rel_err(rstd2[s], r) < 5e-6
        ys = K.bn_act_fwd(x[lo:hi], c, m, r, gamma, beta, act, None if res is None else res[lo:hi], post,
                          None if rowscale is None else rowscale[lo:hi], rows_hw)
        assert rel_err(y[lo:hi], ys) < 1e-5          # (the row-block partition of the statistics differs: last-bit differences)
        dxs = K.bn_act_bwd(x[lo:hi], dy[lo:hi], c, m, r, gamma, beta, act, True, dg2, db2, True, None if y_out is None else y_out[lo:hi],
                           None if not post else rowscale[lo:hi], rows_hw)
        assert rel_err(dx[lo:hi], dxs) < 1e-5
    assert rel_err(rm1, rm2) < 5e-6 and rel_err(rv1, rv2) < 5e-6 and int(nb1) == int(nb2) == 2
    assert rel_err(dg1, dg2) < 1e-5 and rel_err(db1, db2) < 1e-5


def test_merged_pass_with_per_segment_launches(monkeypatch):
    """the fallback that runs the one-segment kernels on the two halves stays equivalent too"""
    from semiuhpe_amd import layers as L
    monkeypatch.setattr(L, "SEGMENT_LAUNCHES", "split")
    test_merged_pass_equals_two_passes("effinetv2", 64, 3, 4)


@pytest.mark.parametrize("hw,c,n,seg,act", [(49, 1536, 6, 2, "silu"), (196, 96, 5, 0, "silu"), (25, 6, 4, 3, "relu"), (1, 512, 8, 0, "relu6")])
def test_batchnorm_apply_with_fused_squeeze(hw, c, n, seg, act):
    """suhpe_bn_act_pool_fwd: BatchNorm + activation and the per-sample channel means of the result from one pass, against
    the apply kernel followed by the pooling kernel (one and two segments)."""
    from semiuhpe_amd import kernels as K
    g = torch.Generator().manual_seed(hw + c)
    x = torch.randn(n, hw, c, generator=g).cuda()
    gamma, beta = (torch.rand(c, generator=g) + 0.5).cuda(), torch.randn(c, generator=g).cuda()
    if seg:
        mean, rstd = K.bn_stats_seg(x, seg * hw, c, 1e-3, 0.1)
        want = K.bn_act_fwd_seg(x, seg * hw, c, mean, rstd, gamma, beta, act)
    else:
        mean, rstd = K.bn_stats(x, c, 1e-3, 0.1)
        want = K.bn_act_fwd(x, c, mean, rstd, gamma, beta, act)
    y, pool = K.bn_act_pool_fwd(x, seg * hw, c, mean, rstd, gamma, beta, act, hw)
    assert torch.equal(y, want)
    assert rel_err(pool, want.double().mean(1)) < 1e-6
    assert rel_err(pool, K.avgpool_fwd(want.view(n, hw, 1, c))) < 1e-6


@pytest.mark.parametrize("hw,c,n,seg,nb", [(196, 512, 6, 2, 3), (49, 2048, 5, 0, 2), (25, 6, 4, 3, 3), (784, 128, 3, 0, 3)])
def test_repvgg_block_tail_fused_batchnorms(hw, c, n, seg, nb):
    """suhpe_repvgg_bn_fwd / _bwd (relu of the sum of 2-3 BatchNorm branches and its backward as one kernel each) against the
    chain of one-branch kernels the block used before (and still uses in eval mode): outputs, dx per branch, dgamma, dbeta."""
    from semiuhpe_amd import kernels as K
    g = torch.Generator().manual_seed(hw + c + nb)
    xs = [(torch.randn(n, hw, 1, c, generator=g) * (1 + j) + 0.3 * j).cuda() for j in range(nb)]
    gam = [(torch.rand(c, generator=g) + 0.5).cuda() for _ in range(nb)]
    bet = [torch.randn(c, generator=g).cuda() for _ in range(nb)]
    dout = torch.randn(n, hw, 1, c, generator=g).cuda()
    rows_a = seg * hw
    if seg:
        st = [K.bn_stats_seg(x, rows_a, c, 1e-5, 0.1) for x in xs]
        fwd = lambda x, m, r, ga, be, act, res: K.bn_act_fwd_seg(x, rows_a, c, m, r, ga, be, act, res)
        bwd = lambda x, dy, m, r, ga, be, act, dg, db, yo: K.bn_act_bwd_seg(x, dy, rows_a, c, m, r, ga, be, act, dg, db, True, yo)
    else:
        st = [K.bn_stats(x, c, 1e-5, 0.1) for x in xs]
        fwd = lambda x, m, r, ga, be, act, res: K.bn_act_fwd(x, c, m, r, ga, be, act, res)
        bwd = lambda x, dy, m, r, ga, be, act, dg, db, yo: K.bn_act_bwd(x, dy, c, m, r, ga, be, act, True, dg, db, True, yo)
    means, rstds = [s[0] for s in st], [s[1] for s in st]
    # the unfused chain: u = BN(x1) (+ BN(x2)); y = relu(BN(x0) + u)
    u = fwd(xs[1], means[1], rstds[1], gam[1], bet[1], "none", None)
    if nb == 3:
        u = fwd(xs[2], means[2], rstds[2], gam[2], bet[2], "none", u)
    want = fwd(xs[0], means[0], rstds[0], gam[0], bet[0], "relu", u)
    y = K.repvgg_bn_fwd(xs, means, rstds, gam, bet, rows_a, c)
    assert rel_err(y, want) < 1e-5
    dg1 = [torch.zeros(c).cuda() for _ in range(nb)]; db1 = [torch.zeros(c).cuda() for _ in range(nb)]
    dg2 = [torch.zeros(c).cuda() for _ in range(nb)]; db2 = [torch.zeros(c).cuda() for _ in range(nb)]
    dxs = K.repvgg_bn_bwd(xs, means, rstds, gam, dout, y, rows_a, c, dg1, db1)
    gmask = K.act_bwd(y, dout, c, "relu")
    ref0 = bwd(xs[0], dout, means[0], rstds[0], gam[0], bet[0], "relu", dg2[0], db2[0], y)
    refs = [ref0] + [bwd(xs[j], gmask, means[j], rstds[j], gam[j], bet[j], "none", dg2[j], db2[j], None) for j in range(1, nb)]
    for j in range(nb):
        assert rel_err(dxs[j], refs[j]) < 2e-5, j
        assert rel_err(dg1[j], dg2[j]) < 2e-5 and rel_err(db1[j], db2[j]) < 2e-5, j
    # suhpe_repvgg_bn_bwd_remask: the ReLU mask recomputed from the branch inputs (the forward's arithmetic) instead of read from y.  The MASK is
    # the same bit for bit -- dbeta = sum of the masked gradient is identical --; the two template instantiations may contract the arithmetic
    # behind it differently (one ulp on some elements of one branch was seen)
    dg3 = [torch.zeros(c).cuda() for _ in range(nb)]; db3 = [torch.zeros(c).cuda() for _ in range(nb)]
    dxr = K.repvgg_bn_bwd(xs, means, rstds, gam, dout, None, rows_a, c, dg3, db3, betas=bet)
    for j in range(nb):
        assert torch.equal(db3[j], db1[j]), j
        assert rel_err(dxr[j], dxs[j]) < 1e-6 and rel_err(dg3[j], dg1[j]) < 1e-6, j
    # ... also where pre-activations sit exactly on the ReLU's edge: all-zero inputs with beta = 0 give a == 0 -> mask off in both forms
    zs = [torch.zeros_like(x) for x in xs]
    zb = [torch.zeros(c).cuda() for _ in range(nb)]
    zm = [torch.zeros_like(m) for m in means]
    y0 = K.repvgg_bn_fwd(zs, zm, rstds, gam, zb, rows_a, c)
    assert float(y0.abs().max()) == 0.0
    da = K.repvgg_bn_bwd(zs, zm, rstds, gam, dout, y0, rows_a, c, [torch.zeros(c).cuda() for _ in range(nb)], [torch.zeros(c).cuda() for _ in range(nb)])
    db_ = K.repvgg_bn_bwd(zs, zm, rstds, gam, dout, None, rows_a, c, [torch.zeros(c).cuda() for _ in range(nb)], [torch.zeros(c).cuda() for _ in range(nb)], betas=zb)
    for j in range(nb):
        assert rel_err(da[j], db_[j]) < 1e-6, j


@pytest.mark.parametrize("hw,c,n,seg,nb", [(196, 512, 6, 2, 3), (49, 2048, 5, 0, 2), (25, 6, 4, 3, 3), (784, 128, 40, 8, 3), (3136, 128, 20, 0, 3)])
def test_repvgg_block_tail_emits_the_statistics_of_its_output(hw, c, n, seg, nb):
    """suhpe_repvgg_bn_fwd_stats: the same output bits as suhpe_repvgg_bn_fwd, and (S, Q, k, n) records of that output which finalise to the
    BatchNorm statistics a separate pass over it gives (one and two batch segments; a post-ReLU channel with a large mean and a tiny spread
    is where an unshifted sum of squares would lose the variance)"""
    from semiuhpe_amd import kernels as K
    g = torch.Generator().manual_seed(hw * 3 + c + nb)
    xs = [(torch.randn(n, hw, 1, c, generator=g) * (1 + j) + 0.3 * j).cuda() for j in range(nb)]
    xs[0][..., 0] = 40.0 + 0.01 * torch.randn(n, hw, 1, generator=g).cuda()          # channel 0: mean ~ 40 sigma, spread 1e-2
    gam = [(torch.rand(c, generator=g) + 0.5).cuda() for _ in range(nb)]
    bet = [torch.randn(c, generator=g).cuda() for _ in range(nb)]
    rows, rows_a = n * hw, seg * hw
    st = [K.bn_stats_seg(x, rows_a, c, 1e-5, 0.1) if seg else K.bn_stats(x, c, 1e-5, 0.1) for x in xs]
    means, rstds = [s_[0] for s_ in st], [s_[1] for s_ in st]
    y0 = K.repvgg_bn_fwd(xs, means, rstds, gam, bet, rows_a, c)
    y, partials = K.repvgg_bn_fwd(xs, means, rstds, gam, bet, rows_a, c, emit_stats=True)
    assert torch.equal(y, y0)
    rm0, rv0, rm1, rv1 = (torch.zeros(c).cuda(), torch.ones(c).cuda(), torch.zeros(c).cuda(), torch.ones(c).cuda())
    nb0, nb1 = torch.zeros(1, dtype=torch.long).cuda(), torch.zeros(1, dtype=torch.long).cuda()
    if seg:
        part, tiles, (first, boundary) = partials
        assert boundary is None and 0 < first < tiles
        m1, r1 = K.bn_stats_from_partials_seg(part, tiles, first, None, rows, rows_a, c, 1e-5, 0.1, rm1, rv1, nb1)
        m0, r0 = K.bn_stats_seg(y, rows_a, c, 1e-5, 0.1, rm0, rv0, nb0)
    else:
        part, tiles = partials
        m1, r1 = K.bn_stats_from_partials(part, tiles, rows, c, 1e-5, 0.1, rm1, rv1, nb1)
        m0, r0 = K.bn_stats(y, c, 1e-5, 0.1, rm0, rv0, nb0)
    assert float((m1 - m0).abs().max()) < 2e-6 * max(1.0, float(m0.abs().max())) and rel_err(r1, r0) < 2e-5
    assert rel_err(rm1, rm0) < 2e-6 and rel_err(rv1, rv0) < 2e-5 and int(nb1) == int(nb0)


def test_repvgg_stage_uses_the_emitted_statistics(monkeypatch):
    """a stage of RepVGG blocks: with the output statistics handed from block to block (default) the step gives the results of the form in which
    every identity BatchNorm runs its own statistics pass (SUHPE_REPVGG_OUT_STATS=0) -- outputs, input gradient, every parameter gradient and the
    running statistics -- and launches no bn_stats pass for the identity branches"""
    from semiuhpe_amd import kernels as K
    from semiuhpe_amd.src import repvgg as R
    from semiuhpe_amd import layers as L
    torch.manual_seed(7)
    net = R.RepVggNet().cuda().train()
    stage = net.layer2                                      # 128 -> 256 (stride 2), then five blocks with an identity branch
    x = torch.randn(6, 28, 28, 128, device="cuda")
    dy = torch.randn(6, 14, 14, 256, device="cuda")

    def run(flag):
        monkeypatch.setattr(R, "OUT_STATS", flag)
        for p in stage.parameters():
            p.grad = None
        state = {k: v.clone() for k, v in stage.state_dict().items()}
        calls = []
        orig = K.bn_stats_seg
        monkeypatch.setattr(K, "bn_stats_seg", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
        with L.batch_segments(2):
            tape = []
            y = stage.fwd(x, tape)
            dx = stage.bwd(dy, tape)
        monkeypatch.setattr(K, "bn_stats_seg", orig)
        out = (y, dx, {k: p.grad.clone() for k, p in stage.named_parameters()}, {k: v.clone() for k, v in stage.state_dict().items() if "running" in k})
        stage.load_state_dict(state)
        return out, len(calls)

    (y1, dx1, g1, r1), n1 = run(True)
    (y0, dx0, g0, r0), n0 = run(False)
    assert n1 == 0 and n0 == 5                              # layer2: a stride-2 block, then five blocks with an identity branch, each fed records
    assert rel_err(y1, y0) < 1e-5 and rel_err(dx1, dx0) < 1e-4
    for k in g0:
        assert rel_err(g1[k], g0[k]) < 2e-4, k
    for k in r0:
        assert rel_err(r1[k], r0[k]) < 1e-5, k


def test_repvgg_tail_backward_falls_back_when_the_weights_moved(monkeypatch):
    """the fused block tail recomputes its ReLU mask from the LIVE gamma / beta (REMASK): exact only while no parameter was written between the
    forward and the backward.  A write in between -- torch-side (_version) or by a raw-pointer kernel (kernels.PARAM_EPOCH: Adam, EMA) -- must
    send the backward to the mask of the stored output: the gradients then equal the SUHPE_REPVGG_REMASK=0 form bit for bit, while a recomputed
    mask from the moved beta would differ"""
    from semiuhpe_amd import kernels as K
    from semiuhpe_amd.src import repvgg as R
    torch.manual_seed(11)
    blk = R.RepVGGBlock(64, 64, 1, 1).cuda().train()
    x = torch.randn(4, 14, 14, 64, device="cuda")
    dy = torch.randn(4, 14, 14, 64, device="cuda")
    calls = []
    orig = K.repvgg_bn_bwd
    monkeypatch.setattr(K, "repvgg_bn_bwd", lambda *a, **k: (calls.append(k.get("betas") is not None), orig(*a, **k))[1])

    def run(remask, disturb):
        monkeypatch.setattr(R, "REMASK", remask)
        state = {k: v.clone() for k, v in blk.state_dict().items()}
        for p in blk.parameters():
            p.grad = None
        tape = []
        blk.fwd(x, tape)
        if disturb == "torch":
            with torch.no_grad():
                blk.rbr_dense.bn.bias.add_(0.75)                  # moves the pre-activation: a mask recomputed from it is another mask
        elif disturb == "kernel":
            K.ema_update(blk.rbr_dense.bn.bias.data, blk.rbr_dense.bn.bias.data + 0.75, 0.0)      # raw-pointer write: only PARAM_EPOCH sees it
        dx = blk.bwd(dy, tape)
        out = (dx.clone(), {k: p.grad.clone() for k, p in blk.named_parameters()})
        blk.load_state_dict(state)
        return out

    for disturb in ("torch", "kernel"):
        calls.clear()
        dx1, g1 = run(True, disturb)
        dx0, g0 = run(False, disturb)
        assert calls == [False, False], (disturb, calls)           # both backward passes took the stored-output mask
        assert torch.equal(dx1, dx0), disturb
        for k in g0:
            assert torch.equal(g1[k], g0[k]), (disturb, k)
    calls.clear()
    run(True, None)
    assert calls == [True]                                          # undisturbed: the recomputed mask (the fast path) is what runs


def test_agent_step_with_no_pseudo_label_selected(tmp_path):
    """Edge of the pseudo-label filter (reference src/agent.py:163-168): when no unlabeled sample passes the entropy threshold the
    unsupervised loss is exactly 0 (not NaN), mask_ratio is 0, loss_all equals the supervised loss and every gradient stays finite."""
    from semiuhpe_amd.src.agent import SSLAgent
    from semiuhpe_amd.synthetic import synthetic_batches
    cfg = SimpleNamespace(
        lr=1e-4, log_dir=str(tmp_path), model_dir=str(tmp_path), distribution="matrixFisher", is_full_range=False, rotate_aug=True,
        train_labeled="300WLP", train_unlabeled="300WLP", cutout_aug=True, cutmix_aug=True, conf_thres=-1e9, type_unsuper="ce", SSL_lambda=1.0,
        is_ema=True, ema_decay=0.999, eman=False, stage1_iteration=0, network="effinetv2", num_classes=9, exp_name="t", date="t", left_ratio=0.95,
        save_feat=False, dynamic_thres=False, autotune=False)
    torch.manual_seed(5)
    agent = SSLAgent(cfg)
    data, ulb = synthetic_batches(4, 8, size=96, seed=1, rotate_aug=True)
    to = lambda d: {k: v.cuda().contiguous() for k, v in d.items()}
    fd, fu, out = agent.train_func(to(data), to(ulb))
    assert float(fu["mask_ratio"]) == 0.0
    assert float(fu["unsuper_loss"].detach()) == 0.0
    assert float(out["loss_all"].detach()) == float(fd["loss"].detach())
    agent.check_finite()
    for n, p in agent.net.named_parameters():
        assert bool(torch.isfinite(p.grad).all()) and bool(torch.isfinite(p).all()), n
