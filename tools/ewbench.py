"""The HBM-bound kernels of an MBConv block on the C2 shapes, one launch at a time, in the cache state the step gives them: caches flushed
by a READ of 1 GiB (clean lines), then the kernel's inputs re-written by a producer copy (as the conv that precedes it would), then the
timed launch.    python tools/ewbench.py [n]        columns: us, algorithmic GB/s (tensor passes x bytes / time)
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from semiuhpe_amd import kernels as K

N = int(sys.argv[1]) if len(sys.argv) > 1 else 160
flush = torch.zeros(1 << 28, dtype=torch.float32, device="cuda")


def timed(fn, inputs, reps=7):
    srcs = [t.clone() for t in inputs]
    ts = []
    for _ in range(reps):
        flush.sum()
        for t, s in zip(inputs, srcs):
            t.copy_(s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(400000)          # ~0.2 ms of device time: the host enqueues the timed launch meanwhile (no host latency in the figure)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2] * 1e3


def line(name, us, passes, nbytes):
    print(f"  {name:34s} {us:8.1f} us {passes * nbytes / us / 1e3:7.0f} GB/s ({passes} passes)", flush=True)


for hw_side, c in ((14, 512), (14, 960), (7, 1536)):
    hw = hw_side * hw_side
    seg = 32 if N > 32 else 0
    x = torch.randn(N, hw_side, hw_side, c, device="cuda"); dy = torch.randn_like(x)
    gamma = torch.rand(c, device="cuda") + 0.5; beta = torch.randn(c, device="cuda") * 0.1
    rows, rows_a = N * hw, seg * hw
    mean2 = torch.zeros(2, c, device="cuda"); rstd2 = torch.ones(2, c, device="cuda")
    mean1, rstd1 = mean2[0].contiguous(), rstd2[0].contiguous()
    m, r = (mean2, rstd2) if seg else (mean1, rstd1)
    s = torch.rand(N, c, device="cuda"); dpool = torch.randn(N, c, device="cuda")
    w = (torch.randn(c, 1, 3, 3, device="cuda") * 0.2).contiguous(memory_format=torch.channels_last)
    nb = x.numel() * 4
    print(f"== {N} x {hw_side}x{hw_side} x {c}  ({nb / 1e6:.1f} MB per tensor), first segment {seg} samples")
    line("bn_act_fwd (silu)", timed(lambda: K.bn_act_fwd_seg(x, rows_a, c, m, r, gamma, beta, "silu"), [x]), 2, nb)
    line("bn_act_pool (squeeze)", timed(lambda: K.bn_act_pool(x, rows_a, c, m, r, gamma, beta, "silu", hw), [x]), 1, nb)
    line("bn_act_scale_fwd", timed(lambda: K.bn_act_scale_fwd(x, rows_a, c, m, r, gamma, beta, "silu", s, hw), [x]), 2, nb)
    line("se_bn_bwd_reduce", timed(lambda: K.se_bn_bwd_reduce(x, dy, rows_a, c, m, r, gamma, beta, "silu", hw), [x, dy]), 2, nb)
    sums = K.se_bn_bwd_reduce(x, dy, rows_a, c, m, r, gamma, beta, "silu", hw)
    dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    chan = K.se_bn_bwd_combine(sums, s, dpool, seg, hw, dg, db)
    line("se_bn_bwd_apply", timed(lambda: K.se_bn_bwd_apply(x, dy, rows_a, c, m, r, gamma, beta, "silu", s, dpool, chan, True, hw), [x, dy]), 3, nb)
    line("bn_act_bwd_sums (silu)", timed(lambda: K.bn_act_bwd_sums(x, dy, c, mean1, rstd1, gamma, beta, "silu"), [x, dy]), 2, nb)
    bsums = K.bn_act_bwd_sums(x, dy, c, mean1, rstd1, gamma, beta, "silu")
    line("bn_act_bwd_apply (silu)", timed(lambda: K.bn_act_bwd_apply(x, dy, c, mean1, rstd1, gamma, beta, "silu", bsums, rows), [x, dy]), 3, nb)
    csq = {512: 32, 960: 40, 1536: 64}[c]
    w1 = torch.randn(csq, c, device="cuda") * 0.05; b1 = torch.zeros(csq, device="cuda"); w2 = torch.randn(c, csq, device="cuda") * 0.05; b2 = torch.zeros(c, device="cuda")
    pl = torch.randn(N, c, device="cuda")
    us = timed(lambda: K.se_mlp_fwd(pl, w1, b1, w2, b2), [pl])
    print(f"  {'se_mlp_fwd':34s} {us:8.1f} us", flush=True)
    line("dwconv_fwd_stats", timed(lambda: K.dwconv_fwd_stats(x, w, 1, 1, seg=seg or None), [x]), 2, nb)
    if K.dwconv_dgrad_bnsums_ok(tuple(x.shape), tuple(w.shape), 1, 1):
        line("dwconv_dgrad_bnsums", timed(lambda: K.dwconv_dgrad_bnsums(dy, w, tuple(x.shape), 1, 1, x, m, r, gamma, beta, "silu", seg, dg, db), [x, dy]), 3, nb)
    line("dwconv_dgrad", timed(lambda: K.conv2d_dgrad(dy, w, tuple(x.shape), 1, 1, c), [dy]), 2, nb)
    line("dwconv_wgrad", timed(lambda: K.conv2d_wgrad(x, dy, tuple(w.shape), 1, 1, c), [x, dy]), 2, nb)
    # round 6: the depthwise conv's whole backward as one launch against the kernels it replaces
    e0 = torch.randn_like(x); dwg = torch.zeros_like(w)
    line("dwconv_bnact_wgrad (replaced)", timed(lambda: K.dwconv_bnact_wgrad(e0, m, r, gamma, beta, "silu", dy, tuple(w.shape), 1, 1, out=dwg, accumulate=True, seg=seg or None), [e0, dy]), 2, nb)
    if K.dwconv_bwd_fused_ok(tuple(x.shape), tuple(w.shape), 1, 1):
        line("dwconv_bwd_fused (plain: dy given)", timed(lambda: K.dwconv_bwd_fused(dy, w, tuple(x.shape), 1, 1, e0, m, r, gamma, beta, "silu", seg, dg, db, dwg), [e0, dy]), 3, nb)
        mid = (x, s, dpool, chan, m, r, gamma, beta, "silu", True)
        line("dwconv_bwd_fused (section form)", timed(lambda: K.dwconv_bwd_fused(dy, w, tuple(x.shape), 1, 1, e0, m, r, gamma, beta, "silu", seg, dg, db, dwg, mid=mid), [e0, dy, x]), 4, nb)
