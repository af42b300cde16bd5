"""FusedMBConv backward pieces on the C2 shapes: plain project dgrad + BatchNorm reduction pass  vs  suhpe_conv2d_dgrad_bnbwd"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from semiuhpe_amd import kernels as K
flush = torch.zeros(1 << 28, dtype=torch.float32, device="cuda")
def timed(fn, reps=7):
    ts = []
    for _ in range(reps):
        flush.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(400000)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2] * 1e3
for n, h, cexp, cout in ((160, 56, 192, 48), (160, 28, 256, 64)):
    y0 = torch.randn(n, h, h, cexp, device="cuda"); dt = torch.randn(n, h, h, cout, device="cuda")
    w = K.to_filter_layout(torch.randn(cout, cexp, 1, 1, device="cuda") / 8)
    gamma, beta = torch.rand(cexp, device="cuda") + 0.5, torch.randn(cexp, device="cuda") * 0.2
    seg = 32
    mean, rstd = K.bn_stats_seg(y0, seg * h * h, cexp, 1e-3, 0.1)
    sc, sh = K.bn_affine_from_stats(mean, rstd, gamma, beta)
    dg, db = torch.zeros(cexp, device="cuda"), torch.zeros(cexp, device="cuda")
    dz = K.conv2d_dgrad(dt, w, y0.shape, 1, 0, 1)
    t0 = timed(lambda: K.conv2d_dgrad(dt, w, y0.shape, 1, 0, 1))
    t1 = timed(lambda: K.bn_act_bwd_seg(y0, dz, seg * h * h, cexp, mean, rstd, gamma, beta, "silu", dg, db, True))
    t2 = timed(lambda: K.conv2d_dgrad_bnbwd(dt, w, y0.shape, y0, mean, rstd, sc, sh, "silu", seg, dg, db))
    g, sums = K.conv2d_dgrad_bnbwd(dt, w, y0.shape, y0, mean, rstd, sc, sh, "silu", seg, dg, db)
    t3 = timed(lambda: K.bn_act_bwd_apply_seg(y0, g, seg * h * h, cexp, mean, rstd, gamma, beta, "none", sums))
    print(f"{n}x{h}x{h} {cexp}->{cout}: dgrad {t0:.1f} us + BN backward (partial, finalize, apply) {t1:.1f} us = {t0 + t1:.1f}  |  dgrad_bnbwd {t2:.1f} + apply {t3:.1f} = {t2 + t3:.1f}")
