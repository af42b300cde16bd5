"""Depthwise 3x3 kernels on the step's shapes, cold, with achieved algorithmic bandwidth."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from semiuhpe_amd import kernels as K

flush = torch.empty(1 << 26, dtype=torch.float32, device="cuda")


def cold(fn, iters=8):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(iters):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e-3


for n, hw, c, st in ((128, 14, 960, 1), (128, 7, 1536, 1), (128, 28, 512, 2), (128, 14, 512, 1), (32, 7, 1536, 1), (128, 14, 960, 2)):
    x = torch.randn(n, hw, hw, c, device="cuda")
    w = K.to_filter_layout(torch.randn(c, 1, 3, 3, device="cuda"))
    y = K.conv2d_fwd(x, w, None, st, 1, c)
    dy = torch.randn_like(y)
    tf = cold(lambda: K.conv2d_fwd(x, w, None, st, 1, c))
    td = cold(lambda: K.conv2d_dgrad(dy, w, x.shape, st, 1, c))
    tw = cold(lambda: K.conv2d_wgrad(x, dy, w.shape, st, 1, c))
    nb = (x.numel() + y.numel()) * 4
    print(f"n={n} {hw}x{hw} c={c} s={st}: fwd {tf*1e6:6.1f} us ({nb/tf/1e9:5.0f} GB/s)  dgrad {td*1e6:6.1f} us ({nb/td/1e9:5.0f} GB/s)  wgrad {tw*1e6:6.1f} us ({nb/tw/1e9:5.0f} GB/s)", flush=True)
