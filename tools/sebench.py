"""SE-MLP kernels, L2-warm and cold (a 1 GiB memset between launches evicts L2 / MALL, as in a real step)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from semiuhpe_amd import kernels as K
from kbench import timeit

flush = torch.empty(1 << 28, dtype=torch.float32, device="cuda")


def cold(fn, iters=10):
    tot = 0.0
    for _ in range(iters):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e-3


for n, c, csq in ((128, 512, 32), (128, 960, 40), (128, 1536, 64), (32, 1536, 64)):
    p = torch.randn(n, c, device="cuda"); ds = torch.randn(n, c, device="cuda")
    w1 = torch.randn(csq, c, device="cuda"); b1 = torch.randn(csq, device="cuda"); w2 = torch.randn(c, csq, device="cuda"); b2 = torch.randn(c, device="cuda")
    a1, s = K.se_mlp_fwd(p, w1, b1, w2, b2)
    g = [torch.zeros_like(w1), torch.zeros_like(b1), torch.zeros_like(w2), torch.zeros_like(b2)]
    tf = timeit(lambda: K.se_mlp_fwd(p, w1, b1, w2, b2), 20)
    tb = timeit(lambda: K.se_mlp_bwd(ds, s, a1, p, w1, w2, *g), 20)
    cf = cold(lambda: K.se_mlp_fwd(p, w1, b1, w2, b2))
    cb = cold(lambda: K.se_mlp_bwd(ds, s, a1, p, w1, w2, *g))
    print(f"n={n} c={c} csq={csq}: warm fwd {tf*1e6:.1f} us bwd(mlp+wgrad) {tb*1e6:.1f} us | cold fwd {cf*1e6:.1f} us bwd {cb*1e6:.1f} us", flush=True)
