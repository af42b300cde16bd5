"""BatchNorm(+act) kernels on the step's dominant shapes, cold (L2/MALL evicted between launches) and warm.
    python tools/bnbench.py            # event timings
    rocprofv3 --kernel-trace --stats ... -- python3 tools/bnbench.py   # per-kernel durations
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from semiuhpe_amd import kernels as K

flush = torch.empty(1 << 28, dtype=torch.float32, device="cuda")


def cold(fn, iters=6):
    tot = 0.0
    for _ in range(iters):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e-3


SILU = "silu"
for shape in ((128, 7, 7, 1536), (128, 14, 14, 960), (32, 7, 7, 1536), (128, 56, 56, 192), (128, 112, 112, 24), (128, 7, 7, 256), (128, 14, 14, 160)):
    c = shape[-1]
    x = torch.randn(shape, device="cuda"); dy = torch.randn(shape, device="cuda")
    gamma = torch.rand(c, device="cuda") + 0.5; beta = torch.randn(c, device="cuda")
    rm, rv, nbt = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda"), torch.zeros((), dtype=torch.int64, device="cuda")
    mean, rstd = K.bn_stats(x, c, 1e-3, 0.1, rm, rv, nbt)
    dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    nb = x.numel() * 4
    ts = cold(lambda: K.bn_stats(x, c, 1e-3, 0.1, rm, rv, nbt))
    tf = cold(lambda: K.bn_act_fwd(x, c, mean, rstd, gamma, beta, SILU))
    tb = cold(lambda: K.bn_act_bwd(x, dy, c, mean, rstd, gamma, beta, SILU, True, dg, db, True))
    print(f"{str(shape):22s} {nb/2**20:6.1f} MiB | stats {ts*1e6:7.1f} us ({nb/ts/1e9:5.0f} GB/s) | fwd {tf*1e6:7.1f} us ({2*nb/tf/1e9:5.0f} GB/s) | "
          f"bwd {tb*1e6:7.1f} us ({3*nb/tb/1e9:5.0f} GB/s algorithmic, {5*nb/tb/1e9:5.0f} actual)", flush=True)
