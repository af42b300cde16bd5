"""A few launches of the fused depthwise backward (section form) on one shape, for rocprofv3 passes:
    python tools/debug/dw_bwd_prof.py [hw=14] [c=960] [n=160] [form=section|plain|fwd]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from semiuhpe_amd import kernels as K

hw = int(sys.argv[1]) if len(sys.argv) > 1 else 14
c = int(sys.argv[2]) if len(sys.argv) > 2 else 960
n = int(sys.argv[3]) if len(sys.argv) > 3 else 160
form = sys.argv[4] if len(sys.argv) > 4 else "section"
seg = 32 if n > 32 else 0
x = torch.randn(n, hw, hw, c, device="cuda"); dy = torch.randn_like(x); e0 = torch.randn_like(x)
gamma = torch.rand(c, device="cuda") + 0.5; beta = torch.randn(c, device="cuda") * 0.1
m = torch.zeros(2 if seg else 1, c, device="cuda").squeeze(0).contiguous() if not seg else torch.zeros(2, c, device="cuda")
r = torch.ones_like(m)
s = torch.rand(n, c, device="cuda"); dpool = torch.randn(n, c, device="cuda")
w = (torch.randn(c, 1, 3, 3, device="cuda") * 0.2).contiguous(memory_format=torch.channels_last)
dg, db, dwg = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda"), torch.zeros_like(w)
chan = torch.randn(2, 2 * c, device="cuda") * 0.01
flush = torch.zeros(1 << 27, dtype=torch.float32, device="cuda")
ts = []
for _ in range(7):
    flush.sum()
    e0.copy_(e0 * 1.0)                   # (the producer has just written the operand: the step's cache state)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(400000)
    ev0.record()
    if form == "fwd":
        K.dwconv_bnact_fwd(e0, m, r, gamma, beta, "silu", w, 1, 1, seg=seg or None, stats=True)
    else:
        mid = (x, s, dpool, chan, m, r, gamma, beta, "silu", True) if form == "section" else None
        K.dwconv_bwd_fused(dy, w, tuple(x.shape), 1, 1, e0, m, r, gamma, beta, "silu", seg, dg, db, dwg, mid=mid)
    ev1.record()
    torch.cuda.synchronize()
    ts.append(ev0.elapsed_time(ev1) * 1e3)
ts.sort()
print(f"{form} {n} x {hw}x{hw} x {c}: median {ts[len(ts) // 2]:.1f} us (min {ts[0]:.1f})")
