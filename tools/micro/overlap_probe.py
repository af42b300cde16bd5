"""How much of an HBM-bound kernel hides under an MFMA-bound GEMM when the two run on different streams?
    python tools/micro/overlap_probe.py
For pairs (GEMM shape, elementwise tensor): time of N GEMMs alone, N elementwise kernels alone, both back to back on one stream,
and both on two streams.  Ideal overlap = max(a, b); no overlap = a + b."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from semiuhpe_amd import kernels as K

dev = torch.device("cuda")
N = 40


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


side = torch.cuda.Stream()
for (n, hw, cin, cout, k) in [(160, 14, 160, 960, 1), (160, 14, 960, 160, 1), (160, 56, 48, 192, 3), (160, 7, 256, 1536, 1)]:
    x = torch.randn(n, hw, hw, cin, device=dev)
    w = K.to_filter_layout(torch.randn(cout, cin, k, k, device=dev))
    pad = (k - 1) // 2
    c = 960 if hw == 14 else (192 if hw == 56 else 1536)
    z = torch.randn(n, hw, hw, c, device=dev)
    mean, rstd = K.bn_stats(z, c, 1e-3, 0.1)
    g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)

    def gemms():
        for _ in range(N):
            K.conv2d_fwd(x, w, None, 1, pad, 1)

    def elems():
        for _ in range(N):
            K.bn_act_fwd(z, c, mean, rstd, g, b, "silu")

    def serial():
        gemms(); elems()

    def both():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            elems()
        gemms()
        torch.cuda.current_stream().wait_stream(side)

    def interleaved_two_streams():
        side.wait_stream(torch.cuda.current_stream())
        for _ in range(N):
            with torch.cuda.stream(side):
                K.bn_act_fwd(z, c, mean, rstd, g, b, "silu")
            K.conv2d_fwd(x, w, None, 1, pad, 1)
        torch.cuda.current_stream().wait_stream(side)

    a, bt, s, o, o2 = timed(gemms), timed(elems), timed(serial), timed(both), timed(interleaved_two_streams)
    print(f"gemm ({n},{hw},{cin}->{cout},k{k}) {a/N*1e3:7.1f} us | bn_act on {z.numel()*4/1e6:.0f} MB {bt/N*1e3:6.1f} us | one stream {s/N*1e3:7.1f} | two streams {o/N*1e3:7.1f} "
          f"(interleaved issue {o2/N*1e3:7.1f}) | ideal {max(a,bt)/N*1e3:7.1f} | hidden {(s-o)/min(a,bt)*100:4.0f} % of the shorter", flush=True)
