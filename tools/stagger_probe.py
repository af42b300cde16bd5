"""Start-up stagger of the GEMM workgroups: time pointwise GEMM shapes of the C2 step per (tile, stagger quanta).

    python tools/stagger_probe.py            (run on the GPU box)

Each launch is timed on its own with HIP events after a cache flush (a 256 MiB fill); median of 9.
"""
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from semiuhpe_amd import _lib, kernels as K  # noqa: E402

lib = _lib.load()
flush = torch.empty(1 << 26, device="cuda")


def time_one(fn, reps=9):
    ts = []
    for _ in range(reps):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return statistics.median(ts), min(ts)


SHAPES = [(31360, 960, 160), (31360, 160, 960), (7840, 1536, 256), (7840, 256, 1536), (25088, 960, 160), (31360, 512, 128)]
TILES = {1: ("64x64", 6), 4: ("128x64", 3), 2: ("128x32", 5)}       # menu index -> (name, workgroups per CU by the launch bounds)

for (m, n, k) in SHAPES:
    x = torch.randn(m, 1, 1, k, device="cuda")
    w = torch.randn(n, k, 1, 1, device="cuda")
    d = K.conv_desc(x.shape, w.shape, 1, 0, 1)
    fl = 2.0 * m * n * k
    for tile, (name, occ) in TILES.items():
        lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, tile)
        row = []
        for q in (0, 1, 2, 3, 4, 6, 8):
            for first in ((256 * occ,) if q else (0,)):
                lib.suhpe_conv2d_set_stagger(q, first)
                med, best = time_one(lambda: K.conv2d_fwd(x, w, None, 1, 0, 1))
                row.append(f"q{q}: {med*1e6:6.1f}us {fl/med/1e12:5.1f}TF")
        lib.suhpe_conv2d_set_stagger(0, 0)
        print(f"M={m} N={n} K={k} {name:7s} | " + " | ".join(row), flush=True)
