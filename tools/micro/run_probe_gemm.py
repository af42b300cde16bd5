"""Where the persistent pointwise GEMM's time goes: the product kernel against copies with (1) no global loads, (2) also
no LDS stores, (3) also no barriers, (4) product loop but no output stores.  Results are meaningless numerically.
    tools/micro/probe_gemm.sh && python tools/micro/run_probe_gemm.py
"""
import ctypes, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", ".."))
probe = os.environ.get("PROBE")
if probe is None:                                         # driver: one child process per library build
    for n in os.environ.get("PROBES", "0 1 2 3 4").split():
        subprocess.call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, PROBE=n))
    sys.exit(0)
import torch
from semiuhpe_amd import _lib
if probe != "0":
    _lib.LIB_PATH = os.path.join(here, f"libsuhpe_probe{probe}.so")
from semiuhpe_amd import kernels as K

lib = _lib.load()
flush = torch.empty(1 << 26, device="cuda")
for (m, n, k) in ((25088, 960, 160), (25088, 160, 960), (6272, 256, 1536), (25088, 960, 2048)):
    x = torch.randn(m, 1, 1, k, device="cuda"); w = torch.randn(n, k, 1, 1, device="cuda")
    d = K.conv_desc(x.shape, w.shape, 1, 0, 1)
    out = []
    for tile in (6, 8):                                   # P64x64, P128x64
        lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, tile)
        K.conv2d_fwd(x, w, None, 1, 0, 1); torch.cuda.synchronize()
        best = 1e9
        for _ in range(6):
            flush.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); K.conv2d_fwd(x, w, None, 1, 0, 1); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        out.append(f"tile{tile} {best*1e3:7.1f} us {2.0*m*n*k/best/1e9:6.1f} TF")
    print(f"probe {probe}: M={m} N={n} K={k}: " + " | ".join(out), flush=True)
