"""Filter-gradient GEMM per shape and tile, in isolation (cold caches): python tools/wgrad_study.py
columns: shape | tile: us (TF/s) ...; the step's pointwise / small-channel filter gradients of C2 (n = 160)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from semiuhpe_amd import _lib, kernels as K

lib = _lib.load()
lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
flush = torch.empty(1 << 26, dtype=torch.float32, device="cuda")


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(iters):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e-3


SHAPES = [(160, 14, 960, 160, 1, 1), (160, 14, 160, 960, 1, 1), (160, 7, 1536, 256, 1, 1), (160, 7, 256, 1536, 1, 1), (160, 112, 24, 24, 3, 1),
          (160, 56, 192, 48, 1, 1), (160, 14, 512, 128, 1, 1), (160, 14, 128, 512, 1, 1), (160, 28, 256, 64, 1, 1), (160, 112, 24, 96, 3, 2),
          (160, 56, 96, 48, 1, 1), (160, 7, 256, 1280, 1, 1)]
if os.environ.get("WG_SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split(",")) for s in os.environ["WG_SHAPES"].split(";")]
names = [lib.suhpe_conv2d_tile_name(2, i).decode() for i in range(lib.suhpe_conv2d_tile_count(2))]
for n, hw, cin, cout, k, st in SHAPES:
    x = torch.randn(n, hw, hw, cin, device="cuda")
    w = K.to_filter_layout(torch.randn(cout, cin, k, k, device="cuda"))
    pad = (k - 1) // 2
    d = K.conv_desc(x.shape, w.shape, st, pad, 1)
    y = K.conv2d_fwd(x, w, None, st, pad, 1)
    dy = torch.randn_like(y)
    fl = K.conv_flops(d)
    out = []
    for idx, name in enumerate(names):
        if lib.suhpe_conv2d_set_tile(ctypes.byref(d), 2, idx) != 0:
            continue
        t = timeit(lambda: K.conv2d_wgrad(x, dy, w.shape, st, pad, 1))
        out.append((t, name.replace("igemm_tn_kernel", "tn").replace("wino_wgrad_kernel", "wino")))
    lib.suhpe_conv2d_set_tile(ctypes.byref(d), 2, -1)
    out.sort()
    print(f"{(n, hw, cin, cout, k, st)}: " + "  ".join(f"{nm} {t * 1e6:.0f}us ({fl / t / 1e12:.0f})" for t, nm in out[:6]), flush=True)
