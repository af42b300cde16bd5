"""times of the Winograd tile on three shapes (forward with a pre-transformed filter, cold caches); used by tools/wino_ablate.sh"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from semiuhpe_amd import _lib, kernels as K
lib = _lib.load()
lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
names = [lib.suhpe_conv2d_tile_name(0, i).decode() for i in range(lib.suhpe_conv2d_tile_count(0))]
flush = torch.empty(1 << 26, dtype=torch.float32, device="cuda")
out = []
SHAPES = ((48, 192, 1, 56, 160), (64, 256, 1, 28, 160), (512, 512, 1, 14, 320))
if os.environ.get("WINO_TIME_SHAPES"):      # "cin,cout,groups,hw,n;..."
    SHAPES = [tuple(int(v) for v in sh.split(",")) for sh in os.environ["WINO_TIME_SHAPES"].split(";")]
for cin, cout, g, hw, n in SHAPES:
    x = torch.randn(n, hw, hw, cin, device="cuda")
    w = K.to_filter_layout(torch.randn(cout, cin // g, 3, 3, device="cuda") * 0.05)
    d = K.conv_desc(x.shape, w.shape, 1, 1, g)
    assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), 0, names.index("wino_nt_kernel<64,64>")) == 0
    fn = (lambda: K.conv2d_fwd_stats(x, w, 1, 1, g, seg=32)) if os.environ.get("WINO_STATS") else (lambda: K.conv2d_fwd(x, w, None, 1, 1, g))
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    out.append(f"{cin}->{cout}@{hw} n{n}: {best*1e3:.0f} us")
print("  " + " | ".join(out))
