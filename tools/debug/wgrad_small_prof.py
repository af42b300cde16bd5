"""a few filter-gradient launches of one 3x3 conv on the Winograd filter-gradient tile (target for rocprofv3 --kernel-trace --stats):
    python3 tools/debug/wgrad_small_prof.py cin cout groups hw n"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from semiuhpe_amd import _lib, kernels as K
cin, cout, g, hw, n = (int(v) for v in sys.argv[1:6])
lib = _lib.load()
lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
names = [lib.suhpe_conv2d_tile_name(2, i).decode() for i in range(lib.suhpe_conv2d_tile_count(2))]
x = torch.randn(n, hw, hw, cin, device="cuda"); dy = torch.randn(n, hw, hw, cout, device="cuda")
w = K.to_filter_layout(torch.randn(cout, cin // g, 3, 3, device="cuda"))
d = K.conv_desc(x.shape, w.shape, 1, 1, g)
assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), 2, names.index("wino_wgrad_kernel<64,64>")) == 0
flush = torch.empty(1 << 26, device="cuda")
for _ in range(5):
    flush.zero_()
    K.conv2d_wgrad(x, dy, w.shape, 1, 1, g)
torch.cuda.synchronize()
