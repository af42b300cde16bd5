"""One 3x3 / stride 1 conv through the Winograd tile, a few launches (target for rocprofv3 --pmc runs):
    python3 tools/wino_pmc.py cin cout groups hw n [pass]      pass: 0 forward (+ statistics), 1 input gradient
"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from semiuhpe_amd import _lib, kernels as K

cin, cout, g, hw, n = (int(v) for v in sys.argv[1:6])
pid = int(sys.argv[6]) if len(sys.argv) > 6 else 0
lib = _lib.load()
lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
names = [lib.suhpe_conv2d_tile_name(0, i).decode() for i in range(lib.suhpe_conv2d_tile_count(0))]
x = torch.randn(n, hw, hw, cin, device="cuda")
w = K.to_filter_layout(torch.randn(cout, cin // g, 3, 3, device="cuda") * 0.05)
dy = torch.randn(n, hw, hw, cout, device="cuda")
d = K.conv_desc(x.shape, w.shape, 1, 1, g)
assert lib.suhpe_conv2d_set_tile(ctypes.byref(d), pid, names.index("wino_nt_kernel<64,64>")) == 0
flush = torch.empty(1 << 26, device="cuda")
for _ in range(4):
    flush.zero_()
    if pid == 0:
        K.conv2d_fwd_stats(x, w, 1, 1, g)
    else:
        K.conv2d_dgrad(dy, w, x.shape, 1, 1, g)
torch.cuda.synchronize()
