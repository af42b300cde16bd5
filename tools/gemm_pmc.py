"""One plain GEMM shape through the conv kernels, a few launches (target for rocprofv3 --pmc runs):
    python3 tools/gemm_pmc.py M N K [pass] [tile]      pass: 0 fwd (igemm_nt), 2 wgrad (igemm_tn); tile: menu index to pin
"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from semiuhpe_amd import _lib, kernels as K

m, n, k = (int(v) for v in sys.argv[1:4])
pid = int(sys.argv[4]) if len(sys.argv) > 4 else 0
tile = int(sys.argv[5]) if len(sys.argv) > 5 else -1
x = torch.randn(m, 1, 1, k, device="cuda")
w = torch.randn(n, k, 1, 1, device="cuda")
dy = torch.randn(m, 1, 1, n, device="cuda")
d = K.conv_desc(x.shape, w.shape, 1, 0, 1)
_lib.load().suhpe_conv2d_set_tile(ctypes.byref(d), pid, tile)
flush = torch.empty(1 << 26, device="cuda")
for _ in range(5):
    flush.zero_()
    if pid == 0:
        K.conv2d_fwd(x, w, None, 1, 0, 1)
    else:
        K.conv2d_wgrad(x, dy, w.shape, 1, 0, 1)
torch.cuda.synchronize()
