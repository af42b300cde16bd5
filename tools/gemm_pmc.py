"""One plain GEMM shape through the conv kernels, a few launches (target for rocprofv3 --pmc runs):
    python3 tools/gemm_pmc.py M N K [pass]      pass: 0 fwd (igemm_nt), 2 wgrad (igemm_tn)
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from semiuhpe_amd import kernels as K

m, n, k = (int(v) for v in sys.argv[1:4])
pid = int(sys.argv[4]) if len(sys.argv) > 4 else 0
x = torch.randn(m, 1, 1, k, device="cuda")
w = torch.randn(n, k, 1, 1, device="cuda")
dy = torch.randn(m, 1, 1, n, device="cuda")
for _ in range(5):
    if pid == 0:
        K.conv2d_fwd(x, w, None, 1, 0, 1)
    else:
        K.conv2d_wgrad(x, dy, w.shape, 1, 0, 1)
torch.cuda.synchronize()
