import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch, torch.nn.functional as F
from semiuhpe_amd import kernels as K
from test_gpu_backbones import G
x = torch.from_numpy(G["repvgg/x"]).cuda()
xn = K.nchw_to_nhwc(x)
g = torch.Generator().manual_seed(1)
for cout in (64, 24):
    wt = (torch.randn(cout, 3, 3, 3, generator=g) / 5).cuda()
    wg = K.to_filter_layout(wt)
    ref = F.conv2d(x.double(), wt.double(), None, 2, 1).permute(0, 2, 3, 1)
    res = K.conv2d_fwd_stats(xn, wg, 2, 1, 1)
    y = res[0]
    e = (y.double() - ref).abs()
    print(cout, "max abs err", float(e.max()), "ref max", float(ref.abs().max()), "where", np.unravel_index(int(e.argmax()), e.shape))
    rows = y.numel() // cout
    m2, r2 = K.bn_stats_from_partials(res[1], res[2], rows, cout, 1e-5, 0.1)
    m0 = ref.reshape(-1, cout).mean(0); v0 = ref.reshape(-1, cout).var(0, unbiased=False)
    print("   mean err", float((m2.double() - m0).abs().max()), "rstd rel err", float((r2.double() / (1 / (v0 + 1e-5).sqrt()) - 1).abs().max()), "records", res[2])
