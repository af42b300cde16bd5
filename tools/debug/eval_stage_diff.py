"""debug: per-stage difference of the eval-mode (no tape) EfficientNetV2-S forward between the HIP path and the CPU oracle"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import networks as ON
from parity_utils import seeded_state_dict
from semiuhpe_amd.src.efficientnet_v2 import EfficientNetV2S
from semiuhpe_amd import kernels as K

n, size = int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 96
ref = ON.EfficientNetV2S()
sd = seeded_state_dict(ref.state_dict(), 23)
ref.load_state_dict(sd); ref.eval()
hip = EfficientNetV2S(); hip.load_state_dict(sd); hip.cuda().eval()
x = torch.randn(n, 3, size, size, generator=torch.Generator().manual_seed(1005))
with torch.no_grad():
    xo = x
    xh = K.nchw_to_nhwc(x.cuda())
    for i, (mo, mh) in enumerate(zip(ref.features, hip.features)):
        subs_o = list(mo) if isinstance(mo, torch.nn.Sequential) and not hasattr(mo, "fwd_is_leaf") else [mo]
        subs_h = list(mh) if i not in (0, 8 - 1) else [mh]
        if i in (0, 7):
            xo = mo(xo); xh = mh.fwd(xh, None)
            d = (xh.permute(0, 3, 1, 2).cpu() - xo).abs().max() / xo.abs().max()
            print(f"features.{i}: rel {float(d):.3e} shape {tuple(xo.shape)}", flush=True)
            continue
        for j, (bo, bh) in enumerate(zip(mo, mh)):
            xin_o, xin_h = xo, xh
            xo = bo(xo); xh = bh.fwd(xh, None)
            d = (xh.permute(0, 3, 1, 2).cpu() - xo).abs().max() / xo.abs().max()
            print(f"features.{i}.{j}: rel {float(d):.3e} shape {tuple(xo.shape)}", flush=True)
            if d > 1e-3:
                # which samples / inner layer
                per = (xh.permute(0, 3, 1, 2).cpu() - xo).abs().flatten(1).max(1)[0]
                print("   per-sample max abs err:", [f"{float(v):.2e}" for v in per])
                # re-feed the oracle's input to isolate the block
                y2 = bh.fwd(xin_o.permute(0, 2, 3, 1).contiguous().cuda(), None)
                d2 = (y2.permute(0, 3, 1, 2).cpu() - xo).abs().max() / xo.abs().max()
                print(f"   block alone on the oracle's input: rel {float(d2):.3e}")
                xh = xo.permute(0, 2, 3, 1).contiguous().cuda()
