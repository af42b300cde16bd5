"""per-block forward activations of RepVGG (train mode) with the stem on the direct kernel vs the GEMM: save / compare"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
if sys.argv[1] == "save":
    from semiuhpe_amd.src.repvgg import RepVggNet, RepVGGBlock
    from semiuhpe_amd import layers as L, kernels as K
    from test_gpu_backbones import seeded_state_dict, G
    m = RepVggNet(); m.load_state_dict(seeded_state_dict(m.state_dict(), 11)); m.cuda().train()
    x = K.nchw_to_nhwc(torch.from_numpy(G["repvgg/x"]).cuda())
    d = {}
    blocks = [(n, b) for n, b in m.named_modules() if isinstance(b, RepVGGBlock)]
    h = x
    for n, b in blocks:
        if n == blocks[0][0]:
            t3 = b.rbr_dense.conv.fwd_stats(h, None)
            d["t3"] = t3[0].cpu().numpy()
            mean, rstd, _ = b.rbr_dense.bn.train_stats(t3[0], t3[1]) if False else (None, None, None)
        h = b.fwd(h, None)
        d[n] = h.detach().cpu().numpy()
    np.savez(sys.argv[2], **d)
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        sc = max(1e-12, np.abs(a[k]).max())
        print(f"{k:12s} max rel diff {np.abs(a[k]-b[k]).max()/sc:.3e}  shape {a[k].shape}")
