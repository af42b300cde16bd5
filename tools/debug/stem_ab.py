"""RepVGG train-mode forward/backward with the stem conv on the direct kernel vs the GEMM (SUHPE_STEM_DIRECT=0): python tools/debug/stem_ab.py save <file> ; ... compare a b"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
if sys.argv[1] == "save":
    from semiuhpe_amd.src.repvgg import RepVggNet
    from semiuhpe_amd import layers as L
    from test_gpu_backbones import seeded_state_dict, G
    m = RepVggNet(); m.load_state_dict(seeded_state_dict(m.state_dict(), 11)); m.cuda().train()
    for mod in m.modules():
        if isinstance(mod, (L.Dropout, L.StochasticDepth)) and hasattr(mod, "p"): mod.p = 0.0
    x, dy = torch.from_numpy(G["repvgg/x"]).cuda(), torch.from_numpy(G["repvgg/dy"]).cuda()
    acts = {}
    blk = m.layer0 if hasattr(m, "layer0") else None
    out = m(x); out.backward(dy)
    d = {"out": out.detach().cpu().numpy()}
    for k, p in m.named_parameters():
        d["g/" + k] = p.grad.detach().cpu().numpy()
    for k, v in m.state_dict().items():
        if "running" in k: d["b/" + k] = v.cpu().numpy()
    np.savez(sys.argv[2], **d)
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    rows = []
    for k in a.files:
        sc = max(1e-12, np.abs(a[k]).max())
        rows.append((np.abs(a[k] - b[k]).max() / sc, k))
    rows.sort(reverse=True)
    for e, k in rows[:12]: print(f"{e:.3e} {k}")
    print("median", np.median([e for e, _ in rows]))
