import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
from semiuhpe_amd.src.repvgg import RepVggNet
from semiuhpe_amd import layers as L
import test_gpu_backbones as T
from oracle import networks as ON
G = T.G
m = RepVggNet(); m.load_state_dict(T.seeded_state_dict(m.state_dict(), 11)); m.cuda().train()
o = ON.RepVggNet(); o.load_state_dict(T.seeded_state_dict(o.state_dict(), 11))
for mod in m.modules():
    if isinstance(mod, (L.Dropout, L.StochasticDepth)) and hasattr(mod, "p"): mod.p = 0.0
x, dy = torch.from_numpy(G["repvgg/x"]), torch.from_numpy(G["repvgg/dy"])
print("input", tuple(x.shape))
out = m(x.cuda()); out.backward(dy.cuda())
truth = T.fp64_truth(o, x, dy)
params = dict(m.named_parameters())
rows = []
for k in sorted({k[len("repvgg/grad/"):].split("@")[0] for k in G.files if k.startswith("repvgg/grad/")}):
    key = "repvgg/grad/" + k
    got = params[k].grad.detach().cpu().double().numpy().reshape(-1); t = truth[k].numpy().reshape(-1)
    if key in G.files: ref, tt, gg = G[key].astype(np.float64).reshape(-1), t, got
    else:
        st = int(G[key + "@stride"]); ref = G[key + "@sample"].astype(np.float64); tt, gg = t[::st][:4096], got[::st][:4096]
    sc = max(1e-6, np.abs(t).max())
    rows.append((np.abs(gg - tt).max() / sc, np.abs(ref - tt).max() / sc, k))
rows.sort(reverse=True)
for e, n, k in rows[:8]: print(f"hip err {e:.2e}  reference-fp32 err {n:.2e}  {k}")
