"""debug: TinyViT-21M train-mode forward/backward, HIP vs fp32 oracle vs fp64 oracle (same weights, masks, upstream gradient),
as one batch and as two batch segments (oracle: two passes)"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import networks as ON
from parity_utils import MaskTape, seeded_state_dict
from semiuhpe_amd import layers as L

from semiuhpe_amd.src.networks import get_network
from types import SimpleNamespace

torch.manual_seed(0)
ref = ON.TinyViT21M()
sd = seeded_state_dict(ref.state_dict(), 23)
ref.load_state_dict(sd)
hip = get_network(SimpleNamespace(network="tinyvit", num_classes=9))
hip.load_state_dict(sd); hip.cuda()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 9
seg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = torch.Generator().manual_seed(5)
x = torch.randn(n, 3, 224, 224, generator=g)
dy = torch.randn(n, 9, generator=g)

def oracle_run(net, dtype):
    net = copy.deepcopy(net).to(dtype).train()
    tape = MaskTape(77)
    ON.set_mask_hook(lambda k, s, p: tape.record(k, s, p).to(dtype))
    for p in net.parameters():
        p.grad = None
    if seg:
        ya = net(x[:seg].to(dtype)); yb = net(x[seg:].to(dtype))
        y = torch.cat([ya, yb])
    else:
        y = net(x.to(dtype))
    y.backward(dy.to(dtype))
    ON.set_mask_hook(None)
    return y.detach(), {k: p.grad.clone() for k, p in net.named_parameters()}, tape

y64, g64, _ = oracle_run(ref, torch.float64)
y32, g32, tape = oracle_run(ref, torch.float32)
hip.train()
if seg:
    k2 = len(tape.masks) // 2
    cur = {"i": 0}
    def replay(kind, shape, keep):
        (ka, sa, ma), (kb, sb, mb) = tape.masks[cur["i"]], tape.masks[k2 + cur["i"]]
        assert ka == kind and tuple(shape) == (sa[0] + sb[0],) + sa[1:], (kind, shape, sa, sb)
        cur["i"] += 1
        return torch.cat([ma, mb])
    L.set_mask_hook(replay)
else:
    L.set_mask_hook(tape.replay)
for p in hip.parameters():
    p.grad = None
with L.batch_segments(seg if seg else None):
    y = hip(x.cuda())
y.backward(dy.cuda())
L.set_mask_hook(None)
rel = lambda a, b: float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
print(f"n={n} seg={seg}: y hip-vs-64 {rel(y, y64):.2e}  y32-vs-64 {rel(y32, y64):.2e}")
gs = max(float(v.abs().max()) for v in g64.values())
rows = []
for k, p in hip.named_parameters():
    t = g64[k]; sc = max(float(t.abs().max()), 1e-4 * gs)
    rows.append((k, float((p.grad.double().cpu() - t).abs().max()) / sc, float((g32[k].double() - t).abs().max()) / sc))
worst = sorted(rows, key=lambda r: -r[1])[:12]
for k, e, nn in worst:
    print(f"  {k:55s} hip {e:.2e}  oracle32 {nn:.2e}")
import statistics
print("median hip err", statistics.median(r[1] for r in rows), "median oracle32 err", statistics.median(r[2] for r in rows))
for k, e, nn in rows:
    if k.startswith("head") or k.startswith("norm_head"):
        print(f"  {k:55s} hip {e:.2e}  oracle32 {nn:.2e}")
