"""A third-party pin for the MBConv block of the EfficientNetV2-S oracle (oracle/networks.py: MBConv; SURVEY 8 a9 -- the reference takes the
network from torchvision, src/networks.py:301-332, which this image lacks): Hugging Face transformers' EfficientNetBlock (the EfficientNet-v1
block: 1x1 expand + BN + SiLU, 3x3 depthwise + BN + SiLU, squeeze-excitation with cin/4 squeeze channels, 1x1 project + BN, skip) on seeded
weights, stride 1 (its stride-2 blocks pad the TensorFlow way, torchvision pads symmetrically: not comparable).  What stays restated from the
published torchvision source only: FusedMBConv, the V2-S stage table, row-mode stochastic depth, the stem / head convs.

    python tools/make_golden_mbconv_hf.py          ->  tests/golden/mbconv_hf.npz

Weights: tests/parity_utils.seeded_state_dict over the oracle block's state dict, copied position by position (both trees list expand conv, BN,
depthwise conv, BN, SE reduce, SE expand, project conv, BN in that order with equal shapes -- asserted)."""
import os, sys
import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import networks as ON
from parity_utils import seeded_state_dict

CASES = {"s4": (128, 4, 7, 3, 41), "s5": (160, 6, 7, 2, 42), "s6": (256, 6, 4, 3, 43)}      # cin = cout, expand, map, batch, seed


def build_pair(cin, expand, seed):
    from transformers import EfficientNetConfig
    from transformers.models.efficientnet.modeling_efficientnet import EfficientNetBlock
    o = ON.MBConv(expand, 3, 1, cin, cin, 0.0)
    sd = seeded_state_dict(o.state_dict(), seed)
    o.load_state_dict(sd)
    cfg = EfficientNetConfig(batch_norm_eps=1e-3, batch_norm_momentum=0.1, squeeze_expansion_ratio=0.25, hidden_act="swish")
    hf = EfficientNetBlock(cfg, cin, cin, 1, expand, 3, 0.0, False, False)          # id_skip=False is its "add the skip" setting
    theirs = hf.state_dict()
    assert [tuple(v.shape) for v in sd.values()] == [tuple(v.shape) for v in theirs.values()]
    hf.load_state_dict(dict(zip(theirs.keys(), sd.values())))
    return o, hf


def sample(t):
    flat = t.reshape(-1)
    return flat[::max(1, flat.numel() // 2048)][:2048].clone()


def run(mod, x, dy):
    mod.eval()
    with torch.no_grad():
        ev = mod(x)
    mod.train()
    xg = x.clone().requires_grad_(True)
    tr = mod(xg)
    for p in mod.parameters():
        p.grad = None
    (tr * dy).sum().backward()
    grads = [p.grad.clone() for p in mod.parameters()]
    stats = [v.clone() for k, v in mod.state_dict().items() if k.endswith("running_var")]
    return ev, tr.detach(), xg.grad, grads, stats


def main():
    store = {"transformers_version": np.array(__import__("transformers").__version__)}
    for tag, (cin, expand, hw, B, seed) in CASES.items():
        o, hf = build_pair(cin, expand, seed)
        g = torch.Generator().manual_seed(seed + 100)
        x, dy = torch.randn(B, cin, hw, hw, generator=g), torch.randn(B, cin, hw, hw, generator=g)
        ev, tr, dx, grads, stats = run(hf, x, dy)
        store.update({f"{tag}/x": x.numpy(), f"{tag}/dy": dy.numpy(), f"{tag}/eval": ev.numpy(), f"{tag}/train": tr.numpy(), f"{tag}/dx": dx.numpy()})
        for i, gr in enumerate(grads):                                      # parameter gradients: a strided sample of <= 2048 values + the L1 norm
            store[f"{tag}/grad{i}"] = sample(gr).numpy()
            store[f"{tag}/grad{i}_l1"] = np.array(float(gr.abs().sum()))
        for i, st in enumerate(stats):
            store[f"{tag}/running_var{i}"] = st.numpy()
    out = os.path.join(ROOT, "tests", "golden", "mbconv_hf.npz")
    np.savez_compressed(out, **store)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
