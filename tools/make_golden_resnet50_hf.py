"""A third-party pin for the ResNet-50 body of oracle/networks.py (the reference takes it from torchvision, which this image
lacks): Hugging Face transformers' ResNetModel (default ResNetConfig = ResNet-50 v1.5: bottleneck layers, stride on the 3x3 conv, the
architecture its `microsoft/resnet-50` weights were converted into from the same published network) run on seeded weights.

    python tools/make_golden_resnet50_hf.py          ->  tests/golden/resnet50_hf.npz

Weights: tests/parity_utils.seeded_state_dict over the ORACLE's state-dict order (seed 31), renamed into the other implementation's keys
by `hf_key` below (the mapping is part of what is pinned).  Stored: the input, the pooled 2048-feature vector in eval mode, the same in
train mode (batch statistics), the gradient of sum(features * dy) w.r.t. the input in train mode, and the BatchNorm running means of
two layers after that step."""
import os, sys
import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import networks as ON
from parity_utils import seeded_state_dict


def hf_key(k):
    """oracle / torchvision state-dict key -> transformers.ResNetModel key (None for the regression head)"""
    parts = k.split(".")
    norm = {"bn1": 0, "bn2": 1, "bn3": 2}
    if parts[0] == "conv1":
        return "embedder.embedder.convolution." + parts[1]
    if parts[0] == "bn1":
        return "embedder.embedder.normalization." + parts[1]
    if parts[0].startswith("layer"):
        stage, blk, what = int(parts[0][5:]) - 1, parts[1], parts[2]
        base = f"encoder.stages.{stage}.layers.{blk}."
        if what == "downsample":
            return base + "shortcut." + ("convolution." if parts[3] == "0" else "normalization.") + parts[4]
        if what.startswith("conv"):
            return base + f"layer.{int(what[4:]) - 1}.convolution." + parts[3]
        return base + f"layer.{norm[what]}.normalization." + parts[3]
    return None


def build_pair(seed=31):
    from transformers import ResNetConfig, ResNetModel
    o = ON.ResNet50()
    sd = seeded_state_dict(o.state_dict(), seed)
    o.load_state_dict(sd)
    hf = ResNetModel(ResNetConfig())
    mapped = {hf_key(k): v for k, v in sd.items() if hf_key(k) is not None}
    assert set(mapped) == set(hf.state_dict()), sorted(set(mapped) ^ set(hf.state_dict()))[:6]
    hf.load_state_dict(mapped)
    return o, hf


def oracle_features(o, x):
    import torch.nn.functional as F
    x = o.maxpool(F.relu(o.bn1(o.conv1(x))))
    return o.layer4(o.layer3(o.layer2(o.layer1(x)))).mean((2, 3))


def main():
    torch.manual_seed(0)
    o, hf = build_pair()
    g = torch.Generator().manual_seed(77)
    x = torch.randn(4, 3, 96, 96, generator=g)
    dy = torch.randn(4, 2048, generator=g)
    hf.eval()
    with torch.no_grad():
        ev = hf(x).pooler_output.flatten(1)
    hf.train()
    xg = x.clone().requires_grad_(True)
    tr = hf(xg).pooler_output.flatten(1)
    (tr * dy).sum().backward()
    sd = hf.state_dict()
    out = os.path.join(ROOT, "tests", "golden", "resnet50_hf.npz")
    np.savez_compressed(out, x=x.numpy(), dy=dy.numpy(), eval_features=ev.numpy(), train_features=tr.detach().numpy(), dx=xg.grad.numpy(),
                        rm_stem=sd["embedder.embedder.normalization.running_mean"].numpy(),
                        rm_last=sd["encoder.stages.3.layers.2.layer.2.normalization.running_mean"].numpy(),
                        transformers_version=np.array(__import__("transformers").__version__))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
