"""GPU parity of RepVGG-B1g2 and TinyViT-21M (HIP layers) against the REFERENCE's own outputs
(tests/golden/backbones.npz: eval/train outputs, gradients, BN running stats from the imported reference modules, with
weights from the shared seeded recipe).  Tolerances: outputs 3e-4 relative, gradients 2e-3 relative to the tensor max."""
import os

import numpy as np
import pytest
import torch

from parity_utils import seeded_state_dict

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "backbones.npz"))


def check(key, got, rtol=3e-4):
    got = got.detach().cpu().double().numpy()
    if key in G.files:
        ref = G[key].astype(np.float64)
        err = np.abs(got.reshape(ref.shape) - ref).max()
        assert err <= rtol * max(1e-6, np.abs(ref).max()), (key, err, np.abs(ref).max())
        return
    stride = int(G[key + "@stride"])
    ref = G[key + "@sample"].astype(np.float64)
    smp = got.reshape(-1)[::stride][:4096]
    err = np.abs(smp - ref).max()
    assert err <= rtol * max(1e-6, np.abs(got).max()), (key, err)
    sums = G[key + "@sums"]
    assert abs(np.abs(got).sum() - sums[1]) <= 2e-3 * sums[1], key


def to_logical(t, like_shape):
    """HIP filters are (Cout,Cg,R,S) logically already; activations need NHWC->NCHW"""
    return t


def run_net(model, tag, x, dy):
    from semiuhpe_amd import layers as L
    model.eval()
    with torch.no_grad():
        check(tag + "/eval_out", model(x.cuda()))
    model.train()
    for m in model.modules():
        if isinstance(m, (L.Dropout, L.StochasticDepth)) or hasattr(m, "drop_prob"):
            if hasattr(m, "p"):
                m.p = 0.0
            if hasattr(m, "drop_prob"):
                m.drop_prob = 0.0
    for p in model.parameters():
        p.grad = None
    out = model(x.cuda())
    out.backward(dy.cuda())
    check(tag + "/train_out", out)
    params = dict(model.named_parameters())
    keys = sorted({k[len(tag + "/grad/"):].split("@")[0] for k in G.files if k.startswith(tag + "/grad/")})
    for k in keys:
        check(tag + "/grad/" + k, params[k].grad, 2e-3)
    sd = model.state_dict()
    for k in G[tag + "/bn_keys"]:
        check(tag + "/bn/" + str(k), sd[str(k)])


def test_repvgg_b1g2_vs_reference():
    from semiuhpe_amd.src.repvgg import RepVggNet
    m = RepVggNet()
    assert len(m.state_dict()) == 467
    m.load_state_dict(seeded_state_dict(m.state_dict(), 11))
    m.cuda()
    run_net(m, "repvgg", torch.from_numpy(G["repvgg/x"]), torch.from_numpy(G["repvgg/dy"]))


@pytest.mark.parametrize("tag,cfg", [("blockA", (64, 128, 2, 1)), ("blockB", (128, 128, 1, 2))])
def test_repvgg_block_vs_reference(tag, cfg):
    from semiuhpe_amd.src.repvgg import RepVGGBlock
    blk = RepVGGBlock(*cfg)
    blk.load_state_dict(seeded_state_dict(blk.state_dict(), 5))
    blk.cuda().train()
    x = torch.from_numpy(G[f"repvgg_{tag}/x"]).permute(0, 2, 3, 1).contiguous().cuda()
    dy = torch.from_numpy(G[f"repvgg_{tag}/dy"]).permute(0, 2, 3, 1).contiguous().cuda()
    tape = []
    y = blk.fwd(x, tape)
    dx = blk.bwd(dy, tape)
    assert not tape
    check(f"repvgg_{tag}/y", y.permute(0, 3, 1, 2))
    check(f"repvgg_{tag}/dx", dx.permute(0, 3, 1, 2), 2e-3)
    for k, p in blk.named_parameters():
        check(f"repvgg_{tag}/grad/{k}", p.grad, 2e-3)


def test_tinyvit_21m_vs_reference():
    tv = pytest.importorskip("semiuhpe_amd.src.tiny_vit")
    m = tv.TinyViT21M()
    assert len(m.state_dict()) == 310
    m.load_state_dict(seeded_state_dict(m.state_dict(), 13))
    m.cuda()
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(int(G["tinyvit/x_seed"]))) * 0.5
    run_net(m, "tinyvit", x, torch.from_numpy(G["tinyvit/dy"]))
