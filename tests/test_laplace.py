"""Rotation-Laplace head: oracle vs the reference's outputs (CPU) and the HIP kernel vs both (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import laplace as OL

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "laplace.npz"))
GRID = os.path.join(os.path.dirname(__file__), "..", "semiuhpe_amd", "src", "laplace", "eq_grids2.npy")
NAMES = [str(n) for n in G["names"]]


def rel(x, ref):
    x = x.detach().cpu().double().numpy()
    return np.abs(x - ref).max() / max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference(name):
    grids = torch.from_numpy(np.load(GRID))
    a = torch.from_numpy(G[name + "/A"]).requires_grad_(True)
    loss, mode = OL.NLL_loss("RLaplace", a, torch.from_numpy(G[name + "/R"]), grids)
    (dA,) = torch.autograd.grad(loss.sum(), a)
    assert rel(loss, G[name + "/nll"]) < 2e-5 and rel(dA, G[name + "/dA"]) < 3e-4 and rel(mode, G[name + "/mode"]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_matches_reference(name):
    from semiuhpe_amd.src.laplace.rotation_laplace import NLL_loss, default_grids
    grids = default_grids("cuda")
    a = torch.from_numpy(G[name + "/A"]).cuda().requires_grad_(True)
    loss, mode = NLL_loss("RLaplace", a, torch.from_numpy(G[name + "/R"]).cuda(), grids)
    loss.sum().backward()
    # the kernel works in fp64; the goldens are the reference's fp32 results, whose 1/p terms near the mode of a
    # concentrated distribution carry ~3e-4 absolute noise (the fp32 oracle above shows the same spread)
    assert rel(loss, G[name + "/nll"]) < 5e-5 and rel(a.grad, G[name + "/dA"]) < 1e-3 and rel(mode, G[name + "/mode"]) < 1e-5
