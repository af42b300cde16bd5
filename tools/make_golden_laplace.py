"""G8: golden vectors of the Rotation-Laplace head by importing the reference (src/laplace/rotation_laplace.py).

    python tools/make_golden_laplace.py      # -> tests/golden/laplace.npz (NLL values, dNLL/dA, analytical mode)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
from src.laplace import rotation_laplace as RL  # noqa: E402  (reference)


def rand_rot(n, g):
    q, r = torch.linalg.qr(torch.randn(n, 3, 3, generator=g, dtype=torch.float64))
    q = q * torch.sign(torch.diagonal(r, dim1=1, dim2=2))[:, None, :]
    q[:, :, 2] *= torch.det(q)[:, None]
    return q.contiguous()


def main():
    grids = torch.from_numpy(np.load(os.path.join(ref_shims.REFERENCE_ROOT, "src", "laplace", "eq_grids2.npy")))
    store, names = {}, []
    for seed, scale in ((0, 0.5), (1, 3.0), (2, 15.0), (3, 60.0)):
        g = torch.Generator().manual_seed(seed)
        A = (torch.randn(16, 9, generator=g, dtype=torch.float64) * scale).float()
        if seed == 2:        # "trained-like": concentrated around a rotation
            Rt = rand_rot(16, g)
            A = (scale * Rt + 0.2 * scale * torch.randn(16, 3, 3, generator=g, dtype=torch.float64)).reshape(16, 9).float()
        R = rand_rot(16, g).float()
        a = A.clone().requires_grad_(True)
        loss, mode = RL.NLL_loss("RLaplace", a, R, grids)
        (dA,) = torch.autograd.grad(loss.sum(), a)
        name = f"s{seed}_x{scale:g}"
        names.append(name)
        store[name + "/A"], store[name + "/R"] = A.numpy(), R.numpy()
        store[name + "/nll"], store[name + "/dA"], store[name + "/mode"] = loss.detach().numpy(), dA.numpy(), mode.detach().numpy()
    store["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "..", "tests", "golden", "laplace.npz"), **store)
    print("laplace.npz:", names)


if __name__ == "__main__":
    main()
