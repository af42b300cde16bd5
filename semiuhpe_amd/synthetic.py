"""Synthetic "224x224 heads" batches with the reference loaders' dict contract: there are no
datasets offline, benchmarks and the smoke test feed these."""
import numpy as np
import torch

from .src.utils import get_6DRepNet_Rot


def synthetic_batches(B, muB, size=224, seed=0, full_range=False, rotate_aug=False):
    g = torch.Generator().manual_seed(1000 + seed)
    data = {"img": torch.randn(B, 3, size, size, generator=g)}
    if full_range:          # DAD3DHeads-like: random SO(3), no Euler labels (dataset_DAD3DHeads.py:150)
        q, r = torch.linalg.qr(torch.randn(B, 3, 3, generator=g, dtype=torch.float64))
        q = q * torch.sign(torch.diagonal(r, dim1=1, dim2=2))[:, None, :]
        q[:, :, 2] *= torch.det(q)[:, None]
        data["rot_mat"] = q.float().contiguous()
    else:                   # 300W-LP-like front range: pitch, roll in +-60, yaw in +-89 degrees
        ang = (torch.rand(B, 3, generator=g, dtype=torch.float64) * 2 - 1) * torch.tensor([60.0, 89.0, 60.0], dtype=torch.float64)
        R = np.stack([get_6DRepNet_Rot(*(a * np.pi / 180).tolist()) for a in ang])
        data["rot_mat"] = torch.from_numpy(R).float()
        data["euler_angles"] = ang.float()
    if muB == 0:
        return data, None
    g2 = torch.Generator().manual_seed(2000 + seed)
    ulb = {"img": torch.randn(muB, 3, size, size, generator=g2),
           "img_strong": torch.randn(muB, 3, size, size, generator=torch.Generator().manual_seed(3000 + seed))}
    deg = torch.round((torch.rand(muB, generator=g2, dtype=torch.float64) * 60 - 30) * 1000) / 1000 if rotate_aug \
        else torch.zeros(muB, dtype=torch.float64)
    rad = deg * np.pi / 180
    aug = torch.zeros(muB, 3, 3, dtype=torch.float64)
    aug[:, 0, 0], aug[:, 0, 1], aug[:, 1, 0], aug[:, 1, 1], aug[:, 2, 2] = torch.cos(rad), torch.sin(rad), -torch.sin(rad), torch.cos(rad), 1.0
    ulb["aug_rot_mat"] = aug.float()
    return data, ulb


def synthetic_head_mesh(n_lat=40, n_lon=56, seed=0):
    """A closed, head-sized triangle mesh for the demo renderer's tests and examples (the reference ships its own asset,
    pose_references/vertices_trans.npy + triangles.npy: 53 149 vertices / 105 694 triangles in [-1, 1]^3): an ellipsoid with a
    nose-like bump and seeded low-frequency ripples, as a latitude / longitude grid with two pole vertices.
    -> vertices (n, 3) float32 in [-1, 1]^3, triangles (m, 3) int32 (counter-clockwise seen from outside)."""
    import numpy as np
    rng = np.random.RandomState(seed)
    lat = np.linspace(0, np.pi, n_lat + 2)[1:-1]
    lon = np.linspace(0, 2 * np.pi, n_lon, endpoint=False)
    la, lo = np.meshgrid(lat, lon, indexing="ij")
    r = 1.0 + 0.18 * np.exp(-((la - 1.7) ** 2 + (np.angle(np.exp(1j * (lo - np.pi / 2)))) ** 2) / 0.08)       # the "nose" towards +y... +z after the swap below
    for _ in range(4):
        r += 0.03 * rng.randn() * np.sin(rng.randint(1, 4) * la + rng.rand() * 6) * np.cos(rng.randint(1, 4) * lo + rng.rand() * 6)
    x, y, z = 0.72 * r * np.sin(la) * np.cos(lo), 0.95 * r * np.cos(la), 0.78 * r * np.sin(la) * np.sin(lo)
    body = np.stack([x, y, z], -1).reshape(-1, 3)
    verts = np.concatenate([body, [[0, 0.95, 0]], [[0, -0.95, 0]]]).astype(np.float32)
    top, bottom = n_lat * n_lon, n_lat * n_lon + 1
    tris = []
    idx = lambda i, j: i * n_lon + (j % n_lon)
    for j in range(n_lon):
        tris.append((top, idx(0, j + 1), idx(0, j)))
        tris.append((bottom, idx(n_lat - 1, j), idx(n_lat - 1, j + 1)))
        for i in range(n_lat - 1):
            tris.append((idx(i, j), idx(i, j + 1), idx(i + 1, j)))
            tris.append((idx(i, j + 1), idx(i + 1, j + 1), idx(i + 1, j)))
    return verts, np.asarray(tris, dtype=np.int32)
