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
