"""CPU oracle (test infrastructure): Euler <-> rotation helpers, error metrics, augmentations, threshold.

Restates src/utils.py:204-260, src/agent.py:357-417 (sort-and-pick part), src/agent.py:447-455,
src/augments.py:12-35,67-94 and pytorch3d-0.7.2's so3_relative_angle (unpinned third party).
"""
import math

import numpy as np
import torch


def get_6DRepNet_Rot(x, y, z):
    """R = Rz(z) Ry(y) Rx(x), right-handed, radians (src/utils.py:204-225)."""
    cx, sx, cy, sy, cz, sz = math.cos(x), math.sin(x), math.cos(y), math.sin(y), math.cos(z), math.sin(z)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def compute_euler_angles_from_rotation_matrices(R, full_range=False):
    """(n,3,3) -> (n,3) radians [pitch(x), yaw(y), roll(z)]  (src/utils.py:232-260).
    sy = sqrt(R00^2 + R10^2); singular when sy < 1e-6 (tested BEFORE the full-range sign flip)."""
    sy = torch.sqrt(R[:, 0, 0] ** 2 + R[:, 1, 0] ** 2)
    singular = (sy < 1e-6).to(R.dtype)
    if full_range:
        sy = torch.where(R[:, 0, 0] < 0, -sy, sy)
    x = torch.atan2(R[:, 2, 1], R[:, 2, 2])
    y = torch.atan2(-R[:, 2, 0], sy)
    z = torch.atan2(R[:, 1, 0], R[:, 0, 0])
    xs = torch.atan2(-R[:, 1, 2], R[:, 1, 1])
    out = torch.stack((x * (1 - singular) + xs * singular, y, z * (1 - singular)), 1)
    return out


def so3_relative_angle(R1, R2, cos_bound=1e-4, eps=1e-4):
    """acos((tr(R1 R2^T) - 1)/2) with pytorch3d's linear extrapolation of acos outside +-(1-cos_bound)."""
    tr = (R1 * R2).sum((1, 2))
    if ((tr < -1 - eps) | (tr > 3 + eps)).any():
        raise ValueError("A matrix has trace outside valid range [-1-eps,3+eps].")
    c = (tr - 1) * 0.5
    b = 1.0 - cos_bound
    slope = -1.0 / math.sqrt(1 - b * b)
    inside = torch.acos(c.clamp(-b, b))
    hi = (c - b) * slope + math.acos(b)
    lo = (c + b) * slope + math.acos(-b)
    return torch.where(c >= b, hi, torch.where(c <= -b, lo, inside))


def compute_err_deg_from_matrices(pred, gt, gt_euler=None):
    """Geodesic degrees, or mean |Euler(pred) - gt_euler| in degrees (src/agent.py:447-455)."""
    if gt_euler is None:
        return torch.rad2deg(so3_relative_angle(pred, gt))
    pd = compute_euler_angles_from_rotation_matrices(pred, full_range=False) * 180 / np.pi
    return (pd - gt_euler).abs().mean(-1)


def draw_boxes(N, width, height, holes=3, normal=True):
    """Box parameters with the reference's CPU-generator call order: normal(cx), normal(cy), randint(size)
    (src/augments.py:15-30).  Returns int32 (N,holes) x0,y0,x1,y1."""
    if not normal:
        cx = torch.randint(20, width - 20, (N, holes)).int()
        cy = torch.randint(20, height - 20, (N, holes)).int()
    else:
        cx = torch.normal(width / 2.0, width / 6.0, size=(N, holes)).int()
        cy = torch.normal(height / 2.0, height / 6.0, size=(N, holes)).int()
    size = torch.randint(10, 20, (N, holes, 2)).int()
    x0 = (cx - size[..., 0]).clamp(0, width)
    y0 = (cy - size[..., 1]).clamp(0, height)
    x1 = (cx + size[..., 0]).clamp(0, width)
    y1 = (cy + size[..., 1]).clamp(0, height)
    return x0, y0, x1, y1


def random_cutout_tensor(image, mask_holes_num=3, normal=False):
    """Zero `mask_holes_num` boxes per image in place (src/augments.py:12-35).  NB the reference names
    dim 2 'width' and dim 3 'height' but slices [y0:y1] on dim 2 and [x0:x1] on dim 3."""
    N, _, width, height = image.shape
    x0, y0, x1, y1 = draw_boxes(N, width, height, mask_holes_num, normal)
    for i in range(N):
        for j in range(mask_holes_num):
            image[i, :, y0[i, j]:y1[i, j], x0[i, j]:x1[i, j]] = 0
    return image


def random_cutmix_tensor(image, mask_holes_num=3, normal=False):
    """Paste boxes from a permuted copy of the batch (src/augments.py:67-94)."""
    N, _, width, height = image.shape
    x0, y0, x1, y1 = draw_boxes(N, width, height, mask_holes_num, normal)
    src = image[torch.randperm(N)]
    for i in range(N):
        for j in range(mask_holes_num):
            image[i, :, y0[i, j]:y1[i, j], x0[i, j]:x1[i, j]] = src[i, :, y0[i, j]:y1[i, j], x0[i, j]:x1[i, j]]
    return image


def entropy_threshold(entropies, left_ratio):
    """sorted(entropies)[int(N * left_ratio)]  (src/agent.py:403-407)."""
    e = np.sort(np.asarray(entropies, dtype=np.float32))
    return e[int(len(e) * left_ratio)]
