"""Device-side replacement of the reference's per-sample image augmentation:
`process_annotated_image` / `process_not_annotated_image` (src/datasets/dataset_300WLP.py:103-161,
dataset_WildHead.py:76-124) and the sample dict built around them (`Dataset_300WLP.__getitem__`, :174-219), for a
whole batch of decoded 224x224 RGB crops that already sit in HBM.

The random decisions are drawn on the host exactly as the reference draws them per sample (numpy global RNG for flip /
blur / angle, torch global RNG inside torchvision's RandomResizedCrop.get_params, in that order), the pixels never leave
the GPU: `suhpe_augment_views` reproduces Pillow's flip / BLUR / rotate(NEAREST, expand) / crop / resize(BILINEAR)
bit for bit and writes the normalised fp32 views in the backbone's pixel-major layout.  JPEG decode and the loose head
crop (`process_ori_mat_anno`) stay on the CPU loaders.
"""
import ctypes
import math

import numpy as np
import torch

from .. import _lib
from .._lib import call, ptr, workspace

SIZE = 224
_tables = {}


def random_resized_crop_params(height, width, scale, ratio=(1.0, 1.0)):
    """torchvision 0.14 `RandomResizedCrop.get_params` (torch global RNG, same draw order): (top, left, h, w)"""
    area = height * width
    log_ratio = torch.log(torch.tensor(ratio))
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        aspect_ratio = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if 0 < w <= width and 0 < h <= height:
            i = torch.randint(0, height - h + 1, size=(1,)).item()
            j = torch.randint(0, width - w + 1, size=(1,)).item()
            return i, j, h, w
    in_ratio = float(width) / float(height)             # fallback: central crop clamped to the ratio range
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def rotate_coefficients(angle, w=SIZE, h=SIZE, center=(112, 112)):
    """PIL `Image.rotate(angle, center=center, expand=True)` as libImaging's 16.16 fixed-point stepping, plus the origin
    of the reference's centre 224 crop in the expanded canvas (dataset_300WLP.py:130-134): 8 integers per image."""
    a = -math.radians(angle % 360.0)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]

    def tf(x, y):
        return m[0] * x + m[1] * y + m[2], m[3] * x + m[4] * y + m[5]
    m[2], m[5] = tf(-center[0], -center[1])
    m[2] += center[0]
    m[5] += center[1]
    corners = [tf(x, y) for x, y in ((0, 0), (w, 0), (w, h), (0, h))]
    nw = math.ceil(max(c[0] for c in corners)) - math.floor(min(c[0] for c in corners))
    nh = math.ceil(max(c[1] for c in corners)) - math.floor(min(c[1] for c in corners))
    m[2], m[5] = tf(-(nw - w) / 2.0, -(nh - h) / 2.0)
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    return (fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5),
            nw // 2 - SIZE // 2, nh // 2 - SIZE // 2)


def draw_params(augment_strong, rotate_aug):
    """One sample's random decisions in the reference's order (process_annotated_image :111-143)."""
    p = dict(flip=bool(np.random.uniform(0, 1) < 0.5), blur=bool(np.random.uniform(0, 1) < 0.05))
    p["weak"] = random_resized_crop_params(SIZE, SIZE, (0.8, 1.25))
    p["angle"] = 0.0
    p["strong"] = None
    if augment_strong:
        if rotate_aug:
            p["angle"] = round(np.random.rand() * 60 - 30, 3)
        p["strong"] = random_resized_crop_params(SIZE, SIZE, (0.6, 1.5))
    return p


def _device_tables(device):
    key = torch.device(device)
    if key not in _tables:
        lib = _lib.load()
        n = lib.suhpe_augment_tables_bytes() // 4
        host = torch.empty(n, dtype=torch.int32)
        rc = lib.suhpe_augment_build_tables(ctypes.c_void_p(host.data_ptr()))
        if rc != 0:
            raise RuntimeError("suhpe_augment_build_tables failed")
        _tables[key] = host.to(key)
    return _tables[key]


def pack_params(params, device, rotate_aug=True):
    """Host decisions of a batch -> the small device arrays of suhpe_augment_views: (flags, weak_box, strong_box, rot_coef)"""
    strong = params[0]["strong"] is not None
    flags = torch.tensor([int(p["flip"]) | (int(p["blur"]) << 1) for p in params], dtype=torch.int32).to(device, non_blocking=True)
    wbox = torch.tensor([p["weak"] for p in params], dtype=torch.int32).to(device, non_blocking=True)
    sbox = coef = None
    if strong:
        sbox = torch.tensor([p["strong"] for p in params], dtype=torch.int32).to(device, non_blocking=True)
        if rotate_aug:
            coef = torch.tensor([rotate_coefficients(p["angle"]) for p in params], dtype=torch.int64).to(device, non_blocking=True)
    return flags, wbox, sbox, coef


def augment_views_packed(images_u8, packed, return_u8=False):
    """The device part alone: images (N, 224, 224, 3) uint8 + pack_params(...) -> normalised fp32 views, pixel-major."""
    assert images_u8.is_cuda and images_u8.dtype == torch.uint8 and images_u8.is_contiguous() and tuple(images_u8.shape[1:]) == (SIZE, SIZE, 3)
    flags, wbox, sbox, coef = packed
    n, dev = images_u8.shape[0], images_u8.device
    assert flags.numel() == n
    img = torch.empty((n, SIZE, SIZE, 3), dtype=torch.float32, device=dev)
    img_s = torch.empty_like(img) if sbox is not None else None
    u8w = torch.empty((n, SIZE, SIZE, 3), dtype=torch.uint8, device=dev) if return_u8 else None
    u8s = torch.empty_like(u8w) if (return_u8 and sbox is not None) else None
    ws = workspace("augment", _lib.load().suhpe_augment_workspace_bytes(n), dev)
    call("suhpe_augment_views", ptr(images_u8), n, ptr(flags), ptr(wbox), ptr(coef), ptr(sbox), ptr(_device_tables(dev)), ptr(ws), ptr(img),
         ptr(img_s), ptr(u8w), ptr(u8s))
    return (img, img_s, u8w, u8s) if return_u8 else (img, img_s)


def augment_views(images_u8, params, rotate_aug=True, return_u8=False):
    """images_u8: (N, 224, 224, 3) uint8 CUDA tensor; params: list of draw_params dicts.
    -> (img, img_strong or None[, weak_u8, strong_u8]) with img (N, 224, 224, 3) fp32 normalised, pixel-major."""
    assert len(params) == images_u8.shape[0]
    return augment_views_packed(images_u8, pack_params(params, images_u8.device, rotate_aug), return_u8)


def get_6DRepNet_Rot(x, y, z):
    """src/utils.py:204-225 (radians): R = Rz(z) Ry(y) Rx(x)"""
    rx = np.array([[1, 0, 0], [0, np.cos(x), -np.sin(x)], [0, np.sin(x), np.cos(x)]])
    ry = np.array([[np.cos(y), 0, np.sin(y)], [0, 1, 0], [-np.sin(y), 0, np.cos(y)]])
    rz = np.array([[np.cos(z), -np.sin(z), 0], [np.sin(z), np.cos(z), 0], [0, 0, 1]])
    return rz.dot(ry.dot(rx))


def process_annotated_batch(images_u8, yaw, pitch, roll, augment_strong=False, config=None):
    """Batched `Dataset_300WLP.__getitem__` tail (:189-219) for decoded crops on the GPU and labels in degrees.
    Returns the loaders' batch dict: img, img_strong (zeros without the strong view), rot_mat (= aug_rot_mat @ R),
    euler_angles (pitch, yaw, roll after the flip), aug_rot_mat — images pixel-major (N, 224, 224, 3)."""
    rotate_aug = bool(getattr(config, "rotate_aug", False))
    n = images_u8.shape[0]
    params, eulers, rots, augs = [], [], [], []
    for s in range(n):
        y, p_, r = (float(v) * np.pi / 180.0 for v in (yaw[s], pitch[s], roll[s]))
        prm = draw_params(augment_strong, rotate_aug)
        if prm["flip"]:
            y, r = -y, -r
        params.append(prm)
        eulers.append([p_ * 180.0 / np.pi, y * 180.0 / np.pi, r * 180.0 / np.pi])
        ra = prm["angle"] * np.pi / 180.0
        aug = np.array([[np.cos(ra), np.sin(ra), 0], [-np.sin(ra), np.cos(ra), 0], [0, 0, 1]])
        augs.append(aug)
        rots.append(torch.FloatTensor(aug) @ torch.FloatTensor(get_6DRepNet_Rot(p_, y, r)))
    img, img_s = augment_views(images_u8, params, rotate_aug)
    if img_s is None:
        img_s = torch.zeros_like(img)
    dev = images_u8.device
    return dict(img=img, img_strong=img_s, rot_mat=torch.stack(rots).to(dev), euler_angles=torch.FloatTensor(eulers).to(dev),
                aug_rot_mat=torch.FloatTensor(np.stack(augs)).to(dev))
