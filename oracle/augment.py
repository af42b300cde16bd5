"""CPU restatement (numpy) of the reference's per-sample image augmentation for a 224x224 RGB crop — test infrastructure
only (see oracle/__init__.py): the checker of the device-side augmentation kernels.

Follows src/datasets/dataset_300WLP.py:103-161 (`process_annotated_image`; `dataset_WildHead.py:76-124` is the same
pipeline without labels) and the third-party image operators it calls, restated from their published algorithms:
  * PIL 9-12 `Image.transpose(FLIP_LEFT_RIGHT)`, `ImageFilter.BLUR` (libImaging/Filter.c 5x5), `Image.rotate(angle,
    center, expand=True)` (Image.py rotate -> libImaging/Geometry.c affine_fixed, NEAREST), `Image.crop`,
    `Image.resize(BILINEAR)` (libImaging/Resample.c: separable triangle filter scaled by the reduction factor,
    22-bit fixed-point coefficients, uint8 intermediate);
  * torchvision 0.14 `RandomResizedCrop.get_params` / `resized_crop` / `ToTensor` / `Normalize`.
PIL is importable in the build container, so every pixel operator here is pinned bit for bit against PIL itself
(tests/test_oracle_augment.py, tools/make_golden_augment.py); torchvision is not, its two dozen lines of parameter
logic are restated (RNG call order included) — "parity unpinned" for that part only.
"""
import math

import numpy as np
import torch

MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)
PRECISION_BITS = 32 - 8 - 2


def flip_lr(img):
    return np.ascontiguousarray(img[:, ::-1])


def blur(img):
    """ImageFilter.BLUR: 5x5 ring-of-ones kernel / 16, the two border rows / columns are copied."""
    h, w, _ = img.shape
    out = img.copy()
    k = np.ones((5, 5), np.float32)
    k[1:4, 1:4] = 0.0
    k /= 16.0
    acc = np.zeros((h - 4, w - 4, 3), np.float32)
    for dy in range(5):
        for dx in range(5):
            if k[dy, dx] != 0.0:
                acc += img[dy:dy + h - 4, dx:dx + w - 4].astype(np.float32) * k[dy, dx]
    out[2:h - 2, 2:w - 2] = np.clip(np.floor(acc + 0.5), 0, 255).astype(np.uint8)
    return out


def _rotate_matrix(angle, w, h, center):
    """Image.rotate's affine matrix (output -> input) and expanded size (PIL/Image.py rotate, expand=True)."""
    angle = angle % 360.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]

    def tf(x, y, mm):
        return mm[0] * x + mm[1] * y + mm[2], mm[3] * x + mm[4] * y + mm[5]
    m[2], m[5] = tf(-center[0], -center[1], m)
    m[2] += center[0]
    m[5] += center[1]
    xx, yy = [], []
    for x, y in ((0, 0), (w, 0), (w, h), (0, h)):
        tx, ty = tf(x, y, m)
        xx.append(tx)
        yy.append(ty)
    nw = math.ceil(max(xx)) - math.floor(min(xx))
    nh = math.ceil(max(yy)) - math.floor(min(yy))
    m[2], m[5] = tf(-(nw - w) / 2.0, -(nh - h) / 2.0, m)
    return m, nw, nh


def rotate_fixed_coeffs(angle, w=224, h=224, center=(112, 112)):
    """16.16 fixed-point stepping of libImaging affine_fixed plus the centre-crop origin of the reference (:130-134):
    returns (a0, a1, a2, a3, a4, a5, x0, y0) so that output pixel (x, y) of the 224x224 crop reads input
    (xin, yin) = ((a2 + a0*(x+x0) + a1*(y+y0)) >> 16, (a5 + a3*(x+x0) + a4*(y+y0)) >> 16), black when outside."""
    m, nw, nh = _rotate_matrix(angle, w, h, center)
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    a0, a1, a3, a4 = fix(m[0]), fix(m[1]), fix(m[3]), fix(m[4])
    a2 = fix(m[2] + m[0] * 0.5 + m[1] * 0.5)
    a5 = fix(m[5] + m[3] * 0.5 + m[4] * 0.5)
    return a0, a1, a2, a3, a4, a5, nw // 2 - 112, nh // 2 - 112


def rotate_center_crop(img, angle):
    """im.rotate(angle, center=(112,112), expand=True) then the centre 224x224 crop (NEAREST, black fill)."""
    h, w, _ = img.shape
    a0, a1, a2, a3, a4, a5, x0, y0 = rotate_fixed_coeffs(angle, w, h)
    ys, xs = np.meshgrid(np.arange(224, dtype=np.int64) + y0, np.arange(224, dtype=np.int64) + x0, indexing="ij")
    xin = (a2 + a0 * xs + a1 * ys) >> 16
    yin = (a5 + a3 * xs + a4 * ys) >> 16
    ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    out = np.zeros((224, 224, 3), np.uint8)
    out[ok] = img[yin[ok], xin[ok]]
    return out


def resample_coeffs(in_size, in0, in1, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the triangle (BILINEAR) filter:
    -> (bounds [out][2] = (first input index, count), integer coefficients [out][ksize])."""
    scale = (in1 - in0) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = np.zeros(ksize, np.float64)
        for x in range(xmax):
            t = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - t if t < 1.0 else 0.0
        ww = w[:xmax].sum()
        if ww != 0.0:
            w[:xmax] /= ww
        for x in range(ksize):
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(0.5 + v) if w[x] >= 0 else int(-0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis(img, bounds, kk, axis):
    out_size = bounds.shape[0]
    src = np.moveaxis(img.astype(np.int64), axis, 0)
    out = np.empty((out_size,) + src.shape[1:], np.int64)
    for o in range(out_size):
        x0, n = bounds[o]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += src[x0 + x] * kk[o, x]
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def resized_crop(img, i, j, h, w, size=224):
    """torchvision F.resized_crop on a PIL image: crop (top i, left j, h, w) then Image.resize((size, size), BILINEAR):
    horizontal pass, uint8 intermediate, vertical pass; a crop that already has the target size is returned as is."""
    crop = img[i:i + h, j:j + w]
    if h == size and w == size:
        return crop.copy()
    bx, kx = resample_coeffs(w, 0, w, size)
    by, ky = resample_coeffs(h, 0, h, size)
    tmp = _resample_axis(crop, bx, kx, 1) if w != size else crop
    return _resample_axis(tmp, by, ky, 0) if h != size else tmp


def random_resized_crop_params(height, width, scale, ratio=(1.0, 1.0)):
    """torchvision 0.14 RandomResizedCrop.get_params: draws from torch's global RNG in the same order."""
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
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def to_tensor_normalize(img):
    """ToTensor + Normalize(ImageNet): (3, H, W) float32"""
    x = img.astype(np.float32) / np.float32(255.0)
    x = (x - MEAN) / STD
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def draw_params(augment_strong, rotate_aug):
    """The random decisions of process_annotated_image in the reference's order (numpy global RNG for flip / blur / angle,
    torch global RNG inside RandomResizedCrop): dict(flip, blur, weak=(i,j,h,w), angle, strong=(i,j,h,w) or None)."""
    p = dict(flip=bool(np.random.uniform(0, 1) < 0.5), blur=bool(np.random.uniform(0, 1) < 0.05))
    p["weak"] = random_resized_crop_params(224, 224, (0.8, 1.25))
    p["angle"] = 0.0
    p["strong"] = None
    if augment_strong:
        if rotate_aug:
            p["angle"] = round(np.random.rand() * 60 - 30, 3)
        p["strong"] = random_resized_crop_params(224, 224, (0.6, 1.5))
    return p


def apply_params(img, p, rotate_aug=True):
    """-> (weak uint8, strong uint8 or None) for a 224x224x3 uint8 crop and the decisions of draw_params"""
    im = flip_lr(img) if p["flip"] else img
    if p["blur"]:
        im = blur(im)
    weak = resized_crop(im, *p["weak"])
    strong = None
    if p["strong"] is not None:
        if rotate_aug:
            im = rotate_center_crop(im, p["angle"])
        strong = resized_crop(im, *p["strong"])
    return weak, strong


def rot6d_label(pitch, yaw, roll):
    """src/utils.py get_6DRepNet_Rot (radians): R = Rz(roll) Ry(yaw) Rx(pitch) in the reference's convention"""
    rx = np.array([[1, 0, 0], [0, np.cos(pitch), -np.sin(pitch)], [0, np.sin(pitch), np.cos(pitch)]])
    ry = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
    rz = np.array([[np.cos(roll), -np.sin(roll), 0], [np.sin(roll), np.cos(roll), 0], [0, 0, 1]])
    return rz.dot(ry.dot(rx))
