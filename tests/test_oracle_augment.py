"""Data-side augmentation, CPU part: the numpy oracle against Pillow itself (bit for bit, PIL ships in the
image) and against committed golden views; the C ABI's host-side resize tables against the oracle's; the host mirror's
random decisions against the reference function run under a torchvision shim (tests/golden/augment_views.npz)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import augment as A

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "augment_views.npz"))


def _sha(a):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def _images():
    rng = np.random.default_rng(0)
    y, x = np.mgrid[0:224, 0:224]
    smooth = np.stack([(np.sin(x / 13.0 + k) + np.cos(y / 7.0 - k)) * 60 + 128 for k in range(3)], -1)
    return [rng.integers(0, 256, (224, 224, 3), dtype=np.uint8), np.clip(smooth + rng.normal(0, 8, smooth.shape), 0, 255).astype(np.uint8)]


def test_operators_match_pillow_bit_for_bit():
    PIL = pytest.importorskip("PIL")
    from PIL import Image, ImageFilter
    for im in _images():
        pil = Image.fromarray(im)
        assert np.array_equal(A.flip_lr(im), np.asarray(pil.transpose(Image.FLIP_LEFT_RIGHT)))
        assert np.array_equal(A.blur(im), np.asarray(pil.filter(ImageFilter.BLUR)))
        for ang in (0.0, 7.123, -29.999, 30.0, -0.001, 29.344, -17.25):
            r = pil.rotate(ang, center=(112, 112), expand=True)
            x0, y0 = r.size[0] // 2 - 112, r.size[1] // 2 - 112
            assert np.array_equal(A.rotate_center_crop(im, ang), np.asarray(r.crop((x0, y0, x0 + 224, y0 + 224)))), ang
        for (i, j, h, w) in ((0, 0, 224, 224), (3, 5, 200, 200), (0, 17, 180, 180), (20, 20, 201, 201), (0, 0, 223, 223), (5, 9, 174, 174)):
            ref = np.asarray(pil.crop((j, i, j + w, i + h)).resize((224, 224), Image.BILINEAR))
            assert np.array_equal(A.resized_crop(im, i, j, h, w), ref), (i, j, h, w)


def test_golden_views_from_the_reference_function():
    """tools/make_golden_augment.py ran the reference's process_annotated_image (Pillow for the pixels, a torchvision shim
    for RandomResizedCrop / ToTensor / Normalize) on seeded inputs: the oracle must reproduce its decisions and views."""
    n = int(GOLD["n"])
    for s in range(n):
        np.random.seed(int(GOLD["np_seed"][s]))
        torch.manual_seed(int(GOLD["torch_seed"][s]))
        p = A.draw_params(True, True)
        assert p["flip"] == bool(GOLD["flip"][s]) and p["blur"] == bool(GOLD["blur"][s])
        assert abs(p["angle"] - float(GOLD["angle"][s])) < 1e-12
        weak, strong = A.apply_params(GOLD["image"][int(GOLD["image_id"][s])], p)
        assert np.array_equal(_sha(weak), GOLD["weak_sha"][s]) and np.array_equal(_sha(strong), GOLD["strong_sha"][s]), s
        assert np.array_equal(_sha(A.to_tensor_normalize(weak)), GOLD["weak_tensor_sha"][s]), s
        if s in list(GOLD["full_ids"]):                 # two cases are kept in full (the others as SHA-256 digests)
            k = list(GOLD["full_ids"]).index(s)
            assert np.array_equal(weak, GOLD["weak"][k]) and np.array_equal(strong, GOLD["strong"][k])
    assert np.array_equal(A.to_tensor_normalize(GOLD["weak"][0]), GOLD["weak_tensor"][0])


def test_product_host_logic_matches_oracle():
    """semiuhpe_amd.src.device_augment draws the same decisions and the same rotation coefficients as the oracle"""
    from semiuhpe_amd.src import device_augment as D
    for seed in range(20):
        np.random.seed(seed); torch.manual_seed(seed)
        a = A.draw_params(True, True)
        np.random.seed(seed); torch.manual_seed(seed)
        b = D.draw_params(True, True)
        assert a == b
        assert D.rotate_coefficients(a["angle"]) == A.rotate_fixed_coeffs(a["angle"])
    np.random.seed(3); torch.manual_seed(3)
    assert D.draw_params(False, True)["strong"] is None


def test_cabi_resize_tables_match_pillow_coefficients():
    """suhpe_augment_build_tables (host code of the library) == Resample.c coefficients as restated by the oracle"""
    from semiuhpe_amd import _lib
    lib = _lib.load()
    n = lib.suhpe_augment_tables_bytes() // 4
    buf = np.zeros(n, np.int32)
    assert lib.suhpe_augment_build_tables(ctypes.c_void_p(buf.ctypes.data)) == 0
    t = buf.reshape(225, 224, 5)
    for size in (1, 2, 100, 173, 174, 199, 200, 223, 224):
        bounds, kk = A.resample_coeffs(size, 0, size, 224)
        assert kk.shape[1] == 3
        assert np.array_equal(t[size, :, 0:2], bounds)
        assert np.array_equal(t[size, :, 2:5], kk)
