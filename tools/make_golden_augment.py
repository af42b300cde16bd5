"""Golden views of the reference's `process_annotated_image` (src/datasets/dataset_300WLP.py:103-161) -> tests/golden/augment_views.npz.

The reference module imports cv2, scipy.io and torchvision.transforms; cv2 and torchvision are absent here, so they are
shimmed: cv2 by an empty module (unused by this function), `torchvision.transforms` by the five classes the function uses,
restated from torchvision 0.14 on top of Pillow (which IS installed, so every pixel comes from the real PIL code):
Compose, Pad(0) (identity), RandomResizedCrop (get_params + PIL crop + PIL resize BILINEAR), ToTensor, Normalize.
Nothing here is product code."""
import math, os, sys, types
import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"


class Compose:
    def __init__(self, ts): self.ts = ts
    def __call__(self, im):
        for t in self.ts: im = t(im)
        return im


class Pad:
    def __init__(self, padding, padding_mode="constant"): self.padding = padding
    def __call__(self, im):
        assert tuple(self.padding) == (0, 0)
        return im


class RandomResizedCrop:
    def __init__(self, size, scale, ratio): self.size, self.scale, self.ratio = size, scale, ratio
    def __call__(self, im):
        width, height = im.size
        area = height * width
        log_ratio = torch.log(torch.tensor(self.ratio))
        box = None
        for _ in range(10):
            target_area = area * torch.empty(1).uniform_(self.scale[0], self.scale[1]).item()
            aspect_ratio = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
            w = int(round(math.sqrt(target_area * aspect_ratio))); h = int(round(math.sqrt(target_area / aspect_ratio)))
            if 0 < w <= width and 0 < h <= height:
                i = torch.randint(0, height - h + 1, size=(1,)).item(); j = torch.randint(0, width - w + 1, size=(1,)).item()
                box = (i, j, h, w); break
        if box is None:
            box = (0, 0, height, width)                   # ratio (1, 1) on a square image: the whole image
        i, j, h, w = box
        return im.crop((j, i, j + w, i + h)).resize(self.size[::-1], Image.BILINEAR)


class ToTensor:
    def __call__(self, im):
        return torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)


class Normalize:
    def __init__(self, mean, std): self.mean, self.std = torch.tensor(mean).view(3, 1, 1), torch.tensor(std).view(3, 1, 1)
    def __call__(self, t): return (t - self.mean) / self.std


def install():
    tv = types.ModuleType("torchvision"); tfs = types.ModuleType("torchvision.transforms")
    for c in (Compose, Pad, RandomResizedCrop, ToTensor, Normalize): setattr(tfs, c.__name__, c)
    tv.transforms = tfs
    sys.modules["torchvision"] = tv; sys.modules["torchvision.transforms"] = tfs
    sys.modules["cv2"] = types.ModuleType("cv2")
    sys.path.insert(0, REF)


if __name__ == "__main__":
    install()
    from src.datasets.dataset_300WLP import process_annotated_image
    cfg = types.SimpleNamespace(rotate_aug=True)
    rng = np.random.default_rng(7)
    y, x = np.mgrid[0:224, 0:224]
    images = [rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)]
    smooth = np.stack([(np.sin(x / 11.0 + k) + np.cos(y / 9.0 - k)) * 55 + 120 for k in range(3)], -1)
    images.append(np.clip(smooth + rng.normal(0, 6, smooth.shape), 0, 255).astype(np.uint8))
    out = dict(image=np.stack(images), image_id=[], np_seed=[], torch_seed=[], flip=[], blur=[], angle=[], weak=[], strong=[], weak_tensor=[],
               euler=[], rot=[])
    seeds = list(range(40))
    picked = 0
    for s in seeds:
        for img_id in (0, 1):
            np.random.seed(s); torch.manual_seed(1000 + s)
            st = np.random.get_state(); u0 = np.random.uniform(0, 1); u1 = np.random.uniform(0, 1); np.random.set_state(st)
            if picked >= 10 and u1 >= 0.05:
                continue                                   # after ten ordinary cases keep only the rare blurred ones
            np.random.seed(s); torch.manual_seed(1000 + s)
            w, wt, st_, stt, eul, rot, ang = process_annotated_image(Image.fromarray(images[img_id]), 10.0, -20.0, 5.0, augment_strong=True, config=cfg)
            out["image_id"].append(img_id); out["np_seed"].append(s); out["torch_seed"].append(1000 + s)
            out["flip"].append(u0 < 0.5); out["blur"].append(u1 < 0.05); out["angle"].append(ang)
            out["weak"].append(np.asarray(w)); out["strong"].append(np.asarray(st_)); out["weak_tensor"].append(wt.numpy())
            out["euler"].append(eul.numpy()); out["rot"].append(rot.numpy())
            picked += 1
    out["n"] = picked
    # keep the fixture small: full views for two cases (one of them blurred), SHA-256 digests of the views for all
    import hashlib
    sha = lambda a: np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)
    out["weak_sha"] = [sha(a) for a in out["weak"]]
    out["strong_sha"] = [sha(a) for a in out["strong"]]
    out["weak_tensor_sha"] = [sha(a) for a in out["weak_tensor"]]
    full = [0, int(np.argmax(out["blur"]))]
    out["full_ids"] = full
    out["weak"] = [out["weak"][i] for i in full]
    out["strong"] = [out["strong"][i] for i in full]
    out["weak_tensor"] = [out["weak_tensor"][full[0]]]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "augment_views.npz"), **{k: np.asarray(v) for k, v in out.items()})
    print("cases", picked, "blurred", int(np.sum(out["blur"])), "flipped", int(np.sum(out["flip"])))
