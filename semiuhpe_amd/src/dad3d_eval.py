"""DAD-3DHeads test-set evaluation (reference eval_DAD3DHeads.py): the head crop around an annotated box and the submission
JSON of predicted rotation matrices.  Host-side data handling around the hot path; the network forward and
the SO(3) projection run through the HIP kernels in batches instead of one image at a time.
"""
import json
import os

import numpy as np
import torch
from PIL import Image

MARGIN = 0.15                       # the reference's fixed enlargement of the box for this test set (eval_DAD3DHeads.py:26)
MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)


def _axis(lo, hi, extra, limit):
    """Grow [lo, hi] by `extra` on both sides inside [0, limit]; what does not fit becomes black padding -> (lo, hi, pad_lo, pad_hi)"""
    lo2, hi2 = lo - extra, hi + extra
    pad_lo, pad_hi = max(0, -lo2), max(0, hi2 - limit)
    return max(lo2, 0), min(hi2, limit), pad_lo, pad_hi


def crop_head(image, bbox, size=224):
    """eval_DAD3DHeads.py:17-68: square box of side (w + h) / 2 enlarged by 15 % around the centre of bbox = [x, y, w, h], clipped
    to the image; the shorter side of the clipped box is then extended symmetrically to a square, parts outside the image are
    black; PIL resize (default filter) to size x size.  `image`: PIL image or path."""
    im = Image.open(image).convert("RGB") if isinstance(image, (str, os.PathLike)) else image.convert("RGB")
    width, height = im.size
    x, y, w, h = bbox
    cx, cy, side = x + w / 2, y + h / 2, (w + h) / 2.0
    half = (0.5 + MARGIN) * side
    left, right = max(int(cx - half), 0), min(int(cx + half), width - 1)
    top, bottom = max(int(cy - half), 0), min(int(cy + half), height - 1)
    bw, bh = right - left, bottom - top
    grow_x, grow_y = (int((bh - bw) / 2), 0) if bh > bw else (0, int((bw - bh) / 2))
    left, right, pad_l, pad_r = _axis(left, right, grow_x, width - 1)
    top, bottom, pad_t, pad_b = _axis(top, bottom, grow_y, height - 1)
    crop = im.crop((left, top, right, bottom))
    canvas = Image.new(crop.mode, (crop.size[0] + pad_l + pad_r, crop.size[1] + pad_t + pad_b), (0, 0, 0))
    canvas.paste(crop, (pad_l, pad_t))
    return canvas.resize([size, size])


def to_input(crops):
    """list of PIL crops -> (n, 3, H, W) float32 tensor, ToTensor + ImageNet normalisation (eval_DAD3DHeads.py:87-88)"""
    arr = np.stack([np.asarray(c, np.float32) / 255.0 for c in crops])
    return torch.from_numpy(((arr - MEAN) / STD).transpose(0, 3, 1, 2).copy())


def experiment_name(config):
    """eval_DAD3DHeads.py:103-106"""
    detail = getattr(config, "exp_detail", "")
    kind = "SemiUHPE" if "Dyna" in detail else ("Baseline" if "r0.05" in detail else "Supervised")
    return config.network + "_" + kind


def predict_rotations(agent, items, db_path, batch=64):
    """items: the entries of test/test.json -> {item_id: submission record} (eval_DAD3DHeads.py:77-101), `batch` crops per forward"""
    from .fisher.fisher_utils import batch_torch_A_to_R
    results = {}
    agent.net.eval()
    for s in range(0, len(items), batch):
        chunk = items[s:s + batch]
        crops = [crop_head(os.path.join(db_path, "test", "images", it["item_id"] + ".png"), it["bbox"]) for it in chunk]
        with torch.no_grad():
            rot = batch_torch_A_to_R(agent.net(to_input(crops).to(agent.device))).reshape(-1, 3, 3).cpu().numpy()
        for it, r in zip(chunk, rot):
            results[it["item_id"]] = {"68_landmarks_2d": [], "N_landmarks_3d": [], "7_landmarks_3d": [], "rotation_matrix": r.tolist()}
    return results


def write_submission(results, config, out_dir="exps"):
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"testset_results_{experiment_name(config)}.json")
    with open(path, "w") as fh:
        json.dump(results, fh)
    return path
