"""Golden head crops of the reference's `process_ori_img` (eval_DAD3DHeads.py:17-68) -> tests/golden/dad3d_crop.npz.

The reference script imports cv2 / torchvision / configargparse at module level (absent here) and runs its evaluation under
`__main__`, so only the one function is taken: its definition is located with `ast` in the reference file and executed with
PIL's `Image` in scope (Pillow IS installed, every pixel comes from the real PIL code).  Inputs are synthetic PNGs written to
a temporary directory; the fixture stores the input images' seeds / sizes / boxes, a SHA-256 of every 224x224 crop and two
crops in full.  Nothing here is product code."""
import ast, hashlib, os, sys, tempfile
import numpy as np
from PIL import Image

REF = "/root/reference/eval_DAD3DHeads.py"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_function():
    tree = ast.parse(open(REF).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "process_ori_img")
    ns = {"Image": Image}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), REF, "exec"), ns)
    return ns["process_ori_img"]


CASES = [  # (seed, width, height, [x, y, w, h])
    (0, 640, 480, [200, 120, 150, 180]), (1, 640, 480, [0, 0, 120, 90]), (2, 640, 480, [520, 380, 119, 99]), (3, 300, 500, [40, 30, 220, 400]),
    (4, 500, 300, [30, 40, 400, 220]), (5, 224, 224, [0, 0, 224, 224]), (6, 1000, 800, [400.5, 300.25, 61.5, 80.75]), (7, 320, 240, [-10, -5, 100, 120]),
    (8, 320, 240, [250, 180, 100, 100]), (9, 800, 600, [100, 50, 600, 500]), (10, 97, 131, [10, 20, 50, 41]), (11, 640, 480, [300, 200, 8, 8]),
]


def synthetic_image(seed, w, h):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 255 // max(1, w - 1)), (yy * 255 // max(1, h - 1)), ((xx + yy) % 256)], -1).astype(np.int32)
    return np.clip(base + rng.integers(-20, 21, size=(h, w, 3)), 0, 255).astype(np.uint8)


def main():
    fn = reference_function()
    out = {"cases": np.array([[c[0], c[1], c[2]] for c in CASES], np.int64), "boxes": np.array([c[3] for c in CASES], np.float64)}
    digests = []
    with tempfile.TemporaryDirectory() as d:
        for k, (seed, w, h, box) in enumerate(CASES):
            path = os.path.join(d, f"{k}.png")
            Image.fromarray(synthetic_image(seed, w, h)).save(path)
            crop = np.asarray(fn(path, box))
            assert crop.shape == (224, 224, 3) and crop.dtype == np.uint8
            digests.append(hashlib.sha256(crop.tobytes()).hexdigest())
            if k in (0, 7):
                out[f"crop{k}"] = crop
    out["sha256"] = np.array(digests)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dad3d_crop.npz"), **out)
    print("wrote tests/golden/dad3d_crop.npz", len(digests), "cases")


if __name__ == "__main__":
    main()
