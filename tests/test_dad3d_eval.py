"""Head crop of the DAD-3DHeads evaluation against crops produced by the reference's own `process_ori_img`
(tools/make_golden_dad3d_crop.py -> tests/golden/dad3d_crop.npz): bit-exact uint8 pixels; plus the submission record format."""
import hashlib
import os
import sys
from types import SimpleNamespace

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "dad3d_crop.npz"))


def test_head_crop_matches_the_reference_bit_for_bit(tmp_path):
    from make_golden_dad3d_crop import synthetic_image
    from semiuhpe_amd.src.dad3d_eval import crop_head
    for k, ((seed, w, h), box) in enumerate(zip(G["cases"], G["boxes"])):
        path = tmp_path / f"{k}.png"
        Image.fromarray(synthetic_image(int(seed), int(w), int(h))).save(path)
        crop = np.asarray(crop_head(str(path), [float(v) for v in box]))
        assert crop.shape == (224, 224, 3) and crop.dtype == np.uint8
        assert hashlib.sha256(crop.tobytes()).hexdigest() == str(G["sha256"][k]), (k, box)
        if f"crop{k}" in G:
            assert np.array_equal(crop, G[f"crop{k}"])


def test_input_tensor_and_experiment_name():
    from semiuhpe_amd.src.dad3d_eval import experiment_name, to_input
    crops = [Image.fromarray(np.full((224, 224, 3), v, np.uint8)) for v in (0, 255)]
    t = to_input(crops)
    assert tuple(t.shape) == (2, 3, 224, 224)
    assert np.allclose(t[0, :, 0, 0].numpy(), (0 - np.array([0.485, 0.456, 0.406])) / np.array([0.229, 0.224, 0.225]), atol=1e-6)
    assert np.allclose(t[1, :, 5, 7].numpy(), (1 - np.array([0.485, 0.456, 0.406])) / np.array([0.229, 0.224, 0.225]), atol=1e-6)
    name = lambda d: experiment_name(SimpleNamespace(network="repvgg", exp_detail=d))
    assert name("SSL1.0_r0.05_ce_tDyna0.75_b32") == "repvgg_SemiUHPE" and name("SSL1.0_r0.05_ce_t-4.3") == "repvgg_Baseline"
    assert name("SSL1.0_r1.0_ce_t-5.3") == "repvgg_Supervised"
