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


def test_submission_records_from_a_stub_agent(tmp_path):
    """predict_rotations / write_submission: file layout in, JSON layout out (reference eval_DAD3DHeads.py:77-108), with a stub
    network so that the host logic is covered without a GPU."""
    import json
    import torch
    from make_golden_dad3d_crop import synthetic_image
    from semiuhpe_amd.src import dad3d_eval as E
    (tmp_path / "test" / "images").mkdir(parents=True)
    items = []
    for k in range(5):
        Image.fromarray(synthetic_image(k, 200, 160)).save(tmp_path / "test" / "images" / f"id{k}.png")
        items.append({"item_id": f"id{k}", "bbox": [30 + k, 20, 90, 100], "attributes": {}})
    eye = torch.eye(3).reshape(1, 9)

    class Net(torch.nn.Module):
        def forward(self, x):
            assert tuple(x.shape[1:]) == (3, 224, 224)
            return eye.repeat(x.shape[0], 1) * (1.0 + x.mean((1, 2, 3)).reshape(-1, 1).abs())

    agent = SimpleNamespace(net=Net(), device=torch.device("cpu"))
    import semiuhpe_amd.src.fisher.fisher_utils as FU
    saved = FU.batch_torch_A_to_R
    FU.batch_torch_A_to_R = lambda A: A / A[:, :1]                      # scaled identities -> identities (no HIP kernel on the CPU)
    try:
        res = E.predict_rotations(agent, items, str(tmp_path), batch=2)
    finally:
        FU.batch_torch_A_to_R = saved
    assert list(res) == [f"id{k}" for k in range(5)]
    for rec in res.values():
        assert set(rec) == {"68_landmarks_2d", "N_landmarks_3d", "7_landmarks_3d", "rotation_matrix"}
        assert np.allclose(rec["rotation_matrix"], np.eye(3))
    path = E.write_submission(res, SimpleNamespace(network="effinetv2", exp_detail="SSL1.0_r0.05_ce_effinetv2_tDyna0.75_b64"), str(tmp_path / "exps"))
    assert path.endswith("testset_results_effinetv2_SemiUHPE.json") and json.load(open(path)).keys() == res.keys()
