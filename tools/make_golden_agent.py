"""Golden vectors of the teacher-student step by running the REFERENCE's SSLAgent (src/agent.py, read-only
/root/reference) with RepVGG-B1g2 on CPU through the import shims.

    python tools/make_golden_agent.py      # -> tests/golden/agent_step.npz

Three SSL steps (EMA alpha 0, 1/2, 2/3), B=4 labeled + 12 unlabeled 64x64 images, rotate_aug + CutOut + CutMix, CE
unsupervised loss, 300WLP rotation convention.  Weights come from the shared seeded recipe, Dropout is disabled on the
instances (p = 0) so that the step is deterministic given torch.manual_seed (which drives the CutOut/CutMix boxes).
Stored: per-step losses, entropies, mask, errors, the threshold used, strided samples of the student parameters,
the EMA teacher and the gradients after each step.
"""
import contextlib
import io
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))
import ref_shims  # noqa: E402

ref_shims.install()
from parity_utils import seeded_state_dict  # noqa: E402
from oracle.agent import synthetic_batches  # noqa: E402  (input recipe only)

with contextlib.redirect_stdout(io.StringIO()):
    import src.agent as ag  # noqa: E402  (reference)
    from src.networks import RepVggNet  # noqa: E402

SAMPLE_KEYS = ["layer0.rbr_dense.conv.weight", "layer1.0.rbr_1x1.conv.weight", "layer2.2.rbr_identity.weight",
               "layer3.5.rbr_dense.bn.bias", "layer4.0.rbr_dense.conv.weight", "linear_reg.1.weight", "linear_reg.7.weight"]


def sample(t):
    flat = t.detach().reshape(-1)
    stride = max(1, flat.numel() // 512)
    return flat[::stride][:512].numpy().copy()


def main():
    torch.set_num_threads(8)

    def factory(cfg):
        with contextlib.redirect_stdout(io.StringIO()):
            m = RepVggNet(pretrained=False, num_classes=9)
        m.load_state_dict(seeded_state_dict(m.state_dict(), 17))
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        return m

    ag.get_network = factory
    tmp = tempfile.mkdtemp()
    cfg = SimpleNamespace(lr=1e-4, log_dir=tmp, model_dir=tmp, distribution="matrixFisher", is_full_range=False, rotate_aug=True,
                          train_labeled="300WLP", train_unlabeled="300WLP", cutout_aug=True, cutmix_aug=True, conf_thres=1e9,
                          type_unsuper="ce", SSL_lambda=1.0, is_ema=True, ema_decay=0.999, eman=False, stage1_iteration=0,
                          network="repvgg", num_classes=9, exp_name="x", date="x", left_ratio=0.95, save_feat=False)
    agent = ag.SSLAgent(cfg)
    store = {"sample_keys": np.array(SAMPLE_KEYS)}
    for step in range(3):
        data, ulb = synthetic_batches(4, 12, size=64, seed=step, rotate_aug=True)
        ulb["rot_mat"] = synthetic_batches(12, 0, size=8, seed=70 + step)[0]["rot_mat"]
        agent.clock.iteration = step
        torch.manual_seed(100 + step)
        store[f"s{step}/conf_thres"] = np.float64(cfg.conf_thres)
        fd, fu, out = agent.train_func({k: v.clone() for k, v in data.items()}, {k: v.clone() for k, v in ulb.items()})
        store[f"s{step}/loss"] = fd["loss"].detach().numpy()
        store[f"s{step}/pred"] = fd["pred"].detach().numpy()
        store[f"s{step}/pred_orth"] = fd["pred_orth"].detach().numpy()
        store[f"s{step}/err_deg"] = fd["err_deg"].detach().numpy()
        store[f"s{step}/unsuper_loss"] = fu["unsuper_loss"].detach().numpy()
        store[f"s{step}/entropy"] = fu["entropy"].detach().numpy()
        store[f"s{step}/mask_ratio"] = fu["mask_ratio"].detach().numpy()
        store[f"s{step}/err_weakAll_gt"] = fu["err_weakAll_gt"].detach().numpy()
        store[f"s{step}/err_weakPseudo_gt_mean"] = fu["err_weakPseudo_gt"].detach().mean().numpy()
        store[f"s{step}/err_strongSuper_pseudo_mean"] = fu["err_strongSuper_pseudo"].detach().mean().numpy()
        store[f"s{step}/loss_all"] = out["loss_all"].detach().numpy()
        params, ema = dict(agent.net.named_parameters()), dict(agent.ema_net.named_parameters())
        for k in SAMPLE_KEYS:
            store[f"s{step}/grad/{k}"] = sample(params[k].grad)
            store[f"s{step}/param/{k}"] = sample(params[k])
            store[f"s{step}/ema/{k}"] = sample(ema[k])
        # next step: threshold = median entropy of this step, so a strict subset passes the filter
        cfg.conf_thres = float(np.median(store[f"s{step}/entropy"]))
    agent.logs_writer.close()
    np.savez_compressed(os.path.join(HERE, "..", "tests", "golden", "agent_step.npz"), **store)
    for s in range(3):
        print(s, float(store[f"s{s}/loss"]), float(store[f"s{s}/unsuper_loss"]), float(store[f"s{s}/mask_ratio"]), float(store[f"s{s}/conf_thres"]))


if __name__ == "__main__":
    main()
