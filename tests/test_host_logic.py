"""CPU tests of the host-side mirror of the reference interface: config surface, exp_detail naming, settings files,
batch-dict contract of the synthetic loaders, EMA alpha schedule, TrainClock."""
import os

import numpy as np
import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_settings_files_carry_the_reference_keys():
    keys = {"gpu_ids", "exp_name", "is_full_range", "train_labeled", "train_unlabeled", "test_set", "ss_ratio", "network",
            "SSL_lambda", "conf_thres", "is_ema", "ema_decay", "type_unsuper", "distribution", "lr", "batch_size", "num_workers",
            "ulb_batch_ratio", "stage1_iteration", "max_iteration", "log_frequency", "val_frequency", "save_frequency"}
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "settings")) if f.endswith(".yml"))
    assert len(files) == 13
    for f in files:
        y = yaml.safe_load(open(os.path.join(ROOT, "settings", f)))
        assert keys <= set(y), (f, keys - set(y))
        assert y["exp_name"] == f[:-4].replace("_RL", "") or y["exp_name"] == f[:-4]
        assert float(y["lr"]) == 1e-4 and y["conf_thres"] == -5.3


def test_config_cli_overrides_yaml_and_exp_detail(tmp_path):
    from semiuhpe_amd.src.config import get_config
    argv = ["--config", os.path.join(ROOT, "settings", "DAD3DHeads_WildHead.yml"), "--network", "effinetv2", "--batch_size", "64",
            "--ulb_batch_ratio", "4", "--is_full_range", "--dynamic_thres", "--cutout_aug", "--cutmix_aug", "--rotate_aug",
            "--left_ratio", "0.75", "--exp_root", str(tmp_path)]
    cfg = get_config("train", argv)
    # name of the released checkpoint's experiment in the reference README
    assert cfg.exp_detail == "SSL1_r0.05_ce_effinetv2_tDyna0.75_b64_ema_RO_CO_CM_full" or \
        cfg.exp_detail == "SSL1.0_r0.05_ce_effinetv2_tDyna0.75_b64_ema_RO_CO_CM_full"
    assert cfg.num_classes == 9 and cfg.is_ema is True and cfg.lr == 1e-4 and cfg.train_labeled == "DAD3DHeads"
    assert os.path.isdir(cfg.log_dir) and os.path.exists(os.path.join(cfg.log_dir, "log.txt"))
    cfg2 = get_config("test", ["--config", os.path.join(ROOT, "settings", "DAD3DHeads_WildHead.yml"), "--exp_root", str(tmp_path),
                               f"{cfg.exp_detail}/{cfg.date}/best"])
    assert cfg2.ckpt == "best" and cfg2.log_dir == cfg.log_dir


def test_synthetic_loader_batch_contract():
    from types import SimpleNamespace

    from semiuhpe_amd.src.datasets import build_loaders
    os.environ["SUHPE_SYNTH_SIZE"] = "32"
    os.environ["SUHPE_SYNTH_BATCHES"] = "2"
    try:
        cfg = SimpleNamespace(exp_name="300WLP_AFLW2000", is_full_range=False, synthetic=True, batch_size=8, ulb_batch_ratio=4,
                              stage1_iteration=1, max_iteration=2, rotate_aug=True)
        train, ulb, test = build_loaders(cfg)
        d = next(iter(train))
        assert set(d) == {"img", "rot_mat", "euler_angles"} and d["img"].shape == (8, 3, 32, 32) and d["rot_mat"].shape == (8, 3, 3)
        R = d["rot_mat"].double()
        assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3, dtype=torch.float64).expand(8, 3, 3), atol=1e-6)
        u = next(iter(ulb))
        assert set(u) == {"img", "img_strong", "aug_rot_mat"} and u["img"].shape[0] == 32
        ang = torch.rad2deg(torch.atan2(u["aug_rot_mat"][:, 0, 1], u["aug_rot_mat"][:, 0, 0]))
        assert float(ang.abs().max()) <= 30.0 and float(ang.abs().max()) > 0
        cfg.exp_name, cfg.is_full_range = "DAD3DHeads_WildHead", True
        d = next(iter(build_loaders(cfg)[0]))
        assert "euler_angles" not in d
        # data-parallel sharding: per-rank batch = global / world, different seeds per rank
        t0 = next(iter(build_loaders(cfg, rank=0, world=2)[0]))
        t1 = next(iter(build_loaders(cfg, rank=1, world=2)[0]))
        assert t0["img"].shape[0] == 4 and not torch.equal(t0["img"], t1["img"])
    finally:
        os.environ.pop("SUHPE_SYNTH_SIZE"); os.environ.pop("SUHPE_SYNTH_BATCHES")


def test_clock_and_ema_schedule():
    from semiuhpe_amd.src.utils import TrainClock
    c = TrainClock()
    for _ in range(3):
        c.tick()
    c.tock()
    ck = c.make_checkpoint()
    assert ck == {"epoch": 1, "minibatch": 0, "iteration": 3}
    c2 = TrainClock(); c2.restore_checkpoint(ck)
    assert c2.iteration == 3
    # alpha = min(1 - 1/(t+1), decay): 0, 1/2, 2/3, ... then the decay (reference agent.py:277-283)
    alphas = [min(1 - 1 / (t + 1), 0.999) for t in (0, 1, 2, 5000)]
    assert alphas[0] == 0 and abs(alphas[1] - 0.5) < 1e-12 and abs(alphas[2] - 2 / 3) < 1e-12 and alphas[3] == 0.999


def test_count_work_reproduces_the_survey_totals():
    """tools/count_work.py: MACs per image-forward, parameter counts and the step totals the roofline uses."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import count_work as cw
    want = {"effinetv2": (2.849e9, 20_901_465, 110), "repvgg": (9.787e9, 44_850_569, 56), "tinyvit": (4.250e9, 20_704_649, 27)}
    r50 = cw.NETS["resnet50"]().totals()
    assert r50["params"] == 24_625_225 and r50["convs"] == 53 and abs(r50["macs"] - 4.09e9) < 0.03e9      # torchvision: 4.09 GFLOPS
    for net, (macs, params, convs) in want.items():
        t = cw.NETS[net]().totals()
        assert abs(t["macs"] - macs) < 1e6 and t["params"] == params and t["convs"] == convs, (net, t)
    assert abs(cw.NETS["tinyvit"]().totals()["attn_macs"] - 0.212e9) < 1e6
    c2 = cw.step_work("effinetv2", 32, 128)
    assert abs(c2["flop"] - 3.46e12) < 0.01e12 and abs(c2["t_flop"] - 22.0e-3) < 0.1e-3
    assert abs(cw.step_work("repvgg", 64, 256)["flop"] - 23.8e12) < 0.05e12
    assert abs(cw.step_work("tinyvit", 32, 128)["flop"] - 5.17e12) < 0.01e12


def test_shard_loader_splits_batch_and_samples():
    """data parallel over the reference's loaders: per-rank batch = global / world, disjoint shards, new order per epoch"""
    import torch
    from torch.utils.data import DataLoader, TensorDataset

    from semiuhpe_amd.src.datasets import shard_loader
    from semiuhpe_amd.src.utils import cycle
    ds = TensorDataset(torch.arange(64))
    base = DataLoader(ds, batch_size=8, shuffle=True, drop_last=True)
    assert shard_loader(base, 0, 1) is base
    shards = [shard_loader(base, r, 2, seed=3) for r in range(2)]
    seen = []
    for ld in shards:
        assert ld.batch_size == 4 and ld.drop_last
        batches = [b[0] for b in ld]
        assert len(batches) == 8 and all(len(b) == 4 for b in batches)
        seen.append(torch.cat(batches))
    assert sorted(torch.cat(seen).tolist()) == list(range(64))          # disjoint and complete
    it = cycle(shards[0])
    first = torch.cat([next(it)[0] for _ in range(8)])
    second = torch.cat([next(it)[0] for _ in range(8)])
    assert not torch.equal(first, second)                                # set_epoch advanced by cycle
    import pytest
    with pytest.raises(ValueError):
        shard_loader(DataLoader(ds, batch_size=9), 0, 2)


def test_build_loaders_uses_importable_reference_style_loaders(tmp_path, monkeypatch):
    """the reference's `src.datasets.dataset_<NAME>.get_dataloader_<NAME>(phase, config)` functions are used
    unchanged when importable.  A stand-in package with that interface (map-style datasets emitting the batch-dict keys of
    SURVEY 8b; NOT the reference's code) is put on sys.path: build_loaders selects the loaders train.py:47-76 selects for the
    experiment name, shards the two training loaders per rank and leaves the test loader whole."""
    import importlib, sys, textwrap
    from types import SimpleNamespace
    pkg = tmp_path / "src" / "datasets"
    pkg.mkdir(parents=True)
    (tmp_path / "src" / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text("")
    for name in ("300WLP", "AFLW2000", "DAD3DHeads", "COCOHead", "WildHead"):
        (pkg / f"dataset_{name}.py").write_text(textwrap.dedent(f"""
            import torch
            from torch.utils.data import Dataset, DataLoader
            CALLS = []
            class _Heads(Dataset):
                def __init__(self, n, phase): self.n, self.phase = n, phase
                def __len__(self): return self.n
                def __getitem__(self, i):
                    return {{"idx": i, "img": torch.full((3, 8, 8), float(i)), "img_strong": torch.zeros(3, 8, 8), "rot_mat": torch.eye(3),
                            "euler_angles": torch.zeros(3), "aug_rot_mat": torch.eye(3), "src": "{name}:" + self.phase}}
            def get_dataloader_{name}(phase, config):
                CALLS.append(phase)
                train = "train" in phase
                bs = config.batch_size if phase in ("train", "train_all") else (round(config.batch_size * config.ulb_batch_ratio) if train else 4)
                return DataLoader(_Heads(96, phase), batch_size=bs, shuffle=train, drop_last=train)
            """))
    for m in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        monkeypatch.delitem(sys.modules, m)
    monkeypatch.syspath_prepend(str(tmp_path))
    from semiuhpe_amd.src.datasets import build_loaders
    cfg = SimpleNamespace(exp_name="300WLP_WildHead", is_full_range=False, synthetic=False, batch_size=8, ulb_batch_ratio=2, stage1_iteration=0,
                          max_iteration=10, rotate_aug=True)
    train, ulb, test = build_loaders(cfg, rank=1, world=2)
    assert importlib.import_module("src.datasets.dataset_300WLP").CALLS == ["train_all"]
    assert importlib.import_module("src.datasets.dataset_WildHead").CALLS == ["ulb_train"]
    assert importlib.import_module("src.datasets.dataset_AFLW2000").CALLS == ["test"]
    b, u, t = next(iter(train)), next(iter(ulb)), next(iter(test))
    assert b["img"].shape == (4, 3, 8, 8) and u["img"].shape == (8, 3, 8, 8) and t["img"].shape[0] == 4      # per-rank shares of 8 and 16; test whole
    assert b["src"][0] == "300WLP:train_all" and u["src"][0] == "WildHead:ulb_train" and t["src"][0] == "AFLW2000:test"
    other = next(iter(build_loaders(cfg, rank=0, world=2)[0]))
    train.sampler.set_epoch(0)
    seen = {int(i) for bb in train for i in bb["idx"]}
    assert len(seen) == 48                                            # this rank's half of the 96 samples, no duplicates
    assert set(b.keys()) == set(other.keys())
    # without the package the error says what to do
    for m in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
        monkeypatch.delitem(sys.modules, m)
    monkeypatch.setattr(sys, "path", [p for p in sys.path if p != str(tmp_path)])
    import pytest as _pt
    try:
        importlib.import_module("src.datasets.dataset_300WLP")
        importable = True
    except ImportError:
        importable = False
    if not importable:
        with _pt.raises(ImportError, match="--synthetic"):
            build_loaders(cfg)
