"""Drop-in for the reference's src/config.py (configargparse) on argparse + PyYAML: same `--config` YAML keys, same
CLI flags (config.py:88-163 of the reference), same derived `exp_detail` string and exps/<exp_name>/<exp_detail>/<date>/
layout.  Additive flags (not in the reference): --synthetic (no datasets offline), --exp_root, --no_autotune and --sync_bn.
"""
import argparse
import os
import sys
from datetime import datetime
from os.path import abspath, dirname, join

import yaml

from .utils import ensure_dirs


def str2type(s):
    """reference config.py:182-191"""
    low = str(s).lower()
    if low == "true":
        return True
    if low == "false":
        return False
    if low == "none":
        return None
    return s


# (flag, type or 'flag', default) in the reference's groups
_OPTIONS = [
    # basic
    ("exp_name", str2type, None), ("exp_detail", str, None), ("ss_ratio", float, None), ("is_full_range", "flag", False),
    # dataset
    ("data_dir_300WLP", str, None), ("data_dir_AFLWFace", str, None), ("data_dir_AFLW2000", str, None),
    ("data_dir_BIWItrain", str, None), ("data_dir_BIWItest", str, None), ("data_dir_WiderFace", str, None),
    ("data_dir_CrowdHuman", str, None), ("data_dir_DAD3DHeads", str, None), ("data_dir_COCOHead", str, None),
    ("data_dir_WildHead", str, None), ("train_labeled", str, None), ("train_unlabeled", str, None), ("test_set", str, None),
    # network
    ("network", str, None),
    # training
    ("lr", float, None), ("batch_size", int, None), ("num_workers", int, None), ("stage1_iteration", int, None),
    ("max_iteration", int, None), ("log_frequency", int, None), ("val_frequency", int, None), ("save_frequency", int, None),
    ("cont_ckpt", str, None), ("debug", "flag", False),
    # ssl
    ("SSL_lambda", float, None), ("conf_thres", float, None), ("dynamic_thres", "flag", False), ("std_ratio", float, 3.0),
    ("left_ratio", float, 0.95), ("ulb_batch_ratio", float, None), ("is_ema", str2type, None), ("ema_decay", float, None),
    ("eman", "flag", False), ("type_unsuper", str, None), ("distribution", str, None), ("cutout_aug", "flag", False),
    ("cutmix_aug", "flag", False), ("rotate_aug", "flag", False), ("save_feat", "flag", False),
]
_NETWORKS = ["mobilenet", "resnet18", "resnet50", "repvgg", "effinetv2", "effinet", "tinyvit"]


def _shared_run_date():
    """The run's time stamp -- the last component of log_dir / model_dir -- must be ONE value for all ranks of a job (only rank
    0 writes best.pth; the others load it at the stage switch): SUHPE_RUN_DATE from the launcher if set, else rank 0's clock
    broadcast over the process group (train.py initialises it before building the config)."""
    date = os.environ.get("SUHPE_RUN_DATE") or datetime.now().strftime("%b%d_%H%M%S")
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            box = [date]
            dist.broadcast_object_list(box, src=0)
            date = box[0]
    except ImportError:
        pass
    return date


class Config(object):
    def __init__(self, phase, argv=None):
        self.is_train = phase == "train"
        args = self.parse(argv)
        self.num_classes = 9
        print("----Experiment Configuration-----")
        for k, v in sorted(args.items()):
            setattr(self, k, v)
            print(f"{k:20}: {v}")
        self.cont = self.cont_ckpt is not None
        if self.is_train:
            if self.debug:
                self.exp_detail, self.date = "debug", "debug"
            elif self.cont:
                self.exp_detail, self.date, self.ckpt = self.cont_ckpt.split("/")
            else:
                self.exp_detail = self.get_exp_detail()
                self.date = _shared_run_date()
        else:
            self.exp_detail, self.date, self.ckpt = self.test_ckpt.split("/")
        print(f"exp name: {self.exp_name}, {self.exp_detail}")
        proj_root = self.exp_root or join(dirname(dirname(dirname(abspath(__file__)))), "exps")
        self.log_dir = join(proj_root, str(self.exp_name), self.exp_detail, self.date)
        self.model_dir = self.log_dir
        if not self.is_train or self.cont:
            assert os.path.exists(self.log_dir), f"Log dir {self.log_dir} does not exist"
        else:
            ensure_dirs([self.log_dir, self.model_dir])
        if self.is_train and int(os.environ.get("RANK", "0")) == 0:
            name = f"log_cont_{datetime.now().strftime('%b%d_%H%M%S')}.txt" if self.cont else "log.txt"
            with open(join(self.log_dir, name), "w") as log:       # the reference also dumps its sources here
                for k, v in sorted(self.__dict__.items()):
                    log.write(f"{k:20}: {v}\n")
        if self.gpu_ids is not None and "LOCAL_RANK" not in os.environ:
            os.environ["CUDA_VISIBLE_DEVICES"] = str(self.gpu_ids)     # HIP honours CUDA_VISIBLE_DEVICES under torch

    def parse(self, argv=None):
        ap = argparse.ArgumentParser()
        ap.add_argument("--config", type=str, help="config file path")
        for name, typ, default in _OPTIONS:
            if typ == "flag":
                ap.add_argument("--" + name, action="store_true")
            elif name == "network":
                ap.add_argument("--network", type=str, choices=_NETWORKS)
            elif name == "type_unsuper":
                ap.add_argument("--type_unsuper", type=str, choices=["ce", "nll"])
            elif name == "distribution":
                ap.add_argument("--distribution", type=str, choices=["matrixFisher", "RotationLaplace"])
            else:
                ap.add_argument("--" + name, type=typ)
        ap.add_argument("-g", "--gpu_ids", type=str)
        ap.add_argument("--synthetic", action="store_true", help="synthetic 224x224 heads instead of dataset loaders")
        ap.add_argument("--exp_root", type=str, default=None, help="where exps/ lives (default: repo root)")
        ap.add_argument("--no_autotune", action="store_true", help="keep the built-in GEMM tile heuristics (no first-step timing)")
        ap.add_argument("--sync_bn", action="store_true", help="data parallel: BatchNorm statistics over the global batch (default: per rank)")
        if not self.is_train:
            ap.add_argument("test_ckpt", type=str)
            ap.add_argument("--hist_low", type=int, default=10)
            ap.add_argument("--hist_high", type=int, default=150)
        ns = vars(ap.parse_args(argv))
        merged = {name: default for name, _, default in _OPTIONS}
        merged.update(gpu_ids=None, synthetic=False, exp_root=None)
        if ns.get("config"):
            with open(ns["config"]) as fh:
                y = yaml.safe_load(fh) or {}
            types = {n: t for n, t, _ in _OPTIONS}
            for k, v in y.items():
                t = types.get(k)
                if t == "flag":
                    v = bool(str2type(v))
                elif t in (int, float) and v is not None:
                    v = t(v)                      # YAML reads 1e-4 as a string
                elif t is str2type:
                    v = str2type(v)
                merged[k] = v
        for k, v in ns.items():                   # the command line overrides the YAML
            if k == "config":
                continue
            is_flag = any(n == k and t == "flag" for n, t, _ in _OPTIONS) or k in ("synthetic", "no_autotune", "sync_bn")
            if (is_flag and v) or (not is_flag and v is not None):
                merged[k] = v
            elif k not in merged:
                merged[k] = v
        merged["autotune"] = not merged.pop("no_autotune", False)
        return merged

    def get_exp_detail(self):
        """reference config.py:165-180"""
        if self.exp_detail is not None:
            return self.exp_detail
        name_thre = "Dyna" + str(self.left_ratio) if self.dynamic_thres else str(self.conf_thres)
        name_ema = "_eman" if self.eman else ("_ema" if self.is_ema else "")
        name_range = "_full" if self.is_full_range else ""
        name_aug = ("_RO" if self.rotate_aug else "") + ("_CO" if self.cutout_aug else "") + ("_CM" if self.cutmix_aug else "")
        name_vis = "_tSNE" if self.save_feat else ""
        return (f"SSL{self.SSL_lambda}_r{self.ss_ratio}_{self.type_unsuper}_{self.network}"
                f"_t{name_thre}_b{self.batch_size}{name_ema}{name_aug}{name_range}{name_vis}")


def get_config(phase, argv=None):
    return Config(phase, argv)
