#!/usr/bin/env python
"""train.py — same command line and log format as the reference's train.py (two-stage loop: supervised stage 1,
teacher-student stage 2, periodic validation, best checkpoint, dynamic entropy threshold), running on the
MI355X hot path.  Single GPU: `python train.py --config settings/X.yml ...`; data parallel on one node:
`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train.py ...` (one process per GPU,
RCCL all-reduce of the flat gradient; --batch_size is the GLOBAL labeled batch and is split across ranks).
"""
import os
from datetime import datetime

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC for RCCL between the ranks' processes (set before HIP starts)

import numpy as np
import torch
import torch.distributed as dist

from semiuhpe_amd.src.agent import get_agent
from semiuhpe_amd.src.config import get_config
from semiuhpe_amd.src.datasets import build_loaders
from semiuhpe_amd.src import distributed as D
from semiuhpe_amd.src.distributed import init_from_env
from semiuhpe_amd.src.utils import cycle, dict_get


def _stamp(clock):
    return datetime.now().strftime("%Y%m%d-%H%M%S") + " {:03d} {:09d}".format(clock.epoch, clock.iteration)


def _log(agent, text):
    agent.logs_writer.write(text)
    agent.logs_writer.flush()


def validate(agent, test_loader, s1, eval_ema, tag):
    """One pass over the test set -> mean test error (reference train.py:163-239); writes the `test*/...` log lines.
    With several ranks every rank takes its share of the batches and the per-sample results are pooled, so all ranks see the
    statistics of the whole set (and agree on the best checkpoint)."""
    losses, errs, ratios, pseudo = [], [], [], []
    for data in D.shard(test_loader):
        if s1:
            fisher_dict = agent.val_func_s1(data)
        else:
            fisher_dict, _, out = agent.val_func(data, eval_ema=eval_ema)
            ratios.append(out["mask_ratio"])
            if out["err_pseudo_gt"] is not None:
                pseudo.append(out["err_pseudo_gt"].detach().cpu().numpy())
        losses.append(fisher_dict["loss"].item())
        errs.append(fisher_dict["err_deg"].detach().cpu().numpy())
    dev = agent.device if D.world()[1] > 1 else "cpu"
    errs, losses = D.gather_values(errs, dev), D.gather_values(losses, dev)
    if not s1:
        ratios, pseudo = D.gather_values([float(r) for r in ratios], dev), ([D.gather_values(pseudo, dev)] if D.world()[1] > 1 else pseudo)
        pseudo = [p for p in pseudo if len(p)]
    _log(agent, _stamp(agent.clock) + " {0}/loss:{1:.6f} {0}/err_mean:{2:.6f}\n".format(tag, np.mean(losses), np.mean(errs)))
    if not s1:
        pseudo = [-1] if len(pseudo) == 0 else np.concatenate(pseudo, 0)
        _log(agent, _stamp(agent.clock) + " {0}/mask_ratio:{1:.6f} {0}/err_pseudo_gt:{2:.6f}\n".format(tag, np.mean(ratios), np.mean(pseudo)))
    return float(np.mean(errs))


def main():
    rank, world = init_from_env(os.environ.get("SUHPE_DIST_BACKEND", "nccl"))      # (gloo: the multi-rank flow rehearsed on one GPU, tests)
    config = get_config("train")
    train_loader, ulb_train_loader, test_loader = build_loaders(config, rank, world)
    ssl = config.stage1_iteration < config.max_iteration
    iter_ulb = cycle(ulb_train_loader) if ssl else None

    agent = get_agent(config)
    if config.cont:
        agent.load_ckpt(config.ckpt)
        agent.clock.tock()
        for group in agent.optimizer.param_groups:
            group["lr"] = config.lr
    clock = agent.clock
    best_mean_error = 360
    done = False
    while not done:
        if hasattr(getattr(train_loader, "sampler", None), "set_epoch"):
            train_loader.sampler.set_epoch(clock.epoch)
        for data in train_loader:
            if clock.iteration == config.stage1_iteration and ssl:      # stage switch: best stage-1 weights, lr x0.1
                stage1_clock = clock.make_checkpoint()
                if world > 1:
                    dist.barrier()
                agent.load_ckpt("best")
                clock.restore_checkpoint(stage1_clock)
                for group in agent.optimizer.param_groups:
                    group["lr"] *= 0.1
            if clock.iteration < config.stage1_iteration:
                s1 = True
                fisher_dict = agent.train_func_s1(data)
            elif not ssl:
                done = True
                break
            else:
                s1 = False
                fisher_dict, fisher_dict_unsuper, out_dict = agent.train_func(data, next(iter_ulb))

            if clock.iteration % config.log_frequency == 0:             # the only host syncs of the training loop
                agent.check_finite()
                vals = [fisher_dict["loss"].reshape(1), fisher_dict["err_deg"].mean().reshape(1)]
                if not s1:
                    u = fisher_dict_unsuper
                    ent = dict_get(u, "entropy", -1)
                    vals += [dict_get(u, "unsuper_loss", -1).reshape(1), ent.mean().reshape(1), ent.std().reshape(1),
                             dict_get(u, "mask_ratio", -1).reshape(1), dict_get(u, "err_weakAll_gt", -1).mean().reshape(1),
                             dict_get(u, "err_weakPseudo_gt", -1).mean().reshape(1), dict_get(u, "err_strongSuper_pseudo", -1).mean().reshape(1)]
                # one small all-reduce: the logged scalars are means over all ranks' shards
                v = D.allreduce_avg_(torch.cat([t.detach().float() for t in vals])).tolist()
                _log(agent, _stamp(clock) + " train/lr:{:.6f} train/loss:{:.6f} train/err_mean:{:.6f}\n".format(
                    agent.optimizer.param_groups[0]["lr"], v[0], v[1]))
                if not s1:
                    _log(agent, _stamp(clock) + " train_SSL/unsuper_loss:{:.6f} train_SSL/entropy_mean:{:.6f} train_SSL/entropy_std:{:.6f}"
                         " train_SSL/mask_ratio:{:.6f}\n".format(v[2], v[3], v[4], v[5]))
                    _log(agent, _stamp(clock) + " train_SSL/err_weakAll_gt:{:.6f} train_SSL/err_weakPseudo_gt:{:.6f}"
                         " train_SSL/err_strongSuper_pseudo:{:.6f}\n".format(v[6], v[7], v[8]))
            clock.tick()

            if clock.iteration % config.val_frequency == 0:
                err = validate(agent, test_loader, s1, False, "test")
                if err < best_mean_error:
                    best_mean_error = err
                    agent.save_ckpt("best")
                if not s1:
                    validate(agent, test_loader, s1, True, "test_ema")

            if clock.iteration % config.save_frequency == 0 and config.dynamic_thres and clock.iteration >= config.stage1_iteration and ssl:
                config.conf_thres = agent.compute_dynamic_entropy_threshold(ulb_train_loader)
                _log(agent, _stamp(clock) + " settings/conf_thres:{:.6f}\n".format(config.conf_thres))
            if clock.iteration >= config.max_iteration:
                done = True
                break
        clock.tock()
    agent.logs_writer.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
