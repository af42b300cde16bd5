#!/usr/bin/env python
"""eval.py — the reference's offline evaluation (eval.py:20-142): student then EMA teacher; Euler MAE
(pitch / yaw / roll / mean) for Euler-labelled test sets, geodesic degrees + ||I - R_pd R_gt^T||_F otherwise.

    python eval.py --config settings/300WLP_AFLW2000.yml --network effinetv2 <exp_detail>/<date>/best [--synthetic]
"""
import json
import os

import numpy as np
import torch

from semiuhpe_amd.src.agent import get_agent
from semiuhpe_amd.src.config import get_config
from semiuhpe_amd.src.datasets import build_loaders
from semiuhpe_amd.src.utils import compute_euler_angles_from_rotation_matrices, limit_angle


def euler_errors_dad_trained(pd_m, gt_e):
    """Model trained on DAD-3DHeads, tested on an Euler-labelled set (reference eval.py:61-72): xyz Euler angles of
    R^T with the 180 degree pitch offset."""
    from scipy.spatial.transform import Rotation
    out = []
    for rot_mat in pd_m.detach().cpu().numpy():
        angle = Rotation.from_matrix(np.transpose(rot_mat)).as_euler("xyz", degrees=True)
        roll, pitch, yaw = list(map(limit_angle, [angle[2], angle[0] - 180, angle[1]]))
        out.append([pitch, yaw, roll])
    return abs(gt_e.cpu().numpy() - np.array(out))


def evaluate(config, test_loader, agent, eval_ema=False, dump_json=True):
    ema_name = "EMA_" if eval_ema else ""
    e_pitch, e_yaw, e_roll, e_rot, e_rel, gt_yaw = [], [], [], [], [], []
    dev = agent.device
    for data in test_loader:
        fisher_dict, _, _ = agent.val_func(data, eval_ema=eval_ema)
        pd_m = fisher_dict["pred_orth"]
        gt_m = data.get("rot_mat").to(dev)
        if "euler_angles" in data:
            gt_e = data.get("euler_angles")
            if config.train_labeled == "DAD3DHeads":
                err_deg = euler_errors_dad_trained(pd_m, gt_e)
            else:
                pd_e = compute_euler_angles_from_rotation_matrices(pd_m, full_range=False) * 180 / np.pi
                err_deg = abs((gt_e.to(dev) - pd_e).detach().cpu().numpy())
            e_pitch.append(err_deg[:, 0]); e_yaw.append(err_deg[:, 1]); e_roll.append(err_deg[:, 2])
            gt_yaw.append(np.asarray(gt_e[:, 1].cpu()))
        else:
            e_rel.append(agent.compute_err_deg_from_matrices(pd_m, gt_m).detach().cpu().numpy())
            R_dist = (pd_m @ gt_m.transpose(1, 2)).detach().cpu().numpy()
            e_rot.append([np.linalg.norm(np.eye(3) - R, "fro") for R in R_dist])
            g = compute_euler_angles_from_rotation_matrices(gt_m, full_range=True) * 180 / np.pi
            gt_yaw.append(g.detach().cpu().numpy()[:, 1])
    print(f"==== {ema_name}exp: {config.exp_name} ====")
    gt_yaw = np.concatenate(gt_yaw, 0)
    if e_rot:
        e_rel, e_rot = np.concatenate(e_rel, 0), np.concatenate(e_rot, 0)
        print(f"{ema_name}Relative Angle mean: {np.mean(e_rel):.4f}; {ema_name}Rotation Matrix mean: {np.mean(e_rot):.4f}")
        result = dict(relative_angle=float(np.mean(e_rel)), rotation_matrix=float(np.mean(e_rot)))
        rows = [[float(a), float(b), float(c)] for a, b, c in zip(gt_yaw, e_rel, e_rot)]
        name = f"./ERR_{config.exp_name}_{config.network}_{config.test_set}.json"
    else:
        p, y, r = np.concatenate(e_pitch, 0), np.concatenate(e_yaw, 0), np.concatenate(e_roll, 0)
        m = (p + y + r) / 3
        print(f"{ema_name}Euler MAE (pitch, yaw, roll, mean): {np.mean(p):.4f}, {np.mean(y):.4f}, {np.mean(r):.4f}, {np.mean(m):.4f}")
        result = dict(pitch=float(np.mean(p)), yaw=float(np.mean(y)), roll=float(np.mean(r)), mean=float(np.mean(m)))
        rows = [[float(a), float(b), float(c), float(d)] for a, b, c, d in zip(gt_yaw, p, y, r)]
        name = f"./exps/MAE_{config.exp_name}_{config.network}_{config.test_set}.json"
    if dump_json and not eval_ema:
        os.makedirs(os.path.dirname(name) or ".", exist_ok=True)
        with open(name, "w") as fh:
            json.dump(rows, fh)
    return result


def test():
    config = get_config("test")
    _, _, test_loader = build_loaders(config)
    agent = get_agent(config)
    agent.load_ckpt(config.ckpt)
    evaluate(config, test_loader, agent)
    evaluate(config, test_loader, agent, eval_ema=True)


if __name__ == "__main__":
    with torch.no_grad():
        test()
