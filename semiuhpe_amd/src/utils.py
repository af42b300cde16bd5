"""Drop-in for the parts of the reference's src/utils.py that sit on the training / metric path."""
import os

import numpy as np
import torch

from .. import kernels as K


class TrainClock(object):
    """reference src/utils.py:11-42: epoch / minibatch / iteration counters (checkpointed)."""

    def __init__(self):
        self.epoch = 0
        self.minibatch = 0
        self.iteration = 0
        self.scratch_iter = 0

    def tick(self):
        self.minibatch += 1
        self.iteration += 1
        self.scratch_iter += 1

    def tock(self):
        self.epoch += 1
        self.minibatch = 0

    def make_checkpoint(self):
        return {"epoch": self.epoch, "minibatch": self.minibatch, "iteration": self.iteration}

    def restore_checkpoint(self, clock_dict):
        self.epoch = clock_dict["epoch"]
        self.minibatch = clock_dict["minibatch"]
        self.iteration = clock_dict["iteration"]


def ensure_dirs(paths):
    if isinstance(paths, (list, tuple)):
        for p in paths:
            os.makedirs(p, exist_ok=True)
    else:
        os.makedirs(paths, exist_ok=True)


def cycle(iterable):
    """reference :175-178 (a DistributedSampler behind the loader gets a new epoch per pass, so every pass reshuffles)"""
    epoch = 0
    while True:
        sampler = getattr(iterable, "sampler", None)
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)
        for x in iterable:
            yield x
        epoch += 1


def requires_grad(xs, req=False):
    """reference :181-185"""
    if not isinstance(xs, (tuple, list)):
        xs = tuple(xs)
    for x in xs:
        x.requires_grad_(req)


def dict_get(d, key, default, default_device="cuda"):
    """reference :188-194"""
    v = d.get(key)
    if v is None or v.nelement() == 0:
        return torch.tensor([default]).float().to(default_device)
    return v


def get_6DRepNet_Rot(x, y, z):
    """reference :204-225: Rz(z) Ry(y) Rx(x), radians."""
    Rx = np.array([[1, 0, 0], [0, np.cos(x), -np.sin(x)], [0, np.sin(x), np.cos(x)]])
    Ry = np.array([[np.cos(y), 0, np.sin(y)], [0, 1, 0], [-np.sin(y), 0, np.cos(y)]])
    Rz = np.array([[np.cos(z), -np.sin(z), 0], [np.sin(z), np.cos(z), 0], [0, 0, 1]])
    return Rz.dot(Ry.dot(Rx))


def compute_euler_angles_from_rotation_matrices(rotation_matrices, full_range=False, use_gpu=True, gpu_id=0):
    """reference :232-260 -> (n,3) radians [pitch, yaw, roll]; one kernel instead of a per-sample Python loop with a
    host sync per item."""
    R = rotation_matrices.reshape(-1, 9)
    return K.euler_from_rot(R if R.is_contiguous() else R.contiguous(), full_range)


def limit_angle(angle, pi=180.0):
    """reference :287-300"""
    if angle < -pi:
        k = -2 * (int(angle / pi) // 2)
        angle = angle + k * pi
    if angle > pi:
        k = 2 * ((int(angle / pi) + 1) // 2)
        angle = angle - k * pi
    return angle
