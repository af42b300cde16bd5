"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm) / xGMI.

The reference has no multi-GPU code at all; this is additive.  The path shards by images: every rank
draws its own B/world labeled + muB/world unlabeled images and holds full replicas of student, teacher and Adam
state.  Per step there is ONE exchange: the flat gradient buffer (83.6 MB for EfficientNetV2-S) is all-reduced as a
single large collective — gradients are pre-scaled by 1/world in backward, so a plain SUM yields the global-batch mean
and no extra pass over the buffer is needed.  BatchNorm running statistics (the teacher's drift per rank because it
runs in train mode) are averaged with one small AVG all-reduce per network.  The dynamic entropy threshold
all-gathers the per-rank entropy vectors so that every rank sorts the same array.
"""
import os

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend="nccl"):
    """torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) -> process group."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1 or dist.is_initialized():
        return world()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {}
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        kw["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend, **kw)
    return world()


def allreduce_sum_(t):
    if world()[1] > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_avg_(t):
    """Average across ranks inside the collective (RCCL implements AVG); only used on device tensors."""
    if world()[1] > 1:
        if dist.get_backend() == "gloo":          # gloo has no AVG (CPU tests, and the single-box rehearsal of bench.py --gpus 2)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t.div_(world()[1])
        else:
            dist.all_reduce(t, op=dist.ReduceOp.AVG)
    return t


def broadcast_(t, src=0):
    if world()[1] > 1:
        dist.broadcast(t, src)
    return t


def gather_varlen(t):
    """Concatenate 1-D tensors of different lengths from all ranks, in rank order, on every rank."""
    rank, ws = world()
    if ws == 1:
        return t
    n = torch.tensor([t.numel()], dtype=torch.long, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
    pad[:t.numel()] = t
    parts = [torch.zeros_like(pad) for _ in range(ws)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)])


def shard(iterable):
    """This rank's share of an iterable of batches (round-robin: batch i goes to rank i % world).  Validation walks the test
    loader through this and pools the per-sample results with gather_values, so N ranks validate N times faster and still
    report the statistics of the whole set."""
    rank, ws = world()
    for i, item in enumerate(iterable):
        if i % ws == rank:
            yield item


def gather_values(values, device="cpu"):
    """1-D float array holding `values` (floats / arrays, flattened) of every rank, rank after rank, on every rank."""
    import numpy as np
    flat = np.concatenate([np.asarray(v, dtype=np.float64).reshape(-1) for v in values]) if len(values) else np.zeros(0)
    if world()[1] == 1:
        return flat
    t = torch.from_numpy(flat).to(device)
    return gather_varlen(t).cpu().numpy()


def grad_seed(device):
    """d(loss_all)/d(loss_all) = 1/world: with per-rank mean losses this makes SUM(all-reduce) the global-batch mean."""
    return torch.full((), 1.0 / world()[1], dtype=torch.float32, device=device)
