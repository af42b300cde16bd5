"""world_size-2 gloo tests (CPU) of the data-parallel plumbing: the flat-gradient SUM all-reduce with 1/world
pre-scaling reproduces the global-batch mean gradient, the variable-length entropy gather gives every rank the same
threshold as a single process would compute, and ranks draw disjoint synthetic shards."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from semiuhpe_amd.src import distributed as D
    from semiuhpe_amd.synthetic import synthetic_batches
    r, w = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    # --- gradient exchange: per-rank mean loss L_r = mean_i f(x_i); seed 1/world; SUM == global mean gradient
    g = torch.Generator().manual_seed(7)
    wgt = torch.randn(16, generator=g, dtype=torch.float64).requires_grad_(True)
    data, _ = synthetic_batches(4, 0, size=8, seed=rank)                    # this rank's shard
    x = data["img"].double().reshape(4, -1)[:, :16]
    loss = ((x @ wgt) ** 2).mean()
    loss.backward(D.grad_seed("cpu").double())
    flat = wgt.grad.clone()
    D.allreduce_sum_(flat)
    # --- entropy gather: different lengths per rank
    ent = torch.arange(5 + 3 * rank, dtype=torch.float32) * (1 + rank) - 7.0
    allent = D.gather_varlen(ent)
    b = torch.tensor([float(rank)])
    D.broadcast_(b, 0)
    # --- sharded validation: batches dealt round-robin, per-sample results pooled on every rank
    mine = list(D.shard([np.arange(3) + 10 * i for i in range(5)]))
    pooled = D.gather_values(mine)
    out[rank] = dict(flat=flat.numpy(), x=x.numpy(), ent=allent.numpy(), b=float(b), mine=[m.tolist() for m in mine], pooled=pooled)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert np.array_equal(r0["flat"], r1["flat"])
    assert not np.array_equal(r0["x"], r1["x"])                            # disjoint shards
    g = torch.Generator().manual_seed(7)
    wgt = torch.randn(16, generator=g, dtype=torch.float64).requires_grad_(True)
    xs = torch.from_numpy(np.concatenate([r0["x"], r1["x"]], 0))
    ((xs @ wgt) ** 2).mean().backward()                                     # single-process global batch
    assert np.allclose(r0["flat"], wgt.grad.numpy(), rtol=1e-12, atol=1e-12)
    expect = np.concatenate([np.arange(5, dtype=np.float32) - 7.0, np.arange(8, dtype=np.float32) * 2 - 7.0])
    assert np.array_equal(r0["ent"], expect) and np.array_equal(r1["ent"], expect)
    from oracle.pose_utils import entropy_threshold
    assert entropy_threshold(r0["ent"], 0.75) == entropy_threshold(r1["ent"], 0.75) == np.sort(expect)[int(13 * 0.75)]
    assert r0["b"] == r1["b"] == 0.0
    assert r0["mine"] == [[0, 1, 2], [20, 21, 22], [40, 41, 42]] and r1["mine"] == [[10, 11, 12], [30, 31, 32]]
    whole = np.sort(np.concatenate([np.arange(3) + 10 * i for i in range(5)]).astype(np.float64))
    assert np.array_equal(np.sort(r0["pooled"]), whole) and np.array_equal(r0["pooled"], r1["pooled"])
