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


def _bucket_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from semiuhpe_amd.src import distributed as D
    D.init_from_env("gloo")
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(10_000, generator=g)
    whole = flat.clone()
    D.allreduce_sum_(whole)                                     # today's single collective
    names = [f"features.{i // 3}.{i % 3}.block.0.weight" for i in range(12)] + ["classifier.1.weight", "classifier.4.weight"]
    sizes = [500] * 12 + [3000, 1000]
    offsets = [sum(sizes[:i]) for i in range(len(sizes))]
    bounds = D.bucket_boundaries(names, offsets, 10_000, 4)
    sent = []
    br = D.BucketedAllReduce(flat, bounds, ready=lambda k: sent.append(k))
    br.begin()
    for off in reversed(offsets):                               # the backward touches the units back to front
        br.touched(off)
    in_flight = list(sent)
    try:                                                        # a write into a bucket that is already on the wire must not pass silently
        br.touched(offsets[-1])
        late = False
    except RuntimeError:
        late = True
    br.finish()
    out[rank] = dict(whole=whole.numpy(), bucketed=flat.numpy(), bounds=bounds, in_flight=in_flight, sent=sent, late_raises=late)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_is_bit_identical_and_early():
    """reverse-layer buckets: cut at unit starts, sent as soon as the backward has left them, same bits as one collective"""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bucket_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert np.array_equal(r0["bucketed"], r0["whole"]) and np.array_equal(r1["bucketed"], r0["whole"])
    b = r0["bounds"]
    assert b[0] == 0 and b[-1] == 10_000 and b == sorted(b) and len(b) >= 4
    assert all(x % 500 == 0 for x in b[1:-1])                   # cut at unit (parameter-block) starts only
    assert r0["sent"] == list(range(len(b) - 2, -1, -1))        # back to front, every bucket once
    assert len(r0["in_flight"]) == len(b) - 2                   # all but the first bucket were sent before finish()
    assert r0["late_raises"] and r1["late_raises"]              # out-of-order gradient writes are an error, not a silent divergence


def test_bucket_boundaries_follow_network_units():
    from semiuhpe_amd.src.distributed import bucket_boundaries
    names = ["features.0.0.weight", "features.0.1.weight", "features.1.0.block.0.0.weight", "features.1.0.block.0.1.weight",
             "features.1.1.block.0.0.weight", "layers.1.blocks.0.attn.qkv.weight", "layers.1.blocks.1.mlp.fc1.weight", "head.1.weight"]
    offs = [0, 10, 20, 40, 60, 100, 140, 180]
    b = bucket_boundaries(names, offs, 200, 3)
    assert b[0] == 0 and b[-1] == 200
    assert set(b[1:-1]) <= {10, 20, 60, 100, 180}               # never inside features.1.0 (40) nor inside layers.1 (140)
    assert bucket_boundaries(names, offs, 200, 1) == [0, 200]
