"""Worker of tests/test_comm_abi_two_ranks.py: one rank of a two-process job that drives the suhpe_comm_* C ABI (include/suhpe.h) over the
host-memory transport (tests/host_build/hostmem_transport.c, selected with SUHPE_RCCL_LIB): argv = mode rank world id_file"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def abi(rank, world, id_file):
    from semiuhpe_amd import _lib
    lib = _lib.load()
    assert lib.suhpe_comm_available() == 1
    if rank == 0:
        buf = ctypes.create_string_buffer(128)
        assert lib.suhpe_comm_unique_id(buf) == 0
        with open(id_file + ".tmp", "wb") as fh:
            fh.write(buf.raw)
        os.replace(id_file + ".tmp", id_file)
    while not os.path.exists(id_file):
        time.sleep(0.01)
    uid = open(id_file, "rb").read()
    assert len(uid) == 128
    h = ctypes.c_void_p()
    # argument validation happens before the transport is touched
    assert lib.suhpe_comm_init(ctypes.byref(h), uid, world, world) != 0 and "outside a world" in _lib.last_error()
    assert lib.suhpe_comm_init(ctypes.byref(h), uid, rank, world) == 0, _lib.last_error()
    assert lib.suhpe_comm_rank(h) == rank and lib.suhpe_comm_world(h) == world
    # all-reduce: sum, then average (in place), odd length
    x = f32(np.arange(1001) * (rank + 1) + 0.25 * rank)
    want = sum(f32(np.arange(1001) * (r + 1) + 0.25 * r) for r in range(world))
    a = x.copy()
    assert lib.suhpe_comm_allreduce(h, ptr(a), a.size, 0, None) == 0
    assert np.array_equal(a, want), "sum"
    b = x.copy()
    assert lib.suhpe_comm_allreduce(h, ptr(b), b.size, 1, None) == 0
    assert np.allclose(b, want / world, rtol=1e-6), "average"
    # zero-length calls return without entering the collective (a rank that has nothing to send must not hang the others)
    assert lib.suhpe_comm_allreduce(h, None, 0, 0, None) == 0
    assert lib.suhpe_comm_broadcast(h, None, 0, 0, None) == 0
    # all-gather: rank r's block lands in row r
    s = f32(np.full(37, 10.0 * rank) + np.arange(37))
    g = np.empty((world, 37), dtype=np.float32)
    assert lib.suhpe_comm_allgather(h, ptr(s), ptr(g), 37, None) == 0
    for r in range(world):
        assert np.array_equal(g[r], f32(np.full(37, 10.0 * r) + np.arange(37))), "allgather"
    # broadcast from the LAST rank (root != 0 exercises the root argument)
    c = f32(np.full(64, 100.0 + rank))
    assert lib.suhpe_comm_broadcast(h, ptr(c), c.size, world - 1, None) == 0
    assert np.array_equal(c, f32(np.full(64, 100.0 + world - 1))), "broadcast"
    # error mapping: a root outside the world is SUHPE_EINVAL on every rank, without a collective
    assert lib.suhpe_comm_broadcast(h, ptr(c), c.size, world, None) == -1 and "root" in _lib.last_error()
    assert lib.suhpe_comm_allreduce(h, None, 5, 0, None) == -1
    assert lib.suhpe_comm_allgather(h, ptr(s), None, 37, None) == -1
    assert lib.suhpe_comm_destroy(h) == 0
    print(f"rank {rank}: abi ok", flush=True)


def dist_native(rank, world, port):
    """the Python host of the same ABI: semiuhpe_amd.src.distributed in SUHPE_COMM=native mode (torch.distributed on gloo as the host channel,
    every collective of the step through suhpe_comm_*), on host tensors"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from semiuhpe_amd.src import distributed as D
    D.init_from_env("nccl")                                    # native mode turns this into gloo: no torch GPU communicator
    assert dist.get_backend() == "gloo" and D.native() is not None and D.native().world == world
    t = torch.arange(10, dtype=torch.float32) + rank
    D.allreduce_sum_(t)
    assert torch.equal(t, torch.arange(10, dtype=torch.float32) * world + sum(range(world)))
    u = torch.full((7,), float(rank))
    D.allreduce_avg_(u)
    assert torch.allclose(u, torch.full((7,), sum(range(world)) / world))
    w = torch.full((5,), float(rank))
    D.broadcast_(w, 0)
    assert torch.equal(w, torch.zeros(5))
    # ragged gather (the entropy vectors of the dynamic threshold: sharded loaders leave the ranks different counts), incl. an empty contribution
    v = torch.arange(3 * rank, dtype=torch.float32) + 100 * rank
    got = D.gather_varlen(v)
    want = torch.cat([torch.arange(3 * r, dtype=torch.float32) + 100 * r for r in range(world)])
    assert torch.equal(got, want), (got, want)
    # bucketed exchange: bit-identical to one all-reduce, every bucket through the library's communicator
    torch.manual_seed(rank)
    flat = torch.randn(1000)
    ref = flat.clone()
    D.allreduce_sum_(ref)
    bk = D.BucketedAllReduce(flat, [0, 100, 400, 1000])
    bk.touched(900); bk.touched(300); bk.touched(50)
    bk.finish()
    assert torch.equal(flat, ref)
    D.barrier("cpu")
    D.native().close()
    dist.destroy_process_group()
    print(f"rank {rank}: distributed native ok", flush=True)


if __name__ == "__main__":
    mode, rank, world, arg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    if mode == "abi":
        abi(rank, world, arg)
    else:
        dist_native(rank, world, int(arg))
