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


def native_mode():
    """SUHPE_COMM=native: every DEVICE collective of the job goes through suhpe_comm_* (ONE library-owned RCCL communicator, so one order of
    collectives), and torch.distributed is brought up on gloo -- a host channel for the rendezvous, the unique id and Python objects only."""
    return os.environ.get("SUHPE_COMM", "") == "native"


def init_from_env(backend="nccl"):
    """torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) -> process group."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1 or dist.is_initialized():
        return world()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {}
    if native_mode() and backend == "nccl":
        # no second GPU communicator beside the library's: torch.distributed keeps the host side (store, object broadcasts, host barriers)
        backend = "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        kw["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend, **kw)
    return world()


# ------------------------------------------------------------------------------------------------ the C-ABI communicator
class NativeComm:
    """suhpe_comm_* (include/suhpe.h): this rank's RCCL communicator behind the C ABI -- float32 device tensors, enqueued on the
    current stream of the tensor's device.  The 128-byte unique id travels from rank 0 over the torch.distributed process group
    (any backend), or is passed in.  Opt-in (SUHPE_COMM=native, see native()): torch.distributed's own RCCL binding is the default
    transport of the training step, this is the path a non-PyTorch host would take."""

    def __init__(self, rank=None, world_size=None, unique_id=None):
        import ctypes
        from .. import _lib
        self._ct, self._libmod = ctypes, _lib
        self.lib = _lib.load()
        if not self.lib.suhpe_comm_available():
            raise RuntimeError("semiuhpe_amd: librccl.so.1 not found: suhpe_comm_* unavailable")
        r, w = world()
        self.rank = r if rank is None else rank
        self.world = w if world_size is None else world_size
        if unique_id is None:
            box = [None]
            if self.rank == 0:
                buf = ctypes.create_string_buffer(128)
                self._check(self.lib.suhpe_comm_unique_id(buf))
                box[0] = buf.raw
            if self.world > 1:
                dist.broadcast_object_list(box, src=0)
            unique_id = box[0]
        self.unique_id = bytes(unique_id)
        self.handle = ctypes.c_void_p()
        self._check(self.lib.suhpe_comm_init(ctypes.byref(self.handle), self.unique_id, self.rank, self.world))

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self._libmod.last_error())

    # SUHPE_COMM_HOST_BUFFERS=1 (tests only, together with SUHPE_RCCL_LIB = the host-memory transport of tests/host_build): host tensors are accepted
    HOST_OK = os.environ.get("SUHPE_COMM_HOST_BUFFERS", "0") == "1"

    @classmethod
    def _chk(cls, t):
        assert (t.is_cuda or cls.HOST_OK) and t.dtype == torch.float32 and t.is_contiguous(), "suhpe_comm_*: contiguous float32 device tensors"

    def _stream(self, t):
        return self._ct.c_void_p(torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0)

    def barrier(self, device):
        """all ranks have enqueued everything before this point on the current stream (a one-element SUM all-reduce; the host does not wait)"""
        t = torch.zeros(1, dtype=torch.float32, device=device)
        return self.allreduce_(t)

    def gather_varlen(self, t):
        """1-D float32 tensors of different lengths from all ranks, concatenated in rank order on every rank (two all-gathers: lengths, padded data)"""
        self._chk(t)
        n = torch.tensor([float(t.numel())], dtype=torch.float32, device=t.device)      # (exact below 2^24 elements per rank)
        assert t.numel() < (1 << 24)
        sizes = [int(v) for v in self.allgather(n).reshape(-1).tolist()]
        mx = max(max(sizes), 1)
        pad = torch.zeros(mx, dtype=torch.float32, device=t.device)
        pad[:t.numel()] = t
        parts = self.allgather(pad)
        return torch.cat([parts[r, :sizes[r]] for r in range(self.world)])

    def allreduce_(self, t, average=False):
        self._chk(t)
        self._check(self.lib.suhpe_comm_allreduce(self.handle, self._ct.c_void_p(t.data_ptr()), t.numel(), int(bool(average)), self._stream(t)))
        return t

    def allgather(self, t):
        """-> (world, *t.shape): rank r's tensor in row r (equal shapes on every rank)"""
        self._chk(t)
        out = torch.empty((self.world,) + tuple(t.shape), dtype=torch.float32, device=t.device)
        self._check(self.lib.suhpe_comm_allgather(self.handle, self._ct.c_void_p(t.data_ptr()), self._ct.c_void_p(out.data_ptr()), t.numel(), self._stream(t)))
        return out

    def broadcast_(self, t, src=0):
        self._chk(t)
        self._check(self.lib.suhpe_comm_broadcast(self.handle, self._ct.c_void_p(t.data_ptr()), t.numel(), int(src), self._stream(t)))
        return t

    def close(self):
        if self.handle:
            self._check(self.lib.suhpe_comm_destroy(self.handle))
            self.handle = self._ct.c_void_p()


_NATIVE = None


def native():
    """the process-wide NativeComm when SUHPE_COMM=native and more than one rank runs (created on first use), else None.
    In this mode the library's communicator is the ONLY one that touches the GPUs (init_from_env brings torch.distributed up on gloo), so the
    order of device collectives is the order this process enqueues them -- gradient buckets, BatchNorm-buffer average, broadcasts, the entropy
    gather and barriers alike.  Two-rank coverage without GPUs: tests/test_comm_abi_two_ranks.py (host-memory transport behind the same C ABI);
    on RCCL itself the mode has run with one rank only (no multi-GPU box in the pool): the default transport stays torch.distributed."""
    global _NATIVE
    if _NATIVE is None and native_mode() and world()[1] > 1 and (torch.cuda.is_available() or NativeComm.HOST_OK):
        _NATIVE = NativeComm()
    return _NATIVE


def _native_for(t):
    return native() if ((t.is_cuda or NativeComm.HOST_OK) and t.dtype == torch.float32 and t.is_contiguous()) else None


def barrier(device=None):
    """device-side barrier in native mode (no torch GPU communicator exists there), torch.distributed's otherwise"""
    if world()[1] > 1:
        nc = native()
        if nc is not None and device is not None:
            nc.barrier(device)
            if torch.device(device).type == "cuda":
                torch.cuda.current_stream(device).synchronize()
        dist.barrier()


def allreduce_sum_(t):
    if world()[1] > 1:
        nc = _native_for(t)
        if nc is not None:
            return nc.allreduce_(t)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_avg_(t):
    """Average across ranks inside the collective (RCCL implements AVG); only used on device tensors."""
    if world()[1] > 1:
        nc = _native_for(t)
        if nc is not None:
            return nc.allreduce_(t, average=True)
        if dist.get_backend() == "gloo":          # gloo has no AVG (CPU tests, and the single-box rehearsal of bench.py --gpus 2)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t.div_(world()[1])
        else:
            dist.all_reduce(t, op=dist.ReduceOp.AVG)
    return t


def broadcast_(t, src=0):
    if world()[1] > 1:
        nc = _native_for(t)
        if nc is not None:
            return nc.broadcast_(t, src)
        dist.broadcast(t, src)
    return t


def gather_varlen(t):
    """Concatenate 1-D tensors of different lengths from all ranks, in rank order, on every rank."""
    rank, ws = world()
    if ws == 1:
        return t
    nc = _native_for(t) if t.dim() == 1 else None
    if nc is not None:
        return nc.gather_varlen(t)
    if native_mode() and t.is_cuda:
        return gather_varlen(t.cpu()).to(t.device)      # (other dtypes: over the host channel -- no torch GPU communicator in native mode)
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


# ------------------------------------------------------------------------------------------------ bucketed gradient exchange
def bucket_boundaries(names, offsets, numel, n_buckets):
    """Flat-buffer offsets [0 = b0 < b1 < ... < numel] that cut the flat gradient into ~n_buckets buckets of similar size, each
    cut at the start of a UNIT of the network: a unit is what the parameter names share up to their last leading numeric
    component ('features.4.2', 'layer3.5', 'layers.1', 'classifier.4'), i.e. a block whose backward has finished before the
    backward of the block in front of it starts.  Buckets are filled from the END of the buffer (the order the backward runs)."""
    def unit(name):
        parts = name.split(".")
        key = [parts[0]]
        for q in parts[1:]:
            if not q.isdigit():
                break
            key.append(q)
        return tuple(key)

    starts, last = [], None
    for name, off in zip(names, offsets):
        u = unit(name)
        if u != last:
            starts.append(off)
            last = u
    target = max(1, numel // max(1, n_buckets))
    cuts, hi = [], numel
    for st in reversed(starts):
        if st > 0 and hi - st >= target:
            cuts.append(st)
            hi = st
    return [0] + sorted(cuts) + [numel]


class BucketedAllReduce:
    """SUM all-reduce of one flat gradient buffer in reverse-layer buckets, each issued as soon as the backward has moved past
    it, so that the exchange over xGMI runs under the rest of the backward.

    `touched(offset)` is called (semiuhpe_amd.layers.grad_of) with the flat offset of every gradient slice a layer is about to
    accumulate into.  The backward visits the units of a network back to front, so the first touch below a bucket's start means
    every kernel that writes into that bucket has been enqueued: `ready(k)` lets the owner flush what it defers (the split-K
    slab reduction of the filter gradients), then bucket k's collective is enqueued on a communication stream behind the
    compute streams.  `finish()` sends what is left (bucket 0 at least) and joins.  Elementwise sums do not depend on how the
    buffer is cut: results are bit-identical to ONE all-reduce of the whole buffer (tests/test_distributed_cpu.py)."""

    def __init__(self, flat, boundaries, ready=None, streams=None):
        self.flat, self.bounds = flat, list(boundaries)
        self.ready = ready                      # callback(k) before bucket k is sent (flush deferred work)
        self.streams = streams                  # callable -> compute streams the collective has to wait for (CUDA only)
        self.comm = None
        self.exposed = []                       # (event before, event after) of the main stream's wait for the communication stream, per step
        self.exposed_host_s = 0.0               # host tensors (gloo tests): seconds spent in the handles' wait()
        self.begin()

    def begin(self):
        self.next = len(self.bounds) - 2        # highest bucket not yet sent
        self.handles = []
        self.active = True

    def _send(self, k):
        if self.ready is not None:
            self.ready(k)
        view = self.flat[self.bounds[k]:self.bounds[k + 1]]
        if self.flat.is_cuda:
            if self.comm is None:
                self.comm = torch.cuda.Stream(device=self.flat.device)
            for st in (self.streams() if self.streams else [torch.cuda.current_stream(self.flat.device)]):
                self.comm.wait_stream(st)
            with torch.cuda.stream(self.comm):
                nc = native()
                if nc is not None:
                    nc.allreduce_(view)          # enqueued on the communication stream: stream order is the only handle needed
                else:
                    self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))
        else:
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))

    def touched(self, offset):
        """a layer is about to write gradients at flat offset `offset`"""
        if self.active and self.next + 1 < len(self.bounds) - 1 and offset >= self.bounds[self.next + 1]:
            # the slice lies in a bucket whose collective is already enqueued: this gradient would never be exchanged and the
            # ranks would step apart silently (a module registered out of forward order, a shared parameter, a second backward)
            raise RuntimeError(f"BucketedAllReduce: gradient write at flat offset {offset} after its bucket "
                               f"[{self.bounds[self.next + 1]}, ...) was sent; the backward does not visit the parameters in reverse "
                               "registration order -- run with SUHPE_GRAD_BUCKETS=1 (one exchange after the backward)")
        while self.active and self.next > 0 and offset < self.bounds[self.next]:
            self._send(self.next)
            self.next -= 1

    def finish(self):
        while self.next >= 0:
            self._send(self.next)
            self.next -= 1
        if self.flat.is_cuda and self.comm is not None:
            with torch.cuda.stream(self.comm):
                for h in self.handles:
                    h.wait()
            # what the step pays for the exchange: the time the compute stream sits in this wait (everything else ran under the backward)
            cur = torch.cuda.current_stream(self.flat.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            cur.wait_stream(self.comm)
            e1.record(cur)
            self.exposed.append((e0, e1))
            if len(self.exposed) > 64:
                del self.exposed[:-64]
        else:
            import time
            t0 = time.perf_counter()
            for h in self.handles:
                h.wait()
            self.exposed_host_s += time.perf_counter() - t0
        self.handles = []
        self.active = False


    def exposed_ms(self, last=None):
        """mean milliseconds per step the compute stream waited for the gradient exchange in finish() (device runs: the last `last` steps, after a
        synchronize by the caller), i.e. the part of the all-reduce the backward did not hide"""
        ev = self.exposed[-last:] if last else self.exposed
        if not ev:
            return None
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev)
