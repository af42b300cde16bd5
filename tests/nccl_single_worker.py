"""Worker of tests/test_gpu_train_eval.py::test_rccl_api_on_one_rank: the data-parallel plumbing on the backend it is written for
(nccl = RCCL), with the one GPU this pool's boxes have: a single-rank process group.  Every collective is then an identity, but the
calls, stream hand-offs and async handles are the real RCCL ones: init with device_id, SUM / AVG all-reduce, broadcast,
variable-length gather, broadcast_object_list, and the bucketed gradient exchange issued from inside a training step's backward."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
from types import SimpleNamespace
import torch
import torch.distributed as dist
from semiuhpe_amd import kernels as K, layers as L
from semiuhpe_amd.src import distributed as D
from semiuhpe_amd.src.agent import SSLAgent
from semiuhpe_amd.src.config import _shared_run_date
from semiuhpe_amd.synthetic import synthetic_batches

if __name__ == "__main__":
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    t = torch.arange(8, dtype=torch.float32, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.AVG)                     # RCCL implements AVG (distributed.allreduce_avg_)
    dist.broadcast(t, 0)
    assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))
    n = torch.tensor([t.numel()], dtype=torch.long, device="cuda")
    sizes = [torch.zeros_like(n)]
    dist.all_gather(sizes, n)
    assert int(sizes[0]) == 8
    box = ["Oct03_000000"]
    dist.broadcast_object_list(box, src=0)
    assert box[0] == "Oct03_000000" and isinstance(_shared_run_date(), str)
    # the bucketed exchange on RCCL handles, driven by a real backward
    torch.manual_seed(3)
    cfg = SimpleNamespace(lr=1e-4, log_dir="/tmp/suhpe_nccl1", model_dir="/tmp/suhpe_nccl1", distribution="matrixFisher", is_full_range=False,
                          rotate_aug=True, train_labeled="300WLP", train_unlabeled="300WLP", cutout_aug=True, cutmix_aug=True, conf_thres=1e9,
                          type_unsuper="ce", SSL_lambda=1.0, is_ema=True, ema_decay=0.999, eman=False, stage1_iteration=0, network="effinetv2",
                          num_classes=9, exp_name="t", date="t", left_ratio=0.95, save_feat=False, autotune=False)
    agent = SSLAgent(cfg)
    fb = agent.optimizer.flat_buffers[0]
    names = [k for k, _ in agent.net.named_parameters()]
    bounds = D.bucket_boundaries(names, fb.offsets, fb.numel, 4)
    sent = []
    agent._buckets = D.BucketedAllReduce(fb.grad, bounds, ready=lambda k: (sent.append(k), K.flush_deferred_now()),
                                         streams=lambda: K.backward_streams(agent.device))
    agent._grad_base = fb.grad.data_ptr()
    agent.world_size = 2                                          # takes the multi-rank branches of _exchange (1-rank collectives)
    data, ulb = synthetic_batches(4, 8, size=64, seed=0, rotate_aug=True)
    data = {k: v.cuda() for k, v in data.items()}
    ulb = {k: v.cuda() for k, v in ulb.items()}
    ref = SSLAgent(cfg)                                           # same seed state? no: copy the weights instead
    ref.net.load_state_dict(agent.net.state_dict()); ref.ema_net.load_state_dict(agent.ema_net.state_dict())
    for a in (agent, ref):
        K.rng_counter(a.device, seed=11)
        torch.manual_seed(5)
        a.train_func(data, ulb)
    torch.cuda.synchronize()
    assert sent == list(range(len(bounds) - 2, -1, -1)), sent
    g1, g2 = fb.grad, ref.optimizer.flat_buffers[0].grad
    assert torch.equal(g1, g2) and float(g1.abs().max()) > 0      # identity collectives: same gradient as the plain step
    assert torch.equal(fb.data, ref.optimizer.flat_buffers[0].data)
    # the C-ABI communicator (suhpe_comm_*) on the same single rank: its own RCCL communicator next to torch's
    nc = D.NativeComm(rank=0, world_size=1)
    v = torch.arange(1000, dtype=torch.float32, device="cuda")
    nc.allreduce_(v); nc.allreduce_(v, average=True); nc.broadcast_(v, 0)
    got = nc.allgather(v)
    torch.cuda.synchronize()
    assert got.shape == (1, 1000) and torch.equal(got[0], v) and torch.equal(v.cpu(), torch.arange(1000, dtype=torch.float32))
    fb2 = fb.grad.clone()
    D._NATIVE = nc                                                # the bucketed exchange through suhpe_comm_allreduce
    bk = D.BucketedAllReduce(fb2, bounds)
    bk.finish()
    torch.cuda.synchronize()
    assert torch.equal(fb2, fb.grad)
    D._NATIVE = None
    nc.close()
    ent = D.gather_varlen(torch.randn(5, device="cuda"))
    assert ent.numel() == 5
    dist.barrier()
    dist.destroy_process_group()
    print("NCCL_SINGLE_OK")
