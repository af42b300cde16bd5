"""Worker of tests/test_gpu_train_eval.py::test_bucketed_exchange_bit_identical: one of two data-parallel ranks (gloo over CUDA tensors,
both on GPU 0) runs two SSL steps and one supervised step of SSLAgent and writes the parameters / gradients it ends with."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from types import SimpleNamespace
import torch
from semiuhpe_amd.src import distributed as D
from semiuhpe_amd.src.agent import SSLAgent
from semiuhpe_amd.synthetic import synthetic_batches

if __name__ == "__main__":
    out = sys.argv[1]
    torch.cuda.set_device(0)
    rank, world = D.init_from_env("gloo")
    torch.manual_seed(11)
    cfg = SimpleNamespace(lr=1e-4, log_dir=out, model_dir=out, distribution="matrixFisher", is_full_range=False, rotate_aug=True,
                          train_labeled="300WLP", train_unlabeled="300WLP", cutout_aug=True, cutmix_aug=True, conf_thres=1e9, type_unsuper="ce",
                          SSL_lambda=1.0, is_ema=True, ema_decay=0.999, eman=False, stage1_iteration=0, network="effinetv2", num_classes=9,
                          exp_name="t", date="t", left_ratio=0.95, save_feat=False, autotune=False)
    agent = SSLAgent(cfg)
    data, ulb = synthetic_batches(4, 8, size=64, seed=rank, rotate_aug=True)
    data = {k: v.cuda() for k, v in data.items()}
    ulb = {k: v.cuda() for k, v in ulb.items()}
    for step in range(2):
        agent.train_func(data, ulb)
        agent.clock.tick()
    agent.train_func_s1(data)
    torch.cuda.synchronize()
    fb = agent.optimizer.flat_buffers[0]
    torch.save({"params": fb.data.cpu(), "grads": fb.grad.cpu(), "ema": agent._ema_flat.data.cpu(), "bn": agent._bn_flat_all.data.cpu(),
                "buckets": None if agent._buckets is None else list(agent._buckets.bounds)}, os.path.join(out, f"rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
