"""Worker of tests/test_gpu_syncbn.py: one of two ranks (gloo over CUDA tensors, both on GPU 0) runs a train-mode forward/backward of
a small conv + BatchNorm stack on ITS half of a batch with layers.SYNC_BN on, and writes outputs / gradients / running statistics."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from semiuhpe_amd import kernels as K, layers as L
from semiuhpe_amd.src import distributed as D
from semiuhpe_amd.src.resnet import Bottleneck


def build():
    torch.manual_seed(3)
    down = L.Sequential(L.Conv2d(32, 64, 1, 2), L.BatchNorm2d(64))
    blk = Bottleneck(32, 16, 2, down)
    for m in blk.modules():
        if isinstance(m, L.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
    return blk.cuda().train()


def data():
    g = torch.Generator().manual_seed(9)
    return torch.randn(8, 12, 12, 32, generator=g).cuda(), torch.randn(8, 6, 6, 64, generator=g).cuda()


def run(x, dy, blk):
    tape = []
    y = blk.fwd(x.contiguous(), tape)
    dx = blk.bwd(dy.contiguous(), tape)
    return y, dx


if __name__ == "__main__":
    out = sys.argv[1]
    torch.cuda.set_device(0)
    rank, world = D.init_from_env("gloo")
    L.SYNC_BN = world > 1
    blk = build()
    x, dy = data()
    lo, hi = rank * 4, rank * 4 + 4
    y, dx = run(x[lo:hi], dy[lo:hi], blk)
    grads = {n: p.grad.clone() for n, p in blk.named_parameters()}
    for g in grads.values():
        D.allreduce_sum_(g)                                   # what the flat-gradient all-reduce does in the agent
    torch.save({"y": y.cpu(), "dx": dx.cpu(), "grads": {k: v.cpu() for k, v in grads.items()},
                "buffers": {k: v.cpu() for k, v in blk.named_buffers()}}, os.path.join(out, f"rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
