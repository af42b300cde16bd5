"""Worker of tests/test_gpu_train_eval.py::test_ranks_agree_on_measured_tiles: one of two data-parallel ranks (gloo over CUDA tensors, both on
GPU 0) with the autotuner ON and an EMPTY tile database, i.e. every shape of the step is "a batch size absent from tiles_mi355x.json".  Runs two
SSL steps and one supervised step and writes the tile table it ends with, the kernel the library reports for every key of that table, and its
parameters."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from types import SimpleNamespace
import torch
from semiuhpe_amd import _lib, kernels as K
from semiuhpe_amd.src import distributed as D
from semiuhpe_amd.src.agent import SSLAgent
from semiuhpe_amd.synthetic import synthetic_batches

if __name__ == "__main__":
    out = sys.argv[1]
    torch.cuda.set_device(0)
    rank, world = D.init_from_env("gloo")
    torch.manual_seed(11)
    cfg = SimpleNamespace(lr=1e-4, log_dir=out, model_dir=out, distribution="matrixFisher", is_full_range=False, rotate_aug=False,
                          train_labeled="300WLP", train_unlabeled="300WLP", cutout_aug=True, cutmix_aug=True, conf_thres=1e9, type_unsuper="ce",
                          SSL_lambda=1.0, is_ema=True, ema_decay=0.999, eman=False, stage1_iteration=0, network="effinetv2", num_classes=9,
                          exp_name="t", date="t", left_ratio=0.95, save_feat=False, autotune=True)
    agent = SSLAgent(cfg)
    assert K.AUTOTUNE and K.tile_table() == {}, "the test needs the autotuner on and an empty database (SUHPE_AUTOTUNE=1, SUHPE_TILE_DB=<empty>)"
    data, ulb = synthetic_batches(8, 16, size=112, seed=rank, rotate_aug=False)
    data = {k: v.cuda() for k, v in data.items()}
    ulb = {k: v.cuda() for k, v in ulb.items()}
    for step in range(2):
        agent.train_func(data, ulb)
        agent.clock.tick()
    agent.train_func_s1(data)
    torch.cuda.synchronize()
    lib = _lib.load()
    table = K.tile_table()
    running = {}
    for key in table:
        v = [int(x) for x in key.split(",")]
        pass_id, (n, h, w, cin, cout, r, s, stride, pad, groups) = v[0], v[1:]
        d = _lib.ConvDesc(n, h, w, cin, (h + 2 * pad - r) // stride + 1, (w + 2 * pad - s) // stride + 1, cout, r, s, stride, pad, groups)
        running[key] = lib.suhpe_conv2d_kernel_name(ctypes.byref(d), pass_id).decode()
    fb = agent.optimizer.flat_buffers[0]
    torch.save({"table": table, "running": running, "params": fb.data.cpu(), "ema": agent._ema_flat.data.cpu(), "bn": agent._bn_flat_all.data.cpu(),
                "agreed": sorted(agent._tiles_agreed)}, os.path.join(out, f"rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
