"""Step time of the other BASELINE configs on one GPU (parity-test cases, not bench lines):
    python tools/step_bench.py repvgg 64 256      # C3 shape
    python tools/step_bench.py tinyvit 32 128     # C5 per-GPU shape
    python tools/step_bench.py effinetv2 32 128 1 1   # [warm-up steps] [timed steps]: a short run for rocprofv3 --pmc passes
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from semiuhpe_amd.src.agent import SSLAgent
from semiuhpe_amd.synthetic import synthetic_batches

net, B, muB = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
warm, n = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (3, 8)
cfg = bench.make_config("/tmp/suhpe_stepbench")
cfg.network = net
full = net == "repvgg"
cfg.is_full_range, cfg.rotate_aug = full, True
cfg.train_labeled = "DAD3DHeads" if full else "300WLP"
agent = SSLAgent(cfg)
data, ulb = synthetic_batches(B, muB, size=224, seed=0, full_range=full, rotate_aug=True)
data = {k: v.cuda() for k, v in data.items()}; ulb = {k: v.cuda() for k, v in ulb.items()}
cfg.conf_thres = float(agent.compute_dynamic_entropy_threshold([ulb]))
for _ in range(warm):
    agent.train_func(data, ulb)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    fd, fu, out = agent.train_func(data, ulb)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"{net} B={B}+{muB}: {dt*1e3:.1f} ms/step, {(B+muB)/dt:.1f} img/s, loss_all {float(out['loss_all']):.4f}, mask_ratio {float(fu['mask_ratio']):.3f}, "
      f"mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
