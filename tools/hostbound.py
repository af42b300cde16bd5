"""Host-side cost of one SSL step: run the C2 step on 32x32 images (GPU work ~0) so that step time ~ Python + launch cost."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from semiuhpe_amd.src.agent import SSLAgent
from semiuhpe_amd.synthetic import synthetic_batches

cfg = bench.make_config("/tmp/suhpe_hostbound")
agent = SSLAgent(cfg)
cfg.conf_thres = 1e9
for size in (32, 224):
    data, ulb = synthetic_batches(32, 128, size=size, seed=0)
    data = {k: v.cuda() for k, v in data.items()}; ulb = {k: v.cuda() for k, v in ulb.items()}
    for _ in range(3):
        agent.train_func(data, ulb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        agent.train_func(data, ulb)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"size {size}: enqueue {1e3*(t1-t0)/10:.1f} ms/step, with sync {1e3*(t2-t0)/10:.1f} ms/step", flush=True)
