"""Where the host spends its time enqueuing one step (cProfile over 32x32-image steps, GPU work ~0)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from semiuhpe_amd.src.agent import SSLAgent
from semiuhpe_amd.synthetic import synthetic_batches

cfg = bench.make_config("/tmp/suhpe_hostprof")
cfg.autotune = False
agent = SSLAgent(cfg)
cfg.conf_thres = 1e9
data, ulb = synthetic_batches(32, 128, size=32, seed=0)
data = {k: v.cuda() for k, v in data.items()}; ulb = {k: v.cuda() for k, v in ulb.items()}
for _ in range(3):
    agent.train_func(data, ulb)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    agent.train_func(data, ulb)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 30)
