"""Achieved HBM rate of the memory-bound wrappers over one training step (which ones sit far from the ~6 TB/s roof):
    python tools/mem_profile.py [effinetv2|repvgg|tinyvit] [B muB]
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from semiuhpe_amd import kernels as K
from semiuhpe_amd.src.agent import SSLAgent
from semiuhpe_amd.synthetic import synthetic_batches

net = sys.argv[1] if len(sys.argv) > 1 else "effinetv2"
B, muB = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (32, 128)
cfg = bench.make_config("/tmp/suhpe_memprof")
cfg.network = net
agent = SSLAgent(cfg)
data, ulb = synthetic_batches(B, muB, size=224, seed=0, rotate_aug=True)
data = {k: v.cuda() for k, v in data.items()}; ulb = {k: v.cuda() for k, v in ulb.items()}
cfg.conf_thres = float(agent.compute_dynamic_entropy_threshold([ulb]))
for _ in range(3):
    agent.train_func(data, ulb)
torch.cuda.synchronize()
K.MEM_PROFILE = []
nprof = 2
for _ in range(nprof):
    agent.train_func(data, ulb)
torch.cuda.synchronize()
recs, K.MEM_PROFILE = K.MEM_PROFILE, None
byname, byshape = {}, {}
for name, nbytes, shape, e0, e1 in recs:
    dt = e0.elapsed_time(e1) * 1e-3
    for agg, key in ((byname, name), (byshape, (name, shape))):
        a = agg.setdefault(key, [0.0, 0.0, 0])
        a[0] += dt; a[1] += nbytes; a[2] += 1
print("per wrapper (event-bracketed, includes launch gaps of multi-kernel wrappers):")
for k, (secs, nb, cnt) in sorted(byname.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:24s} x{cnt//nprof:4d} {secs/nprof*1e3:8.3f} ms/step {nb/secs/1e9:8.0f} GB/s  ideal@6TB/s {nb/nprof/6e12*1e3:7.3f} ms")
print("worst shapes by lost time:")
rows = sorted(((secs - nb / 6e12, k, secs, nb, cnt) for k, (secs, nb, cnt) in byshape.items()), reverse=True)
for lost, k, secs, nb, cnt in rows[:int(os.environ.get("TOP", 40))]:
    print(f"{k[0]:20s} {str(k[1]):28s} x{cnt//nprof:3d} {secs/nprof*1e3:7.3f} ms {nb/secs/1e9:7.0f} GB/s lost {lost/nprof*1e3:6.3f} ms  ({secs/cnt*1e6:6.1f} us each)")
