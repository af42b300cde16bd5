"""Per-shape time of the dense conv / linear launches of one C2 step (which layers lose the most against the MFMA /
HBM bound):   python tools/shape_profile.py [effinetv2|repvgg|tinyvit] [B muB]
columns: pass (0 fwd, 1 dgrad, 2 wgrad), n h w cin cout r stride groups | launches/step | ms/step | TF/s | GB/s | lost ms
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
cfg = bench.make_config("/tmp/suhpe_shapeprof")
cfg.network = net
agent = SSLAgent(cfg)
if os.environ.get("SERIAL") == "1":                         # one stream: the times are the kernels' own
    agent.overlap_teacher, K.WGRAD_OVERLAP = False, False
data, ulb = synthetic_batches(B, muB, size=224, seed=0, rotate_aug=True)
data = {k: v.cuda() for k, v in data.items()}; ulb = {k: v.cuda() for k, v in ulb.items()}
cfg.conf_thres = float(agent.compute_dynamic_entropy_threshold([ulb]))
for _ in range(3):
    agent.train_func(data, ulb)
torch.cuda.synchronize()
K.PROFILE = []
nprof = 3
for _ in range(nprof):
    agent.train_func(data, ulb)
torch.cuda.synchronize()
recs, K.PROFILE = K.PROFILE, None
agg = {}
for name, flops, nbytes, e0, e1, shape in recs:
    a = agg.setdefault((shape, name), [0.0, 0.0, 0.0, 0])
    a[0] += e0.elapsed_time(e1) * 1e-3; a[1] += flops; a[2] += nbytes; a[3] += 1
rows = []
for (shape, name), (secs, flops, nbytes, cnt) in agg.items():
    ideal = max(flops / 157.3e12, nbytes / 6.0e12)
    rows.append((secs - ideal, shape, name, secs, flops, nbytes, cnt))
rows.sort(reverse=True)
tot = sum(r[3] for r in rows) / nprof * 1e3
print(f"total dense GEMM time {tot:.2f} ms/step over {sum(r[6] for r in rows)//nprof} launches")
for lost, shape, name, secs, flops, nbytes, cnt in rows[:int(os.environ.get('TOP', 45))]:
    print(f"{str(shape):48s} {name[6:40]:34s} x{cnt//nprof:3d} {secs/nprof*1e3:7.3f} ms {flops/secs/1e12:6.1f} TF {nbytes/secs/1e9:7.0f} GB/s lost {lost/nprof*1e3:6.3f}")
