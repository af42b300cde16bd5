"""Throughput of the eval-mode (no tape) forward: the teacher's threshold scan / validation pass.
    python tools/eval_bench.py [effinetv2|repvgg|tinyvit] [batch]
BatchNorm, activation and skip adds ride in the conv epilogues; RepVGG blocks run re-parameterised (one 3x3 conv each)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from semiuhpe_amd.src.networks import get_network
from semiuhpe_amd.src.fisher.fisher_utils import fisher_entropy

net_name = sys.argv[1] if len(sys.argv) > 1 else "effinetv2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = bench.make_config("/tmp/suhpe_evalbench")
cfg.network = net_name
net = get_network(cfg).cuda().eval()
x = torch.randn(B, 3, 224, 224, device="cuda")
from semiuhpe_amd import kernels as K
K.AUTOTUNE = True
with torch.no_grad():
    for _ in range(3):
        ent = fisher_entropy(net(x))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        ent = fisher_entropy(net(x))
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"{net_name} eval forward + entropy, batch {B}: {dt*1e3:.1f} ms = {B/dt:,.0f} img/s", flush=True)
