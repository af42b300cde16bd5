"""Upper bound of merging the two student passes (labeled 32 + strong 128) into one batch of 160: forward + backward time
of the separate calls against one call (plain BatchNorm over the merged batch: timing only, not the reference's semantics)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from semiuhpe_amd import kernels as K
from semiuhpe_amd.src.networks import get_network

net_name = sys.argv[1] if len(sys.argv) > 1 else "effinetv2"
cfg = bench.make_config("/tmp/suhpe_mergeprobe")
cfg.network = net_name
K.AUTOTUNE = True
net = get_network(cfg).cuda().train()
xa, xb = torch.randn(32, 3, 224, 224, device="cuda"), torch.randn(128, 3, 224, 224, device="cuda")
xc = torch.cat([xa, xb])


def run(fn, n=6):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


def separate():
    (net(xa).sum() + net(xb).sum()).backward()


def merged():
    net(xc).sum().backward()


print(f"{net_name}: separate 32 + 128: {run(separate):.1f} ms   merged 160: {run(merged):.1f} ms", flush=True)
