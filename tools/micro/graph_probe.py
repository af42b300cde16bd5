"""Does hipGraph replay shorten a launch-dense kernel sequence on this stack?  Captures (a) the eval-mode forward of EfficientNetV2-S
(about 150 launches, no host-side randomness) and (b) the train-mode forward without a tape (teacher pass shape: ~420 launches incl.
statistics finalisers) with torch.cuda.CUDAGraph and compares replay with eager launches of the same sequence.
    python tools/micro/graph_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from semiuhpe_amd import kernels as K, layers as L
from semiuhpe_amd.src.efficientnet_v2 import EfficientNetV2S

torch.manual_seed(0)
net = EfficientNetV2S().cuda()
x = torch.randn(128, 3, 224, 224, device="cuda")
K.AUTOTUNE = True


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for mode in ("eval", "train"):
    net.train(mode == "train")
    ones = {}
    def hook(kind, shape, keep):                         # no host RNG inside the capture: constant keep-everything masks
        if shape not in ones:
            ones[shape] = torch.ones(shape, device="cuda")
        return ones[shape]
    L.set_mask_hook(hook if mode == "train" else None)
    with torch.no_grad():
        for _ in range(3):
            y = net(x)
        torch.cuda.synchronize()
        eager = timed(lambda: net(x))
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            net(x)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            yg = net(x)
        replay = timed(g.replay)
    print(f"{mode}: eager {eager:.3f} ms, graph replay {replay:.3f} ms ({(1 - replay / eager) * 100:.1f} % shorter)", flush=True)
    L.set_mask_hook(None)
