"""Which Python lines put torch copy / fill / elementwise kernels into a steady-state training step?  (~150
__amd_rocclr_copyBuffer + a few at::native kernels per step.)  One C2 step under a TorchDispatchMode that records, for every aten op
that touches a CUDA tensor, the innermost frames inside this repo.
    python tools/debug/trace_copies.py [network=effinetv2]"""
import collections, os, sys, traceback
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from semiuhpe_amd.src.agent import SSLAgent

net = sys.argv[1] if len(sys.argv) > 1 else "effinetv2"
cfg = bench.make_config("/tmp/suhpe_trace")
cfg.network = net
agent = SSLAgent(cfg)
data, ulb = bench.synthetic(0, torch.device("cuda", 0), 32, 128)
cfg.conf_thres = float(agent.compute_dynamic_entropy_threshold([ulb]))
for _ in range(3):
    agent.train_func(data, ulb)
torch.cuda.synchronize()
sites = collections.Counter()


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        flat = [a for a in list(args) + list((kwargs or {}).values()) if torch.is_tensor(a)]
        outs = [o for o in (out if isinstance(out, (tuple, list)) else [out]) if torch.is_tensor(o)]
        if any(t.is_cuda for t in flat + outs) and not any(k in name for k in ("aten.view", "aten.detach", "aten.alias", "aten._unsafe_view", "aten.as_strided",
                                                                              "aten.slice", "aten.select", "aten.t.", "aten.expand", "aten.unsqueeze", "aten.squeeze",
                                                                              "aten.empty", "aten.reshape", "aten.permute", "aten.transpose", "aten.split", "aten.unbind",
                                                                              "aten._reshape_alias", "aten.lift_fresh", "aten.record_stream", "aten.is_pinned")):
            fr = [f for f in traceback.extract_stack() if "/semiuhpe_amd/" in f.filename or f.filename.endswith("bench.py")]
            where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr[-3:]))
            srcdev = "cpu->cuda " if (any(not t.is_cuda for t in flat) and any(t.is_cuda for t in outs + flat)) else ""
            sites[(name.replace("aten.", ""), srcdev + where)] += 1
        return out


# custom autograd Functions run their backward on the autograd engine's thread, where this thread's mode is not active: wrap them
import torch.autograd.function as AF
from semiuhpe_amd import layers as L
from semiuhpe_amd.src import agent as AG
from semiuhpe_amd.src.fisher import fisher_utils as FU
for mod in (L, AG, FU):
    for name in dir(mod):
        cls = getattr(mod, name)
        if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function:
            orig = cls.backward

            def wrapped(ctx, *g, _orig=orig):
                with Mode():
                    return _orig(ctx, *g)
            cls.backward = staticmethod(wrapped)
with Mode():
    agent.train_func(data, ulb)
torch.cuda.synchronize()
tot = sum(sites.values())
print(f"{tot} torch ops on CUDA tensors in one {net} step (views excluded):")
for (op, where), c in sites.most_common(60):
    print(f"  x{c:4d}  {op:28s} {where}")
