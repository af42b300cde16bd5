"""Crash sweep over configuration variants of the agent step (backbone x distribution x unsupervised loss x EMAN x rotation
consistency): python tools/variant_sweep.py   (needs an MI355X)"""
import sys, os, itertools
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from semiuhpe_amd.src.agent import SSLAgent
from semiuhpe_amd.synthetic import synthetic_batches
for net, dist_, unsup, eman, rot in [("effinetv2","matrixFisher","nll",True,True), ("repvgg","RotationLaplace","ce",False,True), ("tinyvit","RotationLaplace","nll",True,False),
                                     ("effinetv2","matrixFisher","ce",False,True), ("resnet50","matrixFisher","ce",False,True)]:
    cfg = bench.make_config("/tmp/suhpe_var"); cfg.network = net; cfg.distribution = dist_; cfg.type_unsuper = unsup; cfg.eman = eman; cfg.rotate_aug = rot
    cfg.autotune = False
    full = net == "repvgg"; cfg.is_full_range = full; cfg.train_labeled = "DAD3DHeads" if full else "300WLP"
    a = SSLAgent(cfg)
    data, ulb = synthetic_batches(6, 12, size=224 if net=="tinyvit" else 96, seed=0, full_range=full, rotate_aug=True)
    data = {k: v.cuda() for k, v in data.items()}; ulb = {k: v.cuda() for k, v in ulb.items()}
    cfg.conf_thres = 1e9
    for i in range(3):
        fd, fu, out = a.train_func(data, ulb); a.clock.tick()
    s1 = a.train_func_s1(data)
    v = a.val_func(data)
    torch.cuda.synchronize(); a.check_finite()
    print(net, dist_, unsup, eman, "loss_all", float(out["loss_all"]), "s1 loss", float(s1["loss"]), flush=True)
print("VARIANTS OK")
