"""debug: TinyViT agent step, HIP vs fp32 oracle gradients under (merged passes, teacher overlap) on / off"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_agent_variants as T
from oracle import networks as ON
from parity_utils import MaskTape
from semiuhpe_amd import layers as L, kernels as K

net = sys.argv[1] if len(sys.argv) > 1 else "tinyvit"
size = 224 if net == "tinyvit" else 96
for merge, overlap, wgrad_overlap in ((True, True, True), (True, False, True), (True, False, False), (False, False, False)):
    oracle, agent, cfg_o, cfg_h = T._pair(network=net, rotate_aug=True, train_labeled="300WLP")
    agent.merge_student_passes, agent.overlap_teacher = merge, overlap
    K.WGRAD_OVERLAP = wgrad_overlap
    data, ulb = T._batches(3, 6, size, 0)
    tape = MaskTape(100)
    ON.set_mask_hook(tape.record)
    torch.manual_seed(50)
    oracle.iteration = 0
    fd_o, fu_o, out_o = oracle.train_func({k: v.clone() for k, v in data.items()}, {k: v.clone() for k, v in ulb.items()})
    ON.set_mask_hook(None)
    L.set_mask_hook(tape.replay_merged if merge else tape.replay)
    torch.manual_seed(50)
    agent.clock.iteration = 0
    fd, fu, out = agent.train_func({k: v.cuda() for k, v in data.items()}, {k: v.cuda() for k, v in ulb.items()})
    L.set_mask_hook(None)
    torch.cuda.synchronize()
    g_ref = {n: p.grad.clone() for n, p in oracle.net.named_parameters()}
    errs = []
    for n, p in agent.net.named_parameters():
        r = g_ref[n].double()
        errs.append((float((p.grad.double().cpu() - r).abs().max() / r.abs().max().clamp_min(1e-20)), n))
    errs.sort(reverse=True)
    import statistics
    print(f"merge={merge} overlap={overlap} wgrad_overlap={wgrad_overlap}: pred err {T.rel_err(fd['pred'], fd_o['pred']):.2e} loss_all {float(out['loss_all']):.6f} vs {float(out_o['loss_all']):.6f} "
          f"median grad err {statistics.median(e for e, _ in errs):.2e}, worst {errs[0][0]:.2e} {errs[0][1]}; head.7.weight {[e for e, n in errs if n == ('head.7.weight' if net == 'tinyvit' else 'classifier.7.weight')]}", flush=True)
