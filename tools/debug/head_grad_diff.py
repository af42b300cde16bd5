"""debug: Fisher NLL / CE gradients of the HIP head kernels vs float64 autograd of the oracle on TinyViT-like outputs"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import fisher as OF, networks as ON
from oracle.agent import synthetic_batches
from parity_utils import seeded_state_dict
from semiuhpe_amd import kernels as K

ref = ON.TinyViT21M(); ref.load_state_dict(seeded_state_dict(ref.state_dict(), 23)); ref.train()
for m in ref.modules():
    if hasattr(m, "p"):
        try: m.p = 0.0
        except Exception: pass
data, ulb = synthetic_batches(3, 6, size=224, seed=0, rotate_aug=True)
with torch.no_grad():
    A_l = ref(data["img"]); A_s = ref(ulb["img_strong"]); A_w = ref(ulb["img"])
print("|A| labeled", A_l.abs().max().item(), "sv:", torch.linalg.svdvals(A_l.reshape(-1, 3, 3)))
for scale in (1.0, 5.0):
    Al, As, Aw = (A_l * scale).double(), (A_s * scale).double(), (A_w * scale).double()
    a = Al.clone().requires_grad_(True)
    loss, _ = OF.vmf_loss(a, data["rot_mat"].double(), overreg=1.025)
    loss.sum().backward()
    l_h, dA_h, _ = K.fisher_nll(Al.float().cuda().contiguous(), data["rot_mat"].reshape(-1, 9).cuda().contiguous(), 1.025)
    print(f"scale {scale}: NLL value err {float((l_h.cpu().double() - loss.detach()).abs().max()):.2e}  dA err {float((dA_h.cpu().double() - a.grad.reshape(-1, 9)).abs().max()):.2e} of max {float(a.grad.abs().max()):.2e}")
    s = As.clone().requires_grad_(True)
    ce = OF.fisher_CE(Aw, s)
    ce.sum().backward()
    ce_h, dA2 = K.fisher_ce(Aw.float().cuda().contiguous(), As.float().cuda().contiguous(), None)
    print(f"          CE value err {float((ce_h.cpu().double() - ce.detach()).abs().max()):.2e}  dA2 err {float((dA2.cpu().double() - s.grad.reshape(-1, 9)).abs().max()):.2e} of max {float(s.grad.abs().max()):.2e}")
    # fp32 oracle for scale
    s32 = As.float().clone().requires_grad_(True)
    ce32 = OF.fisher_CE(Aw.float(), s32); ce32.sum().backward()
    print(f"          oracle32 CE dA2 err {float((s32.grad.double().reshape(-1, 9) - s.grad.reshape(-1, 9)).abs().max()):.2e}")
    print("          per-sample dA2 err", [f"{float(v):.1e}" for v in (dA2.cpu().double() - s.grad.reshape(-1, 9)).abs().max(1)[0]])
