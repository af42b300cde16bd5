"""Generate golden vectors for the matrix-Fisher head, Euler conversion, tensor augmentations and
the dynamic threshold by IMPORTING the reference (read-only, /root/reference) in the build container.

    python tools/make_golden_head.py            # writes tests/golden/{head,euler,augs,threshold}.npz

The fixtures hold inputs and the reference's outputs only (float32 as the reference computes them,
plus the float64 inputs used to create them). No reference source is stored.
Fixture groups follow SURVEY.md §8c: G1 head, G2 rotation adjust, G3 Euler, G4 augs, G7 threshold.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shims  # noqa: E402

ref_shims.install()

from src.fisher import fisher_utils  # noqa: E402  (reference)
import torch_norm_factor  # noqa: E402  (reference, top-level after fisher_utils tweaks sys.path)
from src import utils as ref_utils  # noqa: E402
from src import augments as ref_aug  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def rand_rot(n, gen):
    q, r = torch.linalg.qr(torch.randn(n, 3, 3, generator=gen, dtype=torch.float64))
    q = q * torch.sign(torch.diagonal(r, dim1=1, dim2=2))[:, None, :]
    det = torch.det(q)
    q[:, :, 2] *= det[:, None]
    return q.contiguous()


def head_case(A, R, A1):
    """A: student output (n,9) f32; R: gt rotation (n,3,3) f32; A1: teacher output (n,9) f32."""
    out = {}
    A_ = A.clone().requires_grad_(True)
    loss, Rest = fisher_utils.vmf_loss(A_, R, overreg=1.025)
    (dA,) = torch.autograd.grad(loss.sum(), A_)
    out["nll"] = loss.detach().numpy()
    out["nll_dA"] = dA.numpy()
    out["Rest"] = Rest.detach().numpy()
    out["entropy"] = fisher_utils.fisher_entropy(A.clone()).detach().numpy()
    out["log_pdf"] = fisher_utils.fisher_log_pdf(A.view(-1, 3, 3).clone(), R).detach().numpy()
    A2 = A.clone().requires_grad_(True)
    ce = fisher_utils.fisher_CE(A1.clone(), A2)
    (dA2,) = torch.autograd.grad(ce.sum(), A2)
    out["ce"] = ce.detach().numpy()
    out["ce_dA2"] = dA2.numpy()
    # normaliser on the signed singular values
    U, S, V = torch.svd(A.view(-1, 3, 3))
    s3 = torch.det(U @ V.transpose(1, 2))
    Ss = torch.cat((S[:, :2], S[:, 2:] * s3[:, None]), -1).clone().requires_grad_(True)
    lc = torch_norm_factor.logC_F(Ss)
    (dS,) = torch.autograd.grad(lc.sum(), Ss)
    out["S_signed"] = Ss.detach().numpy()
    out["logC"] = lc.detach().numpy()
    out["logC_dS"] = dS.numpy()
    return out


def gen_head():
    store = {}
    names = []
    n = 16
    for seed in (0, 1, 2):
        for scale in (0.1, 1.0, 5.0, 20.0, 60.0, 150.0):
            g = torch.Generator().manual_seed(1000 * seed + int(scale * 10))
            A = (torch.randn(n, 9, generator=g, dtype=torch.float64) * scale).float()
            A1 = (torch.randn(n, 9, generator=g, dtype=torch.float64) * scale).float()
            R = rand_rot(n, g).float()
            name = f"s{seed}_x{scale:g}"
            names.append(name)
            store[name + "/A"], store[name + "/A1"], store[name + "/R"] = A.numpy(), A1.numpy(), R.numpy()
            for k, v in head_case(A, R, A1).items():
                store[name + "/" + k] = v
    # "trained-like" cases: A = k * R_true + noise (what a converged net emits; det>0, well separated)
    for seed, k in ((7, 3.0), (8, 15.0), (9, 40.0)):
        g = torch.Generator().manual_seed(seed)
        Rt = rand_rot(n, g)
        A = (k * Rt + 0.3 * k * torch.randn(n, 3, 3, generator=g, dtype=torch.float64)).reshape(n, 9).float()
        A1 = (k * Rt + 0.3 * k * torch.randn(n, 3, 3, generator=g, dtype=torch.float64)).reshape(n, 9).float()
        R = Rt.float()
        name = f"trained_k{k:g}"
        names.append(name)
        store[name + "/A"], store[name + "/A1"], store[name + "/R"] = A.numpy(), A1.numpy(), R.numpy()
        for kk, v in head_case(A, R, A1).items():
            store[name + "/" + kk] = v
    # special: forced det<0 (mirror one row) on a well-conditioned matrix
    g = torch.Generator().manual_seed(77)
    A = torch.randn(n, 3, 3, generator=g, dtype=torch.float64) * 4
    sgn = torch.sign(torch.det(A))
    A[:, 0, :] *= -sgn[:, None]  # now every det < 0
    A1 = torch.randn(n, 3, 3, generator=g, dtype=torch.float64) * 4
    R = rand_rot(n, g).float()
    A, A1 = A.reshape(n, 9).float(), A1.reshape(n, 9).float()
    name = "detneg"
    names.append(name)
    store[name + "/A"], store[name + "/A1"], store[name + "/R"] = A.numpy(), A1.numpy(), R.numpy()
    for kk, v in head_case(A, R, A1).items():
        store[name + "/" + kk] = v
    store["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "head.npz"), **store)
    print("head.npz:", len(names), "cases")

    # forward-only degenerate cases (gradients through coincident singular values are ill-defined in the
    # reference as well; only values are pinned): near-equal s1≈s2, one zero singular value, all-zero.
    g = torch.Generator().manual_seed(5)
    U, V = rand_rot(8, g), rand_rot(8, g)
    svals = torch.tensor([[3.0, 3.0 - 1e-3, 1.0], [5.0, 2.0, 2.0 - 1e-3], [4.0, 2.0, 0.0], [2.0, 1.0, -1.0 + 1e-3],
                          [10.0, 10.0 - 1e-2, 0.5], [1.0, 0.5, 0.0], [8.0, 3.0, -1.0], [0.5, 0.25, 0.125]],
                         dtype=torch.float64)
    A = (U @ torch.diag_embed(svals) @ V.transpose(1, 2)).reshape(8, 9).float()
    deg = {"A": A.numpy(), "svals": svals.numpy()}
    deg["entropy"] = fisher_utils.fisher_entropy(A.clone()).detach().numpy()
    deg["Rest"] = fisher_utils.batch_torch_A_to_R(A.clone()).numpy()
    R = rand_rot(8, g).float()
    deg["R"] = R.numpy()
    deg["nll"] = fisher_utils.vmf_loss(A.clone(), R, overreg=1.025)[0].numpy()
    np.savez_compressed(os.path.join(OUT, "head_degenerate.npz"), **deg)


def gen_rot_adjust():
    """G2: capture `pred_weak_adjusted` of agent.forward (agent.py:110-122) for both branches by running the
    reference agent with a stub network and recording the first argument of fisher_CE."""
    import src.agent as ag
    import tempfile

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.randn(3 * 8 * 8, 9) * 0.5)

        def forward(self, x):
            return x[:, :, :8, :8].reshape(x.shape[0], -1) @ self.w

    store = {}
    for branch in ("DAD3DHeads", "300WLP"):
        torch.manual_seed(11)
        ag.get_network = lambda cfg: Stub()
        tmp = tempfile.mkdtemp()
        cfg = SimpleNamespace(lr=1e-4, log_dir=tmp, model_dir=tmp, distribution="matrixFisher", is_full_range=False,
                              rotate_aug=True, train_labeled=branch, train_unlabeled="X", cutout_aug=False,
                              cutmix_aug=False, conf_thres=1e9, type_unsuper="ce", SSL_lambda=1.0, is_ema=True,
                              ema_decay=0.999, eman=False, stage1_iteration=0, network="stub", num_classes=9,
                              exp_name="x", date="x", left_ratio=0.95, save_feat=False)
        agent = ag.SSLAgent(cfg)
        rec = {}
        orig_ce = ag.fisher_CE

        def spy(a1, a2, rec=rec, orig=orig_ce):
            rec["adj"] = a1.detach().clone()
            return orig(a1, a2)

        ag.fisher_CE = spy
        g = torch.Generator().manual_seed(3)
        B, M = 4, 12
        ang = torch.round((torch.rand(M, generator=g) * 60 - 30) * 1000) / 1000
        rad = ang * np.pi / 180
        aug = torch.zeros(M, 3, 3)
        aug[:, 0, 0], aug[:, 0, 1] = torch.cos(rad), torch.sin(rad)
        aug[:, 1, 0], aug[:, 1, 1] = -torch.sin(rad), torch.cos(rad)
        aug[:, 2, 2] = 1
        data = dict(img=torch.randn(B, 3, 16, 16, generator=g), rot_mat=rand_rot(B, g).float())
        ulb = dict(img=torch.randn(M, 3, 16, 16, generator=g), img_strong=torch.randn(M, 3, 16, 16, generator=g),
                   rot_mat=rand_rot(M, g).float(), aug_rot_mat=aug)
        with torch.no_grad():
            pred_weak = agent.ema_net(ulb["img"]).clone()
        agent.net.train(); agent.ema_net.train()
        agent.forward(data, ulb)
        ag.fisher_CE = orig_ce
        agent.logs_writer.close()
        store[branch + "/pred_weak"] = pred_weak.numpy()
        store[branch + "/aug"] = aug.numpy()
        store[branch + "/adjusted"] = rec["adj"].numpy()
    np.savez_compressed(os.path.join(OUT, "rot_adjust.npz"), **store)
    print("rot_adjust.npz done")


def gen_euler():
    g = torch.Generator().manual_seed(21)
    n = 64
    R = rand_rot(n, g).float()
    # singular: sy < 1e-6  <=>  R00 = R10 = 0 (yaw = +-90 deg)
    for i, (p, y, r) in enumerate([(0.3, np.pi / 2, 0.0), (-0.7, -np.pi / 2, 0.4), (1.0, np.pi / 2, -1.2)]):
        R[i] = torch.from_numpy(ref_utils.get_6DRepNet_Rot(p, y, r)).float()
    store = {"R": R.numpy()}
    for fr in (False, True):
        e = ref_utils.compute_euler_angles_from_rotation_matrices(R.clone(), full_range=fr, use_gpu=False)
        store[f"euler_full{int(fr)}"] = e.numpy()
    ang = (torch.rand(32, 3, generator=g, dtype=torch.float64) - 0.5) * torch.tensor([2.0, 3.0, 2.0])
    Rs = np.stack([ref_utils.get_6DRepNet_Rot(*a.tolist()) for a in ang])
    store["angles_xyz"] = ang.numpy()
    store["Rot_from_angles"] = Rs
    # geodesic error (agent.compute_err_deg_from_matrices with gt_euler=None)
    import src.agent as ag
    R2 = rand_rot(n, g).float()
    R2[:8] = R[:8]                      # identical pairs -> extrapolated acos branch
    store["R2"] = R2.numpy()
    store["geodesic_deg"] = ag.SSLAgent.compute_err_deg_from_matrices(R, R2).numpy()
    ge = (torch.rand(n, 3, generator=g) - 0.5) * 120
    store["gt_euler"] = ge.numpy()
    store["euler_err_deg"] = ag.SSLAgent.compute_err_deg_from_matrices(R, R2, gt_euler=ge).numpy()
    np.savez_compressed(os.path.join(OUT, "euler.npz"), **store)
    print("euler.npz done")


def gen_augs():
    store = {}
    for seed in (0, 5):
        for shape in ((6, 3, 48, 48), (12, 3, 224, 224)):
            g = torch.Generator().manual_seed(100 + seed)
            img = torch.randn(*shape, generator=g)
            tag = f"seed{seed}_{shape[0]}x{shape[2]}"
            torch.manual_seed(seed)
            co = ref_aug.random_cutout_tensor(img.clone(), 3, normal=True)
            torch.manual_seed(seed)
            cm = ref_aug.random_cutmix_tensor(img.clone(), 3, normal=True)
            torch.manual_seed(seed)
            both = ref_aug.random_cutmix_tensor(ref_aug.random_cutout_tensor(img.clone(), 3, normal=True), 3, normal=True)
            torch.manual_seed(seed)
            uni = ref_aug.random_cutout_tensor(img.clone(), 3, normal=False) if shape[2] > 48 else None
            if shape[2] <= 48:
                store[tag + "/cutout"], store[tag + "/cutmix"], store[tag + "/both"] = co.numpy(), cm.numpy(), both.numpy()
            for nm, t in (("cutout", co), ("cutmix", cm), ("both", both)) + ((("uniform", uni),) if uni is not None else ()):
                d = t.double()
                store[tag + f"/{nm}_sum"] = d.sum((1, 2, 3)).numpy()
                store[tag + f"/{nm}_abs"] = d.abs().sum((1, 2, 3)).numpy()
                store[tag + f"/{nm}_zeros"] = (t == 0).sum((1, 2, 3)).numpy()
                w = torch.arange(d[0].numel(), dtype=torch.float64).reshape(d[0].shape) % 977
                store[tag + f"/{nm}_wsum"] = (d * w).sum((1, 2, 3)).numpy()
    np.savez_compressed(os.path.join(OUT, "augs.npz"), **store)
    print("augs.npz done")


def gen_threshold():
    """G7: the sort-and-pick of agent.compute_dynamic_entropy_threshold (agent.py:403-407), run through the
    reference agent with a stub teacher."""
    import src.agent as ag
    import tempfile

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.randn(3 * 4 * 4, 9) * 2.0)

        def forward(self, x):
            return x[:, :, :4, :4].reshape(x.shape[0], -1) @ self.w

    torch.manual_seed(5)
    ag.get_network = lambda cfg: Stub()
    tmp = tempfile.mkdtemp()
    store = {}
    for lr in (0.75, 0.95):
        cfg = SimpleNamespace(lr=1e-4, log_dir=tmp, model_dir=tmp, distribution="matrixFisher", is_full_range=False,
                              conf_thres=-5.3, left_ratio=lr, save_feat=False, network="stub", num_classes=9)
        torch.manual_seed(5)
        agent = ag.SSLAgent(cfg)
        g = torch.Generator().manual_seed(9)
        batches = [dict(img=torch.randn(16, 3, 8, 8, generator=g)) for _ in range(5)]
        with torch.no_grad():
            preds = torch.cat([agent.ema_net(b["img"]) for b in batches])
        thr = agent.compute_dynamic_entropy_threshold(batches)
        agent.logs_writer.close()
        store[f"lr{lr}/preds"] = preds.numpy()
        store[f"lr{lr}/thr"] = np.float32(thr)
        store[f"lr{lr}/conf_thres_after"] = np.float32(cfg.conf_thres)
    np.savez_compressed(os.path.join(OUT, "threshold.npz"), **store)
    print("threshold.npz done")


if __name__ == "__main__":
    torch.set_num_threads(4)
    gen_head()
    gen_rot_adjust()
    gen_euler()
    gen_augs()
    gen_threshold()
