"""CPU oracle (test infrastructure): the teacher-student step of the reference's SSLAgent (src/agent.py:27-455),
restated on torch-CPU with the oracle head (oracle/fisher.py) and oracle backbones (oracle/networks.py).

Pinned against the imported reference by tests/golden/agent_step.npz (tools/make_golden_agent.py).  Also the CPU
baseline that bench.py times on the GPU box's host cores (`cpu_baseline.kind = "port"`).
"""
import numpy as np
import torch

from . import fisher as OF
from . import laplace as OL
from . import pose_utils as OP


class OracleAgent:
    def __init__(self, config, net_factory):
        self.config = config
        self.net = net_factory()
        self.ema_net = net_factory()
        for p in self.ema_net.parameters():             # src/agent.py:34-35
            p.detach_()
        self.optimizer = torch.optim.Adam(self.net.parameters(), config.lr)     # :37
        self.iteration = 0
        self.grids = None
        if getattr(config, "distribution", "matrixFisher") == "RotationLaplace":      # :43-58: eq_grids2.npy for both ranges
            import os
            path = getattr(config, "laplace_grids", None) or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "semiuhpe_amd", "src",
                                                                          "laplace", "eq_grids2.npy")
            self.grids = torch.from_numpy(np.load(path))

    def _nll(self, pred, gt):
        """src/agent.py:78-81 / :156-162: the NLL of the configured distribution"""
        if getattr(self.config, "distribution", "matrixFisher") == "RotationLaplace":
            return OL.NLL_loss("RLaplace", pred, gt, self.grids)
        return OF.vmf_loss(pred, gt, overreg=1.025)

    # src/agent.py:61-192
    def forward(self, data, ulb_data, eval_ema=False):
        cfg = self.config
        img, gt = data["img"], data["rot_mat"]
        gt_euler = data.get("euler_angles")
        net = self.ema_net if eval_ema else self.net
        fisher_out = net(img)
        losses, pred_orth = self._nll(fisher_out, gt)
        loss = losses.mean()
        err_deg = OP.compute_err_deg_from_matrices(pred_orth, gt, gt_euler)
        fisher_dict = dict(loss=loss, pred=fisher_out, pred_orth=pred_orth, err_deg=err_deg)
        if ulb_data is None:
            return fisher_dict, None
        weak, strong = ulb_data["img"], ulb_data["img_strong"]
        ulb_gt = ulb_data.get("rot_mat")
        pred_weak = self.ema_net(weak).detach()
        if cfg.rotate_aug:
            pred_weak_adjusted = OF.rotation_adjust(pred_weak, ulb_data["aug_rot_mat"], cfg.train_labeled)
        else:
            pred_weak_adjusted = pred_weak.clone()
        if cfg.cutout_aug:
            strong = OP.random_cutout_tensor(strong, 3, normal=True)
        if cfg.cutmix_aug:
            strong = OP.random_cutmix_tensor(strong, 3, normal=True)
        pred_strong = self.net(strong)
        entropy = OF.fisher_entropy(pred_weak)
        mask = entropy < cfg.conf_thres
        mask_ratio = mask.sum() / len(mask)
        if mask_ratio > 0:
            if cfg.type_unsuper == "ce":
                ul = OF.fisher_CE(pred_weak_adjusted[mask], pred_strong[mask])
            else:
                pseudo = OF.batch_torch_A_to_R(pred_weak_adjusted[mask])
                ul, _ = self._nll(pred_strong[mask], pseudo)
            unsuper_loss = ul.mean()
        else:
            unsuper_loss = torch.zeros(())
        unsuper_loss = unsuper_loss * mask_ratio
        pseudo_all = OF.batch_torch_A_to_R(pred_weak_adjusted)
        out = dict(unsuper_loss=unsuper_loss, entropy=entropy, mask_ratio=mask_ratio, mask=mask,
                   pred_weak=pred_weak, pred_weak_adjusted=pred_weak_adjusted, pred_strong=pred_strong,
                   err_strongSuper_pseudo=OP.compute_err_deg_from_matrices(
                       OF.batch_torch_A_to_R(pred_strong.detach()[mask]), pseudo_all[mask]))
        if ulb_gt is not None:
            out["err_weakAll_gt"] = OP.compute_err_deg_from_matrices(pseudo_all, ulb_gt)
            out["err_weakPseudo_gt"] = OP.compute_err_deg_from_matrices(pseudo_all[mask], ulb_gt[mask])
        return fisher_dict, out

    # src/agent.py:277-299
    def update_ema_variables(self, is_ema, alpha, global_step):
        alpha = min(1 - 1 / (global_step + 1), alpha) if is_ema else 0
        if getattr(self.config, "eman", False):         # :285-297 EMAN: every state_dict entry (parameters AND BatchNorm buffers)
            for (k, v_main), (k_ema, v_ema) in zip(self.net.state_dict().items(), self.ema_net.state_dict().items()):
                assert k == k_ema and v_main.shape == v_ema.shape
                if "num_batches_tracked" in k_ema:
                    v_ema.copy_(v_main)
                else:
                    v_ema.copy_(v_ema * alpha + (1.0 - alpha) * v_main)
            return
        for e, p in zip(self.ema_net.parameters(), self.net.parameters()):
            e.data.mul_(alpha).add_(p.detach(), alpha=1 - alpha)

    # src/agent.py:194-218
    def train_func(self, data, ulb_data):
        self.net.train()
        self.ema_net.train()
        self.update_ema_variables(self.config.is_ema, self.config.ema_decay, self.iteration - self.config.stage1_iteration)
        fd, fu = self.forward(data, ulb_data)
        loss_all = fd["loss"] + self.config.SSL_lambda * fu["unsuper_loss"]
        self.optimizer.zero_grad()
        loss_all.backward()
        self.optimizer.step()
        return fd, fu, dict(SSL_lambda=self.config.SSL_lambda, loss_all=loss_all)

    # src/agent.py:254-266
    def train_func_s1(self, data):
        self.net.train()
        fd, _ = self.forward(data, None)
        self.optimizer.zero_grad()
        fd["loss"].backward()
        self.optimizer.step()
        return fd

    # src/agent.py:269-274
    def val_func_s1(self, data):
        self.net.eval()
        with torch.no_grad():
            fd, _ = self.forward(data, None)
            return fd

    # src/agent.py:220-251
    def val_func(self, data, eval_ema=False):
        self.net.eval()
        self.ema_net.eval()
        with torch.no_grad():
            fd, _ = self.forward(data, None, eval_ema=eval_ema)
            entropy = OF.fisher_entropy(fd["pred"])
            mask = entropy < self.config.conf_thres
            ratio = (mask.sum() / len(mask)).item()
            err = None
            if ratio > 0:
                ge = data["euler_angles"][mask] if "euler_angles" in data else None
                err = OP.compute_err_deg_from_matrices(fd["pred_orth"][mask], data["rot_mat"][mask], ge)
            return fd, None, dict(mask_ratio=ratio, err_pseudo_gt=err)

    # src/agent.py:357-417
    def compute_dynamic_entropy_threshold(self, batches):
        self.ema_net.eval()
        ents = []
        with torch.no_grad():
            for b in batches:
                ents.append(OF.fisher_entropy(self.ema_net(b["img"])).numpy())
        thr = OP.entropy_threshold(np.concatenate(ents, 0), self.config.left_ratio)
        self.config.conf_thres = thr
        return thr


def synthetic_batches(B, muB, size=224, seed=0, full_range=False, rotate_aug=False, dtype=torch.float32):
    """The "224x224 synthetic heads" of SURVEY.md §8d: N(0,1) images, uniform Euler labels in the front range
    (or random SO(3) for full range), +-30 degree in-plane aug rotations rounded to 3 decimals."""
    g = torch.Generator().manual_seed(1000 + seed)
    data = {"img": torch.randn(B, 3, size, size, generator=g, dtype=dtype)}
    if full_range:
        q, r = torch.linalg.qr(torch.randn(B, 3, 3, generator=g, dtype=torch.float64))
        q = q * torch.sign(torch.diagonal(r, dim1=1, dim2=2))[:, None, :]
        q[:, :, 2] *= torch.det(q)[:, None]
        data["rot_mat"] = q.to(dtype).contiguous()
    else:
        ang = (torch.rand(B, 3, generator=g, dtype=torch.float64) * 2 - 1) * torch.tensor([60.0, 89.0, 60.0], dtype=torch.float64)
        R = np.stack([OP.get_6DRepNet_Rot(*(a * np.pi / 180).tolist()) for a in ang])
        data["rot_mat"] = torch.from_numpy(R).to(dtype)
        data["euler_angles"] = ang.to(dtype)
    if muB == 0:
        return data, None
    g2 = torch.Generator().manual_seed(2000 + seed)
    ulb = {"img": torch.randn(muB, 3, size, size, generator=g2, dtype=dtype),
           "img_strong": torch.randn(muB, 3, size, size, generator=torch.Generator().manual_seed(3000 + seed), dtype=dtype)}
    if rotate_aug:
        deg = torch.round((torch.rand(muB, generator=g2, dtype=torch.float64) * 60 - 30) * 1000) / 1000
    else:
        deg = torch.zeros(muB, dtype=torch.float64)
    rad = deg * np.pi / 180
    aug = torch.zeros(muB, 3, 3, dtype=torch.float64)
    aug[:, 0, 0], aug[:, 0, 1], aug[:, 1, 0], aug[:, 1, 1], aug[:, 2, 2] = torch.cos(rad), torch.sin(rad), -torch.sin(rad), torch.cos(rad), 1.0
    ulb["aug_rot_mat"] = aug.to(dtype)
    return data, ulb
