"""Drop-in for the reference's src/agent.py: `get_agent(config)` -> SSLAgent with the same public surface
(net, ema_net, optimizer, clock, logs_writer, train_func, train_func_s1, val_func, val_func_s1,
compute_dynamic_entropy_threshold, update_ema_variables, save_ckpt, load_ckpt).

What differs underneath (behaviour on the same inputs is the reference's):
  * every head call is one HIP kernel on the training stream (the reference round-trips to the CPU 8-10x per step);
  * the pseudo-label filter is a device-side mask folded into the CE kernel (sum mask*CE / muB == mean(CE[mask]) *
    mask_ratio, src/agent.py:163-168) — no boolean compaction, no `if mask_ratio > 0` host sync;
  * CutOut+CutMix+NCHW->NHWC is one pass; rotation-consistency adjust is one kernel;
  * Adam / EMA / zero_grad are one kernel each over flat buffers; with world_size > 1 the flat gradient is
    all-reduced over RCCL (one process per GPU);
  * the per-step os.listdir + debug JPG dump of src/agent.py:128-136 is dropped (filesystem work in the hot loop).
"""
import os
from datetime import datetime

import numpy as np
import torch

from .. import kernels as K
from . import distributed as D
from . import utils
from .. import layers as L
from ..layers import batch_segments
from .augments import strong_view_nhwc
from .fisher.fisher_utils import batch_torch_A_to_R, fisher_CE, fisher_entropy
from .fisher.fisher_utils import vmf_loss as fisher_NLL
from .networks import get_network
from .optim import FlatBuffer, FusedAdam
from .utils import compute_euler_angles_from_rotation_matrices  # noqa: F401  (re-exported like the reference)


def get_agent(config):
    return SSLAgent(config)


class _Mean(torch.autograd.Function):
    """mean over a 1-D device tensor with the division folded into the reduction kernel"""

    @staticmethod
    def forward(ctx, x, denom):
        ctx.n, ctx.denom = x.numel(), denom
        return K.weighted_sum(x.contiguous(), None, 1.0 / denom)

    @staticmethod
    def backward(ctx, g):
        return K.broadcast_scale(g.contiguous(), 1.0 / ctx.denom, ctx.n), None


class _Axpby(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, beta):
        ctx.beta = beta
        return K.axpby(a, 1.0, b, beta)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        return g, K.axpby(g, ctx.beta), None


def _mode(net, training):
    """net.train() / net.eval() without re-walking ~600 submodules every step when the mode is already set"""
    if net.training != training:
        net.train(training)


def _dev(t, device):
    if t.device == device:
        return t
    return t.to(device, non_blocking=True)


class SSLAgent:
    def __init__(self, config):
        self.config = config
        self.clock = utils.TrainClock()
        # time the GEMM tile menu on each new conv shape during the first steps and pin the winners (kernels._autotune)
        K.AUTOTUNE = bool(getattr(config, "autotune", True)) and os.environ.get("SUHPE_AUTOTUNE", "1") != "0"
        self.merge_student_passes = bool(getattr(config, "merge_student_passes", True)) and os.environ.get("SUHPE_MERGE", "1") != "0"
        # the teacher's forward pass does not depend on the student's: it is enqueued on a second stream so that the two
        # passes fill each other's launch gaps and partial waves
        self.overlap_teacher = bool(getattr(config, "overlap_teacher", True)) and os.environ.get("SUHPE_OVERLAP", "1") != "0"
        self._teacher_stream = None
        self.net = get_network(config)
        self.ema_net = get_network(config)
        # optional SyncBN (global-batch BatchNorm statistics across ranks): exact global-batch semantics at the price of two small
        # all-reduces per BatchNorm layer and pass; it uses the reference's two student passes
        L.SYNC_BN = (bool(getattr(config, "sync_bn", False)) or os.environ.get("SUHPE_SYNC_BN") == "1") and D.world()[1] > 1
        if L.SYNC_BN:
            self.merge_student_passes = False
            self.overlap_teacher = False              # the per-layer collectives of both networks stay in one stream's program order
        for param in self.ema_net.parameters():      # reference :34-35: the teacher is never trained by gradients
            param.detach_()
            param.requires_grad_(False)
        self.device = next(self.net.parameters()).device
        self.optimizer = FusedAdam(self.net.parameters(), config.lr)
        self.optimizer.flatten()
        self._ema_flat = FlatBuffer(list(self.ema_net.parameters()), with_grad=False)
        self.rank, self.world_size = D.world()
        # BatchNorm running statistics of each network as one flat buffer (averaged across ranks once per step)
        bufs = [b for net in (self.net, self.ema_net) for n, b in net.named_buffers() if b.dtype == torch.float32]
        self._bn_flat_all = FlatBuffer(bufs, with_grad=False) if bufs else None
        if self.world_size > 1:                       # identical replicas to start from (rank 0's init)
            for fb in self.optimizer.flat_buffers:
                D.broadcast_(fb.data, 0)
            D.broadcast_(self._ema_flat.data, 0)
            if self._bn_flat_all is not None:
                D.broadcast_(self._bn_flat_all.data, 0)
        self._grad_seed = D.grad_seed(self.device)
        # The host enqueues a step in a third of the time the GPU needs for it and nothing in the step synchronises, so it would run
        # many steps ahead; every tensor that crosses to the side stream (record_stream) stays unavailable to the caching allocator
        # until the GPU has passed it, so the RESERVED pool grew to ~5 steps' worth of activations (74 GiB for a 14 GiB step).  The
        # host now waits for the end of step i - depth before it enqueues step i: the launch queue stays 1-2 steps deep (the GPU never
        # idles) and the pool holds what `depth` steps need.
        self._flight_depth = max(1, int(getattr(config, "steps_in_flight", os.environ.get("SUHPE_STEPS_IN_FLIGHT", "2"))))
        self._flight_events = []
        # gradient exchange in reverse-layer buckets, overlapped with the backward (one stream for RCCL beside the compute
        # streams); SUHPE_GRAD_BUCKETS=1 (or config.grad_buckets) = one collective after the backward
        self._buckets = None
        nb = int(getattr(config, "grad_buckets", os.environ.get("SUHPE_GRAD_BUCKETS", "4")))
        if self.world_size > 1 and nb > 1 and len(self.optimizer.flat_buffers) == 1:
            fb = self.optimizer.flat_buffers[0]
            names = [n for n, _ in self.net.named_parameters()]
            bounds = D.bucket_boundaries(names, fb.offsets, fb.numel, nb)
            if len(bounds) > 2:
                self._buckets = D.BucketedAllReduce(fb.grad, bounds, ready=lambda k: K.flush_deferred_now(),
                                                    streams=lambda: K.backward_streams(self.device))
                self._grad_base = fb.grad.data_ptr()
        self._tiles_agreed = set()
        self._thres_host = None
        self._thres_dev = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._finite_flag = None

        if self.rank == 0:
            nowstr = datetime.now().strftime("%Y%m%d-%H%M%S")
            os.makedirs(self.config.log_dir, exist_ok=True)
            self.logs_writer = open(os.path.join(self.config.log_dir, "txt_" + nowstr + ".log"), "w")
        else:
            self.logs_writer = open(os.devnull, "w")
        self.grids = None
        if getattr(config, "distribution", "matrixFisher") == "RotationLaplace":     # reference :43-58
            from .laplace.rotation_laplace import default_grids
            self.grids = default_grids(self.device)

    # ------------------------------------------------------------------------------------------------ helpers
    def _conf_thres(self):
        """Device copy of config.conf_thres, refreshed only when the host value changes (it is mutated at runtime by
        compute_dynamic_entropy_threshold, reference :416)."""
        t = float(self.config.conf_thres)
        if t != self._thres_host:
            self._thres_host = t
            self._thres_dev = torch.tensor([t], dtype=torch.float32).to(self.device)
        return self._thres_dev

    def _masked_mean(self, x, mask, count):
        """sum(mask*x)/count as a 1-element device tensor (nan when nothing is selected, like mean of an empty tensor)"""
        return K.masked_mean(x, mask)

    def _teacher_forward(self, img):
        """teacher(weak view) on the side stream; the caller joins (wait_stream) before it reads the result"""
        if self._teacher_stream is None:
            self._teacher_stream = torch.cuda.Stream(device=self.device)
        side = self._teacher_stream
        side.wait_stream(torch.cuda.current_stream())       # EMA update and the input batch are ordered before the pass
        with torch.cuda.stream(side), torch.no_grad():
            img.record_stream(side)
            return self.ema_net(img)

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, data, ulb_data, eval_ema=False):
        cfg, dev = self.config, self.device
        ### 1. supervised loss (reference :62-96)
        img = _dev(data.get("img"), dev)
        gt = _dev(data.get("rot_mat"), dev).reshape(-1, 9).contiguous()
        gt_euler = _dev(data.get("euler_angles"), dev).contiguous() if "euler_angles" in data else None
        net = self.ema_net if eval_ema else self.net
        # The two student passes of a step (labeled batch, strong views) share the weights: run them as one batch whose
        # BatchNorm statistics are kept per segment (layers.batch_segments) -- same arithmetic per sample, larger GEMMs.
        merged = (self.merge_student_passes and ulb_data is not None and not eval_ema and net.training and torch.is_grad_enabled())
        self._merged_step = merged
        pred_strong = pred_weak = None
        if ulb_data is not None and self.overlap_teacher:
            pred_weak = self._teacher_forward(_dev(ulb_data.get("img"), dev))
        if merged:
            strong = _dev(ulb_data.get("img_strong"), dev)
            n_l, n_u = img.shape[0], strong.shape[0]
            both = torch.empty((n_l + n_u, img.shape[2], img.shape[3], img.shape[1]), dtype=torch.float32, device=dev)
            K.nchw_to_nhwc(img.contiguous(), out=both[:n_l])
            strong_view_nhwc(strong, cfg.cutout_aug, cfg.cutmix_aug, 3, normal=True, out=both[n_l:])
            with batch_segments(n_l):
                both = net(both, nhwc=True)
            fisher_out, pred_strong = both[:img.shape[0]], both[img.shape[0]:]
        else:
            fisher_out = net(img)
        if cfg.distribution == "RotationLaplace":
            from .laplace.rotation_laplace import NLL_loss as laplace_NLL
            losses, pred_orth = laplace_NLL("RLaplace", fisher_out, gt, self.grids)
        else:
            losses, pred_orth = fisher_NLL(fisher_out, gt, overreg=1.025)
        loss = _Mean.apply(losses, losses.numel()).squeeze(0)
        err_deg = self.compute_err_deg_from_matrices(pred_orth, gt, gt_euler=gt_euler)
        fisher_dict = dict(loss=loss, pred=fisher_out, pred_orth=pred_orth, err_deg=err_deg)
        if ulb_data is None:
            return fisher_dict, None

        ### 2. unsupervised loss (reference :98-192)
        ulb_img_weak = _dev(ulb_data.get("img"), dev)
        ulb_gt = _dev(ulb_data.get("rot_mat"), dev).reshape(-1, 9).contiguous() if "rot_mat" in ulb_data else None
        if pred_weak is None:
            with torch.no_grad():
                pred_weak = self.ema_net(ulb_img_weak)
        else:                                               # issued on the teacher stream before the student pass: join
            main = torch.cuda.current_stream()
            main.wait_stream(self._teacher_stream)
            pred_weak.record_stream(main)
        mu_b = pred_weak.shape[0]
        if cfg.rotate_aug:
            aug = _dev(ulb_data.get("aug_rot_mat"), dev).reshape(-1, 9).contiguous()
            if cfg.train_labeled == "DAD3DHeads":
                pred_weak_adjusted = K.rot_adjust(pred_weak, aug, 0)
            elif cfg.train_labeled == "300WLP":
                pred_weak_adjusted = K.rot_adjust(pred_weak, aug, 1)
            else:
                raise ValueError(f"rotate_aug: unknown train_labeled '{cfg.train_labeled}'")
        else:
            pred_weak_adjusted = pred_weak
        # CutOut / CutMix on the strong view fused with the layout change (reference :125-126)
        if pred_strong is None:
            strong_nhwc = strong_view_nhwc(_dev(ulb_data.get("img_strong"), dev), cfg.cutout_aug, cfg.cutmix_aug, 3, normal=True)
            pred_strong = self.net(strong_nhwc, nhwc=True)
        entropy = fisher_entropy(pred_weak)
        mask, count = K.entropy_mask(entropy, self._conf_thres())
        mask_ratio = K.axpby(count, 1.0 / mu_b).squeeze(0)
        if cfg.type_unsuper == "ce":
            unsuper_losses = fisher_CE(pred_weak_adjusted, pred_strong, mask)
            unsuper_loss = _Mean.apply(unsuper_losses, mu_b).squeeze(0)       # == mean(CE[mask]) * mask_ratio
        else:  # 'nll' against the projected pseudo labels (reference :157)
            pseudo = batch_torch_A_to_R(pred_weak_adjusted).reshape(-1, 9)
            if cfg.distribution == "RotationLaplace":
                from .laplace.rotation_laplace import NLL_loss as laplace_NLL
                nll, _ = laplace_NLL("RLaplace", pred_strong, pseudo, self.grids)
            else:
                nll, _ = fisher_NLL(pred_strong, pseudo, overreg=1.025)
            unsuper_loss = _MaskedSum.apply(nll, mask, 1.0 / mu_b).squeeze(0)
        self._finite_flag = unsuper_loss.detach()

        # diagnostics (reference :171-181), masked means stay on the device
        pseudo_all = batch_torch_A_to_R(pred_weak_adjusted)
        if ulb_gt is not None:
            err_all = self.compute_err_deg_from_matrices(pseudo_all, ulb_gt)
            err_weakAll_gt = err_all
            err_weakPseudo_gt = self._masked_mean(err_all, mask, count)
        else:
            err_weakAll_gt = torch.zeros(1, device=dev)
            err_weakPseudo_gt = torch.zeros(1, device=dev)
        err_ss = self.compute_err_deg_from_matrices(batch_torch_A_to_R(pred_strong), pseudo_all)
        err_strongSuper_pseudo = self._masked_mean(err_ss, mask, count)

        fisher_dict_unsuper = dict(unsuper_loss=unsuper_loss, entropy=entropy, mask_ratio=mask_ratio,
                                   err_weakAll_gt=err_weakAll_gt, err_weakPseudo_gt=err_weakPseudo_gt,
                                   err_strongSuper_pseudo=err_strongSuper_pseudo)
        return fisher_dict, fisher_dict_unsuper

    # ------------------------------------------------------------------------------------------------ steps
    def _agree_on_tiles(self, kind, dry):
        """Multi-rank tile selection without losing replica determinism.  With more than one rank the lazy autotuner measures
        nothing (kernels._autotune), so a batch size absent from the shipped tile database would run on the heuristic tiles (never Winograd, >= 10 %
        slower).  The FIRST training step of each kind (`train_func`, `train_func_s1`) is a point every rank reaches together -- the step ends in the
        gradient all-reduce -- so there rank 0 alone runs one dry forward + backward of that step on its own batch with the autotuner measuring
        (kernels.SOLO_TUNE), state saved and restored around it (BatchNorm buffers, mask-generator position, host RNGs; the gradient buffer is zeroed
        by the step that follows; no optimiser step, no collective inside), and broadcasts its {shape: tile name} table over the host channel; every
        rank adopts it, so every replica pins the same kernels for the same shapes.  Validation / threshold-scan shapes stay on the database or the
        heuristic (the ranks do not reach them together)."""
        if self.world_size <= 1 or not K.AUTOTUNE or kind in self._tiles_agreed:
            return
        self._tiles_agreed.add(kind)
        import random
        import numpy as np
        if self.rank == 0:
            bufs = [b for net in (self.net, self.ema_net) for b in net.buffers()]
            saved = [b.clone() for b in bufs]
            rng = (torch.get_rng_state(), torch.cuda.get_rng_state(self.device), np.random.get_state(), random.getstate(), dict(K._host_offsets))
            K.SOLO_TUNE = True
            try:
                dry()
                torch.cuda.synchronize(self.device)
            finally:
                K.SOLO_TUNE = False
                L.set_grad_watch(None)
            with torch.no_grad():
                for b, v in zip(bufs, saved):
                    b.copy_(v)
            torch.set_rng_state(rng[0]); torch.cuda.set_rng_state(rng[1], self.device); np.random.set_state(rng[2]); random.setstate(rng[3])
            K._host_offsets.update(rng[4])
            for hook in K._reseed_hooks:                   # masks drawn ahead during the dry pass belong to positions that will be drawn again
                hook()
            K.flush_deferred_now()
        K.share_tiles_from_rank0()

    def _watch_grads(self, single_backward):
        """arm the bucketed exchange for the backward that follows (only when ONE network backward produces the whole gradient:
        the merged student pass or a supervised step; with two student passes the buckets are complete only after both)"""
        if self._buckets is None:
            return
        self._buckets.begin()
        if single_backward:
            base = self._grad_base
            L.set_grad_watch(lambda g: self._buckets.touched((g.data_ptr() - base) >> 2))
        else:
            self._buckets.active = False

    def _exchange(self):
        """The exchange step of data-parallel training: SUM all-reduce of the flat (pre-scaled) gradient -- in buckets already
        under way since the backward left them, see _watch_grads -- and AVG of the BatchNorm running statistics of student and
        teacher (ONE collective: both networks' buffers live in one flat buffer)."""
        if self.world_size > 1:
            if self._buckets is not None:
                L.set_grad_watch(None)
                self._buckets.finish()
            else:
                for fb in self.optimizer.flat_buffers:
                    D.allreduce_sum_(fb.grad)
            if self._bn_flat_all is not None:
                D.allreduce_avg_(self._bn_flat_all.data)

    def _throttle(self):
        """keep at most `steps_in_flight` training steps enqueued ahead of the GPU (bounds the caching allocator's pool)"""
        if len(self._flight_events) >= self._flight_depth:
            self._flight_events.pop(0).synchronize()

    def _step_done(self):
        ev = torch.cuda.Event()
        ev.record()
        self._flight_events.append(ev)

    def train_func(self, data, ulb_data):
        """one SSL step (reference :194-218)"""
        self._throttle()
        _mode(self.net, True)
        _mode(self.ema_net, True)
        if self.world_size > 1 and "ssl" not in self._tiles_agreed:
            def dry():
                fd, fu = self.forward(data, ulb_data)
                loss = _Axpby.apply(fd["loss"].reshape(1), fu["unsuper_loss"].reshape(1), float(self.config.SSL_lambda)).squeeze(0)
                with K.filter_cache(self):
                    loss.backward(self._grad_seed)
            self._agree_on_tiles("ssl", dry)
        stage2_iter = self.clock.iteration - self.config.stage1_iteration
        self.update_ema_variables(self.config.is_ema, self.config.ema_decay, stage2_iter)
        self.optimizer.zero_grad()
        fisher_dict, fisher_dict_unsuper = self.forward(data, ulb_data)
        SSL_lambda = self.config.SSL_lambda
        loss_all = _Axpby.apply(fisher_dict["loss"].reshape(1), fisher_dict_unsuper["unsuper_loss"].reshape(1),
                                float(SSL_lambda)).squeeze(0)
        self._watch_grads(self._merged_step)
        with K.filter_cache(self):                  # transposed filters: one batched launch per step (and shared by both passes)
            loss_all.backward(self._grad_seed)
        self._exchange()
        self.optimizer.step()
        self._step_done()
        return fisher_dict, fisher_dict_unsuper, dict(SSL_lambda=SSL_lambda, loss_all=loss_all.detach())

    def train_func_s1(self, data):
        """supervised step (reference :254-266); the EMA teacher is not touched in stage 1"""
        self._throttle()
        _mode(self.net, True)
        if self.world_size > 1 and "s1" not in self._tiles_agreed:
            self._agree_on_tiles("s1", lambda: self.forward(data, None)[0]["loss"].backward(self._grad_seed))
        self.optimizer.zero_grad()
        fisher_dict, _ = self.forward(data, None)
        self._watch_grads(True)
        fisher_dict["loss"].backward(self._grad_seed)
        self._exchange()
        self.optimizer.step()
        self._step_done()
        return fisher_dict

    def val_func(self, data, eval_ema=False):
        """one validation step (reference :220-251); host syncs are fine here (the reference calls .item() too)"""
        _mode(self.net, False)
        _mode(self.ema_net, False)
        with torch.no_grad():
            fisher_dict, fisher_dict_unsuper = self.forward(data, None, eval_ema=eval_ema)
            entropy = fisher_entropy(fisher_dict["pred"])
            fisher_mask = entropy < self.config.conf_thres
            fisher_mask_ratio = (fisher_mask.sum() / len(fisher_mask)).item()
            if fisher_mask_ratio > 0:
                gt_euler = _dev(data.get("euler_angles"), self.device)[fisher_mask] if "euler_angles" in data else None
                fisher_err_pseudo_gt = self.compute_err_deg_from_matrices(
                    fisher_dict["pred_orth"][fisher_mask], _dev(data.get("rot_mat"), self.device)[fisher_mask], gt_euler=gt_euler)
            else:
                fisher_err_pseudo_gt = None
            return fisher_dict, fisher_dict_unsuper, dict(mask_ratio=fisher_mask_ratio, err_pseudo_gt=fisher_err_pseudo_gt)

    def val_func_s1(self, data):
        _mode(self.net, False)
        with torch.no_grad():
            fisher_dict, _ = self.forward(data, None)
            return fisher_dict

    def update_ema_variables(self, is_ema, alpha, global_step):
        """reference :277-299"""
        alpha = min(1 - 1 / (global_step + 1), alpha) if is_ema else 0
        student = self.optimizer.flat_buffers[0]
        K.ema_update(self._ema_flat.data, student.data, alpha)
        if getattr(self.config, "eman", False):      # EMAN: buffers too, num_batches_tracked copied
            for (k, v_main), (_, v_ema) in zip(self.net.named_buffers(), self.ema_net.named_buffers()):
                if "num_batches_tracked" in k:
                    v_ema.copy_(v_main)
                else:
                    K.ema_update(v_ema, v_main, alpha)

    def check_finite(self):
        """Lazy replacement for the reference's per-step `assert not isnan/isinf` on the CE (fisher_utils.py:98):
        call at log time; one device->host read."""
        if self._finite_flag is not None:
            assert bool(torch.isfinite(self._finite_flag).all()), "unsupervised loss became nan/inf"

    # ------------------------------------------------------------------------------------------------ checkpoints
    def save_ckpt(self, name=None):
        """reference :302-331 — same file name and dict keys; tensors are copied to the host without moving the
        live networks off the device (that would detach the flat-buffer views)."""
        if self.rank != 0:
            return
        if name is None:
            save_path = os.path.join(self.config.model_dir, "ckpt_iteration{}.pth".format(self.clock.iteration))
            print("[{}/{}] Saving checkpoint iteration {}...".format(self.config.exp_name, self.config.date, self.clock.iteration))
        else:
            save_path = os.path.join(self.config.model_dir, "{}.pth".format(name))
            print("[{}/{}] Saving checkpoint {}...".format(self.config.exp_name, self.config.date, name))

        def host(sd):
            return {k: v.detach().cpu().contiguous() for k, v in sd.items()}

        # Optimizer.state_dict() hands out the LIVE per-parameter dicts (sd["state"][i] is optimizer.state[p]): build a fresh
        # dict of host copies and leave the live one -- views of the flat Adam buffers, shared step scalar -- untouched
        sd = self.optimizer.state_dict()
        opt = {"state": {i: {k: (v.detach().cpu().contiguous().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                         for i, st in sd["state"].items()},
               "param_groups": [dict(g) for g in sd["param_groups"]]}
        torch.save({"clock": self.clock.make_checkpoint(), "model_state_dict": host(self.net.state_dict()),
                    "model_state_dict_ema": host(self.ema_net.state_dict()), "optimizer_state_dict": opt}, save_path)

    def load_ckpt(self, name=None):
        """reference :333-354"""
        load_path = name if os.path.isabs(name) else os.path.join(self.config.model_dir, "{}.pth".format(name))
        if not os.path.exists(load_path):
            raise ValueError("Checkpoint {} not exists.".format(load_path))
        checkpoint = torch.load(load_path, map_location=torch.device("cpu"))
        print("Loading checkpoint from {} ...".format(load_path))
        self.net.load_state_dict(checkpoint["model_state_dict"])
        if "model_state_dict_ema" in checkpoint.keys():
            self.ema_net.load_state_dict(checkpoint["model_state_dict_ema"])
        self.optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
        self.clock.restore_checkpoint(checkpoint["clock"])

    # ------------------------------------------------------------------------------------------------ threshold
    # the layer whose output the reference dumps for t-SNE (src/agent.py:376-379: the second Linear of the regression head);
    # EfficientNetV2 / TinyViT have no entry there (the reference crashes on them): their head's layer 4 is used the same way
    _FEAT_LAYER = {"mobilenet": "classifier.4", "resnet18": "fc.4", "resnet50": "fc.4", "repvgg": "linear_reg.4",
                   "effinetv2": "classifier.4", "tinyvit": "head.4"}

    def compute_dynamic_entropy_threshold(self, ulb_train_bar):
        """reference :357-417: teacher (eval mode) entropies over the unlabeled set, sorted, left_ratio quantile.
        With world_size > 1 each rank scans its own loader shard and the entropy vectors are all-gathered.
        config.save_feat (:369-414): the head's 64/128-d feature of every unlabeled sample is written with its entropy to
        <log_dir>/ulb_feats_iter<iteration>_thre<threshold>.json as {str(idx): [feature list, entropy]}."""
        _mode(self.ema_net, False)
        chunks, feats, idxs = [], [], []
        hooked = None
        if getattr(self.config, "save_feat", False):
            name = self._FEAT_LAYER.get(self.config.network)
            mods = dict(self.ema_net.named_modules())
            if name not in mods:
                raise NotImplementedError(f"--save_feat: no feature layer known for network '{self.config.network}'")
            hooked, fmap = mods[name], {}
            orig_fwd = hooked.fwd

            def fwd_and_keep(x, tape):
                y = orig_fwd(x, tape)
                fmap["output"] = y
                return y
            hooked.fwd = fwd_and_keep
        try:
            with torch.no_grad():
                seen = 0
                for ulb_data in ulb_train_bar:
                    pred_weak = self.ema_net(_dev(ulb_data.get("img"), self.device))
                    chunks.append(fisher_entropy(pred_weak))
                    if hooked is not None:
                        feats.append(fmap["output"].reshape(pred_weak.shape[0], -1))
                        idx = ulb_data.get("idx")
                        n = pred_weak.shape[0]
                        idxs += [str(i) for i in (idx.tolist() if torch.is_tensor(idx) else idx)] if idx is not None else \
                            [f"r{self.rank}_{seen + i}" for i in range(n)]
                        seen += n
        finally:
            if hooked is not None:
                del hooked.fwd                      # drop the instance attribute: the class's fwd is back
        ent_local = torch.cat(chunks) if chunks else torch.zeros(0, device=self.device)
        ent = D.gather_varlen(ent_local)
        entropy_all = ent.cpu().numpy()
        entropy_all.sort()
        index = int(len(entropy_all) * self.config.left_ratio)
        entropy_thre = entropy_all[index]
        if self.rank == 0:
            print("The best dynamic entropy threshold is:", entropy_thre)
        if hooked is not None:
            import json
            f = torch.cat(feats).cpu().numpy().tolist() if feats else []
            e = ent_local.cpu().numpy().tolist()
            table = {k: [fv, ev] for k, fv, ev in zip(idxs, f, e)}
            if self.world_size > 1:
                import torch.distributed as dist
                parts = [None] * self.world_size
                dist.all_gather_object(parts, table)
                table = {k: v for part in parts for k, v in part.items()}
            if self.rank == 0:
                path = os.path.join(self.config.log_dir, "ulb_feats_iter%d_thre%s.json" % (self.clock.iteration, str(entropy_thre)))
                with open(path, "w") as fh:
                    json.dump(table, fh)
        self.config.conf_thres = entropy_thre
        return entropy_thre

    # ------------------------------------------------------------------------------------------------ metrics
    @staticmethod
    def compute_err_deg_from_matrices(pred, gt, gt_euler=None):
        """reference :447-455"""
        pred = pred.reshape(-1, 9)
        gt = gt.reshape(-1, 9)
        return K.pose_err_deg(pred if pred.is_contiguous() else pred.contiguous(),
                              gt if gt.is_contiguous() else gt.contiguous(), gt_euler)


class _MaskedSum(torch.autograd.Function):
    """scale * sum(mask * x)"""

    @staticmethod
    def forward(ctx, x, mask, scale):
        ctx.save_for_backward(mask)
        ctx.scale = scale
        return K.weighted_sum(x.contiguous(), mask, scale)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        gb = K.broadcast_scale(g.contiguous(), ctx.scale, mask.numel())
        return K.mul(gb, mask), None, None
