"""GPU: SSLAgent with RepVGG-B1g2 (C3's backbone) against the REFERENCE's own SSLAgent outputs for the same step
(tests/golden/agent_step.npz), then three steps against the CPU oracle with the state re-synchronised each step
(DAD3DHeads rotation convention, geodesic metric, CutOut+CutMix, strict-subset pseudo-label masks, EMA alpha 0, 1/2, 2/3).

Gradient yardstick: 28 train-mode BatchNorm layers over 4-12 images of 64x64 amplify fp32 rounding to the percent level
in some filters (the reference's own fp32 gradients deviate that much from the exact ones).  So gradients are compared
with the float64 oracle as the truth: the HIP path must be as close to the exact gradient as the fp32 reference/oracle is
(x4, floor 3e-3 of the tensor's max)."""
import numpy as np
import pytest
import torch

import agent_fixture as AF
from parity_utils import rel_err, seeded_state_dict

pytestmark = pytest.mark.gpu


def _hip_agent(cfg):
    from semiuhpe_amd import layers as L
    from semiuhpe_amd.src.agent import SSLAgent
    agent = SSLAgent(cfg)
    sd = seeded_state_dict(agent.net.state_dict(), 17)
    agent.net.load_state_dict(sd)
    agent.ema_net.load_state_dict(sd)
    for net in (agent.net, agent.ema_net):
        for m in net.modules():
            if isinstance(m, L.Dropout):
                m.p = 0.0
    return agent


def _oracle_agent(cfg, dtype=torch.float32):
    from oracle import networks as ON
    from oracle.agent import OracleAgent

    def factory():
        m = ON.RepVggNet()
        m.load_state_dict(seeded_state_dict(m.state_dict(), 17))
        for mod in m.modules():
            if isinstance(mod, ON.Dropout):
                mod.p = 0.0
        return m.to(dtype)

    return OracleAgent(cfg, factory)


def _yardstick(got, ref32, truth64, floor=3e-3, mult=4.0, gscale=0.0):
    """gscale: max |gradient| over the whole network; tensors whose exact gradient is (analytically) zero — a bias that
    feeds a train-mode BatchNorm — are measured against 1e-4 of it instead of their own rounding noise."""
    t = truth64.detach().double().reshape(-1)
    scale = max(1e-12, float(t.abs().max()), 1e-4 * gscale)
    noise = float((ref32.detach().double().reshape(-1) - t).abs().max()) / scale
    err = float((got.detach().cpu().double().reshape(-1) - t).abs().max()) / scale
    return err <= max(floor, mult * noise), err, noise


def test_first_step_matches_reference_agent():
    """values against the reference's fixture; gradient samples against the fp64 truth with the fixture as yardstick"""
    from oracle.agent import synthetic_batches
    cfg = AF.config()
    agent = _hip_agent(cfg)
    AF.run_and_compare(agent, cfg, lambda s: setattr(agent.clock, "iteration", s), lambda d: {k: v.cuda() for k, v in d.items()},
                       value_tol=3e-4, grad_tol=1e9, param_tol=2e-5, steps=1, orth_tol=3e-4)
    cfg64 = AF.config()
    truth = _oracle_agent(cfg64, torch.float64)
    data, ulb = synthetic_batches(4, 12, size=64, seed=0, rotate_aug=True, dtype=torch.float64)
    ulb["rot_mat"] = synthetic_batches(12, 0, size=8, seed=70, dtype=torch.float64)[0]["rot_mat"]
    truth.iteration = 0
    torch.manual_seed(100)
    truth.train_func(data, ulb)
    tp, hp = dict(truth.net.named_parameters()), dict(agent.net.named_parameters())
    for k in AF.KEYS:
        ok, err, noise = _yardstick(torch.from_numpy(AF.sample(hp[k].grad)), torch.from_numpy(AF.G[f"s0/grad/{k}"].astype(np.float64)),
                                    torch.from_numpy(AF.sample(tp[k].grad)))
        assert ok, (k, err, noise)


def test_three_steps_match_oracle_with_resync():
    from oracle.agent import synthetic_batches
    cfg_o, cfg_t, cfg_h = (AF.config(train_labeled="DAD3DHeads") for _ in range(3))
    oracle, truth, agent = _oracle_agent(cfg_o), _oracle_agent(cfg_t, torch.float64), _hip_agent(cfg_h)
    for step in range(3):
        data, ulb = synthetic_batches(4, 12, size=64, seed=step, full_range=True, rotate_aug=True)     # no euler labels
        d64, u64 = synthetic_batches(4, 12, size=64, seed=step, full_range=True, rotate_aug=True, dtype=torch.float64)
        cfg_o.conf_thres = cfg_t.conf_thres = cfg_h.conf_thres = float(AF.G[f"s{step}/conf_thres"])
        oracle.iteration = truth.iteration = agent.clock.iteration = step
        torch.manual_seed(100 + step)
        fd_o, fu_o, out_o = oracle.train_func({k: v.clone() for k, v in data.items()}, {k: v.clone() for k, v in ulb.items()})
        g_ref = {n: p.grad.clone() for n, p in oracle.net.named_parameters()}
        torch.manual_seed(100 + step)
        fd_t, _, _ = truth.train_func(d64, u64)
        g_true = {n: p.grad.clone() for n, p in truth.net.named_parameters()}
        torch.manual_seed(100 + step)
        fd, fu, out = agent.train_func({k: v.cuda() for k, v in data.items()}, {k: v.cuda() for k, v in ulb.items()})
        assert rel_err(fd["loss"], fd_o["loss"]) < 3e-4 and rel_err(fd["err_deg"], fd_o["err_deg"]) < 2e-3      # geodesic metric
        # train-mode BN on 4 images amplifies fp32 rounding: the 1e-4 bound on rotation matrices is asserted in eval mode
        # (test_gpu_backbones.py); here the HIP path must sit as close to the float64 result as the fp32 oracle does (x4)
        orth_noise = float((fd_o["pred_orth"].double() - fd_t["pred_orth"]).abs().max())
        orth_err = float((fd["pred_orth"].cpu().double() - fd_t["pred_orth"]).abs().max())
        assert orth_err <= max(1e-4, 4.0 * orth_noise), (step, orth_err, orth_noise)
        assert float((fd["pred_orth"].cpu() - fd_o["pred_orth"]).abs().max()) < 1e-3
        assert rel_err(fu["entropy"], fu_o["entropy"]) < 3e-4
        assert abs(float(fu["mask_ratio"]) - float(fu_o["mask_ratio"])) < 1e-6 and 0 < float(fu["mask_ratio"]) <= 1
        assert rel_err(fu["unsuper_loss"], fu_o["unsuper_loss"]) < 2e-3
        bad = {}
        gscale = max(float(g.abs().max()) for g in g_true.values())
        for n, p in agent.net.named_parameters():
            ok, err, noise = _yardstick(p.grad, g_ref[n], g_true[n], gscale=gscale)
            if not ok:
                bad[n] = (err, noise)
        assert not bad, (step, list(bad.items())[:6])
        for (n, pe), (_, po) in zip(agent.ema_net.named_parameters(), oracle.ema_net.named_parameters()):
            assert rel_err(pe, po) < 1e-6, n
        # re-synchronise all three replicas on the fp32 oracle's post-step state
        sd, sde = oracle.net.state_dict(), oracle.ema_net.state_dict()
        agent.net.load_state_dict(sd); agent.ema_net.load_state_dict(sde)
        agent.optimizer.load_state_dict(oracle.optimizer.state_dict())
        truth.net.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in sd.items()})
        truth.ema_net.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in sde.items()})
        st = oracle.optimizer.state_dict()
        st = {"param_groups": st["param_groups"],
              "state": {i: {kk: (vv.double() if kk != "step" else vv.clone()) for kk, vv in s_.items()} for i, s_ in st["state"].items()}}
        truth.optimizer.load_state_dict(st)
