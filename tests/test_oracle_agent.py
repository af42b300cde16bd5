"""Pins oracle/agent.py (the CPU restatement of SSLAgent.forward / train_func / EMA) against three steps of the
REFERENCE's own SSLAgent with RepVGG-B1g2 (tests/golden/agent_step.npz from tools/make_golden_agent.py)."""
import torch

import agent_fixture as AF
from oracle import networks as ON
from oracle.agent import OracleAgent
from parity_utils import seeded_state_dict


def test_oracle_agent_reproduces_reference_steps():
    def factory():
        m = ON.RepVggNet()
        m.load_state_dict(seeded_state_dict(m.state_dict(), 17))
        for mod in m.modules():
            if isinstance(mod, ON.Dropout):
                mod.p = 0.0
        return m

    cfg = AF.config()
    agent = OracleAgent(cfg, factory)

    def set_it(step):
        agent.iteration = step

    AF.run_and_compare(agent, cfg, set_it, lambda d: {k: v.clone() for k, v in d.items()}, value_tol=2e-5, grad_tol=2e-4)
