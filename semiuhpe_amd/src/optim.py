"""Flat parameter / gradient / optimiser-state buffers and the fused Adam step.

The reference builds `optim.Adam(net.parameters(), lr)` (src/agent.py:37) — ~450 small foreach kernels per step — and
updates the EMA teacher with a mul_/add_ pair per parameter tensor (src/agent.py:298-299).  Here every parameter of
a network is re-pointed to a view of ONE contiguous HBM buffer (same for gradients and Adam moments), so that
  * zero_grad is one fill, Adam is one kernel, the EMA update is one kernel,
  * data-parallel training all-reduces one flat buffer over RCCL (a few large collectives, sized for xGMI),
  * layer backward kernels accumulate straight into their slice of the flat gradient (semiuhpe_amd/layers.grad_of).
`FusedAdam` keeps torch.optim.Adam's state_dict layout (state[i] = {step, exp_avg, exp_avg_sq}) so checkpoints are
interchangeable with the reference's (`optimizer_state_dict`, src/agent.py:327,353).
"""
import torch

from .. import kernels as K

_ALIGN = 4   # floats (16 bytes): every slice starts on a float4 boundary


class FlatBuffer:
    """Flattens the parameters of `params` (list) into one fp32 buffer and re-points each parameter to its slice."""

    def __init__(self, params, with_grad=True):
        params = [p for p in params]
        assert params, "no parameters to flatten"
        dev = params[0].device
        if not params[0].is_cuda:
            raise RuntimeError("semiuhpe_amd: parameters must be on the HIP device before building the optimiser "
                               "(no CPU path)")
        self.params = params
        self.offsets = []
        off = 0
        for p in params:
            assert p.device == dev and p.dtype == torch.float32
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev) if with_grad else None
        for p, o in zip(params, self.offsets):
            if not _dense(p):
                p.data = p.data.contiguous()
            view = self.data.as_strided(p.shape, p.stride(), o)
            view.copy_(p.data)
            p.data = view
            if with_grad:
                p.grad = self.grad.as_strided(p.shape, p.stride(), o)

    def view_like(self, flat, i):
        p = self.params[i]
        return flat.as_strided(p.shape, p.stride(), self.offsets[i])

    def repoint(self):
        """Re-attach parameters (and grads) to the flat storage, e.g. after load_state_dict-style `.data` swaps or
        `zero_grad(set_to_none=True)`."""
        for i, p in enumerate(self.params):
            v = self.view_like(self.data, i)
            if p.data.data_ptr() != v.data_ptr():
                v.copy_(p.data)
                p.data = v
            if self.grad is not None and (p.grad is None or p.grad.data_ptr() != self.view_like(self.grad, i).data_ptr()):
                p.grad = self.view_like(self.grad, i)


def _dense(p):
    """non-overlapping and dense (any permutation of a contiguous layout)"""
    sizes_strides = sorted(((st, sz) for sz, st in zip(p.shape, p.stride()) if sz > 1))
    expect = 1
    for st, sz in sizes_strides:
        if st != expect:
            return False
        expect *= sz
    return True


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (no amsgrad), one HIP kernel per parameter group."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat = None

    # -- flat storage ------------------------------------------------------------------------------------
    def flatten(self):
        if self._flat is not None:
            return self._flat
        self._flat = []
        for g in self.param_groups:
            fb = FlatBuffer(g["params"], with_grad=True)
            fb.exp_avg = torch.zeros_like(fb.data)
            fb.exp_avg_sq = torch.zeros_like(fb.data)
            fb.step = 0
            fb.step_t = torch.tensor(0.0)          # one host scalar shared by every state entry of the group
            self._flat.append(fb)
            for i, p in enumerate(fb.params):
                self.state[p] = {"step": fb.step_t, "exp_avg": fb.view_like(fb.exp_avg, i),
                                 "exp_avg_sq": fb.view_like(fb.exp_avg_sq, i)}
        return self._flat

    @property
    def flat_buffers(self):
        return self.flatten()

    def zero_grad(self, set_to_none=True):
        """One fill of the flat gradient buffer (the reference's set_to_none semantics would drop the views the layer
        kernels accumulate into)."""
        for fb in self.flatten():
            K.fill_(fb.grad, 0.0)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for g, fb in zip(self.param_groups, self.flatten()):
            fb.step += 1
            b1, b2 = g["betas"]
            K.adam_step(fb.data, fb.grad, fb.exp_avg, fb.exp_avg_sq, g["lr"], b1, b2, g["eps"], g["weight_decay"], fb.step)
            fb.step_t.fill_(float(fb.step))
        return loss

    # -- checkpoint compatibility -------------------------------------------------------------------------
    def load_state_dict(self, state_dict):
        self.flatten()
        super().load_state_dict(state_dict)
        for fb in self._flat:
            steps = []
            for i, p in enumerate(fb.params):
                st = self.state.get(p)
                if not st:
                    continue
                m, v = fb.view_like(fb.exp_avg, i), fb.view_like(fb.exp_avg_sq, i)
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
                steps.append(int(float(st["step"])))
                self.state[p] = {"step": fb.step_t, "exp_avg": m, "exp_avg_sq": v}
            if steps:
                fb.step = max(steps)
                fb.step_t.fill_(float(fb.step))
            fb.repoint()
