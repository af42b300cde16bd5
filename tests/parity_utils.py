"""Helpers shared by the GPU parity tests: identical Bernoulli masks on both paths, gradient comparison."""
import numpy as np
import torch


class MaskTape:
    """Records the masks the oracle draws (own seeded generator) and replays them, in order, to the HIP path."""

    def __init__(self, seed):
        self.gen = torch.Generator().manual_seed(seed)
        self.masks = []
        self.cursor = 0

    def record(self, kind, shape, keep):
        m = (torch.rand(shape, generator=self.gen) < keep).float() / keep
        self.masks.append((kind, tuple(shape), m))
        return m

    def replay(self, kind, shape, keep):
        k, s, m = self.masks[self.cursor]
        assert k == kind and s == tuple(shape), f"mask order mismatch: oracle drew {k}{s}, HIP asks {kind}{tuple(shape)}"
        self.cursor += 1
        return m

    def replay_merged(self, kind, shape, keep):
        """For an agent that runs its two student passes as one batch: the oracle drew [labeled pass | teacher pass | strong
        pass], K masks each.  The merged student asks for K masks over the concatenated batch; the teacher (which may be
        enqueued before or after the student) for its K.  Requests are told apart by their batch dimension."""
        k3 = len(self.masks) // 3
        if not hasattr(self, "cur_student"):
            self.cur_student, self.cur_teacher = 0, 0
        (ka, sa, ma), (kb, sb, mb) = self.masks[self.cur_student % k3], self.masks[2 * k3 + self.cur_student % k3]
        if self.cur_student < k3 and tuple(shape) == (sa[0] + sb[0],) + sa[1:] and ka == kb == kind:
            self.cur_student += 1
            return torch.cat([ma, mb])
        k, s_, m = self.masks[k3 + self.cur_teacher]
        assert k == kind and s_ == tuple(shape), f"mask order mismatch: oracle drew {k}{s_}, HIP asks {kind}{tuple(shape)}"
        self.cur_teacher += 1
        return m

    def done(self):
        if hasattr(self, "cur_student"):
            return self.cur_student == self.cur_teacher == len(self.masks) // 3
        return self.cursor == len(self.masks)


def rel_err(x, ref):
    x = x.detach().cpu().double().reshape(-1)
    ref = ref.detach().cpu().double().reshape(-1)
    return float((x - ref).abs().max() / max(1e-12, float(ref.abs().max())))


def grad_mismatches(named_ref, named_got, rtol, atol_frac=1e-5):
    """Per-tensor max |got-ref| <= rtol*max|ref| + atol_frac*global_max|ref|.  The absolute term covers gradients
    that are analytically zero (e.g. a bias feeding a train-mode BatchNorm) and hold only rounding noise."""
    gmax = max(float(g.abs().max()) for g in named_ref.values())
    bad = {}
    for n, g in named_ref.items():
        got = named_got[n].detach().cpu().double()
        err = float((got - g.double()).abs().max())
        lim = rtol * float(g.abs().max()) + atol_frac * gmax
        if err > lim:
            bad[n] = (err, lim)
    return bad


def seeded_state_dict(template, seed):
    """Deterministic weights for a module tree too large to commit: walk `template` (a state_dict) in key order with
    one seeded generator.  Conv/Linear weights ~ N(0, 1/fan_in) * 1.5, BN/LN weight ~ U(0.6,1.4), biases ~ N(0,0.05),
    running_mean ~ N(0,0.1), running_var ~ U(0.6,1.4), attention biases ~ N(0,0.5); integer buffers are kept."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in template.items():
        if not torch.is_floating_point(v):
            out[k] = v.clone()
            continue
        shape = tuple(v.shape)
        if k.endswith("running_var"):
            t = torch.rand(shape, generator=g) * 0.8 + 0.6
        elif k.endswith("running_mean"):
            t = torch.randn(shape, generator=g) * 0.1
        elif k.endswith("attention_biases"):
            t = torch.randn(shape, generator=g) * 0.5
        elif v.ndim == 1 and k.endswith("weight"):
            t = torch.rand(shape, generator=g) * 0.8 + 0.6
        elif v.ndim == 1:
            t = torch.randn(shape, generator=g) * 0.05
        else:
            fan_in = int(np.prod(shape[1:]))
            t = torch.randn(shape, generator=g) * (1.5 / fan_in ** 0.5)
        out[k] = t.to(v.dtype)
    return out


def fp32_noise_samples(model32, x, dy, k=2, off=()):
    """More draws of the fp32 rounding noise of a whole-net gradient: the SAME fp32 model on inputs moved by one unit in the last place
    (up / down per element, seeded).  Through dozens of train-mode BatchNorm + ReLU layers on a handful of images the distance of an fp32
    gradient from the exact (float64) one is a chaotic quantity -- one reference run is ONE sample of it, and a yardstick built from a
    single sample makes any other rounding pattern look like an error half of the time.  Returns k dicts {name: grad}; the tests take the
    largest distance to the float64 truth over the reference run and these as the noise of the problem (the control of
    tools/mae_parity.drift_trend, applied to one backward pass).  `off`: module classes whose `p` is set to 0 (dropout / stochastic depth)."""
    import copy
    out = []
    for i in range(k):
        m = copy.deepcopy(model32).float().train()
        for mod in m.modules():
            if off and isinstance(mod, tuple(off)):
                mod.p = 0.0
        for q in m.parameters():
            q.grad = None
        g = torch.Generator().manual_seed(4242 + i)
        up = torch.rand(x.shape, generator=g) < 0.5
        xi = torch.where(up, torch.nextafter(x, torch.full_like(x, float("inf"))), torch.nextafter(x, torch.full_like(x, float("-inf"))))
        m(xi).backward(dy)
        out.append({n: q.grad.detach().clone() for n, q in m.named_parameters()})
    return out
