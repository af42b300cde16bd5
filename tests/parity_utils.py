"""Helpers shared by the GPU parity tests: identical Bernoulli masks on both paths, gradient comparison."""
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

    def done(self):
        return self.cursor == len(self.masks)


def rel_err(x, ref):
    x = x.detach().cpu().double().reshape(-1)
    ref = ref.detach().cpu().double().reshape(-1)
    return float((x - ref).abs().max() / max(1e-12, float(ref.abs().max())))
