"""CPU oracle (test infrastructure) for the Rotation-Laplace head (src/laplace/rotation_laplace.py:24-173).

power(A, X) = -sqrt(max(s1+s2+s3 - tr(A^T X), 1e-8)) with s3 signed by det A (:140-173);
logF = c + log(mean_g exp(p_g - c) / (-p_g)) over the SO(3) grid (:58-72); nll = logF - p + log(-p) (:76-99);
mode = U diag(1,1,det(UV^T)) V^T (:102-115).  Pinned by tests/golden/laplace.npz (reference outputs)."""
import torch

EPS = 1e-8


def _signed_sum(A):
    S = torch.linalg.svdvals(A)
    return S[..., 0] + S[..., 1] + S[..., 2] * torch.sign(torch.det(A))


def power_fn(A, X):
    tr = (A * X).sum((-1, -2))
    return -torch.sqrt(torch.clamp_min(_signed_sum(A) - tr, EPS))


def NLL_loss(fn_type, pred, gt, grids):
    assert fn_type == "RLaplace"
    A = pred.reshape(-1, 3, 3)
    p = power_fn(A[:, None], grids[None].to(A.dtype))                    # (b, N)
    c = p.max(-1)[0]
    logF = c + torch.log((torch.exp(p - c[:, None]) / (-p)).sum(1) / grids.shape[0])
    px = power_fn(A, gt.reshape(-1, 3, 3))
    losses = logF - px + torch.log(-px)
    U, _, Vh = torch.linalg.svd(A.detach())
    d = torch.det(U @ Vh)
    mode = U @ torch.diag_embed(torch.stack((torch.ones_like(d), torch.ones_like(d), d), -1)) @ Vh
    return losses, mode
