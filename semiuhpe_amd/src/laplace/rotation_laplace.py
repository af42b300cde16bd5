"""Drop-in for the reference's src/laplace/rotation_laplace.py (Rotation-Laplace head): same `NLL_loss(fn_type, pred,
gt, grids)` -> (losses (b,), pred_orth (b,3,3)); one HIP launch per call (value, analytic gradient, analytical mode)
instead of a (b, 4608, 3, 3) broadcast matmul plus CPU svdvals."""
import os

import numpy as np
import torch

from ... import kernels as K


def default_grids(device):
    """the reference always loads eq_grids2.npy (4608 rotations) for both ranges (src/agent.py:46-58)"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "eq_grids2.npy")
    return torch.from_numpy(np.load(path)).to(device).contiguous()


class _LaplaceNLL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, grids):
        A = pred.reshape(-1, 9)
        loss, dA, mode = K.laplace_nll(A if A.is_contiguous() else A.contiguous(), gt.reshape(-1, 9).contiguous(), grids)
        ctx.save_for_backward(dA)
        ctx.shape = pred.shape
        ctx.mark_non_differentiable(mode)
        return loss, mode

    @staticmethod
    def backward(ctx, g, _):
        (dA,) = ctx.saved_tensors
        return K.scale_rows(dA, g.contiguous()).view(ctx.shape), None, None


def NLL_loss(fn_type, pred, gt, grids):
    if fn_type != "RLaplace":
        raise NotImplementedError("only the 'RLaplace' power function is used by the reference's agent (src/agent.py:81,162)")
    return _LaplaceNLL.apply(pred, gt, grids)


def analytical_mode(pred, fn_type="RLaplace"):
    return K.fisher_svd_project(pred.detach().reshape(-1, 9).contiguous())
