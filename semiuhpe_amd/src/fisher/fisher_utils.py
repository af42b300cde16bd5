"""Drop-in for the reference's src/fisher/fisher_utils.py — same names, arguments and return values — with the
math in one HIP kernel per call (semiuhpe_amd/csrc/head.hip) instead of .cpu() -> torch.svd -> .cuda().

Each function is a torch.autograd.Function whose backward multiplies the per-sample analytic gradient produced by
the forward kernel with the incoming gradient (suhpe_scale_rows); no host synchronisation anywhere.
"""
import torch

from ... import kernels as K


def _rows9(A):
    A = A.reshape(-1, 9)
    return A if A.is_contiguous() else A.contiguous()


class _FisherNLL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, R, overreg):
        loss, dA, Rest = K.fisher_nll(_rows9(A), _rows9(R), overreg)
        ctx.save_for_backward(dA)
        ctx.shape = A.shape
        ctx.mark_non_differentiable(Rest)
        return loss, Rest

    @staticmethod
    def backward(ctx, gloss, _grest):
        (dA,) = ctx.saved_tensors
        return K.scale_rows(dA, gloss.contiguous()).view(ctx.shape), None, None


def vmf_loss(net_out, R, overreg=1.05):
    """reference src/fisher/fisher_utils.py:14-18 -> (loss (b,), Rest (b,3,3))"""
    return _FisherNLL.apply(net_out, R, overreg)


def KL_Fisher(A, R, overreg=1.05):
    """reference :21-36"""
    return _FisherNLL.apply(A, R, overreg)[0]


def batch_torch_A_to_R(A):
    """reference :39-48"""
    return K.fisher_svd_project(_rows9(A.detach()))


def fisher_log_pdf(A, R):
    """reference :51-67 (no gradient is taken through it anywhere in the reference)"""
    return K.fisher_log_pdf(_rows9(A.detach()), _rows9(R))


def fisher_entropy(A):
    """reference :70-81"""
    return K.fisher_entropy(_rows9(A.detach()))


class _FisherCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A1, A2, mask):
        ce, dA2 = K.fisher_ce(_rows9(A1), _rows9(A2), mask)
        ctx.save_for_backward(dA2)
        ctx.shape = A2.shape
        return ce

    @staticmethod
    def backward(ctx, g):
        (dA2,) = ctx.saved_tensors
        return None, K.scale_rows(dA2, g.contiguous()).view(ctx.shape), None


def fisher_CE(A1, A2, mask=None):
    """reference :84-99 (A1 = teacher / target, A2 = prediction).  `mask` (float 0/1 per sample, optional, additive
    to the reference's signature) zeroes unselected samples inside the kernel instead of compacting the batch with
    a host sync.  The reference's `assert not isnan/isinf` is replaced by a device-side finite flag
    (`semiuhpe_amd.src.agent.SSLAgent.check_finite`)."""
    return _FisherCE.apply(A1.detach(), A2, mask)
