"""CPU oracle (test infrastructure) for the matrix-Fisher rotation head.

Restates, quaternion-free, what the reference computes in
  src/fisher/fisher_utils.py:14-99, src/fisher/torch_norm_factor.py:4-92,
  src/fisher/between_bingham_fisher.py:63-200, src/fisher/bingham_utils.py:5-73.
The algebra that removes the quaternion / Bingham detour is SURVEY.md Appendix C; every
identity there is re-checked against the reference's own outputs by tests/test_oracle_head.py
(fixtures tests/golden/head*.npz, produced by tools/make_golden_head.py).

All functions are dtype-generic (float32 like the reference, or float64 for a "truth" run).
"""
import math

import torch

# Abramowitz & Stegun 9.8.1 / 9.8.2 polynomial coefficients for I0 (ascending powers), the same
# published constants the reference uses (torch_norm_factor.py:10-11).
_I0_SMALL = (1.0, 3.5156229, 3.0899424, 1.2067492, 0.2659732, 0.360768e-1, 0.45813e-2)
_I0_LARGE = (0.39894228, 0.1328592e-1, 0.225319e-2, -0.157565e-2, 0.916281e-2,
             -0.2057706e-1, 0.2635537e-1, -0.1647633e-1, 0.392377e-2)
N_QUAD = 512          # torch_norm_factor.py:69  (trapezoid nodes on [-1, 1])
LOG_2PI2 = math.log(2.0 * math.pi ** 2)


def _poly(coef, x):
    acc = torch.full_like(x, coef[-1])
    for c in reversed(coef[:-1]):
        acc = acc * x + c
    return acc


def bessel_i0e(x):
    """exp(-|x|) * I0(x) by the two A&S polynomials split at 3.75 (torch_norm_factor.py:12-19)."""
    ax = x.abs()
    small = _poly(_I0_SMALL, (ax / 3.75) ** 2) * torch.exp(-ax)
    safe = torch.where(ax > 3.75, ax, torch.full_like(ax, 3.75))
    large = _poly(_I0_LARGE, 3.75 / safe) / torch.sqrt(safe)
    return torch.where(ax <= 3.75, small, large)


def _trapz_nodes(dtype):
    # x_j = j * (2/(N-1)) - 1, weights 1 except the two ends = 1/2  (torch_norm_factor.py:21-31)
    x = torch.arange(N_QUAD, dtype=dtype) * (2.0 / (N_QUAD - 1)) - 1.0
    w = torch.ones(N_QUAD, dtype=dtype)
    w[0] = w[-1] = 0.5
    return x.view(1, -1), w.view(1, -1) * (2.0 / (N_QUAD - 1))


def _half_integral(sa, sb, sc, with_x):
    """1/2 * int_{-1}^{1} i0e((sb-sc)/2 (1-x)) i0e((sb+sc)/2 (1+x)) exp((sa+sc)(x-1)) [x] dx
    (integrand_CF :33-47 / integrand_Cdiff :50-63 once the caller has ordered sb >= sc)."""
    x, w = _trapz_nodes(sa.dtype)
    f = bessel_i0e(((sb - sc) / 2)[:, None] * (1 - x)) * bessel_i0e(((sb + sc) / 2)[:, None] * (1 + x)) \
        * torch.exp((sa + sc)[:, None] * (x - 1))
    if with_x:
        f = f * x
    return 0.5 * (f * w).sum(1)


def logc_parts(S):
    """S (n,3): proper singular values (s1>=s2>=|s3|, s3 signed).
    Returns (L, g): L = log(1/2 * integral) = logC_F(S) - sum(S)   (class_logC_F.forward :68-78)
                    g = d logC_F / dS  (class_logC_F.backward :80-91), g in [-1,1]^3."""
    S = S.detach()
    factor = _half_integral(S[:, 0], S[:, 1], S[:, 2], False)
    g = []
    for i in range(3):
        a, b, c = S[:, i], S[:, (i + 1) % 3], S[:, (i + 2) % 3]
        hi, lo = torch.maximum(b, c), torch.minimum(b, c)
        g.append(_half_integral(a, hi, lo, True) / factor)
    return torch.log(factor), torch.stack(g, 1)


class _LogCF(torch.autograd.Function):
    """logC_F with the reference's first-order custom backward (torch_norm_factor.py:66-92)."""

    @staticmethod
    def forward(ctx, S):
        L, g = logc_parts(S)
        ctx.save_for_backward(g)
        return L + S.sum(1)

    @staticmethod
    def backward(ctx, grad):
        (g,) = ctx.saved_tensors
        return g * grad[:, None]


log_c_f = _LogCF.apply


def proper_svd(A):
    """A (n,3,3) = U diag(s) V^T with det U = det V = +1 and s1>=s2>=|s3| (between_bingham_fisher.py:63-82).
    The +-1 determinants are constants for autograd (their derivative vanishes on O(3))."""
    U, S, Vh = torch.linalg.svd(A)
    V = Vh.transpose(1, 2)
    du = torch.det(U).detach()
    dv = torch.det(V).detach()
    U = torch.cat((U[:, :, :2], U[:, :, 2:] * du[:, None, None]), -1)
    V = torch.cat((V[:, :, :2], V[:, :, 2:] * dv[:, None, None]), -1)
    S = torch.cat((S[:, :2], S[:, 2:] * (du * dv)[:, None]), -1)
    return U, S, V


def batch_torch_A_to_R(A):
    """Nearest rotation U diag(1,1,det(UV^T)) V^T = (proper U)(proper V)^T  (fisher_utils.py:39-48)."""
    U, _, V = proper_svd(A.reshape(-1, 3, 3))
    return U @ V.transpose(1, 2)


def vmf_loss(net_out, R, overreg=1.05):
    """-<A,R> + overreg * logC_F(S)  and the mode rotation (fisher_utils.py:14-36)."""
    A = net_out.reshape(-1, 3, 3)
    _, S, _ = proper_svd(A)
    loss = -(A.reshape(-1, 9) * R.reshape(-1, 9)).sum(1) + overreg * log_c_f(S)
    return loss, batch_torch_A_to_R(A.detach())


def fisher_log_pdf(A, R):
    """<A,R> - logC_F(S)  (fisher_utils.py:51-67)."""
    A = A.reshape(-1, 3, 3)
    _, S, _ = proper_svd(A)
    return (A.reshape(-1, 9) * R.reshape(-1, 9)).sum(1) - log_c_f(S)


def fisher_entropy(A):
    """H = logC_F(S) - sum_i s_i dlogC/ds_i  (== fisher_utils.py:70-81 via bingham_entropy)."""
    _, S, _ = proper_svd(A.reshape(-1, 3, 3).detach())
    L, g = logc_parts(S)
    return L + S.sum(1) - (S * g).sum(1)


def _bingham_weights(g):
    """bingham_dF/bingham_F for the three non-zero Bingham parameters (bingham_utils.py:59-73)."""
    g1, g2, g3 = g[:, 0], g[:, 1], g[:, 2]
    return torch.stack((0.25 * (g1 - g2 - g3) + 0.25, 0.25 * (-g1 + g2 - g3) + 0.25,
                        0.25 * (-g1 - g2 + g3) + 0.25), 1)


def _branches(U, V):
    """P = U V^T and M_j = U (2 e_j e_j^T - I) V^T = 2 u_j v_j^T - P  (between_bingham_fisher.py:118-126)."""
    P = U @ V.transpose(1, 2)
    M = [2 * U[:, :, j:j + 1] @ V[:, :, j:j + 1].transpose(1, 2) - P for j in range(3)]
    return P, torch.stack(M, 1)              # (n,3,3), (n,3,3,3)


def fisher_CE(A1, A2):
    """Cross entropy teacher A1 (no grad) -> student A2 AS WRITTEN in the reference, including the
    row-index quirk of bingham_CE (bingham_utils.py:24-29):
        CE = L(s2) - sum_i Lam2_i [ beta_i + sum_k (alpha_ik - beta_i) w1_k ]
        beta_i  = (1 + <P1, M2_i>)/4      alpha_ik = (1 + <M1_i, M2_k>)/4
    (fisher_utils.py:84-99).  Gradient flows to A2 through s2 and through the singular vectors."""
    A1 = A1.reshape(-1, 3, 3).detach()
    A2 = A2.reshape(-1, 3, 3)
    U1, S1, V1 = proper_svd(A1)
    U2, S2, V2 = proper_svd(A2)
    _, g1 = logc_parts(S1)
    w1 = _bingham_weights(g1)                                   # (n,3)
    P1, M1 = _branches(U1, V1)
    _, M2 = _branches(U2, V2)
    lam2 = torch.stack((-2 * (S2[:, 1] + S2[:, 2]), -2 * (S2[:, 0] + S2[:, 2]), -2 * (S2[:, 0] + S2[:, 1])), 1)
    beta = (1 + (P1[:, None] * M2).sum((2, 3))) / 4             # (n,3)  index i (student branch)
    alpha = (1 + (M1[:, :, None] * M2[:, None, :]).sum((3, 4))) / 4   # (n,i,k): teacher i, student k
    inner = beta + ((alpha - beta[:, :, None]) * w1[:, None, :]).sum(2)
    L2 = log_c_f(S2) - S2.sum(1)
    ce = L2 - (lam2 * inner).sum(1)
    assert torch.isfinite(ce).all()                              # fisher_utils.py:98
    return ce


def rotation_adjust(pred_weak, aug_rot_mat, train_labeled):
    """Pose-altering rotation consistency on the teacher's 9-D output (agent.py:110-122)."""
    A = pred_weak.reshape(-1, 3, 3)
    if train_labeled == "DAD3DHeads":
        return (aug_rot_mat @ A).reshape(-1, 9)
    if train_labeled == "300WLP":
        r180 = torch.diag(torch.tensor([1.0, -1.0, -1.0], dtype=A.dtype))
        return (r180 @ (aug_rot_mat @ (r180 @ A.transpose(1, 2)))).transpose(1, 2).reshape(-1, 9)
    raise ValueError(train_labeled)
