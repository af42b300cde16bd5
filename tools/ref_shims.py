"""Import shims that let the *reference* (/root/reference, read-only) be imported in the
build container, which has no GPU and lacks cv2 / torchvision / timm / pytorch3d / pytorchcv.

Used only by tools/make_golden*.py to GENERATE golden vectors (tests/golden/*.npz).
Nothing here travels into the product path and nothing here is reference source:
the stubs restate the public behaviour of the third-party functions the reference calls.

Call `install()` BEFORE importing anything from the reference.
"""
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


# ---------------------------------------------------------------- pytorch3d.transforms (0.7.2)
def quaternion_to_matrix(q):
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
            two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
            two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j),
        ),
        -1,
    )
    return o.reshape(q.shape[:-1] + (3, 3))


def _sqrt_positive_part(x):
    ret = torch.zeros_like(x)
    pos = x > 0
    ret[pos] = torch.sqrt(x[pos])
    return ret


def matrix_to_quaternion(matrix):
    batch_dim = matrix.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(
        matrix.reshape(batch_dim + (9,)), dim=-1)
    q_abs = _sqrt_positive_part(
        torch.stack(
            [1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22,
             1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1))
    quat_by_rijk = torch.stack(
        [
            torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
            torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
            torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
            torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1),
        ],
        dim=-2,
    )
    flr = torch.tensor(0.1).to(dtype=q_abs.dtype, device=q_abs.device)
    quat_candidates = quat_by_rijk / (2.0 * q_abs[..., None].max(flr))
    idx = q_abs.argmax(dim=-1)
    onehot = torch.nn.functional.one_hot(idx, num_classes=4) > 0.5
    return quat_candidates[onehot, :].reshape(batch_dim + (4,))


def _acos_linear_extrapolation(x, bound=1.0 - 1e-4):
    # pytorch3d.transforms.math.acos_linear_extrapolation with bounds (-b, b)
    import math
    lower, upper = -bound, bound
    acos_x = torch.empty_like(x)
    inside = (x > lower) & (x < upper)
    acos_x[inside] = torch.acos(x[inside])

    def extrap(v, x0):
        return (v - x0) * (-1.0 / math.sqrt(1.0 - x0 * x0)) + math.acos(x0)

    hi = x >= upper
    lo = x <= lower
    acos_x[hi] = extrap(x[hi], upper)
    acos_x[lo] = extrap(x[lo], lower)
    return acos_x


def so3_relative_angle(R1, R2, cos_angle=False, cos_bound=1e-4, eps=1e-4):
    R12 = torch.bmm(R1, R2.permute(0, 2, 1))
    rot_trace = R12[:, 0, 0] + R12[:, 1, 1] + R12[:, 2, 2]
    if ((rot_trace < -1.0 - eps) + (rot_trace > 3.0 + eps)).any():
        raise ValueError("A matrix has trace outside valid range [-1-eps,3+eps].")
    phi_cos = (rot_trace - 1.0) * 0.5
    if cos_angle:
        return phi_cos
    if cos_bound > 0.0:
        return _acos_linear_extrapolation(phi_cos, 1.0 - cos_bound)
    return torch.acos(phi_cos)


# ---------------------------------------------------------------- timm pieces
class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        mask = x.new_empty(shape).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


def _to_2tuple(x):
    if isinstance(x, (tuple, list)):
        return tuple(x)
    return (x, x)


def _build_model_with_cfg(cls, variant, pretrained, **kwargs):
    kwargs.pop("pretrained_cfg", None)
    kwargs.pop("pretrained_filter_fn", None)
    return cls(**kwargs)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    """Install every shim. Idempotent."""
    if getattr(install, "_done", False):
        return
    install._done = True
    import matplotlib
    matplotlib.use("Agg")

    # no GPU in the build container: make .cuda() the identity
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self

    _mod("cv2", imwrite=lambda *a, **k: True)
    tv = _mod("torchvision")
    tv.models = _mod("torchvision.models")
    pcv = _mod("pytorchcv")
    pcv.model_provider = _mod("pytorchcv.model_provider", get_model=lambda *a, **k: None)

    timm = _mod("timm", __version__="0.6.13")
    timm.models = _mod("timm.models")
    timm.models.layers = _mod("timm.models.layers", DropPath=DropPath, to_2tuple=_to_2tuple,
                              trunc_normal_=nn.init.trunc_normal_)
    timm.models.registry = _mod("timm.models.registry", register_model=lambda f: f)
    timm.models._builder = _mod("timm.models._builder", build_model_with_cfg=_build_model_with_cfg)

    p3d = _mod("pytorch3d")
    p3d.transforms = _mod("pytorch3d.transforms", quaternion_to_matrix=quaternion_to_matrix,
                          matrix_to_quaternion=matrix_to_quaternion,
                          so3_relative_angle=so3_relative_angle)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
