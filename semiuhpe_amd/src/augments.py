"""Drop-in for the reference's src/augments.py (tensor CutOut / CutMix).

Box parameters are drawn on the host with the reference's exact torch-RNG call order (so a seeded run produces
the same boxes), uploaded asynchronously, and applied by ONE kernel (suhpe_strong_aug) instead of 2*N*3 slice
assignments with 12*N scalar device->host syncs.
"""
import torch

from .. import kernels as K


def draw_boxes(N, width, height, mask_holes_num=3, normal=False):
    """reference src/augments.py:15-30: normal(cx), normal(cy), randint(size) on the CPU generator -> int32 (N,holes,4)
    [x0,y0,x1,y1]."""
    if not normal:
        cx = torch.randint(20, width - 20, (N, mask_holes_num)).int()
        cy = torch.randint(20, height - 20, (N, mask_holes_num)).int()
    else:
        cx = torch.normal(width / 2.0, width / 6.0, size=(N, mask_holes_num)).int()
        cy = torch.normal(height / 2.0, height / 6.0, size=(N, mask_holes_num)).int()
    size = torch.randint(10, 20, (N, mask_holes_num, 2)).int()
    x0 = (cx - size[..., 0]).clamp_(0, width)
    y0 = (cy - size[..., 1]).clamp_(0, height)
    x1 = (cx + size[..., 0]).clamp_(0, width)
    y1 = (cy + size[..., 1]).clamp_(0, height)
    return torch.stack((x0, y0, x1, y1), -1).contiguous()


def _to_dev(t, device):
    return t.pin_memory().to(device, non_blocking=True) if torch.cuda.is_available() else t.to(device)


def random_cutout_tensor(image, mask_holes_num=3, normal=False):
    """reference :12-35.  image: (N,C,W,H) device tensor, modified in place and returned."""
    N, _, width, height = image.shape
    boxes = _to_dev(draw_boxes(N, width, height, mask_holes_num, normal), image.device)
    out = K.strong_aug(image.contiguous(), boxes, None, None, mask_holes_num, src_nhwc=False, dst_nhwc=False)
    image.copy_(out)
    return image


def random_cutmix_tensor(image, mask_holes_num=3, normal=False):
    """reference :67-94."""
    N, _, width, height = image.shape
    boxes = _to_dev(draw_boxes(N, width, height, mask_holes_num, normal), image.device)
    perm = _to_dev(torch.randperm(N).int(), image.device)
    out = K.strong_aug(image.contiguous(), None, boxes, perm, mask_holes_num, src_nhwc=False, dst_nhwc=False)
    image.copy_(out)
    return image


def strong_view_nhwc(image, cutout, cutmix, mask_holes_num=3, normal=True, out=None):
    """Fused path used by the agent: CutOut then CutMix (same RNG order as calling the two functions above in
    sequence) while converting the NCHW loader batch to the backbone's NHWC layout; one read, one write.  The step's random tables
    (CutOut boxes, CutMix boxes, permutation: host RNG, the reference's draw order) travel in ONE pinned upload."""
    N, _, width, height = image.shape
    co = draw_boxes(N, width, height, mask_holes_num, normal) if cutout else None
    cm = perm = None
    if cutmix:
        cm = draw_boxes(N, width, height, mask_holes_num, normal)
        perm = torch.randperm(N).int()
    parts = [t for t in (co, cm, perm) if t is not None]
    if parts:
        sizes = [t.numel() for t in parts]
        packed = _to_dev(torch.cat([t.reshape(-1) for t in parts]), image.device)
        views, off = [], 0
        for t, n in zip(parts, sizes):
            views.append(packed[off:off + n].view(t.shape))
            off += n
        it = iter(views)
        co = next(it) if co is not None else None
        cm = next(it) if cm is not None else None
        perm = next(it) if perm is not None else None
    return K.strong_aug(image.contiguous(), co, cm, perm, mask_holes_num, src_nhwc=False, dst_nhwc=True, out=out)
