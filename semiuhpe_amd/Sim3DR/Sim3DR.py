"""`get_normal` / `rasterize` with the reference's signatures (Sim3DR/Sim3DR.py:8-29), computed by suhpe_mesh_normals /
suhpe_rasterize (csrc/raster.hip) on the current HIP device.  numpy arrays in and out, as the reference's callers expect
(src/renderer.py, Sim3DR/lighting.py); the arrays cross to the device for the call.  No CPU path: without libsuhpe.so or a GPU the
calls raise (semiuhpe_amd._lib)."""
import hashlib

import numpy as np
import torch

from .. import _lib
from .._lib import call, ptr

_adjacency = {}


def _vertex_triangles(triangles, nver):
    """vertex -> triangles in CSR form, each vertex's triangles in increasing index (the order _get_normal accumulates them in,
    rasterize_kernel.cpp:191-201); cached per mesh (the demo renders one fixed mesh)"""
    # keyed on the CONTENT: `triangles` may be a temporary of get_normal (its address is reused by the next mesh of the same shape)
    key = (hashlib.blake2b(triangles.tobytes(), digest_size=16).digest(), triangles.shape[0], nver)
    hit = _adjacency.get(key)
    if hit is None:
        flat = triangles.reshape(-1).astype(np.int64)
        order = np.argsort(flat, kind="stable")                 # stable: ties (same vertex) keep triangle order
        start = np.zeros(nver + 1, dtype=np.int32)
        np.cumsum(np.bincount(flat, minlength=nver), out=start[1:])
        tris = (order // 3).astype(np.int32)
        hit = (torch.from_numpy(start).cuda(), torch.from_numpy(tris).cuda())
        if len(_adjacency) > 8:
            _adjacency.clear()
        _adjacency[key] = hit
    return hit


def get_normal(vertices, triangles):
    _lib.require_gpu()
    vertices = np.ascontiguousarray(vertices, dtype=np.float32)
    triangles = np.ascontiguousarray(triangles, dtype=np.int32)
    nver, ntri = vertices.shape[0], triangles.shape[0]
    start, tris = _vertex_triangles(triangles, nver)
    v, t = torch.from_numpy(vertices).cuda(), torch.from_numpy(triangles).cuda()
    normal = torch.empty_like(v)
    call("suhpe_mesh_normals", ptr(v), ptr(t), ptr(start), ptr(tris), nver, ntri, ptr(normal))
    return normal.cpu().numpy()


def rasterize(vertices, triangles, colors, bg=None, height=None, width=None, channel=None, reverse=False, alpha=1.0):
    """paints into `bg` IN PLACE and returns it, as the reference does (rasterize.pyx:90-102 writes through the array pointer)"""
    _lib.require_gpu()
    if bg is not None:
        height, width, channel = bg.shape
    else:
        assert height is not None and width is not None and channel is not None
        bg = np.zeros((height, width, channel), dtype=np.uint8)
    assert bg.dtype == np.uint8
    if colors.dtype != np.float32:
        colors = colors.astype(np.float32)
    v = torch.from_numpy(np.ascontiguousarray(vertices, dtype=np.float32)).cuda()
    t = torch.from_numpy(np.ascontiguousarray(triangles, dtype=np.int32)).cuda()
    col = torch.from_numpy(np.ascontiguousarray(colors)).cuda()
    img = torch.from_numpy(np.ascontiguousarray(bg)).cuda()
    depth = torch.full((height, width), -1e8, dtype=torch.float32, device="cuda")       # Sim3DR.py:24
    ntri = t.shape[0]
    nbytes = _lib.load().suhpe_rasterize_workspace_bytes(ntri)
    ws = _lib.workspace("raster", nbytes)
    call("suhpe_rasterize", ptr(img), ptr(v), ptr(t), ptr(col), ptr(depth), ntri, height, width, channel, float(alpha), int(bool(reverse)), ptr(ws),
         ws.numel())
    bg[...] = img.cpu().numpy()
    return bg
