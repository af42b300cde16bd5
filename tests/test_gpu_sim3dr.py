"""f4 (demo renderer) on the GPU: suhpe_mesh_normals / suhpe_rasterize (csrc/raster.hip) behind the reference's Sim3DR API against the
CPU oracle bit for bit (random scenes: triangles off-frame, degenerate triangles, exact depth ties, frames that are not multiples of
the 16-pixel tile, 1 / 3 / 4 channels, alpha < 1, reversed rows) and the whole Renderer against the images the reference produced."""
import ctypes
import os

import numpy as np
import pytest
import torch

from test_oracle_sim3dr import G, O, mesh, random_scene  # noqa: F401  (O: the oracle fixture)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,alpha,reverse,c,hw", [(0, 1.0, False, 3, (70, 90)), (1, 0.6, True, 3, (64, 48)), (2, 0.25, False, 1, (33, 130)),
                                                       (3, 1.0, True, 4, (17, 15)), (4, 0.9, False, 3, (224, 224))])
def test_rasterize_and_normals_bit_identical_to_oracle(O, seed, alpha, reverse, c, hw):
    from semiuhpe_amd import Sim3DR
    v, t, col, img = random_scene(seed, h=hw[0], w=hw[1], c=c, nver=600 if hw[0] > 100 else 400, ntri=2500 if hw[0] > 100 else 900)
    a, b = img.copy(), img.copy()
    O.rasterize(v, t, col, bg=a, alpha=alpha, reverse=reverse)
    out = Sim3DR.rasterize(v, t, col, bg=b, alpha=alpha, reverse=reverse)
    assert out is b and np.array_equal(a, b) and not np.array_equal(a, img)
    touched = np.unique(t)
    with np.errstate(invalid="ignore", divide="ignore"):
        n_cpu = O.get_normal(v, t)
    n_gpu = Sim3DR.get_normal(v, t)
    assert np.array_equal(n_cpu[touched], n_gpu[touched], equal_nan=True)


def test_renderer_reproduces_the_reference_images():
    from semiuhpe_amd import Sim3DR
    from semiuhpe_amd.src.renderer import Renderer
    verts, tris = mesh()
    R = Renderer(vertices=verts, triangles=tris)
    tv = R.transform_vertices(G["frame"], [np.array(p) for p in G["poses"]])
    assert np.array_equal(Sim3DR.get_normal(np.ascontiguousarray(tv[0], dtype=np.float32), R.triangles), G["normal0"])
    over = G["frame"].copy()
    for v, stage in zip(tv, (G["stage0"], G["stage1"])):
        over = R.render_app(np.ascontiguousarray(v, dtype=np.float32), R.triangles, over)
        assert np.array_equal(over, stage)
    res = R.render(G["frame"], tv, alpha=0.9)
    assert np.array_equal(res, np.clip(np.rint((1 - 0.9) * G["frame"].astype(np.float64) + 0.9 * G["stage1"]), 0, 255).astype(np.uint8))
    img = G["frame"].copy()
    Sim3DR.rasterize(np.ascontiguousarray(tv[0], dtype=np.float32), R.triangles, G["colors"], bg=img, alpha=0.6, reverse=True)
    assert np.array_equal(img, G["img_alpha06_reversed"])


def test_rasterize_c_abi_errors():
    from semiuhpe_amd import _lib
    lib = _lib.load()
    buf = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    p = ctypes.c_void_p(buf.data_ptr())
    assert lib.suhpe_rasterize(p, p, p, p, p, 10, 8, 8, 5, ctypes.c_float(1.0), 0, p, 1024, None) == -1        # 5 channels
    assert lib.suhpe_rasterize(p, p, p, p, p, 100, 8, 8, 3, ctypes.c_float(1.0), 0, p, 16, None) != 0          # workspace too small
    assert lib.suhpe_rasterize(None, p, p, p, p, 10, 8, 8, 3, ctypes.c_float(1.0), 0, p, 1024, None) == -1
    assert lib.suhpe_rasterize_workspace_bytes(100) == 1600
