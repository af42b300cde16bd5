"""f4 (demo renderer), CPU side: the plain-C oracle (oracle/sim3dr_oracle.c) against the golden vectors the reference produced
(tools/make_golden_sim3dr.py: the reference's C++ under the reference's Python) and, where the compiled reference is present
(oracle/_ref, built by `make -C oracle` in the build container and shipped to the GPU box), against the reference directly on
random meshes; the host-side pose / vertex transforms of the product against the same goldens."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "sim3dr.npz"))


@pytest.fixture(scope="module")
def O():
    if not os.path.exists(os.path.join(ROOT, "oracle", "libsim3dr_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    from oracle import sim3dr
    return sim3dr


def mesh():
    from semiuhpe_amd.synthetic import synthetic_head_mesh
    return synthetic_head_mesh()


def random_scene(seed, nver=400, ntri=900, h=70, w=90, c=3):
    """vertices partly outside the frame, a few degenerate (repeated-vertex) triangles, depths with exact ties"""
    rng = np.random.RandomState(seed)
    v = np.stack([rng.uniform(-15, w + 15, nver), rng.uniform(-15, h + 15, nver), np.round(rng.uniform(0, 40, nver), 1)], 1).astype(np.float32)
    t = rng.randint(0, nver, (ntri, 3)).astype(np.int32)
    t[::97, 1] = t[::97, 0]
    near = rng.randint(0, nver, ntri // 2)
    t[: ntri // 2] = np.stack([near, (near + 1) % nver, (near + 7) % nver], 1)       # small triangles between neighbours too
    v[1::2, :2] = v[::2, :2][: len(v[1::2])] + rng.uniform(-6, 6, (len(v[1::2]), 2)).astype(np.float32)
    return v, t, rng.rand(nver, c).astype(np.float32), rng.randint(0, 256, (h, w, c)).astype(np.uint8)


def test_oracle_matches_reference_goldens(O):
    verts, tris = mesh()
    tv0, tv1 = G["tv0"].astype(np.float32), G["tv1"].astype(np.float32)
    assert np.array_equal(O.get_normal(np.ascontiguousarray(tv0), tris), G["normal0"])
    img = G["frame"].copy()
    O.rasterize(np.ascontiguousarray(tv0), tris, G["colors"], bg=img, alpha=0.6, reverse=True)
    assert np.array_equal(img, G["img_alpha06_reversed"])
    pipe = O.RenderPipeline()
    over = G["frame"].copy()
    for tv, stage in ((tv0, G["stage0"]), (tv1, G["stage1"])):
        over = pipe(np.ascontiguousarray(tv), tris, over)
        assert np.array_equal(over, stage)
    res = O.add_weighted(G["frame"], 0.1, over, 0.9)
    assert res.dtype == np.uint8 and int(np.abs(res.astype(int) - (0.1 * G["frame"] + 0.9 * over)).max()) <= 1


@pytest.mark.parametrize("seed,alpha,reverse,c", [(0, 1.0, False, 3), (1, 0.6, True, 3), (2, 0.25, False, 1), (3, 1.0, True, 4)])
def test_oracle_matches_compiled_reference(O, seed, alpha, reverse, c):
    if not O.have_ref():
        pytest.skip("oracle/_ref/libsim3dr_ref.so not built (needs /root/reference): the goldens pin the oracle instead")
    v, t, col, img = random_scene(seed, c=c)
    a, b = img.copy(), img.copy()
    O.rasterize(v, t, col, bg=a, alpha=alpha, reverse=reverse)
    O.rasterize(v, t, col, bg=b, alpha=alpha, reverse=reverse, ref=True)
    assert np.array_equal(a, b) and not np.array_equal(a, img)
    with np.errstate(invalid="ignore", divide="ignore"):
        na, nb = O.get_normal(v, t), O.get_normal(v, t, ref=True)
    assert np.array_equal(na, nb, equal_nan=True)                # (vertices no triangle touches: 0 / 0, as the reference)


def test_host_side_pose_and_vertex_transforms_match_reference():
    from semiuhpe_amd.src import vis_plot
    from semiuhpe_amd.src.renderer import Renderer, add_weighted
    K = G["intrinsics"]
    poses = [vis_plot.convert_euler_bbox_to_6dof(list(e), tuple(int(x) for x in b), K) for e, b in zip(G["eulers"], G["boxes"])]
    assert np.allclose(poses, G["poses"], rtol=1e-12, atol=1e-12)
    assert np.allclose(vis_plot.convert_rotmat_bbox_to_6dof(G["rot"], tuple(int(x) for x in G["boxes"][0]), K), G["pose_rotmat"], rtol=1e-12, atol=1e-12)
    verts, tris = mesh()
    R = Renderer(vertices=verts, triangles=tris)
    tv = R.transform_vertices(G["frame"], [np.array(p) for p in G["poses"]])
    assert np.array_equal(tv[0], G["tv0"]) and np.array_equal(tv[1], G["tv1"])
    assert np.array_equal(add_weighted(G["frame"], 0.1, G["stage1"], 0.9), np.clip(np.rint(0.1 * G["frame"].astype(np.float64) + 0.9 * G["stage1"]), 0, 255))
    img = np.zeros((60, 80, 3), np.uint8)
    vis_plot.draw_axis_ypr(img, 20.0, -10.0, 5.0, size=25)
    assert (img[:, :, 2] == 255).any() and (img[:, :, 1] == 255).any() and (img[:, :, 0] == 255).any()
