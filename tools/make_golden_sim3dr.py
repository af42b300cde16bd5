"""Golden vectors of the f4 row (demo renderer) from the REFERENCE itself, run in the build container:
  * the reference's C++ (Sim3DR/lib/rasterize_kernel.cpp, compiled where it lies by oracle/Makefile -> oracle/_ref/libsim3dr_ref.so),
  * under the reference's own Python: Sim3DR/lighting.py (RenderPipeline), src/renderer.py (Renderer.transform_vertices / render_app),
    src/vis_plot.py (pose conversions) -- imported from /root/reference with two stubs: `Sim3DR_Cython` (the py3.8 extension
    module that cannot load here) forwards to the compiled reference C++ through ctypes exactly as rasterize.pyx:63-102 does, and
    `cv2` (absent; only imported, not called on this path).
Inputs are the seeded synthetic mesh of semiuhpe_amd.synthetic.synthetic_head_mesh and a seeded frame; outputs go to
tests/golden/sim3dr.npz.      python tools/make_golden_sim3dr.py
"""
import os, sys, tempfile, types
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import sim3dr as O                                   # only for the ctypes binding of oracle/_ref (ref=True)
from semiuhpe_amd.synthetic import synthetic_head_mesh

assert O.have_ref(), "run `make -C oracle` first (needs /root/reference)"
cy = types.ModuleType("Sim3DR_Cython")
def _get_normal(normal, vertices, triangles, nver, ntri):
    normal[...] = O.get_normal(vertices, triangles, ref=True)
def _rasterize(bg, vertices, triangles, colors, buffer, ntri, h, w, c, alpha=1, reverse=False):
    import ctypes
    fn = O._lib(True)._Z10_rasterizePhPfPiS0_S0_iiiifb
    fn.restype = None
    fn(O._p(bg), O._p(vertices), O._p(triangles), O._p(colors), O._p(buffer), ctypes.c_int(ntri), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_int(c),
       ctypes.c_float(alpha), ctypes.c_bool(reverse))
cy.get_normal, cy.rasterize = _get_normal, _rasterize
sys.modules["Sim3DR_Cython"] = cy
sys.modules["cv2"] = types.ModuleType("cv2")
sys.path.insert(0, "/root/reference")
import Sim3DR as RefSim3DR                                       # the reference package (lighting.py, Sim3DR.py)
from src import renderer as ref_renderer, vis_plot as ref_vis

verts, tris = synthetic_head_mesh()
tmp = tempfile.mkdtemp()
np.save(os.path.join(tmp, "v.npy"), verts)
np.save(os.path.join(tmp, "t.npy"), np.ascontiguousarray(tris.T))
R = ref_renderer.Renderer(os.path.join(tmp, "v.npy"), os.path.join(tmp, "t.npy"))
rng = np.random.RandomState(7)
h, w = 168, 200
frame = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
K = np.array([[w + h, 0, w // 2], [0, w + h, h // 2], [0, 0, 1]], dtype=np.float64)
eulers = [(12.0, -35.0, 8.0), (-20.0, 60.0, -15.0)]
boxes = [(30, 20, 120, 130), (90, 40, 190, 150)]
rot = np.array([[0.8, -0.1, 0.59], [0.2, 0.97, -0.1], [-0.56, 0.2, 0.8]])
rot, _ = np.linalg.qr(rot)
poses = [ref_vis.convert_euler_bbox_to_6dof(list(e), b, K) for e, b in zip(eulers, boxes)]
pose_rm = ref_vis.convert_rotmat_bbox_to_6dof(rot, boxes[0], K)
tv = R.transform_vertices(frame, [np.array(p) for p in poses])
overlap = frame.copy()
stages = []
for v in tv:
    v32 = ref_renderer._to_ctype(v.astype(np.float32))
    overlap = R.render_app(v32, R.triangles, overlap)
    stages.append(overlap.copy())
normal0 = RefSim3DR.get_normal(np.ascontiguousarray(tv[0], dtype=np.float32), R.triangles)
# the rasteriser alone at alpha < 1 and with reversed rows (the Cython default is alpha = 1): straight calls of the reference C++
cols = rng.rand(verts.shape[0], 3).astype(np.float32)
img_a = frame.copy()
_rasterize(img_a, np.ascontiguousarray(tv[0], dtype=np.float32), R.triangles, cols, np.zeros((h, w), np.float32) - 1e8, tris.shape[0], h, w, 3, 0.6, True)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sim3dr.npz"), frame=frame, intrinsics=K, eulers=np.array(eulers), boxes=np.array(boxes), rot=rot,
                    poses=np.array(poses), pose_rotmat=np.array(pose_rm), tv0=tv[0], tv1=tv[1], stage0=stages[0], stage1=stages[1], normal0=normal0,
                    colors=cols, img_alpha06_reversed=img_a)
print("wrote tests/golden/sim3dr.npz:", {k: v.shape for k, v in np.load(os.path.join(ROOT, "tests", "golden", "sim3dr.npz")).items()})
print("pixels painted:", int((stages[1] != frame).any(-1).sum()), "of", h * w)
