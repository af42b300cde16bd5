"""oracle/sim3dr.py -- TEST INFRASTRUCTURE ONLY: the CPU side of the f4 row (demo renderer), SURVEY.md 8.

* `get_normal` / `rasterize`: ctypes over oracle/libsim3dr_oracle.so (our plain-C restatement, oracle/sim3dr_oracle.c) or, with
  `ref=True`, over oracle/_ref/libsim3dr_ref.so = the reference's own Sim3DR/lib/rasterize_kernel.cpp compiled by oracle/Makefile
  (its C++ entry points by mangled name; the Cython wrapper Sim3DR/lib/rasterize.pyx:63-102 only forwards the array pointers).
* `RenderPipeline`: numpy restatement of Sim3DR/lighting.py:24-72 (ambient + diffuse + specular per-vertex light, then rasterize).
* `transform_vertices` / `render` / `add_weighted`: src/renderer.py:58-102 (`cv2.addWeighted` restated: saturate(round-half-even)).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


def _lib(ref):
    if ref not in _libs:
        path = os.path.join(_HERE, "_ref", "libsim3dr_ref.so") if ref else os.path.join(_HERE, "libsim3dr_oracle.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: build it with `make -C oracle` (__graft_entry__.build() does)")
        _libs[ref] = ctypes.CDLL(path)
    return _libs[ref]


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libsim3dr_ref.so"))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def get_normal(vertices, triangles, ref=False):
    """Sim3DR/Sim3DR.py:8-11"""
    vertices = np.ascontiguousarray(vertices, dtype=np.float32)
    triangles = np.ascontiguousarray(triangles, dtype=np.int32)
    normal = np.zeros_like(vertices, dtype=np.float32)
    fn = _lib(ref)._Z11_get_normalPfS_Piii if ref else _lib(ref).sim3dr_get_normal
    fn.restype = None
    fn(_p(normal), _p(vertices), _p(triangles), ctypes.c_int(vertices.shape[0]), ctypes.c_int(triangles.shape[0]))
    return normal


def rasterize(vertices, triangles, colors, bg=None, height=None, width=None, channel=None, reverse=False, alpha=1.0, ref=False):
    """Sim3DR/Sim3DR.py:14-29 (alpha: rasterize.pyx:95 default 1); paints into `bg` in place and returns it"""
    if bg is not None:
        height, width, channel = bg.shape
    else:
        bg = np.zeros((height, width, channel), dtype=np.uint8)
    assert bg.dtype == np.uint8 and bg.flags.c_contiguous
    buffer = np.zeros((height, width), dtype=np.float32) - 1e8
    vertices = np.ascontiguousarray(vertices, dtype=np.float32)
    triangles = np.ascontiguousarray(triangles, dtype=np.int32)
    colors = np.ascontiguousarray(colors, dtype=np.float32)
    if ref:
        fn = _lib(True)._Z10_rasterizePhPfPiS0_S0_iiiifb
        fn.restype = None
        fn(_p(bg), _p(vertices), _p(triangles), _p(colors), _p(buffer), ctypes.c_int(triangles.shape[0]), ctypes.c_int(height), ctypes.c_int(width),
           ctypes.c_int(channel), ctypes.c_float(alpha), ctypes.c_bool(reverse))
    else:
        fn = _lib(False).sim3dr_rasterize
        fn.restype = None
        fn(_p(bg), _p(vertices), _p(triangles), _p(colors), _p(buffer), ctypes.c_int(triangles.shape[0]), ctypes.c_int(height), ctypes.c_int(width),
           ctypes.c_int(channel), ctypes.c_float(alpha), ctypes.c_int(int(reverse)))
    return bg


_norm = lambda arr: arr / np.sqrt(np.sum(arr ** 2, axis=1))[:, None]


def norm_vertices(vertices):
    """Sim3DR/lighting.py:9-14"""
    vertices -= vertices.min(0)[None, :]
    vertices /= vertices.max()
    vertices *= 2
    vertices -= vertices.max(0)[None, :] / 2
    return vertices


def _row(obj):
    return np.array(obj, dtype=np.float32)[None, :] if isinstance(obj, (tuple, list)) else obj


class RenderPipeline:
    """Sim3DR/lighting.py:24-72 with the configuration of src/renderer.py:47-56 as defaults"""

    def __init__(self, intensity_ambient=0.3, intensity_directional=0.6, intensity_specular=0.1, specular_exp=5, color_ambient=(1, 1, 1),
                 color_directional=(1, 1, 1), light_pos=(0, 0, 5), view_pos=(0, 0, 5), ref=False):
        self.ia, self.idr, self.isp, self.exp = intensity_ambient, intensity_directional, intensity_specular, specular_exp
        self.ca, self.cd, self.lp, self.vp = _row(color_ambient), _row(color_directional), _row(light_pos), _row(view_pos)
        self.ref = ref

    def light(self, vertices, triangles):
        normal = get_normal(vertices, triangles, ref=self.ref)
        light = np.zeros_like(vertices, dtype=np.float32)
        if self.ia > 0:
            light += self.ia * self.ca
        vn = norm_vertices(vertices.copy())
        if self.idr > 0:
            direction = _norm(self.lp - vn)
            cos = np.sum(normal * direction, axis=1)[:, None]
            light += self.idr * (self.cd * np.clip(cos, 0, 1))
            if self.isp > 0:
                v2v = _norm(self.vp - vn)
                reflection = 2 * cos * normal - direction
                spe = np.sum((v2v * reflection) ** self.exp, axis=1)[:, None]
                spe = np.where(cos != 0, np.clip(spe, 0, 1), np.zeros_like(spe))
                light += self.isp * self.cd * np.clip(spe, 0, 1)
        return np.clip(light, 0, 1)

    def __call__(self, vertices, triangles, bg, texture=None):
        light = self.light(vertices, triangles)
        if texture is not None:
            texture *= light
            light = texture
        return rasterize(vertices, triangles, light, bg=bg, ref=self.ref)


def transform_vertices(mesh_vertices, img_shape, poses, global_intrinsics=None):
    """src/renderer.py:58-85 (+ plot_3d_landmark :9-19, transform_points :6-7): mesh -> image-plane vertices (x, y, scaled depth) per pose"""
    from scipy.spatial.transform import Rotation
    h, w = img_shape[:2]
    if global_intrinsics is None:
        global_intrinsics = np.array([[w + h, 0, w // 2], [0, w + h, h // 2], [0, 0, 1]])
    out = []
    for pose in poses:
        pose = np.asarray(pose, dtype=np.float64)
        trans = mesh_vertices.dot(Rotation.from_rotvec(pose[:3]).as_matrix().T) + pose[3:]
        proj = global_intrinsics.dot(trans.T).T
        lms = np.zeros_like(mesh_vertices)
        lms[:, :2] = proj[:, :2] / np.tile(proj[:, 2], (2, 1)).T
        lms[:, 2] = proj[:, 2] * -1
        range_x = np.max(lms[:, 0]) - np.min(lms[:, 0])
        range_y = np.max(lms[:, 1]) - np.min(lms[:, 1])
        lms[:, 2] *= (h + w) / pose[5]
        lms[:, 2] += (range_x + range_y) * 3
        out.append(lms)
    return out


def add_weighted(a, wa, b, wb):
    """cv2.addWeighted(a, wa, b, wb, 0) on uint8 images (src/renderer.py:95): saturate_cast<uchar>(a*wa + b*wb), rounded half to even"""
    return np.clip(np.rint(a.astype(np.float64) * wa + b.astype(np.float64) * wb), 0, 255).astype(np.uint8)


def render(pipeline, triangles, img, transformed_vertices, alpha=0.9):
    """src/renderer.py:87-102"""
    img = np.asarray(img)
    overlap = img.copy()
    for vertices in transformed_vertices:
        overlap = pipeline(np.ascontiguousarray(vertices, dtype=np.float32), triangles, overlap)
    return add_weighted(img, 1 - alpha, overlap, alpha)
