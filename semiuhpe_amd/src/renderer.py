"""The demo renderer with the reference's interface (src/renderer.py:37-131): a head mesh is posed per detection
(`transform_vertices`), lit and rasterised over the frame on the GPU (`semiuhpe_amd.Sim3DR`), blended back with the frame
(`render`), or dumped as a coloured OBJ (`save_to_obj`).  cv2 is not needed: `cv2.addWeighted` on uint8 frames is restated
(saturate(round-half-even(a * wa + b * wb))), frames are numpy (h, w, 3) uint8 arrays."""
import numpy as np
from scipy.spatial.transform import Rotation

from ..Sim3DR import RenderPipeline


def transform_points(points, pose):
    return points.dot(Rotation.from_rotvec(pose[:3]).as_matrix().T) + pose[3:]


def plot_3d_landmark(verts, campose, intrinsics):
    """-> (pixel coordinates (n, 2), homogeneous projection (n, 3)) of the posed vertices (renderer.py:9-19)"""
    projected = intrinsics.dot(transform_points(verts, campose).T).T
    return projected[:, :2] / projected[:, 2:3], projected


def add_weighted(a, wa, b, wb):
    return np.clip(np.rint(a.astype(np.float64) * wa + b.astype(np.float64) * wb), 0, 255).astype(np.uint8)


def get_colors(img, ver):
    """nearest-pixel colours of (3, n) vertices, clamped to the frame (renderer.py:28-35; clamps `ver` in place)"""
    h, w, _ = img.shape
    ver[0, :] = np.clip(ver[0, :], 0, w - 1)
    ver[1, :] = np.clip(ver[1, :], 0, h - 1)
    ind = np.round(ver).astype(np.int32)
    return (img[ind[1, :], ind[0, :], :] / 255.0).copy()


class Renderer:
    LIGHT = dict(intensity_ambient=0.3, color_ambient=(1, 1, 1), intensity_directional=0.6, color_directional=(1, 1, 1),
                 intensity_specular=0.1, specular_exp=5, light_pos=(0, 0, 5), view_pos=(0, 0, 5))

    def __init__(self, vertices_path="../pose_references/vertices_trans.npy", triangles_path="../pose_references/triangles.npy",
                 vertices=None, triangles=None):
        """the mesh comes from the reference's .npy pair (vertices (n, 3), triangles (3, m)) or directly as arrays"""
        self.vertices = np.load(vertices_path) if vertices is None else np.array(vertices)
        tri = np.load(triangles_path) if triangles is None else np.asarray(triangles)
        self.triangles = np.ascontiguousarray(tri.T if tri.shape[0] == 3 and tri.shape[1] != 3 else tri, dtype=np.int32)
        self.vertices[:, 0] *= -1
        self.cfg = dict(self.LIGHT)
        self.render_app = RenderPipeline(**self.cfg)

    def transform_vertices(self, img, poses, global_intrinsics=None):
        h, w = img.shape[:2]
        if global_intrinsics is None:
            global_intrinsics = np.array([[w + h, 0, w // 2], [0, w + h, h // 2], [0, 0, 1]])
        out = []
        for pose in poses:
            pose = np.asarray(pose, dtype=np.float64)
            lms = np.zeros_like(self.vertices)
            lms[:, :2], proj = plot_3d_landmark(self.vertices, pose, global_intrinsics)
            lms[:, 2] = proj[:, 2] * -1
            spread = (np.max(lms[:, 0]) - np.min(lms[:, 0])) + (np.max(lms[:, 1]) - np.min(lms[:, 1]))
            lms[:, 2] *= (h + w) / pose[5]
            lms[:, 2] += spread * 3
            out.append(lms)
        return out

    def render(self, img, transformed_vertices, alpha=0.9, save_path=None):
        img = np.asarray(img)
        overlap = img.copy()
        for vertices in transformed_vertices:
            overlap = self.render_app(np.ascontiguousarray(vertices, dtype=np.float32), self.triangles, overlap)
        res = add_weighted(img, 1 - alpha, overlap, alpha)
        if save_path is not None:
            from PIL import Image
            Image.fromarray(res[:, :, ::-1]).save(save_path)     # frames are BGR (cv2 convention of the callers)
            print(f"Save visualization result to {save_path}")
        return res

    def save_to_obj(self, img, ver_lst, height, save_path):
        if len(ver_lst) <= 0:
            return
        n_vertex = ver_lst[0].shape[0]
        with open(save_path, "w") as f:
            for ver in ver_lst:
                ver = ver.T
                colors = get_colors(img, ver)
                for j in range(n_vertex):
                    x, y, z = ver[:, j]
                    f.write(f"v {x:.2f} {height - y:.2f} {z:.2f} {colors[j, 2]:.2f} {colors[j, 1]:.2f} {colors[j, 0]:.2f}\n")
            for i in range(len(ver_lst)):
                off = i * n_vertex
                for a, b, c in self.triangles:
                    f.write(f"f {c + 1 + off} {b + 1 + off} {a + 1 + off}\n")
        print(f"Dump tp {save_path}")
