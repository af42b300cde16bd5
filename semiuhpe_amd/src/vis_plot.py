"""Pose conversions of the demo (src/vis_plot.py:7-104): a head pose predicted inside a face crop (Euler angles or a rotation matrix)
becomes the 6-DoF pose of the head mesh in the full frame, which `Renderer.transform_vertices` consumes.  `draw_axis_ypr`
(:106-135) draws the three pose axes with a small numpy line rasteriser (cv2 is not in this image: 2-pixel-wide Bresenham lines,
same end points and colours, not cv2's anti-aliasing)."""
import numpy as np
from scipy.spatial.transform import Rotation


def bbox_is_dict(bbox):
    if isinstance(bbox, dict):
        return bbox
    return {"left": bbox[0], "top": bbox[1], "right": bbox[2], "bottom": bbox[3]}


def get_bbox_intrinsics(image_intrinsics, bbox):
    """the frame's camera with its principal point moved to the crop centre (:19-29)"""
    k = image_intrinsics.copy()
    k[0, 2] = bbox["left"] + ((bbox["right"] - bbox["left"]) // 2)
    k[1, 2] = bbox["top"] + ((bbox["bottom"] - bbox["top"]) // 2)
    return k


def pose_bbox_to_full_image(pose, image_intrinsics, bbox):
    """(:31-70) crop-local 6-DoF pose -> frame pose: depth rescaled by focal length / half the crop's mean side, translation and
    rotation re-projected from the crop camera to the frame camera"""
    bbox = bbox_is_dict(bbox)
    rvec, tvec = pose[:3].copy(), pose[3:].copy()
    crop_k = get_bbox_intrinsics(image_intrinsics, bbox)
    bbox_size = ((bbox["right"] - bbox["left"]) + (bbox["bottom"] - bbox["top"])) * (0.5 * 0.5)
    tvec[2] *= image_intrinsics[0, 0] / bbox_size
    tvec = crop_k.dot(tvec.T).dot(np.linalg.inv(image_intrinsics.T))
    rmat = np.linalg.inv(image_intrinsics).dot(crop_k.dot(Rotation.from_rotvec(rvec).as_matrix()))
    return np.concatenate([Rotation.from_matrix(rmat).as_rotvec(), tvec])


def convert_euler_bbox_to_6dof(euler_angle, bbox, global_intrinsics):
    """(:74-87) [pitch, yaw, roll] in degrees, as the 300WLP / AFLW2000 / BIWI labels and this repo's Euler kernel give them"""
    pitch, yaw, roll = euler_angle
    rot = Rotation.from_euler("xyz", [pitch, -yaw, -roll], degrees=True).as_matrix().T
    rv = Rotation.from_matrix(rot).as_rotvec()
    local = np.array([rv[0], rv[1], rv[2], 0, 0, 1])
    return pose_bbox_to_full_image(local, global_intrinsics, bbox_is_dict(bbox)).tolist()


def convert_rotmat_bbox_to_6dof(rot_mat, bbox, global_intrinsics):
    """(:90-103) DAD3DHeads-convention rotation matrices: turned by 180 degrees about x, transposed, first rotation-vector entry negated"""
    rot = (np.diag([1.0, -1.0, -1.0]) @ rot_mat).T
    rv = Rotation.from_matrix(rot).as_rotvec()
    local = np.array([-rv[0], rv[1], rv[2], 0, 0, 1])
    return pose_bbox_to_full_image(local, global_intrinsics, bbox_is_dict(bbox)).tolist()


def _line(img, p, q, color, width=2):
    x0, y0, x1, y1 = int(p[0]), int(p[1]), int(q[0]), int(q[1])
    n = max(abs(x1 - x0), abs(y1 - y0), 1)
    xs = np.rint(np.linspace(x0, x1, n + 1)).astype(int)
    ys = np.rint(np.linspace(y0, y1, n + 1)).astype(int)
    for dx in range(width):
        for dy in range(width):
            ok = (xs + dx >= 0) & (xs + dx < img.shape[1]) & (ys + dy >= 0) & (ys + dy < img.shape[0])
            img[ys[ok] + dy, xs[ok] + dx] = color


def draw_axis_ypr(img, yaw, pitch, roll, tdx=None, tdy=None, size=100):
    """(:106-135) x axis red, y axis green, z axis (out of the screen) blue, BGR frame, drawn in place"""
    pitch, yaw, roll = pitch * np.pi / 180, -(yaw * np.pi / 180), roll * np.pi / 180
    if tdx is None or tdy is None:
        tdx, tdy = img.shape[1] / 2, img.shape[0] / 2
    ends = [(size * (np.cos(yaw) * np.cos(roll)) + tdx, size * (np.cos(pitch) * np.sin(roll) + np.cos(roll) * np.sin(pitch) * np.sin(yaw)) + tdy, (0, 0, 255)),
            (size * (-np.cos(yaw) * np.sin(roll)) + tdx, size * (np.cos(pitch) * np.cos(roll) - np.sin(pitch) * np.sin(yaw) * np.sin(roll)) + tdy, (0, 255, 0)),
            (size * np.sin(yaw) + tdx, size * (-np.cos(yaw) * np.sin(pitch)) + tdy, (255, 0, 0))]
    for x, y, color in ends:
        _line(img, (tdx, tdy), (x, y), color)
    return img
