"""`RenderPipeline` with the reference's call surface (Sim3DR/lighting.py:24-72: keyword configuration, `update_light_pos`,
`pipeline(vertices, triangles, bg, texture=None)`): Phong-style light per vertex -- a handful of numpy expressions over the
mesh's vertices, host side as in the reference and in its float32 operation order, so the colours that reach the rasteriser
are the reference's bit for bit -- then normals and rasterisation on the device (.Sim3DR -> csrc/raster.hip)."""
import numpy as np

from .Sim3DR import get_normal, rasterize

_DEFAULTS = dict(intensity_ambient=0.3, intensity_directional=0.6, intensity_specular=0.1, specular_exp=5,
                 color_ambient=(1, 1, 1), color_directional=(1, 1, 1), light_pos=(0, 0, 5), view_pos=(0, 0, 5))


def _as_row(value):
    """tuples / lists become (1, k) float32 rows that broadcast over the vertices; scalars stay Python numbers (lighting.py:17-20)"""
    return np.asarray(value, dtype=np.float32)[None, :] if isinstance(value, (tuple, list)) else value


def _unit_rows(a):
    return a / np.sqrt(np.sum(a ** 2, axis=1))[:, None]


def norm_vertices(vertices):
    """in place: shift to the origin, scale the largest coordinate to 2, centre (lighting.py:9-14)"""
    vertices -= vertices.min(0)[None, :]
    vertices /= vertices.max()
    vertices *= 2
    vertices -= vertices.max(0)[None, :] / 2
    return vertices


class RenderPipeline:
    def __init__(self, **kwargs):
        unknown = set(kwargs) - set(_DEFAULTS)
        if unknown:
            raise TypeError(f"RenderPipeline: unknown option(s) {sorted(unknown)}")
        for name, default in _DEFAULTS.items():
            setattr(self, name, _as_row(kwargs.get(name, default)))

    def update_light_pos(self, light_pos):
        self.light_pos = _as_row(light_pos)

    def vertex_light(self, vertices, normal):
        """(nver, 3) colours in [0, 1]: ambient + diffuse (clipped cosine to the light) + specular ((view . reflection)^exp per
        channel, summed; zero where the cosine is exactly zero), lighting.py:40-63"""
        light = np.zeros_like(vertices, dtype=np.float32)
        if self.intensity_ambient > 0:
            light += self.intensity_ambient * self.color_ambient
        if self.intensity_directional > 0:
            unit = norm_vertices(vertices.copy())
            to_light = _unit_rows(self.light_pos - unit)
            cos = np.sum(normal * to_light, axis=1)[:, None]
            light += self.intensity_directional * (self.color_directional * np.clip(cos, 0, 1))
            if self.intensity_specular > 0:
                to_view = _unit_rows(self.view_pos - unit)
                mirrored = 2 * cos * normal - to_light
                shine = np.sum((to_view * mirrored) ** self.specular_exp, axis=1)[:, None]
                shine = np.where(cos != 0, np.clip(shine, 0, 1), np.zeros_like(shine))
                light += self.intensity_specular * self.color_directional * np.clip(shine, 0, 1)
        return np.clip(light, 0, 1)

    def __call__(self, vertices, triangles, bg, texture=None):
        light = self.vertex_light(vertices, get_normal(vertices, triangles))
        if texture is not None:
            texture *= light                                    # (in place, as the reference)
            light = texture
        return rasterize(vertices, triangles, light, bg=bg)
