"""Sim3DR on the MI355X: the reference's package surface (Sim3DR/__init__.py: `from .Sim3DR import get_normal, rasterize`,
`from .lighting import RenderPipeline`) over the HIP kernels of csrc/raster.hip."""
from .Sim3DR import get_normal, rasterize      # noqa: F401
from .lighting import RenderPipeline           # noqa: F401
