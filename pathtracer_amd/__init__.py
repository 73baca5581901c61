"""pathtracer_amd — MI355X (gfx950) backend for the Pathtracer `--gpu` path.

Public surface mirrors the reference's boundary: `Scene.load("scene.json")`
(src/scene.rs:94) and `hip_renderer.render(scene)` (src/gpu_renderer.rs:11).
"""
from . import hip_renderer  # noqa: F401
from ._abi import HipptError  # noqa: F401
from .scene import Scene  # noqa: F401
