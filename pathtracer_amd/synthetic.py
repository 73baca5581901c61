"""Synthetic many-sphere GGX scene of BASELINE config C4.

synthetic_spheres(n=100000, seed=0x5EED): splitmix64-driven; sphere centres uniform in
[-40,40] x [0.1,12] x [4,84], radii uniform in [0.05,0.35]; 64 materials (metallic in {0,1}
50/50, roughness log-uniform in [0.05,1], rgb uniform in [0.2,0.9], ior 1, no volume); one
ground plane point=(0,0,44), u=(60,0,0), v=(0,0,-60); one light pos=(0,30,44), u=(-20,0,0),
v=(0,0,-20) (normal -> -y), intensity 40; camera pos=(0,6,-10), look_at=(0,3,20), fov 60,
aperture 0.02. Draw order: per material {metallic, roughness, r, g, b}, then per sphere
{cx, cy, cz, radius, material}. Scene order: the plane first, then the spheres.
"""
import numpy as np

from .scene import MAT_DTYPE, SPHERE_DTYPE, Scene


def splitmix64_uniforms(seed, count):
    """count doubles in [0,1) from splitmix64(seed)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def synthetic_spheres(n=100000, seed=0x5EED, width=1920, height=1080, samples=256, n_materials=64):
    u = splitmix64_uniforms(seed, 5 * n_materials + 5 * n)
    um = u[:5 * n_materials].reshape(n_materials, 5)
    us = u[5 * n_materials:].reshape(n, 5)
    s = Scene()
    s.width, s.height, s.samples = width, height, samples
    mats = np.zeros(n_materials + 2, MAT_DTYPE)
    mats["metallic"][:n_materials] = (um[:, 0] < 0.5).astype(np.float32)
    mats["roughness"][:n_materials] = np.exp(np.log(0.05) + um[:, 1] * (np.log(1.0) - np.log(0.05)))
    mats["rgb"][:n_materials] = 0.2 + 0.7 * um[:, 2:5]
    mats["ior"] = 1.0
    ground = n_materials
    mats["rgb"][ground], mats["roughness"][ground] = (0.6, 0.6, 0.6), 1.0
    mats["rgb"][ground + 1], mats["roughness"][ground + 1] = (1.0, 0.0, 1.0), 1.0   # unused spare (magenta default)
    s.materials = mats
    s.add_plane((0, 0, 44), (60, 0, 0), (0, 0, -60), ground)            # scene order 0
    sp = np.zeros(n, SPHERE_DTYPE)
    sp["center"][:, 0] = -40 + 80 * us[:, 0]
    sp["center"][:, 1] = 0.1 + 11.9 * us[:, 1]
    sp["center"][:, 2] = 4 + 80 * us[:, 2]
    sp["radius"] = 0.05 + 0.30 * us[:, 3]
    sp["material"] = np.minimum((us[:, 4] * n_materials).astype(np.uint32), n_materials - 1)
    sp["order"] = 1 + np.arange(n, dtype=np.uint32)
    s.spheres = sp
    s.add_light((0, 30, 44), (-20, 0, 0), (0, 0, -20), (40, 40, 40))
    s.set_camera((0, 6, -10), (0, 3, 20), (0, 1, 0), 60.0, 0.02)
    return s
