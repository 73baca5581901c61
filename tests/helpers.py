"""Seeded input generators shared by the parity tests."""
import numpy as np


def unit_vectors(rng, n):
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v.astype(np.float32)


def example_rays(rng, n):
    """Rays inside the scene.example room: origins in the camera/corridor volume, random directions."""
    ro = np.stack([rng.uniform(-2.4, 2.4, n), rng.uniform(-0.7, 3.0, n), rng.uniform(-1.0, 7.0, n)], 1)
    return ro.astype(np.float32), unit_vectors(rng, n)


def rel_err_vec(a, b):
    """|a-b| relative to max(1, |b|) per row (unit vectors and radiances)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.linalg.norm(a - b, axis=-1) if a.ndim > 1 else np.abs(a - b)
    s = np.linalg.norm(b, axis=-1) if b.ndim > 1 else np.abs(b)
    return d / np.maximum(s, 1.0)


def rel_err(a, b, floor=1e-30):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)
