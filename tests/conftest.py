import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def example_scene():
    from pathtracer_amd import Scene
    return Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"))


@pytest.fixture(scope="session")
def hip():
    """HIP unit kernels through the C ABI; fails loudly without library or GPU."""
    from pathtracer_amd import hip_renderer
    return hip_renderer.unit_ops(0)
