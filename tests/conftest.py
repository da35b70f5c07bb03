import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_scan(name):
    """Vertex positions of one of the reference's scans (tests/golden/*.xyz.f32, see tools/make_fixtures.py)."""
    return np.fromfile(os.path.join(GOLDEN, name + ".xyz.f32"), dtype="<f4").reshape(-1, 3).astype(np.float32)


@pytest.fixture(scope="session")
def bunny000():
    return load_scan("bunny000")


@pytest.fixture(scope="session")
def bunny045():
    return load_scan("bunny045")


@pytest.fixture(scope="session")
def bun270():
    return load_scan("bun270")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    """One icpx context on cuda:0 for the whole GPU session. Fails (does not skip) when the HIP library
    or the device is missing: the GPU tests must never pass on a fallback."""
    from icp_amd.api import Context
    ctx = Context(0)
    yield ctx
    ctx.close()


def synth_surface(n, seed, noise=0.0):
    """Points on the 'bumpy sphere' r = 0.35 (1 + 0.15 sin 3theta cos 5phi), float32."""
    rng = np.random.default_rng(seed)
    u = rng.uniform(-1.0, 1.0, n)
    phi = rng.uniform(0.0, 2 * np.pi, n)
    theta = np.arccos(u)
    r = 0.35 * (1 + 0.15 * np.sin(3 * theta) * np.cos(5 * phi))
    p = np.stack([r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(theta)], 1)
    if noise:
        p = p + rng.normal(0.0, noise, p.shape)
    return p.astype(np.float32)


def rigid(axis, deg, t):
    a = np.asarray(axis, float); a /= np.linalg.norm(a)
    th = np.deg2rad(deg)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    return T
