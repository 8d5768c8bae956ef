import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DATA = os.path.join(ROOT, "tests", "golden", "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def data_dir():
    return DATA


@pytest.fixture(scope="session")
def load_scene():
    from raytracing_amd import host

    cache = {}

    def _load(name):
        if name not in cache:
            cache[name] = host.Scene.load(os.path.join(DATA, name))
        return cache[name]

    return _load


def synthetic_vertex_sha256(scene):
    """SHA-256 of the flattened vertex buffer (tri_verts, 9 f64 per triangle) of a scene."""
    import hashlib

    import numpy as np
    d = scene.desc
    buf = np.ctypeslib.as_array(d.tri_verts, shape=(int(d.n_tris) * 9,))
    return hashlib.sha256(memoryview(buf)).hexdigest()


@pytest.fixture(scope="session")
def synthetic_sha():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "synthetic_sha256.json")))


@pytest.fixture(scope="session", autouse=True)
def _pipeline_from_env():
    """QD_PIPELINE=0|1 runs the GPU suite on one scheduling form of the frame (qd_set_pipeline); unset = library default."""
    v = os.environ.get("QD_PIPELINE")
    if v is not None:
        from raytracing_amd import device
        device.set_pipeline(int(v))
    yield
