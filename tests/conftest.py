import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


_scene_cache = {}


def get_scene(name, lazy):
    from raytracer_amd.scene import FlatScene
    key = (name, lazy)
    if key not in _scene_cache:
        _scene_cache[key] = FlatScene.builtin(name, lazy=lazy)
    return _scene_cache[key]


@pytest.fixture(scope="session")
def cpu_scene():
    """flattened built-in scene without touching the GPU library (lazy binding of the host library)"""
    return lambda name: get_scene(name, True)


_oracle_cache = {}


@pytest.fixture(scope="session")
def oracle_for():
    from oracle.binding import Oracle

    def make(name, lazy=True):
        if name not in _oracle_cache:
            _oracle_cache[name] = Oracle(get_scene(name, lazy))
        return _oracle_cache[name]
    return make


def image_metrics(rgb8, ref8, invalid=None):
    """acceptance metrics of BASELINE.json: fraction of pixels whose 8-bit RGB differs by <= 1, and PSNR (dB)"""
    a, b = rgb8.astype(np.int32), ref8.astype(np.int32)
    ok = np.abs(a - b).max(axis=2) <= 1
    valid = np.ones(ok.shape, bool) if invalid is None else ~invalid
    frac = float((ok & valid).sum() / ok.size)  # invalid reference pixels count as mismatches
    mse = float(((a - b)[valid] ** 2).mean()) if valid.any() else 0.0
    psnr = float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
    return frac, psnr


def render_params(g):
    return json.loads(str(g["params"]))
