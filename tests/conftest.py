import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

REFERENCE = "/root/reference"  # exists only in the build container, never on the GPU box


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """libmi_pt.so and the oracle must exist; build them if the tree is fresh."""
    from master_amd import build as mb

    mb.build(force=False)
    import oracle

    oracle.build()


def scene_path(name):
    return os.path.join(ROOT, "scenes", name + ".miscene")


@pytest.fixture(scope="session")
def cornell():
    import master_amd as ma

    return ma.Scene.load(scene_path("CornellBoxDiffuse"))


def load_scene(name):
    import master_amd as ma

    return ma.Scene.load(scene_path(name))
