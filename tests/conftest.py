import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def uvs():
    return importlib.import_module("uv-slam_amd")


@pytest.fixture(scope="session")
def oracle(uvs):
    from oracle_binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def gpu_api(uvs):
    """The product binding; only usable on a GPU box (uvs_create fails loudly otherwise)."""
    return uvs.api
