"""pytest config: register the ``gpu`` marker, put the repo root (oracle/) and the product package dir
(mri-diffusion-superresolution_amd/ -> ``mrisr``) on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mri-diffusion-superresolution_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# the tile / split-K table shipped with the library (read-only; see bench.py): GPU tests run the kernels the bench runs
os.environ.setdefault("MRISR_TUNE_CACHE", os.path.join(ROOT, "profiles", "r03_tune_cache.tsv"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
