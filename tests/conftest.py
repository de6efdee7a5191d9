"""pytest configuration: markers and import paths.

`gpu` marks tests that need a real MI355X (run by the driver with `-m gpu`);
everything else must pass on a CPU-only box.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "difffe-physics-lab_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (HIP kernels are executed)")
