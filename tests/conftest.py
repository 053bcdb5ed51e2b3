"""pytest configuration: markers + import paths.

* ``gpu`` marks tests that need a real MI355X (run by the driver with ``-m gpu``).
* The product package lives in ``neural-astar_amd/`` (import name ``neural_astar``, the reference's own
  package name, so it is a drop-in); the oracle lives in ``oracle/`` and is imported by tests only.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "neural-astar_amd"), ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (run with -m gpu)")
