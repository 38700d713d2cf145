import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box)")


@pytest.fixture(scope="session")
def golden_util():
    return np.load(os.path.join(GOLDEN, "util_tables.npz"))


@pytest.fixture(scope="session")
def golden_pcg():
    return np.load(os.path.join(GOLDEN, "pcg_cases.npz"))


@pytest.fixture(scope="session")
def golden_stage():
    return np.load(os.path.join(GOLDEN, "stage_n16.npz"))


def fresh_state(sc, oracle, vel=None):
    """Inputs of a first substep: no stale face weights, no valid flags."""
    z32 = lambda: np.zeros(oracle.shape, np.float32)  # noqa: E731
    return dict(pos=sc.pos, vel=sc.vel if vel is None else vel, mass=sc.mass, phi_fluid=sc.phi_fluid,
                phi_solid=sc.phi_solid, w=[z32() for _ in range(3)],
                valid=[np.zeros(oracle.shape, np.uint8) for _ in range(3)])


def desc_from_golden(g):
    d, f = g["desc"], g["desc_f"]
    return dict(nx=int(d[0]), ny=int(d[1]), nz=int(d[2]), ox=int(d[3]), oy=int(d[4]), oz=int(d[5]), dx=float(f[0]),
                phi_fluid_bg=float(f[1]), phi_solid_bg=float(f[2]), rest_density=float(f[3]))
