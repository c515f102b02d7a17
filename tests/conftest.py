import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# One process, one HIP runtime: some GPU tests build their clouds with torch on the device, and PyTorch-ROCm ships its own
# libamdhip64 / libhsa-runtime64 under the SONAMEs libsmhip.so links against.  Whichever is loaded first is the one the
# process uses; with libsmhip.so first, torch's later initialisation finds "No HIP GPUs are available" -- so torch goes first.
try:
    import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cfg1():
    """BASELINE config #1 (plumbing): two 5k-pt clouds on three noisy planes, known SE(3) offset."""
    from staticmapping_amd import synth
    from oracle import cref
    tgt, src, T = synth.three_planes_pair(5000, seed=1)
    q, n, _ = cref.calculate_normals(tgt[:, :3].astype(np.float64))
    return dict(src=src, tgt=tgt, q=q, n=n, T=T)


def _velodyne_case(n_points):
    from staticmapping_amd import synth
    from oracle import cref
    a, b, T = synth.scan_pair("cfg2", n_points=n_points)
    q, n, _ = cref.calculate_normals(a[:, :3].astype(np.float64))
    ok = np.isfinite(n).all(axis=1)
    return dict(src=b, tgt=a, q=q[ok], n=n[ok], T=T, guess=synth.make_pose(t=(0.6, 0.0, 0.0)))


@pytest.fixture(scope="session")
def velo20k():
    return _velodyne_case(20_000)


@pytest.fixture(scope="session")
def cfg2():
    """BASELINE config #2: 120k-pt Velodyne-64 scan pair (seeded synthetic)."""
    return _velodyne_case(120_000)
