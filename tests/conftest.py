import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Parity bound from BASELINE.json north_star: <= 1e-5 relative fp32
# (max-abs error / max-abs reference, and rel-L2, per transform: SURVEY.md 8(c)).
REL_TOL = 1e-5
# Absolute bound the reference's own tests use (examples/basic_inverse.rs:250).
REF_ABS_TOL = 1e-5


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")
    # built artefacts are git-ignored: (re)build them with the dependency-aware Makefile, as
    # __graft_entry__.build() does; without hipcc the ABI/GPU tests fail with the loader's clear message
    import shutil
    import subprocess
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "fft_wgpu_amd", "csrc"), "-j8", "all", "lab"])


@pytest.fixture(scope="session")
def k4():
    return np.load(os.path.join(GOLDEN, "k4_random.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def known_answers():
    with open(os.path.join(GOLDEN, "reference_known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    import oracle as _o
    _o.build()
    return _o
