import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pylidar-slam_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / at round end)")


@pytest.fixture(scope="session")
def golden_components():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "components.npz"))


@pytest.fixture(scope="session")
def golden_c1():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "c1_sequence.npz"))


@pytest.fixture(scope="session")
def c1_scans(golden_c1):
    """The C1 synthetic scans, regenerated from the seeded generator and checked against the hashes stored with
    the reference outputs (the fixtures hold the reference's OUTPUTS; inputs are reproducible from the seed)."""
    import hashlib
    import numpy as np
    from pylidar_slam_amd.synthetic import SceneConfig, make_sequence
    h, w = (int(v) for v in golden_c1["hw"])
    scans, gt = make_sequence(SceneConfig(height=h, width=w), len(golden_c1["scan_sha"]))
    for s, ref in zip(scans, golden_c1["scan_sha"]):
        if hashlib.sha1(np.ascontiguousarray(s).tobytes()).hexdigest() != str(ref):
            pytest.fail("the seeded synthetic generator no longer reproduces the scans the reference was run on "
                        "(tests/golden/c1_sequence.npz holds their sha1): the only reference-pinned end-to-end "
                        "sequence would silently drop out — regenerate the goldens or fix the generator")
    return scans, gt
