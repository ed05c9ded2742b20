"""CPU: the C-ABI library builds, loads and exports every symbol include/icp_mi355x.h declares; without a GPU the
product path refuses to run (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from pylidar_slam_amd import _lib
    if not os.path.exists(_lib.library_path()):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "pylidar-slam_amd", "csrc"), "-j", "8"], check=True)
    return _lib.library_path()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "icp_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(icp_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(lib_path):
    from pylidar_slam_amd import _lib
    lib = ctypes.CDLL(lib_path)
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/icp_mi355x.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared  # the ctypes binding covers exactly the header


def test_struct_layouts_match_header(lib_path):
    from pylidar_slam_amd import _lib
    assert ctypes.sizeof(_lib.IcpConfig) == 14 * 4
    assert ctypes.sizeof(_lib.IcpRegisterResult) == 16 * 4 + 6 * 4 + 4 * 4 + 8
    lib = _lib.load_library()
    cfg = _lib.IcpConfig()
    lib.icp_default_config(ctypes.byref(cfg))
    assert (cfg.height, cfg.width, cfg.max_num_alignments, cfg.local_map_size, cfg.num_neighbors_normals) == \
        (64, 1024, 100, 20, 10)
    assert abs(cfg.threshold_delta_pose - 1e-4) < 1e-10 and cfg.scheme == 0


def test_no_cpu_fallback_without_gpu(lib_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from pylidar_slam_amd.engine import IcpContext
    from pylidar_slam_amd._lib import IcpLibraryError
    with pytest.raises(IcpLibraryError, match="no CPU fallback"):
        IcpContext()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pylidar-slam_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(d, f)).read()
                assert "icp_oracle" not in src and "import oracle" not in src, os.path.join(d, f)


def test_documented_options_are_the_accepted_options():
    """`icp_set_option`: every option name the header documents is one the library accepts, and every name the library
    accepts (the chain of `k == "name"` comparisons in api.hip) is documented in the header — dev-only switches excepted."""
    header = open(os.path.join(ROOT, "include", "icp_mi355x.h")).read()
    start = header.index("MI355X-side tuning options by name")
    block = header[start:header.index("*/", start)]
    documented = set(re.findall(r'"([a-z_0-9]+)"', block))
    api = open(os.path.join(ROOT, "pylidar-slam_amd", "csrc", "api.hip")).read()
    body = api[api.index("int icp_set_option("):]
    body = body[:body.index("\nint icp_set_cost(")]
    accepted = set(re.findall(r'k == "([a-z_0-9]+)"', body))
    assert documented, "no option documented?"
    missing_in_library = documented - accepted
    assert not missing_in_library, f"documented but not accepted: {sorted(missing_in_library)}"
    undocumented = accepted - documented
    assert not undocumented, f"accepted but not documented in include/icp_mi355x.h: {sorted(undocumented)}"
