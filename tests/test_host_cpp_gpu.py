"""Runs the C++ host-mirror test program (tests/cpp/test_host_api.cpp), which drives
the reference-named functors of visualslam_amd/host/vslam_host.hpp over the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_host_api")
XML = os.path.join(ROOT, "visualslam_amd", "data", "camera.xml")


@pytest.mark.gpu
def test_cpp_host_mirror():
    assert os.path.exists(EXE), "tests/cpp/test_host_api is not built: run __graft_entry__.build()"
    r = subprocess.run([EXE, XML], capture_output=True, text=True, timeout=100)
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert "0 failures" in r.stdout
