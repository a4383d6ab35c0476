"""GPU: a plain C++ host program (no Python, no torch in the process) links libsnac_hip.so against the system HIP
runtime, drives the C ABI with hipMalloc'ed arrays on its own stream, and compares with the C oracle linked beside it."""
import os
import subprocess

import pytest

import helpers

pytestmark = pytest.mark.gpu


def test_cpp_host_links_and_matches_oracle(tmp_path):
    helpers.oracle().build()
    root = helpers.ROOT
    exe = str(tmp_path / "host_parity")
    hipcc = "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "-O2", "-std=c++17", "-I", os.path.join(root, "include"), "-I", os.path.join(root, "oracle"),
           os.path.join(root, "tests", "native", "host_parity.cpp"), "-L", os.path.join(root, "snac_amd"), "-lsnac_hip",
           "-L", os.path.join(root, "oracle"), "-lsnac_oracle", "-Wl,-rpath," + os.path.join(root, "snac_amd"),
           "-Wl,-rpath," + os.path.join(root, "oracle"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PARITY OK" in r.stdout, (r.stdout[-500:], r.stderr[-500:])
