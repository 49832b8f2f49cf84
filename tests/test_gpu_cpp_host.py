"""Runs the C++ host-mirror tests (tests/cpp/visual_tests.cpp): the reference's visual_blend / visual_filters tests
written against include/pfx.hpp, checked against the reference goldens with tolerance 0."""
import os
import struct
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_mirror(golden, tmp_path):
    exe = os.path.join(ROOT, "tests", "cpp", "visual_tests")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    path = tmp_path / "golden.bin"
    with open(path, "wb") as f:
        for key in golden.files:
            if key.startswith(("blend/", "filters/")):
                img = golden[key]
                f.write(key.encode() + b"\0" + struct.pack("<II", img.shape[1], img.shape[0]) + img.tobytes())
    p = subprocess.run([exe, str(path)], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert " 0 failed" in p.stdout
