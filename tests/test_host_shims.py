"""The host-side C++ mirrors of the reference surfaces (slam-module_amd/host/) compile, link against the C ABI and run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "slam-module_amd", "lib", "host_shim_smoke")


def _build():
    lib = os.path.join(ROOT, "slam-module_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "slam-module_amd", "host"),
                           os.path.join(ROOT, "tests", "host_shim_smoke.cpp"), "-o", EXE, "-L", lib, "-lmi355slam", "-Wl,-rpath," + lib,
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64"])


def test_host_shims_compile_and_link():
    _build()
    out = subprocess.check_output([EXE, "--no-gpu"], text=True)
    assert "link ok" in out


@pytest.mark.gpu
def test_host_shims_run_end_to_end():
    _build()          # always rebuild: the ABI structs in the header may have changed since a stale binary was built
    out = subprocess.check_output([EXE], text=True, timeout=120)
    assert "host shims ok" in out, out
