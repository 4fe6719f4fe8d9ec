"""The C ABI from a plain C99 program (tests/c_abi/smoke.c): no Python, no torch, no C++ between the caller and libafhip.so.
CPU: it compiles and links against include/af_hip.h + the shipped library with gcc.  GPU: it runs and checks BN folding, a
3x1x1 convolution + BN + residual + ReLU and a max-pool against loops on the host, and that a NULL argument is refused with an
error text instead of a launch."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "spatiotemporal-deepfake-detection-for-live-video-calls_amd")
SRC = os.path.join(ROOT, "tests", "c_abi", "smoke.c")


def _build(tmp_path):
    if shutil.which("gcc") is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("gcc / ROCm headers not available")
    if not os.path.exists(os.path.join(PKG, "libafhip.so")):
        pytest.skip("libafhip.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    exe = os.path.join(str(tmp_path), "c_abi_smoke")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", SRC, "-o", exe,
           "-L" + PKG, "-lafhip", "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_caller_compiles_and_links(tmp_path):
    exe = _build(tmp_path)
    assert os.path.getsize(exe) > 0
    # every entry point the program uses is resolved from libafhip.so, not from somewhere else
    needed = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
    assert "libafhip.so" in needed


@pytest.mark.gpu
def test_c_caller_runs(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "c_abi_smoke OK" in r.stdout, (r.returncode, r.stdout, r.stderr)
