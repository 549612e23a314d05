"""The drop-in boundary from plain C: tests/c/abi_smoke.c is compiled with gcc against include/rfx_hip.h, linked with
librfx.so and run as its own process -- no Python or torch between the host code and the library."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pure_c_host(built, tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    exe = str(tmp_path / "abi_smoke")
    lib = os.path.join(ROOT, "rayforce_amd")
    subprocess.run(["gcc", "-std=c11", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", exe,
                    "-L", lib, "-lrfx", "-lm", f"-Wl,-rpath,{lib}"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "c host ok" in r.stdout


def test_pure_c_host_of_the_planner(built, tmp_path):
    """tests/c/exec_smoke.c: a C host that plans through include/rfx_exec.h -- three shards on one device, a two-level where: tree, scalar and
    grouped aggregates with FIRST -- what rfx_select itself calls, with nothing above it."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    exe = str(tmp_path / "exec_smoke")
    lib = os.path.join(ROOT, "rayforce_amd")
    subprocess.run(["gcc", "-std=c11", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "exec_smoke.c"), "-o", exe,
                    "-L", lib, "-lrfx", "-lm", f"-Wl,-rpath,{lib}"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "c planner host ok" in r.stdout
