"""The drop-in boundary used from plain C (examples/capi_demo.c): compiled with gcc against include/enoki_hip.h only,
linked to libenoki-hip.so, run on the GPU.  The header must stay valid C11 (also checked without a GPU)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "capi_demo.c")
LIBDIR = os.path.join(ROOT, "enoki_amd")


def build(out):
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", f"-I{os.path.join(ROOT, 'include')}", SRC, f"-L{LIBDIR}", "-lenoki-hip",
           f"-Wl,-rpath,{LIBDIR}", "-lm", "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_c_example_compiles_as_c11(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "libenoki-hip.so")):
        pytest.skip("libenoki-hip.so has not been built")
    build(tmp_path / "capi_demo")


@pytest.mark.gpu
def test_c_example_runs(tmp_path):
    exe = tmp_path / "capi_demo"
    build(exe)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "scatter_add bins wrong: 0" in r.stdout
