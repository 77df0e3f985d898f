"""The reference's OWN test sources, compiled unmodified against this repository's headers with the device array types
substituted for the CPU ones (tests/cpp/refshim; built by enoki_amd/_build.py where /root/reference exists, the binary
travels to the GPU box).  SURVEY 8b(i): this is the templated code of the reference acting as the caller of the
HIPArray / DiffArray<HIPArray> member concept."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(name):
    exe = os.path.join(HERE, "cpp", name + ".bin")
    if not os.path.exists(exe):
        pytest.skip(f"{exe} was not built (needs /root/reference at build time)")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(r.stdout)
    m = re.search(r"(\d+)/(\d+) passed", r.stdout)
    assert m, r.stdout[-2000:]
    return int(m.group(1)), int(m.group(2)), r.stdout


def test_reference_autodiff_suite_on_device():
    """tests/autodiff.cpp of the reference: 47 tests on DiffArray<HIPArray<float>>"""
    passed, total, log = _run("reftest_autodiff_hip")
    assert total == 47, log[-3000:]
    assert passed == total, "\n".join(l for l in log.splitlines() if "failure --" in l or "FAILED" in l)
