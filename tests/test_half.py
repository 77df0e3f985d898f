"""enoki/half.h (binary16 storage type) and load / store of static arrays (array.h): tests/cpp/half_host.cpp checks every
encoding, 8.7 M roundings against an independent rounding in double arithmetic (and 70 M more against F16C when the
machine has it), the reference's own Array<half, 4> test (tests/float.cpp:215-238) and the load / store forms."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_half_and_load_store_on_the_host():
    out = subprocess.run([os.path.join(HERE, "cpp", "half_host.bin")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    assert "65536 encodings" in out.stdout


def test_router_helpers_on_host_packets():
    """hmean, *_nested, *_inner, any_or / all_or / none_or, rad_to_deg, abs_dot, copysign_neg, fmaddsub, rol_array, low / high
    (array_router.h / array_static.h routines that compose from the core operations): tests/cpp/router_host.cpp"""
    out = subprocess.run([os.path.join(HERE, "cpp", "router_host.bin")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "router_host:" in out.stdout


def test_every_reference_header_name_is_includable(tmp_path):
    """<enoki/fwd.h>, <enoki/array_traits.h>, <enoki/array_router.h>, ... <enoki/array_math.h> and the type headers in one
    translation unit (tests/cpp/headers_host.cpp): compile-only"""
    root = os.path.dirname(HERE)
    out = subprocess.run(["g++", "-std=c++17", f"-I{os.path.join(root, 'include')}", "-c", os.path.join(HERE, "cpp", "headers_host.cpp"),
                          "-o", str(tmp_path / "headers_host.o")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
