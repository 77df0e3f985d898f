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
    out = subprocess.run(["g++", "-std=c++17", f"-I{os.path.join(root, 'include')}", f"-I{os.path.join(root, 'compat')}", "-c", os.path.join(HERE, "cpp", "headers_host.cpp"),
                          "-o", str(tmp_path / "headers_host.o")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]


def test_dynamic_array_alias_is_opt_in(tmp_path):
    """compat/enoki/dynamic.h (opt-in include root): the reference's DynamicArray<Packet<T>> is a HOST array (reference dynamic.h:54-60).  Using the name
    without -DENOKI_HIP_DYNAMIC_IS_DEVICE must stop the compilation with an explanation; with it the name is HIPArray<T>."""
    root = os.path.dirname(HERE)
    src = tmp_path / "dyn.cpp"
    src.write_text("#include <enoki/dynamic.h>\n"
                   "using FloatX = enoki::DynamicArray<enoki::Packet<float, 8>>;\n"
                   "#if defined(ENOKI_HIP_DYNAMIC_IS_DEVICE)\n"
                   "static_assert(std::is_same_v<FloatX, enoki::HIPArray<float>>);\n"
                   "#endif\n"
                   "size_t f() { return sizeof(FloatX); }\n")
    cmd = ["g++", "-std=c++17", f"-I{os.path.join(root, 'include')}", f"-I{os.path.join(root, 'compat')}", "-fsyntax-only", str(src)]
    off = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert off.returncode != 0 and "ENOKI_HIP_DYNAMIC_IS_DEVICE" in off.stderr and "HOST array" in off.stderr, off.stderr[-2000:]
    on = subprocess.run(cmd + ["-DENOKI_HIP_DYNAMIC_IS_DEVICE=1"], capture_output=True, text=True, timeout=600)
    assert on.returncode == 0, on.stderr[-2000:]
