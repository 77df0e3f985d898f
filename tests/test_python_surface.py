"""The names the reference's python bindings define (every `def("...")` of src/python/*.h / *.cpp, collected once) must
be present somewhere on the modules / classes of enoki_amd.hip and enoki_amd.hip_autodiff -- a guard for the name-by-name
comparison described in INTEGRATION.md.  Import-only: runs without a GPU."""
import pytest

# src/python/* of the reference, `def(...)` / `def_static(...)` / `def_property*(...)` names
REFERENCE_NAMES = """
T __enter__ __exit__ __floordiv__ __ge__ __getitem__ __gt__ __iter__ __le__ __len__ __lt__ __matmul__ __mod__ __mul__
__pow__ __repr__ __setitem__ __truediv__ abs abs_dot acos acosh advance all all_nested allclose any any_nested arange arg
asin asinh atan atan2 atanh backward cbrt ceil clamp compress conj copysign copysign_neg cos cosh cot coth count count_nested
cross csc csch data det detach dot empty eq erf erfinv eval exp floor fmadd fmsub fnmadd fnmsub forward full gather
gradient gradient_index graphviz hmax hmax_nested hmean hmean_nested hmin hmin_nested hprod hprod_nested hsum hsum_nested
identity imag index inverse inverse_transpose isfinite isinf isnan lerp lgamma linspace log log2i log_level look_at lzcnt
managed matrix_to_quat max meshgrid min mulhi mulsign mulsign_neg neq next_float32 next_uint32 next_uint32_bounded
next_uint64 next_uint64_bounded none none_nested norm normalize numpy partition popcnt pow psum quat_to_euler
quat_to_matrix rcp real reattach requires_gradient resize reverse rotate round rsqrt safe_acos safe_asin safe_rsqrt
safe_sqrt scale scatter scatter_add sec sech seed select set_gradient set_graph_simplification set_label set_log_level
set_requires_gradient set_slices shape sign simplify_graph sin sincos sincosh sinh slerp slices sqr sqrt squared_norm tan
tanh tgamma torch transform_compose transform_decompose translate transpose trunc tzcnt w whos x y z zero binary_search
""".split()

# spelled differently on purpose: host-array protocol (device arrays export __cuda_array_interface__), python-level iterator
# object (iteration goes through __iter__), the cuda_* runtime names (hip_* here; package `enoki` provides the cuda_* spellings)
NOT_CARRIED = {"__array_interface__", "__next__", "cuda_eval", "cuda_log_level", "cuda_malloc_trim", "cuda_mem_get_info",
               "cuda_set_log_level", "cuda_sync", "cuda_whos"}


def _namespaces():
    import enoki_amd.hip as c
    import enoki_amd.hip_autodiff as d
    spaces = [c, d]
    for m in (c, d):
        for name in dir(m):
            obj = getattr(m, name)
            if isinstance(obj, type):
                spaces.append(obj)
                spaces += [getattr(obj, n) for n in dir(obj) if isinstance(getattr(obj, n, None), type) and not n.startswith("__")]
    return spaces


def test_every_reference_binding_name_exists():
    spaces = _namespaces()
    missing = [n for n in REFERENCE_NAMES if not any(hasattr(s, n) for s in spaces)]
    assert not missing, missing


def test_cuda_spellings_live_in_the_compat_package():
    import enoki
    for n in sorted(NOT_CARRIED - {"__array_interface__", "__next__"}):
        assert hasattr(enoki, n), n
