"""`import enoki` compatibility layer over the MI355X backend.

The reference ships `enoki.cuda` / `enoki.cuda_autodiff` plus type aliases and free functions in the top-level
`enoki` namespace (src/python/main.cpp:79-140: `FloatC`, `FloatD`, `ek.atan2(...)`, `ek.set_requires_gradient(...)`).
This package offers the same spelling on top of `enoki_amd.hip` / `enoki_amd.hip_autodiff`:

    import enoki as ek
    x = ek.FloatD.linspace(0, 1, 10); ek.set_requires_gradient(x)
    ek.backward(ek.hsum(ek.atan2(x, ek.FloatD(2.0)))); g = ek.gradient(x)

`enoki.hip` and `enoki.hip_autodiff` are the two extension modules themselves (also reachable as `enoki.cuda` /
`enoki.cuda_autodiff` for scripts written against the reference); free functions dispatch on the argument types.
"""
import sys as _sys

from enoki_amd import hip, hip_autodiff

cuda, cuda_autodiff = hip, hip_autodiff
for _name, _mod in (("hip", hip), ("hip_autodiff", hip_autodiff), ("cuda", hip), ("cuda_autodiff", hip_autodiff)):
    _sys.modules[__name__ + "." + _name] = _mod

# type aliases: <Type>C = plain device array, <Type>D = differentiable device array
_TYPES = ["Float32", "Float64", "Int32", "UInt32", "Int64", "UInt64", "Mask"] + \
         [f"Vector{n}{k}" for n in range(5) for k in "miufd"] + \
         [f"Matrix{n}{k}" for n in (2, 3, 4) for k in "fd"] + ["Complex2f", "Complex2d", "Quaternion4f", "Quaternion4d"]
_SHORT = {"Float32": "Float", "Mask": "Bool"}
for _t in _TYPES:
    for _mod, _suffix in ((hip, "C"), (hip_autodiff, "D")):
        if hasattr(_mod, _t):
            globals()[_t + _suffix] = getattr(_mod, _t)
            if _t in _SHORT:
                globals()[_SHORT[_t] + _suffix] = getattr(_mod, _t)
PCG32C = hip.PCG32


def _dispatcher(name, candidates):
    def call(*args, **kwargs):
        error = None
        for fn in candidates:
            try:
                return fn(*args, **kwargs)
            except TypeError as e:          # pybind11: "incompatible function arguments" -> try the other module
                error = e
        raise error
    call.__name__ = name
    call.__doc__ = f"enoki.{name}: dispatches to enoki.hip_autodiff.{name} / enoki.hip.{name} by argument type"
    return call


def __getattr__(name):
    candidates = [getattr(m, name) for m in (hip_autodiff, hip) if hasattr(m, name) and callable(getattr(m, name))
                  and not isinstance(getattr(m, name), type)]
    if not candidates:
        raise AttributeError(f"module 'enoki' has no attribute '{name}'")
    fn = candidates[0] if len(candidates) == 1 else _dispatcher(name, candidates)
    globals()[name] = fn
    return fn


def cuda_eval():
    """no-op: the backend is eager"""
    hip.hip_eval()


def cuda_sync():
    hip.hip_sync()


def cuda_whos():
    return hip.hip_whos()


def cuda_set_log_level(level):
    hip.hip_set_log_level(level)


def cuda_log_level():
    return hip.hip_log_level()


def cuda_mem_get_info():
    return hip.hip_mem_get_info()


def shape(a):
    """(components..., entries) of an array, like the reference's ek.shape"""
    try:
        return (len(a), max(len(a[i]) for i in range(len(a)))) if hasattr(a, "x") else (len(a),)
    except TypeError:
        return (len(a),)


def cuda_malloc_trim():
    """name used by scripts written for the reference"""
    hip.hip_malloc_trim()
