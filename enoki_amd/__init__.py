"""enoki_amd -- MI355X (gfx950) native array + autodiff backend behind the Enoki API.

Layout (only what the north-star hot path needs):
    csrc/      HIP kernels + C ABI            -> libenoki-hip.so            (include/enoki_hip.h)
    src/       Tape<HIPArray<float>> instance -> libenoki-hip-autodiff.so   (include/enoki/autodiff.h)
    python/    pybind11 modules               -> enoki_amd.hip, enoki_amd.hip_autodiff
    capi.py    ctypes view of the C ABI (tests, tools)
    dist.py    index-range sharding across GPUs (torch.distributed / RCCL)

Nothing here falls back to the CPU: importing a submodule without its shared library raises.
"""
__version__ = "0.1.0"
