"""Minimal ctypes view of the HIP runtime for measurement plumbing (events on the library's stream)."""
import ctypes

_hip = ctypes.CDLL("libamdhip64.so")


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with hipError {rc}")


class Event:
    def __init__(self):
        self.h = ctypes.c_void_p()
        _chk(_hip.hipEventCreate(ctypes.byref(self.h)), "hipEventCreate")

    def record(self, stream):
        _chk(_hip.hipEventRecord(self.h, ctypes.c_void_p(stream)), "hipEventRecord")

    def synchronize(self):
        _chk(_hip.hipEventSynchronize(self.h), "hipEventSynchronize")

    def elapsed_ms(self, end):
        ms = ctypes.c_float()
        _chk(_hip.hipEventElapsedTime(ctypes.byref(ms), self.h, end.h), "hipEventElapsedTime")
        return ms.value

    def __del__(self):
        try:
            _hip.hipEventDestroy(self.h)
        except Exception:
            pass


def time_region(stream, fn, iters=20, warmup=3):
    """average milliseconds per call of ``fn`` (which enqueues work on ``stream``)"""
    for _ in range(warmup):
        fn()
    e0, e1 = Event(), Event()
    e0.record(stream)
    for _ in range(iters):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_ms(e1) / iters
