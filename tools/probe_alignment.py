"""Do streams that start at the same offset modulo the allocation alignment collide in HBM?  fmadd (3 reads, 1 write) over
64 Mi floats with the four arrays staggered by different byte offsets -- C ABI, raw pointers."""
import ctypes, os, statistics, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi, hiprt
capi.init(); st = capi.stream(); P = ctypes.c_void_p
n = 1 << 26
pad = 64 << 20
bufs = [capi.Buf(np.uint8, 4 * n + pad) for _ in range(4)]
print("base addresses:", [hex(b.ptr) for b in bufs])
for b in bufs[1:]:
    capi.lib.ek_hip_memset(P(b.ptr), 0, ctypes.c_size_t(4 * n + pad))
def run(offsets):
    ptrs = [b.ptr + o for b, o in zip(bufs, offsets)]
    ops = [capi.Operand(p, 0, n) for p in ptrs[1:]]
    f = lambda: capi.check(capi.lib.ek_hip_ternary(capi.TERNARY["fmadd"], capi.NP2EK[np.dtype(np.float32)], P(ptrs[0]), ctypes.byref(ops[0]),
                                                   ctypes.byref(ops[1]), ctypes.byref(ops[2]), ctypes.c_size_t(n)))
    return statistics.median(hiprt.time_region(st, f, iters=10, warmup=2) for _ in range(5))
for name, offs in [("aligned", (0, 0, 0, 0)), ("256 B steps", (0, 256, 512, 768)), ("4 KiB steps", (0, 4096, 8192, 12288)),
                   ("64 KiB steps", (0, 65536, 131072, 196608)), ("1 MiB steps", (0, 1 << 20, 2 << 20, 3 << 20)),
                   ("1 MiB + 4 KiB steps", (0, (1 << 20) + 4096, (2 << 20) + 8192, (3 << 20) + 12288)),
                   ("16 MiB steps", (0, 16 << 20, 32 << 20, 48 << 20)), ("odd", (0, 4096 * 37, 4096 * 91, 4096 * 153))]:
    ms = run(offs)
    print(f"{name:22s} {ms:.4f} ms  {16 * n / ms / 1e9:.3f} TB/s", flush=True)
