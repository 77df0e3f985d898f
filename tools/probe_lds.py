"""LDS atomic throughput on one MI355X (csrc/probe.hip k_probe_lds_atomic)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi, hiprt
capi.init(); st = capi.stream()
sink = capi.Buf(np.float32, 1024)
names = {0: "ds_add_f32 random", 1: "ds_add_u32 random", 2: "plain RMW random (racy)", 3: "ds_add_f32 conflict-free", 4: "ds_add_rtn_u32 random"}
iters = 2048
for bins_log2 in (6, 10, 14):
    for blocks_per_cu in (1, 2):
        blocks = 256 * blocks_per_cu
        for v in range(5):
            f = lambda: capi.check(capi.probe_lib().ek_hip_probe_lds_atomic(v, blocks, iters, bins_log2, ctypes.c_void_p(sink.ptr)))
            ms = hiprt.time_region(st, f, iters=5, warmup=1)
            ops = blocks * 512 * iters
            print(f"bins=2^{bins_log2:2d} blocks/CU={blocks_per_cu} {names[v]:28s} {ms:8.4f} ms  {ops / ms / 1e6:9.1f} Gop/s  "
                  f"{ms * 1e-3 * 2.4e9 / (ops / 256 / 64):7.1f} cycles per wave-instruction per CU")
