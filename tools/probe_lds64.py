"""LDS update rates for the adjoint sums of k_bucket_pair_forward_adjoint (csrc/probe_lds64.hip) on one MI355X.

    python tools/probe_lds64.py            # the rate table (profiles/probe_lds64_r06.txt)
    python tools/probe_lds64.py --isa      # (no GPU) the DS instructions of every variant, from the built probe library
"""
import ctypes, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

#        name                                   variant entry_bytes planes updates-per-slot DS-instr-per-slot
ROWS = [("ds_read_b64", 0, 8, 1, 1),
        ("ds_write_b64", 1, 8, 1, 1),
        ("ds_add_u32", 2, 4, 1, 1),
        ("ds_add_rtn_u32", 3, 4, 1, 1),
        ("ds_add_u64", 4, 8, 1, 1),
        ("ds_add_rtn_u64", 5, 8, 1, 1),
        ("ds_add_f32", 6, 4, 1, 1),
        ("ds_add_f32, MODE f32 denorm = flush", 7, 4, 1, 1),
        ("ds_add_f64", 8, 8, 1, 1),
        ("ds_pk_add_f16", 17, 4, 1, 1),
        ("exchange lock: wrxchg_rtn_b64 + write_b64", 9, 8, 1, 2),
        ("2 x ds_add_u64, 16-byte entries", 10, 16, 1, 2),
        ("2 x ds_add_u64, two planes", 11, 8, 2, 2),
        ("2 x ds_add_u32, 8-byte entries", 12, 8, 1, 2),
        ("2 x ds_add_f64, 16-byte entries", 13, 16, 1, 2),
        ("read_b64 + 2 x ds_add_u64 (proposed, 3 planes)", 14, 8, 3, 3),
        ("read_b64 + exchange lock (today, no retries)", 15, 8, 2, 3),
        ("read_b64 + 1 x ds_add_u64 (2 x 32-bit fields)", 16, 8, 2, 2)]

if "--isa" in sys.argv:
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "enoki_amd", "libenoki-hip-probe.so")
    tmp = "/tmp/probe_lds64_isa"
    os.makedirs(tmp, exist_ok=True)
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={lib}", f"--output={tmp}/dev.co", "--unbundle"], check=False)
    print("use: /opt/rocm/lib/llvm/bin/llvm-objdump -d build/obj/probe_lds64.hip.o (device section) -- see tools/kernel_resources.py")
    sys.exit(0)

from enoki_amd import capi, hiprt
capi.init(); st = capi.stream()
sink = capi.Buf(np.float32, 1024)
lib = capi.probe_lib()
iters = 256
print("# csrc/probe_lds64.hip: 256 workgroups x 1024 threads (16 waves per CU), 8 updates in flight per lane, random entries")
print("# cycles = LDS-clock cycles per WAVE-INSTRUCTION GROUP (one slot of 64 lanes) per CU at 2.4 GHz; 'per DS instr' divides by the instructions of the slot")
for entries in (8192, 4096):
    for name, v, eb, planes, ninstr in ROWS:
        if entries * eb * planes > 160 * 1024:
            continue
        f = lambda: capi.check(lib.ek_hip_probe_lds64(v, 256, iters, entries, eb, planes, ctypes.c_void_p(sink.ptr)))
        try:
            ms = hiprt.time_region(st, f, iters=5, warmup=1)
        except Exception as e:          # noqa
            print(f"entries={entries:5d} {name:50s} FAILED {e}")
            continue
        slots = 256 * 1024 * iters * 8
        cyc = ms * 1e-3 * 2.4e9 / (slots / 256 / 64)
        print(f"entries={entries:5d} {name:50s} {ms:8.4f} ms {slots / ms / 1e6:9.1f} G slots/s  {cyc:7.2f} cycles per slot  "
              f"{cyc / ninstr:6.2f} per DS instr   -> 64 Mi elements: {64 * 2**20 / 256 / 64 * cyc / 2.4e9 * 1e6:6.1f} us of LDS")
