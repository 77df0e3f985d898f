"""VALU issue rates per instruction class on one MI355X (csrc/probe_valu.hip): cycles of a SIMD per wave64 instruction with 4 waves
per SIMD (one 1024-thread workgroup per CU, the shape of the bucket kernels)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enoki_amd import capi, hiprt
capi.init(); st = capi.stream()
lib = capi.probe_lib()
names = ["v_fma_f32", "v_add_f32", "v_mul_f32", "v_add_u32", "v_and_b32", "v_xor_b32", "v_lshlrev_b32", "v_cndmask_b32", "v_cmp_lt_f32",
         "v_cvt_i32_f32", "v_cvt_f32_i32", "v_mov_b32", "v_lshl_add_u32", "v_mul_lo_u32", "v_mul_u32_u24", "v_bfe_u32", "v_and_or_b32",
         "v_max_f32", "v_rcp_f32", "v_pk_fma_f32", "v_bfi_b32", "v_sub_f32", "v_cmp_eq_u32", "v_mad_u32_u24",
         "v_cndmask_b32_e64 (SGPR pair)", "v_cmp_lt_f32_e64 -> SGPR pair", "v_cmp_e64 + v_cndmask_e64 (2 instr)", "v_or_b32", "v_sub_u32", "v_lshrrev_b32", "v_min_f32", "v_fmac_f32", "v_add3_u32",
         "v_fma_f32 |a|, -b", "v_and_b32 literal", "v_mul_f32 literal", "v_cndmask_b32 vcc (vcc set once)", "v_trunc_f32", "v_rndne_f32", "v_lshlrev_b64", "v_cmp vcc + v_cndmask vcc (2 instr)", "v_xor_b32 literal", "v_bfe_i32", "v_and_b32 sgpr",
         "v_cvt_f64_f32", "v_fma_f64", "v_fma_f64 sgpr pair", "v_add_f64", "v_mul_f32_e64 |a|, sgpr", "v_mul_f32_e64 |a|, vgpr", "v_fma_f32 sgpr", "v_mul_f32 sgpr (VOP2)",
         "v_ashrrev_i32", "v_cmp_class_f32", "v_mul_legacy_f32", "v_bitop3_b32", "v_bitop3_b32 sgpr", "v_and_b32_sdwa WORD_1", "v_pk_mul_f32", "v_fmamk_f32 literal",
         "v_lshl_add_u64", "v_add_u32 sgpr", "v_cvt_i32_f64"]
first = int(os.environ.get("PROBE_FIRST", "0"))
out = capi.Buf.from_numpy(np.zeros(2, dtype=np.uint64))
iters = 512
print("# 256 workgroups x 1024 threads (4 waves per SIMD), 64 instructions of one class per loop trip in 8 independent chains")
print("# cycles per instruction PER SIMD = (mean s_memtime cycles of a wave) / instructions of a wave / 4 waves sharing the SIMD")
for v, name in enumerate(names):
    if v < first:
        continue
    f = lambda: capi.check(lib.ek_hip_probe_valu(v, 256, iters, ctypes.c_void_p(out.ptr)))
    f(); capi.sync()
    base = int(out.numpy()[0])
    ms = hiprt.time_region(st, f, iters=3, warmup=0)
    capi.sync()
    total = int(out.numpy()[0]) - base
    waves = 256 * 16 * 3
    per_wave = total / waves
    n_inst = iters * 64
    print(f"{name:40s} {per_wave / n_inst / 4:6.2f} cycles per instruction per SIMD   ({per_wave / n_inst:6.2f} per instruction of one of 4 waves; {ms * 1e3:7.1f} us per launch)")
