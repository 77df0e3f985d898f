// LDS update-rate probe, round 6: what ONE update of a per-entry {sum0, sum1} pair costs under the access pattern of
// k_bucket_pair_forward_adjoint (1024 threads per CU, random entries of an 8 Ki-entry bucket), for every instruction form that
// could carry it.  The round-1 probe (probe.hip: k_probe_lds_atomic) spent ~10 vector instructions per update on its hash, which
// at 8-16 waves per CU is itself ~10 cycles per wave-instruction: its "ds_add_u32 = 10 cycles" is that floor, not the LDS.  Here an
// address costs two full-rate vector instructions (xor with a uniform value, and with the byte mask): XOR by a uniform value
// permutes the entries without changing which lanes of a wave collide on a bank, so each (wave, slot) keeps the conflict pattern
// of its random draw and the average over 16 waves x 8 slots x 256 workgroups is that of random indices.
//
// The instructions are written as inline assembly so that the table names what was measured (the disassembly is checked by
// tools/probe_lds64.py --isa).
#include "ek_internal.h"
#include <hip/hip_runtime.h>

namespace ek {

enum {
    LV_READ_B64 = 0,      // ds_read_b64                      the record read of the kernel (reference point)
    LV_WRITE_B64,         // ds_write_b64
    LV_ADD_U32,           // ds_add_u32
    LV_ADD_RTN_U32,       // ds_add_rtn_u32
    LV_ADD_U64,           // ds_add_u64
    LV_ADD_RTN_U64,       // ds_add_rtn_u64
    LV_ADD_F32,           // ds_add_f32                       MODE as the library runs (f32 denormals preserved)
    LV_ADD_F32_FTZ,       // ds_add_f32                       MODE.fp_denorm(f32) = flush, set around the loop
    LV_ADD_F64,           // ds_add_f64
    LV_XCHG_WRITE,        // ds_wrxchg_rtn_b64 + ds_write_b64 the exchange lock of today: claim, (add), release -- no retry round
    LV_2ADD_U64,          // 2 x ds_add_u64 (offset:0, offset:8)  a 16-byte {s0, s1} entry, 64-bit fixed point (each add sees HALF the bank pairs)
    LV_2ADD_U64_SOA,      // 2 x ds_add_u64 into two 8-byte-stride planes s0[l], s1[l]
    LV_2ADD_U32,          // 2 x ds_add_u32 (offset:0, offset:4)  an 8-byte entry, 32-bit fixed point
    LV_2ADD_F64,          // 2 x ds_add_f64
    LV_READ_2ADD_U64,     // ds_read_b64 + 2 x ds_add_u64     the whole per-element LDS work of the proposed kernel: planes rec[l], s0[l], s1[l]
    LV_READ_XCHG_WRITE,   // ds_read_b64 + exchange + write   the whole per-element LDS work of today's kernel (no retries): planes rec[l], pair[l]
    LV_READ_ADD_U64,      // ds_read_b64 + 1 x ds_add_u64     (two 32-bit fixed-point fields in one 64-bit add): planes rec[l], packed[l]
    LV_PK_ADD_F16,        // ds_pk_add_f16                    (rate only; not a candidate: 11-bit sums)
    LV_COUNT
};

template <int Variant>
__global__ __launch_bounds__(1024) void k_probe_lds64(float *__restrict__ sink, int iters, unsigned entries, unsigned entry_bytes, unsigned planes) {
    extern __shared__ __align__(16) unsigned char raw[];
    const unsigned plane = entries * entry_bytes, total = plane * planes;
    for (unsigned j = threadIdx.x * 4u; j < total; j += 4096u) *reinterpret_cast<unsigned *>(raw + j) = 0u;
    __syncthreads();
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const unsigned base = (unsigned) (uintptr_t) (lds_byte *) raw;
    // eight random entries per lane (as byte offsets of entry starts)
    unsigned r[8];
    unsigned h = (blockIdx.x * 1024u + threadIdx.x) * 8u + 1u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        unsigned v = h + (unsigned) j;
        v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
        r[j] = v;
    }
    const unsigned shift = entry_bytes == 4 ? 2 : entry_bytes == 8 ? 3 : 4;      // entry_bytes: 4, 8 or 16
    // entries is a power of two here (the rate does not depend on it); the mask is applied to the BYTE offset
    const unsigned bmask = (entries - 1u) << shift;
    unsigned long long acc = 0;
    unsigned old_mode = 0;
    if constexpr (Variant == LV_ADD_F32_FTZ) {
        old_mode = __builtin_amdgcn_s_getreg((1 /* MODE */) | (4 << 6) | (1 << 11));
        __builtin_amdgcn_s_setreg((1 /* MODE */) | (4 << 6) | (1 << 11), 0);           // f32 denormals: flush in and out
    }
    const unsigned long long one64 = 0x0000000100000001ull;
    const double oned = 1.0;
    const float onef = 1.0f;
    for (int it = 0; it < iters; ++it) {
        const unsigned salt = (unsigned) it * 0x9E3779B1u;
        // (the compiler does not know that an inline-assembly DS instruction returns LATER: every returned value stays in its own
        // register until the s_waitcnt behind the eight slots, and is only consumed after it)
        unsigned long long ret[8] = {};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned a = base + ((r[j] ^ salt) & bmask);
            unsigned long long &v = ret[j];
            if constexpr (Variant == LV_READ_B64) {
                asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a) : "memory");
            } else if constexpr (Variant == LV_WRITE_B64) {
                asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(one64) : "memory");
            } else if constexpr (Variant == LV_ADD_U32) {
                asm volatile("ds_add_u32 %0, %1" :: "v"(a), "v"(1u) : "memory");
            } else if constexpr (Variant == LV_ADD_RTN_U32) {
                unsigned v32;
                asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(v32) : "v"(a), "v"(1u) : "memory");
                asm volatile("" : "=v"(v) : "0"((unsigned long long) v32));       // (keeps v32's register reserved: v aliases it)
            } else if constexpr (Variant == LV_ADD_U64) {
                asm volatile("ds_add_u64 %0, %1" :: "v"(a), "v"(one64) : "memory");
            } else if constexpr (Variant == LV_ADD_RTN_U64) {
                asm volatile("ds_add_rtn_u64 %0, %1, %2" : "=v"(v) : "v"(a), "v"(one64) : "memory");
            } else if constexpr (Variant == LV_ADD_F32 || Variant == LV_ADD_F32_FTZ) {
                asm volatile("ds_add_f32 %0, %1" :: "v"(a), "v"(onef) : "memory");
            } else if constexpr (Variant == LV_ADD_F64) {
                asm volatile("ds_add_f64 %0, %1" :: "v"(a), "v"(oned) : "memory");
            } else if constexpr (Variant == LV_XCHG_WRITE) {
                unsigned long long x;
                asm volatile("ds_wrxchg_rtn_b64 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x) : "v"(a), "v"(one64) : "memory");
                x += one64;
                asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(x) : "memory");
            } else if constexpr (Variant == LV_2ADD_U64) {
                asm volatile("ds_add_u64 %0, %1\n\tds_add_u64 %0, %1 offset:8" :: "v"(a), "v"(one64) : "memory");
            } else if constexpr (Variant == LV_2ADD_U64_SOA) {
                const unsigned a1 = a + plane;
                asm volatile("ds_add_u64 %0, %2\n\tds_add_u64 %1, %2" :: "v"(a), "v"(a1), "v"(one64) : "memory");
            } else if constexpr (Variant == LV_2ADD_U32) {
                asm volatile("ds_add_u32 %0, %1\n\tds_add_u32 %0, %1 offset:4" :: "v"(a), "v"(1u) : "memory");
            } else if constexpr (Variant == LV_2ADD_F64) {
                asm volatile("ds_add_f64 %0, %1\n\tds_add_f64 %0, %1 offset:8" :: "v"(a), "v"(oned) : "memory");
            } else if constexpr (Variant == LV_READ_2ADD_U64) {
                const unsigned a1 = a + plane, a2 = a1 + plane;
                asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a) : "memory");
                asm volatile("ds_add_u64 %0, %2\n\tds_add_u64 %1, %2" :: "v"(a1), "v"(a2), "v"(one64) : "memory");
            } else if constexpr (Variant == LV_READ_ADD_U64) {
                const unsigned a1 = a + plane;
                asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a) : "memory");
                asm volatile("ds_add_u64 %0, %1" :: "v"(a1), "v"(one64) : "memory");
            } else if constexpr (Variant == LV_READ_XCHG_WRITE) {
                unsigned long long x, w;
                const unsigned a1 = a + plane;
                asm volatile("ds_read_b64 %0, %2\n\tds_wrxchg_rtn_b64 %1, %3, %4\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w), "=&v"(x) : "v"(a), "v"(a1), "v"(one64) : "memory");
                x += w;
                asm volatile("ds_write_b64 %0, %1" :: "v"(a1), "v"(x) : "memory");
            } else if constexpr (Variant == LV_PK_ADD_F16) {
                asm volatile("ds_pk_add_f16 %0, %1" :: "v"(a), "v"(0x3C003C00u) : "memory");
            }
        }
        if constexpr (Variant == LV_READ_B64 || Variant == LV_ADD_RTN_U32 || Variant == LV_ADD_RTN_U64 || Variant == LV_READ_2ADD_U64 ||
                      Variant == LV_READ_ADD_U64) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ret[0]), "+v"(ret[1]), "+v"(ret[2]), "+v"(ret[3]), "+v"(ret[4]), "+v"(ret[5]), "+v"(ret[6]), "+v"(ret[7]) :: "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += ret[j];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (Variant == LV_ADD_F32_FTZ) __builtin_amdgcn_s_setreg((1 /* MODE */) | (4 << 6) | (1 << 11), old_mode);
    __syncthreads();
    const float probe = *reinterpret_cast<float *>(raw + (threadIdx.x & 255u) * 4u);
    if (probe == 12345.678f || acc == 0xdeadbeefdeadbeefull) sink[threadIdx.x] = probe;
}

} // namespace ek

using namespace ek;

/// one launch of `blocks` workgroups of 1024 threads, `iters` x 8 updates per lane into `entries` entries of `entry_bytes` bytes,
/// `planes` such arrays behind one another (the variants with a record read next to the sums use one plane each)
extern "C" EK_API int ek_hip_probe_lds64(int variant, int blocks, int iters, unsigned entries, unsigned entry_bytes, unsigned planes, float *sink) {
    if (int rc = ensure_init()) return rc;
    Context &cx = ctx();
    const size_t lds = (size_t) entries * entry_bytes * planes;
    if (lds > 160 * 1024 || (entries & (entries - 1)) != 0 || (entry_bytes != 4 && entry_bytes != 8 && entry_bytes != 16)) return EK_ERR_INVALID;
#define EK_LV(V) case V: \
        if (lds > 64 * 1024) (void) hipFuncSetAttribute(reinterpret_cast<const void *>(&k_probe_lds64<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); \
        hipLaunchKernelGGL((k_probe_lds64<V>), dim3(blocks), dim3(1024), lds, cx.stream, sink, iters, entries, entry_bytes, planes); break;
    switch (variant) {
        EK_LV(0) EK_LV(1) EK_LV(2) EK_LV(3) EK_LV(4) EK_LV(5) EK_LV(6) EK_LV(7) EK_LV(8) EK_LV(9) EK_LV(10) EK_LV(11) EK_LV(12) EK_LV(13)
        EK_LV(14) EK_LV(15) EK_LV(16) EK_LV(17)
        default: return EK_ERR_INVALID;
    }
#undef EK_LV
    EK_LAUNCH_CHECK("probe_lds64", (size_t) blocks * 1024 * iters * 8, 0);
    return EK_OK;
}
