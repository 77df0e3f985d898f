// Bandwidth probe: the streaming-kernel design space (vectors in flight per lane, non-temporal
// loads/stores, grid shape) instantiated for a handful of representative bodies so that the
// compile-time choices in ek_map.h (EK_MAP_U, EK_MAP_NT) and the blocks_per_cu default can be
// re-measured on hardware.  tools/probe_bw.py drives it; results are kept under profiles/.
#include "ek_map.h"
#include <enoki/device/ek_math.h>

namespace ek {

using V4 = __attribute__((ext_vector_type(4))) float;

template <bool NTL> __device__ __forceinline__ V4 ld(const V4 *p) {
    if constexpr (NTL) return __builtin_nontemporal_load(p); else return *p;
}
template <bool NTS> __device__ __forceinline__ void st(V4 *p, V4 v) {
    if constexpr (NTS) __builtin_nontemporal_store(v, p); else *p = v;
}

// Body 0: copy (1R 1W)   1: fmadd (3R 1W)   2: sincos (1R 2W)   3: read-only sum (1R)   4: scale (1R 1W, scalar operand)
template <int Body, int U, bool NTL, bool NTS, bool OneShot>
__global__ __launch_bounds__(256) void k_probe(V4 *__restrict__ o0, V4 *__restrict__ o1, const V4 *__restrict__ a,
                                               const V4 *__restrict__ b, const V4 *__restrict__ c, size_t nvec) {
    const size_t gid = (size_t) blockIdx.x * 256 + threadIdx.x, total = (size_t) gridDim.x * 256;
    V4 acc = { 0, 0, 0, 0 };
    for (size_t v0 = gid; v0 < nvec; v0 += total * U) {
        V4 pa[U], pb[U], pc[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            size_t v = v0 + k * total;
            if (v < nvec) {
                pa[k] = ld<NTL>(a + v);
                if constexpr (Body == 1) { pb[k] = ld<NTL>(b + v); pc[k] = ld<NTL>(c + v); }
                if constexpr (Body == 5) { pb[k] = ld<NTL>(b + v); }
            }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            size_t v = v0 + k * total;
            if (v < nvec) {
                if constexpr (Body == 0) {
                    st<NTS>(o0 + v, pa[k]);
                } else if constexpr (Body == 1) {
                    V4 r;
#pragma unroll
                    for (int i = 0; i < 4; ++i) r[i] = __builtin_fmaf(pa[k][i], pb[k][i], pc[k][i]);
                    st<NTS>(o0 + v, r);
                } else if constexpr (Body == 2) {
                    V4 s, co;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { float ss, cc; dev::sincos_f32<true, true>(pa[k][i], ss, cc); s[i] = ss; co[i] = cc; }
                    st<NTS>(o0 + v, s);
                    st<NTS>(o1 + v, co);
                } else if constexpr (Body == 5) {           // 2 inputs, 1 output (safe_mul(x, g))
                    st<NTS>(o0 + v, pa[k] * pb[k]);
                } else if constexpr (Body == 6) {           // 1 input, 1 output, transcendental (sin)
                    V4 r;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { float ss, cc; dev::sincos_f32<true, false>(pa[k][i], ss, cc); r[i] = ss; }
                    st<NTS>(o0 + v, r);
                } else if constexpr (Body == 3) {
                    acc += pa[k];
                } else {
                    st<NTS>(o0 + v, pa[k] * 1.5f);
                }
            }
        }
        if (OneShot) break;
    }
    if constexpr (Body == 3) {
        if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) o0[gid] = acc;   // keep the loads alive
    }
}

template <int Body, int U, bool NTL, bool NTS>
int probe_launch(int blocks_per_cu, void *o0, void *o1, const void *a, const void *b, const void *c, size_t n) {
    Context &cx = ctx();
    size_t nvec = n / 4;
    if (blocks_per_cu > 0) {
        unsigned grid = stream_grid((nvec + U - 1) / U, blocks_per_cu);
        hipLaunchKernelGGL((k_probe<Body, U, NTL, NTS, false>), dim3(grid), dim3(256), 0, cx.stream, (V4 *) o0, (V4 *) o1,
                           (const V4 *) a, (const V4 *) b, (const V4 *) c, nvec);
    } else {
        size_t blocks = (nvec + 256 * (size_t) U - 1) / (256 * (size_t) U);
        hipLaunchKernelGGL((k_probe<Body, U, NTL, NTS, true>), dim3((unsigned) blocks), dim3(256), 0, cx.stream, (V4 *) o0,
                           (V4 *) o1, (const V4 *) a, (const V4 *) b, (const V4 *) c, nvec);
    }
    EK_LAUNCH_CHECK("probe", n, 0);
    return EK_OK;
}

template <int Body, int U>
int probe_nt(int ntl, int nts, int bpc, void *o0, void *o1, const void *a, const void *b, const void *c, size_t n) {
    if (ntl && nts) return probe_launch<Body, U, true, true>(bpc, o0, o1, a, b, c, n);
    if (ntl) return probe_launch<Body, U, true, false>(bpc, o0, o1, a, b, c, n);
    if (nts) return probe_launch<Body, U, false, true>(bpc, o0, o1, a, b, c, n);
    return probe_launch<Body, U, false, false>(bpc, o0, o1, a, b, c, n);
}

template <int Body>
int probe_u(int u, int ntl, int nts, int bpc, void *o0, void *o1, const void *a, const void *b, const void *c, size_t n) {
    switch (u) {
        case 1: return probe_nt<Body, 1>(ntl, nts, bpc, o0, o1, a, b, c, n);
        case 2: return probe_nt<Body, 2>(ntl, nts, bpc, o0, o1, a, b, c, n);
        case 4: return probe_nt<Body, 4>(ntl, nts, bpc, o0, o1, a, b, c, n);
        case 8: return probe_nt<Body, 8>(ntl, nts, bpc, o0, o1, a, b, c, n);
        default: return fail(EK_ERR_INVALID, "ek_hip_probe(): unroll must be 1, 2, 4 or 8");
    }
}

// ---- scatter_add experiments ---------------------------------------------------------------------
// mode 0: every lane adds into ONE shared table (what ek_hip_scatter_add does);
// mode 1: every workgroup adds into the copy of the table that belongs to its XCD (HW_REG_XCC_ID),
//         so a table line is only ever owned by one XCD's L2; the 8 copies are summed afterwards.
template <int Mode>
__global__ __launch_bounds__(256) void k_probe_scatter_add(float *__restrict__ table, size_t table_size,
                                                           const V4 *__restrict__ val, const uint4 *__restrict__ idx,
                                                           size_t nvec) {
    size_t v = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (v >= nvec) return;
    float *t = table;
    if constexpr (Mode == 1) t += (size_t) (__builtin_amdgcn_s_getreg(6164) & 7u) * table_size;   // hwreg(XCC_ID, 0, 4)
    V4 pv = __builtin_nontemporal_load(val + v);
    uint4 pi = idx[v];
    unsafeAtomicAdd(t + pi.x, pv[0]);
    unsafeAtomicAdd(t + pi.y, pv[1]);
    unsafeAtomicAdd(t + pi.z, pv[2]);
    unsafeAtomicAdd(t + pi.w, pv[3]);
}

__global__ __launch_bounds__(256) void k_probe_fold8(float *__restrict__ out, const float *__restrict__ copies, size_t k) {
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= k) return;
    float s = out[i];
#pragma unroll
    for (int c = 0; c < 8; ++c) s += copies[(size_t) c * k + i];
    out[i] = s;
}

// ---- gather experiments ---------------------------------------------------------------------------
// E elements per lane (index loaded as one vector when E == 4), table load policy: 0 plain, 1 non-temporal,
// 2 via __builtin_amdgcn_global_load with sc bits is not expressible in HIP -> only 0/1 here.
template <int E, int Policy>
__global__ __launch_bounds__(256) void k_probe_gather(float *__restrict__ out, const float *__restrict__ table,
                                                      const uint32_t *__restrict__ idx, size_t n) {
    size_t e = ((size_t) blockIdx.x * 256 + threadIdx.x) * E;
    if (e + E > n) return;
    uint32_t ix[E];
    float v[E];
    if constexpr (E == 4) {
        using U4 = __attribute__((ext_vector_type(4))) uint32_t;
        U4 p = __builtin_nontemporal_load(reinterpret_cast<const U4 *>(idx + e));
        ix[0] = p[0]; ix[1] = p[1]; ix[2] = p[2]; ix[3] = p[3];
    } else {
#pragma unroll
        for (int k = 0; k < E; ++k) ix[k] = __builtin_nontemporal_load(idx + e + k);
    }
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rsrc;
    if constexpr (Policy >= 3) rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(table), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int k = 0; k < E; ++k) {
        if constexpr (Policy == 1) v[k] = __builtin_nontemporal_load(table + ix[k]);
        else if constexpr (Policy == 2) v[k] = __hip_atomic_load(table + ix[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1
        else if constexpr (Policy == 3) v[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, ix[k] * 4u, 0, 0));
        else if constexpr (Policy == 4) v[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, ix[k] * 4u, 0, 1));    // sc0
        else if constexpr (Policy == 5) v[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, ix[k] * 4u, 0, 17));   // sc0 sc1
        else v[k] = table[ix[k]];
    }
    if constexpr (E == 4) {
        V4 r = { v[0], v[1], v[2], v[3] };
        __builtin_nontemporal_store(r, reinterpret_cast<V4 *>(out + e));
    } else {
#pragma unroll
        for (int k = 0; k < E; ++k) __builtin_nontemporal_store(v[k], out + e + k);
    }
}

// ---- shared-index gather pairs, fused into their consumer (round 2) ---------------------------------
// a = gather(A, idx), b = gather(B, idx), u = fmadd(a, x, b): what does it cost when
//   variant 0: two 4-byte lookups per element, a and b written out            (= k_gather_multi<2>)
//   variant 1: ONE 8-byte lookup per element from an interleaved {A[k], B[k]} table, a and b written out
//   variant 2: two 4-byte lookups, consumed in place: only u = fma(a, x, b) is written
//   variant 3: one 8-byte lookup, consumed in place
//   variant 4: a streamed (materialised by an earlier launch), b looked up, u written
//   variant 5: one 4-byte lookup consumed in place: u = fma(a, x, c) with c streamed
// A/B are K floats each; AB is the interleaved table (2K floats).
using U4 = __attribute__((ext_vector_type(4))) uint32_t;
using V2 = __attribute__((ext_vector_type(2))) float;

template <int Variant>
__global__ __launch_bounds__(256) void k_probe_gather_pair(float *__restrict__ o0, float *__restrict__ o1,
                                                           const float *__restrict__ A, const float *__restrict__ B,
                                                           const V2 *__restrict__ AB, const float *__restrict__ x,
                                                           const float *__restrict__ s, const uint32_t *__restrict__ idx, size_t n) {
    size_t e = ((size_t) blockIdx.x * 256 + threadIdx.x) * 4;
    if (e + 4 > n) return;
    U4 p = __builtin_nontemporal_load(reinterpret_cast<const U4 *>(idx + e));
    V4 a, b;
    if constexpr (Variant == 0 || Variant == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = A[p[k]];
#pragma unroll
        for (int k = 0; k < 4; ++k) b[k] = B[p[k]];
    } else if constexpr (Variant == 1 || Variant == 3) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { V2 t = AB[p[k]]; a[k] = t[0]; b[k] = t[1]; }
    } else if constexpr (Variant == 4) {
        a = __builtin_nontemporal_load(reinterpret_cast<const V4 *>(s + e));
#pragma unroll
        for (int k = 0; k < 4; ++k) b[k] = B[p[k]];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = A[p[k]];
        b = __builtin_nontemporal_load(reinterpret_cast<const V4 *>(s + e));
    }
    if constexpr (Variant <= 1) {
        __builtin_nontemporal_store(a, reinterpret_cast<V4 *>(o0 + e));
        __builtin_nontemporal_store(b, reinterpret_cast<V4 *>(o1 + e));
    } else {
        V4 xv = __builtin_nontemporal_load(reinterpret_cast<const V4 *>(x + e)), r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = __builtin_fmaf(a[k], xv[k], b[k]);
        __builtin_nontemporal_store(r, reinterpret_cast<V4 *>(o0 + e));
    }
}

// L2 blocking in time for the pair lookup: launch h serves the elements whose index falls into slice h of the table
// (the interleaved slice is K / S * 8 bytes: with S = 2 and K = 1 Mi each launch works out of 4 MiB, one XCD's L2); both
// launches stream idx and x, each element's result is written by exactly one of them (4-byte stores under a lane mask).
__global__ __launch_bounds__(256) void k_probe_gather_pair_sliced(float *__restrict__ o0, const V2 *__restrict__ AB,
                                                                  const float *__restrict__ x, const uint32_t *__restrict__ idx,
                                                                  size_t n, uint32_t lo, uint32_t hi) {
    size_t e = ((size_t) blockIdx.x * 256 + threadIdx.x) * 4;
    if (e + 4 > n) return;
    U4 p = __builtin_nontemporal_load(reinterpret_cast<const U4 *>(idx + e));
    bool in[4];
    bool any = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) { in[k] = p[k] >= lo && p[k] < hi; any = any || in[k]; }
    if (!any) return;
    V4 xv = __builtin_nontemporal_load(reinterpret_cast<const V4 *>(x + e));
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (in[k]) { V2 t = AB[p[k]]; r[k] = __builtin_fmaf(t[0], xv[k], t[1]); }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (in[k]) o0[e + k] = r[k];
}

__global__ __launch_bounds__(256) void k_probe_interleave(V2 *__restrict__ AB, const float *__restrict__ A,
                                                          const float *__restrict__ B, size_t k) {
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i < k) { V2 t = { A[i], B[i] }; AB[i] = t; }
}

// ---- LDS atomic throughput -------------------------------------------------------------------------
// variant 0: ds_add_f32 random bins   1: ds_add_u32 random bins   2: plain ds read-modify-write (racy, bound only)
// 3: ds_add_f32, lane-private bins (no conflicts)   4: ds_add_rtn_u32 random (returning)
template <int Variant>
__global__ __launch_bounds__(512) void k_probe_lds_atomic(float *__restrict__ sink, int iters, int bins_log2) {
    extern __shared__ __align__(16) unsigned char raw[];
    float *accf = reinterpret_cast<float *>(raw);
    unsigned *accu = reinterpret_cast<unsigned *>(raw);
    const unsigned bins = 1u << bins_log2;
    for (unsigned j = threadIdx.x; j < bins; j += 512) accf[j] = 0.f;
    __syncthreads();
    unsigned h = blockIdx.x * 512u + threadIdx.x + 1u;
    unsigned ret = 0;
    for (int it = 0; it < iters; ++it) {
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        unsigned b = h & (bins - 1);
        if constexpr (Variant == 0) atomicAdd(&accf[b], 1.0f);
        else if constexpr (Variant == 1) atomicAdd(&accu[b], 1u);
        else if constexpr (Variant == 2) accf[b] = accf[b] + 1.0f;
        else if constexpr (Variant == 3) atomicAdd(&accf[(threadIdx.x + (unsigned) it * 512u) & (bins - 1)], 1.0f);
        else ret += atomicAdd(&accu[b], 1u);
    }
    __syncthreads();
    if (accf[threadIdx.x] == 12345.678f || ret == 0xdeadbeefu) sink[threadIdx.x] = accf[threadIdx.x];
}

} // namespace ek

using namespace ek;

extern "C" EK_API int ek_hip_probe_lds_atomic(int variant, int blocks, int iters, int bins_log2, float *sink) {
    if (int rc = ensure_init()) return rc;
    Context &cx = ctx();
    size_t lds = (size_t) 4 << bins_log2;
    switch (variant) {
        case 0: hipLaunchKernelGGL((k_probe_lds_atomic<0>), dim3(blocks), dim3(512), lds, cx.stream, sink, iters, bins_log2); break;
        case 1: hipLaunchKernelGGL((k_probe_lds_atomic<1>), dim3(blocks), dim3(512), lds, cx.stream, sink, iters, bins_log2); break;
        case 2: hipLaunchKernelGGL((k_probe_lds_atomic<2>), dim3(blocks), dim3(512), lds, cx.stream, sink, iters, bins_log2); break;
        case 3: hipLaunchKernelGGL((k_probe_lds_atomic<3>), dim3(blocks), dim3(512), lds, cx.stream, sink, iters, bins_log2); break;
        default: hipLaunchKernelGGL((k_probe_lds_atomic<4>), dim3(blocks), dim3(512), lds, cx.stream, sink, iters, bins_log2); break;
    }
    EK_LAUNCH_CHECK("probe_lds_atomic", (size_t) blocks * 512 * iters, 0);
    return EK_OK;
}

// L2 blocking in time: one launch per table slice [lo, hi); lanes whose index falls outside skip (their output element is
// written by another launch).  `out` is written with per-lane 4-byte stores.
__global__ __launch_bounds__(256) void k_probe_gather_range(float *__restrict__ out, const float *__restrict__ table,
                                                            const uint32_t *__restrict__ idx, size_t n, uint32_t lo, uint32_t hi) {
    size_t e = ((size_t) blockIdx.x * 256 + threadIdx.x) * 4;
    if (e + 4 > n) return;
    using U4 = __attribute__((ext_vector_type(4))) uint32_t;
    U4 p = __builtin_nontemporal_load(reinterpret_cast<const U4 *>(idx + e));
    float v[4];
    bool in[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { in[k] = p[k] >= lo && p[k] < hi; v[k] = in[k] ? table[p[k]] : 0.0f; }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (in[k]) out[e + k] = v[k];
}

extern "C" EK_API int ek_hip_probe_gather_sliced(int slices, float *out, const float *table, size_t table_size, const uint32_t *idx, size_t n) {
    if (int rc = ensure_init()) return rc;
    Context &cx = ctx();
    for (int s = 0; s < slices; ++s) {
        uint32_t lo = (uint32_t) (table_size * (size_t) s / slices), hi = (uint32_t) (table_size * (size_t) (s + 1) / slices);
        hipLaunchKernelGGL(k_probe_gather_range, dim3((unsigned) ((n / 4 + 255) / 256)), dim3(256), 0, cx.stream, out, table, idx, n, lo, hi);
    }
    EK_LAUNCH_CHECK("probe_gather_sliced", n, 12 * n);
    return EK_OK;
}

extern "C" EK_API int ek_hip_probe_gather(int elems, int policy, float *out, const float *table, const uint32_t *idx, size_t n) {
    if (int rc = ensure_init()) return rc;
    Context &cx = ctx();
#define EK_PG(E, P) hipLaunchKernelGGL((k_probe_gather<E, P>), dim3((unsigned) ((n / E + 255) / 256)), dim3(256), 0, cx.stream, out, table, idx, n)
    if (elems == 1 && policy == 0) EK_PG(1, 0); else if (elems == 1) EK_PG(1, 1);
    else if (elems == 4 && policy == 0) EK_PG(4, 0); else if (elems == 4 && policy == 1) EK_PG(4, 1);
    else if (elems == 4 && policy == 2) EK_PG(4, 2); else if (elems == 4 && policy == 3) EK_PG(4, 3);
    else if (elems == 4 && policy == 4) EK_PG(4, 4); else if (elems == 4 && policy == 5) EK_PG(4, 5);
    else if (elems == 8 && policy == 0) EK_PG(8, 0); else EK_PG(8, 1);
#undef EK_PG
    EK_LAUNCH_CHECK("probe_gather", n, 12 * n);
    return EK_OK;
}

// scatter_add experiment: `table` must hold 8 * table_size floats for mode 1 (copies zeroed by the caller);
// `fold_into` (table_size floats) receives table += sum of the copies when non-null.
extern "C" EK_API int ek_hip_probe_scatter_add(int mode, float *table, size_t table_size, float *fold_into,
                                               const float *val, const uint32_t *idx, size_t n) {
    if (int rc = ensure_init()) return rc;
    Context &cx = ctx();
    size_t nvec = n / 4;
    unsigned grid = (unsigned) ((nvec + 255) / 256);
    if (mode == 0)
        hipLaunchKernelGGL((k_probe_scatter_add<0>), dim3(grid), dim3(256), 0, cx.stream, table, table_size,
                           (const V4 *) val, (const uint4 *) idx, nvec);
    else
        hipLaunchKernelGGL((k_probe_scatter_add<1>), dim3(grid), dim3(256), 0, cx.stream, table, table_size,
                           (const V4 *) val, (const uint4 *) idx, nvec);
    if (fold_into)
        hipLaunchKernelGGL(k_probe_fold8, dim3((unsigned) ((table_size + 255) / 256)), dim3(256), 0, cx.stream, fold_into,
                           table, table_size);
    EK_LAUNCH_CHECK("probe_scatter_add", n, 8 * n);
    return EK_OK;
}

// Diagnostic entry point (f32 only): body 0 copy, 1 fmadd, 2 sincos, 3 read-only, 4 scale.
// blocks_per_cu <= 0 selects the "one shot" grid (every lane handles exactly `unroll` vectors).
extern "C" EK_API int ek_hip_probe(int body, int unroll, int nt_load, int nt_store, int blocks_per_cu, void *out0,
                                   void *out1, const void *a, const void *b, const void *c, size_t n) {
    if (int rc = ensure_init()) return rc;
    switch (body) {
        case 0: return probe_u<0>(unroll, nt_load, nt_store, blocks_per_cu, out0, out1, a, b, c, n);
        case 1: return probe_u<1>(unroll, nt_load, nt_store, blocks_per_cu, out0, out1, a, b, c, n);
        case 2: return probe_u<2>(unroll, nt_load, nt_store, blocks_per_cu, out0, out1, a, b, c, n);
        case 3: return probe_u<3>(unroll, nt_load, nt_store, blocks_per_cu, out0, out1, a, b, c, n);
        case 4: return probe_u<4>(unroll, nt_load, nt_store, blocks_per_cu, out0, out1, a, b, c, n);
        case 5: return probe_u<5>(unroll, nt_load, nt_store, blocks_per_cu, out0, out1, a, b, c, n);
        case 6: return probe_u<6>(unroll, nt_load, nt_store, blocks_per_cu, out0, out1, a, b, c, n);
        default: return fail(EK_ERR_INVALID, "ek_hip_probe(): unknown body %d", body);
    }
}

extern "C" EK_API int ek_hip_probe_gather_pair(int variant, float *o0, float *o1, const float *A, const float *B, const float *AB,
                                               const float *x, const float *s, const uint32_t *idx, size_t n) {
    if (int rc = ensure_init()) return rc;
    Context &cx = ctx();
    unsigned grid = (unsigned) ((n / 4 + 255) / 256);
#define EK_PGP(V) hipLaunchKernelGGL((k_probe_gather_pair<V>), dim3(grid), dim3(256), 0, cx.stream, o0, o1, A, B, (const V2 *) AB, x, s, idx, n)
    switch (variant) {
        case 0: EK_PGP(0); break;
        case 1: EK_PGP(1); break;
        case 2: EK_PGP(2); break;
        case 3: EK_PGP(3); break;
        case 4: EK_PGP(4); break;
        default: EK_PGP(5); break;
    }
#undef EK_PGP
    EK_LAUNCH_CHECK("probe_gather_pair", n, 0);
    return EK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
//  Partition experiment (round 2): 64-way split of (idx, c, x) into bucket-ordered 12-byte RECORDS {bucket-local index,
//  c * x, c} written with one dwordx3 store per element, against the product's SoA pair lists (2-byte + 4-byte + 4-byte
//  streams, values restaged per stream).  Bucket space is reserved per tile with one global atomic per bucket (the
//  caller hands in regions of `capacity` records per bucket), so there is no count / scan pass either.
// ---------------------------------------------------------------------------------------------------------------------
struct Rec3 { uint32_t key; float a, b; };

template <int Threads, int PerThread, int Shift>
__global__ __launch_bounds__(Threads) void k_probe_partition_aos(Rec3 *__restrict__ out, uint32_t *__restrict__ cursor,
                                                                  uint32_t capacity, const uint32_t *__restrict__ index,
                                                                  const float *__restrict__ c, const float *__restrict__ x,
                                                                  size_t n, size_t chunk) {
    constexpr int Tile = Threads * PerThread, Runs = PerThread / 4, Buckets = 64;
    __shared__ uint32_t hist[Buckets], off[Buckets], gbase[Buckets];
    __shared__ Rec3 stage[Tile];
    if (threadIdx.x < Buckets) hist[threadIdx.x] = 0;
    __syncthreads();
    const size_t begin = (size_t) blockIdx.x * chunk, end = begin + chunk < n ? begin + chunk : n;
    for (size_t base = begin; base + Tile <= end; base += Tile) {
        uint32_t ix[PerThread], rank[PerThread];
        float va[PerThread], vb[PerThread];
#pragma unroll
        for (int h = 0; h < Runs; ++h) {
            const size_t e = base + (size_t) h * (Tile / Runs) + (size_t) threadIdx.x * 4;
            Pack<uint32_t, 4> pi = pack_load<uint32_t, 4, true>(index + e);
            Pack<float, 4> pc = pack_load<float, 4, true>(c + e), px = pack_load<float, 4, true>(x + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) { ix[h * 4 + j] = pi.v[j]; vb[h * 4 + j] = pc.v[j]; va[h * 4 + j] = pc.v[j] * px.v[j]; }
        }
#pragma unroll
        for (int k = 0; k < PerThread; ++k) rank[k] = atomicAdd(&hist[ix[k] >> Shift], 1u);
        __syncthreads();
        if (threadIdx.x < 64) {
            const uint32_t h = hist[threadIdx.x];
            uint32_t incl = h;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t up = __shfl_up(incl, d, 64);
                if ((int) threadIdx.x >= d) incl += up;
            }
            off[threadIdx.x] = incl - h;
            gbase[threadIdx.x] = threadIdx.x * capacity + atomicAdd(&cursor[threadIdx.x], h);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PerThread; ++k) {
            const uint32_t b = ix[k] >> Shift;
            stage[off[b] + rank[k]] = Rec3{ ix[k], va[k], vb[k] };
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PerThread; ++k) {
            const uint32_t j = k * Threads + threadIdx.x;
            Rec3 r = stage[j];
            const uint32_t b = r.key >> Shift;
            r.key &= (1u << Shift) - 1u;
            out[gbase[b] + (j - off[b])] = r;
        }
        if (threadIdx.x < Buckets) hist[threadIdx.x] = 0;
        __syncthreads();
    }
}

extern "C" EK_API int ek_hip_probe_partition_aos(int variant, void *out, uint32_t *cursor, uint32_t capacity, const uint32_t *index,
                                                 const float *c, const float *x, size_t n) {
    if (int rc = ensure_init()) return rc;
    Context &cx = ctx();
    if (hipMemsetAsync(cursor, 0, 64 * sizeof(uint32_t), cx.stream) != hipSuccess) return fail(EK_ERR_HIP, "memset");
#define EK_PPA(T, P, BPC) {                                                                                            \
        constexpr int Tile = T * P;                                                                                     \
        unsigned blocks = (unsigned) std::min<size_t>((size_t) cx.num_cu * BPC, n / Tile);                              \
        size_t chunk = ((n / blocks) / Tile) * Tile;                                                                    \
        hipLaunchKernelGGL((k_probe_partition_aos<T, P, 14>), dim3(blocks), dim3(T), 0, cx.stream, (Rec3 *) out, cursor, \
                           capacity, index, c, x, chunk * blocks, chunk); }
    switch (variant) {
        case 0: EK_PPA(512, 8, 6) break;       // 4096-record tiles (48 KiB): 3 workgroups per CU
        case 1: EK_PPA(1024, 4, 6) break;      // same tile, 1024 threads
        case 2: EK_PPA(512, 16, 2) break;      // 8192-record tiles (96 KiB): 1 workgroup per CU
        case 3: EK_PPA(256, 8, 12) break;      // 2048-record tiles (24 KiB): 6 workgroups per CU
        default: EK_PPA(1024, 8, 2) break;     // 8192-record tiles, 1024 threads
    }
#undef EK_PPA
    EK_LAUNCH_CHECK("probe_partition_aos", n, n * 24);
    return EK_OK;
}

extern "C" EK_API int ek_hip_probe_interleave(float *AB, const float *A, const float *B, size_t k) {
    if (int rc = ensure_init()) return rc;
    hipLaunchKernelGGL(k_probe_interleave, dim3((unsigned) ((k + 255) / 256)), dim3(256), 0, ctx().stream, (V2 *) AB, A, B, k);
    EK_LAUNCH_CHECK("probe_interleave", k, 16 * k);
    return EK_OK;
}

extern "C" EK_API int ek_hip_probe_gather_pair_sliced(int slices, float *o0, const float *AB, size_t table_size, const float *x,
                                                      const uint32_t *idx, size_t n) {
    if (int rc = ensure_init()) return rc;
    Context &cx = ctx();
    for (int s = 0; s < slices; ++s) {
        uint32_t lo = (uint32_t) (table_size * (size_t) s / slices), hi = (uint32_t) (table_size * (size_t) (s + 1) / slices);
        hipLaunchKernelGGL(k_probe_gather_pair_sliced, dim3((unsigned) ((n / 4 + 255) / 256)), dim3(256), 0, cx.stream, o0,
                           (const V2 *) AB, x, idx, n, lo, hi);
    }
    EK_LAUNCH_CHECK("probe_gather_pair_sliced", n, 0);
    return EK_OK;
}

// ---- single-pass paged partition (ek_paged.h): validation + timing outside the product path ---------------------------
// (-DEK_PG_TIMING: per-phase cycle counters and wall-clock stamps of k_page_partition, read by tools/probe_paged.py)
#include "ek_paged.h"

// geometry[0..7] <- page_shift, cap, W, slots, chunk, page_slots, lds bytes, n_buckets
extern "C" EK_API int ek_hip_probe_page_plan(size_t n, size_t table_size, int shift, uint64_t *geometry) {
    if (int rc = ek::ensure_init()) return rc;
    const int n_buckets = (int) ((table_size + ((size_t) 1 << shift) - 1) >> shift);
    const ek::PagedPlan p = ek::paged_plan(n, n_buckets, ek::ctx().num_cu);
    geometry[0] = p.page_shift; geometry[1] = p.cap; geometry[2] = p.W; geometry[3] = p.slots; geometry[4] = p.chunk;
    geometry[5] = p.page_slots; geometry[6] = p.lds; geometry[7] = (uint64_t) n_buckets;
    return EK_OK;
}

// meta: uint32 words laid out as  counter block [kPgCounterWords = 4352] | base_full[257] | base_part[257] | piece_prefix[257] | cnt_full[nb*W] | loff[nb*W] | part[nb*W]
extern "C" EK_API int ek_hip_probe_page_partition(int nts, int index64, const void *index, const float *x, const uint8_t *mask,
                                                  size_t n, size_t table_size, int shift, uint32_t target_pieces, uint16_t *lp,
                                                  float *xp, uint32_t *wdir, uint32_t *wlist, uint32_t *glist_full,
                                                  uint32_t *glist_part, uint32_t *meta, int directory, unsigned long long *dbg) {
    using namespace ek;
    if (int rc = ensure_init()) return rc;
    Context &c = ctx();
    const int n_buckets = (int) ((table_size + ((size_t) 1 << shift) - 1) >> shift);
    if (n_buckets > kMaxBuckets) return fail(EK_ERR_INVALID, "too many buckets");
    const PagedPlan p = paged_plan(n, n_buckets, c.num_cu);
    PagedOut<float> out;
    out.lp = lp; out.xp = xp; out.wdir = wdir; out.wlist = wlist;
    out.gtotal = meta;
    out.active = meta + kPgMetaBase;
    out.lo = 0; out.span = (uint32_t) std::min<size_t>(table_size, 0xFFFFFFFFu);
    out.class_w = nullptr; out.class_stamp = nullptr; out.class_band = 0; out.wdir_lds = 0;             // (equal chunks: the probe measures the kernel, not the balancing)
#ifdef EK_PG_TIMING
    out.dbg = dbg;
#else
    (void) dbg;
#endif
    uint32_t *base_full = meta + kPgCounterWords, *base_part = base_full + kMaxBuckets + 1, *piece_prefix = base_part + kMaxBuckets + 1;
    out.cnt_full = piece_prefix + kMaxBuckets + 1;
    out.loff = out.cnt_full + (size_t) n_buckets * p.W;
    out.part = out.loff + (size_t) n_buckets * p.W;
    EK_HIP_CHECK(hipMemsetAsync(meta, 0, kPgCounterWords * sizeof(uint32_t), c.stream));
    const Arg<uint8_t> m{ mask, 1, mask ? 1u : 0u };
    const int vec_ok = aligned16(index) && aligned16(x) && (!mask || aligned16(mask));
#define EK_PP_LAUNCH(I, PS, HM)                                                                                                        \
    {                                                                                                                                  \
        EK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_page_partition<float, I, PS, HM>),                          \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int) p.lds));                                   \
        hipLaunchKernelGGL((k_page_partition<float, I, PS, HM>), dim3(p.W), dim3(kPgThreads), p.lds, c.stream, out, (const I *) index,  \
                           m, x, n, p.chunk, n_buckets, shift, p.cap, p.slots, vec_ok);                                               \
    }
    (void) nts;
    if (index64) {
        if (p.page_shift == 6) { if (mask) EK_PP_LAUNCH(uint64_t, 6, true) else EK_PP_LAUNCH(uint64_t, 6, false) }
        else { if (mask) EK_PP_LAUNCH(uint64_t, 5, true) else EK_PP_LAUNCH(uint64_t, 5, false) }
    } else {
        if (p.page_shift == 6) { if (mask) EK_PP_LAUNCH(uint32_t, 6, true) else EK_PP_LAUNCH(uint32_t, 6, false) }
        else { if (mask) EK_PP_LAUNCH(uint32_t, 5, true) else EK_PP_LAUNCH(uint32_t, 5, false) }
    }
#undef EK_PP_LAUNCH
    EK_LAUNCH_CHECK("probe_page_partition", n, n * 14);
    if (directory) {
        hipLaunchKernelGGL(k_page_directory, dim3(n_buckets, kPgDirSlices), dim3(256), 0, c.stream, glist_full, glist_part, base_full, base_part,
                           piece_prefix, out.gtotal, (const uint32_t *) out.cnt_full, (const uint32_t *) out.loff,
                           (const uint32_t *) out.part, (const uint32_t *) wlist, p.W, p.slots, n_buckets, target_pieces,
                           (uint32_t *) nullptr, (const uint32_t *) nullptr, 0u);
        EK_LAUNCH_CHECK("probe_page_directory", (size_t) n_buckets, 0);
    }
    return EK_OK;
}
