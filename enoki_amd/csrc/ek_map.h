// Streaming "map" kernels for CDNA4.
//
// Shape (measured with csrc/probe.hip on MI355X, see profiles/probe_r01.txt): 256-thread blocks
// (4 waves of 64), every lane moves ONE 16-byte vector per array (U = 1), and the grid covers the
// array exactly once ("one shot": ceil(nvec / 256) blocks, 65536 blocks for 64 Mi floats) instead
// of a capped grid-stride loop -- retiring waves overlap their stores with the next waves' loads
// and the dispatcher keeps every CU full.  Loads and stores use the non-temporal cache policy
// (`nt`): the data is touched once per kernel, so it should not displace the few reused lines
// (gather tables) in L2.  On the probe this moved a 3-input/1-output body from 4.9 to 6.7 TB/s.
//
// All vertical ops of the array backend are instances of these
// templates; size-1 operands and immediates are broadcast from a register (never materialised),
// mirroring how the reference's JIT passes scalars as `ldu` loads (src/cuda/jit.cu:1131-1136).
#pragma once

#include "ek_internal.h"

// Cache policy, measured two ways on the MI355X (profiles/):
//   * single kernels re-reading the same buffers (tools/probe_bw.py --sweep) prefer plain loads for
//     one-input bodies -- but that is the 256 MiB Infinity Cache serving repeats of the same array;
//   * the same kernels chained as a producer -> consumer PIPELINE like the real tape
//     (tools/probe_pipeline.py, cfg3a emulation) are fastest with non-temporal loads AND stores and one
//     vector per lane everywhere: 0.6035 ms / 83.4 % of 8 TB/s, vs 0.628 ms for the per-kernel optimum,
//     0.664 ms for plain loads + nt stores and 0.681 ms without nt.
// The pipeline is what the backend executes, so: U = 1, nt loads, nt stores.
namespace ek {

template <typename T, int N> struct alignas(sizeof(T) * N) Pack { T v[N]; };

template <typename... Ts> struct max_size;
template <typename T> struct max_size<T> { static constexpr size_t value = sizeof(T); };
template <typename T, typename... Ts> struct max_size<T, Ts...> {
    static constexpr size_t rest = max_size<Ts...>::value;
    static constexpr size_t value = sizeof(T) > rest ? sizeof(T) : rest;
};

template <int Bytes> struct raw_vec;
template <> struct raw_vec<16> { using type = __attribute__((ext_vector_type(4))) uint32_t; };
template <> struct raw_vec<8> { using type = __attribute__((ext_vector_type(2))) uint32_t; };
template <> struct raw_vec<4> { using type = uint32_t; };
template <> struct raw_vec<2> { using type = uint16_t; };
template <> struct raw_vec<1> { using type = uint8_t; };

template <typename T, int N, bool NT> __device__ __forceinline__ Pack<T, N> pack_load(const T *p) {
    using R = typename raw_vec<sizeof(T) * N>::type;
    R raw;
    if constexpr (NT) raw = __builtin_nontemporal_load(reinterpret_cast<const R *>(p));
    else raw = *reinterpret_cast<const R *>(p);
    Pack<T, N> r;
    __builtin_memcpy(&r, &raw, sizeof(r));
    return r;
}

template <typename T, int N, bool NT> __device__ __forceinline__ void pack_store(T *p, const Pack<T, N> &v) {
    using R = typename raw_vec<sizeof(T) * N>::type;
    R raw;
    __builtin_memcpy(&raw, &v, sizeof(v));
    if constexpr (NT) __builtin_nontemporal_store(raw, reinterpret_cast<R *>(p));
    else *reinterpret_cast<R *>(p) = raw;
}

template <typename T> __device__ __forceinline__ T arg_scalar(const Arg<T> &a) {
    return a.ptr ? a.ptr[0] : a.imm;
}

// Elements [e, e + N) of operand `a`: one vector load when `fast`, guarded scalar loads otherwise
// (tail of the array or pointers that are not 16-byte aligned), broadcast for size-1 operands.
template <typename T, int N, bool NT>
__device__ __forceinline__ Pack<T, N> arg_load(const Arg<T> &a, T s, size_t e, size_t n, bool fast) {
    Pack<T, N> p;
    if (!a.vec) {
#pragma unroll
        for (int i = 0; i < N; ++i) p.v[i] = s;
    } else if (fast) {
        p = pack_load<T, N, NT>(a.ptr + e);
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) p.v[i] = (e + i < n) ? a.ptr[e + i] : T(0);
    }
    return p;
}

template <typename T, int N, bool NT>
__device__ __forceinline__ void out_store(T *out, const Pack<T, N> &p, size_t e, size_t n, bool fast) {
    if (fast) {
        pack_store<T, N, NT>(out + e, p);
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (e + i < n) out[e + i] = p.v[i];
    }
}

// first element handled by this lane for unroll slot k
template <int N, int U> __device__ __forceinline__ size_t lane_elem(int k) {
    return ((size_t) blockIdx.x * (256 * U) + (size_t) k * 256 + threadIdx.x) * N;
}

// ---- arity 1 ----------------------------------------------------------------------------------
template <typename F, typename TO, typename TA, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_map1(TO *__restrict__ out, size_t n, int vec_ok, Arg<TA> a) {
    constexpr int N = 16 / max_size<TO, TA>::value;
    const TA sa = a.vec ? TA(0) : arg_scalar(a);
    Pack<TA, N> pa[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
        size_t e = lane_elem<N, U>(k);
        if (e < n) pa[k] = arg_load<TA, N, NTL>(a, sa, e, n, vec_ok && e + N <= n);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
        size_t e = lane_elem<N, U>(k);
        if (e < n) {
            Pack<TO, N> po;
#pragma unroll
            for (int i = 0; i < N; ++i) po.v[i] = F::apply(pa[k].v[i]);
            out_store<TO, N, NTS>(out, po, e, n, vec_ok && e + N <= n);
        }
    }
}

// one input, two outputs (sincos)
template <typename F, typename TO, typename TA, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_map1x2(TO *__restrict__ out0, TO *__restrict__ out1, size_t n, int vec_ok,
                                                Arg<TA> a) {
    constexpr int N = 16 / max_size<TO, TA>::value;
    const TA sa = a.vec ? TA(0) : arg_scalar(a);
    Pack<TA, N> pa[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
        size_t e = lane_elem<N, U>(k);
        if (e < n) pa[k] = arg_load<TA, N, NTL>(a, sa, e, n, vec_ok && e + N <= n);
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
        size_t e = lane_elem<N, U>(k);
        if (e < n) {
            Pack<TO, N> p0, p1;
#pragma unroll
            for (int i = 0; i < N; ++i) F::apply(pa[k].v[i], p0.v[i], p1.v[i]);
            bool fast = vec_ok && e + N <= n;
            out_store<TO, N, NTS>(out0, p0, e, n, fast);
            out_store<TO, N, NTS>(out1, p1, e, n, fast);
        }
    }
}

// ---- arity 2 ----------------------------------------------------------------------------------
template <typename F, typename TO, typename TA, typename TB, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_map2(TO *__restrict__ out, size_t n, int vec_ok, Arg<TA> a, Arg<TB> b) {
    constexpr int N = 16 / max_size<TO, TA, TB>::value;
    const TA sa = a.vec ? TA(0) : arg_scalar(a);
    const TB sb = b.vec ? TB(0) : arg_scalar(b);
    Pack<TA, N> pa[U];
    Pack<TB, N> pb[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
        size_t e = lane_elem<N, U>(k);
        if (e < n) {
            bool fast = vec_ok && e + N <= n;
            pa[k] = arg_load<TA, N, NTL>(a, sa, e, n, fast);
            pb[k] = arg_load<TB, N, NTL>(b, sb, e, n, fast);
        }
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
        size_t e = lane_elem<N, U>(k);
        if (e < n) {
            Pack<TO, N> po;
#pragma unroll
            for (int i = 0; i < N; ++i) po.v[i] = F::apply(pa[k].v[i], pb[k].v[i]);
            out_store<TO, N, NTS>(out, po, e, n, vec_ok && e + N <= n);
        }
    }
}

// ---- arity 3 ----------------------------------------------------------------------------------
template <typename F, typename TO, typename TA, typename TB, typename TC, int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_map3(TO *__restrict__ out, size_t n, int vec_ok, Arg<TA> a, Arg<TB> b,
                                              Arg<TC> c) {
    constexpr int N = 16 / max_size<TO, TA, TB, TC>::value;
    const TA sa = a.vec ? TA(0) : arg_scalar(a);
    const TB sb = b.vec ? TB(0) : arg_scalar(b);
    const TC sc = c.vec ? TC(0) : arg_scalar(c);
    Pack<TA, N> pa[U];
    Pack<TB, N> pb[U];
    Pack<TC, N> pc[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
        size_t e = lane_elem<N, U>(k);
        if (e < n) {
            bool fast = vec_ok && e + N <= n;
            pa[k] = arg_load<TA, N, NTL>(a, sa, e, n, fast);
            pb[k] = arg_load<TB, N, NTL>(b, sb, e, n, fast);
            pc[k] = arg_load<TC, N, NTL>(c, sc, e, n, fast);
        }
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
        size_t e = lane_elem<N, U>(k);
        if (e < n) {
            Pack<TO, N> po;
#pragma unroll
            for (int i = 0; i < N; ++i) po.v[i] = F::apply(pa[k].v[i], pb[k].v[i], pc[k].v[i]);
            out_store<TO, N, NTS>(out, po, e, n, vec_ok && e + N <= n);
        }
    }
}

// ---- host-side launchers -----------------------------------------------------------------------
template <int N, int U> inline unsigned oneshot_grid(size_t n) {
    size_t per_block = (size_t) 256 * U * N;
    size_t blocks = (n + per_block - 1) / per_block;
    return (unsigned) (blocks ? blocks : 1);
}

/// Functors may declare `static constexpr bool heavy = true` (transcendental bodies)
template <typename F, typename = void> struct is_heavy : std::false_type { };
template <typename F> struct is_heavy<F, std::void_t<decltype(F::heavy)>> : std::bool_constant<F::heavy> { };

#define EK_MAP_LAUNCH(KERNEL, TYPES, N_, U_, NTL_, NTS_, n_, ...)                                 \
    hipLaunchKernelGGL((KERNEL<TYPES, U_, NTL_, NTS_>), dim3(oneshot_grid<N_, U_>(n_)), dim3(256), 0, \
                       ctx().stream, __VA_ARGS__)

#define EK_COMMA ,

template <typename F, typename TO, typename TA>
int launch_map1(const char *name, TO *out, size_t n, const Arg<TA> &a) {
    constexpr int N = 16 / max_size<TO, TA>::value;
    int vec_ok = aligned16(out) && arg_aligned(a);
    EK_MAP_LAUNCH(k_map1, F EK_COMMA TO EK_COMMA TA, N, 1, true, true, n, out, n, vec_ok, a);
    EK_LAUNCH_CHECK(name, n, n * sizeof(TO) + arg_bytes(a, n));
    return EK_OK;
}

template <typename F, typename TO, typename TA>
int launch_map1x2(const char *name, TO *out0, TO *out1, size_t n, const Arg<TA> &a) {
    constexpr int N = 16 / max_size<TO, TA>::value;
    int vec_ok = aligned16(out0) && aligned16(out1) && arg_aligned(a);
    EK_MAP_LAUNCH(k_map1x2, F EK_COMMA TO EK_COMMA TA, N, 1, true, true, n, out0, out1, n, vec_ok, a);
    EK_LAUNCH_CHECK(name, n, 2 * n * sizeof(TO) + arg_bytes(a, n));
    return EK_OK;
}

template <typename F, typename TO, typename TA, typename TB>
int launch_map2(const char *name, TO *out, size_t n, const Arg<TA> &a, const Arg<TB> &b) {
    constexpr int N = 16 / max_size<TO, TA, TB>::value;
    int vec_ok = aligned16(out) && arg_aligned(a) && arg_aligned(b);
    EK_MAP_LAUNCH(k_map2, F EK_COMMA TO EK_COMMA TA EK_COMMA TB, N, 1, true, true, n, out, n, vec_ok, a, b);
    // algorithmic bytes = DISTINCT input arrays + output (x * x reads x once)
    const bool b_dup = b.vec && a.vec && (const void *) b.ptr == (const void *) a.ptr;
    EK_LAUNCH_CHECK(name, n, n * sizeof(TO) + arg_bytes(a, n) + (b_dup ? 0 : arg_bytes(b, n)));
    return EK_OK;
}

template <typename F, typename TO, typename TA, typename TB, typename TC>
int launch_map3(const char *name, TO *out, size_t n, const Arg<TA> &a, const Arg<TB> &b, const Arg<TC> &c) {
    constexpr int N = 16 / max_size<TO, TA, TB, TC>::value;
    int vec_ok = aligned16(out) && arg_aligned(a) && arg_aligned(b) && arg_aligned(c);
    EK_MAP_LAUNCH(k_map3, F EK_COMMA TO EK_COMMA TA EK_COMMA TB EK_COMMA TC, N, 1, true, true, n, out, n, vec_ok, a, b, c);
    const bool b_dup = b.vec && a.vec && (const void *) b.ptr == (const void *) a.ptr;
    const bool c_dup = c.vec && ((a.vec && (const void *) c.ptr == (const void *) a.ptr) ||
                                 (b.vec && (const void *) c.ptr == (const void *) b.ptr));
    EK_LAUNCH_CHECK(name, n, n * sizeof(TO) + arg_bytes(a, n) + (b_dup ? 0 : arg_bytes(b, n)) +
                             (c_dup ? 0 : arg_bytes(c, n)));
    return EK_OK;
}

} // namespace ek
