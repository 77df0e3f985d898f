// Horizontal reductions for CDNA4: hsum / hprod / hmin / hmax, mask all / any / count, the fused
// backward reduction hsum(safe_mul(w, g)), and the inclusive prefix sum.  SURVEY.md rows a8, a11.
//
// Replaces the reference's CUB calls (src/cuda/horiz.cu:162-354).  Structure (hand-written):
//   stage 1  grid of (reduce_blocks_per_cu x #CU) blocks of 256 threads; every lane streams
//            16-byte vectors (4 in flight), keeps 4 independent accumulators, then
//            wave64 butterfly with __shfl_down (6 steps) -> 4 wave partials in LDS -> 1 block partial;
//   stage 2  one 256-thread block folds the <= 2048 block partials the same way.
// No floating point atomics: the result is deterministic for a given (n, grid), i.e. run-to-run
// reproducible, but the summation ORDER differs from the CPU's lane-wise packet accumulation
// (dynamic.h:632-702), so fp results agree to the order-dependent bound documented in tests/.
#include "ek_unary.h"

#include <algorithm>
#include <limits>

namespace ek {

template <int Op, typename T> struct Reducer {
    static __device__ __host__ __forceinline__ T identity() {
        if constexpr (Op == EK_HSUM) return T(0);
        else if constexpr (Op == EK_HPROD) return T(1);
        else if constexpr (std::is_floating_point_v<T>) return std::numeric_limits<T>::quiet_NaN();   // minNum / maxNum: see combine()
        else if constexpr (Op == EK_HMIN) return std::numeric_limits<T>::max();
        else return std::numeric_limits<T>::lowest();
    }
    static __device__ __forceinline__ T combine(T acc, T v) {
        using U = wrap_t<T>;
        if constexpr (Op == EK_HSUM) return (T) ((U) acc + (U) v);
        else if constexpr (Op == EK_HPROD) return (T) ((U) acc * (U) v);
        else if constexpr (std::is_floating_point_v<T>) {
            // Floating point hmin / hmax are IEEE minNum / maxNum reductions (v_min_f32 / v_max_f32): NaN entries are
            // ignored unless every entry is NaN, and -0 < +0 -- well defined and independent of the reduction order.
            // The reference's AVX path is NOT: MINPS returns its second operand on unordered or equal compares, so
            // whether a NaN (or which zero) survives depends on its position modulo the packet width
            // (dynamic.h:669-702); on NaN-free data without mixed zeros both agree bit for bit.
            if constexpr (sizeof(T) == 4) return Op == EK_HMIN ? __builtin_fminf(acc, v) : __builtin_fmaxf(acc, v);
            else return Op == EK_HMIN ? __builtin_fmin(acc, v) : __builtin_fmax(acc, v);
        }
        else if constexpr (Op == EK_HMIN) return v < acc ? v : acc;
        else return v > acc ? v : acc;
    }
};

template <typename T> __device__ __forceinline__ T shfl_down(T v, int delta) {
    if constexpr (sizeof(T) == 8) {
        uint64_t u;
        __builtin_memcpy(&u, &v, 8);
        uint32_t lo = (uint32_t) u, hi = (uint32_t) (u >> 32);
        lo = __shfl_down(lo, delta, 64);
        hi = __shfl_down(hi, delta, 64);
        u = ((uint64_t) hi << 32) | lo;
        __builtin_memcpy(&v, &u, 8);
        return v;
    } else {
        return __shfl_down(v, delta, 64);
    }
}

// 256 threads -> one value in thread 0
template <typename R, typename T> __device__ __forceinline__ T block_reduce(T v) {
    __shared__ T wave_part[4];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        v = R::combine(v, shfl_down(v, d));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_part[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0)
        v = R::combine(R::combine(wave_part[0], wave_part[1]), R::combine(wave_part[2], wave_part[3]));
    return v;
}

// Loaders turn (index) -> value; they let the same reduction serve plain arrays, u8 masks and the
// fused hsum(safe_mul(w, g)) of Tape::backward (autodiff.cpp:867-871).
// N = elements consumed per 16-byte access, M = values handed to the reducer per access.
template <typename T> struct PlainLoader {
    const T *ptr;
    static constexpr int N = 16 / sizeof(T), M = N;
    __device__ __forceinline__ void init() { }
    __device__ __forceinline__ Pack<T, N> load_pack(size_t v) const { return pack_load<T, N, true>(ptr + v * N); }
    __device__ __forceinline__ T load(size_t i) const { return ptr[i]; }
};

struct MaskCountLoader {   // u8 mask -> u64 count, 16 mask bytes per lane per access
    const uint8_t *ptr;
    static constexpr int N = 16, M = 1;
    __device__ __forceinline__ void init() { }
    __device__ __forceinline__ Pack<uint64_t, 1> load_pack(size_t v) const {
        Pack<uint8_t, 16> raw = pack_load<uint8_t, 16, true>(ptr + v * 16);
        Pack<uint64_t, 1> r;
        unsigned cnt = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) cnt += raw.v[i] ? 1u : 0u;
        r.v[0] = cnt;
        return r;
    }
    __device__ __forceinline__ uint64_t load(size_t i) const { return ptr[i] ? 1u : 0u; }
};

template <typename T> struct SafeMulLoader {
    Arg<T> w, g;
    T sw, sg;
    static constexpr int N = 16 / sizeof(T), M = N;
    __device__ __forceinline__ void init() {
        sw = w.vec ? T(0) : arg_scalar(w);
        sg = g.vec ? T(0) : arg_scalar(g);
    }
    __device__ __forceinline__ Pack<T, N> load_pack(size_t v) const {
        Pack<T, N> pw = arg_load<T, N, true>(w, sw, v * N, (size_t) -1, true),
                   pg = arg_load<T, N, true>(g, sg, v * N, (size_t) -1, true), r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = dev::safe_mul(pw.v[i], pg.v[i]);
        return r;
    }
    __device__ __forceinline__ T load(size_t i) const {
        return dev::safe_mul(w.vec ? w.ptr[i] : sw, g.vec ? g.ptr[i] : sg);
    }
};

// A unary operation applied while loading: hsum(sin(x)) reads x once and writes nothing (HIPArray leaves the result of a
// fusable unary op unevaluated until its first consumer, include/enoki/hip.h); same values, same reduction tree as
// the two-kernel version.
template <int Map, typename T> struct MapLoader {
    const T *ptr;
    static constexpr int N = 16 / sizeof(T), M = N;
    __device__ __forceinline__ void init() { }
    __device__ __forceinline__ Pack<T, N> load_pack(size_t v) const {
        Pack<T, N> p = pack_load<T, N, true>(ptr + v * N);
#pragma unroll
        for (int i = 0; i < N; ++i) p.v[i] = UnaryOp<Map, T>::apply(p.v[i]);
        return p;
    }
    __device__ __forceinline__ T load(size_t i) const { return UnaryOp<Map, T>::apply(ptr[i]); }
};

template <typename R, typename T, typename Loader>
__global__ __launch_bounds__(256) void k_reduce_stage1(T *__restrict__ partials, size_t n, int vec_ok, Loader ld) {
    constexpr int N = Loader::N, M = Loader::M, U = 4;
    ld.init();
    const size_t gid = (size_t) blockIdx.x * 256 + threadIdx.x, total = (size_t) gridDim.x * 256;
    T acc[U];
#pragma unroll
    for (int k = 0; k < U; ++k) acc[k] = R::identity();
    size_t done = 0;
    if (vec_ok) {
        const size_t nvec = n / N;
        for (size_t v0 = gid; v0 < nvec; v0 += total * U) {
            Pack<T, M> p[U];
#pragma unroll
            for (int k = 0; k < U; ++k)
                if (v0 + k * total < nvec) p[k] = ld.load_pack(v0 + k * total);
#pragma unroll
            for (int k = 0; k < U; ++k) {
                if (v0 + k * total < nvec) {
#pragma unroll
                    for (int i = 0; i < M; ++i) acc[k] = R::combine(acc[k], p[k].v[i]);
                }
            }
        }
        done = nvec * N;
    }
    for (size_t i = done + gid; i < n; i += total)
        acc[0] = R::combine(acc[0], ld.load(i));
    T v = R::combine(R::combine(acc[0], acc[1]), R::combine(acc[2], acc[3]));
    v = block_reduce<R>(v);
    if (threadIdx.x == 0) partials[blockIdx.x] = v;
}

template <typename R, typename T>
__global__ __launch_bounds__(256) void k_reduce_stage2(T *__restrict__ out, const T *__restrict__ partials, unsigned count) {
    T v = R::identity();
    for (unsigned i = threadIdx.x; i < count; i += 256)
        v = R::combine(v, partials[i]);
    v = block_reduce<R>(v);
    if (threadIdx.x == 0) out[0] = v;
}

template <typename R, typename T, typename Loader>
int reduce_launch(const char *name, T *out, size_t n, int vec_ok, const Loader &ld, size_t bytes) {
    RoctxRange range("enoki-hip: horizontal reduction");
    Context &c = ctx();
    constexpr int N = Loader::N;
    size_t items = (n / N + 3) / 4 + 1;
    unsigned grid = stream_grid(items, c.tuning.reduce_blocks_per_cu);
    if (grid > 2048) grid = 2048;
    void *scratch = nullptr;
    if (int rc = reduce_scratch((size_t) grid * sizeof(T), &scratch)) return rc;
    hipLaunchKernelGGL((k_reduce_stage1<R, T, Loader>), dim3(grid), dim3(256), 0, c.stream, (T *) scratch, n, vec_ok, ld);
    EK_LAUNCH_CHECK(name, n, bytes);
    hipLaunchKernelGGL((k_reduce_stage2<R, T>), dim3(1), dim3(256), 0, c.stream, out, (const T *) scratch, grid);
    EK_LAUNCH_CHECK("reduce_stage2", (size_t) grid, (size_t) grid * sizeof(T) + sizeof(T));
    return EK_OK;
}

template <int Op, typename T> int reduce_typed(void *out, const void *in, size_t n) {
    PlainLoader<T> ld{ (const T *) in };
    return reduce_launch<Reducer<Op, T>>(Op == EK_HSUM ? "hsum" : Op == EK_HPROD ? "hprod" : Op == EK_HMIN ? "hmin" : "hmax",
                                         (T *) out, n, aligned16(in), ld, n * sizeof(T));
}

template <typename T> int reduce_dispatch(int op, void *out, const void *in, size_t n) {
    switch (op) {
        case EK_HSUM: return reduce_typed<EK_HSUM, T>(out, in, n);
        case EK_HPROD: return reduce_typed<EK_HPROD, T>(out, in, n);
        case EK_HMIN: return reduce_typed<EK_HMIN, T>(out, in, n);
        case EK_HMAX: return reduce_typed<EK_HMAX, T>(out, in, n);
        default: return fail(EK_ERR_INVALID, "ek_hip_reduce(): unknown op %d", op);
    }
}

template <int Op, int Map, typename T> int reduce_map_typed(void *out, const void *in, size_t n) {
    MapLoader<Map, T> ld{ (const T *) in };
    return reduce_launch<Reducer<Op, T>>(Op == EK_HSUM ? "hsum_map" : Op == EK_HPROD ? "hprod_map" : Op == EK_HMIN ? "hmin_map" : "hmax_map",
                                         (T *) out, n, aligned16(in), ld, n * sizeof(T));
}

template <int Op, typename T> int reduce_map_select(int map, void *out, const void *in, size_t n) {
    switch (map) {
        case EK_NEG: return reduce_map_typed<Op, EK_NEG, T>(out, in, n);
        case EK_ABS: return reduce_map_typed<Op, EK_ABS, T>(out, in, n);
        case EK_SQRT: return reduce_map_typed<Op, EK_SQRT, T>(out, in, n);
        case EK_RCP: return reduce_map_typed<Op, EK_RCP, T>(out, in, n);
        case EK_RSQRT: return reduce_map_typed<Op, EK_RSQRT, T>(out, in, n);
        case EK_SIN: return reduce_map_typed<Op, EK_SIN, T>(out, in, n);
        case EK_COS: return reduce_map_typed<Op, EK_COS, T>(out, in, n);
        case EK_EXP: return reduce_map_typed<Op, EK_EXP, T>(out, in, n);
        case EK_LOG: return reduce_map_typed<Op, EK_LOG, T>(out, in, n);
        case EK_RCP_SQR: return reduce_map_typed<Op, EK_RCP_SQR, T>(out, in, n);
        case EK_RSQRT_SQR: return reduce_map_typed<Op, EK_RSQRT_SQR, T>(out, in, n);
        case EK_RSQRT_CUBE: return reduce_map_typed<Op, EK_RSQRT_CUBE, T>(out, in, n);
        case EK_TAN: return reduce_map_typed<Op, EK_TAN, T>(out, in, n);
        case EK_TANH: return reduce_map_typed<Op, EK_TANH, T>(out, in, n);
        case EK_ATAN: return reduce_map_typed<Op, EK_ATAN, T>(out, in, n);
        case EK_SINH: return reduce_map_typed<Op, EK_SINH, T>(out, in, n);
        case EK_COSH: return reduce_map_typed<Op, EK_COSH, T>(out, in, n);
        case EK_SEC_SQR: return reduce_map_typed<Op, EK_SEC_SQR, T>(out, in, n);
        case EK_SECH_SQR: return reduce_map_typed<Op, EK_SECH_SQR, T>(out, in, n);
        case EK_RCP_1P_SQR: return reduce_map_typed<Op, EK_RCP_1P_SQR, T>(out, in, n);
        default: return fail(EK_ERR_UNSUPPORTED, "ek_hip_reduce_map(): op %d cannot be applied on load", map);
    }
}

template <typename T> int reduce_map_dispatch(int op, int map, void *out, const void *in, size_t n) {
    switch (op) {
        case EK_HSUM: return reduce_map_select<EK_HSUM, T>(map, out, in, n);
        case EK_HPROD: return reduce_map_select<EK_HPROD, T>(map, out, in, n);
        case EK_HMIN: return reduce_map_select<EK_HMIN, T>(map, out, in, n);
        case EK_HMAX: return reduce_map_select<EK_HMAX, T>(map, out, in, n);
        default: return fail(EK_ERR_INVALID, "ek_hip_reduce_map(): unknown op %d", op);
    }
}

// ---- chains: base(src0, src1, src2) under up to three unary maps, applied while the operands are loaded -------------------
// What the reference's JIT does with hsum(sin(exp(fmadd(a, x, b)))) -- BASELINE configs[1] -- is ONE kernel that reads a, x, b and
// writes a partial sum per block (src/cuda/jit.cu:1066-1217 assemble, :1418-1508 eval): 12 B/elt.  Run op by op the same
// expression moves 36 B/elt, with the reduction applying its last map on load 28.  HIPArray leaves the fma AND the maps on top
// of it unevaluated (include/enoki/hip.h, kind 4 / kind 1 nodes); the reduction that finally consumes the chain gets it as a
// DESCRIPTOR -- a base op over at most three operands and at most three unary op codes -- not as a kernel per combination:
// a lane holds one 16-byte vector of every operand, and each stage is a wave-uniform switch AROUND the loop over the
// vector's elements, so every op body exists once per kernel and the switch costs a scalar branch per stage and vector.
// Same functors as the one-op kernels (UnaryOp / TernaryOp bodies): the values are bit-identical to the op-by-op evaluation.
enum { CH_COPY = 0, CH_ADD, CH_SUB, CH_MUL, CH_TERNARY };     // + ek_ternary_op

template <typename T> struct ChainArgs {
    Arg<T> src[3];
    int base, n_maps;
    int map_ops[3];
};

template <typename T, int E>
__device__ __forceinline__ void chain_apply(T (&r)[E], const T (&b)[E], const T (&c)[E], const ChainArgs<T> &ch) {
    auto fma_ = [](T x, T y, T z) -> T {
        if constexpr (sizeof(T) == 4) return __builtin_fmaf(x, y, z); else return __builtin_fma(x, y, z);
    };
#define EK_CH_BASE(CODE, EXPR) case CODE: _Pragma("unroll") for (int e = 0; e < E; ++e) r[e] = (EXPR); break;
    switch (ch.base) {
        EK_CH_BASE(CH_ADD, r[e] + b[e]) EK_CH_BASE(CH_SUB, r[e] - b[e]) EK_CH_BASE(CH_MUL, r[e] * b[e])
        EK_CH_BASE(CH_TERNARY + EK_FMADD, fma_(r[e], b[e], c[e])) EK_CH_BASE(CH_TERNARY + EK_FMSUB, fma_(r[e], b[e], -c[e]))
        EK_CH_BASE(CH_TERNARY + EK_FNMADD, fma_(-r[e], b[e], c[e])) EK_CH_BASE(CH_TERNARY + EK_FNMSUB, fma_(-r[e], b[e], -c[e]))
        EK_CH_BASE(CH_TERNARY + EK_MULADD, r[e] * b[e] + c[e]) EK_CH_BASE(CH_TERNARY + EK_MULSUB, r[e] * b[e] - c[e])
        EK_CH_BASE(CH_TERNARY + EK_NMULADD, c[e] - r[e] * b[e])
        default: break;                                                     // CH_COPY
    }
#undef EK_CH_BASE
#define EK_CH_MAP(OP) case OP: _Pragma("unroll") for (int e = 0; e < E; ++e) r[e] = UnaryOp<OP, T>::apply(r[e]); break;
    for (int s = 0; s < ch.n_maps; ++s) {
        switch (ch.map_ops[s]) {
            EK_CH_MAP(EK_NEG) EK_CH_MAP(EK_ABS) EK_CH_MAP(EK_SQRT) EK_CH_MAP(EK_RCP) EK_CH_MAP(EK_RSQRT) EK_CH_MAP(EK_SIN)
            EK_CH_MAP(EK_COS) EK_CH_MAP(EK_EXP) EK_CH_MAP(EK_LOG) EK_CH_MAP(EK_RCP_SQR) EK_CH_MAP(EK_RSQRT_SQR) EK_CH_MAP(EK_RSQRT_CUBE)
#ifndef EK_CHAIN_FIRST_WAVE_ONLY      /* (measurement builds: what the second-wave cases cost the kernels that do not use them) */
            EK_CH_MAP(EK_TAN) EK_CH_MAP(EK_TANH) EK_CH_MAP(EK_ATAN) EK_CH_MAP(EK_SINH) EK_CH_MAP(EK_COSH)
            EK_CH_MAP(EK_SEC_SQR) EK_CH_MAP(EK_SECH_SQR) EK_CH_MAP(EK_RCP_1P_SQR)
#endif
            default: break;
        }
    }
#undef EK_CH_MAP
}

template <typename T, int N>
__device__ __forceinline__ void chain_load(const ChainArgs<T> &ch, const T (&s)[3], size_t e, size_t n, bool fast, T (&r)[N], T (&b)[N], T (&c)[N]) {
    const Pack<T, N> p0 = arg_load<T, N, true>(ch.src[0], s[0], e, n, fast), p1 = arg_load<T, N, true>(ch.src[1], s[1], e, n, fast),
                     p2 = arg_load<T, N, true>(ch.src[2], s[2], e, n, fast);
#pragma unroll
    for (int i = 0; i < N; ++i) { r[i] = p0.v[i]; b[i] = p1.v[i]; c[i] = p2.v[i]; }
}

// The reduction TREE is that of k_reduce_stage1 over a plain array, slot for slot (same grid, four vectors per lane and trip, one
// accumulator per vector slot, the same tail and the same combine order): the result is bit-identical to evaluating the chain
// op by op and reducing the array -- deferred evaluation never changes a bit (tests/cpp fuzz programs compare exactly that).
template <typename R, typename T>
__global__ __launch_bounds__(256) void k_chain_reduce(T *__restrict__ partials, size_t n, int vec_ok, ChainArgs<T> ch) {
    constexpr int N = 16 / sizeof(T), U = 4, E = N * U;
    T s[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) s[k] = ch.src[k].vec ? T(0) : arg_scalar(ch.src[k]);
    const size_t gid = (size_t) blockIdx.x * 256 + threadIdx.x, total = (size_t) gridDim.x * 256;
    T acc[U];
#pragma unroll
    for (int k = 0; k < U; ++k) acc[k] = R::identity();
    size_t done = 0;
    if (vec_ok) {
        const size_t nvec = n / N;
        for (size_t v0 = gid; v0 < nvec; v0 += total * U) {
            T r[E], b[E], c[E];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                // (a slot beyond the end reads the first vector again: its values are computed and not used)
                const size_t v = v0 + k * total < nvec ? v0 + k * total : 0;
                const Pack<T, N> p0 = arg_load<T, N, true>(ch.src[0], s[0], v * N, n, true), p1 = arg_load<T, N, true>(ch.src[1], s[1], v * N, n, true),
                                 p2 = arg_load<T, N, true>(ch.src[2], s[2], v * N, n, true);
#pragma unroll
                for (int i = 0; i < N; ++i) { r[k * N + i] = p0.v[i]; b[k * N + i] = p1.v[i]; c[k * N + i] = p2.v[i]; }
            }
            chain_apply<T, E>(r, b, c, ch);
#pragma unroll
            for (int k = 0; k < U; ++k) {
                if (v0 + k * total < nvec) {
#pragma unroll
                    for (int i = 0; i < N; ++i) acc[k] = R::combine(acc[k], r[k * N + i]);
                }
            }
        }
        done = nvec * N;
    }
    for (size_t i = done + gid; i < n; i += total) {
        T r[1], b[1], c[1];
        r[0] = ch.src[0].vec ? ch.src[0].ptr[i] : s[0];
        b[0] = ch.src[1].vec ? ch.src[1].ptr[i] : s[1];
        c[0] = ch.src[2].vec ? ch.src[2].ptr[i] : s[2];
        chain_apply<T, 1>(r, b, c, ch);
        acc[0] = R::combine(acc[0], r[0]);
    }
    T v = R::combine(R::combine(acc[0], acc[1]), R::combine(acc[2], acc[3]));
    v = block_reduce<R>(v);
    if (threadIdx.x == 0) partials[blockIdx.x] = v;
}

// the same chain written out (what forcing an unevaluated chain costs: one pass instead of one per op).  Tail: the optional factor
// of a scaled map (one more rounding, as the eager product has) and the optional SECOND output  out2 = w (safe-)times out  -- the
// step of Tape::backward() at u = fmadd(a, x, b) under sin: grad_b = cos(u) and grad_a = x cos(u) are two results of one pass
// over a, x, b (20 B/elt) instead of cos written, re-read and multiplied (24 after a stored u, 28 after a recomputed one).
struct ChainTail {
    int scaled, product;            // product: 0 none, 1 w * out, 2 safe_mul(w, out)
};

#ifndef EK_CHAIN_TAIL_VECTORS
#define EK_CHAIN_TAIL_VECTORS 1              // 16-byte vectors per lane of the kernel with a tail (A/B switch)
#endif
#ifndef EK_CHAIN_TAIL_NT
#define EK_CHAIN_TAIL_NT 1                   // nontemporal stores of its outputs
#endif

template <typename T, bool Tail>
__global__ __launch_bounds__(256) void k_chain_map(T *__restrict__ out, size_t n, int vec_ok, ChainArgs<T> ch, ChainTail tail, T scale,
                                                    const T *__restrict__ w, T *__restrict__ out2) {
    constexpr int N = 16 / sizeof(T), U = Tail ? EK_CHAIN_TAIL_VECTORS : 1;
    constexpr bool NT = Tail ? EK_CHAIN_TAIL_NT != 0 : true;
    T s[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) s[k] = ch.src[k].vec ? T(0) : arg_scalar(ch.src[k]);
    size_t e[U];
    bool fast[U];
    T r[U * N], b[U * N], c[U * N];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        e[u] = lane_elem<N, U>(u);
        fast[u] = vec_ok && e[u] + N <= n;
        if (e[u] >= n) {                                   // (only the last vectors of the grid: computed on zeros, not stored)
#pragma unroll
            for (int i = 0; i < N; ++i) r[u * N + i] = b[u * N + i] = c[u * N + i] = T(0);
            continue;
        }
        const Pack<T, N> p0 = arg_load<T, N, true>(ch.src[0], s[0], e[u], n, fast[u]), p1 = arg_load<T, N, true>(ch.src[1], s[1], e[u], n, fast[u]),
                         p2 = arg_load<T, N, true>(ch.src[2], s[2], e[u], n, fast[u]);
#pragma unroll
        for (int i = 0; i < N; ++i) { r[u * N + i] = p0.v[i]; b[u * N + i] = p1.v[i]; c[u * N + i] = p2.v[i]; }
    }
    if (e[0] >= n) return;
    Pack<T, N> pw[U];
    if constexpr (Tail) {
        if (tail.product) {
            const Arg<T> wa{ w, T(0), 1u };
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (e[u] < n) pw[u] = arg_load<T, N, true>(wa, T(0), e[u], n, fast[u]);
        }
    }
    chain_apply<T, U * N>(r, b, c, ch);
    if constexpr (Tail) {
        if (tail.scaled) {
#pragma unroll
            for (int i = 0; i < U * N; ++i) r[i] = r[i] * scale;
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (e[u] >= n) continue;
        Pack<T, N> po;
#pragma unroll
        for (int i = 0; i < N; ++i) po.v[i] = r[u * N + i];
        if (!Tail || out) out_store<T, N, NT>(out, po, e[u], n, fast[u]);
        if constexpr (Tail) {
            if (tail.product) {
                Pack<T, N> pp;
                if (tail.product == 2) {
#pragma unroll
                    for (int i = 0; i < N; ++i) pp.v[i] = dev::safe_mul(pw[u].v[i], r[u * N + i]);
                } else {
#pragma unroll
                    for (int i = 0; i < N; ++i) pp.v[i] = pw[u].v[i] * r[u * N + i];
                }
                out_store<T, N, NT>(out2, pp, e[u], n, fast[u]);
            }
        }
    }
}

template <typename T> int chain_args(const ek_chain *chain, size_t n, ChainArgs<T> &ch, size_t &bytes, int &aligned, const char *what) {
    if (!chain) return fail(EK_ERR_INVALID, "%s: null chain", what);
    if (chain->arity < 1 || chain->arity > 3 || chain->n_maps < 0 || chain->n_maps > 3)
        return fail(EK_ERR_INVALID, "%s: arity %d with %d maps", what, chain->arity, chain->n_maps);
    const int op = chain->base_op;
    if (chain->arity == 1) ch.base = CH_COPY;
    else if (chain->arity == 2 && (op == EK_ADD || op == EK_SUB || op == EK_MUL)) ch.base = op == EK_ADD ? CH_ADD : op == EK_SUB ? CH_SUB : CH_MUL;
    else if (chain->arity == 3 && (op == EK_FMADD || op == EK_FMSUB || op == EK_FNMADD || op == EK_FNMSUB || op == EK_MULADD ||
                                   op == EK_MULSUB || op == EK_NMULADD)) ch.base = CH_TERNARY + op;
    else return fail(EK_ERR_UNSUPPORTED, "%s: base op %d of arity %d cannot head a chain", what, op, chain->arity);
    bytes = 0;
    aligned = 1;
    for (int k = 0; k < 3; ++k) {
        ch.src[k] = Arg<T>{ nullptr, T(0), 0u };
        if (k >= chain->arity) continue;
        if (int rc = make_arg<T>(&chain->src[k], n, ch.src[k], what)) return rc;
        bool dup = false;
        for (int j = 0; j < k; ++j) dup = dup || (ch.src[j].vec && ch.src[k].vec && ch.src[j].ptr == ch.src[k].ptr);
        if (!dup) bytes += arg_bytes(ch.src[k], n);
        aligned = aligned && arg_aligned(ch.src[k]);
    }
    ch.n_maps = chain->n_maps;
    for (int k = 0; k < 3; ++k) {
        ch.map_ops[k] = k < chain->n_maps ? chain->map_ops[k] : (int) EK_COPY;
        if (k < chain->n_maps && !unary_chainable(ch.map_ops[k]))
            return fail(EK_ERR_UNSUPPORTED, "%s: op %d cannot be applied on load", what, ch.map_ops[k]);
    }
    return EK_OK;
}

template <typename T> int chain_reduce(int op, void *out, const ek_chain *chain, size_t n) {
    ChainArgs<T> ch;
    size_t bytes;
    int aligned;
    if (int rc = chain_args<T>(chain, n, ch, bytes, aligned, "ek_hip_reduce_chain()")) return rc;
    RoctxRange range("enoki-hip: horizontal reduction of a chain");
    Context &c = ctx();
    constexpr int N = 16 / sizeof(T);
    // the grid of reduce_launch() over a plain array of n elements: same partial sums, same second stage
    unsigned grid = stream_grid((n / N + 3) / 4 + 1, c.tuning.reduce_blocks_per_cu);
    if (grid > 2048) grid = 2048;
    void *scratch = nullptr;
    if (int rc = reduce_scratch((size_t) grid * sizeof(T), &scratch)) return rc;
#define EK_CHAIN_REDUCE(OP) case OP: hipLaunchKernelGGL((k_chain_reduce<Reducer<OP, T>, T>), dim3(grid), dim3(256), 0, c.stream, (T *) scratch, n, aligned, ch); \
                                     EK_LAUNCH_CHECK("reduce_chain", n, bytes); \
                                     hipLaunchKernelGGL((k_reduce_stage2<Reducer<OP, T>, T>), dim3(1), dim3(256), 0, c.stream, (T *) out, (const T *) scratch, grid); break;
    switch (op) {
        EK_CHAIN_REDUCE(EK_HSUM) EK_CHAIN_REDUCE(EK_HPROD) EK_CHAIN_REDUCE(EK_HMIN) EK_CHAIN_REDUCE(EK_HMAX)
        default: return fail(EK_ERR_INVALID, "ek_hip_reduce_chain(): unknown op %d", op);
    }
#undef EK_CHAIN_REDUCE
    EK_LAUNCH_CHECK("reduce_stage2", (size_t) grid, (size_t) grid * sizeof(T) + sizeof(T));
    return EK_OK;
}

template <typename T> int chain_map(void *out, const ek_chain *chain, size_t n) {
    ChainArgs<T> ch;
    size_t bytes;
    int aligned;
    if (int rc = chain_args<T>(chain, n, ch, bytes, aligned, "ek_hip_map_chain()")) return rc;
    constexpr int N = 16 / sizeof(T);
    hipLaunchKernelGGL((k_chain_map<T, false>), dim3(oneshot_grid<N, 1>(n)), dim3(256), 0, ctx().stream, (T *) out, n, aligned && aligned16(out), ch,
                       ChainTail{ 0, 0 }, T(1), (const T *) nullptr, (T *) nullptr);
    EK_LAUNCH_CHECK("map_chain", n, bytes + n * sizeof(T));
    return EK_OK;
}

template <typename T> int chain_map_product(void *out, void *out2, const ek_chain *chain, const ek_operand *scale, int op2, const ek_operand *w, size_t n) {
    const char *what = "ek_hip_map_chain_product()";
    ChainArgs<T> ch;
    size_t bytes;
    int aligned;
    if (int rc = chain_args<T>(chain, n, ch, bytes, aligned, what)) return rc;
    ChainTail tail{ 0, 0 };
    T factor = T(1);
    if (!out && !out2) return fail(EK_ERR_INVALID, "%s: null pointer", what);
    if (scale) {
        if (scale->ptr) return fail(EK_ERR_INVALID, "%s: the factor is an immediate", what);
        memcpy(&factor, &scale->imm, sizeof(T));
        tail.scaled = 1;
    }
    const T *wp = nullptr;
    if (out2) {
        if (op2 != EK_MUL && op2 != EK_SAFE_MUL) return fail(EK_ERR_UNSUPPORTED, "%s: second output through op %d", what, op2);
        if (!w || !w->ptr || w->size != n) return fail(EK_ERR_INVALID, "%s: the other factor is an array of n elements", what);
        wp = (const T *) w->ptr;
        tail.product = op2 == EK_SAFE_MUL ? 2 : 1;
        aligned = aligned && aligned16(wp) && aligned16(out2);
        bool dup = false;                                   // (the factor usually IS one of the chain's operands: x of fmadd(a, x, b))
        for (int k = 0; k < 3; ++k) dup = dup || (ch.src[k].vec && ch.src[k].ptr == wp);
        bytes += (dup ? 1 : 2) * n * sizeof(T);
    }
    if (out) { aligned = aligned && aligned16(out); bytes += n * sizeof(T); }
    constexpr int N = 16 / sizeof(T);
    hipLaunchKernelGGL((k_chain_map<T, true>), dim3(oneshot_grid<N, EK_CHAIN_TAIL_VECTORS>(n)), dim3(256), 0, ctx().stream, (T *) out, n, aligned, ch, tail,
                       factor, wp, (T *) out2);
    EK_LAUNCH_CHECK("map_chain_product", n, bytes);
    return EK_OK;
}

// Identities of EMPTY inputs follow the CPU reference literally (dynamic.h:633, 651, 669, 687):
// hsum -> 0, hprod -> 1, hmin -> numeric_limits::max(), hmax -> numeric_limits::min() (which is the
// smallest positive normal for floating point types -- a reference quirk we keep).
template <typename T> uint64_t empty_identity_bits(int op) {
    T v;
    switch (op) {
        case EK_HSUM: v = T(0); break;
        case EK_HPROD: v = T(1); break;
        case EK_HMIN: v = std::numeric_limits<T>::max(); break;
        default: v = std::numeric_limits<T>::min(); break;
    }
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    return bits;
}

// ---- inclusive prefix sum ------------------------------------------------------------------------
// Three-phase scan: per-block sums -> scan of block sums (single block) -> per-block scan + offset.
// Blocks own contiguous chunks so results are deterministic.  Only Tape::append_psum needs this
// (autodiff.cpp:473-521); it is not on the timed path.
template <typename T>
__global__ __launch_bounds__(256) void k_scan_block_sums(T *__restrict__ sums, const T *__restrict__ in, size_t n, size_t chunk) {
    size_t begin = (size_t) blockIdx.x * chunk, end = begin + chunk < n ? begin + chunk : n;
    T v = T(0);
    for (size_t i = begin + threadIdx.x; i < end; i += 256) v += in[i];
    v = block_reduce<Reducer<EK_HSUM, T>>(v);
    if (threadIdx.x == 0) sums[blockIdx.x] = v;
}

template <typename T>
__global__ void k_scan_sums_serial(T *__restrict__ sums, unsigned count) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        T run = T(0);
        for (unsigned i = 0; i < count; ++i) { T s = sums[i]; sums[i] = run; run += s; }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_scan_apply(T *__restrict__ out, const T *__restrict__ in, const T *__restrict__ sums,
                                                    size_t n, size_t chunk) {
    __shared__ T tile[256];
    size_t begin = (size_t) blockIdx.x * chunk, end = begin + chunk < n ? begin + chunk : n;
    T carry = sums[blockIdx.x];
    for (size_t base = begin; base < end; base += 256) {
        size_t i = base + threadIdx.x;
        T v = i < end ? in[i] : T(0);
        tile[threadIdx.x] = v;
        __syncthreads();
        // Hillis-Steele within the 256-tile
        for (int d = 1; d < 256; d <<= 1) {
            T add = threadIdx.x >= (unsigned) d ? tile[threadIdx.x - d] : T(0);
            __syncthreads();
            tile[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < end) out[i] = tile[threadIdx.x] + carry;
        carry += tile[255];
        __syncthreads();
    }
}

template <typename T> int psum_single_pass(void *out, const void *in, size_t n);      // scan.hip: decoupled look-back

template <typename T> int psum_typed(void *out, const void *in, size_t n) {
    Context &c = ctx();
    // one pass over the data (8 B per 4-byte element) unless a floating point sum has to be run-to-run reproducible:
    // the look-back's association depends on timing, the three kernels below have a fixed shape
    if (!(std::is_floating_point_v<T> && c.tuning.deterministic)) return psum_single_pass<T>(out, in, n);
    unsigned blocks = (unsigned) std::min<size_t>((n + 4095) / 4096, (size_t) c.num_cu * 4);
    if (blocks == 0) blocks = 1;
    size_t chunk = (n + blocks - 1) / blocks;
    chunk = (chunk + 255) / 256 * 256;
    blocks = (unsigned) ((n + chunk - 1) / chunk);
    void *sums = nullptr;
    if (int rc = ek_hip_malloc((size_t) blocks * sizeof(T), &sums)) return rc;
    hipLaunchKernelGGL((k_scan_block_sums<T>), dim3(blocks), dim3(256), 0, c.stream, (T *) sums, (const T *) in, n, chunk);
    hipLaunchKernelGGL((k_scan_sums_serial<T>), dim3(1), dim3(64), 0, c.stream, (T *) sums, blocks);
    hipLaunchKernelGGL((k_scan_apply<T>), dim3(blocks), dim3(256), 0, c.stream, (T *) out, (const T *) in, (const T *) sums, n, chunk);
    ek_hip_free(sums);   // stream-ordered reuse
    EK_LAUNCH_CHECK("psum", n, 3 * n * sizeof(T));
    return EK_OK;
}

} // namespace ek

using namespace ek;

extern "C" {

int ek_hip_reduce(int op, int type, void *out, const void *in, size_t n) {
    if (int rc = ensure_init()) return rc;
    if (!out) return fail(EK_ERR_INVALID, "ek_hip_reduce(): null output pointer");
    if (op < 0 || op >= EK_REDUCE_COUNT) return fail(EK_ERR_INVALID, "ek_hip_reduce(): unknown op %d", op);
    if (n == 0) {
        uint64_t bits;
        switch (type) {
            case EK_I32: bits = empty_identity_bits<int32_t>(op); break;
            case EK_U32: bits = empty_identity_bits<uint32_t>(op); break;
            case EK_I64: bits = empty_identity_bits<int64_t>(op); break;
            case EK_U64: bits = empty_identity_bits<uint64_t>(op); break;
            case EK_F32: bits = empty_identity_bits<float>(op); break;
            case EK_F64: bits = empty_identity_bits<double>(op); break;
            default: return fail(EK_ERR_UNSUPPORTED, "ek_hip_reduce(): unsupported type %d", type);
        }
        return ek_hip_fill(type, out, bits, 1);
    }
    if (!in) return fail(EK_ERR_INVALID, "ek_hip_reduce(): null input pointer");
    if (n == 1) return ek_hip_memcpy_device(out, in, type_size(type));
    switch (type) {
        case EK_I32: return reduce_dispatch<int32_t>(op, out, in, n);
        case EK_U32: return reduce_dispatch<uint32_t>(op, out, in, n);
        case EK_I64: return reduce_dispatch<int64_t>(op, out, in, n);
        case EK_U64: return reduce_dispatch<uint64_t>(op, out, in, n);
        case EK_F32: return reduce_dispatch<float>(op, out, in, n);
        case EK_F64: return reduce_dispatch<double>(op, out, in, n);
        default: return fail(EK_ERR_UNSUPPORTED, "ek_hip_reduce(): unsupported type %d", type);
    }
}

int ek_hip_reduce_chain(int reduce_op, int type, void *out, const ek_chain *chain, size_t n) {
    if (int rc = ensure_init()) return rc;
    if (!out) return fail(EK_ERR_INVALID, "ek_hip_reduce_chain(): null pointer");
    if (n == 0) return fail(EK_ERR_INVALID, "ek_hip_reduce_chain(): empty input");
    switch (type) {
        case EK_F32: return chain_reduce<float>(reduce_op, out, chain, n);
        case EK_F64: return chain_reduce<double>(reduce_op, out, chain, n);
        default: return fail(EK_ERR_UNSUPPORTED, "ek_hip_reduce_chain(): floating point types only");
    }
}

int ek_hip_map_chain(int type, void *out, const ek_chain *chain, size_t n) {
    if (int rc = ensure_init()) return rc;
    if (n == 0) return EK_OK;
    if (!out) return fail(EK_ERR_INVALID, "ek_hip_map_chain(): null pointer");
    switch (type) {
        case EK_F32: return chain_map<float>(out, chain, n);
        case EK_F64: return chain_map<double>(out, chain, n);
        default: return fail(EK_ERR_UNSUPPORTED, "ek_hip_map_chain(): floating point types only");
    }
}

int ek_hip_map_chain_product(int type, void *out, void *out2, const ek_chain *chain, const ek_operand *scale, int op2, const ek_operand *w, size_t n) {
    if (int rc = ensure_init()) return rc;
    if (n == 0) return EK_OK;
    switch (type) {
        case EK_F32: return chain_map_product<float>(out, out2, chain, scale, op2, w, n);
        case EK_F64: return chain_map_product<double>(out, out2, chain, scale, op2, w, n);
        default: return fail(EK_ERR_UNSUPPORTED, "ek_hip_map_chain_product(): floating point types only");
    }
}

int ek_hip_reduce_map(int op, int map_op, int type, void *out, const void *in, size_t n) {
    if (int rc = ensure_init()) return rc;
    if (!out || !in) return fail(EK_ERR_INVALID, "ek_hip_reduce_map(): null pointer");
    if (n == 0) return fail(EK_ERR_INVALID, "ek_hip_reduce_map(): empty input");
    if (!unary_chainable(map_op)) return fail(EK_ERR_UNSUPPORTED, "ek_hip_reduce_map(): op %d cannot be applied on load", map_op);
    switch (type) {
        case EK_F32: return reduce_map_dispatch<float>(op, map_op, out, in, n);
        case EK_F64: return reduce_map_dispatch<double>(op, map_op, out, in, n);
        default: return fail(EK_ERR_UNSUPPORTED, "ek_hip_reduce_map(): floating point types only");
    }
}

int ek_hip_hsum_safe_mul(int type, void *out, const ek_operand *w, const ek_operand *g, size_t n) {
    if (int rc = ensure_init()) return rc;
    if (!out) return fail(EK_ERR_INVALID, "ek_hip_hsum_safe_mul(): null output pointer");
    if (n == 0) return ek_hip_fill(type, out, 0, 1);
    if (type == EK_F32) {
        SafeMulLoader<float> ld;
        if (int rc = make_arg<float>(w, n, ld.w, "ek_hip_hsum_safe_mul")) return rc;
        if (int rc = make_arg<float>(g, n, ld.g, "ek_hip_hsum_safe_mul")) return rc;
        ld.sw = ld.sg = 0;   // fetched on the device by Loader::init()
        return reduce_launch<Reducer<EK_HSUM, float>>("hsum_safe_mul", (float *) out, n,
                                                      arg_aligned(ld.w) && arg_aligned(ld.g), ld,
                                                      arg_bytes(ld.w, n) + arg_bytes(ld.g, n));
    } else if (type == EK_F64) {
        SafeMulLoader<double> ld;
        if (int rc = make_arg<double>(w, n, ld.w, "ek_hip_hsum_safe_mul")) return rc;
        if (int rc = make_arg<double>(g, n, ld.g, "ek_hip_hsum_safe_mul")) return rc;
        ld.sw = ld.sg = 0;
        return reduce_launch<Reducer<EK_HSUM, double>>("hsum_safe_mul", (double *) out, n,
                                                       arg_aligned(ld.w) && arg_aligned(ld.g), ld,
                                                       arg_bytes(ld.w, n) + arg_bytes(ld.g, n));
    }
    return fail(EK_ERR_UNSUPPORTED, "ek_hip_hsum_safe_mul(): floating point types only");
}

int ek_hip_mask_reduce(int op, const uint8_t *mask, size_t n, uint64_t *host_result) {
    if (int rc = ensure_init()) return rc;
    if (!host_result) return fail(EK_ERR_INVALID, "ek_hip_mask_reduce(): null result pointer");
    if (op < 0 || op >= EK_MASK_REDUCE_COUNT) return fail(EK_ERR_INVALID, "ek_hip_mask_reduce(): unknown op %d", op);
    uint64_t count = 0;
    if (n != 0) {
        if (!mask) return fail(EK_ERR_INVALID, "ek_hip_mask_reduce(): null mask pointer");
        void *dev_count = nullptr;
        if (int rc = ek_hip_malloc(sizeof(uint64_t), &dev_count)) return rc;
        MaskCountLoader ld{ mask };
        int rc = reduce_launch<Reducer<EK_HSUM, uint64_t>>("mask_reduce", (uint64_t *) dev_count, n, aligned16(mask), ld, n);
        if (!rc) rc = ek_hip_memcpy_to_host(&count, dev_count, sizeof(uint64_t));   // synchronizes
        ek_hip_free(dev_count);
        if (rc) return rc;
    }
    switch (op) {
        case EK_ALL: *host_result = count == n; break;      // empty -> true  (dynamic.h:721)
        case EK_ANY: *host_result = count != 0; break;      // empty -> false (dynamic.h:705)
        default: *host_result = count; break;
    }
    return EK_OK;
}

int ek_hip_psum(int type, void *out, const void *in, size_t n) {
    if (int rc = ensure_init()) return rc;
    if (n == 0) return EK_OK;
    if (!out || !in) return fail(EK_ERR_INVALID, "ek_hip_psum(): null pointer");
    switch (type) {
        case EK_I32: case EK_U32: return psum_typed<uint32_t>(out, in, n);
        case EK_I64: case EK_U64: return psum_typed<uint64_t>(out, in, n);
        case EK_F32: return psum_typed<float>(out, in, n);
        case EK_F64: return psum_typed<double>(out, in, n);
        default: return fail(EK_ERR_UNSUPPORTED, "ek_hip_psum(): unsupported type %d", type);
    }
}

} // extern "C"
