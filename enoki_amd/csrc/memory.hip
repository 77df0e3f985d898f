// Initialisation and indexed-memory kernels: fill / arange / linspace / reverse and masked
// gather / scatter / scatter_add.  SURVEY.md rows a6, a7.
//
// Reference semantics:
//   gather_      out[i] = mask[i] ? base[index[i]] : 0            cuda.h:845-864, dynamic.h:478-496
//   scatter_     if (mask[i]) base[index[i]] = value[i]            cuda.h:866-890, dynamic.h:498-515
//   scatter_add_ if (mask[i]) base[index[i]] += value[i]           cuda.h:892-905 (atom.global.add)
// Indices are element offsets (stride = sizeof(T)), signed or unsigned 32/64 bit.  Index and
// mask vectors are read as 16-byte (resp. 4-byte) packs per lane so that the streaming side of
// these kernels stays coalesced; the random side goes through L2 / Infinity Cache.
#include "ek_map.h"

namespace ek {

// ---- fill / arange / linspace / reverse -----------------------------------------------------------
template <typename T> struct FillOp {
    static __device__ __forceinline__ T apply(T x) { return x; }
};

template <typename T>
__global__ __launch_bounds__(256) void k_arange(T *__restrict__ out, size_t n, T start, T step) {
    using U = wrap_t<T>;
    const size_t gid = (size_t) blockIdx.x * 256 + threadIdx.x, total = (size_t) gridDim.x * 256;
    for (size_t i = gid; i < n; i += total) {
        if constexpr (std::is_floating_point_v<T>) {
            // fmadd(index, step, start)  (cuda.h:649-652)
            if constexpr (sizeof(T) == 4) out[i] = __builtin_fmaf((float) (uint32_t) i, step, start);
            else out[i] = __builtin_fma((double) (uint32_t) i, step, start);
        } else {
            out[i] = (T) ((U) start + (U) i * (U) step);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_reverse(T *__restrict__ out, const T *__restrict__ in, size_t n) {
    const size_t gid = (size_t) blockIdx.x * 256 + threadIdx.x, total = (size_t) gridDim.x * 256;
    for (size_t i = gid; i < n; i += total)
        out[i] = in[n - 1 - i];
}

// out[i] = mask[i] && address[i] ? *(T *) (address[i] + byte_offset) : 0 -- a data member read out of instance memory through a
// pointer array (ENOKI_CALL_SUPPORT_GETTER, array_call.h:269-283: gather<Return, 1>(nullptr, self + offset, mask))
template <typename T>
__global__ __launch_bounds__(256) void k_gather_address(T *__restrict__ out, Arg<uint64_t> address, int64_t byte_offset, Arg<uint8_t> mask,
                                                        size_t n) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t a = address.vec ? address.ptr[i] : arg_scalar(address);
    const bool on = (mask.vec ? mask.ptr[i] : arg_scalar(mask)) != 0 && a != 0;
    out[i] = on ? *reinterpret_cast<const T *>(a + (uint64_t) byte_offset) : T(0);
}

// ---- concat ------------------------------------------------------------------------------------
constexpr int kConcatMax = 8;
struct ConcatArgs { const void *src[kConcatMax]; size_t end[kConcatMax]; unsigned bcast = 0; /* bit k: source k is ONE entry copied to every row (k_concat_rows) */ };

// out[r][end[k-1] + j] = src_k[r * c_k + j]: every source seen as [rows, c_k], concatenated along the columns (a.end = column
// boundaries within one output row of `row` entries)
template <typename T> __global__ __launch_bounds__(256) void k_concat_rows(T *__restrict__ out, ConcatArgs a, size_t row, size_t total) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t r = i / row, col = i - r * row;
    const void *src = a.src[0];
    size_t begin = 0, width = a.end[0];
    unsigned bc = a.bcast & 1u;
#pragma unroll
    for (int k = 0; k < kConcatMax - 1; ++k)
        if (col >= a.end[k]) { src = a.src[k + 1]; begin = a.end[k]; width = a.end[k + 1] - a.end[k]; bc = (a.bcast >> (k + 1)) & 1u; }
    out[i] = static_cast<const T *>(src)[bc ? 0 : r * width + (col - begin)];
}

template <typename T> __global__ __launch_bounds__(256) void k_concat(T *__restrict__ out, ConcatArgs a, size_t total) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const void *src = a.src[0];
    size_t begin = 0;
#pragma unroll
    for (int k = 0; k < kConcatMax - 1; ++k)            // ends ascend: the last boundary at or below i wins
        if (i >= a.end[k]) { src = a.src[k + 1]; begin = a.end[k]; }
    out[i] = static_cast<const T *>(src)[i - begin];
}

// ---- gather ------------------------------------------------------------------------------------
template <typename I> __device__ __forceinline__ int64_t index_offset(I i) { return (int64_t) i; }

template <typename T, typename I, int N>
__global__ __launch_bounds__(256) void k_gather(T *__restrict__ out, const T *__restrict__ base, Arg<I> index,
                                                Arg<uint8_t> mask, size_t n, int vec_ok) {
    const I si = index.vec ? I(0) : arg_scalar(index);
    const uint8_t sm = mask.vec ? uint8_t(0) : arg_scalar(mask);
    const size_t e = lane_elem<N, 1>(0);
    if (e >= n) return;
    const bool fast = vec_ok && e + N <= n;
    Pack<I, N> pi = arg_load<I, N, true>(index, si, e, n, fast);
    Pack<uint8_t, N> pm = arg_load<uint8_t, N, true>(mask, sm, e, n, fast);
    Pack<T, N> po;
#pragma unroll
    for (int k = 0; k < N; ++k)
        po.v[k] = (pm.v[k] && e + k < n) ? base[index_offset(pi.v[k])] : T(0);
    out_store<T, N, true>(out, po, e, n, fast);
}

// Gather of a structure-of-arrays value (Array<HIPArray<T>, C> = C tables sharing one index array, array_struct.h:9-40):
// the index and mask vectors are read once and every lane issues its C x N lookups back to back.
template <typename T, int C> struct TablePtrs { const T *base[C]; T *out[C]; };

template <typename T, typename I, int N, int C>
__global__ __launch_bounds__(256) void k_gather_multi(TablePtrs<T, C> t, Arg<I> index, Arg<uint8_t> mask, size_t n, int vec_ok) {
    const I si = index.vec ? I(0) : arg_scalar(index);
    const uint8_t sm = mask.vec ? uint8_t(0) : arg_scalar(mask);
    const size_t e = lane_elem<N, 1>(0);
    if (e >= n) return;
    const bool fast = vec_ok && e + N <= n;
    Pack<I, N> pi = arg_load<I, N, true>(index, si, e, n, fast);
    Pack<uint8_t, N> pm = arg_load<uint8_t, N, true>(mask, sm, e, n, fast);
    Pack<T, N> po[C];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int k = 0; k < N; ++k)
            po[c].v[k] = (pm.v[k] && e + k < n) ? t.base[c][index_offset(pi.v[k])] : T(0);
#pragma unroll
    for (int c = 0; c < C; ++c) out_store<T, N, true>(t.out[c], po[c], e, n, fast);
}

template <typename T, typename I, int C>
int gather_multi_launch(void *const *outs, const void *const *bases, const ek_operand *index, const ek_operand *mask, size_t n) {
    constexpr int N = 16 / (sizeof(T) > sizeof(I) ? sizeof(T) : sizeof(I));
    Arg<I> ii;
    Arg<uint8_t> mm;
    if (int rc = make_arg<I>(index, n, ii, "ek_hip_gather_multi")) return rc;
    if (int rc = make_arg<uint8_t>(mask, n, mm, "ek_hip_gather_multi")) return rc;
    TablePtrs<T, C> t;
    int vec_ok = arg_aligned(ii) && arg_aligned(mm);
    for (int c = 0; c < C; ++c) {
        if (!outs[c] || !bases[c]) return fail(EK_ERR_INVALID, "ek_hip_gather_multi(): null pointer");
        t.base[c] = (const T *) bases[c];
        t.out[c] = (T *) outs[c];
        vec_ok = vec_ok && aligned16(outs[c]);
    }
    Context &c = ctx();
    unsigned grid = (unsigned) ((n + (size_t) 256 * N - 1) / ((size_t) 256 * N));
    hipLaunchKernelGGL((k_gather_multi<T, I, N, C>), dim3(grid), dim3(256), 0, c.stream, t, ii, mm, n, vec_ok);
    EK_LAUNCH_CHECK("gather", n, arg_bytes(ii, n) + arg_bytes(mm, n) + (size_t) C * 2 * n * sizeof(T));
    return EK_OK;
}

template <typename T, typename I>
int gather_multi_dispatch(int count, void *const *outs, const void *const *bases, const ek_operand *index, const ek_operand *mask,
                          size_t n) {
    switch (count) {
        case 2: return gather_multi_launch<T, I, 2>(outs, bases, index, mask, n);
        case 3: return gather_multi_launch<T, I, 3>(outs, bases, index, mask, n);
        case 4: return gather_multi_launch<T, I, 4>(outs, bases, index, mask, n);
        default: return fail(EK_ERR_INVALID, "ek_hip_gather_multi(): 2, 3 or 4 components expected, got %d", count);
    }
}

// Struct gathers through staged records.  Every lookup that misses the L2 is one 64-byte request to the fabric whatever
// its width, and that request rate -- not bytes -- bounds a random gather (profiles/rocprof_l2_r0*.txt).  C tables of K
// entries are therefore first interleaved into ONE table of K records of R = 2 or 4 slots (8 or 16 bytes; a third
// component is padded to four), a streaming pass of (C + R) * sizeof(T) * K bytes, and the gather issues one R-slot load
// per element instead of C scalar ones.
template <typename T, int C, int R>
__global__ __launch_bounds__(256) void k_stage_records(Pack<T, R> *__restrict__ rec, TablePtrs<T, C> t, size_t k, int vec_ok) {
    constexpr int N = 16 / sizeof(T);
    const size_t e = ((size_t) blockIdx.x * 256 + threadIdx.x) * N;
    if (e >= k) return;
    if (vec_ok && e + N <= k) {
        Pack<T, N> p[C];
#pragma unroll
        for (int c = 0; c < C; ++c) p[c] = pack_load<T, N, false>(t.base[c] + e);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            Pack<T, R> r;
#pragma unroll
            for (int c = 0; c < R; ++c) r.v[c] = c < C ? p[c < C ? c : 0].v[j] : T(0);
            rec[e + j] = r;
        }
    } else {
        for (size_t i = e; i < k && i < e + N; ++i) {
            Pack<T, R> r;
#pragma unroll
            for (int c = 0; c < R; ++c) r.v[c] = c < C ? t.base[c < C ? c : 0][i] : T(0);
            rec[i] = r;
        }
    }
}

template <typename T, typename I, int N, int C, int R>
__global__ __launch_bounds__(256) void k_gather_records(TablePtrs<T, C> t, const Pack<T, R> *__restrict__ rec, Arg<I> index,
                                                        Arg<uint8_t> mask, size_t n, int vec_ok) {
    const I si = index.vec ? I(0) : arg_scalar(index);
    const uint8_t sm = mask.vec ? uint8_t(0) : arg_scalar(mask);
    const size_t e = lane_elem<N, 1>(0);
    if (e >= n) return;
    const bool fast = vec_ok && e + N <= n;
    Pack<I, N> pi = arg_load<I, N, true>(index, si, e, n, fast);
    Pack<uint8_t, N> pm = arg_load<uint8_t, N, true>(mask, sm, e, n, fast);
    Pack<T, R> got[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        // (not `on ? rec[...] : zero`: a select between two records is a select between two ADDRESSES, and the zero record
        // then lives in scratch memory)
#pragma unroll
        for (int c = 0; c < R; ++c) got[k].v[c] = T(0);
        if (pm.v[k] && e + k < n) got[k] = rec[index_offset(pi.v[k])];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        Pack<T, N> po;
#pragma unroll
        for (int k = 0; k < N; ++k) po.v[k] = got[k].v[c];
        out_store<T, N, true>(t.out[c], po, e, n, fast);
    }
}

/// Worth staging?  Fitted to profiles/probe_gather_records_r02.txt (C = 2..4, K = 256 Ki..32 Mi entries, 1 Mi..64 Mi lookups):
///  * a table that fits the 4 MiB L2 of an XCD is left alone -- its lookups mostly hit, the gain is a few percent, and the
///    binding can keep such gathers unevaluated and fuse them into their consumers, which is worth as much;
///  * beyond that every component lookup that misses is one fabric request of ~16 ps, and records save C - 1 of them
///    for the share of the table that does not fit (1 - 4 MiB / table), of which 80 % is counted;
///  * staging streams (C + R) * sizeof(T) bytes per entry at ~5 TB/s plus ~10 us of launch and allocation, with a 25 %
///    margin.  The rule reproduces every win / loss of the probe outside +-8 %.
static bool gather_records_pays(int mode, int count, size_t elem, size_t base_size, size_t n) {
    if (mode == 0 || base_size == 0) return false;
    if (mode == 2) return true;
    const double table = (double) elem * (double) base_size, l2 = (double) ((size_t) 4 << 20);
    if (table <= l2) return false;
    const int slots = count == 2 ? 2 : 4;
    const double saved_ps = (double) (count - 1) * 16.0 * (1.0 - l2 / table) * 0.8 * (double) n;
    const double stage_ps = (double) (count + slots) * table * 0.2 + 1e7;
    return saved_ps > 1.25 * stage_ps;
}

template <typename T, typename I, int C>
int gather_records_launch(void *const *outs, const void *const *bases, size_t base_size, const ek_operand *index,
                          const ek_operand *mask, size_t n) {
    constexpr int R = C == 2 ? 2 : 4;
    constexpr int N = 16 / (sizeof(T) > sizeof(I) ? sizeof(T) : sizeof(I));
    Arg<I> ii;
    Arg<uint8_t> mm;
    if (int rc = make_arg<I>(index, n, ii, "ek_hip_gather_multi_sized")) return rc;
    if (int rc = make_arg<uint8_t>(mask, n, mm, "ek_hip_gather_multi_sized")) return rc;
    TablePtrs<T, C> t;
    int vec_ok = arg_aligned(ii) && arg_aligned(mm), stage_vec = 1;
    for (int c = 0; c < C; ++c) {
        if (!outs[c] || !bases[c]) return fail(EK_ERR_INVALID, "ek_hip_gather_multi_sized(): null pointer");
        t.base[c] = (const T *) bases[c];
        t.out[c] = (T *) outs[c];
        vec_ok = vec_ok && aligned16(outs[c]);
        stage_vec = stage_vec && aligned16(bases[c]);
    }
    void *rec = nullptr;
    if (ek_hip_malloc(base_size * sizeof(Pack<T, R>), &rec) != EK_OK)      // no room for the records: the plain kernels need none
        return gather_multi_launch<T, I, C>(outs, bases, index, mask, n);
    Context &c = ctx();
    constexpr size_t per_block = 256 * (16 / sizeof(T));
    hipLaunchKernelGGL((k_stage_records<T, C, R>), dim3((unsigned) ((base_size + per_block - 1) / per_block)), dim3(256), 0, c.stream,
                       (Pack<T, R> *) rec, t, base_size, stage_vec);
    note_launch("gather_stage_records", base_size, (size_t) (C + R) * sizeof(T) * base_size);
    unsigned grid = (unsigned) ((n + (size_t) 256 * N - 1) / ((size_t) 256 * N));
    hipLaunchKernelGGL((k_gather_records<T, I, N, C, R>), dim3(grid), dim3(256), 0, c.stream, t, (const Pack<T, R> *) rec, ii, mm, n, vec_ok);
    int rc = EK_OK;
    if (hipError_t e = hipGetLastError(); e != hipSuccess) rc = hip_fail(e, "k_gather_records", __FILE__, __LINE__);
    else note_launch("gather_records", n, arg_bytes(ii, n) + arg_bytes(mm, n) + (size_t) C * 2 * n * sizeof(T));
    ek_hip_free(rec);                 // stream-ordered reuse: the kernel above is already enqueued
    return rc;
}

// ---- scatter / scatter_add ---------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void atomic_add(T *addr, T v) {
    if constexpr (std::is_same_v<T, float>) {
        unsafeAtomicAdd(addr, v);                       // global_atomic_add_f32, no CAS loop
    } else if constexpr (std::is_same_v<T, double>) {
        unsafeAtomicAdd(addr, v);                       // global_atomic_add_f64
    } else if constexpr (sizeof(T) == 4) {
        atomicAdd(reinterpret_cast<unsigned int *>(addr), (unsigned int) v);
    } else {
        atomicAdd(reinterpret_cast<unsigned long long *>(addr), (unsigned long long) v);
    }
}

template <typename T, typename I, int N, bool Add>
__global__ __launch_bounds__(256) void k_scatter(T *__restrict__ base, Arg<T> value, Arg<I> index,
                                                 Arg<uint8_t> mask, size_t n, int vec_ok) {
    const T sv = value.vec ? T(0) : arg_scalar(value);
    const I si = index.vec ? I(0) : arg_scalar(index);
    const uint8_t sm = mask.vec ? uint8_t(0) : arg_scalar(mask);
    const size_t e = lane_elem<N, 1>(0);
    if (!Add && e >= n) return;
    // (scatter_add: lanes past the end stay in the wave -- they take part in the shuffles below with `on` = false;
    //  arg_load bounds-checks every element on the non-vector path)
    const bool fast = vec_ok && e + N <= n;
    Pack<T, N> pv = arg_load<T, N, true>(value, sv, e, n, fast);
    Pack<I, N> pi = arg_load<I, N, true>(index, si, e, n, fast);
    Pack<uint8_t, N> pm = arg_load<uint8_t, N, true>(mask, sm, e, n, fast);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        bool on = pm.v[k] && e + k < n;
        if constexpr (Add) {
            // Same-address device atomics retire at only ~0.08 G/s (profiles/probe_skew_r01.txt), so hot bins are
            // pre-combined inside the wave: the lanes that share the first active lane's index reduce their values
            // and issue ONE atomic; two rounds peel the two hottest bins of the instruction.  With well spread
            // indices this costs a shuffle, a compare and a ballot per element.
            const I my = pi.v[k];
            const int lane = threadIdx.x & 63;
            using U = wrap_t<T>;
#pragma unroll
            for (int round = 0; round < 2; ++round) {
                const unsigned long long act = __ballot(on);
                if (act == 0) break;
                const int leader = __ffsll((long long) act) - 1;
                const I hot = __shfl(my, leader);
                const bool same = on && my == hot;
                if (__popcll(__ballot(same)) < 2) break;         // the leader's bin is not shared: nothing to combine
                U total = same ? (U) pv.v[k] : U(0);
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) total = (U) (total + (U) __shfl_xor(total, d));
                if (lane == leader) atomic_add(base + index_offset(hot), (T) total);
                on = on && !same;
            }
            if (on) atomic_add(base + index_offset(my), pv.v[k]);
        } else {
            if (on) base[index_offset(pi.v[k])] = pv.v[k];
        }
    }
}

template <typename T, typename I>
int gather_launch(void *out, const void *base, const ek_operand *index, const ek_operand *mask, size_t n) {
    constexpr int N = 16 / (sizeof(T) > sizeof(I) ? sizeof(T) : sizeof(I));
    Arg<I> ii;
    Arg<uint8_t> mm;
    if (int rc = make_arg<I>(index, n, ii, "ek_hip_gather")) return rc;
    if (int rc = make_arg<uint8_t>(mask, n, mm, "ek_hip_gather")) return rc;
    int vec_ok = aligned16(out) && arg_aligned(ii) && arg_aligned(mm);
    Context &c = ctx();
    unsigned grid = (unsigned) ((n + (size_t) 256 * N - 1) / ((size_t) 256 * N));
    hipLaunchKernelGGL((k_gather<T, I, N>), dim3(grid), dim3(256), 0, c.stream, (T *) out, (const T *) base, ii, mm,
                       n, vec_ok);
    // algorithmic bytes: index + mask + output + one table element per lane (SURVEY.md 8d counts the
    // table read as sizeof(T) per element even when the table is cache resident)
    EK_LAUNCH_CHECK("gather", n, arg_bytes(ii, n) + arg_bytes(mm, n) + 2 * n * sizeof(T));
    return EK_OK;
}

template <typename T, typename I, bool Add>
int scatter_launch(void *base, const ek_operand *value, const ek_operand *index, const ek_operand *mask, size_t n) {
    constexpr int N = 16 / (sizeof(T) > sizeof(I) ? sizeof(T) : sizeof(I));
    Arg<T> vv;
    Arg<I> ii;
    Arg<uint8_t> mm;
    if (int rc = make_arg<T>(value, n, vv, "ek_hip_scatter")) return rc;
    if (int rc = make_arg<I>(index, n, ii, "ek_hip_scatter")) return rc;
    if (int rc = make_arg<uint8_t>(mask, n, mm, "ek_hip_scatter")) return rc;
    int vec_ok = arg_aligned(vv) && arg_aligned(ii) && arg_aligned(mm);
    Context &c = ctx();
    unsigned grid = (unsigned) ((n + (size_t) 256 * N - 1) / ((size_t) 256 * N));
    hipLaunchKernelGGL((k_scatter<T, I, N, Add>), dim3(grid), dim3(256), 0, c.stream, (T *) base, vv, ii, mm, n,
                       vec_ok);
    EK_LAUNCH_CHECK(Add ? "scatter_add" : "scatter", n, arg_bytes(vv, n) + arg_bytes(ii, n) + arg_bytes(mm, n));
    return EK_OK;
}

#define EK_INDEX_SWITCH(itype, CALL, WHAT)                                                        \
    switch (itype) {                                                                             \
        case EK_I32: { using I = int32_t; return CALL; }                                          \
        case EK_U32: { using I = uint32_t; return CALL; }                                         \
        case EK_I64: { using I = int64_t; return CALL; }                                          \
        case EK_U64: { using I = uint64_t; return CALL; }                                         \
        default: return fail(EK_ERR_INVALID, WHAT ": index type %d is not an integer type", itype); \
    }

} // namespace ek

using namespace ek;

extern "C" {

int ek_hip_fill(int type, void *out, uint64_t imm, size_t n) {
    if (int rc = ensure_init()) return rc;
    if (n == 0) return EK_OK;
    if (!out) return fail(EK_ERR_INVALID, "ek_hip_fill(): null output pointer");
    ek_operand op = { nullptr, imm, 1 };
    // moves bit patterns: dispatch on the element width only (cuda_fill, common.cu:56-80)
    switch (type_size(type)) {
        case 1: { Arg<uint8_t> a; make_arg<uint8_t>(&op, n, a, "fill"); return launch_map1<FillOp<uint8_t>>("fill", (uint8_t *) out, n, a); }
        case 4: { Arg<uint32_t> a; make_arg<uint32_t>(&op, n, a, "fill"); return launch_map1<FillOp<uint32_t>>("fill", (uint32_t *) out, n, a); }
        case 8: { Arg<uint64_t> a; make_arg<uint64_t>(&op, n, a, "fill"); return launch_map1<FillOp<uint64_t>>("fill", (uint64_t *) out, n, a); }
        default: return fail(EK_ERR_INVALID, "ek_hip_fill(): unknown type %d", type);
    }
}

int ek_hip_arange(int type, void *out, int64_t start, int64_t step, size_t n) {
    if (int rc = ensure_init()) return rc;
    if (n == 0) return EK_OK;
    if (!out) return fail(EK_ERR_INVALID, "ek_hip_arange(): null output pointer");
    Context &c = ctx();
    unsigned grid = stream_grid(n, c.tuning.blocks_per_cu);
#define EK_ARANGE(T) hipLaunchKernelGGL((k_arange<T>), dim3(grid), dim3(256), 0, c.stream, (T *) out, n, (T) start, (T) step); break
    switch (type) {
        case EK_I32: EK_ARANGE(int32_t);
        case EK_U32: EK_ARANGE(uint32_t);
        case EK_I64: EK_ARANGE(int64_t);
        case EK_U64: EK_ARANGE(uint64_t);
        case EK_F32: EK_ARANGE(float);
        case EK_F64: EK_ARANGE(double);
        default: return fail(EK_ERR_INVALID, "ek_hip_arange(): unsupported type %d", type);
    }
#undef EK_ARANGE
    EK_LAUNCH_CHECK("arange", n, n * type_size(type));
    return EK_OK;
}

int ek_hip_linspace(int type, void *out, double min, double max, size_t n) {
    if (int rc = ensure_init()) return rc;
    if (n == 0) return EK_OK;
    if (!out) return fail(EK_ERR_INVALID, "ek_hip_linspace(): null output pointer");
    Context &c = ctx();
    unsigned grid = stream_grid(n, c.tuning.blocks_per_cu);
    if (type == EK_F32) {
        float lo = (float) min, hi = (float) max;
        float step = (hi - lo) / (float) (n - 1);           // cuda.h:661
        hipLaunchKernelGGL((k_arange<float>), dim3(grid), dim3(256), 0, c.stream, (float *) out, n, lo, step);
    } else if (type == EK_F64) {
        double step = (max - min) / (double) (n - 1);
        hipLaunchKernelGGL((k_arange<double>), dim3(grid), dim3(256), 0, c.stream, (double *) out, n, min, step);
    } else {
        return fail(EK_ERR_UNSUPPORTED, "ek_hip_linspace(): floating point types only");
    }
    EK_LAUNCH_CHECK("linspace", n, n * type_size(type));
    return EK_OK;
}

int ek_hip_reverse(int type, void *out, const void *in, size_t n) {
    if (int rc = ensure_init()) return rc;
    if (n == 0) return EK_OK;
    if (!out || !in) return fail(EK_ERR_INVALID, "ek_hip_reverse(): null pointer");
    Context &c = ctx();
    unsigned grid = stream_grid(n, c.tuning.blocks_per_cu);
    switch (type_size(type)) {
        case 1: hipLaunchKernelGGL((k_reverse<uint8_t>), dim3(grid), dim3(256), 0, c.stream, (uint8_t *) out, (const uint8_t *) in, n); break;
        case 4: hipLaunchKernelGGL((k_reverse<uint32_t>), dim3(grid), dim3(256), 0, c.stream, (uint32_t *) out, (const uint32_t *) in, n); break;
        case 8: hipLaunchKernelGGL((k_reverse<uint64_t>), dim3(grid), dim3(256), 0, c.stream, (uint64_t *) out, (const uint64_t *) in, n); break;
        default: return fail(EK_ERR_INVALID, "ek_hip_reverse(): unknown type %d", type);
    }
    EK_LAUNCH_CHECK("reverse", n, 2 * n * type_size(type));
    return EK_OK;
}

int ek_hip_concat_rows(int type, void *out, size_t rows, int count, const void *const *srcs, const size_t *sizes) {
    if (int rc = ensure_init()) return rc;
    if (count < 1 || count > kConcatMax) return fail(EK_ERR_INVALID, "ek_hip_concat_rows(): 1 to %d arrays expected, got %d", kConcatMax, count);
    if (!out || !srcs || !sizes || rows == 0) return fail(EK_ERR_INVALID, "ek_hip_concat_rows(): null pointer / no rows");
    ConcatArgs a;
    size_t row = 0;
    for (int i = 0; i < kConcatMax; ++i) {
        a.src[i] = nullptr;
        a.end[i] = row;
        if (i < count) {
            if (!srcs[i] || (sizes[i] != 1 && sizes[i] % rows))
                return fail(EK_ERR_INVALID, "ek_hip_concat_rows(): array %d: null, or neither one entry nor a multiple of %zu rows", i, rows);
            a.src[i] = srcs[i];
            if (sizes[i] == 1) { a.bcast |= 1u << i; row += 1; }      // one entry: the same value in every row
            else row += sizes[i] / rows;
            a.end[i] = row;
        }
    }
    if (row == 0) return EK_OK;
    Context &c = ctx();
    const size_t total = rows * row;
    unsigned grid = (unsigned) ((total + 255) / 256);
    switch (type_size(type)) {
        case 4: hipLaunchKernelGGL((k_concat_rows<uint32_t>), dim3(grid), dim3(256), 0, c.stream, (uint32_t *) out, a, row, total); break;
        case 8: hipLaunchKernelGGL((k_concat_rows<uint64_t>), dim3(grid), dim3(256), 0, c.stream, (uint64_t *) out, a, row, total); break;
        default: return fail(EK_ERR_INVALID, "ek_hip_concat_rows(): 4- and 8-byte types only");
    }
    EK_LAUNCH_CHECK("concat_rows", total, 2 * total * type_size(type));
    return EK_OK;
}

int ek_hip_concat(int type, void *out, int count, const void *const *srcs, const size_t *sizes) {
    if (int rc = ensure_init()) return rc;
    if (count < 1 || count > kConcatMax) return fail(EK_ERR_INVALID, "ek_hip_concat(): 1 to %d arrays expected, got %d", kConcatMax, count);
    if (!out || !srcs || !sizes) return fail(EK_ERR_INVALID, "ek_hip_concat(): null pointer");
    ConcatArgs a;
    size_t total = 0;
    for (int i = 0; i < kConcatMax; ++i) {
        a.src[i] = nullptr;
        a.end[i] = total;
        if (i < count) {
            if (sizes[i] && !srcs[i]) return fail(EK_ERR_INVALID, "ek_hip_concat(): null pointer");
            a.src[i] = srcs[i];
            total += sizes[i];
            a.end[i] = total;
        }
    }
    if (total == 0) return EK_OK;
    Context &c = ctx();
    unsigned grid = (unsigned) ((total + 255) / 256);
    switch (type_size(type)) {
        case 1: hipLaunchKernelGGL((k_concat<uint8_t>), dim3(grid), dim3(256), 0, c.stream, (uint8_t *) out, a, total); break;
        case 4: hipLaunchKernelGGL((k_concat<uint32_t>), dim3(grid), dim3(256), 0, c.stream, (uint32_t *) out, a, total); break;
        case 8: hipLaunchKernelGGL((k_concat<uint64_t>), dim3(grid), dim3(256), 0, c.stream, (uint64_t *) out, a, total); break;
        default: return fail(EK_ERR_INVALID, "ek_hip_concat(): unknown type %d", type);
    }
    EK_LAUNCH_CHECK("concat", total, 2 * total * type_size(type));
    return EK_OK;
}

int ek_hip_sort_pairs(int key_bits, const uint32_t *keys, size_t n, uint32_t *keys_out, uint32_t *perm_out) {
    if (int rc = ensure_init()) return rc;
    if (n == 0) return EK_OK;
    if (!keys || !keys_out || !perm_out) return fail(EK_ERR_INVALID, "ek_hip_sort_pairs(): null pointer");
    if (n >= ((size_t) 1 << 32)) return fail(EK_ERR_UNSUPPORTED, "ek_hip_sort_pairs(): more than 2^32 - 1 entries");
    if (key_bits > 32) key_bits = 32;
    return sort_pairs_u32(key_bits, keys, n, keys_out, perm_out);
}

int ek_hip_gather(int type, int index_type, void *out, const void *base, const ek_operand *index,
                  const ek_operand *mask, size_t n) {
    if (int rc = ensure_init()) return rc;
    if (n == 0) return EK_OK;
    if (!out || !base) return fail(EK_ERR_INVALID, "ek_hip_gather(): null pointer");
    switch (type_size(type)) {
        case 1: EK_INDEX_SWITCH(index_type, (gather_launch<uint8_t, I>(out, base, index, mask, n)), "ek_hip_gather()")
        case 4: EK_INDEX_SWITCH(index_type, (gather_launch<uint32_t, I>(out, base, index, mask, n)), "ek_hip_gather()")
        case 8: EK_INDEX_SWITCH(index_type, (gather_launch<uint64_t, I>(out, base, index, mask, n)), "ek_hip_gather()")
        default: return fail(EK_ERR_INVALID, "ek_hip_gather(): unknown type %d", type);
    }
}

int ek_hip_gather_address(int type, void *out, const ek_operand *address, int64_t byte_offset, const ek_operand *mask, size_t n) {
    if (int rc = ensure_init()) return rc;
    if (n == 0) return EK_OK;
    if (!out || !address) return fail(EK_ERR_INVALID, "ek_hip_gather_address(): null pointer");
    Arg<uint64_t> aa;
    Arg<uint8_t> mm;
    if (int rc = make_arg<uint64_t>(address, n, aa, "ek_hip_gather_address")) return rc;
    if (int rc = make_arg<uint8_t>(mask, n, mm, "ek_hip_gather_address")) return rc;
    Context &c = ctx();
    const unsigned grid = (unsigned) ((n + 255) / 256);
    switch (type_size(type)) {
        case 1: hipLaunchKernelGGL((k_gather_address<uint8_t>), dim3(grid), dim3(256), 0, c.stream, (uint8_t *) out, aa, byte_offset, mm, n); break;
        case 4: hipLaunchKernelGGL((k_gather_address<uint32_t>), dim3(grid), dim3(256), 0, c.stream, (uint32_t *) out, aa, byte_offset, mm, n); break;
        case 8: hipLaunchKernelGGL((k_gather_address<uint64_t>), dim3(grid), dim3(256), 0, c.stream, (uint64_t *) out, aa, byte_offset, mm, n); break;
        default: return fail(EK_ERR_INVALID, "ek_hip_gather_address(): unknown type %d", type);
    }
    EK_LAUNCH_CHECK("gather_address", n, arg_bytes(aa, n) + arg_bytes(mm, n) + 2 * n * type_size(type));
    return EK_OK;
}

int ek_hip_gather_multi(int type, int index_type, int count, void *const *outs, const void *const *bases,
                        const ek_operand *index, const ek_operand *mask, size_t n) {
    if (int rc = ensure_init()) return rc;
    if (n == 0) return EK_OK;
    if (!outs || !bases) return fail(EK_ERR_INVALID, "ek_hip_gather_multi(): null pointer");
    switch (type_size(type)) {
        case 4: EK_INDEX_SWITCH(index_type, (gather_multi_dispatch<uint32_t, I>(count, outs, bases, index, mask, n)), "ek_hip_gather_multi()")
        case 8: EK_INDEX_SWITCH(index_type, (gather_multi_dispatch<uint64_t, I>(count, outs, bases, index, mask, n)), "ek_hip_gather_multi()")
        default: return fail(EK_ERR_UNSUPPORTED, "ek_hip_gather_multi(): 4- and 8-byte element types only");
    }
}

int ek_hip_gather_multi_plan(int type, int index_type, int count, size_t base_size, size_t n) {
    if (ensure_init()) return EK_GATHER_ONE_LAUNCH;
    const size_t elem = type_size(type);
    const bool narrow = index_type == EK_U32 || index_type == EK_I32;
    const bool shape_ok = (elem == 4 && count >= 2 && count <= 4) || (elem == 8 && count == 2);
    if (narrow && shape_ok && base_size <= ((size_t) 1 << 32) &&
        gather_records_pays(ctx().tuning.gather_records, count, elem, base_size, n))
        return EK_GATHER_RECORDS;
    // One launch reads the indices once, but its working set is ALL tables: when one table fits the 4 MiB L2 of an XCD
    // and the set does not, one launch per table is up to 2x faster (profiles/probe_gather_multi_r01.txt)
    if (base_size != 0 && (size_t) count * elem * base_size > ((size_t) 3 << 20) && elem * base_size < ((size_t) 128 << 20))
        return EK_GATHER_PER_TABLE;
    return EK_GATHER_ONE_LAUNCH;
}

int ek_hip_gather_multi_sized(int type, int index_type, int count, void *const *outs, const void *const *bases, size_t base_size,
                              const ek_operand *index, const ek_operand *mask, size_t n) {
    if (int rc = ensure_init()) return rc;
    if (n == 0) return EK_OK;
    if (!outs || !bases) return fail(EK_ERR_INVALID, "ek_hip_gather_multi_sized(): null pointer");
    const size_t elem = type_size(type);
    int plan = ek_hip_gather_multi_plan(type, index_type, count, base_size, n);
    if (plan == EK_GATHER_RECORDS && (!index || !index->ptr)) plan = EK_GATHER_ONE_LAUNCH;     // one record for all lanes
    if (plan == EK_GATHER_PER_TABLE) {
        for (int c = 0; c < count; ++c)
            if (int rc = ek_hip_gather(type, index_type, outs[c], bases[c], index, mask, n)) return rc;
        return EK_OK;
    }
    if (plan == EK_GATHER_ONE_LAUNCH) return ek_hip_gather_multi(type, index_type, count, outs, bases, index, mask, n);
    RoctxRange range("enoki-hip: struct gather through staged records");
#define EK_RECORDS(T, C)                                                                                               \
    (index_type == EK_U32 ? gather_records_launch<T, uint32_t, C>(outs, bases, base_size, index, mask, n)              \
                          : gather_records_launch<T, int32_t, C>(outs, bases, base_size, index, mask, n))
    if (elem == 8) return EK_RECORDS(uint64_t, 2);
    switch (count) {
        case 2: return EK_RECORDS(uint32_t, 2);
        case 3: return EK_RECORDS(uint32_t, 3);
        default: return EK_RECORDS(uint32_t, 4);
    }
#undef EK_RECORDS
}

int ek_hip_scatter(int type, int index_type, void *base, const ek_operand *value, const ek_operand *index,
                   const ek_operand *mask, size_t n) {
    if (int rc = ensure_init()) return rc;
    if (n == 0) return EK_OK;
    if (!base) return fail(EK_ERR_INVALID, "ek_hip_scatter(): null pointer");
    switch (type_size(type)) {
        case 1: EK_INDEX_SWITCH(index_type, (scatter_launch<uint8_t, I, false>(base, value, index, mask, n)), "ek_hip_scatter()")
        case 4: EK_INDEX_SWITCH(index_type, (scatter_launch<uint32_t, I, false>(base, value, index, mask, n)), "ek_hip_scatter()")
        case 8: EK_INDEX_SWITCH(index_type, (scatter_launch<uint64_t, I, false>(base, value, index, mask, n)), "ek_hip_scatter()")
        default: return fail(EK_ERR_INVALID, "ek_hip_scatter(): unknown type %d", type);
    }
}

/// 64-bit index ARRAYS into a table of known size <= 2^32 are narrowed once (12 bytes per element, streamed) so that a
/// large scatter_add can take the binned / sorted paths, which are written for 32-bit indices.  The reference's tape
/// records the offsets of every gather as Int64 (autodiff.cpp:355-366, 524-535), so the adjoint of a gather arrives
/// here with 64-bit indices when the reference's own autodiff layer drives this library;
/// the device atomics it would otherwise fall back to are ~8x slower on large inputs.
static bool narrowable_index(int index_type, const ek_operand *index, size_t base_size, size_t n) {
    return (index_type == EK_I64 || index_type == EK_U64) && index && index->ptr && index->size == n &&
           base_size != 0 && base_size <= ((size_t) 1 << 32) && n >= ((size_t) 1 << 18) && n < ((size_t) 1 << 32);
}

struct NarrowedIndex {
    void *ptr = nullptr;
    ek_operand operand { nullptr, 0, 0 };
    int make(int index_type, const ek_operand *index, size_t n) {
        if (int rc = ek_hip_malloc(n * sizeof(uint32_t), &ptr)) return rc;
        if (int rc = ek_hip_cast(index_type, EK_U32, ptr, index, n)) return rc;
        operand = ek_operand{ ptr, 0, n };
        return EK_OK;
    }
    ~NarrowedIndex() { if (ptr) ek_hip_free(ptr); }       // stream-ordered reuse: the consumers are already enqueued
};

int ek_hip_scatter_add(int type, int index_type, void *base, size_t base_size, const ek_operand *value,
                       const ek_operand *index, const ek_operand *mask, size_t n, int mode) {
    if (int rc = ensure_init()) return rc;
    if (n == 0) return EK_OK;
    if (!base) return fail(EK_ERR_INVALID, "ek_hip_scatter_add(): null pointer");
    if (narrowable_index(index_type, index, base_size, n) && type != EK_BOOL &&
        (mode == 1 || ctx().tuning.deterministic ||
         (ctx().tuning.scatter_add_binned && scatter_add_binned_applicable(base_size, n, true, type_size(type))))) {
        NarrowedIndex narrow;
        if (int rc = narrow.make(index_type, index, n)) return rc;
        return ek_hip_scatter_add(type, EK_U32, base, base_size, value, &narrow.operand, mask, n, mode);
    }
    bool is_fp = type == EK_F32 || type == EK_F64;
    // the process-wide switch (ENOKI_HIP_DETERMINISTIC / tuning "deterministic") PROMOTES mode 0; an explicit mode 1 is a demand
    const bool promoted = mode == 0 && ctx().tuning.deterministic;
    if (promoted) mode = 1;
    if (mode == 1 && is_fp) {
        // deterministic: bit-identical to the CPU reference's element-order accumulation
        const bool sortable = index && mask && value && index->ptr != nullptr && index->size == n && base_size != 0 &&
                              (index_type == EK_U32 || index_type == EK_I32) && n < ((size_t) 1 << 32);
        if (!sortable && !promoted)
            return fail(EK_ERR_UNSUPPORTED, "ek_hip_scatter_add(): deterministic mode needs a 32-bit index ARRAY of size n "
                                            "and the target size (base_size)");
        if (!sortable) {
            // promoted by the global switch but not expressible as a sort (64-bit / broadcast index, unknown target size):
            // programs that work in the default mode keep working -- this call takes the unordered path
            if (ctx().log_level >= 1)
                fprintf(stderr, "enoki-hip: scatter_add(): deterministic order not available for this call (needs a 32-bit "
                                "index array and the target size); using the unordered path\n");
            mode = 0;
        }
    }
    if (mode == 1 && is_fp) {
        Arg<uint8_t> mm;
        if (int rc = make_arg<uint8_t>(mask, n, mm, "ek_hip_scatter_add")) return rc;
#define EK_SORTED_CALL(T, I)                                                                                          \
        do {                                                                                                          \
            Arg<T> vv; Arg<I> ii;                                                                                     \
            if (int rc = make_arg<T>(value, n, vv, "ek_hip_scatter_add")) return rc;                                  \
            if (int rc = make_arg<I>(index, n, ii, "ek_hip_scatter_add")) return rc;                                  \
            return scatter_add_sorted<T, I>((T *) base, base_size, vv, ii, mm, n);                                    \
        } while (0)
        if (type == EK_F32) { if (index_type == EK_U32) EK_SORTED_CALL(float, uint32_t); else EK_SORTED_CALL(float, int32_t); }
        else                { if (index_type == EK_U32) EK_SORTED_CALL(double, uint32_t); else EK_SORTED_CALL(double, int32_t); }
#undef EK_SORTED_CALL
    }
    // large inputs into tables that fit 256 LDS buckets: partition + ds_add instead of global atomics
    if (mode == 0 && ctx().tuning.scatter_add_binned && index && mask && value &&
        scatter_add_binned_applicable(base_size, n, index->ptr != nullptr && index->size == n, type_size(type)) &&
        (index_type == EK_U32 || index_type == EK_I32) && type != EK_BOOL) {
        Arg<uint8_t> mm;
        if (int rc = make_arg<uint8_t>(mask, n, mm, "ek_hip_scatter_add")) return rc;
#define EK_BINNED_CALL(T)                                                                                              \
        do {                                                                                                          \
            Arg<T> vv;                                                                                                \
            if (int rc = make_arg<T>(value, n, vv, "ek_hip_scatter_add")) return rc;                                  \
            if (index_type == EK_U32) { Arg<uint32_t> ii; if (int rc = make_arg<uint32_t>(index, n, ii, "ek_hip_scatter_add")) return rc; return scatter_add_binned<T, uint32_t>((T *) base, base_size, vv, ii, mm, n); } \
            else                      { Arg<int32_t> ii;  if (int rc = make_arg<int32_t>(index, n, ii, "ek_hip_scatter_add")) return rc;  return scatter_add_binned<T, int32_t>((T *) base, base_size, vv, ii, mm, n); }  \
        } while (0)
        if (type == EK_F64) EK_BINNED_CALL(double);
        if (type == EK_I64 || type == EK_U64) EK_BINNED_CALL(uint64_t);
#undef EK_BINNED_CALL
        if (type == EK_F32) {
            Arg<float> vv;
            if (int rc = make_arg<float>(value, n, vv, "ek_hip_scatter_add")) return rc;
            if (index_type == EK_U32) { Arg<uint32_t> ii; if (int rc = make_arg<uint32_t>(index, n, ii, "ek_hip_scatter_add")) return rc; return scatter_add_binned<float, uint32_t>((float *) base, base_size, vv, ii, mm, n); }
            else                      { Arg<int32_t> ii;  if (int rc = make_arg<int32_t>(index, n, ii, "ek_hip_scatter_add")) return rc;  return scatter_add_binned<float, int32_t>((float *) base, base_size, vv, ii, mm, n); }
        } else {
            Arg<uint32_t> vv;
            if (int rc = make_arg<uint32_t>(value, n, vv, "ek_hip_scatter_add")) return rc;
            if (index_type == EK_U32) { Arg<uint32_t> ii; if (int rc = make_arg<uint32_t>(index, n, ii, "ek_hip_scatter_add")) return rc; return scatter_add_binned<uint32_t, uint32_t>((uint32_t *) base, base_size, vv, ii, mm, n); }
            else                      { Arg<int32_t> ii;  if (int rc = make_arg<int32_t>(index, n, ii, "ek_hip_scatter_add")) return rc;  return scatter_add_binned<uint32_t, int32_t>((uint32_t *) base, base_size, vv, ii, mm, n); }
        }
    }
    switch (type) {
        case EK_F32: EK_INDEX_SWITCH(index_type, (scatter_launch<float, I, true>(base, value, index, mask, n)), "ek_hip_scatter_add()")
        case EK_F64: EK_INDEX_SWITCH(index_type, (scatter_launch<double, I, true>(base, value, index, mask, n)), "ek_hip_scatter_add()")
        case EK_I32: case EK_U32:
            EK_INDEX_SWITCH(index_type, (scatter_launch<uint32_t, I, true>(base, value, index, mask, n)), "ek_hip_scatter_add()")
        case EK_I64: case EK_U64:
            EK_INDEX_SWITCH(index_type, (scatter_launch<uint64_t, I, true>(base, value, index, mask, n)), "ek_hip_scatter_add()")
        default: return fail(EK_ERR_UNSUPPORTED, "ek_hip_scatter_add(): unsupported type %d", type);
    }
}

} // extern "C"

namespace {
template <typename T, typename I, int C, bool Sorted = false>
int scatter_add_multi_fused(void *const *bases, size_t base_size, const ek_operand *const *values, const ek_operand *const *weights,
                            const ek_operand *index, const ek_operand *mask, size_t n, const int *ops = nullptr) {
    Arg<T> vv[C], ww[C];
    Arg<I> ii;
    Arg<uint8_t> mm;
    unsigned weighted = 0;
    T *tables[C];
    for (int c = 0; c < C; ++c) {
        if (int rc = make_arg<T>(values[c], n, vv[c], "ek_hip_scatter_add_multi")) return rc;
        ww[c] = Arg<T>{ nullptr, T(1), 0u };
        if (weights && weights[c]) {
            if (int rc = make_arg<T>(weights[c], n, ww[c], "ek_hip_scatter_add_multi")) return rc;
            weighted |= 1u << c;
        }
        tables[c] = (T *) bases[c];
    }
    if (int rc = make_arg<I>(index, n, ii, "ek_hip_scatter_add_multi")) return rc;
    if (int rc = make_arg<uint8_t>(mask, n, mm, "ek_hip_scatter_add_multi")) return rc;
    if constexpr (Sorted) return scatter_add_sorted_multi<T, I, C>(tables, base_size, vv, ww, weighted, ii, mm, n);
    else return scatter_add_binned_multi<T, I, C>(tables, base_size, vv, ww, weighted, ii, mm, n, ops);
}

// deterministic mode: 2 or 3 streams per sort (one stream goes through ek_hip_scatter_add)
template <typename T, typename I>
int scatter_add_multi_sorted_count(int count, void *const *bases, size_t base_size, const ek_operand *const *values,
                                   const ek_operand *const *weights, const ek_operand *index, const ek_operand *mask, size_t n) {
    switch (count) {
        case 2: return scatter_add_multi_fused<T, I, 2, true>(bases, base_size, values, weights, index, mask, n);
        case 3: return scatter_add_multi_fused<T, I, 3, true>(bases, base_size, values, weights, index, mask, n);
        default: {
            if (int rc = scatter_add_multi_fused<T, I, 2, true>(bases, base_size, values, weights, index, mask, n)) return rc;
            return scatter_add_multi_fused<T, I, 2, true>(bases + 2, base_size, values + 2, weights ? weights + 2 : nullptr, index, mask, n);
        }
    }
}

template <typename T, typename I>
int scatter_add_multi_fused_count(int count, void *const *bases, size_t base_size, const ek_operand *const *values,
                                  const ek_operand *const *weights, const ek_operand *index, const ek_operand *mask, size_t n,
                                  const int *ops = nullptr) {
    switch (count) {
        case 1: return scatter_add_multi_fused<T, I, 1>(bases, base_size, values, weights, index, mask, n, ops);
        case 2: return scatter_add_multi_fused<T, I, 2>(bases, base_size, values, weights, index, mask, n, ops);
        case 3: return scatter_add_multi_fused<T, I, 3>(bases, base_size, values, weights, index, mask, n, ops);
        default: {
            // four streams would cost the partition kernel its second workgroup per CU (152 VGPRs): run 2 + 2
            if (int rc = scatter_add_multi_fused<T, I, 2>(bases, base_size, values, weights, index, mask, n, ops)) return rc;
            return scatter_add_multi_fused<T, I, 2>(bases + 2, base_size, values + 2, weights ? weights + 2 : nullptr, index, mask, n,
                                                    ops ? ops + 2 : nullptr);
        }
    }
}
} // namespace

extern "C" {

int ek_hip_scatter_add_multi(int type, int index_type, int count, void *const *bases, size_t base_size,
                             const ek_operand *const *values, const ek_operand *const *weights, const ek_operand *index,
                             const ek_operand *mask, size_t n, int mode) {
    if (int rc = ensure_init()) return rc;
    if (count < 1 || count > 4) return fail(EK_ERR_INVALID, "ek_hip_scatter_add_multi(): 1 to 4 streams expected, got %d", count);
    if (!bases || !values || !index || !mask) return fail(EK_ERR_INVALID, "ek_hip_scatter_add_multi(): null pointer");
    for (int c = 0; c < count; ++c) {
        if (!bases[c] || !values[c]) return fail(EK_ERR_INVALID, "ek_hip_scatter_add_multi(): null pointer");
        for (int d = 0; d < c; ++d)
            if (bases[d] == bases[c]) return fail(EK_ERR_INVALID, "ek_hip_scatter_add_multi(): the tables must be distinct");
    }
    if (n == 0) return EK_OK;
    if (narrowable_index(index_type, index, base_size, n) && type != EK_BOOL) {
        NarrowedIndex narrow;
        if (int rc = narrow.make(index_type, index, n)) return rc;
        return ek_hip_scatter_add_multi(type, EK_U32, count, bases, base_size, values, weights, &narrow.operand, mask, n, mode);
    }
    if (mode == 0 && ctx().tuning.deterministic) mode = 1;
    const bool fused = mode == 0 && ctx().tuning.scatter_add_binned && type != EK_BOOL &&
                       (index_type == EK_U32 || index_type == EK_I32) &&
                       scatter_add_binned_multi_applicable(base_size, n, index->ptr != nullptr && index->size == n, type_size(type));
    if (fused) {
        if (type == EK_F32) {
            if (index_type == EK_U32) return scatter_add_multi_fused_count<float, uint32_t>(count, bases, base_size, values, weights, index, mask, n);
            return scatter_add_multi_fused_count<float, int32_t>(count, bases, base_size, values, weights, index, mask, n);
        }
        if (type == EK_F64) {
            if (index_type == EK_U32) return scatter_add_multi_fused_count<double, uint32_t>(count, bases, base_size, values, weights, index, mask, n);
            return scatter_add_multi_fused_count<double, int32_t>(count, bases, base_size, values, weights, index, mask, n);
        }
        bool any_weight = false;
        for (int c = 0; c < count; ++c) any_weight = any_weight || (weights && weights[c]);
        if (!any_weight) {      // integer streams carry no gradients; the fused path takes them unweighted
            if (type_size(type) == 8) {
                if (index_type == EK_U32) return scatter_add_multi_fused_count<uint64_t, uint32_t>(count, bases, base_size, values, weights, index, mask, n);
                return scatter_add_multi_fused_count<uint64_t, int32_t>(count, bases, base_size, values, weights, index, mask, n);
            }
            if (index_type == EK_U32) return scatter_add_multi_fused_count<uint32_t, uint32_t>(count, bases, base_size, values, weights, index, mask, n);
            return scatter_add_multi_fused_count<uint32_t, int32_t>(count, bases, base_size, values, weights, index, mask, n);
        }
    }
    // deterministic mode with several fp streams: one stable sort of the keys carries all value streams
    if (mode == 1 && count >= 2 && (type == EK_F32 || type == EK_F64) && (index_type == EK_U32 || index_type == EK_I32) &&
        index->ptr != nullptr && index->size == n && base_size > 0 && n < ((size_t) 1 << 32)) {
        if (type == EK_F32) {
            if (index_type == EK_U32) return scatter_add_multi_sorted_count<float, uint32_t>(count, bases, base_size, values, weights, index, mask, n);
            return scatter_add_multi_sorted_count<float, int32_t>(count, bases, base_size, values, weights, index, mask, n);
        }
        if (index_type == EK_U32) return scatter_add_multi_sorted_count<double, uint32_t>(count, bases, base_size, values, weights, index, mask, n);
        return scatter_add_multi_sorted_count<double, int32_t>(count, bases, base_size, values, weights, index, mask, n);
    }
    // everything else: one scatter_add per stream, products materialised first
    for (int c = 0; c < count; ++c) {
        const ek_operand *v = values[c];
        ek_operand product;
        void *tmp = nullptr;
        if (weights && weights[c]) {
            if (type != EK_F32 && type != EK_F64) return fail(EK_ERR_UNSUPPORTED, "ek_hip_scatter_add_multi(): weights need a floating point type");
            const size_t m = (values[c]->ptr && values[c]->size != 1) || (weights[c]->ptr && weights[c]->size != 1) ? n : 1;
            if (int rc = ek_hip_malloc(m * type_size(type), &tmp)) return rc;
            int rc = ek_hip_binary(EK_SAFE_MUL, type, tmp, weights[c], values[c], m);
            if (rc) { ek_hip_free(tmp); return rc; }
            product = ek_operand{ tmp, 0, m };
            v = &product;
        }
        int rc = ek_hip_scatter_add(type, index_type, bases[c], base_size, v, index, mask, n, mode);
        if (tmp) ek_hip_free(tmp);
        if (rc) return rc;
    }
    return EK_OK;
}

int ek_hip_scatter_add_multi_map(int type, int index_type, int count, void *const *bases, size_t base_size,
                                 const ek_operand *const *values, const int *value_ops, const ek_operand *const *weights,
                                 const ek_operand *index, const ek_operand *mask, size_t n, int mode) {
    if (int rc = ensure_init()) return rc;
    bool mapped = false;
    if (value_ops) {
        if (count < 1 || count > 4) return fail(EK_ERR_INVALID, "ek_hip_scatter_add_multi_map(): 1 to 4 streams expected, got %d", count);
        for (int c = 0; c < count; ++c) {
            if (value_ops[c] == EK_COPY) continue;
            if (!unary_fusable(value_ops[c]) || (type != EK_F32 && type != EK_F64))
                return fail(EK_ERR_UNSUPPORTED, "ek_hip_scatter_add_multi_map(): op %d cannot be applied on load", value_ops[c]);
            mapped = true;
        }
    }
    if (!mapped) return ek_hip_scatter_add_multi(type, index_type, count, bases, base_size, values, weights, index, mask, n, mode);
    if (!bases || !values || !index || !mask) return fail(EK_ERR_INVALID, "ek_hip_scatter_add_multi_map(): null pointer");
    for (int c = 0; c < count; ++c) {
        if (!bases[c] || !values[c]) return fail(EK_ERR_INVALID, "ek_hip_scatter_add_multi_map(): null pointer");
        for (int d = 0; d < c; ++d)
            if (bases[d] == bases[c]) return fail(EK_ERR_INVALID, "ek_hip_scatter_add_multi_map(): the tables must be distinct");
    }
    if (n == 0) return EK_OK;
    if (narrowable_index(index_type, index, base_size, n)) {
        NarrowedIndex narrow;
        if (int rc = narrow.make(index_type, index, n)) return rc;
        return ek_hip_scatter_add_multi_map(type, EK_U32, count, bases, base_size, values, value_ops, weights, &narrow.operand, mask, n, mode);
    }
    const bool deterministic = mode == 1 || (mode == 0 && ctx().tuning.deterministic);
    const bool fused = !deterministic && ctx().tuning.scatter_add_binned && (index_type == EK_U32 || index_type == EK_I32) &&
                       scatter_add_binned_multi_applicable(base_size, n, index->ptr != nullptr && index->size == n, type_size(type));
    if (fused) {
        if (type == EK_F32) {
            if (index_type == EK_U32) return scatter_add_multi_fused_count<float, uint32_t>(count, bases, base_size, values, weights, index, mask, n, value_ops);
            return scatter_add_multi_fused_count<float, int32_t>(count, bases, base_size, values, weights, index, mask, n, value_ops);
        }
        if (index_type == EK_U32) return scatter_add_multi_fused_count<double, uint32_t>(count, bases, base_size, values, weights, index, mask, n, value_ops);
        return scatter_add_multi_fused_count<double, int32_t>(count, bases, base_size, values, weights, index, mask, n, value_ops);
    }
    // paths without an on-load map (deterministic mode, small tables, atomics): evaluate the mapped streams first; streams
    // that share array and op share the temporary
    ek_operand mapped_value[4];
    const ek_operand *vv[4];
    void *tmp[4] = { nullptr, nullptr, nullptr, nullptr };
    int rc = EK_OK;
    for (int c = 0; c < count && rc == EK_OK; ++c) {
        vv[c] = values[c];
        if (value_ops[c] == EK_COPY) continue;
        int same = -1;
        for (int d = 0; d < c; ++d)
            if (tmp[d] && value_ops[d] == value_ops[c] && values[d]->ptr == values[c]->ptr && values[d]->size == values[c]->size) same = d;
        if (same >= 0) { vv[c] = vv[same]; continue; }
        const size_t m = values[c]->ptr && values[c]->size != 1 ? n : 1;
        rc = ek_hip_malloc(m * type_size(type), &tmp[c]);
        if (rc == EK_OK) rc = ek_hip_unary(value_ops[c], type, tmp[c], values[c], m);
        mapped_value[c] = ek_operand{ tmp[c], 0, m };
        vv[c] = &mapped_value[c];
    }
    if (rc == EK_OK) rc = ek_hip_scatter_add_multi(type, index_type, count, bases, base_size, vv, weights, index, mask, n, mode);
    for (int c = 0; c < count; ++c)
        if (tmp[c]) ek_hip_free(tmp[c]);
    return rc;
}

} // extern "C"
