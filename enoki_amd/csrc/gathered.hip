// Vertical ops that consume a gather in place: out[i] = op(..., mask[i] ? table[index[i]] : 0, ...).
//
// The reference never materialises `gather(A, idx)` that feeds an arithmetic op: its JIT emits the `ld.global` of
// cuda.h:845-864 into the same kernel as the consumer (jit.cu:1066-1217).  An eager backend pays 4 B/elt to write the
// gathered array and 4 B/elt to read it back.  HIPArray therefore keeps a large gather *deferred* (include/enoki/hip.h)
// and hands it to the first arithmetic consumer, which runs one of the kernels below:
//
//   slot   binary ADD / SUB / MUL         ternary FMADD / FMSUB / FNMADD / FNMSUB
//   A      op(G, s0)                      fma(G, s0, s1)
//   B      op(s0, G)   (SUB only)         --            (the product commutes: the caller swaps)
//   C      --                             fma(s0, s1, G)
//   PAIR   --                             fma(Ga, s0, Gc), Ga and Gc gathered through the SAME index / mask arrays
//
// PAIR is the shape of a parameter lookup `fmadd(gather(A, idx), x, gather(B, idx))` (BASELINE config 3b).  Random
// 4-byte lookups are bound by the REQUEST rate of the vector memory pipeline / L2 (~190 G requests/s for a 4 MiB
// table, profiles/rocprof_l2_r01.txt), not by bytes, so the two tables are first interleaved into one table of
// {A[k], B[k]} records (K * 16 B of traffic, microseconds) and every element issues ONE 8-byte request instead of two
// 4-byte ones (profiles/probe_gather_pair_r02.txt: 0.67 ms instead of 0.93 ms for 64 Mi elements, K = 1 Mi).
#include "ek_map.h"

namespace ek {

enum { G_SLOT_A = 0, G_SLOT_B = 1, G_SLOT_C = 2, G_SLOT_PAIR = 3 };

template <typename T> struct GArg {
    const T *table;          // slot A/B/C: the table;  PAIR: interleaved {a, c} records
    Arg<uint32_t> index;
    Arg<uint8_t> mask;
};

template <int Op, typename T> struct GBinary {
    static __device__ __forceinline__ T apply(T x, T y) {
        if constexpr (Op == EK_ADD) return x + y;
        else if constexpr (Op == EK_SUB) return x - y;
        else return x * y;
    }
};

template <int Op, typename T> struct GTernary {
    static __device__ __forceinline__ T fma_(T a, T b, T c) {
        if constexpr (sizeof(T) == 4) return __builtin_fmaf(a, b, c); else return __builtin_fma(a, b, c);
    }
    static __device__ __forceinline__ T apply(T x, T y, T z) {
        if constexpr (Op == EK_FMADD) return fma_(x, y, z);
        else if constexpr (Op == EK_FMSUB) return fma_(x, y, -z);
        else if constexpr (Op == EK_FNMADD) return fma_(-x, y, z);
        // the operator spellings: a product and a sum with a rounding each (-ffp-contract=off: never fused)
        else if constexpr (Op == EK_MULADD) return x * y + z;
        else if constexpr (Op == EK_MULSUB) return x * y - z;
        else if constexpr (Op == EK_NMULADD) return z - x * y;
        else return fma_(-x, y, -z);
    }
};

// One 16-byte vector of the output per lane (N = 4 floats / 2 doubles); the index vector is N * 4 bytes.
template <typename F, typename T, int Arity, int Slot>
__global__ __launch_bounds__(256) void k_map_gathered(T *__restrict__ out, size_t n, int vec_ok, GArg<T> g, Arg<T> s0, Arg<T> s1) {
    constexpr int N = 16 / sizeof(T);
    const uint8_t sm = g.mask.vec ? uint8_t(0) : arg_scalar(g.mask);
    const uint32_t si = g.index.vec ? 0u : arg_scalar(g.index);         // n == 1: the index "array" is a single entry
    const T v0 = s0.vec ? T(0) : arg_scalar(s0);
    const T v1 = (Arity == 3 && !s1.vec) ? arg_scalar(s1) : T(0);
    const size_t e = lane_elem<N, 1>(0);
    if (e >= n) return;
    const bool fast = vec_ok && e + N <= n;
    Pack<uint32_t, N> pi = arg_load<uint32_t, N, true>(g.index, si, e, n, fast);
    Pack<uint8_t, N> pm = arg_load<uint8_t, N, true>(g.mask, sm, e, n, fast);
    Pack<T, N> ga, gc;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const bool on = pm.v[k] && e + k < n;
        if constexpr (Slot == G_SLOT_PAIR) {
            Pack<T, 2> rec;
            rec.v[0] = T(0); rec.v[1] = T(0);
            if (on) rec = pack_load<T, 2, false>(g.table + 2 * (size_t) pi.v[k]);
            ga.v[k] = rec.v[0];
            gc.v[k] = rec.v[1];
        } else {
            ga.v[k] = on ? g.table[pi.v[k]] : T(0);
        }
    }
    Pack<T, N> p0 = arg_load<T, N, true>(s0, v0, e, n, fast), p1;
    if constexpr (Arity == 3 && Slot != G_SLOT_PAIR) p1 = arg_load<T, N, true>(s1, v1, e, n, fast);
    Pack<T, N> po;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if constexpr (Arity == 2) po.v[k] = Slot == G_SLOT_A ? F::apply(ga.v[k], p0.v[k]) : F::apply(p0.v[k], ga.v[k]);
        else if constexpr (Slot == G_SLOT_A) po.v[k] = F::apply(ga.v[k], p0.v[k], p1.v[k]);
        else if constexpr (Slot == G_SLOT_C) po.v[k] = F::apply(p0.v[k], p1.v[k], ga.v[k]);
        else po.v[k] = F::apply(ga.v[k], p0.v[k], gc.v[k]);
    }
    out_store<T, N, true>(out, po, e, n, fast);
}

// {a[k], b[k]} records.  Four entries per lane: two 16-byte loads, two 16-byte stores (f32); the scalar body covers the
// tail and tables that are not 16-byte aligned.
template <typename T>
__global__ __launch_bounds__(256) void k_interleave2(Pack<T, 2> *__restrict__ out, const T *__restrict__ a, const T *__restrict__ b, size_t k,
                                                     int vec_ok) {
    constexpr int N = 16 / sizeof(T);
    const size_t e = ((size_t) blockIdx.x * 256 + threadIdx.x) * N;
    if (e >= k) return;
    if (vec_ok && e + N <= k) {
        Pack<T, N> pa = pack_load<T, N, false>(a + e), pb = pack_load<T, N, false>(b + e);
        if constexpr (N == 4) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                Pack<T, 4> r;
                r.v[0] = pa.v[2 * h]; r.v[1] = pb.v[2 * h]; r.v[2] = pa.v[2 * h + 1]; r.v[3] = pb.v[2 * h + 1];
                pack_store<T, 4, false>(reinterpret_cast<T *>(out + e) + 4 * h, r);
            }
        } else {                     // 8-byte elements: one record per 16-byte store
            Pack<T, 2> r0, r1;
            r0.v[0] = pa.v[0]; r0.v[1] = pb.v[0];
            r1.v[0] = pa.v[1]; r1.v[1] = pb.v[1];
            out[e] = r0;
            out[e + 1] = r1;
        }
    } else {
        for (size_t i = e; i < k && i < e + N; ++i) {
            Pack<T, 2> r;
            r.v[0] = a[i];
            r.v[1] = b[i];
            out[i] = r;
        }
    }
}

static int make_garg_common(const ek_gathered *g, size_t n, Arg<uint32_t> &index, Arg<uint8_t> &mask) {
    if (!g->table) return fail(EK_ERR_INVALID, "ek_hip_map_gathered(): null table");
    if (g->index_type != EK_U32 && g->index_type != EK_I32)
        return fail(EK_ERR_UNSUPPORTED, "ek_hip_map_gathered(): 32-bit index arrays only");
    if (!g->index.ptr || g->index.size != n)
        return fail(EK_ERR_UNSUPPORTED, "ek_hip_map_gathered(): the index must be an array of the op's size");
    // valid int32 indices are non-negative: same bits as uint32
    if (int rc = make_arg<uint32_t>(&g->index, n, index, "ek_hip_map_gathered")) return rc;
    return make_arg<uint8_t>(&g->mask, n, mask, "ek_hip_map_gathered");
}

template <typename F, typename T, int Arity, int Slot>
int launch_gathered(const char *name, T *out, size_t n, const GArg<T> &g, const Arg<T> &s0, const Arg<T> &s1) {
    constexpr int N = 16 / sizeof(T);
    int vec_ok = aligned16(out) && arg_aligned(g.index) && arg_aligned(g.mask) && arg_aligned(s0) && arg_aligned(s1);
    unsigned grid = oneshot_grid<N, 1>(n);
    hipLaunchKernelGGL((k_map_gathered<F, T, Arity, Slot>), dim3(grid), dim3(256), 0, ctx().stream, out, n, vec_ok, g, s0, s1);
    // algorithmic bytes: index + mask + one table element per lookup (SURVEY 8d prices the cache-resident table read at
    // sizeof(T)) + streamed operands + output
    const size_t lookups = Slot == G_SLOT_PAIR ? 2 : 1;
    EK_LAUNCH_CHECK(name, n, arg_bytes(g.index, n) + arg_bytes(g.mask, n) + lookups * n * sizeof(T) + arg_bytes(s0, n) +
                             arg_bytes(s1, n) + n * sizeof(T));
    return EK_OK;
}

template <typename T>
int map_gathered(int arity, int op, void *out, const ek_operand *const *o, const ek_gathered *const *g, size_t n) {
    const Arg<T> none{ nullptr, T(0), 0u };
    GArg<T> ga;
    Arg<T> s0 = none, s1 = none;
    T *outp = (T *) out;
    if (arity == 2) {
        const int slot = g[0] ? G_SLOT_A : G_SLOT_B;
        if ((g[0] != nullptr) == (g[1] != nullptr))
            return fail(EK_ERR_UNSUPPORTED, "ek_hip_map_gathered(): exactly one gathered operand expected");
        const ek_gathered *gg = g[slot];
        if (int rc = make_garg_common(gg, n, ga.index, ga.mask)) return rc;
        ga.table = (const T *) gg->table;
        if (int rc = make_arg<T>(o[1 - slot], n, s0, "ek_hip_map_gathered")) return rc;
        if (slot == G_SLOT_A) {
            switch (op) {
                case EK_ADD: return launch_gathered<GBinary<EK_ADD, T>, T, 2, G_SLOT_A>("gather_add", outp, n, ga, s0, s1);
                case EK_SUB: return launch_gathered<GBinary<EK_SUB, T>, T, 2, G_SLOT_A>("gather_sub", outp, n, ga, s0, s1);
                case EK_MUL: return launch_gathered<GBinary<EK_MUL, T>, T, 2, G_SLOT_A>("gather_mul", outp, n, ga, s0, s1);
                default: break;
            }
        } else {
            switch (op) {      // a + G and a * G commute bit for bit
                case EK_ADD: return launch_gathered<GBinary<EK_ADD, T>, T, 2, G_SLOT_A>("gather_add", outp, n, ga, s0, s1);
                case EK_MUL: return launch_gathered<GBinary<EK_MUL, T>, T, 2, G_SLOT_A>("gather_mul", outp, n, ga, s0, s1);
                case EK_SUB: return launch_gathered<GBinary<EK_SUB, T>, T, 2, G_SLOT_B>("gather_sub", outp, n, ga, s0, s1);
                default: break;
            }
        }
        return fail(EK_ERR_UNSUPPORTED, "ek_hip_map_gathered(): binary op %d cannot consume a gather", op);
    }
    if (arity != 3) return fail(EK_ERR_INVALID, "ek_hip_map_gathered(): arity must be 2 or 3");
    const bool two_roundings = op == EK_MULADD || op == EK_MULSUB || op == EK_NMULADD;
    if (op != EK_FMADD && op != EK_FMSUB && op != EK_FNMADD && op != EK_FNMSUB && !two_roundings)
        return fail(EK_ERR_UNSUPPORTED, "ek_hip_map_gathered(): ternary op %d cannot consume a gather", op);
    if (two_roundings && !(g[2] && (g[0] || g[1])))
        return fail(EK_ERR_UNSUPPORTED, "ek_hip_map_gathered(): op %d takes a gathered factor AND a gathered addend", op);
    const ek_operand *oa = o[0], *ob = o[1];
    const ek_gathered *g0 = g[0], *g1 = g[1], *g2 = g[2];
    if (g0 && g1) return fail(EK_ERR_UNSUPPORTED, "ek_hip_map_gathered(): both factors of the product are gathered");
    if (g1) { g0 = g1; g1 = nullptr; ob = oa; oa = nullptr; }         // fma(a, G, c) == fma(G, a, c)
    int slot;
    void *pair_table = nullptr;
    if (g0 && g2) {
        // one 8-byte lookup from the interleaved {a, c} table: needs the same index and mask arrays and equal table sizes
        if (g0->index.ptr != g2->index.ptr || g0->index.size != g2->index.size || g0->index_type != g2->index_type ||
            g0->mask.ptr != g2->mask.ptr || g0->mask.imm != g2->mask.imm || g0->mask.size != g2->mask.size ||
            g0->table_size != g2->table_size || g0->table_size == 0)
            return fail(EK_ERR_UNSUPPORTED, "ek_hip_map_gathered(): a gathered pair must share index, mask and table size");
        slot = G_SLOT_PAIR;
        if (int rc = make_garg_common(g0, n, ga.index, ga.mask)) return rc;
        const size_t K = g0->table_size;
        if (int rc = ek_hip_malloc(2 * K * sizeof(T), &pair_table)) return rc;
        constexpr size_t per_block = 256 * (16 / sizeof(T));
        hipLaunchKernelGGL((k_interleave2<T>), dim3((unsigned) ((K + per_block - 1) / per_block)), dim3(256), 0, ctx().stream,
                           (Pack<T, 2> *) pair_table, (const T *) g0->table, (const T *) g2->table, K,
                           (int) (aligned16(g0->table) && aligned16(g2->table)));
        note_launch("gather_interleave", K, 4 * K * sizeof(T));
        ga.table = (const T *) pair_table;
        if (int rc = make_arg<T>(ob, n, s0, "ek_hip_map_gathered")) { ek_hip_free(pair_table); return rc; }
    } else if (g0) {
        slot = G_SLOT_A;
        if (int rc = make_garg_common(g0, n, ga.index, ga.mask)) return rc;
        ga.table = (const T *) g0->table;
        if (int rc = make_arg<T>(ob, n, s0, "ek_hip_map_gathered")) return rc;
        if (int rc = make_arg<T>(o[2], n, s1, "ek_hip_map_gathered")) return rc;
    } else if (g2) {
        slot = G_SLOT_C;
        if (int rc = make_garg_common(g2, n, ga.index, ga.mask)) return rc;
        ga.table = (const T *) g2->table;
        if (int rc = make_arg<T>(oa, n, s0, "ek_hip_map_gathered")) return rc;
        if (int rc = make_arg<T>(ob, n, s1, "ek_hip_map_gathered")) return rc;
    } else {
        return fail(EK_ERR_INVALID, "ek_hip_map_gathered(): no gathered operand");
    }
    int rc = EK_ERR_INVALID;
#define EK_G3(OP, NAME)                                                                                                       \
    case OP:                                                                                                                  \
        rc = slot == G_SLOT_A ? launch_gathered<GTernary<OP, T>, T, 3, G_SLOT_A>("gather_" NAME, outp, n, ga, s0, s1)         \
           : slot == G_SLOT_C ? launch_gathered<GTernary<OP, T>, T, 3, G_SLOT_C>("gather_" NAME, outp, n, ga, s0, s1)         \
                              : launch_gathered<GTernary<OP, T>, T, 3, G_SLOT_PAIR>("gather_pair_" NAME, outp, n, ga, s0, s1); \
        break;
#define EK_G3_PAIR(OP, NAME) case OP: rc = launch_gathered<GTernary<OP, T>, T, 3, G_SLOT_PAIR>("gather_pair_" NAME, outp, n, ga, s0, s1); break;
    switch (op) {
        EK_G3(EK_FMADD, "fmadd") EK_G3(EK_FMSUB, "fmsub") EK_G3(EK_FNMADD, "fnmadd") EK_G3(EK_FNMSUB, "fnmsub")
        EK_G3_PAIR(EK_MULADD, "muladd") EK_G3_PAIR(EK_MULSUB, "mulsub") EK_G3_PAIR(EK_NMULADD, "nmuladd")
        default: break;
    }
#undef EK_G3_PAIR
#undef EK_G3
    if (pair_table) ek_hip_free(pair_table);     // stream-ordered reuse: the kernel above is already enqueued
    return rc;
}

} // namespace ek

using namespace ek;

extern "C" int ek_hip_map_gathered(int arity, int op, int type, void *out, const ek_operand *const *operands,
                                   const ek_gathered *const *gathered, size_t n) {
    if (int rc = ensure_init()) return rc;
    if (n == 0) return EK_OK;
    RoctxRange range("enoki-hip: gather consumed in place");
    if (!out || !operands || !gathered) return fail(EK_ERR_INVALID, "ek_hip_map_gathered(): null pointer");
    for (int k = 0; k < arity && k < 3; ++k)
        if (!operands[k] && !gathered[k]) return fail(EK_ERR_INVALID, "ek_hip_map_gathered(): operand %d is missing", k);
    if (type == EK_F32) return map_gathered<float>(arity, op, out, operands, gathered, n);
    if (type == EK_F64) return map_gathered<double>(arity, op, out, operands, gathered, n);
    return fail(EK_ERR_UNSUPPORTED, "ek_hip_map_gathered(): floating point types only");
}
