// PCG32 (M. O'Neill, "PCG: A Family of Simple Fast Space-Efficient Statistically Good Algorithms for
// Random Number Generation", 2014; XSH-RR 64/32 variant) as ONE fused kernel per draw.
//
// The reference implements the generator generically over array types (include/enoki/random.h:38-330):
// on its CUDA backend every draw is ~12 traced u64/u32 ops that the JIT fuses.  Eager execution of the
// same composition would move ~200 B per sample through HBM; this kernel reads state + inc (+ mask) and
// writes the new state + the sample: 28-37 B per sample.
//
//   step      state' = mask ? state * 0x5851f42d4c957f2d + inc : state          (random.h:68-75, 78-84)
//   output    xorshifted = uint32(((old >> 18) ^ old) >> 27);  rot = uint32(old >> 59);
//             u32 = ror(xorshifted, rot)                                        (random.h:71-74)
//   float32   reinterpret((u32 >> 9) | 0x3f800000) - 1                          (random.h:112-114)
//   float64   reinterpret((uint64(u32) << 20) | 0x3ff0000000000000) - 1         (random.h:128-133)
//   uint64    two steps; first draw = high word (operand order of the pinned build)  (random.h:87-94)
// Integer work: results are bit-exact against the reference (tests/test_random_gpu.py).
#include "ek_map.h"

namespace ek {

static constexpr uint64_t kPcgMult = 0x5851f42d4c957f2dull;

__device__ __forceinline__ uint32_t pcg_output(uint64_t old) {
    uint32_t xorshifted = (uint32_t) (((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t) (old >> 59);
    return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
}

template <int Kind> struct pcg_out;
template <> struct pcg_out<EK_PCG32_UINT32> { using type = uint32_t; };
template <> struct pcg_out<EK_PCG32_FLOAT32> { using type = float; };
template <> struct pcg_out<EK_PCG32_UINT64> { using type = uint64_t; };
template <> struct pcg_out<EK_PCG32_FLOAT64> { using type = double; };

template <int Kind>
__global__ __launch_bounds__(256) void k_pcg32(typename pcg_out<Kind>::type *__restrict__ out,
                                               uint64_t *__restrict__ state_out, Arg<uint64_t> state,
                                               Arg<uint64_t> inc, Arg<uint8_t> mask, size_t n, int vec_ok) {
    using TO = typename pcg_out<Kind>::type;
    constexpr int N = 2;                                   // two u64 states = one 16-byte vector per lane
    const uint64_t ss = state.vec ? 0 : arg_scalar(state), si = inc.vec ? 0 : arg_scalar(inc);
    const uint8_t sm = mask.vec ? uint8_t(0) : arg_scalar(mask);
    const size_t e = lane_elem<N, 1>(0);
    if (e >= n) return;
    const bool fast = vec_ok && e + N <= n;
    Pack<uint64_t, N> ps = arg_load<uint64_t, N, true>(state, ss, e, n, fast);
    Pack<uint64_t, N> pi = arg_load<uint64_t, N, true>(inc, si, e, n, fast);
    Pack<uint8_t, N> pm = arg_load<uint8_t, N, true>(mask, sm, e, n, fast);
    Pack<TO, N> po;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        uint64_t s = ps.v[k];
        const bool m = pm.v[k] != 0;
        uint32_t u = pcg_output(s);
        if (m) s = s * kPcgMult + pi.v[k];
        if constexpr (Kind == EK_PCG32_UINT32) {
            po.v[k] = u;
        } else if constexpr (Kind == EK_PCG32_FLOAT32) {
            po.v[k] = __uint_as_float((u >> 9) | 0x3f800000u) - 1.0f;
        } else if constexpr (Kind == EK_PCG32_FLOAT64) {
            po.v[k] = __longlong_as_double((long long) (((uint64_t) u << 20) | 0x3ff0000000000000ull)) - 1.0;
        } else {
            // first draw -> HIGH word: `UInt64(next_uint32()) | sl<32>(UInt64(next_uint32()))` leaves the operand
            // order to the compiler, and the pinned g++ reference build evaluates the right operand first
            uint32_t second = pcg_output(s);
            if (m) s = s * kPcgMult + pi.v[k];
            po.v[k] = (uint64_t) second | ((uint64_t) u << 32);
        }
        ps.v[k] = s;
    }
    out_store<TO, N, true>(out, po, e, n, fast);
    out_store<uint64_t, N, true>(state_out, ps, e, n, fast);
}

template <int Kind>
int pcg32_launch(void *out, uint64_t *state_out, const Arg<uint64_t> &st, const Arg<uint64_t> &ic,
                 const Arg<uint8_t> &mk, size_t n) {
    using TO = typename pcg_out<Kind>::type;
    int vec_ok = aligned16(out) && aligned16(state_out) && arg_aligned(st) && arg_aligned(ic) &&
                 (!mk.vec || (reinterpret_cast<uintptr_t>(mk.ptr) & 1u) == 0);
    Context &c = ctx();
    unsigned grid = (unsigned) ((n + 511) / 512);
    hipLaunchKernelGGL((k_pcg32<Kind>), dim3(grid), dim3(256), 0, c.stream, (TO *) out, state_out, st, ic, mk, n,
                       vec_ok);
    EK_LAUNCH_CHECK("pcg32", n, n * (sizeof(TO) + sizeof(uint64_t)) + arg_bytes(st, n) + arg_bytes(ic, n) +
                                arg_bytes(mk, n));
    return EK_OK;
}

} // namespace ek

using namespace ek;

extern "C" int ek_hip_pcg32_next(int kind, void *out, uint64_t *state_out, const ek_operand *state,
                                 const ek_operand *inc, const ek_operand *mask, size_t n) {
    if (int rc_ = ensure_init()) return rc_;
    if (n == 0) return EK_OK;
    if (!out) return fail(EK_ERR_INVALID, "ek_hip_pcg32_next(): null output pointer");
    if (!state_out) return fail(EK_ERR_INVALID, "ek_hip_pcg32_next(): null state output");
    Arg<uint64_t> st, ic;
    Arg<uint8_t> mk;
    if (int rc = make_arg<uint64_t>(state, n, st, "ek_hip_pcg32_next")) return rc;
    if (int rc = make_arg<uint64_t>(inc, n, ic, "ek_hip_pcg32_next")) return rc;
    if (int rc = make_arg<uint8_t>(mask, n, mk, "ek_hip_pcg32_next")) return rc;
    switch (kind) {
        case EK_PCG32_UINT32: return pcg32_launch<EK_PCG32_UINT32>(out, state_out, st, ic, mk, n);
        case EK_PCG32_FLOAT32: return pcg32_launch<EK_PCG32_FLOAT32>(out, state_out, st, ic, mk, n);
        case EK_PCG32_UINT64: return pcg32_launch<EK_PCG32_UINT64>(out, state_out, st, ic, mk, n);
        case EK_PCG32_FLOAT64: return pcg32_launch<EK_PCG32_FLOAT64>(out, state_out, st, ic, mk, n);
        default: return fail(EK_ERR_INVALID, "ek_hip_pcg32_next(): unknown kind %d", kind);
    }
}
