/*
    enoki/random.h -- PCG32 pseudorandom number generator over HIPArray types

    Same interface as the reference's enoki::PCG32<T> (include/enoki/random.h:38-330): `seed`,
    `next_uint32/64`, `next_float32/64`, `next_uint32/64_bounded`, `advance`, `operator-`, public
    `state` / `inc`.  The algorithm is M. O'Neill's PCG XSH-RR 64/32 (pcg-random.org); the reference
    spells every draw as ~12 array operations that its JIT fuses -- here a draw is ONE kernel
    (`ek_hip_pcg32_next`, csrc/random.hip) that reads state + inc and writes state' + sample.
*/
#pragma once

#include <enoki/hip.h>

#define PCG32_DEFAULT_STATE  0x853c49e6748fea9bULL
#define PCG32_DEFAULT_STREAM 0xda3e39cb94b95bdbULL
#define PCG32_MULT           0x5851f42d4c957f2dULL

namespace enoki {

template <typename T> struct PCG32 {
    static_assert(is_array_v<T> && !T::IsDiff, "PCG32: instantiate with a HIPArray type, e.g. PCG32<HIPArray<float>>");
    using Int64     = typename T::template ReplaceValue<int64_t>;
    using UInt64    = typename T::template ReplaceValue<uint64_t>;
    using UInt32    = typename T::template ReplaceValue<uint32_t>;
    using Float64   = typename T::template ReplaceValue<double>;
    using Float32   = typename T::template ReplaceValue<float>;
    using Mask      = typename T::MaskType;
    using UInt32Mask = Mask;
    using UInt64Mask = Mask;

    PCG32(const UInt64 &initstate = UInt64(PCG32_DEFAULT_STATE), const UInt64 &initseq = UInt64(PCG32_DEFAULT_STREAM)) {
        seed(initstate, initseq);
    }

    /// State initializer + sequence selection constant (stream id), random.h:62-68
    void seed(const UInt64 &initstate, const UInt64 &initseq) {
        state = UInt64(uint64_t(0));
        inc = (initseq << UInt64(uint64_t(1))) | UInt64(uint64_t(1));
        next_uint32();
        state = state + initstate;
        next_uint32();
    }

    UInt32 next_uint32(const Mask &mask = Mask(true)) { return UInt32::pcg32_next_(EK_PCG32_UINT32, state, inc, mask); }
    UInt64 next_uint64(const Mask &mask = Mask(true)) { return UInt64::pcg32_next_(EK_PCG32_UINT64, state, inc, mask); }
    /// Uniform on [0, 1), 23 random mantissa bits
    Float32 next_float32(const Mask &mask = Mask(true)) { return Float32::pcg32_next_(EK_PCG32_FLOAT32, state, inc, mask); }
    /// Uniform on [0, 1), 32 random mantissa bits
    Float64 next_float64(const Mask &mask = Mask(true)) { return Float64::pcg32_next_(EK_PCG32_FLOAT64, state, inc, mask); }

    template <typename Value> Value next_uint(const Mask &mask = Mask(true)) {
        if constexpr (sizeof(scalar_t<Value>) == 8) return Value(next_uint64(mask));
        else return Value(next_uint32(mask));
    }

    template <typename Value> Value next_float(const Mask &mask = Mask(true)) {
        if constexpr (sizeof(scalar_t<Value>) == 8) return next_float64(mask);
        else return next_float32(mask);
    }

    /// Uniform integer in [0, bound) by rejection (random.h:153-199): lanes that drew below the threshold
    /// redraw; the others stop advancing their generator
    UInt32 next_uint32_bounded(uint32_t bound, Mask mask = Mask(true)) {
        const uint32_t threshold = (~bound + 1u) % bound;
        UInt32 result = UInt32(0u);
        do {
            result = select(mask, next_uint32(mask), result);
            mask = mask & (result < UInt32(threshold));
        } while (any(mask));
        return result % UInt32(bound);
    }

    /// next_uint32_bounded / next_uint64_bounded selected by the requested array type (random.h:248-256)
    template <typename Value> Value next_uint_bounded(scalar_t<Value> bound, Mask mask = Mask(true)) {
        if constexpr (sizeof(scalar_t<Value>) == 8) return Value(next_uint64_bounded((uint64_t) bound, mask));
        else return Value(next_uint32_bounded((uint32_t) bound, mask));
    }

    UInt64 next_uint64_bounded(uint64_t bound, Mask mask = Mask(true)) {
        const uint64_t threshold = (~bound + (uint64_t) 1) % bound;
        UInt64 result = UInt64(uint64_t(0));
        do {
            result = select(mask, next_uint64(mask), result);
            mask = mask & (result < UInt64(threshold));
        } while (any(mask));
        return result % UInt64(bound);
    }

    /// Jump ahead (or back, with a negative delta) in O(log delta): Brown, "Random Number Generation with
    /// Arbitrary Stride", 1994 (random.h:256-284)
    void advance(const Int64 &delta_) {
        UInt64 cur_mult = UInt64(uint64_t(PCG32_MULT)), cur_plus = inc, acc_mult = UInt64(uint64_t(1)),
               acc_plus = UInt64(uint64_t(0));
        UInt64 delta(delta_), one = UInt64(uint64_t(1)), zero_ = UInt64(uint64_t(0));
        while (any(neq(delta, zero_))) {
            Mask bit = neq(delta & one, zero_);
            acc_mult = select(bit, acc_mult * cur_mult, acc_mult);
            acc_plus = select(bit, acc_plus * cur_mult + cur_plus, acc_plus);
            cur_plus = (cur_mult + one) * cur_plus;
            cur_mult = cur_mult * cur_mult;
            delta = delta >> one;
        }
        state = acc_mult * state + acc_plus;
    }

    /// Distance between two generators on the same stream (random.h:287-309)
    Int64 operator-(const PCG32 &other) const {
        UInt64 cur_mult = UInt64(uint64_t(PCG32_MULT)), cur_plus = inc, cur_state = other.state,
               the_bit = UInt64(uint64_t(1)), distance = UInt64(uint64_t(0)), one = UInt64(uint64_t(1));
        while (any(neq(state, cur_state))) {
            Mask differs = neq(state & the_bit, cur_state & the_bit);
            cur_state = select(differs, cur_state * cur_mult + cur_plus, cur_state);
            distance = select(differs, distance | the_bit, distance);
            the_bit = the_bit << one;
            cur_plus = (cur_mult + one) * cur_plus;
            cur_mult = cur_mult * cur_mult;
        }
        return Int64(distance);
    }

    bool operator==(const PCG32 &other) const { return all(eq(state, other.state)) && all(eq(inc, other.inc)); }
    bool operator!=(const PCG32 &other) const { return !operator==(other); }

    UInt64 state;  // RNG state.  All values are possible.
    UInt64 inc;    // Controls which RNG sequence (stream) is selected.  Always odd.
};

} // namespace enoki
