/*
    enoki/morton.h -- Morton (Z-order) codes for any unsigned array type (reference: include/enoki/morton.h)

        code = morton_encode(Array<UInt, D>(x_0, ..., x_{D-1}))      bit b of x_i lands on bit b * D + i
        Array<UInt, D> coords = morton_decode<Array<UInt, D>>(code)

    Each coordinate contributes its low floor(bits / D) bits (16 of 32 for D = 2, 10 for D = 3), like the reference's
    pdep / pext path.  The bits are spread and collected by the usual doubling steps -- x = (x | x << s) & mask with masks
    generated at compile time -- written with the array type's own shift / and / or, so the same code serves scalars,
    HIPArray<uint32_t / uint64_t> (a handful of integer kernels) and the packets of vectorize().  Integer work: bit-exact.
*/
#pragma once

#include <enoki/array.h>

namespace enoki {

namespace detail {
    /// bits j of a word whose block index (j / block) is a multiple of `dim`, at most floor(bits / dim) of them from the bottom
    template <typename S> constexpr S morton_mask(size_t dim, size_t block) {
        S m = 0;
        size_t taken = 0;
        const size_t bits = sizeof(S) * 8, most = bits / dim;
        for (size_t j = 0; j < bits; ++j)
            if ((j / block) % dim == 0 && taken < most) { m |= S(S(1) << j); ++taken; }
        return m;
    }
    template <size_t Dim, size_t Block, typename Value> inline Value morton_spread_step(Value x) {
        using S = scalar_t<Value>;
        if constexpr (Block == 0) {
            return x;
        } else {
            constexpr size_t shift = Block * (Dim - 1);
            if constexpr (shift > 0 && shift < sizeof(S) * 8) x = x | (x << Value(S(shift)));
            x = x & Value(morton_mask<S>(Dim, Block));
            return morton_spread_step<Dim, Block / 2>(x);
        }
    }
    template <size_t Dim, size_t Block, size_t Top, typename Value> inline Value morton_collect_step(Value x) {
        using S = scalar_t<Value>;
        if constexpr (Block > Top) {
            return x;
        } else {
            x = x & Value(morton_mask<S>(Dim, Block));
            constexpr size_t shift = Block * (Dim - 1);
            if constexpr (shift > 0 && shift < sizeof(S) * 8) x = x | (x >> Value(S(shift)));
            return morton_collect_step<Dim, Block * 2, Top>(x);
        }
    }
    template <size_t Dim, typename Value> inline Value morton_spread(const Value &x) {
        using S = scalar_t<Value>;
        constexpr size_t bits = sizeof(S) * 8, most = bits / Dim;
        constexpr S low = most >= bits ? S(~S(0)) : S((S(1) << most) - 1);
        if constexpr (Dim == 1) return x;
        else return morton_spread_step<Dim, bits / 2>(x & Value(low));
    }
    template <size_t Dim, typename Value> inline Value morton_collect(const Value &x) {
        using S = scalar_t<Value>;
        constexpr size_t bits = sizeof(S) * 8, most = bits / Dim;
        constexpr S low = most >= bits ? S(~S(0)) : S((S(1) << most) - 1);
        if constexpr (Dim == 1) return x;
        else return morton_collect_step<Dim, 1, bits / 2>(x) & Value(low);
    }
}

template <typename Coords, typename Value = value_t<Coords>> inline Value morton_encode(const Coords &a) {
    static_assert(std::is_unsigned_v<scalar_t<Coords>>, "morton_encode() requires unsigned arguments");
    using S = scalar_t<Value>;
    constexpr size_t D = Coords::Size;
    Value code = detail::morton_spread<D>(Value(a.coeff(0)));
    for (size_t i = 1; i < D; ++i) code = code | (detail::morton_spread<D>(Value(a.coeff(i))) << Value(S(i)));
    return code;
}

template <typename Coords, typename Value = value_t<Coords>> inline Coords morton_decode(const Value &code) {
    static_assert(std::is_unsigned_v<scalar_t<Coords>>, "morton_decode() requires unsigned arguments");
    using S = scalar_t<Value>;
    constexpr size_t D = Coords::Size;
    Coords r;
    r.coeff(0) = detail::morton_collect<D>(code);
    for (size_t i = 1; i < D; ++i) r.coeff(i) = detail::morton_collect<D>(code >> Value(S(i)));
    return r;
}

} // namespace enoki
