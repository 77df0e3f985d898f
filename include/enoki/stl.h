/*
    enoki/stl.h -- std::pair, std::tuple and std::array as structures of arrays (reference: include/enoki/stl.h)

    With this header included, pairs and tuples whose members are arrays behave like ENOKI_STRUCT types: zero / empty /
    slices / set_slices / gather / scatter / select work member by member, and enoki::vectorize() slices them as arguments
    and builds them as results -- `vectorize([](auto &&x) { return sincos(x); }, x)` returns a
    std::pair<HIPArray<float>, HIPArray<float>> computed by one fused kernel.  std::array<T, N> is covered the same way
    (N members of one type).
*/
#pragma once

#include <enoki/array.h>

#include <array>
#include <tuple>
#include <utility>

namespace enoki {

template <typename A, typename B> struct struct_support<std::pair<A, B>> {
    static constexpr bool Defined = true;
    using Value = std::pair<A, B>;
    static constexpr size_t leaf_count = dynamic_leaf_count<std::decay_t<A>>::value + dynamic_leaf_count<std::decay_t<B>>::value;
    template <typename F> static void apply(Value &v, F &&fn) { fn(v.first); fn(v.second); }
    template <typename F> static void apply(const Value &v, F &&fn) { fn(v.first); fn(v.second); }
    template <typename V2, typename F> static void apply2(Value &v, const V2 &w, F &&fn) { fn(v.first, w.first); fn(v.second, w.second); }
    template <typename V2, typename V3, typename F> static void apply3(Value &v, const V2 &w, const V3 &u, F &&fn) {
        fn(v.first, w.first, u.first); fn(v.second, w.second, u.second);
    }
};

template <typename... T> struct struct_support<std::tuple<T...>> {
    static constexpr bool Defined = true;
    using Value = std::tuple<T...>;
    static constexpr size_t leaf_count = (size_t(0) + ... + dynamic_leaf_count<std::decay_t<T>>::value);
    template <typename V, typename F, size_t... I> static void visit1(V &v, F &fn, std::index_sequence<I...>) { (fn(std::get<I>(v)), ...); }
    template <typename V, typename V2, typename F, size_t... I> static void visit2(V &v, const V2 &w, F &fn, std::index_sequence<I...>) {
        (fn(std::get<I>(v), std::get<I>(w)), ...);
    }
    template <typename V, typename V2, typename V3, typename F, size_t... I>
    static void visit3(V &v, const V2 &w, const V3 &u, F &fn, std::index_sequence<I...>) {
        (fn(std::get<I>(v), std::get<I>(w), std::get<I>(u)), ...);
    }
    template <typename F> static void apply(Value &v, F &&fn) { visit1(v, fn, std::index_sequence_for<T...>()); }
    template <typename F> static void apply(const Value &v, F &&fn) { visit1(v, fn, std::index_sequence_for<T...>()); }
    template <typename V2, typename F> static void apply2(Value &v, const V2 &w, F &&fn) { visit2(v, w, fn, std::index_sequence_for<T...>()); }
    template <typename V2, typename V3, typename F> static void apply3(Value &v, const V2 &w, const V3 &u, F &&fn) {
        visit3(v, w, u, fn, std::index_sequence_for<T...>());
    }
};

template <typename T, size_t N> struct struct_support<std::array<T, N>> {
    static constexpr bool Defined = true;
    using Value = std::array<T, N>;
    static constexpr size_t leaf_count = N * dynamic_leaf_count<std::decay_t<T>>::value;
    template <typename F> static void apply(Value &v, F &&fn) { for (size_t i = 0; i < N; ++i) fn(v[i]); }
    template <typename F> static void apply(const Value &v, F &&fn) { for (size_t i = 0; i < N; ++i) fn(v[i]); }
    template <typename V2, typename F> static void apply2(Value &v, const V2 &w, F &&fn) { for (size_t i = 0; i < N; ++i) fn(v[i], w[i]); }
    template <typename V2, typename V3, typename F> static void apply3(Value &v, const V2 &w, const V3 &u, F &&fn) {
        for (size_t i = 0; i < N; ++i) fn(v[i], w[i], u[i]);
    }
};

} // namespace enoki
