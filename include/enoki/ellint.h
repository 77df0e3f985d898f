/*
    enoki/ellint.h -- Carlson symmetric forms and the Legendre elliptic integrals built on them
                      (reference: include/enoki/special.h:314-672; included by <enoki/special.h>)

        carlson_rf(x, y, z)      R_F = 1/2 int_0^inf ((t+x)(t+y)(t+z))^(-1/2) dt
        carlson_rd(x, y, z)      R_D = 3/2 int_0^inf (t+x)^(-1/2) (t+y)^(-1/2) (t+z)^(-3/2) dt
        carlson_rc(x, y)         R_C = 1/2 int_0^inf (t+x)^(-1/2) (t+y)^(-1) dt
        carlson_rj(x, y, z, r)   R_J = 3/2 int_0^inf ((t+x)(t+y)(t+z))^(-1/2) (t+r)^(-1) dt
        ellint_1/2/3, comp_ellint_1/2/3 (k enters SQUARED, as in the reference and in std::ellint_*)

    Algorithm: B. C. Carlson, "Computing elliptic integrals by duplication", Numer. Math. 33 (1979): the arguments are
    pulled towards their mean by the duplication theorem (x <- (x + lambda) / 4) until the relative deviations drop below
    eps^(1/6)-type thresholds, then a short Taylor series in the deviations finishes (coefficient groupings as in
    Numerical Recipes, 3rd ed., section 6.12, which the reference also uses).  The functions take and return VALUES, not
    vectors of values: every array flavour of this backend works -- HIPArray (one kernel per operation; the convergence
    test reads one flag back per round, at most 10), DiffArray (differentiates through the iteration like the reference),
    and the one-element packets of enoki::vectorize(), where the whole integral is ONE fused kernel with a per-lane loop:

        FloatC K = vectorize([](auto &&phi, auto &&k) { return ellint_1(phi, k); }, phi, k);

    Parity: every step is evaluated in the reference's operation order (separately rounded products, fused multiply-adds
    only where the reference spells fmadd / dot), so float64 results are bit-identical to the reference build; float32
    differs by the reference's rcpps-based rcp() only (class C, a few ulp).  tests/test_special.py, tests/golden/ellint.npz.
*/
#pragma once

#include <enoki/array.h>

namespace enoki {

namespace detail {
    template <typename Scalar> constexpr Scalar ellint_tolerance(double scale) {
        // eps^(1/6): 0.0024608 (double), 0.070154 (float); scaled by 0.6 for R_D / R_J and 0.48 for R_C
        return Scalar((sizeof(Scalar) == 8 ? 0.0024608 : 0.070154) * scale);
    }
    template <typename Value, typename Mask> inline void ellint_step(Value &v, const Mask &active, const Value &lambda) {
        using Scalar = scalar_t<Value>;
        v = select(active, (v + lambda) * Value(Scalar(0.25)), v);
    }
    /// lambda = sqrt(x) sqrt(y) + sqrt(y) sqrt(z) + sqrt(z) sqrt(x), summed as the reference's dot(shuffle<1, 2, 0>(s), s)
    template <typename Value> inline Value ellint_lambda(const Value &sx, const Value &sy, const Value &sz) {
        return fmadd(sx, sz, fmadd(sz, sy, sy * sx));
    }
    template <typename Value> inline Value lit_v(double c) { return Value(scalar_t<Value>(c)); }
}

template <typename Value> Value carlson_rf(Value x, Value y, Value z) {
    using Scalar = scalar_t<Value>;
    using Mask = mask_t<Value>;
    const Value one = Value(Scalar(1)), tol = Value(detail::ellint_tolerance<Scalar>(1.0));
    Value X, Y, Z, mu_inv;
    Mask active(true);
    for (int round = 1;; ++round) {
        Value lambda = detail::ellint_lambda(sqrt(x), sqrt(y), sqrt(z));
        Value mu = ((x + y) + z) * detail::lit_v<Value>(1.0 / 3.0);
        mu_inv = rcp(mu);
        X = fnmadd(x, mu_inv, one); Y = fnmadd(y, mu_inv, one); Z = fnmadd(z, mu_inv, one);
        active = active & (max(max(abs(X), abs(Y)), abs(Z)) > tol);
        if (none(active) || round == 10) break;
        detail::ellint_step(x, active, lambda); detail::ellint_step(y, active, lambda); detail::ellint_step(z, active, lambda);
    }
    Value e2 = X * Y - Z * Z, e3 = (X * Y) * Z;
    Value er = ((detail::lit_v<Value>(1.0 / 24.0) * e2 - detail::lit_v<Value>(1.0 / 10.0)) - detail::lit_v<Value>(3.0 / 44.0) * e3) * e2 +
               detail::lit_v<Value>(1.0 / 14.0) * e3;
    return sqrt(mu_inv) * (one + er);
}

template <typename Value> Value carlson_rd(Value x, Value y, Value z) {
    using Scalar = scalar_t<Value>;
    using Mask = mask_t<Value>;
    const Value one = Value(Scalar(1)), tol = Value(detail::ellint_tolerance<Scalar>(0.6)), quarter = Value(Scalar(0.25));
    Value X, Y, Z, mu_inv, sum = Value(Scalar(0)), num = one;
    Mask active(true);
    for (int round = 1;; ++round) {
        Value lambda = detail::ellint_lambda(sqrt(x), sqrt(y), sqrt(z));
        Value mu = ((x * detail::lit_v<Value>(1.0 / 5.0) + y * detail::lit_v<Value>(1.0 / 5.0)) + z * detail::lit_v<Value>(3.0 / 5.0));
        mu_inv = rcp(mu);
        X = fnmadd(x, mu_inv, one); Y = fnmadd(y, mu_inv, one); Z = fnmadd(z, mu_inv, one);
        active = active & (max(max(abs(X), abs(Y)), abs(Z)) > tol);
        if (none(active) || round == 10) break;
        sum = select(active, sum + num / (sqrt(z) * (z + lambda)), sum);
        num = select(active, num * quarter, num);
        detail::ellint_step(x, active, lambda); detail::ellint_step(y, active, lambda); detail::ellint_step(z, active, lambda);
    }
    Value ea = X * Y, eb = Z * Z, ec = ea - eb, ed = fnmadd(detail::lit_v<Value>(6.0), eb, ea), ee = fmadd(ec, detail::lit_v<Value>(2.0), ed);
    Value p = ed * ((-detail::lit_v<Value>(3.0 / 14.0) + detail::lit_v<Value>(9.0 / 88.0) * ed) - (detail::lit_v<Value>(1.0 / 4.0) * Z) * ee) +
              Z * (detail::lit_v<Value>(1.0 / 6.0) * ee + Z * (-detail::lit_v<Value>(9.0 / 22.0) * ec + (Z * detail::lit_v<Value>(3.0 / 26.0)) * ea));
    return detail::lit_v<Value>(3.0) * sum + ((num * mu_inv) * sqrt(mu_inv)) * (one + p);
}

template <typename Value> Value carlson_rc(Value x, Value y) {
    using Scalar = scalar_t<Value>;
    using Mask = mask_t<Value>;
    const Value one = Value(Scalar(1)), tol = Value(detail::ellint_tolerance<Scalar>(0.48));
    Value mu_inv, s;
    Mask active(true);
    for (int round = 1;; ++round) {
        Value lambda = sqrt(x) * sqrt(y);
        lambda = lambda + (lambda + y);
        Value mu = fmadd(x, detail::lit_v<Value>(1.0 / 3.0), y * detail::lit_v<Value>(2.0 / 3.0));
        mu_inv = rcp(mu);
        s = (y - mu) * mu_inv;
        active = active & (abs(s) > tol);
        if (none(active) || round == 10) break;
        detail::ellint_step(x, active, lambda); detail::ellint_step(y, active, lambda);
    }
    return sqrt(mu_inv) * (one + (s * s) * (detail::lit_v<Value>(0.3) + s * (detail::lit_v<Value>(1.0 / 7.0) +
                                           s * (detail::lit_v<Value>(0.375) + s * detail::lit_v<Value>(9.0 / 22.0)))));
}

template <typename Value> Value carlson_rj(Value x, Value y, Value z, Value r) {
    using Scalar = scalar_t<Value>;
    using Mask = mask_t<Value>;
    const Value one = Value(Scalar(1)), tol = Value(detail::ellint_tolerance<Scalar>(0.6)), quarter = Value(Scalar(0.25));
    Value X, Y, Z, R, mu_inv, sum = Value(Scalar(0)), num = one;
    Mask active(true);
    for (int round = 1;; ++round) {
        Value sx = sqrt(x), sy = sqrt(y), sz = sqrt(z);
        Value lambda = detail::ellint_lambda(sx, sy, sz);
        Value mu = ((((x + y) + z) + r) + r) * detail::lit_v<Value>(1.0 / 5.0);
        mu_inv = rcp(mu);
        X = fnmadd(x, mu_inv, one); Y = fnmadd(y, mu_inv, one); Z = fnmadd(z, mu_inv, one); R = fnmadd(r, mu_inv, one);
        active = active & (max(max(max(abs(X), abs(Y)), abs(Z)), abs(R)) > tol);
        Value alpha = r * ((sx + sy) + sz) + sqrt((x * y) * z);
        alpha = alpha * alpha;
        Value beta = (r * (r + lambda)) * (r + lambda);
        if (none(active) || round == 10) break;
        sum = select(active, sum + num * carlson_rc(alpha, beta), sum);
        num = select(active, num * quarter, num);
        detail::ellint_step(x, active, lambda); detail::ellint_step(y, active, lambda);
        detail::ellint_step(z, active, lambda); detail::ellint_step(r, active, lambda);
    }
    Value ea = X * (Y + Z) + Y * Z, eb = (X * Y) * Z, ec = R * R, ed = ea - detail::lit_v<Value>(3.0) * ec,
          ee = eb + (detail::lit_v<Value>(2.0) * R) * (ea - ec);
    Value series = ((((one + ed * ((-detail::lit_v<Value>(3.0 / 14.0) + detail::lit_v<Value>(9.0 / 88.0) * ed) - detail::lit_v<Value>(9.0 / 52.0) * ee)) +
                      eb * (detail::lit_v<Value>(1.0 / 6.0) + R * (-detail::lit_v<Value>(3.0 / 11.0) + R * detail::lit_v<Value>(3.0 / 26.0)))) +
                     (R * ea) * (detail::lit_v<Value>(1.0 / 3.0) - R * detail::lit_v<Value>(3.0 / 22.0))) -
                    (detail::lit_v<Value>(1.0 / 3.0) * R) * ec);
    return detail::lit_v<Value>(3.0) * sum + ((num * mu_inv) * sqrt(mu_inv)) * series;
}

// ---------------------------------------------------------------------------------------------------------------------
//  Legendre forms.  Arguments outside [-pi/2, pi/2] are reduced by n = floor(phi / pi + 1/2) periods:
//  F(phi, k) = 2 n K(k) + F(phi - n pi, k), likewise for E and Pi.
// ---------------------------------------------------------------------------------------------------------------------
template <typename K, typename Value = expr_t<K>> Value comp_ellint_1(const K &k_) {
    using Scalar = scalar_t<Value>;
    Value k = Value(k_);
    return carlson_rf(Value(Scalar(0)), Value(Scalar(1)) - k * k, Value(Scalar(1)));
}

template <typename K, typename Value = expr_t<K>> Value comp_ellint_2(const K &k_) {
    using Scalar = scalar_t<Value>;
    Value k = Value(k_), k2 = k * k, zero = Value(Scalar(0)), one = Value(Scalar(1));
    return carlson_rf(zero, one - k2, one) - (detail::lit_v<Value>(1.0 / 3.0) * k2) * carlson_rd(zero, one - k2, one);
}

template <typename K, typename Nu, typename Value = expr_t<K, Nu>> Value comp_ellint_3(const K &k_, const Nu &nu_) {
    using Scalar = scalar_t<Value>;
    Value k = Value(k_), nu = Value(nu_), k2 = k * k, zero = Value(Scalar(0)), one = Value(Scalar(1));
    return carlson_rf(zero, one - k2, one) - (detail::lit_v<Value>(1.0 / 3.0) * nu) * carlson_rj(zero, one - k2, one, one + nu);
}

namespace detail {
    /// phi -> (n, phi - n pi); `periods` tells whether any lane left the principal interval
    template <typename Value> inline Value ellint_reduce(Value &phi, bool &periods) {
        using Scalar = scalar_t<Value>;
        Value n = floor(fmadd(phi, Value(Scalar(1.0 / 3.14159265358979323846)), Value(Scalar(0.5))));
        periods = any_nested(neq(n, Value(Scalar(0))));
        if (periods) phi = fnmadd(n, Value(Scalar(3.14159265358979323846)), phi);
        return n;
    }
}

template <typename Phi, typename K, typename Value = expr_t<Phi, K>> Value ellint_1(const Phi &phi_, const K &k_) {
    using Scalar = scalar_t<Value>;
    Value phi = Value(phi_), k = Value(k_), one = Value(Scalar(1)), result = Value(Scalar(0));
    bool periods;
    Value n = detail::ellint_reduce(phi, periods);
    if (periods) result = (comp_ellint_1(k) * n) * Value(Scalar(2));
    auto [s, c] = sincos(phi);
    return result + s * carlson_rf(c * c, one - ((k * k) * s) * s, one);
}

template <typename Phi, typename K, typename Value = expr_t<Phi, K>> Value ellint_2(const Phi &phi_, const K &k_) {
    using Scalar = scalar_t<Value>;
    Value phi = Value(phi_), k = Value(k_), k2 = k * k, one = Value(Scalar(1)), result = Value(Scalar(0));
    bool periods;
    Value n = detail::ellint_reduce(phi, periods);
    if (periods) result = (comp_ellint_2(k) * n) * Value(Scalar(2));
    auto [s, c] = sincos(phi);
    Value s2k2 = (s * s) * k2, x = c * c, y = one - s2k2;
    return result + s * (carlson_rf(x, y, one) - (detail::lit_v<Value>(1.0 / 3.0) * s2k2) * carlson_rd(x, y, one));
}

template <typename Phi, typename K, typename Nu, typename Value = expr_t<expr_t<Phi, K>, Nu>>
Value ellint_3(const Phi &phi_, const K &k_, const Nu &nu_) {
    using Scalar = scalar_t<Value>;
    Value phi = Value(phi_), k = Value(k_), nu = Value(nu_), k2 = k * k, one = Value(Scalar(1)), result = Value(Scalar(0));
    bool periods;
    Value n = detail::ellint_reduce(phi, periods);
    if (periods) result = (comp_ellint_3(k, nu) * n) * Value(Scalar(2));
    auto [s, c] = sincos(phi);
    Value s2 = s * s, x = c * c, y = one - k2 * s2;
    return result + s * (carlson_rf(x, y, one) - ((detail::lit_v<Value>(1.0 / 3.0) * nu) * s2) * carlson_rj(x, y, one, one + nu * s2));
}

} // namespace enoki
