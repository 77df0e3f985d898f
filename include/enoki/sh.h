/*
    enoki/sh.h -- real spherical harmonics of a unit direction (reference: include/enoki/sh.h)

        sh_eval(d, order, out)        out[l * (l + 1) + m] = Y_l^m(d),  0 <= l <= order,  -l <= m <= l

    Same basis, ordering and signs as the reference (orthonormal real harmonics with the Condon-Shortley phase:
    Y_1^{+1} = -0.4886 x, Y_1^{-1} = -0.4886 y, Y_1^0 = 0.4886 z).  The reference ships generated code for orders 0..9
    (P.-P. Sloan, "Efficient Spherical Harmonic Evaluation", JCGT 2(2), 2013); this header evaluates the same factorisation
    for ANY order with the recurrences the generator unrolls:

        Y_l^{+m} = N_l^m Q_l^m(z) c_m,   Y_l^{-m} = N_l^m Q_l^m(z) s_m,   Y_l^0 = N_l^0 Q_l^0(z)
        c_m + i s_m = (x + i y)^m                                  (two fused multiply-adds per m)
        Q_m^m = (-1)^m (2m - 1)!!,  Q_{m+1}^m = (2m + 1) z Q_m^m,  (l - m) Q_l^m = (2l - 1) z Q_{l-1}^m - (l + m - 1) Q_{l-2}^m
        N_l^m = sqrt((2 - [m = 0]) (2l + 1) / (4 pi) (l - m)! / (l + m)!)

    with the constants folded in double precision at compile time.  Works on every array flavour (scalars, HIPArray,
    DiffArray, vectorize() packets).  Agreement with the reference's generated code: a few ulp (different association of
    the constants), tests/test_sh.py.
*/
#pragma once

#include <enoki/array.h>

#include <cmath>
#include <stdexcept>

namespace enoki {

namespace detail {
    inline double sh_norm(int l, int m) {              // N_l^m above
        double ratio = 1.0;                            // (l - m)! / (l + m)!
        for (int k = l - m + 1; k <= l + m; ++k) ratio /= (double) k;
        return std::sqrt((m == 0 ? 1.0 : 2.0) * (2.0 * l + 1.0) / (4.0 * 3.14159265358979323846) * ratio);
    }
}

template <typename Vector3, typename Value = value_t<Vector3>> void sh_eval(const Vector3 &d, size_t order, Value *out) {
    static_assert(Vector3::Size == 3, "sh_eval(): the direction must be a 3D vector");
    using Scalar = scalar_t<Value>;
    if (order > 64) throw std::runtime_error("sh_eval(): order too high!");
    const int L = (int) order;
    const Value x = d.coeff(0), y = d.coeff(1), z = d.coeff(2);
    Value c = Value(Scalar(1)), s = Value(Scalar(0));          // (x + i y)^m
    double qmm = 1.0;                                           // (-1)^m (2m - 1)!!
    for (int m = 0; m <= L; ++m) {
        if (m > 0) {
            Value c_next = fmsub(x, c, y * s), s_next = fmadd(x, s, y * c);
            c = c_next; s = s_next;
            qmm *= -(2.0 * m - 1.0);
        }
        // Q_l^m for l = m, m + 1, ...: two-term recurrence in z, started from a constant
        Value q_prev2, q_prev = Value(Scalar(qmm));
        for (int l = m; l <= L; ++l) {
            Value q;
            if (l == m) q = q_prev;
            else if (l == m + 1) q = (z * Value(Scalar(2.0 * m + 1.0))) * q_prev;
            else q = fmsub(z * Value(Scalar((2.0 * l - 1.0) / (l - m))), q_prev, Value(Scalar((l + m - 1.0) / (l - m))) * q_prev2);
            if (l > m) { q_prev2 = q_prev; q_prev = q; }
            const Value nq = Value(Scalar(detail::sh_norm(l, m))) * q;
            if (m == 0) {
                out[l * (l + 1)] = nq;
            } else {
                out[l * (l + 1) + m] = nq * c;
                out[l * (l + 1) - m] = nq * s;
            }
        }
    }
}

} // namespace enoki
