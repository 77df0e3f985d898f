/*
    enoki/matrix.h -- small square matrices over array types

    Matrix<Value, N> is N columns of Array<Value, N> (column-major, like the reference's
    include/enoki/matrix.h:20-150), so Matrix<HIPArray<float>, 4> is a structure of 16 device arrays and
    `m * v` transforms as many vectors as the arrays have entries.  Products use the reference's
    operation order -- per result column `c0 * s0`, then `fmadd(c_i, s_i, sum)` (matrix.h:152-180) -- so the
    results are bit-identical to the CPU path for the same inputs.

    Provided: element access m(i, j), column access, zero / identity / diag, matrix * matrix, matrix * vector,
    matrix * scalar, transpose, trace, frob, det and inverse for N = 2, 3 (matrix.h:262-318, same operation order;
    they contain one rcp(), parity class C) and for N = 4.  The 4 x 4 case is a Laplace expansion over 2 x 2 minors
    -- NOT the reference's SSE shuffle formulation (matrix.h:320-415) -- so it agrees with the reference to rounding
    (a few ulp times the condition number), not bit for bit.  The polar decomposition is not provided.
*/
#pragma once

#include <enoki/array.h>

#include <utility>

namespace enoki {

template <typename Value_, size_t Size_> struct Matrix : Array<Array<Value_, Size_>, Size_> {
    using Entry = Value_;
    using Column = Array<Value_, Size_>;
    using Base = Array<Column, Size_>;
    static constexpr size_t Size = Size_;
    static constexpr bool IsMatrix = true;

    Matrix() = default;
    Matrix(const Base &b) : Base(b) { }

    /// Diagonal matrix with `v` on the diagonal (matrix.h:53-59)
    Matrix(const Entry &v) {
        for (size_t j = 0; j < Size; ++j)
            for (size_t i = 0; i < Size; ++i)
                this->coeff(j).coeff(i) = i == j ? v : Entry(scalar_t<Entry>(0));
    }

    /// From N columns
    template <typename... Cols, enable_if_t<sizeof...(Cols) == Size_ && (std::is_same_v<std::decay_t<Cols>, Column> && ...)> = 0>
    Matrix(const Cols &... cols) {
        size_t k = 0;
        ((this->coeff(k++) = cols), ...);
    }

    /// From N * N entries in ROW-major order, as one writes a matrix down (matrix.h:107-115)
    template <typename... Args, enable_if_t<sizeof...(Args) == Size_ * Size_ && (Size_ > 1)> = 0>
    Matrix(const Args &... args) {
        Entry entries[] = { Entry(args)... };
        for (size_t i = 0; i < Size; ++i)
            for (size_t j = 0; j < Size; ++j)
                this->coeff(j).coeff(i) = entries[i * Size + j];
    }

    Entry &operator()(size_t i, size_t j) { return this->coeff(j).coeff(i); }
    const Entry &operator()(size_t i, size_t j) const { return this->coeff(j).coeff(i); }
    Column &col(size_t j) { return this->coeff(j); }
    const Column &col(size_t j) const { return this->coeff(j); }
    Column row(size_t i) const {
        Column r;
        for (size_t j = 0; j < Size; ++j) r.coeff(j) = (*this)(i, j);
        return r;
    }
};

template <typename T> constexpr bool is_matrix_v = false;
template <typename V, size_t N> constexpr bool is_matrix_v<Matrix<V, N>> = true;

template <typename M, enable_if_t<is_matrix_v<M>> = 0> inline M identity(size_t size = 1) {
    using E = typename M::Entry;
    M r;
    for (size_t j = 0; j < M::Size; ++j)
        for (size_t i = 0; i < M::Size; ++i)
            r(i, j) = i == j ? full<E>(scalar_t<E>(1), size) : zero<E>(size);
    return r;
}

template <typename M, enable_if_t<is_matrix_v<M>> = 0> inline M diag(const typename M::Column &v) {
    M r;
    for (size_t j = 0; j < M::Size; ++j)
        for (size_t i = 0; i < M::Size; ++i)
            r(i, j) = i == j ? v.coeff(i) : typename M::Entry(scalar_t<typename M::Entry>(0));
    return r;
}

template <typename V, size_t N> inline Array<V, N> diag(const Matrix<V, N> &m) {
    Array<V, N> r;
    for (size_t i = 0; i < N; ++i) r.coeff(i) = m(i, i);
    return r;
}

/// matrix * matrix (matrix.h:152-167)
template <typename V, size_t N> inline Matrix<V, N> operator*(const Matrix<V, N> &a, const Matrix<V, N> &b) {
    using Column = typename Matrix<V, N>::Column;
    Matrix<V, N> r;
    for (size_t j = 0; j < N; ++j) {
        Column sum = a.col(0) * Column(b(0, j));
        for (size_t i = 1; i < N; ++i) sum = fmadd(a.col(i), Column(b(i, j)), sum);
        r.col(j) = sum;
    }
    return r;
}

/// matrix * vector (matrix.h:169-178)
template <typename V, size_t N> inline Array<V, N> operator*(const Matrix<V, N> &m, const Array<V, N> &v) {
    using Column = Array<V, N>;
    Column sum = m.col(0) * Column(v.coeff(0));
    for (size_t i = 1; i < N; ++i) sum = fmadd(m.col(i), Column(v.coeff(i)), sum);
    return sum;
}

/// matrix * scalar entry, scalar * matrix (matrix.h:179-194)
template <typename V, size_t N> inline Matrix<V, N> operator*(const Matrix<V, N> &m, const V &s) {
    Matrix<V, N> r;
    for (size_t j = 0; j < N; ++j) r.col(j) = m.col(j) * Array<V, N>(s);
    return r;
}
template <typename V, size_t N> inline Matrix<V, N> operator*(const V &s, const Matrix<V, N> &m) {
    Matrix<V, N> r;
    for (size_t j = 0; j < N; ++j) r.col(j) = Array<V, N>(s) * m.col(j);
    return r;
}
template <typename V, size_t N> inline Matrix<V, N> operator+(const Matrix<V, N> &a, const Matrix<V, N> &b) {
    Matrix<V, N> r;
    for (size_t j = 0; j < N; ++j) r.col(j) = a.col(j) + b.col(j);
    return r;
}
template <typename V, size_t N> inline Matrix<V, N> operator-(const Matrix<V, N> &a, const Matrix<V, N> &b) {
    Matrix<V, N> r;
    for (size_t j = 0; j < N; ++j) r.col(j) = a.col(j) - b.col(j);
    return r;
}

template <typename V, size_t N> inline Matrix<V, N> transpose(const Matrix<V, N> &m) {
    Matrix<V, N> r;
    for (size_t j = 0; j < N; ++j)
        for (size_t i = 0; i < N; ++i)
            r(i, j) = m(j, i);
    return r;
}

/// Sum of the diagonal, in index order (matrix.h:205-211)
template <typename V, size_t N> inline V trace(const Matrix<V, N> &m) {
    V r = m(0, 0);
    for (size_t i = 1; i < N; ++i) r = r + m(i, i);
    return r;
}

/// Squared Frobenius norm (matrix.h:213-219)
template <typename V, size_t N> inline V frob(const Matrix<V, N> &m) {
    Array<V, N> r = m.col(0) * m.col(0);
    for (size_t i = 1; i < N; ++i) r = fmadd(m.col(i), m.col(i), r);
    return hsum(r);
}

template <typename V> inline V det(const Matrix<V, 2> &m) { return fmsub(m(0, 0), m(1, 1), m(0, 1) * m(1, 0)); }

template <typename V> inline Matrix<V, 2> inverse(const Matrix<V, 2> &m) {
    V inv_det = rcp(fmsub(m(0, 0), m(1, 1), m(0, 1) * m(1, 0)));
    return Matrix<V, 2>(m(1, 1) * inv_det, -m(0, 1) * inv_det,
                        -m(1, 0) * inv_det, m(0, 0) * inv_det);
}

template <typename V> inline V det(const Matrix<V, 3> &m) { return dot(m.col(0), cross(m.col(1), m.col(2))); }

/// Rows of the inverse are cross products of the columns (matrix.h:286-311)
template <typename V> inline Matrix<V, 3> inverse_transpose(const Matrix<V, 3> &m) {
    using Vector = Array<V, 3>;
    Vector row0 = cross(m.col(1), m.col(2)), row1 = cross(m.col(2), m.col(0)), row2 = cross(m.col(0), m.col(1));
    Vector inv_det = Vector(rcp(dot(m.col(0), row0)));
    return Matrix<V, 3>(Vector(row0 * inv_det), Vector(row1 * inv_det), Vector(row2 * inv_det));
}
template <typename V> inline Matrix<V, 3> inverse(const Matrix<V, 3> &m) { return transpose(inverse_transpose(m)); }

namespace detail {
    /// The twelve 2 x 2 minors of the upper (s) and lower (c) row pairs of a 4 x 4 matrix
    template <typename V> struct Minors4 {
        V s[6], c[6];
        explicit Minors4(const Matrix<V, 4> &m) {
            static constexpr int pairs[6][2] = { { 0, 1 }, { 0, 2 }, { 0, 3 }, { 1, 2 }, { 1, 3 }, { 2, 3 } };
            for (int k = 0; k < 6; ++k) {
                const int a = pairs[k][0], b = pairs[k][1];
                s[k] = fmsub(m(0, a), m(1, b), m(1, a) * m(0, b));
                c[k] = fmsub(m(2, a), m(3, b), m(3, a) * m(2, b));
            }
        }
        V det() const {
            V d = s[0] * c[5];
            d = fnmadd(s[1], c[4], d);
            d = fmadd(s[2], c[3], d);
            d = fmadd(s[3], c[2], d);
            d = fnmadd(s[4], c[1], d);
            return fmadd(s[5], c[0], d);
        }
    };
    /// a * x - b * y + c * z
    template <typename V> inline V expand3(const V &a, const V &x, const V &b, const V &y, const V &c, const V &z) {
        return fmadd(c, z, fmsub(a, x, b * y));
    }
}

template <typename V> inline V det(const Matrix<V, 4> &m) { return detail::Minors4<V>(m).det(); }

template <typename V> inline Matrix<V, 4> inverse(const Matrix<V, 4> &m) {
    detail::Minors4<V> k(m);
    const V *s = k.s, *c = k.c;
    V inv_det = rcp(k.det());
    using detail::expand3;
    Matrix<V, 4> r;
    r(0, 0) =  expand3(m(1, 1), c[5], m(1, 2), c[4], m(1, 3), c[3]) * inv_det;
    r(0, 1) = -expand3(m(0, 1), c[5], m(0, 2), c[4], m(0, 3), c[3]) * inv_det;
    r(0, 2) =  expand3(m(3, 1), s[5], m(3, 2), s[4], m(3, 3), s[3]) * inv_det;
    r(0, 3) = -expand3(m(2, 1), s[5], m(2, 2), s[4], m(2, 3), s[3]) * inv_det;
    r(1, 0) = -expand3(m(1, 0), c[5], m(1, 2), c[2], m(1, 3), c[1]) * inv_det;
    r(1, 1) =  expand3(m(0, 0), c[5], m(0, 2), c[2], m(0, 3), c[1]) * inv_det;
    r(1, 2) = -expand3(m(3, 0), s[5], m(3, 2), s[2], m(3, 3), s[1]) * inv_det;
    r(1, 3) =  expand3(m(2, 0), s[5], m(2, 2), s[2], m(2, 3), s[1]) * inv_det;
    r(2, 0) =  expand3(m(1, 0), c[4], m(1, 1), c[2], m(1, 3), c[0]) * inv_det;
    r(2, 1) = -expand3(m(0, 0), c[4], m(0, 1), c[2], m(0, 3), c[0]) * inv_det;
    r(2, 2) =  expand3(m(3, 0), s[4], m(3, 1), s[2], m(3, 3), s[0]) * inv_det;
    r(2, 3) = -expand3(m(2, 0), s[4], m(2, 1), s[2], m(2, 3), s[0]) * inv_det;
    r(3, 0) = -expand3(m(1, 0), c[3], m(1, 1), c[1], m(1, 2), c[0]) * inv_det;
    r(3, 1) =  expand3(m(0, 0), c[3], m(0, 1), c[1], m(0, 2), c[0]) * inv_det;
    r(3, 2) = -expand3(m(3, 0), s[3], m(3, 1), s[1], m(3, 2), s[0]) * inv_det;
    r(3, 3) =  expand3(m(2, 0), s[3], m(2, 1), s[1], m(2, 2), s[0]) * inv_det;
    return r;
}

template <typename V> inline Matrix<V, 4> inverse_transpose(const Matrix<V, 4> &m) { return transpose(inverse(m)); }

/// Polar decomposition A = Q P (Q orthogonal, P symmetric positive semi-definite) by the scaled Newton iteration
/// Q <- (g Q + Q^-T / g) / 2, g = sqrt(|Q^-T|_F / |Q|_F) (N. Higham, "Computing the polar decomposition -- with
/// applications", SIAM J. Sci. Stat. Comput. 7 (1986); reference matrix.h:526-539: the same update, `it` rounds).
template <typename V, size_t N> inline std::pair<Matrix<V, N>, Matrix<V, N>> polar_decomp(const Matrix<V, N> &A, size_t it = 10) {
    using S = scalar_t<V>;
    Matrix<V, N> Q = A;
    for (size_t round = 0; round < it; ++round) {
        Matrix<V, N> Qi = inverse_transpose(Q);
        V gamma = sqrt(frob(Qi) / frob(Q));
        V a = gamma * V(S(0.5)), b = rcp(gamma) * V(S(0.5));
        for (size_t j = 0; j < N; ++j)
            for (size_t i = 0; i < N; ++i) Q(i, j) = fmadd(Q(i, j), a, Qi(i, j) * b);
    }
    return { Q, transpose(Q) * A };
}

} // namespace enoki
