/*
    enoki/transform.h -- homogeneous transformation matrices over any entry type (float, HIPArray<float>, DiffArray, ...)
                         (reference: include/enoki/transform.h:20-150)

        translate<Matrix4>(v)   scale<Matrix4>(v)   rotate<Matrix3>(angle)   rotate<Matrix4>(axis, angle)
        perspective<Matrix4>(fov, near, far, aspect)   frustum / ortho<Matrix4>(left, right, bottom, top, near, far)
        look_at<Matrix4>(origin, target, up)

    Conventions are the reference's (OpenGL-style clip space, column-major storage, `fov` in RADIANS, the rotation of
    Rodrigues' formula in the sign convention of a right-handed system).  Every entry is evaluated in the reference's
    operation order, so results on device arrays agree with its CPU arrays bit for bit wherever no rcp() / normalize() is
    involved and to their class-C bound otherwise (tests/test_matrix.py, tests/golden/transform.npz).
    transform_decompose / transform_compose / transform_compose_inverse split an affine 4 x 4 matrix into scale / shear
    (symmetric 3 x 3), rotation (quaternion) and translation through the polar decomposition of matrix.h, and back.
*/
#pragma once

#include <enoki/matrix.h>
#include <enoki/quaternion.h>

#include <tuple>

namespace enoki {

namespace detail {
    template <typename E> inline E tr_lit(double c) { return E(scalar_t<E>(c)); }
}

/// Last column = (v, 1)
template <typename M, typename Vector, enable_if_t<is_matrix_v<M>> = 0> inline M translate(const Vector &v) {
    M t = identity<M>();
    for (size_t i = 0; i + 1 < M::Size; ++i) t(i, M::Size - 1) = typename M::Entry(v.coeff(i));
    return t;
}

/// diag(v, 1)
template <typename M, typename Vector, enable_if_t<is_matrix_v<M>> = 0> inline M scale(const Vector &v) {
    using E = typename M::Entry;
    M t = identity<M>();
    for (size_t i = 0; i + 1 < M::Size; ++i) t(i, i) = E(v.coeff(i));
    return t;
}

/// 2-D rotation in homogeneous coordinates (3 x 3)
template <typename M, enable_if_t<is_matrix_v<M> && M::Size == 3> = 0> inline M rotate(const typename M::Entry &angle) {
    using E = typename M::Entry;
    const E z = detail::tr_lit<E>(0), o = detail::tr_lit<E>(1);
    auto sc = sincos(angle);
    return M(sc.second, -sc.first, z, sc.first, sc.second, z, z, z, o);
}

/// Rotation by `angle` about the UNIT vector `axis` (4 x 4)
template <typename M, typename Vector3, enable_if_t<is_matrix_v<M> && M::Size == 4> = 0>
inline M rotate(const Vector3 &axis, const typename M::Entry &angle) {
    using E = typename M::Entry;
    auto sc = sincos(angle);
    const E s = sc.first, c = sc.second, cm = detail::tr_lit<E>(1) - c, z = detail::tr_lit<E>(0), o = detail::tr_lit<E>(1);
    const E a[3] = { E(axis.coeff(0)), E(axis.coeff(1)), E(axis.coeff(2)) };
    E d[3], p[3], q[3];          // d: a_i a_i cm + c;  p: a_i a_(i+1) cm + a_(i+2) s;  q: a_i a_(i+2) cm - a_(i+1) s
    for (int i = 0; i < 3; ++i) {
        const E &n1 = a[(i + 1) % 3], &n2 = a[(i + 2) % 3];
        d[i] = fmadd(a[i] * a[i], cm, c);
        p[i] = fmadd(a[i] * n1, cm, n2 * s);
        q[i] = fmsub(a[i] * n2, cm, n1 * s);
    }
    using Col = typename M::Column;
    return M(Col(d[0], p[0], q[0], z), Col(q[1], d[1], p[1], z), Col(p[2], q[2], d[2], z), Col(z, z, z, o));
}

template <typename M, enable_if_t<is_matrix_v<M> && M::Size == 4> = 0>
inline M perspective(const typename M::Entry &fov, const typename M::Entry &near_, const typename M::Entry &far_,
                     const typename M::Entry &aspect = typename M::Entry(scalar_t<typename M::Entry>(1))) {
    using E = typename M::Entry;
    const E recip = rcp(near_ - far_), c = cot(detail::tr_lit<E>(0.5) * fov), z = detail::tr_lit<E>(0);
    M t = diag<M>(typename M::Column(c / aspect, c, (near_ + far_) * recip, z));
    t(2, 3) = ((detail::tr_lit<E>(2) * near_) * far_) * recip;
    t(3, 2) = detail::tr_lit<E>(-1);
    return t;
}

template <typename M, enable_if_t<is_matrix_v<M> && M::Size == 4> = 0>
inline M frustum(const typename M::Entry &left, const typename M::Entry &right, const typename M::Entry &bottom,
                 const typename M::Entry &top, const typename M::Entry &near_, const typename M::Entry &far_) {
    using E = typename M::Entry;
    const E rl = rcp(right - left), tb = rcp(top - bottom), fn = rcp(far_ - near_), two = detail::tr_lit<E>(2);
    M t = M(detail::tr_lit<E>(0));
    t(0, 0) = (two * near_) * rl;
    t(1, 1) = (two * near_) * tb;
    t(0, 2) = (right + left) * rl;
    t(1, 2) = (top + bottom) * tb;
    t(2, 2) = -(far_ + near_) * fn;
    t(3, 2) = detail::tr_lit<E>(-1);
    t(2, 3) = ((detail::tr_lit<E>(-2) * far_) * near_) * fn;
    return t;
}

template <typename M, enable_if_t<is_matrix_v<M> && M::Size == 4> = 0>
inline M ortho(const typename M::Entry &left, const typename M::Entry &right, const typename M::Entry &bottom,
               const typename M::Entry &top, const typename M::Entry &near_, const typename M::Entry &far_) {
    using E = typename M::Entry;
    const E rl = rcp(right - left), tb = rcp(top - bottom), fn = rcp(far_ - near_), two = detail::tr_lit<E>(2);
    M t = M(detail::tr_lit<E>(0));
    t(0, 0) = two * rl;
    t(1, 1) = two * tb;
    t(2, 2) = detail::tr_lit<E>(-2) * fn;
    t(3, 3) = detail::tr_lit<E>(1);
    t(0, 3) = -(right + left) * rl;
    t(1, 3) = -(top + bottom) * tb;
    t(2, 3) = -(far_ + near_) * fn;
    return t;
}

/// Columns: (left, 0), (up', 0), (-dir, 0), (-left.o, -up'.o, dir.o, 1)  -- the reference's layout
template <typename M, typename Point, typename Vector, enable_if_t<is_matrix_v<M> && M::Size == 4> = 0>
inline M look_at(const Point &origin, const Point &target, const Vector &up) {
    using E = typename M::Entry;
    using Col = typename M::Column;
    const E z = detail::tr_lit<E>(0), o = detail::tr_lit<E>(1);
    auto dir = normalize(target - origin);
    auto left = normalize(cross(dir, up));
    auto new_up = cross(left, dir);
    return M(Col(left.coeff(0), left.coeff(1), left.coeff(2), z), Col(new_up.coeff(0), new_up.coeff(1), new_up.coeff(2), z),
             Col(-dir.coeff(0), -dir.coeff(1), -dir.coeff(2), z), Col(-dot(left, origin), -dot(new_up, origin), dot(dir, origin), o));
}

/// A = T(t) R(q) S: returns (S, q, t).  A reflection is moved from the rotation into S (det R = +1), like the reference
/// (transform.h:152-173).
template <typename V> inline std::tuple<Matrix<V, 3>, Quaternion<V>, Array<V, 3>> transform_decompose(const Matrix<V, 4> &A, size_t it = 10) {
    using M3 = Matrix<V, 3>;
    M3 sub;
    for (size_t j = 0; j < 3; ++j)
        for (size_t i = 0; i < 3; ++i) sub(i, j) = A(i, j);
    auto qp = polar_decomp(sub, it);
    M3 Q = qp.first, P = qp.second;
    if (any_nested(isnan(Q(0, 0)))) Q = identity<M3>();          // singular input
    V sign = det(Q);
    for (size_t j = 0; j < 3; ++j)
        for (size_t i = 0; i < 3; ++i) { Q(i, j) = mulsign(Q(i, j), sign); P(i, j) = mulsign(P(i, j), sign); }
    return { P, matrix_to_quat(Q), Array<V, 3>(A(0, 3), A(1, 3), A(2, 3)) };
}

template <typename V, typename Vector3> inline Matrix<V, 4> transform_compose(const Matrix<V, 3> &S, const Quaternion<V> &q, const Vector3 &t) {
    using E = scalar_t<V>;
    Matrix<V, 3> RS = quat_to_matrix<Matrix<V, 3>>(q) * S;
    Matrix<V, 4> r = identity<Matrix<V, 4>>();
    for (size_t j = 0; j < 3; ++j)
        for (size_t i = 0; i < 3; ++i) r(i, j) = RS(i, j);
    for (size_t i = 0; i < 3; ++i) r(i, 3) = V(t.coeff(i));
    (void) sizeof(E);
    return r;
}

template <typename V, typename Vector3> inline Matrix<V, 4> transform_compose_inverse(const Matrix<V, 3> &S, const Quaternion<V> &q, const Vector3 &t) {
    Matrix<V, 3> inv = inverse(quat_to_matrix<Matrix<V, 3>>(q) * S);
    Array<V, 3> back = inv * Array<V, 3>(-V(t.coeff(0)), -V(t.coeff(1)), -V(t.coeff(2)));
    Matrix<V, 4> r = identity<Matrix<V, 4>>();
    for (size_t j = 0; j < 3; ++j)
        for (size_t i = 0; i < 3; ++i) r(i, j) = inv(i, j);
    for (size_t i = 0; i < 3; ++i) r(i, 3) = back.coeff(i);
    return r;
}

} // namespace enoki
