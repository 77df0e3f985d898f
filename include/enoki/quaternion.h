/*
    enoki/quaternion.h -- quaternions over array types

    Quaternion<Value> is {x, y, z, w} = Array<Value, 4> with the real part LAST (reference include/enoki/quaternion.h:
    the constructor Quaternion(i, j, k, r)), so Quaternion<HIPArray<float>> is four device arrays and every operation
    processes as many quaternions as the arrays have entries.  Operation order follows the reference so that results
    agree with its CPU path wherever the real functions do:

      product   (quaternion.h:142-160) written out per component from the reference's shuffle formulation:
                  x = (q0x q1w + q0y q1z) + (q0w q1x - q0z q1y)        [fmadd / fmsub pairs, then one add]
                  y = (q0y q1w + q0z q1x) + (q0w q1y - q0x q1z)
                  z = (q0z q1w + q0x q1y) + (q0w q1z - q0y q1x)
                  w = -(q0x q1x + q0y q1y) + (q0w q1w - q0z q1z)
      rcp       conj(q) * (1 / squared_norm(q));  q0 / q1 = q0 * rcp(q1)
      exp, log, pow, sqrt (through Complex), slerp, rotate, quat_to_matrix (3x3 / 4x4), matrix_to_quat (Mike Day's
      branch-free selection), quat_to_euler -- quaternion.h:191-382.
*/
#pragma once

#include <enoki/complex.h>
#include <enoki/matrix.h>

namespace enoki {

template <typename Value_> struct Quaternion : Array<Value_, 4> {
    using Value = Value_;
    using Base = Array<Value_, 4>;
    static constexpr bool IsQuaternion = true;

    Quaternion() = default;
    Quaternion(const Base &b) : Base(b) { }
    /// A real number: (0, 0, 0, w)
    Quaternion(const Value &w) : Base(Value(scalar_t<Value>(0)), Value(scalar_t<Value>(0)), Value(scalar_t<Value>(0)), w) { }
    Quaternion(const Value &x, const Value &y, const Value &z, const Value &w) : Base(x, y, z, w) { }
    /// Imaginary part + real part
    Quaternion(const Array<Value, 3> &im, const Value &re) : Base(im.x(), im.y(), im.z(), re) { }
};

template <typename T> constexpr bool is_quaternion_v = false;
template <typename V> constexpr bool is_quaternion_v<Quaternion<V>> = true;

template <typename Q, enable_if_t<is_quaternion_v<Q>> = 0> inline Q identity(size_t size = 1) {
    using V = typename Q::Value;
    V z = zero<V>(size), o = full<V>(scalar_t<V>(1), size);
    return Q(z, z, z, o);
}

template <typename V> inline V real(const Quaternion<V> &q) { return q.w(); }
template <typename V> inline Array<V, 3> imag(const Quaternion<V> &q) { return Array<V, 3>(q.x(), q.y(), q.z()); }
template <typename V> inline V dot(const Quaternion<V> &a, const Quaternion<V> &b) {
    return dot((const Array<V, 4> &) a, (const Array<V, 4> &) b);
}
template <typename V> inline V squared_norm(const Quaternion<V> &q) { return squared_norm((const Array<V, 4> &) q); }
template <typename V> inline V norm(const Quaternion<V> &q) { return norm((const Array<V, 4> &) q); }
template <typename V> inline V abs(const Quaternion<V> &q) { return norm(q); }
template <typename V> inline Quaternion<V> normalize(const Quaternion<V> &q) { return Quaternion<V>(normalize((const Array<V, 4> &) q)); }
template <typename V> inline Quaternion<V> conj(const Quaternion<V> &q) { return Quaternion<V>(-q.x(), -q.y(), -q.z(), q.w()); }

template <typename V> inline Quaternion<V> operator+(const Quaternion<V> &a, const Quaternion<V> &b) {
    return Quaternion<V>((const Array<V, 4> &) a + (const Array<V, 4> &) b);
}
template <typename V> inline Quaternion<V> operator-(const Quaternion<V> &a, const Quaternion<V> &b) {
    return Quaternion<V>((const Array<V, 4> &) a - (const Array<V, 4> &) b);
}
template <typename V> inline Quaternion<V> operator-(const Quaternion<V> &a) { return Quaternion<V>(-(const Array<V, 4> &) a); }
template <typename V> inline Quaternion<V> operator*(const Quaternion<V> &a, const V &s) { return Quaternion<V>((const Array<V, 4> &) a * s); }
template <typename V> inline Quaternion<V> operator*(const V &s, const Quaternion<V> &a) { return Quaternion<V>((const Array<V, 4> &) a * s); }
template <typename V> inline Quaternion<V> operator/(const Quaternion<V> &a, const V &s) { return Quaternion<V>((const Array<V, 4> &) a / s); }

/// Hamilton product (see the header comment for the association)
template <typename V> inline Quaternion<V> operator*(const Quaternion<V> &a, const Quaternion<V> &b) {
    V t1x = fmadd(a.x(), b.w(), a.y() * b.z()), t1y = fmadd(a.y(), b.w(), a.z() * b.x()),
      t1z = fmadd(a.z(), b.w(), a.x() * b.y()), t1w = -fmadd(a.x(), b.x(), a.y() * b.y());
    V t2x = fmsub(a.w(), b.x(), a.z() * b.y()), t2y = fmsub(a.w(), b.y(), a.x() * b.z()),
      t2z = fmsub(a.w(), b.z(), a.y() * b.x()), t2w = fmsub(a.w(), b.w(), a.z() * b.z());
    return Quaternion<V>(t1x + t2x, t1y + t2y, t1z + t2z, t1w + t2w);
}

template <typename V> inline Quaternion<V> rcp(const Quaternion<V> &q) {
    return conj(q) * (V(scalar_t<V>(1)) / squared_norm(q));
}
template <typename V> inline Quaternion<V> operator/(const Quaternion<V> &a, const Quaternion<V> &b) { return a * rcp(b); }

template <typename V> inline Quaternion<V> exp(const Quaternion<V> &q) {
    Array<V, 3> qi = imag(q);
    V ri = norm(qi), exp_w = exp(real(q));
    auto sc = sincos(ri);
    return Quaternion<V>(qi * (sc.first * exp_w / ri), sc.second * exp_w);
}
template <typename V> inline Quaternion<V> log(const Quaternion<V> &q) {
    Array<V, 3> qi_n = normalize(imag(q));
    V rq = norm(q), acos_rq = acos(real(q) / rq), log_rq = log(rq);
    return Quaternion<V>(qi_n * acos_rq, log_rq);
}
template <typename V> inline Quaternion<V> pow(const Quaternion<V> &a, const Quaternion<V> &b) { return exp(log(a) * b); }
template <typename V> inline Quaternion<V> sqrt(const Quaternion<V> &q) {
    V ri = norm(imag(q));
    Complex<V> cs = sqrt(Complex<V>(real(q), ri));
    return Quaternion<V>(imag(q) * (rcp(ri) * imag(cs)), real(cs));
}

/// Rotation about `axis` (unit length) by `angle`
template <typename Q, typename V = typename Q::Value, enable_if_t<is_quaternion_v<Q>> = 0>
inline Q rotate(const Array<V, 3> &axis, const V &angle) {
    auto sc = sincos(angle * V(scalar_t<V>(0.5)));
    return Q(axis * sc.first, sc.second);
}

/// Spherical linear interpolation; nearly parallel inputs (cos > 0.9995) fall back to a normalised lerp
template <typename V> inline Quaternion<V> slerp(const Quaternion<V> &q0, const Quaternion<V> &q1_, const V &t) {
    using S = scalar_t<V>;
    V cos_theta = dot(q0, q1_);
    Quaternion<V> q1((const Array<V, 4> &) q1_ * Array<V, 4>(sign(cos_theta)));      // mulsign: take the short way round
    cos_theta = abs(cos_theta);
    V theta = acos(cos_theta);
    auto sc = sincos(theta * t);
    Quaternion<V> qperp = normalize(q1 - q0 * cos_theta), result = q0 * sc.second + qperp * sc.first;
    Quaternion<V> close = normalize(q0 * (V(S(1)) - t) + q1 * t);
    return Quaternion<V>(select(cos_theta > V(S(0.9995)), (const Array<V, 4> &) close, (const Array<V, 4> &) result));
}

/// Rotation matrix of a unit quaternion (N = 3 or 4)
template <typename Mat, typename V, enable_if_t<Mat::Size == 3 || Mat::Size == 4> = 0>
inline Mat quat_to_matrix(const Quaternion<V> &q_) {
    using S = scalar_t<V>;
    Quaternion<V> q = q_ * V(S(1.41421356237309504880));
    V xx = q.x() * q.x(), yy = q.y() * q.y(), zz = q.z() * q.z(), xy = q.x() * q.y(), xz = q.x() * q.z(), yz = q.y() * q.z(),
      xw = q.x() * q.w(), yw = q.y() * q.w(), zw = q.z() * q.w();
    const V one = V(S(1)), zero_v = V(S(0));
    if constexpr (Mat::Size == 4)
        return Mat(one - (yy + zz), xy - zw, xz + yw, zero_v,
                   xy + zw, one - (xx + zz), yz - xw, zero_v,
                   xz - yw, yz + xw, one - (xx + yy), zero_v,
                   zero_v, zero_v, zero_v, one);
    else
        return Mat(one - (yy + zz), xy - zw, xz + yw,
                   xy + zw, one - (xx + zz), yz - xw,
                   xz - yw, yz + xw, one - (xx + yy));
}

/// "Converting a Rotation Matrix to a Quaternion" (Mike Day): four candidates, the best conditioned one selected per entry
template <typename V, size_t N, enable_if_t<N == 3 || N == 4> = 0>
inline Quaternion<V> matrix_to_quat(const Matrix<V, N> &mat) {
    using S = scalar_t<V>;
    using Q = Quaternion<V>;
    const V c1 = V(S(1));
    V t0 = c1 + mat(0, 0) - mat(1, 1) - mat(2, 2);
    Q q0(t0, mat(1, 0) + mat(0, 1), mat(0, 2) + mat(2, 0), mat(2, 1) - mat(1, 2));
    V t1 = c1 - mat(0, 0) + mat(1, 1) - mat(2, 2);
    Q q1(mat(1, 0) + mat(0, 1), t1, mat(2, 1) + mat(1, 2), mat(0, 2) - mat(2, 0));
    V t2 = c1 - mat(0, 0) - mat(1, 1) + mat(2, 2);
    Q q2(mat(0, 2) + mat(2, 0), mat(2, 1) + mat(1, 2), t2, mat(1, 0) - mat(0, 1));
    V t3 = c1 + mat(0, 0) + mat(1, 1) + mat(2, 2);
    Q q3(mat(2, 1) - mat(1, 2), mat(0, 2) - mat(2, 0), mat(1, 0) - mat(0, 1), t3);
    auto mask0 = mat(0, 0) > mat(1, 1);
    V t01 = select(mask0, t0, t1);
    Array<V, 4> q01 = select(mask0, (const Array<V, 4> &) q0, (const Array<V, 4> &) q1);
    auto mask1 = mat(0, 0) < -mat(1, 1);
    V t23 = select(mask1, t2, t3);
    Array<V, 4> q23 = select(mask1, (const Array<V, 4> &) q2, (const Array<V, 4> &) q3);
    auto mask2 = mat(2, 2) < V(S(0));
    V t0123 = select(mask2, t01, t23);
    Array<V, 4> q0123 = select(mask2, q01, q23);
    return Q(q0123 * (rsqrt(t0123) * V(S(0.5))));
}

/// (roll, pitch, yaw); the pitch is clamped to +-pi/2 where |sin(pitch)| >= 1
template <typename Vector3, typename V> inline Vector3 quat_to_euler(const Quaternion<V> &q) {
    using S = scalar_t<V>;
    V q_y_2 = sqr(q.y());
    V sinr_cosp = V(S(2)) * fmadd(q.w(), q.x(), q.y() * q.z());
    V cosr_cosp = fnmadd(V(S(2)), fmadd(q.x(), q.x(), q_y_2), V(S(1)));
    V roll = atan2(sinr_cosp, cosr_cosp);
    V sinp = V(S(2)) * fmsub(q.w(), q.y(), q.z() * q.x());
    V pitch = select(abs(sinp) >= V(S(1)), copysign(V(S(1.57079632679489661923)), sinp), asin(sinp));
    V siny_cosp = V(S(2)) * fmadd(q.w(), q.z(), q.x() * q.y());
    V cosy_cosp = fnmadd(V(S(2)), fmadd(q.z(), q.z(), q_y_2), V(S(1)));
    V yaw = atan2(siny_cosp, cosy_cosp);
    return Vector3(roll, pitch, yaw);
}

} // namespace enoki
