// Device-side scalar math for the HIP kernels.
//
// The parity target is the reference's *CPU* algorithm (include/enoki/array_math.h), not the
// hardware approximations its CUDA backend emits (cuda.h:433-467).  Each function restates the
// published CEPHES-derived algorithm with every rounding step made explicit: the library is
// compiled with -ffp-contract=off, so `a * b + c` is two roundings and only __builtin_fmaf()
// fuses -- exactly the operations the reference spells as enoki::fmadd().
// Citations are file:line in /root/reference.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace ek {
namespace dev {

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// float -> int32 with the x86 cvttps2dq convention the CPU path has (NaN / out of range ->
// 0x80000000); the native v_cvt_i32_f32 saturates instead.
__device__ __forceinline__ int32_t cvtt_i32(float a) {
    return (a > -2147483904.0f && a < 2147483648.0f) ? (int32_t) a : (int32_t) 0x80000000;
}

// Joint sine/cosine, float branch of detail::sincos_approx (array_math.h:261-367):
// octant index j = (trunc(|x| * 4/pi) + 1) & ~1, three-term Cody-Waite reduction written with
// plain operators (:320-323, separately rounded), degree-2 polynomials in z = y^2 (:334-340),
// s = fma(s, y, y), c = fma(c, z, fma(z, -.5, 1)) (:357-358), quadrant swap and sign fix-up by
// xor-ing sign bits (:360-366, mulsign = array_router.h:447).
template <bool Sin, bool Cos>
__device__ __forceinline__ void sincos_f32(float x, float &s_out, float &c_out) {
    float xa = __builtin_fabsf(x);
    int32_t j = cvtt_i32(xa * 1.2732395447351626862f);
    j = (int32_t) (((uint32_t) j + 1u) & ~1u);
    float y = (float) j;

    uint32_t sign_sin = ((uint32_t) j << 29) ^ f2u(x);
    uint32_t sign_cos = (~((uint32_t) j - 2u)) << 29;

    float t = xa - y * 0.78515625f;
    t = t - y * 2.4187564849853515625e-4f;
    t = t - y * 3.77489497744594108e-8f;
    y = t;

    float z = y * y;
    if (xa == __builtin_inff()) z = u2f(0xffffffffu);   // z |= eq(xa, inf)  (:331)

    float z2 = z * z;
    float s = __builtin_fmaf(z2, -1.9515295891e-4f, __builtin_fmaf(z, 8.3321608736e-3f, -1.6666654611e-1f)) * z;
    float c = __builtin_fmaf(z2, 2.443315711809948e-5f,
                             __builtin_fmaf(z, -1.388731625493765e-3f, 4.166664568298827e-2f)) * z;

    s = __builtin_fmaf(s, y, y);
    c = __builtin_fmaf(c, z, __builtin_fmaf(z, -0.5f, 1.0f));

    bool polymask = (j & 2) == 0;
    if (Sin) s_out = u2f(f2u(polymask ? s : c) ^ (sign_sin & 0x80000000u));
    if (Cos) c_out = u2f(f2u(polymask ? c : s) ^ (sign_cos & 0x80000000u));
}

// exp, float branch (array_math.h:711-776): n = floor(fma(log2e, x, .5)); two-step fnmadd range
// reduction (:742-743); poly5 in Estrin form (:752-754, array_math.h:49-55);
// fma(z, xr*xr, xr + 1) (:755); ldexp by adding n to the exponent field (:677-680);
// overflow / underflow selects (:774-775).
__device__ __forceinline__ float exp_f32(float x) {
    bool overflow = x > 88.3762588501f, underflow = x < -88.3762588501f;
    float n = __builtin_floorf(__builtin_fmaf(1.4426950408889634073599f, x, 0.5f));
    float xr = x;
    xr = __builtin_fmaf(-n, 0.693359375f, xr);
    xr = __builtin_fmaf(-n, -2.12194440e-4f, xr);
    float x2 = xr * xr, x4 = x2 * x2;
    float z = __builtin_fmaf(x2, __builtin_fmaf(xr, 8.3334519073e-3f, 4.1665795894e-2f),
                             __builtin_fmaf(x4, __builtin_fmaf(xr, 1.9875691500e-4f, 1.3981999507e-3f),
                                            __builtin_fmaf(xr, 1.6666665459e-1f, 5.0000001201e-1f)));
    z = __builtin_fmaf(z, xr * xr, xr + 1.0f);
    int32_t ni = cvtt_i32(n);
    float r = z * u2f(((uint32_t) ni + 0x7fu) << 23);
    return overflow ? __builtin_inff() : (underflow ? 0.0f : r);
}

// log, float branch without AVX-512 (array_math.h:778-898): frexp by bit masks (:682-709; the
// reference applies it to x itself, so denormals take the "normal" path with exponent -127),
// sqrt(1/2) split (:815-822), poly8 (:825-829, array_math.h:75-82), two-term ln2 recombination
// (:834-836), specials (:894-897).
__device__ __forceinline__ float log_f32(float x) {
    bool valid = x >= 0.0f;
    uint32_t xi = f2u(x);
    uint32_t exponent_bits = xi & 0x7f800000u;
    bool is_normal = (x != 0.0f) && (exponent_bits != 0x7f800000u);
    int32_t exponent_i = (int32_t) (exponent_bits >> 23) - 0x7f;
    uint32_t mantissa = (xi & ~0x7f800000u) | 0x3f000000u;
    float xm = u2f(is_normal ? mantissa : xi);
    float e = (float) (is_normal ? exponent_i : 0);

    bool ge = xm >= 0.70710678118654752440f;
    if (ge) e += 1.0f;
    xm += (ge ? 0.0f : xm) - 1.0f;

    float z = xm * xm;
    float x2 = z, x4 = x2 * x2, x8 = x4 * x4;
    float y = __builtin_fmaf(
        x4,
        __builtin_fmaf(x2, __builtin_fmaf(xm, -1.1514610310e-1f, 1.1676998740e-1f),
                       __builtin_fmaf(xm, -1.2420140846e-1f, 1.4249322787e-1f)),
        __builtin_fmaf(x2, __builtin_fmaf(xm, -1.6668057665e-1f, 2.0000714765e-1f),
                       __builtin_fmaf(xm, -2.4999993993e-1f, 3.3333331174e-1f) + 7.0376836292e-2f * x8));
    y *= xm * z;
    y = __builtin_fmaf(e, -2.12194440e-4f, y);
    z = __builtin_fmaf(z, -0.5f, xm + y);
    float r = __builtin_fmaf(e, 0.693359375f, z);

    if (x == __builtin_inff()) r = __builtin_inff();
    if (x == 0.0f) r = -__builtin_inff();
    return valid ? r : u2f(0xffffffffu);
}

// safe_mul / safe_fmadd: CPU branch of src/autodiff/autodiff.cpp:1191-1221
// (w == 0 || g == 0) ? 0 : w*g     resp.    (w == 0 || g == 0) ? acc : fma(w, g, acc)
template <typename T> __device__ __forceinline__ T safe_mul(T w, T g) {
    return (w == T(0) || g == T(0)) ? T(0) : w * g;
}
__device__ __forceinline__ float safe_fmadd(float w, float g, float acc) {
    return (w == 0.0f || g == 0.0f) ? acc : __builtin_fmaf(w, g, acc);
}
__device__ __forceinline__ double safe_fmadd(double w, double g, double acc) {
    return (w == 0.0 || g == 0.0) ? acc : __builtin_fma(w, g, acc);
}

} // namespace dev
} // namespace ek
