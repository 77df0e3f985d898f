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

// The same for an argument that is never negative (|x| * 4/pi), where the quantity that is USED is (j + 1) & ~1: the native
// conversion saturates at 0x7fffffff, and (0x7fffffff + 1) & ~1 = 0x80000000 = (0x80000000 + 1) & ~1 -- the octant index of an
// out-of-range argument comes out the same without the two compares and the select of cvtt_i32 (12 of the ~110 issue cycles of a
// sincos on gfx950: compares and selects are 4-cycle instructions, profiles/probe_valu_r06.txt).  A NaN converts to 0 instead of
// 0x80000000: bits 1 and 2 of the octant index (the quadrant swap, the two sign flips) are clear either way, and the NaN reaches
// the result through the reduction and the polynomials whatever j is -- same bits (tests/test_kernels_gpu.py: specials).
__device__ __forceinline__ int32_t cvt_sat_i32(float a) {
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(a));
    return r;
}

// Joint sine/cosine, float branch of detail::sincos_approx (array_math.h:261-367):
// octant index j = (trunc(|x| * 4/pi) + 1) & ~1, three-term Cody-Waite reduction written with
// plain operators (:320-323, separately rounded), degree-2 polynomials in z = y^2 (:334-340),
// s = fma(s, y, y), c = fma(c, z, fma(z, -.5, 1)) (:357-358), quadrant swap and sign fix-up by
// xor-ing sign bits (:360-366, mulsign = array_router.h:447).
// Finite: the caller guarantees a finite x (k_bucket_pair_forward_adjoint's piece guard) -- no fix-up of an infinity.
template <bool Sin, bool Cos, bool Finite = false>
__device__ __forceinline__ void sincos_f32(float x, float &s_out, float &c_out) {
    float xa = __builtin_fabsf(x);
    int32_t j = cvt_sat_i32(xa * 1.2732395447351626862f);
    j = (int32_t) (((uint32_t) j + 1u) & ~1u);
    float y = (float) j;

    uint32_t sign_sin = ((uint32_t) j << 29) ^ f2u(x);
    uint32_t sign_cos = (~((uint32_t) j - 2u)) << 29;

    float t = xa - y * 0.78515625f;
    t = t - y * 2.4187564849853515625e-4f;
    t = t - y * 3.77489497744594108e-8f;
    y = t;

    float z = y * y;
    // z |= eq(xa, inf)  (:331) -- a branch that no wave takes on finite data instead of a select per element
    if constexpr (!Finite) {
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(xa == __builtin_inff()) != 0, 0)) {
            if (xa == __builtin_inff()) z = u2f(0xffffffffu);
        }
    }

    float z2 = z * z;
    float s = __builtin_fmaf(z2, -1.9515295891e-4f, __builtin_fmaf(z, 8.3321608736e-3f, -1.6666654611e-1f)) * z;
    float c = __builtin_fmaf(z2, 2.443315711809948e-5f,
                             __builtin_fmaf(z, -1.388731625493765e-3f, 4.166664568298827e-2f)) * z;

    s = __builtin_fmaf(s, y, y);
    c = __builtin_fmaf(c, z, __builtin_fmaf(z, -0.5f, 1.0f));

    // quadrant swap (polymask = (j & 2) == 0 ? s : c) and sign fix-up without the condition register: bit 1 of j spread over a word
    // selects bit by bit, and x ^ (sign & 0x80000000) is one three-input operation as well -- v_bfe_i32 + 4 x v_bitop3_b32 (~11
    // issue cycles of a SIMD) against v_and, v_cmp, the wait states of vcc, two v_cndmask and two v_bitop3 with a scalar operand
    // (~23, profiles/probe_valu_r06.txt).  The sign-bit constant is pinned in a vector register: as a scalar operand it halves
    // the rate of the instruction.
    const uint32_t swap = (uint32_t) (((int32_t) ((uint32_t) j << 30)) >> 31);     // all ones: take the other polynomial
    uint32_t sign_bit;
    asm("v_mov_b32 %0, 0x80000000" : "=v"(sign_bit));
    if (Sin) s_out = u2f(__builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(swap, f2u(c), f2u(s), 0xCA), sign_sin, sign_bit, 0x78));
    if (Cos) c_out = u2f(__builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(swap, f2u(s), f2u(c), 0xCA), sign_cos, sign_bit, 0x78));
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
    // (saturating conversion: wherever n leaves the int32 range the overflow / underflow select below discards r, and a NaN gives
    //  (0 + 0x7f) << 23 = 1.0 = (0x80000000 + 0x7f) << 23 -- same bits as the cvttps2dq convention without its compares)
    int32_t ni = cvt_sat_i32(n);
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

// ------------------------------------------------------------------------------------------------
//  Second wave: tan/cot, asin/acos/atan/atan2, cbrt, pow, hyperbolic and
//  inverse hyperbolic functions -- float branches of array_math.h.
//
//  Polynomials use the reference's Estrin groupings (array_math.h:25-105), one overload per
//  degree; c[k] multiplies x^k.  Coefficients the reference writes as double literals and then
//  narrows (`S(c)`) are written the same way here so that both sides round the literal once.
//  Where the reference calls rcp() the AVX2 path uses rcpps + one Newton step
//  (array_avx.h:324-357, machine dependent); the kernels divide exactly, so tan/cot/sinh/cosh/tanh
//  are parity class C (few ulp), everything else in this block is bit-exact (class A).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

__device__ __forceinline__ float estrin(float x, float c0, float c1, float c2) {
    float x2 = x * x;
    return fma_(x2, c2, fma_(x, c1, c0));
}
__device__ __forceinline__ float estrin(float x, float c0, float c1, float c2, float c3) {
    float x2 = x * x;
    return fma_(x2, fma_(x, c3, c2), fma_(x, c1, c0));
}
__device__ __forceinline__ float estrin(float x, float c0, float c1, float c2, float c3, float c4) {
    float x2 = x * x, x4 = x2 * x2;
    return fma_(x2, fma_(x, c3, c2), fma_(x, c1, c0) + c4 * x4);
}
__device__ __forceinline__ float estrin(float x, float c0, float c1, float c2, float c3, float c4, float c5) {
    float x2 = x * x, x4 = x2 * x2;
    return fma_(x2, fma_(x, c3, c2), fma_(x4, fma_(x, c5, c4), fma_(x, c1, c0)));
}
__device__ __forceinline__ float estrin(float x, float c0, float c1, float c2, float c3, float c4, float c5,
                                        float c6) {
    float x2 = x * x, x4 = x2 * x2;
    return fma_(x4, fma_(x2, c6, fma_(x, c5, c4)), fma_(x2, fma_(x, c3, c2), fma_(x, c1, c0)));
}

__device__ __forceinline__ float copysign_f32(float mag, float sgn) {       // array_router.h:397
    return u2f((f2u(mag) & 0x7fffffffu) | (f2u(sgn) & 0x80000000u));
}
__device__ __forceinline__ float mulsign_f32(float v, float sgn) {          // array_router.h:449
    return u2f(f2u(v) ^ (f2u(sgn) & 0x80000000u));
}
__device__ __forceinline__ float nan_mask_f32() { return u2f(0xffffffffu); }

// frexp / ldexp by exponent-field surgery (array_math.h:677-709); zero, denormal-as-zero-exponent,
// inf and NaN pass through frexp unchanged with exponent 0
__device__ __forceinline__ float frexp_f32(float x, float &e_out) {
    uint32_t xi = f2u(x), eb = xi & 0x7f800000u;
    bool normal = (x != 0.0f) && (eb != 0x7f800000u);
    e_out = (float) (normal ? (int32_t) (eb >> 23) - 0x7f : 0);
    return u2f(normal ? ((xi & ~0x7f800000u) | 0x3f000000u) : xi);
}
__device__ __forceinline__ float ldexp_f32(float x, float e) {
    return x * u2f(((uint32_t) cvtt_i32(e) + 0x7fu) << 23);
}

// tan / cot: detail::tancot_approx (array_math.h:369-442)
template <bool Tan> __device__ __forceinline__ float tancot_f32(float x) {
    float xa = __builtin_fabsf(x);
    int32_t j = cvtt_i32(xa * (float) 1.2732395447351626862);
    j = (int32_t) (((uint32_t) j + 1u) & ~1u);
    float y = (float) j;

    float t = xa - y * (float) 0.78515625;
    t = t - y * (float) 2.4187564849853515625e-4;
    t = t - y * (float) 3.77489497744594108e-8;
    y = t;

    float z = y * y;
    if (xa == __builtin_inff()) z = nan_mask_f32();

    float r = estrin(z, (float) 3.33331568548e-1, (float) 1.33387994085e-1, (float) 5.34112807005e-2,
                     (float) 2.44301354525e-2, (float) 3.11992232697e-3, (float) 9.38540185543e-3);
    r = fma_(r, z * y, y);

    bool recip = Tan ? (j & 2) != 0 : (j & 2) == 0;
    if (xa < (float) 1e-4) r = y;
    if (recip) r = 1.0f / r;

    uint32_t sign = ((uint32_t) j << 30) ^ f2u(x);
    return u2f(f2u(r) ^ (sign & 0x80000000u));
}

// shared front end of asin / acos (array_math.h:489-506, :571-585); c0 differs in the last digits
__device__ __forceinline__ float asin_core_f32(float x, float c0, bool &big) {
    float xa = __builtin_fabsf(x), x2 = x * x;
    big = xa > 0.5f;
    float x1 = 0.5f * (1.0f - xa);
    float x3 = big ? x1 : x2;
    float x4 = big ? __builtin_sqrtf(x1) : xa;
    float z1 = estrin(x3, c0, 7.4953002686e-2f, 4.5470025998e-2f, 2.4181311049e-2f, 4.2163199048e-2f);
    return fma_(z1, x3 * x4, x4);
}

__device__ __forceinline__ float asin_f32(float x) {                        // array_math.h:474-553
    bool big;
    float z1 = asin_core_f32(x, 1.6666752422e-1f, big);
    float r = big ? (float) 1.57079632679489661923 - (z1 + z1) : z1;
    return copysign_f32(r, x);
}

__device__ __forceinline__ float acos_f32(float x) {                        // array_math.h:555-601
    bool big;
    float z1 = asin_core_f32(x, 1.666675242e-1f, big);
    float z2 = z1 + z1;
    if (x < 0.0f) z2 = (float) 3.14159265358979323846 - z2;
    float z3 = (float) 1.57079632679489661923 - copysign_f32(z1, x);
    return big ? z2 : z3;
}

// atan2(y, x): minimax fit in min/max form (array_math.h:603-664); min/max keep the x86
// operand convention of BinaryOp (second operand returned on unordered compares)
__device__ __forceinline__ float atan2_f32(float y, float x) {
    float abs_x = __builtin_fabsf(x), abs_y = __builtin_fabsf(y);
    float min_val = abs_x < abs_y ? abs_x : abs_y;      // min(abs_y, abs_x)
    float max_val = abs_y > abs_x ? abs_y : abs_x;      // max(abs_x, abs_y)
    float scale = 1.0f / max_val;
    float scaled_min = min_val * scale;
    float z = scaled_min * scaled_min;

    float t = estrin(z, (float) 0.99999934166683966009, (float) -0.33326497518773606976,
                     (float) 0.19881342388439013552, (float) -0.13486708938456973185,
                     (float) 0.083863120428809689910, (float) -0.037006525670417265220,
                     (float) 0.0078613793713198150252);
    t = t * scaled_min;
    if (abs_y > abs_x) t = (float) 1.57079632679489661923 - t;
    if (x < 0.0f) t = (float) 3.14159265358979323846 - t;
    float r = y < 0.0f ? u2f(f2u(t) ^ 0x80000000u) : t;
    return max_val != 0.0f ? r : 0.0f;
}

// cbrt (array_math.h:900-954): polynomial seed on the frexp mantissa, exponent / 3 with a
// cbrt(2), cbrt(4) fix-up, one Newton step
__device__ __forceinline__ float cbrt_f32(float x) {
    const float CBRT2 = (float) 1.25992104989487316477, CBRT4 = (float) 1.58740105196819947475,
                THIRD = (float) (1.0 / 3.0);
    float xa = __builtin_fabsf(x), xe;
    float xm = frexp_f32(xa, xe);
    xe += 1.0f;

    float xea = __builtin_fabsf(xe), xea1 = __builtin_floorf(xea * THIRD), rem = fma_(-xea1, 3.0f, xea);

    xm = estrin(xm, (float) 0.40238979564544752126924, (float) 1.1399983354717293273738,
                (float) -0.95438224771509446525043, (float) 0.54664601366395524503440,
                (float) -0.13466110473359520655053);

    float f1 = xe >= 0.0f ? CBRT2 : 1.0f / CBRT2, f2 = xe >= 0.0f ? CBRT4 : 1.0f / CBRT4;
    float f = rem == 1.0f ? f1 : f2;
    if (rem != 0.0f) xm *= f;

    float r = ldexp_f32(xm, mulsign_f32(xea1, xe));
    r = mulsign_f32(r, x);
    r -= (r - (x / (r * r))) * THIRD;
    return __builtin_fabsf(x) < __builtin_inff() ? r : x;
}

__device__ __forceinline__ float pow_f32(float x, float y) { return exp_f32(log_f32(x) * y); }   // array_math.h:956-958

__device__ __forceinline__ float fmod_f32(float x, float y) {               // array_math.h:1381-1383
    return fma_(-__builtin_truncf(x / y), y, x);
}

// odd polynomial used by sinh / sincosh for |x| <= 1 (array_math.h:1028-1031)
__device__ __forceinline__ float sinh_small_f32(float x) {
    float x2 = x * x;
    return fma_(estrin(x2, (float) 1.66667160211e-1, (float) 8.33028376239e-3, (float) 2.03721912945e-4), x2 * x, x);
}

__device__ __forceinline__ float sinh_f32(float x) {                        // array_math.h:997-1046
    const bool big = __builtin_fabsf(x) > 1.0f;
    float r_big = 0.0f, r_small = 0.0f;
    if (__any(big)) { float e0 = exp_f32(x), e1 = 1.0f / e0; r_big = (e0 - e1) * 0.5f; }     // wave-uniform early outs, like
    if (!__all(big)) r_small = sinh_small_f32(x);                                             // any_nested() at :1017-1024
    return big ? r_big : r_small;
}

__device__ __forceinline__ float cosh_f32(float x) {                        // array_math.h:1048-1065
    float e0 = exp_f32(x), e1 = 1.0f / e0;
    return (e0 + e1) * 0.5f;
}

__device__ __forceinline__ void sincosh_f32(float x, float &s, float &c) {  // array_math.h:1067-1127
    float e0 = exp_f32(x), e1 = 1.0f / e0;
    s = __builtin_fabsf(x) > 1.0f ? (e0 - e1) * 0.5f : sinh_small_f32(x);
    c = 0.5f * (e0 + e1);
}

__device__ __forceinline__ float tanh_f32(float x) {                        // array_math.h:1129-1179
    const bool big = __builtin_fabsf(x) >= 0.625f;
    float r_small = 0.0f, r_big = 0.0f;
    if (!__all(big)) {
        float x2 = x * x;
        r_small = estrin(x2, (float) -3.33332819422e-1, (float) 1.33314422036e-1, (float) -5.37397155531e-2,
                         (float) 2.06390887954e-2, (float) -5.70498872745e-3);
        r_small = fma_(r_small, x2 * x, x);
    }
    if (__any(big)) {
        float e = exp_f32(x + x), e2 = 1.0f / (e + 1.0f);
        r_big = 1.0f - (e2 + e2);
    }
    return big ? r_big : r_small;
}

__device__ __forceinline__ float asinh_f32(float x) {                       // array_math.h:1185-1237
    float x2 = x * x, xa = __builtin_fabsf(x);
    bool big = xa >= (float) 0.51, huge = xa >= (float) 1e10;
    float r_small = 0.0f, r_big = 0.0f;
    if (!__all(big)) {
        r_small = estrin(x2, (float) -1.6666288134e-1, (float) 7.4847586088e-2, (float) -4.2699340972e-2,
                         (float) 2.0122003309e-2);
        r_small = fma_(r_small, x2 * x, x);
    }
    if (__any(big)) {
        r_big = log_f32(xa + (huge ? 0.0f : __builtin_sqrtf(x2 + 1.0f)));
        if (huge) r_big += (float) 0.693147180559945309417;
    }
    return big ? copysign_f32(r_big, x) : r_small;
}

__device__ __forceinline__ float acosh_f32(float x) {                       // array_math.h:1239-1293
    float x1 = x - 1.0f;
    bool big = x1 >= (float) 0.49, huge = x1 >= (float) 1e10;
    float r_small = 0.0f, r_big = 0.0f;
    if (!__all(big)) {
        r_small = estrin(x1, (float) 1.4142135263e+0, (float) -1.1784741703e-1, (float) 2.6454905019e-2,
                         (float) -7.5272886713e-3, (float) 1.7596881071e-3);
        r_small *= __builtin_sqrtf(x1);
        if (x1 < 0.0f) r_small = nan_mask_f32();
    }
    if (__any(big)) {
        r_big = log_f32(x + (huge ? 0.0f : __builtin_sqrtf(fma_(x, x, -1.0f))));
        if (huge) r_big += (float) 0.693147180559945309417;
    }
    return big ? r_big : r_small;
}

__device__ __forceinline__ float atanh_f32(float x) {                       // array_math.h:1295-1348
    float xa = __builtin_fabsf(x), x2 = x * x;
    const bool big = xa >= 0.5f;
    float r_small = 0.0f, r_big = 0.0f;
    if (!__all(big)) {
        r_small = estrin(x2, (float) 3.33337300303e-1, (float) 1.99782164500e-1, (float) 1.46691431730e-1,
                         (float) 8.24370301058e-2, (float) 1.81740078349e-1);
        r_small = fma_(r_small, x2 * x, x);
    }
    if (__any(big)) r_big = log_f32((1.0f + xa) / (1.0f - xa)) * 0.5f;
    return big ? copysign_f32(r_big, x) : r_small;
}

// ------------------------------------------------------------------------------------------------
//  float64 branches of sin/cos/sincos, exp, log (array_math.h:325-327, 342-354, 745-746, 761-771,
//  838-887).  Same structure as the f32 functions above; integers are 64 bit, the int conversion
//  follows cvttpd2qq / the scalar fallback of the AVX2 path (indefinite = 0x8000000000000000).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t d2u(double f) { return (uint64_t) __double_as_longlong(f); }
__device__ __forceinline__ double u2d(uint64_t u) { return __longlong_as_double((long long) u); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

__device__ __forceinline__ int64_t cvtt_i64(double a) {
    return (a > -9223372036854777856.0 && a < 9223372036854775808.0) ? (int64_t) a : (int64_t) 0x8000000000000000ull;
}

__device__ __forceinline__ double estrin(double x, double c0, double c1, double c2) {
    double x2 = x * x;
    return fma_(x2, c2, fma_(x, c1, c0));
}
__device__ __forceinline__ double estrin(double x, double c0, double c1, double c2, double c3) {
    double x2 = x * x;
    return fma_(x2, fma_(x, c3, c2), fma_(x, c1, c0));
}
__device__ __forceinline__ double estrin(double x, double c0, double c1, double c2, double c3, double c4, double c5) {
    double x2 = x * x, x4 = x2 * x2;
    return fma_(x2, fma_(x, c3, c2), fma_(x4, fma_(x, c5, c4), fma_(x, c1, c0)));
}

template <bool Sin, bool Cos>
__device__ __forceinline__ void sincos_f64(double x, double &s_out, double &c_out) {
    double xa = __builtin_fabs(x);
    int64_t j = cvtt_i64(xa * 1.2732395447351626862);
    j = (int64_t) (((uint64_t) j + 1ull) & ~1ull);
    double y = (double) j;

    uint64_t sign_sin = ((uint64_t) j << 61) ^ d2u(x);
    uint64_t sign_cos = (~((uint64_t) j - 2ull)) << 61;

    double t = xa - y * 7.85398125648498535156e-1;
    t = t - y * 3.77489470793079817668e-8;
    t = t - y * 2.69515142907905952645e-15;
    y = t;

    double z = y * y;
    if (xa == __builtin_inf()) z = u2d(~0ull);

    double s = estrin(z, -1.66666666666666307295e-1, 8.33333333332211858878e-3, -1.98412698295895385996e-4,
                      2.75573136213857245213e-6, -2.50507477628578072866e-8, 1.58962301576546568060e-10) * z;
    double c = estrin(z, 4.16666666666665929218e-2, -1.38888888888730564116e-3, 2.48015872888517045348e-5,
                      -2.75573141792967388112e-7, 2.08757008419747316778e-9, -1.13585365213876817300e-11) * z;

    s = fma_(s, y, y);
    c = fma_(c, z, fma_(z, -0.5, 1.0));

    bool polymask = (j & 2) == 0;
    const uint64_t sb = 0x8000000000000000ull;
    if (Sin) s_out = u2d(d2u(polymask ? s : c) ^ (sign_sin & sb));
    if (Cos) c_out = u2d(d2u(polymask ? c : s) ^ (sign_cos & sb));
}

__device__ __forceinline__ double exp_f64(double x) {
    bool overflow = x > 7.0943613930310391424428e2, underflow = x < -7.0943613930310391424428e2;
    double n = __builtin_floor(fma_(1.4426950408889634073599, x, 0.5));
    double xr = x;
    xr = fma_(-n, 6.93145751953125e-1, xr);
    xr = fma_(-n, 1.42860682030941723212e-6, xr);
    double z = xr * xr;
    double p = estrin(z, 9.99999999999999999910e-1, 3.02994407707441961300e-2, 1.26177193074810590878e-4) * xr;
    double q = estrin(z, 2.00000000000000000009e0, 2.27265548208155028766e-1, 2.52448340349684104192e-3,
                      3.00198505138664455042e-6);
    double pq = p / (q - p);
    z = pq + pq + 1.0;
    double r = z * u2d(((uint64_t) cvtt_i64(n) + 0x3ffull) << 52);
    return overflow ? __builtin_inf() : (underflow ? 0.0 : r);
}

__device__ __forceinline__ double log_f64(double x) {
    bool valid = x >= 0.0;
    uint64_t xi = d2u(x), exponent_bits = xi & 0x7ff0000000000000ull;
    bool is_normal = (x != 0.0) && (exponent_bits != 0x7ff0000000000000ull);
    int64_t exponent_i = (int64_t) (exponent_bits >> 52) - 0x3ff;
    uint64_t mantissa = (xi & ~0x7ff0000000000000ull) | 0x3fe0000000000000ull;
    double xm = u2d(is_normal ? mantissa : xi);
    double e = (double) (is_normal ? exponent_i : 0);

    bool e_big = __builtin_fabs(e) > 2.0;              // evaluated before the sqrt(1/2) adjustment (:815)
    bool ge = xm >= 0.70710678118654752440;
    if (ge) e += 1.0;

    double r_big = 0.0, r_small = 0.0;
    if (__any(e_big)) {       // |e| > 2: log(x) = z + z^3 P(z)/Q(z), z = 2(x-1)/(x+1)   (:842-861, any_nested early out)
        double zb = xm - 0.5;
        if (ge) zb -= 0.5;
        double yb = 0.5 * (ge ? xm : zb) + 0.5;
        double x2b = zb / yb;
        double z2 = x2b * x2b;
        double rb = x2b * (z2 * estrin(z2, -6.41409952958715622951e1, 1.63866645699558079767e1, -7.89580278884799154124e-1) /
                           estrin(z2, -7.69691943550460008604e2, 3.12093766372244180303e2, -3.56722798256324312549e1, 1.0));
        r_big = fma_(-e, 2.121944400546905827679e-4, rb) + x2b;
    }
    if (!__all(e_big)) {      // otherwise: log(1+x) = x - x^2/2 + x^3 P(x)/Q(x)          (:863-884)
        double x2s = (ge ? xm : xm + xm) - 1.0;
        double zs = x2s * x2s;
        double ys = x2s * (zs * estrin(x2s, 7.70838733755885391666e0, 1.79368678507819816313e1, 1.44989225341610930846e1,
                                       4.70579119878881725854e0, 4.97494994976747001425e-1, 1.01875663804580931796e-4) /
                           estrin(x2s, 2.31251620126765340583e1, 7.11544750618563894466e1, 8.29875266912776603211e1,
                                  4.52279145837532221105e1, 1.12873587189167450590e1, 1.0));
        ys = fma_(-e, 2.121944400546905827679e-4, ys);
        r_small = x2s + fma_(-0.5, zs, ys);
    }

    double r = fma_(e, 0.693359375, e_big ? r_big : r_small);
    if (x == __builtin_inf()) r = __builtin_inf();
    if (x == 0.0) r = -__builtin_inf();
    return valid ? r : u2d(~0ull);
}

// ------------------------------------------------------------------------------------------------
//  float64 branches of the second wave (array_math.h:405-441, 509-551, 591-600, 640-663, 950-951,
//  1033-1041, 1160-1166, 1216-1225, 1270-1280, 1327-1337).  Rational CEPHES approximations; rcp()
//  appears in tan/cot/sinh/cosh/tanh only (class C), the rest is bit-exact.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double estrin(double x, double c0, double c1, double c2, double c3, double c4) {
    double x2 = x * x, x4 = x2 * x2;
    return fma_(x2, fma_(x, c3, c2), fma_(x, c1, c0) + c4 * x4);
}
__device__ __forceinline__ double estrin(double x, double c0, double c1, double c2, double c3, double c4, double c5,
                                         double c6) {
    double x2 = x * x, x4 = x2 * x2;
    return fma_(x4, fma_(x2, c6, fma_(x, c5, c4)), fma_(x2, fma_(x, c3, c2), fma_(x, c1, c0)));
}
__device__ __forceinline__ double copysign_f64(double mag, double sgn) {
    return u2d((d2u(mag) & 0x7fffffffffffffffull) | (d2u(sgn) & 0x8000000000000000ull));
}
__device__ __forceinline__ double mulsign_f64(double v, double sgn) {
    return u2d(d2u(v) ^ (d2u(sgn) & 0x8000000000000000ull));
}
__device__ __forceinline__ double frexp_f64(double x, double &e_out) {
    uint64_t xi = d2u(x), eb = xi & 0x7ff0000000000000ull;
    bool normal = (x != 0.0) && (eb != 0x7ff0000000000000ull);
    e_out = (double) (normal ? (int64_t) (eb >> 52) - 0x3ff : 0);
    return u2d(normal ? ((xi & ~0x7ff0000000000000ull) | 0x3fe0000000000000ull) : xi);
}
__device__ __forceinline__ double ldexp_f64(double x, double e) {
    return x * u2d(((uint64_t) cvtt_i64(e) + 0x3ffull) << 52);
}

template <bool Tan> __device__ __forceinline__ double tancot_f64(double x) {
    double xa = __builtin_fabs(x);
    int64_t j = cvtt_i64(xa * 1.2732395447351626862);
    j = (int64_t) (((uint64_t) j + 1ull) & ~1ull);
    double y = (double) j;
    double t = xa - y * 7.85398125648498535156e-1;
    t = t - y * 3.77489470793079817668e-8;
    t = t - y * 2.69515142907905952645e-15;
    y = t;
    double z = y * y;
    if (xa == __builtin_inf()) z = u2d(~0ull);
    double r = estrin(z, -1.79565251976484877988e7, 1.15351664838587416140e6, -1.30936939181383777646e4) /
               estrin(z, -5.38695755929454629881e7, 2.50083801823357915839e7, -1.32089234440210967447e6,
                      1.36812963470692954678e4, 1.0);
    r = fma_(r, z * y, y);
    bool recip = Tan ? (j & 2) != 0 : (j & 2) == 0;
    if (xa < 1e-4) r = y;
    if (recip) r = 1.0 / r;
    uint64_t sign = ((uint64_t) j << 62) ^ d2u(x);
    return u2d(d2u(r) ^ (sign & 0x8000000000000000ull));
}

__device__ __forceinline__ double asin_f64(double x) {
    double xa = __builtin_fabs(x), x2 = x * x;
    bool big = xa > 0.625;
    // |x| > 0.625: asin(1 - t) = pi/2 - sqrt(2 t) (1 + R(t))
    double zz = 1.0 - xa;
    double p = estrin(zz, 2.853665548261061424989e1, -2.556901049652824852289e1, 6.968710824104713396794e0,
                      -5.634242780008963776856e-1, 2.967721961301243206100e-3) /
               estrin(zz, 3.424398657913078477438e2, -3.838770957603691357202e2, 1.470656354026814941758e2,
                      -2.194779531642920639778e1, 1.0) * zz;
    zz = __builtin_sqrt(zz + zz);
    double zb = 0.78539816339744830962 - zz;
    double r_big = zb - fma_(zz, p, -6.123233995736765886130e-17) + 0.78539816339744830962;
    // otherwise a rational in x^2
    double zs = estrin(x2, -8.198089802484824371615e0, 1.956261983317594739197e1, -1.626247967210700244449e1,
                       5.444622390564711410273e0, -6.019598008014123785661e-1, 4.253011369004428248960e-3) /
                estrin(x2, -4.918853881490881290097e1, 1.395105614657485689735e2, -1.471791292232726029859e2,
                       7.049610280856842141659e1, -1.474091372988853791896e1, 1.0) * x2;
    zs = fma_(xa, zs, xa);
    if (xa < 1e-8) zs = xa;
    return copysign_f64(big ? r_big : zs, x);
}

__device__ __forceinline__ double acos_f64(double x) {
    bool mask = x > 0.5;
    double y = asin_f64(mask ? __builtin_sqrt(fma_(-0.5, x, 0.5)) : x);
    return mask ? y + y : 0.78539816339744830962 - y + 6.123233995736765886130e-17 + 0.78539816339744830962;
}

__device__ __forceinline__ double atan2_f64(double y, double x) {
    double abs_x = __builtin_fabs(x), abs_y = __builtin_fabs(y);
    double min_val = abs_x < abs_y ? abs_x : abs_y, max_val = abs_y > abs_x ? abs_y : abs_x;
    double scale = 1.0 / max_val, scaled_min = min_val * scale, z = scaled_min * scaled_min;
    double t = estrin(z, 9.9999999999999999419e-1, 2.50554429737833465113e0, 2.28289058385464073556e0,
                      9.20960512187107069075e-1, 1.59189681028889623410e-1, 9.35911604785115940726e-3,
                      8.07005540507283419124e-5) /
               estrin(z, 1.00000000000000000000e0, 2.83887763071166519407e0, 3.02918312742541450749e0,
                      1.50576983803701596773e0, 3.49719171130492192607e-1, 3.29968942624402204199e-2,
                      8.26619391703564168942e-4);
    t = t * scaled_min;
    if (abs_y > abs_x) t = 1.57079632679489661923 - t;
    if (x < 0.0) t = 3.14159265358979323846 - t;
    double r = y < 0.0 ? u2d(d2u(t) ^ 0x8000000000000000ull) : t;
    return max_val != 0.0 ? r : 0.0;
}

__device__ __forceinline__ double cbrt_f64(double x) {
    const double CBRT2 = 1.25992104989487316477, CBRT4 = 1.58740105196819947475, THIRD = 1.0 / 3.0;
    double xa = __builtin_fabs(x), xe;
    double xm = frexp_f64(xa, xe);
    xe += 1.0;
    double xea = __builtin_fabs(xe), xea1 = __builtin_floor(xea * THIRD), rem = fma_(-xea1, 3.0, xea);
    xm = estrin(xm, 0.40238979564544752126924, 1.1399983354717293273738, -0.95438224771509446525043,
                0.54664601366395524503440, -0.13466110473359520655053);
    double f1 = xe >= 0.0 ? CBRT2 : 1.0 / CBRT2, f2 = xe >= 0.0 ? CBRT4 : 1.0 / CBRT4;
    double f = rem == 1.0 ? f1 : f2;
    if (rem != 0.0) xm *= f;
    double r = ldexp_f64(xm, mulsign_f64(xea1, xe));
    r = mulsign_f64(r, x);
    r -= (r - (x / (r * r))) * THIRD;
    r -= (r - (x / (r * r))) * THIRD;
    return __builtin_fabs(x) < __builtin_inf() ? r : x;
}

__device__ __forceinline__ double sinh_small_f64(double x) {
    double x2 = x * x;
    return fma_(estrin(x2, -3.51754964808151394800e5, -1.15614435765005216044e4, -1.63725857525983828727e2,
                       -7.89474443963537015605e-1) /
                estrin(x2, -2.11052978884890840399e6, 3.61578279834431989373e4, -2.77711081420602794433e2, 1.0),
                x2 * x, x);
}
__device__ __forceinline__ double sinh_f64(double x) {
    double e0 = exp_f64(x), e1 = 1.0 / e0;
    return __builtin_fabs(x) > 1.0 ? (e0 - e1) * 0.5 : sinh_small_f64(x);
}
__device__ __forceinline__ double cosh_f64(double x) {
    double e0 = exp_f64(x), e1 = 1.0 / e0;
    return (e0 + e1) * 0.5;
}
__device__ __forceinline__ void sincosh_f64(double x, double &s, double &c) {
    double e0 = exp_f64(x), e1 = 1.0 / e0;
    s = __builtin_fabs(x) > 1.0 ? (e0 - e1) * 0.5 : sinh_small_f64(x);
    c = 0.5 * (e0 + e1);
}
__device__ __forceinline__ double tanh_f64(double x) {
    double x2 = x * x;
    double r_small = estrin(x2, -1.61468768441708447952e3, -9.92877231001918586564e1, -9.64399179425052238628e-1) /
                     estrin(x2, 4.84406305325125486048e3, 2.23548839060100448583e3, 1.12811678491632931402e2, 1.0);
    r_small = fma_(r_small, x2 * x, x);
    double e = exp_f64(x + x), e2 = 1.0 / (e + 1.0);
    double r_big = 1.0 - (e2 + e2);
    return __builtin_fabs(x) >= 0.625 ? r_big : r_small;
}
__device__ __forceinline__ double asinh_f64(double x) {
    double x2 = x * x, xa = __builtin_fabs(x);
    bool big = xa >= 0.533, huge = xa >= 1e20;
    double r_small = 0.0, r_big = 0.0;
    if (!__all(big)) {
        r_small = estrin(x2, -5.56682227230859640450e0, -9.09030533308377316566e0, -4.37390226194356683570e0,
                         -5.91750212056387121207e-1, -4.33231683752342103572e-3) /
                  estrin(x2, 3.34009336338516356383e1, 6.95722521337257608734e1, 4.86042483805291788324e1,
                         1.28757002067426453537e1, 1.0);
        r_small = fma_(r_small, x2 * x, x);
    }
    if (__any(big)) {
        r_big = log_f64(xa + (huge ? 0.0 : __builtin_sqrt(x2 + 1.0)));
        if (huge) r_big += 0.693147180559945309417;
    }
    return big ? copysign_f64(r_big, x) : r_small;
}
__device__ __forceinline__ double acosh_f64(double x) {
    double x1 = x - 1.0;
    bool big = x1 >= 0.49, huge = x1 >= 1e10;
    double r_small = estrin(x1, 1.10855947270161294369E5, 1.08102874834699867335E5, 3.43989375926195455866E4,
                            3.94726656571334401102E3, 1.18801130533544501356E2) /
                     estrin(x1, 7.83869920495893927727E4, 8.29725251988426222434E4, 2.97683430363289370382E4,
                            4.15352677227719831579E3, 1.86145380837903397292E2, 1.0);
    r_small *= __builtin_sqrt(x1);
    if (x1 < 0.0) r_small = u2d(~0ull);
    double r_big = log_f64(x + (huge ? 0.0 : __builtin_sqrt(fma_(x, x, -1.0))));
    if (huge) r_big += 0.693147180559945309417;
    return big ? r_big : r_small;
}
__device__ __forceinline__ double atanh_f64(double x) {
    double xa = __builtin_fabs(x), x2 = x * x;
    const bool big = xa >= 0.5;
    double r_small = 0.0, r_big = 0.0;
    if (!__all(big)) {
        r_small = estrin(x2, -3.09092539379866942570e1, 6.54566728676544377376e1, -4.61252884198732692637e1,
                         1.20426861384072379242e1, -8.54074331929669305196e-1) /
                  estrin(x2, -9.27277618139601130017e1, 2.52006675691344555838e2, -2.49839401325893582852e2,
                         1.08938092147140262656e2, -1.95638849376911654834e1, 1.0);
        r_small = fma_(r_small, x2 * x, x);
    }
    if (__any(big)) r_big = log_f64((1.0 + xa) / (1.0 - xa)) * 0.5;
    return big ? copysign_f64(r_big, x) : r_small;
}
__device__ __forceinline__ double pow_f64(double x, double y) { return exp_f64(log_f64(x) * y); }

// safe_mul / safe_fmadd: CPU branch of src/autodiff/autodiff.cpp:1191-1221
// (w == 0 || g == 0) ? 0 : w*g     resp.    (w == 0 || g == 0) ? acc : fma(w, g, acc)
template <typename T> __device__ __forceinline__ T safe_mul(T w, T g) {
    return (w == T(0) || g == T(0)) ? T(0) : w * g;
}
#if defined(__HIP_DEVICE_COMPILE__) && (defined(__gfx900__) || defined(__gfx906__) || defined(__gfx908__) || defined(__gfx90a__) || \
                                       defined(__gfx940__) || defined(__gfx941__) || defined(__gfx942__) || defined(__gfx950__))
// (GFX9 only: the mnemonic does not exist on gfx10+, where user kernels that include this header through enoki::vectorize get the
// generic template above.  Denormal operands follow the kernel's fp32 denormal mode exactly like the plain multiplication of the
// generic form does -- the library builds with denormals on; tests/test_kernels_gpu.py covers denormal operands.)
// float: v_mul_legacy_f32 IS this function -- (+-0) * anything = +0, an IEEE multiplication otherwise -- in one instruction
// instead of two compares, a multiplication and a select (bit-exact against the literal form: tests/test_kernels_gpu.py)
template <> __device__ __forceinline__ float safe_mul<float>(float w, float g) {
    float r;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(w), "v"(g));
    return r;
}
#endif
__device__ __forceinline__ float safe_fmadd(float w, float g, float acc) {
    return (w == 0.0f || g == 0.0f) ? acc : __builtin_fmaf(w, g, acc);
}
__device__ __forceinline__ double safe_fmadd(double w, double g, double acc) {
    return (w == 0.0 || g == 0.0) ? acc : __builtin_fma(w, g, acc);
}

} // namespace dev
} // namespace ek
