/*
 * oracle/enoki_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's scalar/AVX2 `DynamicArray<Packet<T,8>>`
 * path for the north-star hot path.  It is the *checker* for
 * the HIP kernels: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it.  The product (libenoki-hip.so and everything above it) never links,
 * imports or calls anything in this directory.
 *
 * Parity pinning: every function here is compared bit-for-bit (integer / mask / index /
 * IEEE ops and sin/cos/exp/log) or within the documented class bounds (rcp, rsqrt,
 * order-dependent reductions) against oracle/_ref/libenoki_ref.so -- the UNMODIFIED
 * reference headers compiled with the pinned flags -- by tests/test_oracle_vs_ref.py,
 * and against the committed fixtures in tests/golden/ (generated from that same build by
 * tests/golden/make_golden.py).
 *
 * Build: gcc -std=c11 -O2 -mavx2 -mfma -ffp-contract=off -fno-math-errno (oracle/Makefile).
 * -ffp-contract=off matters: only the explicit fmaf() calls below are fused, mirroring the
 * explicit enoki::fmadd() calls of the reference.
 *
 * All file:line citations are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define PKT 8 /* Packet<float,8>: BASELINE.json config 1 */

/* type codes shared with include/enoki_hip.h */
enum { T_BOOL = 0, T_I32 = 1, T_U32 = 2, T_I64 = 3, T_U64 = 4, T_F32 = 5, T_F64 = 6 };

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

/* ------------------------------------------------------------------------------------ */
/*  f32 building blocks, AVX2 packet semantics                                            */
/* ------------------------------------------------------------------------------------ */

/* include/enoki/array_avx.h min_/max_ issue _mm256_min_ps(b, a) / _mm256_max_ps(b, a) (operands
   swapped), i.e. (b < a) ? b : a -- the FIRST enoki operand wins when the compare is unordered. */
static inline float min_ps(float a, float b) { return b < a ? b : a; }
static inline float max_ps(float a, float b) { return b > a ? b : a; }

/* cvttps2dq: truncation; NaN / out of range -> 0x80000000 ("integer indefinite") */
static inline int32_t cvtt_f32_i32(float a) {
    if (!(a > -2147483904.0f && a < 2147483648.0f)) return INT32_MIN;
    return (int32_t) a;
}

/* detail::sincos_approx<Sin, Cos>, include/enoki/array_math.h:261-367 (float branch) */
static void sincos_f32(float x, float *s_out, float *c_out) {
    float xa = fabsf(x);                                         /* :297 */
    int32_t j = cvtt_f32_i32(xa * 1.2732395447351626862f);       /* :300 IntArray j(xa * 4/pi) */
    j = (int32_t) (((uint32_t) j + 1u) & ~1u);                   /* :303 */
    float y = (float) j;                                         /* :306 */

    uint32_t sign_sin = ((uint32_t) j << 29) ^ f2u(x);           /* :313 sl<29>(j) ^ x */
    uint32_t sign_cos = (~((uint32_t) j - 2u)) << 29;            /* :316 */

    /* :320-323 -- operators, not fmadd: four separately rounded mul/sub with contraction off */
    float t = xa - y * 0.78515625f;
    t = t - y * 2.4187564849853515625e-4f;
    t = t - y * 3.77489497744594108e-8f;
    y = t;

    float z = y * y;                                             /* :330 */
    if (xa == INFINITY) z = u2f(f2u(z) | 0xffffffffu);           /* :331 z |= eq(xa, inf) */

    /* poly2 (array_math.h:25-29): fmadd(x2, c2, fmadd(x, c1, c0)) with x2 = x*x */
    float z2 = z * z;
    float s = fmaf(z2, -1.9515295891e-4f, fmaf(z, 8.3321608736e-3f, -1.6666654611e-1f)) * z;   /* :334 */
    float c = fmaf(z2, 2.443315711809948e-5f, fmaf(z, -1.388731625493765e-3f, 4.166664568298827e-2f)) * z; /* :338 */

    s = fmaf(s, y, y);                                           /* :357 */
    c = fmaf(c, z, fmaf(z, -0.5f, 1.0f));                        /* :358 */

    int polymask = (j & 2) == 0;                                 /* :360 */
    if (s_out) *s_out = u2f(f2u(polymask ? s : c) ^ (sign_sin & 0x80000000u)); /* :363 mulsign */
    if (c_out) *c_out = u2f(f2u(polymask ? c : s) ^ (sign_cos & 0x80000000u)); /* :366 */
}

/* exp, include/enoki/array_math.h:711-776 (float branch) */
static float exp_f32(float x) {
    int overflow = x > 88.3762588501f, underflow = x < -88.3762588501f;   /* :727-731 */
    float n = floorf(fmaf(1.4426950408889634073599f, x, 0.5f));             /* :738 */
    float xr = x;
    xr = fmaf(-n, 0.693359375f, xr);                                         /* :742 fnmadd */
    xr = fmaf(-n, -2.12194440e-4f, xr);                                      /* :743 */
    /* poly5 (array_math.h:49-55) */
    float x2 = xr * xr, x4 = x2 * x2;
    float z = fmaf(x2, fmaf(xr, 8.3334519073e-3f, 4.1665795894e-2f),
                   fmaf(x4, fmaf(xr, 1.9875691500e-4f, 1.3981999507e-3f),
                        fmaf(xr, 1.6666665459e-1f, 5.0000001201e-1f)));      /* :752-754 */
    z = fmaf(z, xr * xr, xr + 1.0f);                                         /* :755 */
    /* ldexp (:677-680): z * reinterpret(sl<23>(int(n) + 0x7f)) */
    int32_t ni = cvtt_f32_i32(n);
    float scale = u2f(((uint32_t) ni + 0x7fu) << 23);
    float r = z * scale;
    return overflow ? INFINITY : (underflow ? 0.0f : r);                     /* :774-775 */
}

/* log, include/enoki/array_math.h:778-898 (float branch, !has_avx512f) */
static float log_f32(float x) {
    int valid = x >= 0.0f;                                                   /* :797 */
    /* frexp(x) (:682-709) -- note the reference calls frexp on x, not on max(x, limit) (:811) */
    uint32_t xi = f2u(x);
    uint32_t exponent_bits = xi & 0x7f800000u;
    int is_normal = (x != 0.0f) && (exponent_bits != 0x7f800000u);
    int32_t exponent_i = (int32_t) (exponent_bits >> 23) - 0x7f;
    uint32_t mantissa = (xi & ~0x7f800000u) | f2u(0.5f);
    float xm = u2f(is_normal ? mantissa : xi);
    float e = (float) (is_normal ? exponent_i : 0);

    int ge = xm >= 0.70710678118654752440f;                                  /* :815 */
    if (ge) e += 1.0f;                                                       /* :818 */

    xm += (ge ? 0.0f : xm) - 1.0f;                                           /* :822 xm += (xm & ~mask) - 1 */

    float z = xm * xm;                                                       /* :824 */
    /* poly8 (array_math.h:75-82) */
    float x2 = z, x4 = x2 * x2, x8 = x4 * x4;
    float y = fmaf(x4, fmaf(x2, fmaf(xm, -1.1514610310e-1f, 1.1676998740e-1f),
                              fmaf(xm, -1.2420140846e-1f, 1.4249322787e-1f)),
                   fmaf(x2, fmaf(xm, -1.6668057665e-1f, 2.0000714765e-1f),
                        fmaf(xm, -2.4999993993e-1f, 3.3333331174e-1f) + 7.0376836292e-2f * x8)); /* :825-829 */
    y *= xm * z;                                                             /* :832 */
    y = fmaf(e, -2.12194440e-4f, y);                                         /* :834 */
    z = fmaf(z, -0.5f, xm + y);                                              /* :835 */
    float r = fmaf(e, 0.693359375f, z);                                      /* :836 */

    if (x == INFINITY) r = INFINITY;                                         /* :894 */
    if (x == 0.0f) r = -INFINITY;                                            /* :895 */
    return valid ? r : u2f(f2u(r) | 0xffffffffu);                            /* :897 r | ~valid_mask */
}

/* ------------------------------------------------------------------------------------ */
/*  Second-wave f32 functions (array_math.h:369-442, 466-668, 900-958, 997-1348, 1381)     */
/*                                                                                        */
/*  Estrin evaluation with the groupings of array_math.h:25-105; c[k] multiplies x^k and  */
/*  is narrowed from the double literal like the reference's `S(c)`.  rcp() is spelled as */
/*  an exact division (class C versus the reference's rcpps + Newton step).               */
/* ------------------------------------------------------------------------------------ */
static inline float P2(float x, const double *c) {                           /* :25-29 */
    float x2 = x * x;
    return fmaf(x2, (float) c[2], fmaf(x, (float) c[1], (float) c[0]));
}
static inline float P3(float x, const double *c) {                           /* :31-37 */
    float x2 = x * x;
    return fmaf(x2, fmaf(x, (float) c[3], (float) c[2]), fmaf(x, (float) c[1], (float) c[0]));
}
static inline float P4f(float x, const float *c) {                           /* :39-45 */
    float x2 = x * x, x4 = x2 * x2;
    return fmaf(x2, fmaf(x, c[3], c[2]), fmaf(x, c[1], c[0]) + c[4] * x4);
}
static inline float P4(float x, const double *c) {
    float f[5] = { (float) c[0], (float) c[1], (float) c[2], (float) c[3], (float) c[4] };
    return P4f(x, f);
}
static inline float P5(float x, const double *c) {                           /* :47-54 */
    float x2 = x * x, x4 = x2 * x2;
    return fmaf(x2, fmaf(x, (float) c[3], (float) c[2]),
                fmaf(x4, fmaf(x, (float) c[5], (float) c[4]), fmaf(x, (float) c[1], (float) c[0])));
}
static inline float P6(float x, const double *c) {                           /* :56-63 */
    float x2 = x * x, x4 = x2 * x2;
    return fmaf(x4, fmaf(x2, (float) c[6], fmaf(x, (float) c[5], (float) c[4])),
                fmaf(x2, fmaf(x, (float) c[3], (float) c[2]), fmaf(x, (float) c[1], (float) c[0])));
}

static inline float copysign_ps(float mag, float sgn) { return u2f((f2u(mag) & 0x7fffffffu) | (f2u(sgn) & 0x80000000u)); }
static inline float mulsign_ps(float v, float sgn) { return u2f(f2u(v) ^ (f2u(sgn) & 0x80000000u)); }

/* frexp :682-709, ldexp :677-680 */
static inline float frexp_f32(float x, float *e) {
    uint32_t xi = f2u(x), eb = xi & 0x7f800000u;
    int normal = (x != 0.0f) && (eb != 0x7f800000u);
    *e = (float) (normal ? (int32_t) (eb >> 23) - 0x7f : 0);
    return u2f(normal ? ((xi & ~0x7f800000u) | 0x3f000000u) : xi);
}
static inline float ldexp_f32(float x, float e) {
    return x * u2f(((uint32_t) cvtt_f32_i32(e) + 0x7fu) << 23);
}

static float tancot_f32(float x, int want_tan) {                             /* :369-442 */
    static const double c[6] = { 3.33331568548e-1, 1.33387994085e-1, 5.34112807005e-2,
                                 2.44301354525e-2, 3.11992232697e-3, 9.38540185543e-3 };
    float xa = fabsf(x);                                                     /* :388 */
    int32_t j = cvtt_f32_i32(xa * (float) 1.2732395447351626862);            /* :391 */
    j = (int32_t) (((uint32_t) j + 1u) & ~1u);                               /* :394 */
    float y = (float) j;                                                     /* :397 */
    float t = xa - y * (float) 0.78515625;                                   /* :401-403 */
    t = t - y * (float) 2.4187564849853515625e-4;
    t = t - y * (float) 3.77489497744594108e-8;
    y = t;
    float z = y * y;                                                         /* :410 */
    if (xa == INFINITY) z = u2f(0xffffffffu);                                /* :411 */
    float r = P5(z, c);                                                      /* :415-420 */
    r = fmaf(r, z * y, y);                                                   /* :432 */
    int recip = want_tan ? (j & 2) != 0 : (j & 2) == 0;                      /* :434-435 */
    if (xa < (float) 1e-4) r = y;                                            /* :436 */
    if (recip) r = 1.0f / r;                                                 /* :437 rcp(), class C */
    uint32_t sign = ((uint32_t) j << 30) ^ f2u(x);                           /* :439 */
    return u2f(f2u(r) ^ (sign & 0x80000000u));                               /* :441 */
}

static float asin_acos_core(float x, float c0, int *big) {                   /* :489-506 / :571-585 */
    float c[5] = { c0, 7.4953002686e-2f, 4.5470025998e-2f, 2.4181311049e-2f, 4.2163199048e-2f };
    float xa = fabsf(x), x2 = x * x;
    *big = xa > 0.5f;
    float x1 = 0.5f * (1.0f - xa);
    float x3 = *big ? x1 : x2, x4 = *big ? sqrtf(x1) : xa;
    float z1 = P4f(x3, c);
    return fmaf(z1, x3 * x4, x4);
}

static float asin_f32(float x) {                                             /* :474-553 */
    int big;
    float z1 = asin_acos_core(x, 1.6666752422e-1f, &big);
    float r = big ? (float) M_PI_2 - (z1 + z1) : z1;                         /* :508 */
    return copysign_ps(r, x);                                                /* :552 */
}

static float acos_f32(float x) {                                             /* :555-601 */
    int big;
    float z1 = asin_acos_core(x, 1.666675242e-1f, &big);
    float z2 = z1 + z1;                                                      /* :586 */
    if (x < 0.0f) z2 = (float) M_PI - z2;                                    /* :587 */
    float z3 = (float) M_PI_2 - copysign_ps(z1, x);                          /* :589 */
    return big ? z2 : z3;
}

static float atan2_f32(float y, float x) {                                   /* :603-664, called as atan2(y, x) */
    static const double c[7] = { 0.99999934166683966009, -0.33326497518773606976, +0.19881342388439013552,
                                 -0.13486708938456973185, +0.083863120428809689910, -0.037006525670417265220,
                                 0.0078613793713198150252 };
    float abs_x = fabsf(x), abs_y = fabsf(y);                                /* :619-620 */
    float min_val = min_ps(abs_y, abs_x), max_val = max_ps(abs_x, abs_y);    /* :621-622 */
    float scale = 1.0f / max_val, scaled_min = min_val * scale;              /* :623-624 */
    float z = scaled_min * scaled_min;
    float t = P6(z, c) * scaled_min;                                         /* :633-657 */
    if (abs_y > abs_x) t = (float) M_PI_2 - t;                               /* :659 */
    if (x < 0.0f) t = (float) M_PI - t;                                      /* :660 */
    float r = y < 0.0f ? u2f(f2u(t) ^ 0x80000000u) : t;                      /* :661 */
    return max_val != 0.0f ? r : 0.0f;                                       /* :662 */
}

static float cbrt_f32(float x) {                                             /* :900-954 */
    static const double c[5] = { 0.40238979564544752126924, 1.1399983354717293273738, -0.95438224771509446525043,
                                 0.54664601366395524503440, -0.13466110473359520655053 };
    const float CBRT2 = (float) 1.25992104989487316477, CBRT4 = (float) 1.58740105196819947475,
                THIRD = (float) (1.0 / 3.0);
    float xa = fabsf(x), xe;
    float xm = frexp_f32(xa, &xe);                                           /* :923 */
    xe += 1.0f;
    float xea = fabsf(xe), xea1 = floorf(xea * THIRD), rem = fmaf(-xea1, 3.0f, xea);   /* :926-928 */
    xm = P4(xm, c);                                                          /* :932-936 */
    float f1 = xe >= 0.0f ? CBRT2 : 1.0f / CBRT2, f2 = xe >= 0.0f ? CBRT4 : 1.0f / CBRT4;
    float f = rem == 1.0f ? f1 : f2;                                         /* :938-940 */
    if (rem != 0.0f) xm *= f;                                                /* :942 */
    float r = ldexp_f32(xm, mulsign_ps(xea1, xe));                           /* :944 */
    r = mulsign_ps(r, x);
    r -= (r - (x / (r * r))) * THIRD;                                        /* :948 */
    return fabsf(x) < INFINITY ? r : x;                                      /* :953 */
}

static float sinh_small_f32(float x) {                                       /* :1025-1031 */
    static const double c[3] = { 1.66667160211e-1, 8.33028376239e-3, 2.03721912945e-4 };
    float x2 = x * x;
    return fmaf(P2(x2, c), x2 * x, x);
}
static float sinh_f32(float x) {                                             /* :997-1046 */
    float e0 = exp_f32(x), e1 = 1.0f / e0;                                   /* rcp(), class C */
    return fabsf(x) > 1.0f ? (e0 - e1) * 0.5f : sinh_small_f32(x);
}
static float cosh_f32(float x) {                                             /* :1048-1065 */
    float e0 = exp_f32(x), e1 = 1.0f / e0;
    return (e0 + e1) * 0.5f;
}
static float tanh_f32(float x) {                                             /* :1129-1179 */
    static const double c[5] = { -3.33332819422e-1, 1.33314422036e-1, -5.37397155531e-2, 2.06390887954e-2,
                                 -5.70498872745e-3 };
    float x2 = x * x;
    float r_small = fmaf(P4(x2, c), x2 * x, x);                              /* :1154-1169 */
    float e = exp_f32(x + x), e2 = 1.0f / (e + 1.0f);                        /* :1173-1174 rcp(), class C */
    float r_big = 1.0f - (e2 + e2);
    return fabsf(x) >= 0.625f ? r_big : r_small;
}
static float asinh_f32(float x) {                                            /* :1185-1237 */
    static const double c[4] = { -1.6666288134e-1, 7.4847586088e-2, -4.2699340972e-2, 2.0122003309e-2 };
    float x2 = x * x, xa = fabsf(x);
    int big = xa >= (float) 0.51, huge = xa >= (float) 1e10;                 /* :1206-1207 */
    float r_small = fmaf(P3(x2, c), x2 * x, x);                              /* :1211-1227 */
    float r_big = log_f32(xa + (huge ? 0.0f : sqrtf(x2 + 1.0f)));            /* :1231 */
    if (huge) r_big += (float) M_LN2;                                        /* :1232 */
    return big ? copysign_ps(r_big, x) : r_small;
}
static float acosh_f32(float x) {                                            /* :1239-1293 */
    static const double c[5] = { 1.4142135263e+0, -1.1784741703e-1, 2.6454905019e-2, -7.5272886713e-3,
                                 1.7596881071e-3 };
    float x1 = x - 1.0f;
    int big = x1 >= (float) 0.49, huge = x1 >= (float) 1e10;                 /* :1259-1260 */
    float r_small = P4(x1, c) * sqrtf(x1);                                   /* :1264-1283 */
    if (x1 < 0.0f) r_small = u2f(0xffffffffu);                               /* :1284 */
    float r_big = log_f32(x + (huge ? 0.0f : sqrtf(fmaf(x, x, -1.0f))));     /* :1288 */
    if (huge) r_big += (float) M_LN2;
    return big ? r_big : r_small;
}
static float atanh_f32(float x) {                                            /* :1295-1348 */
    static const double c[5] = { 3.33337300303e-1, 1.99782164500e-1, 1.46691431730e-1, 8.24370301058e-2,
                                 1.81740078349e-1 };
    float xa = fabsf(x), x2 = x * x;
    float r_small = fmaf(P4(x2, c), x2 * x, x);                              /* :1321-1339 */
    float r_big = log_f32((1.0f + xa) / (1.0f - xa)) * 0.5f;                 /* :1343 */
    return xa >= 0.5f ? copysign_ps(r_big, x) : r_small;
}

/* ------------------------------------------------------------------------------------ */
/*  float64 branches of sincos / exp / log (array_math.h:325-327, 342-354, 745-771, 838-887) */
/* ------------------------------------------------------------------------------------ */
static inline int64_t cvtt_f64_i64(double a) {                /* cvttsd2si: out of range / NaN -> INT64_MIN */
    if (!(a > -9223372036854777856.0 && a < 9223372036854775808.0)) return INT64_MIN;
    return (int64_t) a;
}
static inline double D2(double x, const double *c) { double x2 = x * x; return fma(x2, c[2], fma(x, c[1], c[0])); }
static inline double D3(double x, const double *c) {
    double x2 = x * x;
    return fma(x2, fma(x, c[3], c[2]), fma(x, c[1], c[0]));
}
static inline double D5(double x, const double *c) {
    double x2 = x * x, x4 = x2 * x2;
    return fma(x2, fma(x, c[3], c[2]), fma(x4, fma(x, c[5], c[4]), fma(x, c[1], c[0])));
}

static void sincos_f64(double x, double *s_out, double *c_out) {             /* :261-367, double branch */
    static const double cs[6] = { -1.66666666666666307295e-1, 8.33333333332211858878e-3, -1.98412698295895385996e-4,
                                  2.75573136213857245213e-6, -2.50507477628578072866e-8, 1.58962301576546568060e-10 };
    static const double cc[6] = { 4.16666666666665929218e-2, -1.38888888888730564116e-3, 2.48015872888517045348e-5,
                                  -2.75573141792967388112e-7, 2.08757008419747316778e-9, -1.13585365213876817300e-11 };
    double xa = fabs(x);
    int64_t j = cvtt_f64_i64(xa * 1.2732395447351626862);                    /* :301 */
    j = (int64_t) (((uint64_t) j + 1ull) & ~1ull);                           /* :304 */
    double y = (double) j;
    uint64_t sign_sin = ((uint64_t) j << 61) ^ d2u(x);                       /* :314 */
    uint64_t sign_cos = (~((uint64_t) j - 2ull)) << 61;                      /* :317 */
    double t = xa - y * 7.85398125648498535156e-1;                           /* :325-327 */
    t = t - y * 3.77489470793079817668e-8;
    t = t - y * 2.69515142907905952645e-15;
    y = t;
    double z = y * y;
    if (xa == INFINITY) z = u2d(~0ull);                                      /* :331 */
    double s = D5(z, cs) * z, c = D5(z, cc) * z;                             /* :342-354 */
    s = fma(s, y, y);                                                        /* :357 */
    c = fma(c, z, fma(z, -0.5, 1.0));                                        /* :358 */
    int polymask = (j & 2) == 0;
    if (s_out) *s_out = u2d(d2u(polymask ? s : c) ^ (sign_sin & 0x8000000000000000ull));
    if (c_out) *c_out = u2d(d2u(polymask ? c : s) ^ (sign_cos & 0x8000000000000000ull));
}

static double exp_f64(double x) {                                            /* :711-776, double branch */
    static const double cp[3] = { 9.99999999999999999910e-1, 3.02994407707441961300e-2, 1.26177193074810590878e-4 };
    static const double cq[4] = { 2.00000000000000000009e0, 2.27265548208155028766e-1, 2.52448340349684104192e-3,
                                  3.00198505138664455042e-6 };
    int overflow = x > 7.0943613930310391424428e2, underflow = x < -7.0943613930310391424428e2;
    double n = floor(fma(1.4426950408889634073599, x, 0.5));                 /* :739 */
    double xr = fma(-n, 6.93145751953125e-1, x);                             /* :745 */
    xr = fma(-n, 1.42860682030941723212e-6, xr);                             /* :746 */
    double z = xr * xr;
    double p = D2(z, cp) * xr, q = D3(z, cq);                                /* :761-768 */
    double pq = p / (q - p);
    z = pq + pq + 1.0;                                                       /* :770-771 */
    double r = z * u2d(((uint64_t) cvtt_f64_i64(n) + 0x3ffull) << 52);       /* ldexp :677-680 */
    return overflow ? INFINITY : (underflow ? 0.0 : r);
}

static double log_f64(double x) {                                            /* :778-898, double branch */
    static const double pb[3] = { -6.41409952958715622951e1, 1.63866645699558079767e1, -7.89580278884799154124e-1 };
    static const double qb[4] = { -7.69691943550460008604e2, 3.12093766372244180303e2, -3.56722798256324312549e1, 1.0 };
    static const double ps[6] = { 7.70838733755885391666e0, 1.79368678507819816313e1, 1.44989225341610930846e1,
                                  4.70579119878881725854e0, 4.97494994976747001425e-1, 1.01875663804580931796e-4 };
    static const double qs[6] = { 2.31251620126765340583e1, 7.11544750618563894466e1, 8.29875266912776603211e1,
                                  4.52279145837532221105e1, 1.12873587189167450590e1, 1.0 };
    int valid = x >= 0.0;
    uint64_t xi = d2u(x), eb = xi & 0x7ff0000000000000ull;                   /* frexp :682-709 */
    int normal = (x != 0.0) && (eb != 0x7ff0000000000000ull);
    double xm = u2d(normal ? ((xi & ~0x7ff0000000000000ull) | 0x3fe0000000000000ull) : xi);
    double e = (double) (normal ? (int64_t) (eb >> 52) - 0x3ff : 0);
    int e_big = fabs(e) > 2.0;                                               /* :815 (before the adjustment) */
    int ge = xm >= 0.70710678118654752440;
    if (ge) e += 1.0;                                                        /* :819 */

    double zb = xm - 0.5;                                                    /* :844-860 */
    if (ge) zb -= 0.5;
    double yb = 0.5 * (ge ? xm : zb) + 0.5;
    double x2b = zb / yb, z2 = x2b * x2b;
    double rb = x2b * (z2 * D2(z2, pb) / D3(z2, qb));
    double r_big = fma(-e, 2.121944400546905827679e-4, rb) + x2b;

    double x2s = (ge ? xm : xm + xm) - 1.0;                                  /* :865-883 */
    double zs = x2s * x2s;
    double ys = x2s * (zs * D5(x2s, ps) / D5(x2s, qs));
    ys = fma(-e, 2.121944400546905827679e-4, ys);
    double r_small = x2s + fma(-0.5, zs, ys);

    double r = fma(e, 0.693359375, e_big ? r_big : r_small);                 /* :886-887 */
    if (x == INFINITY) r = INFINITY;
    if (x == 0.0) r = -INFINITY;
    return valid ? r : u2d(~0ull);
}

/* ------------------------------------------------------------------------------------ */
/*  float64 branches of the second wave (array_math.h:405-441, 509-551, 591-600, 640-663, */
/*  950-951, 1033-1041, 1160-1166, 1216-1225, 1270-1280, 1327-1337)                       */
/* ------------------------------------------------------------------------------------ */
static inline double D4(double x, const double *c) {
    double x2 = x * x, x4 = x2 * x2;
    return fma(x2, fma(x, c[3], c[2]), fma(x, c[1], c[0]) + c[4] * x4);
}
static inline double D6(double x, const double *c) {
    double x2 = x * x, x4 = x2 * x2;
    return fma(x4, fma(x2, c[6], fma(x, c[5], c[4])), fma(x2, fma(x, c[3], c[2]), fma(x, c[1], c[0])));
}
static inline double copysign_pd(double m, double s) { return u2d((d2u(m) & 0x7fffffffffffffffull) | (d2u(s) & 0x8000000000000000ull)); }
static inline double mulsign_pd(double v, double s) { return u2d(d2u(v) ^ (d2u(s) & 0x8000000000000000ull)); }
static inline double frexp_f64(double x, double *e) {
    uint64_t xi = d2u(x), eb = xi & 0x7ff0000000000000ull;
    int normal = (x != 0.0) && (eb != 0x7ff0000000000000ull);
    *e = (double) (normal ? (int64_t) (eb >> 52) - 0x3ff : 0);
    return u2d(normal ? ((xi & ~0x7ff0000000000000ull) | 0x3fe0000000000000ull) : xi);
}
static inline double ldexp_f64(double x, double e) { return x * u2d(((uint64_t) cvtt_f64_i64(e) + 0x3ffull) << 52); }

static double tancot_f64(double x, int want_tan) {                           /* :369-442 */
    static const double pn[3] = { -1.79565251976484877988e7, 1.15351664838587416140e6, -1.30936939181383777646e4 };
    static const double pd[5] = { -5.38695755929454629881e7, 2.50083801823357915839e7, -1.32089234440210967447e6,
                                  1.36812963470692954678e4, 1.0 };
    double xa = fabs(x);
    int64_t j = cvtt_f64_i64(xa * 1.2732395447351626862);
    j = (int64_t) (((uint64_t) j + 1ull) & ~1ull);
    double y = (double) j;
    double t = xa - y * 7.85398125648498535156e-1;
    t = t - y * 3.77489470793079817668e-8;
    t = t - y * 2.69515142907905952645e-15;
    y = t;
    double z = y * y;
    if (xa == INFINITY) z = u2d(~0ull);
    double r = D2(z, pn) / D4(z, pd);                                        /* :422-429 */
    r = fma(r, z * y, y);
    int recip = want_tan ? (j & 2) != 0 : (j & 2) == 0;
    if (xa < 1e-4) r = y;
    if (recip) r = 1.0 / r;                                                  /* rcp(): class C */
    uint64_t sign = ((uint64_t) j << 62) ^ d2u(x);
    return u2d(d2u(r) ^ (sign & 0x8000000000000000ull));
}

static double asin_f64(double x) {                                           /* :509-552 */
    static const double bn[5] = { 2.853665548261061424989e1, -2.556901049652824852289e1, 6.968710824104713396794e0,
                                  -5.634242780008963776856e-1, 2.967721961301243206100e-3 };
    static const double bd[5] = { 3.424398657913078477438e2, -3.838770957603691357202e2, 1.470656354026814941758e2,
                                  -2.194779531642920639778e1, 1.0 };
    static const double sn[6] = { -8.198089802484824371615e0, 1.956261983317594739197e1, -1.626247967210700244449e1,
                                  5.444622390564711410273e0, -6.019598008014123785661e-1, 4.253011369004428248960e-3 };
    static const double sd[6] = { -4.918853881490881290097e1, 1.395105614657485689735e2, -1.471791292232726029859e2,
                                  7.049610280856842141659e1, -1.474091372988853791896e1, 1.0 };
    const double pio4 = 0.78539816339744830962, more_bits = 6.123233995736765886130e-17;
    double xa = fabs(x), x2 = x * x;
    int big = xa > 0.625;
    double zz = 1.0 - xa;
    double p = D4(zz, bn) / D4(zz, bd) * zz;
    zz = sqrt(zz + zz);
    double z = pio4 - zz;
    double r_big = z - fma(zz, p, -more_bits) + pio4;                        /* :531 */
    double zs = D5(x2, sn) / D5(x2, sd) * x2;
    zs = fma(xa, zs, xa);
    if (xa < 1e-8) zs = xa;
    return copysign_pd(big ? r_big : zs, x);
}

static double acos_f64(double x) {                                           /* :591-600 */
    const double pio4 = 0.78539816339744830962, more_bits = 6.123233995736765886130e-17;
    int mask = x > 0.5;
    double y = asin_f64(mask ? sqrt(fma(-0.5, x, 0.5)) : x);
    return mask ? y + y : pio4 - y + more_bits + pio4;
}

static double atan2_f64(double y, double x) {                                /* :603-664 */
    static const double tn[7] = { 9.9999999999999999419e-1, 2.50554429737833465113e0, 2.28289058385464073556e0,
                                  9.20960512187107069075e-1, 1.59189681028889623410e-1, 9.35911604785115940726e-3,
                                  8.07005540507283419124e-5 };
    static const double td[7] = { 1.00000000000000000000e0, 2.83887763071166519407e0, 3.02918312742541450749e0,
                                  1.50576983803701596773e0, 3.49719171130492192607e-1, 3.29968942624402204199e-2,
                                  8.26619391703564168942e-4 };
    double abs_x = fabs(x), abs_y = fabs(y);
    double min_val = abs_x < abs_y ? abs_x : abs_y, max_val = abs_y > abs_x ? abs_y : abs_x;
    double scale = 1.0 / max_val, scaled_min = min_val * scale, z = scaled_min * scaled_min;
    double t = D6(z, tn) / D6(z, td) * scaled_min;
    if (abs_y > abs_x) t = M_PI_2 - t;
    if (x < 0.0) t = M_PI - t;
    double r = y < 0.0 ? u2d(d2u(t) ^ 0x8000000000000000ull) : t;
    return max_val != 0.0 ? r : 0.0;
}

static double cbrt_f64(double x) {                                           /* :900-954 */
    static const double c[5] = { 0.40238979564544752126924, 1.1399983354717293273738, -0.95438224771509446525043,
                                 0.54664601366395524503440, -0.13466110473359520655053 };
    const double CBRT2 = 1.25992104989487316477, CBRT4 = 1.58740105196819947475, THIRD = 1.0 / 3.0;
    double xa = fabs(x), xe;
    double xm = frexp_f64(xa, &xe);
    xe += 1.0;
    double xea = fabs(xe), xea1 = floor(xea * THIRD), rem = fma(-xea1, 3.0, xea);
    xm = D4(xm, c);
    double f1 = xe >= 0.0 ? CBRT2 : 1.0 / CBRT2, f2 = xe >= 0.0 ? CBRT4 : 1.0 / CBRT4;
    double f = rem == 1.0 ? f1 : f2;
    if (rem != 0.0) xm *= f;
    double r = ldexp_f64(xm, mulsign_pd(xea1, xe));
    r = mulsign_pd(r, x);
    r -= (r - (x / (r * r))) * THIRD;
    r -= (r - (x / (r * r))) * THIRD;                                        /* :950-951 */
    return fabs(x) < INFINITY ? r : x;
}

static double sinh_small_f64(double x) {                                     /* :1033-1041 */
    static const double n_[4] = { -3.51754964808151394800e5, -1.15614435765005216044e4, -1.63725857525983828727e2,
                                  -7.89474443963537015605e-1 };
    static const double d_[4] = { -2.11052978884890840399e6, 3.61578279834431989373e4, -2.77711081420602794433e2, 1.0 };
    double x2 = x * x;
    return fma(D3(x2, n_) / D3(x2, d_), x2 * x, x);
}
static double sinh_f64(double x) { double e0 = exp_f64(x), e1 = 1.0 / e0; return fabs(x) > 1.0 ? (e0 - e1) * 0.5 : sinh_small_f64(x); }
static double cosh_f64(double x) { double e0 = exp_f64(x), e1 = 1.0 / e0; return (e0 + e1) * 0.5; }
static double tanh_f64(double x) {                                           /* :1160-1178 */
    static const double n_[3] = { -1.61468768441708447952e3, -9.92877231001918586564e1, -9.64399179425052238628e-1 };
    static const double d_[4] = { 4.84406305325125486048e3, 2.23548839060100448583e3, 1.12811678491632931402e2, 1.0 };
    double x2 = x * x;
    double r_small = fma(D2(x2, n_) / D3(x2, d_), x2 * x, x);
    double e = exp_f64(x + x), e2 = 1.0 / (e + 1.0);
    double r_big = 1.0 - (e2 + e2);
    return fabs(x) >= 0.625 ? r_big : r_small;
}
static double asinh_f64(double x) {                                          /* :1202-1236 */
    static const double n_[5] = { -5.56682227230859640450e0, -9.09030533308377316566e0, -4.37390226194356683570e0,
                                  -5.91750212056387121207e-1, -4.33231683752342103572e-3 };
    static const double d_[5] = { 3.34009336338516356383e1, 6.95722521337257608734e1, 4.86042483805291788324e1,
                                  1.28757002067426453537e1, 1.0 };
    double x2 = x * x, xa = fabs(x);
    int big = xa >= 0.533, huge = xa >= 1e20;
    double r_small = fma(D4(x2, n_) / D4(x2, d_), x2 * x, x);
    double r_big = log_f64(xa + (huge ? 0.0 : sqrt(x2 + 1.0)));
    if (huge) r_big += M_LN2;
    return big ? copysign_pd(r_big, x) : r_small;
}
static double acosh_f64(double x) {                                          /* :1256-1292 */
    static const double n_[5] = { 1.10855947270161294369E5, 1.08102874834699867335E5, 3.43989375926195455866E4,
                                  3.94726656571334401102E3, 1.18801130533544501356E2 };
    static const double d_[6] = { 7.83869920495893927727E4, 8.29725251988426222434E4, 2.97683430363289370382E4,
                                  4.15352677227719831579E3, 1.86145380837903397292E2, 1.0 };
    double x1 = x - 1.0;
    int big = x1 >= 0.49, huge = x1 >= 1e10;
    double r_small = D4(x1, n_) / D5(x1, d_) * sqrt(x1);
    if (x1 < 0.0) r_small = u2d(~0ull);
    double r_big = log_f64(x + (huge ? 0.0 : sqrt(fma(x, x, -1.0))));
    if (huge) r_big += M_LN2;
    return big ? r_big : r_small;
}
static double atanh_f64(double x) {                                          /* :1313-1347 */
    static const double n_[5] = { -3.09092539379866942570e1, 6.54566728676544377376e1, -4.61252884198732692637e1,
                                  1.20426861384072379242e1, -8.54074331929669305196e-1 };
    static const double d_[6] = { -9.27277618139601130017e1, 2.52006675691344555838e2, -2.49839401325893582852e2,
                                  1.08938092147140262656e2, -1.95638849376911654834e1, 1.0 };
    double xa = fabs(x), x2 = x * x;
    double r_small = fma(D4(x2, n_) / D5(x2, d_), x2 * x, x);
    double r_big = log_f64((1.0 + xa) / (1.0 - xa)) * 0.5;
    return xa >= 0.5 ? copysign_pd(r_big, x) : r_small;
}

/* ------------------------------------------------------------------------------------ */
/*  Special functions (include/enoki/special.h:22-312): erf, erfc, erfinv, i0e, dawson,   */
/*  erfi, lgamma, tgamma.  Same operation order as the reference's composition; rcp() and */
/*  rsqrt() are spelled as exact divisions (class C versus rcpps / rsqrtps + Newton in    */
/*  float32, identical in float64 where the reference divides too).                       */
/* ------------------------------------------------------------------------------------ */
static inline float sin_only_f32(float x) { float s; sincos_f32(x, &s, NULL); return s; }
static inline double sin_only_f64(double x) { double s; sincos_f64(x, &s, NULL); return s; }

#define ORC_SPECIAL_GENERIC(T, SUF, FMA, EXP, LOG, SQRT, FABS, RINT, SIN)                                             \
    static inline T S4##SUF(T x, const double *c) {                                      /* poly4, array_math.h:39-45 */ \
        T x2 = x * x, x4 = x2 * x2;                                                                                   \
        return FMA(x2, FMA(x, (T) c[3], (T) c[2]), FMA(x, (T) c[1], (T) c[0]) + (T) c[4] * x4);                        \
    }                                                                                                                 \
    static inline T S5##SUF(T x, const double *c) {                                      /* poly5, :47-54 */            \
        T x2 = x * x, x4 = x2 * x2;                                                                                   \
        return FMA(x2, FMA(x, (T) c[3], (T) c[2]), FMA(x4, FMA(x, (T) c[5], (T) c[4]), FMA(x, (T) c[1], (T) c[0])));    \
    }                                                                                                                 \
    static inline T S6##SUF(T x, const double *c) {                                      /* poly6, :56-63 */            \
        T x2 = x * x, x4 = x2 * x2;                                                                                   \
        return FMA(x4, FMA(x2, (T) c[6], FMA(x, (T) c[5], (T) c[4])),                                                  \
                   FMA(x2, FMA(x, (T) c[3], (T) c[2]), FMA(x, (T) c[1], (T) c[0])));                                   \
    }                                                                                                                 \
    static inline T S7##SUF(T x, const double *c) {                                      /* poly7, :65-73 */            \
        T x2 = x * x, x4 = x2 * x2;                                                                                   \
        return FMA(x4, FMA(x2, FMA(x, (T) c[7], (T) c[6]), FMA(x, (T) c[5], (T) c[4])),                                \
                   FMA(x2, FMA(x, (T) c[3], (T) c[2]), FMA(x, (T) c[1], (T) c[0])));                                   \
    }                                                                                                                 \
    static inline T S8##SUF(T x, const double *c) {                                      /* poly8, :75-83 */            \
        T x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;                                                                     \
        return FMA(x4, FMA(x2, FMA(x, (T) c[7], (T) c[6]), FMA(x, (T) c[5], (T) c[4])),                                \
                   FMA(x2, FMA(x, (T) c[3], (T) c[2]), FMA(x, (T) c[1], (T) c[0]) + (T) c[8] * x8));                   \
    }                                                                                                                 \
    static T chbevl##SUF(T x, const double *c, int n) {                                  /* special.h:22-36 */          \
        T b0 = (T) c[0], b1 = 0, b2 = 0;                                                                              \
        for (int i = 0; i < n; ++i) { b2 = b1; b1 = b0; b0 = FMA(x, b1, -(b2 - (T) c[i])); }                           \
        return (b0 - b2) * (T) 0.5;                                                                                   \
    }                                                                                                                 \
    static T i0e##SUF(T x_) {                                                            /* special.h:168-218 */        \
        static const double A[] = { -1.30002500998624804212E-8, 6.04699502254191894932E-8, -2.67079385394061173391E-7, \
            1.11738753912010371815E-6, -4.41673835845875056359E-6, 1.64484480707288970893E-5,                         \
            -5.75419501008210370398E-5, 1.88502885095841655729E-4, -5.76375574538582365885E-4,                        \
            1.63947561694133579842E-3, -4.32430999505057594430E-3, 1.05464603945949983183E-2,                         \
            -2.37374148058994688156E-2, 4.93052842396707084878E-2, -9.49010970480476444210E-2,                        \
            1.71620901522208775349E-1, -3.04682672343198398683E-1, 6.76795274409476084995E-1 };                       \
        static const double B[] = { 3.39623202570838634515E-9, 2.26666899049817806459E-8, 2.04891858946906374183E-7,  \
            2.89137052083475648297E-6, 6.88975834691682398426E-5, 3.36911647825569408990E-3,                          \
            8.04490411014108831608E-1 };                                                                              \
        T x = FABS(x_);                                                                                               \
        if (x > (T) 8) return chbevl##SUF(FMA((T) 32, (T) 1 / x, -(T) 2), B, 7) * ((T) 1 / SQRT(x));                   \
        return chbevl##SUF(FMA(x, (T) 0.5, -(T) 2), A, 18);                                                           \
    }                                                                                                                 \
    static T erfinv##SUF(T x) {                                                          /* special.h:222-246 */        \
        static const double c1[] = { 1.50140941, 0.246640727, -0.00417768164, -0.00125372503, 0.00021858087,          \
                                     -4.39150654e-06, -3.5233877e-06, 3.43273939e-07, 2.81022636e-08 };               \
        static const double c2[] = { 2.83297682, 1.00167406, 0.00943887047, -0.0076224613, 0.00573950773,             \
                                     -0.00367342844, 0.00134934322, 0.000100950558, -0.000200214257 };                \
        T w = -LOG(((T) 1 - x) * ((T) 1 + x));                                                                        \
        T w1 = w - (T) 2.5, w2 = SQRT(w) - (T) 3;                                                                     \
        T p1 = S8##SUF(w1, c1), p2 = S8##SUF(w2, c2);                                                                 \
        return (w < (T) 5 ? p1 : p2) * x;                                                                             \
    }                                                                                                                 \
    static T dawson##SUF(T x) {                                                          /* special.h:249-265 */        \
        static const double cn[] = { 1.00000080272429, 9.18170212243285e-2, 4.25835373536124e-2, 6.0536496345054e-3,  \
                                     9.88555033724111e-4, 3.64943550840577e-5, 1.55942290996993e-5 };                 \
        static const double cd[] = { 1.0, 7.58517175815194e-1, 2.81364355593059e-1, 6.81783097841267e-2,              \
                                     1.13586116798019e-2, 1.92020805811771e-3, 5.74217664074868e-5,                   \
                                     3.11884331363595e-5 };                                                           \
        T x2 = x * x;                                                                                                 \
        return S6##SUF(x2, cn) / S7##SUF(x2, cd) * x;                                                                 \
    }                                                                                                                 \
    static T erfi##SUF(T x) { return (T) 1.12837916709551257390 * dawson##SUF(x) * EXP(x * x); }   /* :268-272 */      \
    static T lgamma##SUF(T x_) {                                                         /* special.h:275-309 */        \
        static const double coeff[7] = { 1.000000000190015, 76.18009172947146, -86.50532032941677, 24.01409824083091, \
                                         -1.231739572450155, 0.1208650973866179e-2, -0.5395239384953e-5 };            \
        const T pi = (T) 3.14159265358979323846;                                                                      \
        int reflect = x_ < (T) 0.5;                                                                                   \
        T x = reflect ? -x_ : x_ - (T) 1, b = x + (T) 5 + (T) 0.5, sum = 0;                                           \
        for (int i = 6; i >= 1; --i) sum += (T) coeff[i] / (x + (T) i);                                               \
        sum += (T) coeff[0];                                                                                          \
        T result = (((T) 0.91893853320467274178 + LOG(sum)) - b) + LOG(b) * (x + (T) 0.5);                             \
        if (reflect) {                                                                                                \
            result = LOG(FABS(pi / SIN(pi * x_))) - result;                                                           \
            if (x_ == RINT(x_)) result = (T) INFINITY;                                                                \
        }                                                                                                             \
        return result;                                                                                                \
    }                                                                                                                 \
    static T tgamma##SUF(T x) { return EXP(lgamma##SUF(x)); }                             /* special.h:312 */

ORC_SPECIAL_GENERIC(float, _sf32, fmaf, exp_f32, log_f32, sqrtf, fabsf, rintf, sin_only_f32)
ORC_SPECIAL_GENERIC(double, _sf64, fma, exp_f64, log_f64, sqrt, fabs, rint, sin_only_f64)

/* erf / erfc: special.h:56-165; *_core = the function before the other one's fix-up (Recurse = false) */
static float erf_core_f32(float x) {
    static const double c[] = { 1.128379165726710e+0, -3.761262582423300e-1, 1.128358514861418e-1, -2.685381193529856e-2,
                                5.188327685732524e-3, -8.010193625184903e-4, 7.853861353153693e-5 };
    return S6_sf32(x * x, c) * x;
}
static float erfc_core_f32(float x) {
    static const double cs[] = { 5.638259427386472e-1, -2.741127028184656e-1, 3.404879937665872e-1, -4.944515323274145e-1,
                                 6.210004621745983e-1, -5.824733027278666e-1, 3.687424674597105e-1, -1.387039388740657e-1,
                                 2.326819970068386e-2 };
    static const double cl[] = { 5.641895067754075e-1, -2.820767439740514e-1, 4.218463358204948e-1, -1.015265279202700e+0,
                                 2.921019019210786e+0, -7.495518717768503e+0, 1.297719955372516e+1, -1.047766399936249e+1 };
    float xa = fabsf(x), z = exp_f32(-x * x), q = 1.0f / xa, y = q * q;
    float r = z * q * (xa > 2.0f ? S7_sf32(y, cl) : S8_sf32(y, cs));
    return x < 0.0f ? 2.0f - r : r;
}
static float erf_f32(float x) { return fabsf(x) > 1.0f ? 1.0f - erfc_core_f32(x) : erf_core_f32(x); }
static float erfc_f32(float x) { return fabsf(x) < 1.0f ? 1.0f - erf_core_f32(x) : erfc_core_f32(x); }

static double erf_core_f64(double x) {
    static const double p[] = { 5.55923013010394962768e4, 7.00332514112805075473e3, 2.23200534594684319226e3,
                                9.00260197203842689217e1, 9.60497373987051638749e0 };
    static const double q[] = { 4.92673942608635921086e4, 2.26290000613890934246e4, 4.59432382970980127987e3,
                                5.21357949780152679795e2, 3.35617141647503099647e1, 1.00000000000000000000e0 };
    double z = x * x;
    return S4_sf64(z, p) / S5_sf64(z, q) * x;
}
static double erfc_core_f64(double x) {
    static const double ps[] = { 5.57535335369399327526e2, 1.02755188689515710272e3, 9.34528527171957607540e2,
                                 5.26445194995477358631e2, 1.96520832956077098242e2, 4.86371970985681366614e1,
                                 7.46321056442269912687e0, 5.64189564831068821977e-1, 2.46196981473530512524e-10 };
    static const double qs[] = { 5.57535340817727675546e2, 1.65666309194161350182e3, 2.24633760818710981792e3,
                                 1.82390916687909736289e3, 9.75708501743205489753e2, 3.54937778887819891062e2,
                                 8.67072140885989742329e1, 1.32281951154744992508e1, 1.00000000000000000000e0 };
    static const double pl[] = { 2.97886665372100240670e0, 7.40974269950448939160e0, 6.16021097993053585195e0,
                                 5.01905042251180477414e0, 1.27536670759978104416e0, 5.64189583547755073984e-1 };
    static const double ql[] = { 3.36907645100081516050e0, 9.60896809063285878198e0, 1.70814450747565897222e1,
                                 1.20489539808096656605e1, 9.39603524938001434673e0, 2.26052863220117276590e0,
                                 1.00000000000000000000e0 };
    double xa = fabs(x), z = exp_f64(-x * x);
    int large = xa > 8.0;
    double r = (z * (large ? S5_sf64(xa, pl) : S8_sf64(xa, ps))) / (large ? S6_sf64(xa, ql) : S8_sf64(xa, qs));
    if (!(z != 0.0)) r = 0.0;
    return x < 0.0 ? 2.0 - r : r;
}
static double erf_f64(double x) { return fabs(x) > 1.0 ? 1.0 - erfc_core_f64(x) : erf_core_f64(x); }
static double erfc_f64(double x) { return fabs(x) < 1.0 ? 1.0 - erf_core_f64(x) : erfc_core_f64(x); }

/* ------------------------------------------------------------------------------------ */
/*  generic dispatch helpers                                                              */
/* ------------------------------------------------------------------------------------ */

static int is(const char *a, const char *b) { return strcmp(a, b) == 0; }

static inline uint32_t lzcnt32(uint32_t v) { return v ? (uint32_t) __builtin_clz(v) : 32u; }
static inline uint32_t tzcnt32(uint32_t v) { return v ? (uint32_t) __builtin_ctz(v) : 32u; }
static inline uint64_t lzcnt64(uint64_t v) { return v ? (uint64_t) __builtin_clzll(v) : 64u; }
static inline uint64_t tzcnt64(uint64_t v) { return v ? (uint64_t) __builtin_ctzll(v) : 64u; }

/* ------------------------------------------------------------------------------------ */
/*  unary                                                                                 */
/* ------------------------------------------------------------------------------------ */

static int unary_f32(const char *op, const float *a, float *o, size_t n) {
#define LOOP(expr) do { for (size_t i = 0; i < n; ++i) { float x = a[i]; (void) x; o[i] = (expr); } return 0; } while (0)
    if (is(op, "neg"))   LOOP(u2f(f2u(x) ^ 0x80000000u));          /* array_avx.h neg_: xor sign bit */
    if (is(op, "abs"))   LOOP(u2f(f2u(x) & 0x7fffffffu));
    if (is(op, "sqrt"))  LOOP(sqrtf(x));
    if (is(op, "rcp"))   LOOP(1.0f / x);    /* class C: reference uses rcpps + 1 NR step (array_avx.h:324-357) */
    if (is(op, "rsqrt")) LOOP(1.0f / sqrtf(x)); /* class C (array_avx.h:359-395) */
    if (is(op, "floor")) LOOP(floorf(x));
    if (is(op, "ceil"))  LOOP(ceilf(x));
    if (is(op, "round")) LOOP(rintf(x));    /* nearest-even, array_router.h:254 */
    if (is(op, "trunc")) LOOP(truncf(x));
    if (is(op, "exp"))   LOOP(exp_f32(x));
    if (is(op, "log"))   LOOP(log_f32(x));
    if (is(op, "sign"))  LOOP(u2f((f2u(x) & 0x80000000u) | f2u(1.0f))); /* array_router.h:371 */
    if (is(op, "tan"))   LOOP(tancot_f32(x, 1));
    if (is(op, "cot"))   LOOP(tancot_f32(x, 0));
    if (is(op, "asin"))  LOOP(asin_f32(x));
    if (is(op, "acos"))  LOOP(acos_f32(x));
    if (is(op, "atan"))  LOOP(atan2_f32(x, 1.0f));                  /* array_math.h:666-668 */
    if (is(op, "sinh"))  LOOP(sinh_f32(x));
    if (is(op, "cosh"))  LOOP(cosh_f32(x));
    if (is(op, "tanh"))  LOOP(tanh_f32(x));
    if (is(op, "asinh")) LOOP(asinh_f32(x));
    if (is(op, "acosh")) LOOP(acosh_f32(x));
    if (is(op, "atanh")) LOOP(atanh_f32(x));
    if (is(op, "cbrt"))  LOOP(cbrt_f32(x));
    if (is(op, "erf"))    LOOP(erf_f32(x));
    if (is(op, "erfc"))   LOOP(erfc_f32(x));
    if (is(op, "erfinv")) LOOP(erfinv_sf32(x));
    if (is(op, "i0e"))    LOOP(i0e_sf32(x));
    if (is(op, "dawson")) LOOP(dawson_sf32(x));
    if (is(op, "erfi"))   LOOP(erfi_sf32(x));
    if (is(op, "lgamma")) LOOP(lgamma_sf32(x));
    if (is(op, "tgamma")) LOOP(tgamma_sf32(x));
#undef LOOP
    if (is(op, "sin")) { for (size_t i = 0; i < n; ++i) sincos_f32(a[i], &o[i], NULL); return 0; }
    if (is(op, "cos")) { for (size_t i = 0; i < n; ++i) sincos_f32(a[i], NULL, &o[i]); return 0; }
    return -1;
}

static int unary_f64(const char *op, const double *a, double *o, size_t n) {
#define LOOP(expr) do { for (size_t i = 0; i < n; ++i) { double x = a[i]; o[i] = (expr); } return 0; } while (0)
    if (is(op, "neg"))   LOOP(u2d(d2u(x) ^ 0x8000000000000000ull));
    if (is(op, "abs"))   LOOP(u2d(d2u(x) & 0x7fffffffffffffffull));
    if (is(op, "sqrt"))  LOOP(sqrt(x));
    if (is(op, "floor")) LOOP(floor(x));
    if (is(op, "ceil"))  LOOP(ceil(x));
    if (is(op, "round")) LOOP(rint(x));
    if (is(op, "trunc")) LOOP(trunc(x));
    if (is(op, "exp"))   LOOP(exp_f64(x));
    if (is(op, "log"))   LOOP(log_f64(x));
    if (is(op, "tan"))   LOOP(tancot_f64(x, 1));
    if (is(op, "cot"))   LOOP(tancot_f64(x, 0));
    if (is(op, "asin"))  LOOP(asin_f64(x));
    if (is(op, "acos"))  LOOP(acos_f64(x));
    if (is(op, "atan"))  LOOP(atan2_f64(x, 1.0));
    if (is(op, "sinh"))  LOOP(sinh_f64(x));
    if (is(op, "cosh"))  LOOP(cosh_f64(x));
    if (is(op, "tanh"))  LOOP(tanh_f64(x));
    if (is(op, "asinh")) LOOP(asinh_f64(x));
    if (is(op, "acosh")) LOOP(acosh_f64(x));
    if (is(op, "atanh")) LOOP(atanh_f64(x));
    if (is(op, "cbrt"))  LOOP(cbrt_f64(x));
    if (is(op, "erf"))    LOOP(erf_f64(x));
    if (is(op, "erfc"))   LOOP(erfc_f64(x));
    if (is(op, "erfinv")) LOOP(erfinv_sf64(x));
    if (is(op, "i0e"))    LOOP(i0e_sf64(x));
    if (is(op, "dawson")) LOOP(dawson_sf64(x));
    if (is(op, "erfi"))   LOOP(erfi_sf64(x));
    if (is(op, "lgamma")) LOOP(lgamma_sf64(x));
    if (is(op, "tgamma")) LOOP(tgamma_sf64(x));
#undef LOOP
    if (is(op, "sin")) { for (size_t i = 0; i < n; ++i) sincos_f64(a[i], &o[i], NULL); return 0; }
    if (is(op, "cos")) { for (size_t i = 0; i < n; ++i) sincos_f64(a[i], NULL, &o[i]); return 0; }
    return -1;
}

#define DEF_UNARY_INT(NAME, T, UT, BITS, LZ, TZ, POP)                                         \
    static int NAME(const char *op, const T *a, T *o, size_t n) {                             \
        for (size_t i = 0; i < n; ++i) {                                                      \
            UT x = (UT) a[i];                                                                 \
            if      (is(op, "neg"))    o[i] = (T) (0 - x);                                    \
            else if (is(op, "not"))    o[i] = (T) ~x;                                         \
            else if (is(op, "abs"))    o[i] = (T) (((T) x < 0) ? (0 - x) : x);                \
            else if (is(op, "popcnt")) o[i] = (T) POP(x);                                     \
            else if (is(op, "lzcnt"))  o[i] = (T) LZ(x);                                      \
            else if (is(op, "tzcnt"))  o[i] = (T) TZ(x);                                      \
            else return -1;                                                                   \
        }                                                                                     \
        return 0;                                                                             \
    }
DEF_UNARY_INT(unary_i32, int32_t, uint32_t, 32, lzcnt32, tzcnt32, __builtin_popcount)
DEF_UNARY_INT(unary_u32, uint32_t, uint32_t, 32, lzcnt32, tzcnt32, __builtin_popcount)
DEF_UNARY_INT(unary_i64, int64_t, uint64_t, 64, lzcnt64, tzcnt64, __builtin_popcountll)
DEF_UNARY_INT(unary_u64, uint64_t, uint64_t, 64, lzcnt64, tzcnt64, __builtin_popcountll)

int orc_unary(int type, const char *op, const void *a, void *out, size_t n) {
    switch (type) {
        case T_I32: return unary_i32(op, (const int32_t *) a, (int32_t *) out, n);
        case T_U32: return unary_u32(op, (const uint32_t *) a, (uint32_t *) out, n);
        case T_I64: return unary_i64(op, (const int64_t *) a, (int64_t *) out, n);
        case T_U64: return unary_u64(op, (const uint64_t *) a, (uint64_t *) out, n);
        case T_F32: return unary_f32(op, (const float *) a, (float *) out, n);
        case T_F64: return unary_f64(op, (const double *) a, (double *) out, n);
    }
    return -2;
}

int orc_sincos(int type, const void *a_, void *s_, void *c_, size_t n) {
    if (type == T_F64) {
        const double *ad = (const double *) a_;
        double *sd = (double *) s_, *cd = (double *) c_;
        for (size_t i = 0; i < n; ++i) sincos_f64(ad[i], &sd[i], &cd[i]);
        return 0;
    }
    if (type != T_F32) return -2;
    const float *a = (const float *) a_;
    float *s = (float *) s_, *c = (float *) c_;
    for (size_t i = 0; i < n; ++i) sincos_f32(a[i], &s[i], &c[i]);
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/*  binary / ternary                                                                      */
/* ------------------------------------------------------------------------------------ */

/* safe_mul / safe_fmadd: CPU branch of src/autodiff/autodiff.cpp:1191-1221 */
static inline float safe_mul_f32(float w, float g) { return (w == 0.0f || g == 0.0f) ? 0.0f : w * g; }
static inline float safe_fmadd_f32(float w, float g, float acc) {
    return (w == 0.0f || g == 0.0f) ? acc : fmaf(w, g, acc);
}

static int binary_f32(const char *op, const float *a, const float *b, float *o, size_t n) {
#define LOOP(expr) do { for (size_t i = 0; i < n; ++i) { float x = a[i], y = b[i]; o[i] = (expr); } return 0; } while (0)
    if (is(op, "add")) LOOP(x + y);
    if (is(op, "sub")) LOOP(x - y);
    if (is(op, "mul")) LOOP(x * y);
    if (is(op, "div")) LOOP(x / y);
    if (is(op, "min")) LOOP(min_ps(x, y));
    if (is(op, "max")) LOOP(max_ps(x, y));
    if (is(op, "safe_mul")) LOOP(safe_mul_f32(x, y));
    if (is(op, "atan2")) LOOP(atan2_f32(x, y));
    if (is(op, "pow"))   LOOP(exp_f32(log_f32(x) * y));             /* array_math.h:956-958 */
    if (is(op, "fmod"))  LOOP(fmaf(-truncf(x / y), y, x));          /* array_math.h:1381-1383 */
    if (is(op, "ldexp")) LOOP(ldexp_f32(x, y));                     /* array_math.h:677-680 */
#undef LOOP
    return -1;
}

static int binary_f64(const char *op, const double *a, const double *b, double *o, size_t n) {
#define LOOP(expr) do { for (size_t i = 0; i < n; ++i) { double x = a[i], y = b[i]; o[i] = (expr); } return 0; } while (0)
    if (is(op, "add")) LOOP(x + y);
    if (is(op, "sub")) LOOP(x - y);
    if (is(op, "mul")) LOOP(x * y);
    if (is(op, "div")) LOOP(x / y);
    if (is(op, "min")) LOOP(y < x ? y : x);
    if (is(op, "max")) LOOP(y > x ? y : x);
    if (is(op, "atan2")) LOOP(atan2_f64(x, y));
    if (is(op, "pow"))   LOOP(exp_f64(log_f64(x) * y));
    if (is(op, "fmod"))  LOOP(fma(-trunc(x / y), y, x));
    if (is(op, "ldexp")) LOOP(ldexp_f64(x, y));
    if (is(op, "safe_mul")) LOOP((x == 0.0 || y == 0.0) ? 0.0 : x * y);
#undef LOOP
    return -1;
}

/* Integer semantics of the AVX2 packets:
   - add/sub/mul wrap (two's complement);  div/mod are C truncating division (scalar loops in the reference);
   - variable shifts (vpsllvd/vpsrlvd/vpsravd, array_avx2.h sl_/sr_): count >= BITS gives 0 (or sign fill for
     arithmetic right shift) rather than wrapping the count;
   - mulhi: high half of the full product (array_fallbacks.h mulhi_scalar). */
#define DEF_BINARY_INT(NAME, T, UT, WT, BITS, SIGNED)                                          \
    static int NAME(const char *op, const T *a, const T *b, T *o, size_t n) {                  \
        for (size_t i = 0; i < n; ++i) {                                                       \
            T x = a[i], y = b[i];                                                              \
            UT ux = (UT) x, uy = (UT) y;                                                       \
            if      (is(op, "add")) o[i] = (T) (ux + uy);                                      \
            else if (is(op, "sub")) o[i] = (T) (ux - uy);                                      \
            else if (is(op, "mul")) o[i] = (T) (ux * uy);                                      \
            else if (is(op, "div")) o[i] = (T) (x / y);                                        \
            else if (is(op, "mod")) o[i] = (T) (x % y);                                        \
            else if (is(op, "min")) o[i] = x < y ? x : y;                                      \
            else if (is(op, "max")) o[i] = x > y ? x : y;                                      \
            else if (is(op, "and")) o[i] = (T) (ux & uy);                                      \
            else if (is(op, "or"))  o[i] = (T) (ux | uy);                                      \
            else if (is(op, "xor")) o[i] = (T) (ux ^ uy);                                      \
            else if (is(op, "mulhi")) o[i] = (T) (((WT) x * (WT) y) >> BITS);                  \
            else if (is(op, "sl"))  o[i] = (T) (uy >= BITS ? 0 : (UT) (ux << uy));             \
            else if (is(op, "sr")) {                                                           \
                if (SIGNED) o[i] = (T) (uy >= BITS ? (x < 0 ? (T) -1 : (T) 0) : (T) (x >> uy)); \
                else        o[i] = (T) (uy >= BITS ? 0 : (UT) (ux >> uy));                     \
            }                                                                                  \
            else return -1;                                                                    \
        }                                                                                      \
        return 0;                                                                              \
    }
DEF_BINARY_INT(binary_i32, int32_t, uint32_t, int64_t, 32, 1)
DEF_BINARY_INT(binary_u32, uint32_t, uint32_t, uint64_t, 32, 0)
DEF_BINARY_INT(binary_i64, int64_t, uint64_t, __int128, 64, 1)
DEF_BINARY_INT(binary_u64, uint64_t, uint64_t, unsigned __int128, 64, 0)

int orc_binary(int type, const char *op, const void *a, const void *b, void *out, size_t n) {
    switch (type) {
        case T_I32: return binary_i32(op, (const int32_t *) a, (const int32_t *) b, (int32_t *) out, n);
        case T_U32: return binary_u32(op, (const uint32_t *) a, (const uint32_t *) b, (uint32_t *) out, n);
        case T_I64: return binary_i64(op, (const int64_t *) a, (const int64_t *) b, (int64_t *) out, n);
        case T_U64: return binary_u64(op, (const uint64_t *) a, (const uint64_t *) b, (uint64_t *) out, n);
        case T_F32: return binary_f32(op, (const float *) a, (const float *) b, (float *) out, n);
        case T_F64: return binary_f64(op, (const double *) a, (const double *) b, (double *) out, n);
    }
    return -2;
}

int orc_ternary(int type, const char *op, const void *a_, const void *b_, const void *c_, void *out, size_t n) {
    if (type == T_F32) {
        const float *a = (const float *) a_, *b = (const float *) b_, *c = (const float *) c_;
        float *o = (float *) out;
        for (size_t i = 0; i < n; ++i) {
            /* array_avx.h fmadd_/fmsub_/fnmadd_/fnmsub_: vfmadd/vfmsub/vfnmadd/vfnmsub, single rounding */
            if      (is(op, "fmadd"))  o[i] = fmaf(a[i], b[i], c[i]);
            else if (is(op, "fmsub"))  o[i] = fmaf(a[i], b[i], -c[i]);
            else if (is(op, "fnmadd")) o[i] = fmaf(-a[i], b[i], c[i]);
            else if (is(op, "fnmsub")) o[i] = fmaf(-a[i], b[i], -c[i]);
            else if (is(op, "safe_fmadd")) o[i] = safe_fmadd_f32(a[i], b[i], c[i]);
            else return -1;
        }
        return 0;
    } else if (type == T_F64) {
        const double *a = (const double *) a_, *b = (const double *) b_, *c = (const double *) c_;
        double *o = (double *) out;
        for (size_t i = 0; i < n; ++i) {
            if      (is(op, "fmadd"))  o[i] = fma(a[i], b[i], c[i]);
            else if (is(op, "fmsub"))  o[i] = fma(a[i], b[i], -c[i]);
            else if (is(op, "fnmadd")) o[i] = fma(-a[i], b[i], c[i]);
            else if (is(op, "fnmsub")) o[i] = fma(-a[i], b[i], -c[i]);
            else return -1;
        }
        return 0;
    } else if (type == T_I32 || type == T_U32) {
        /* integer fmadd = mul.lo + add (array_generic fallbacks) */
        const uint32_t *a = (const uint32_t *) a_, *b = (const uint32_t *) b_, *c = (const uint32_t *) c_;
        uint32_t *o = (uint32_t *) out;
        for (size_t i = 0; i < n; ++i) {
            if      (is(op, "fmadd"))  o[i] = a[i] * b[i] + c[i];
            else if (is(op, "fmsub"))  o[i] = a[i] * b[i] - c[i];
            else if (is(op, "fnmadd")) o[i] = c[i] - a[i] * b[i];
            else if (is(op, "fnmsub")) o[i] = 0u - a[i] * b[i] - c[i];
            else return -1;
        }
        return 0;
    }
    return -2;
}

/* ------------------------------------------------------------------------------------ */
/*  compare / select / cast                                                               */
/* ------------------------------------------------------------------------------------ */

#define DEF_COMPARE(NAME, T)                                                                  \
    static int NAME(const char *op, const T *a, const T *b, uint8_t *o, size_t n) {           \
        for (size_t i = 0; i < n; ++i) {                                                      \
            T x = a[i], y = b[i];                                                             \
            if      (is(op, "eq"))  o[i] = x == y;                                            \
            else if (is(op, "neq")) o[i] = x != y;   /* unordered-true, _CMP_NEQ_UQ */        \
            else if (is(op, "lt"))  o[i] = x < y;                                             \
            else if (is(op, "le"))  o[i] = x <= y;                                            \
            else if (is(op, "gt"))  o[i] = x > y;                                             \
            else if (is(op, "ge"))  o[i] = x >= y;                                            \
            else return -1;                                                                   \
        }                                                                                     \
        return 0;                                                                             \
    }
DEF_COMPARE(compare_i32, int32_t) DEF_COMPARE(compare_u32, uint32_t)
DEF_COMPARE(compare_i64, int64_t) DEF_COMPARE(compare_u64, uint64_t)
DEF_COMPARE(compare_f32, float)   DEF_COMPARE(compare_f64, double)

int orc_compare(int type, const char *op, const void *a, const void *b, uint8_t *out, size_t n) {
    switch (type) {
        case T_I32: return compare_i32(op, (const int32_t *) a, (const int32_t *) b, out, n);
        case T_U32: return compare_u32(op, (const uint32_t *) a, (const uint32_t *) b, out, n);
        case T_I64: return compare_i64(op, (const int64_t *) a, (const int64_t *) b, out, n);
        case T_U64: return compare_u64(op, (const uint64_t *) a, (const uint64_t *) b, out, n);
        case T_F32: return compare_f32(op, (const float *) a, (const float *) b, out, n);
        case T_F64: return compare_f64(op, (const double *) a, (const double *) b, out, n);
    }
    return -2;
}

static size_t type_size(int type) {
    switch (type) {
        case T_BOOL: return 1;
        case T_I32: case T_U32: case T_F32: return 4;
        case T_I64: case T_U64: case T_F64: return 8;
    }
    return 0;
}

/* select_: blendv on raw bits (array_avx.h select_) */
int orc_select(int type, const uint8_t *m, const void *t, const void *f, void *out, size_t n) {
    size_t sz = type_size(type);
    if (!sz) return -2;
    for (size_t i = 0; i < n; ++i)
        memcpy((char *) out + i * sz, (const char *) (m[i] ? t : f) + i * sz, sz);
    return 0;
}

/* Conversions between array types (array_avx.h / array_avx2.h converting constructors; scalar
   loops for the 64-bit integer cases).  f32->i32 is cvttps2dq (indefinite = INT_MIN);
   everything else matches C casts on in-range inputs, which is all the tests feed. */
int orc_cast(int src, int dst, const void *a, void *out, size_t n) {
#define CAST_LOOP(ST, DT, EXPR) do { const ST *s = (const ST *) a; DT *d = (DT *) out;       \
        for (size_t i = 0; i < n; ++i) { ST x = s[i]; d[i] = (DT) (EXPR); } return 0; } while (0)
#define ROW(SC, ST)                                                                           \
    if (src == SC) {                                                                          \
        switch (dst) {                                                                        \
            case T_I32: if (SC == T_F32) CAST_LOOP(ST, int32_t, cvtt_f32_i32((float) x));    \
                        CAST_LOOP(ST, int32_t, x);                                            \
            case T_U32: CAST_LOOP(ST, uint32_t, x);                                           \
            case T_I64: CAST_LOOP(ST, int64_t, x);                                            \
            case T_U64: CAST_LOOP(ST, uint64_t, x);                                           \
            case T_F32: CAST_LOOP(ST, float, x);                                              \
            case T_F64: CAST_LOOP(ST, double, x);                                             \
        }                                                                                     \
    }
    ROW(T_I32, int32_t) ROW(T_U32, uint32_t) ROW(T_I64, int64_t)
    ROW(T_U64, uint64_t) ROW(T_F32, float) ROW(T_F64, double)
#undef ROW
#undef CAST_LOOP
    return -2;
}

/* ------------------------------------------------------------------------------------ */
/*  gather / scatter / scatter_add  (include/enoki/dynamic.h:478-534)                     */
/* ------------------------------------------------------------------------------------ */

static int64_t load_index(int itype, const void *idx, size_t i) {
    switch (itype) {
        case T_I32: return ((const int32_t *) idx)[i];
        case T_U32: return ((const uint32_t *) idx)[i];
        case T_I64: return ((const int64_t *) idx)[i];
        case T_U64: return (int64_t) ((const uint64_t *) idx)[i];
    }
    return 0;
}

/* out[i] = mask[i] ? base[idx[i]] : 0   (array_router.h:1075-1079 scalar case) */
int orc_gather(int type, int itype, const void *base, size_t src_size, const void *idx,
               const uint8_t *mask, void *out, size_t n) {
    size_t sz = type_size(type);
    (void) src_size;
    if (!sz) return -2;
    for (size_t i = 0; i < n; ++i) {
        if (mask[i]) memcpy((char *) out + i * sz, (const char *) base + load_index(itype, idx, i) * (int64_t) sz, sz);
        else         memset((char *) out + i * sz, 0, sz);
    }
    return 0;
}

/* scatter: element order, last writer wins (dynamic.h:498-515 -> per-packet scatter in lane order).
   scatter_add: sequential read-modify-write in element order (dynamic.h:517-534 ->
   transform, array_static.h:982-991). */
int orc_scatter(int type, int itype, int add, void *base, const void *val, const void *idx,
                const uint8_t *mask, size_t n) {
    size_t sz = type_size(type);
    if (!sz) return -2;
    for (size_t i = 0; i < n; ++i) {
        if (!mask[i]) continue;
        char *dst = (char *) base + load_index(itype, idx, i) * (int64_t) sz;
        const char *src = (const char *) val + i * sz;
        if (!add) { memcpy(dst, src, sz); continue; }
        switch (type) {
            case T_F32: { float d, s; memcpy(&d, dst, 4); memcpy(&s, src, 4); d += s; memcpy(dst, &d, 4); break; }
            case T_F64: { double d, s; memcpy(&d, dst, 8); memcpy(&s, src, 8); d += s; memcpy(dst, &d, 8); break; }
            case T_I32: case T_U32: { uint32_t d, s; memcpy(&d, dst, 4); memcpy(&s, src, 4); d += s; memcpy(dst, &d, 4); break; }
            case T_I64: case T_U64: { uint64_t d, s; memcpy(&d, dst, 8); memcpy(&s, src, 8); d += s; memcpy(dst, &d, 8); break; }
            default: return -2;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/*  horizontal reductions (include/enoki/dynamic.h:632-752)                               */
/* ------------------------------------------------------------------------------------ */

/* DynamicArray::hsum_: lane-wise accumulation over full packets, masked add of the last
   (partial) packet, then the AVX in-packet tree: (lo + hi) [array_avx.h:404] followed by the
   SSE tree of the 4-lane half (array_sse42.h hsum_: (v + movehdup(v)), then + movehl). */
static float hreduce_f32(const float *a, size_t n, int op /* 0 sum, 1 prod, 2 min, 3 max */) {
    if (n == 0) {
        /* dynamic.h:633, 651, 669, 687 -- note hmax of an empty array is numeric_limits::min() */
        switch (op) { case 0: return 0.0f; case 1: return 1.0f; case 2: return 3.402823466e+38f; default: return 1.175494351e-38f; }
    }
    if (n == 1) return a[0];
    float lane[PKT];
    for (int l = 0; l < PKT; ++l) lane[l] = (op == 0) ? 0.0f : (op == 1 ? 1.0f : a[0]);
    size_t packets = (n + PKT - 1) / PKT;
    for (size_t p = 0; p + 1 < packets; ++p) {
        for (int l = 0; l < PKT; ++l) {
            float v = a[p * PKT + l];
            switch (op) {
                case 0: lane[l] += v; break;
                case 1: lane[l] *= v; break;
                case 2: lane[l] = min_ps(lane[l], v); break;
                default: lane[l] = max_ps(lane[l], v); break;
            }
        }
    }
    size_t last = (n - 1) % PKT;
    for (size_t l = 0; l <= last; ++l) {
        float v = a[(packets - 1) * PKT + l];
        switch (op) {
            case 0: lane[l] += v; break;
            case 1: lane[l] *= v; break;
            case 2: lane[l] = min_ps(lane[l], v); break;
            default: lane[l] = max_ps(lane[l], v); break;
        }
    }
    float h[4];
    for (int l = 0; l < 4; ++l) {
        switch (op) {
            case 0: h[l] = lane[l] + lane[l + 4]; break;
            case 1: h[l] = lane[l] * lane[l + 4]; break;
            case 2: h[l] = min_ps(lane[l], lane[l + 4]); break;
            default: h[l] = max_ps(lane[l], lane[l + 4]); break;
        }
    }
    /* SSE tree: t = h + movehdup(h)  -> t0 = h0 (+) h1, t2 = h2 (+) h3;  r = t0 (+) t2 */
    switch (op) {
        case 0: return (h[0] + h[1]) + (h[2] + h[3]);
        case 1: return (h[0] * h[1]) * (h[2] * h[3]);
        case 2: return min_ps(min_ps(h[0], h[1]), min_ps(h[2], h[3]));
        default: return max_ps(max_ps(h[0], h[1]), max_ps(h[2], h[3]));
    }
}

#define DEF_REDUCE_INT(NAME, T, UT, TMAX, TMIN)                                               \
    static T NAME(const T *a, size_t n, int op) {                                             \
        if (n == 0) { switch (op) { case 0: return 0; case 1: return 1; case 2: return TMAX; default: return TMIN; } } \
        T r = (op == 0) ? 0 : (op == 1 ? 1 : a[0]);                                           \
        for (size_t i = 0; i < n; ++i) {                                                      \
            switch (op) {                                                                     \
                case 0: r = (T) ((UT) r + (UT) a[i]); break;                                  \
                case 1: r = (T) ((UT) r * (UT) a[i]); break;                                  \
                case 2: r = a[i] < r ? a[i] : r; break;                                       \
                default: r = a[i] > r ? a[i] : r; break;                                      \
            }                                                                                 \
        }                                                                                     \
        return r;                                                                             \
    }
DEF_REDUCE_INT(hreduce_i32, int32_t, uint32_t, INT32_MAX, INT32_MIN)
DEF_REDUCE_INT(hreduce_u32, uint32_t, uint32_t, UINT32_MAX, 0)
DEF_REDUCE_INT(hreduce_i64, int64_t, uint64_t, INT64_MAX, INT64_MIN)
DEF_REDUCE_INT(hreduce_u64, uint64_t, uint64_t, UINT64_MAX, 0)

int orc_reduce(int type, const char *op_, const void *a, void *out, size_t n) {
    int op = is(op_, "hsum") ? 0 : is(op_, "hprod") ? 1 : is(op_, "hmin") ? 2 : is(op_, "hmax") ? 3 : -1;
    if (op < 0) return -1;
    switch (type) {
        case T_F32: *(float *) out = hreduce_f32((const float *) a, n, op); return 0;
        case T_I32: *(int32_t *) out = hreduce_i32((const int32_t *) a, n, op); return 0;
        case T_U32: *(uint32_t *) out = hreduce_u32((const uint32_t *) a, n, op); return 0;
        case T_I64: *(int64_t *) out = hreduce_i64((const int64_t *) a, n, op); return 0;
        case T_U64: *(uint64_t *) out = hreduce_u64((const uint64_t *) a, n, op); return 0;
    }
    return -2;
}

/* all_/any_/count_ (dynamic.h:704-752): empty -> any false, all true, count 0 */
int orc_mask_reduce(const char *op, const uint8_t *m, uint64_t *out, size_t n) {
    uint64_t cnt = 0;
    for (size_t i = 0; i < n; ++i) cnt += m[i] ? 1 : 0;
    if      (is(op, "all"))   *out = cnt == n;
    else if (is(op, "any"))   *out = cnt != 0;
    else if (is(op, "count")) *out = cnt;
    else return -1;
    return 0;
}

/* arange / linspace (dynamic.h:909-938).  The CPU DynamicArray builds the first packet with the static
   linspace (array_static.h:1119-1141: lane * step' + min with step' = (hi - min) / 7, mul and add rounded
   separately) and then advances packet by packet with `value_p += shift` (shift = step * 8), i.e. the error
   accumulates with the packet index.  (The reference's GPU type uses fmadd(index, step, min) instead,
   cuda.h:655-663 -- that closed form is what the HIP kernel implements; see tests for the tolerance.) */
int orc_arange_f32(float *out, size_t n) {
    float v[PKT];
    for (int l = 0; l < PKT; ++l) v[l] = (float) l;
    for (size_t p = 0; p * PKT < n; ++p)
        for (int l = 0; l < PKT; ++l) { if (p * PKT + l < n) out[p * PKT + l] = v[l]; v[l] += (float) PKT; }
    return 0;
}
int orc_arange_u32(uint32_t *out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = (uint32_t) i; return 0; }
int orc_linspace_f32(float lo, float hi, float *out, size_t n) {
    float step = (hi - lo) / (float) (n - 1);
    float last = lo + step * (float) (PKT - 1);
    float pstep = (last - lo) / (float) (PKT - 1);
    float shift = step * (float) PKT;
    float v[PKT];
    for (int l = 0; l < PKT; ++l) { float t = (float) l * pstep; v[l] = t + lo; }
    for (size_t p = 0; p * PKT < n; ++p)
        for (int l = 0; l < PKT; ++l) { if (p * PKT + l < n) out[p * PKT + l] = v[l]; v[l] += shift; }
    return 0;
}
int orc_reverse_f32(const float *a, float *out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = a[n - 1 - i]; return 0; }
int orc_psum_f32(const float *a, float *out, size_t n) {
    if (n) out[0] = a[0];
    for (size_t i = 1; i < n; ++i) out[i] = out[i - 1] + a[i];
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/*  BASELINE.json configs                                                                 */
/* ------------------------------------------------------------------------------------ */

static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

/* cfg1: hsum(fmadd(a, x, b)), tests/dynamic.cpp-style packet loops */
float orc_cfg1(const float *a, const float *x, const float *b, size_t n, double *seconds) {
    float *u = (float *) malloc(n * sizeof(float));
    double t0 = now_s();
    for (size_t i = 0; i < n; ++i) u[i] = fmaf(a[i], x[i], b[i]);
    float y = hreduce_f32(u, n, 0);
    if (seconds) *seconds = now_s() - t0;
    free(u);
    return y;
}

/* cfg2: hsum(sin(exp(fmadd(a, x, b)))) -- one array pass per op like the reference */
float orc_cfg2(const float *a, const float *x, const float *b, size_t n, double *seconds) {
    float *u = (float *) malloc(n * sizeof(float));
    double t0 = now_s();
    for (size_t i = 0; i < n; ++i) u[i] = fmaf(a[i], x[i], b[i]);
    for (size_t i = 0; i < n; ++i) u[i] = exp_f32(u[i]);
    for (size_t i = 0; i < n; ++i) sincos_f32(u[i], &u[i], NULL);
    float y = hreduce_f32(u, n, 0);
    if (seconds) *seconds = now_s() - t0;
    free(u);
    return y;
}

/* cfg3a: y = hsum(sin(fmadd(a, x, b))); backward(y) with a, b leaves (size n) and x plain.
 *
 * Restates what the reference tape does for exactly this graph:
 *   forward  DiffArray::fmadd_ (autodiff.h:274-286): u = fmadd(a,x,b), edges a<-w=x, b<-w=1
 *            DiffArray::sin_   (autodiff.h:491-501): (s,c) = sincos(u), edge u<-w=c
 *            DiffArray::hsum_  (autodiff.h:1052-1062): y = hsum(s), edge s<-w=1
 *   backward Tape::backward (autodiff.cpp:838-910), nodes in descending index order:
 *            g_s = safe_mul(1, 1)               (:873-874; scalar, broadcast by set_slices :851-861)
 *            g_u = safe_mul(c, g_s)
 *            g_a = safe_mul(x, g_u);  g_b = safe_mul(1, g_u)
 */
float orc_cfg3a(const float *a, const float *x, const float *b, size_t n,
                float *grad_a, float *grad_b, double *seconds) {
    float *s = (float *) malloc(n * sizeof(float)), *c = (float *) malloc(n * sizeof(float));
    float *gu = (float *) malloc(n * sizeof(float));
    double t0 = now_s();
    for (size_t i = 0; i < n; ++i) s[i] = fmaf(a[i], x[i], b[i]);
    for (size_t i = 0; i < n; ++i) sincos_f32(s[i], &s[i], &c[i]);
    float y = hreduce_f32(s, n, 0);
    float gs = safe_mul_f32(1.0f, 1.0f);
    for (size_t i = 0; i < n; ++i) gu[i] = safe_mul_f32(c[i], gs);
    for (size_t i = 0; i < n; ++i) grad_a[i] = safe_mul_f32(x[i], gu[i]);
    for (size_t i = 0; i < n; ++i) grad_b[i] = safe_mul_f32(1.0f, gu[i]);
    if (seconds) *seconds = now_s() - t0;
    free(s); free(c); free(gu);
    return y;
}

/* cfg3b: a = gather(A, idx), b = gather(B, idx) (A, B leaves of size k), rest as cfg3a;
 *   backward additionally runs Gather::backward (autodiff.cpp:384-398): grad_A = zero(k);
 *   scatter_add(grad_A, g_a, idx) -- sequential in element order on the CPU path. */
float orc_cfg3b(const float *A, const float *B, size_t k, const float *x, const uint32_t *idx,
                size_t n, float *grad_A, float *grad_B, double *seconds) {
    float *s = (float *) malloc(n * sizeof(float)), *c = (float *) malloc(n * sizeof(float));
    float *gu = (float *) malloc(n * sizeof(float)), *ga = (float *) malloc(n * sizeof(float));
    float *av = (float *) malloc(n * sizeof(float)), *bv = (float *) malloc(n * sizeof(float));
    double t0 = now_s();
    for (size_t i = 0; i < n; ++i) av[i] = A[idx[i]];
    for (size_t i = 0; i < n; ++i) bv[i] = B[idx[i]];
    for (size_t i = 0; i < n; ++i) s[i] = fmaf(av[i], x[i], bv[i]);
    for (size_t i = 0; i < n; ++i) sincos_f32(s[i], &s[i], &c[i]);
    float y = hreduce_f32(s, n, 0);
    float gs = safe_mul_f32(1.0f, 1.0f);
    for (size_t i = 0; i < n; ++i) gu[i] = safe_mul_f32(c[i], gs);
    /* node order: fmadd's edges are (a <- x), (b <- 1); both gather nodes have a higher index than the
       leaves, and b's gather node (created second) is processed first in descending order. */
    for (size_t i = 0; i < n; ++i) ga[i] = safe_mul_f32(x[i], gu[i]);
    for (size_t i = 0; i < n; ++i) gu[i] = safe_mul_f32(1.0f, gu[i]);
    memset(grad_B, 0, k * sizeof(float));
    for (size_t i = 0; i < n; ++i) grad_B[idx[i]] += gu[i];
    memset(grad_A, 0, k * sizeof(float));
    for (size_t i = 0; i < n; ++i) grad_A[idx[i]] += ga[i];
    if (seconds) *seconds = now_s() - t0;
    free(s); free(c); free(gu); free(ga); free(av); free(bv);
    return y;
}

/* cfg4: masked gather -> ray/sphere intersection -> shading -> masked scatter -> hit count.
 * Arithmetic order of the user-level kernels (tests/sphere.cpp:58-83) on Array<Packet, 3>:
 *   dot(u, v) = fmadd(u2, v2, fmadd(u1, v1, u0*v0))                     (array_static.h:948-960)
 *   a = dot(d,d); b = 2*dot(o,d); c = dot(o,o) - 1; discrim = b*b - (4*a)*c   (operators: separate roundings)
 *   t = (-b + sqrt(discrim)) / (2*a); pos = o + t*d; hit = discrim >= 0; pos = select(hit, pos, 0)
 *   shade = 0.2 + max(dot(pos, (-1,-1,2)), 0) * 90 */
static inline float dot3(const float *u, const float *v) { return fmaf(u[2], v[2], fmaf(u[1], v[1], u[0] * v[0])); }

int orc_cfg4(const float *gx, const float *gy, const uint32_t *perm, const uint8_t *mask, size_t n, float *image,
             uint64_t *hit_count) {
    uint64_t hits = 0;
    float *shade = (float *) malloc(n * sizeof(float));
    uint8_t *hit = (uint8_t *) malloc(n);
    for (size_t i = 0; i < n; ++i) {
        float px = mask[i] ? gx[perm[i]] : 0.0f, py = mask[i] ? gy[perm[i]] : 0.0f;
        float o[3] = { px, py, -1.0f }, d[3] = { 0.0f, 0.0f, 1.0f };
        float a = dot3(d, d), b = 2.0f * dot3(o, d), c = dot3(o, o) - 1.0f;
        float t4 = 4.0f * a;
        float discrim = b * b - t4 * c;
        float t = (-b + sqrtf(discrim)) / (2.0f * a);
        int h = discrim >= 0.0f;
        float pos[3], lightdir[3] = { -1.0f, -1.0f, 2.0f };
        for (int k = 0; k < 3; ++k) { float td = t * d[k]; pos[k] = h ? o[k] + td : 0.0f; }
        shade[i] = 0.2f + max_ps(dot3(pos, lightdir), 0.0f) * 90.0f;
        hit[i] = (uint8_t) (h && mask[i]);
    }
    for (size_t i = 0; i < n; ++i)
        if (hit[i]) { image[perm[i]] = shade[i]; hits++; }
    *hit_count = hits;
    free(shade); free(hit);
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/*  PCG32 (include/enoki/random.h:38-330; O'Neill's XSH-RR 64/32), scalar per lane      */
/*  Same script as oracle/ref_driver.cpp:ref_pcg32.                                     */
/* ------------------------------------------------------------------------------------ */
#define ORC_PCG32_MULT 0x5851f42d4c957f2dull

static inline uint32_t pcg32_draw(uint64_t *state, uint64_t inc, int active) {   /* random.h:68-84 */
    uint64_t old = *state;
    if (active) *state = old * ORC_PCG32_MULT + inc;
    uint32_t xorshifted = (uint32_t) (((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t) (old >> 59);
    return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31u));             /* ror */
}

int orc_pcg32(uint64_t initstate, const uint64_t *initseq, size_t n, int steps, const uint8_t *mask,
              uint32_t *out_u32, float *out_f32, uint64_t *out_u64, double *out_f64, uint32_t bound,
              uint32_t *out_bounded, int64_t delta, uint32_t *out_after, uint64_t *state_out) {
    const uint32_t threshold = (~bound + 1u) % bound;                            /* random.h:176 */
    for (size_t i = 0; i < n; ++i) {
        uint64_t state = 0, inc = (initseq[i] << 1) | 1u;                        /* seed(), random.h:62-68 */
        pcg32_draw(&state, inc, 1);
        state += initstate;
        pcg32_draw(&state, inc, 1);

        for (int s = 0; s < steps; ++s)
            out_u32[(size_t) s * n + i] = pcg32_draw(&state, inc, mask[i] != 0);

        out_f32[i] = u2f((pcg32_draw(&state, inc, 1) >> 9) | 0x3f800000u) - 1.0f;                 /* :112-114 */
        /* :87-89 `UInt64(next_uint32()) | sl<32>(UInt64(next_uint32()))`: operand evaluation order is
           unspecified in C++; the pinned g++ build evaluates the RIGHT operand first, so the first draw
           lands in the HIGH word.  The oracle (and the kernel) follow the pinned build. */
        uint64_t hi = pcg32_draw(&state, inc, 1), lo = pcg32_draw(&state, inc, 1);
        out_u64[i] = lo | (hi << 32);
        out_f64[i] = u2d(((uint64_t) pcg32_draw(&state, inc, 1) << 20) | 0x3ff0000000000000ull) - 1.0; /* :128-133 */

        uint32_t r;                                                              /* :178-190, per-lane rejection */
        do { r = pcg32_draw(&state, inc, 1); } while (r < threshold);
        out_bounded[i] = r % bound;

        uint64_t cur_mult = ORC_PCG32_MULT, cur_plus = inc, acc_mult = 1, acc_plus = 0, d = (uint64_t) delta;
        while (d != 0) {                                                         /* advance(), :262-283 */
            if (d & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
            cur_plus = (cur_mult + 1) * cur_plus;
            cur_mult *= cur_mult;
            d >>= 1;
        }
        state = acc_mult * state + acc_plus;

        out_after[i] = pcg32_draw(&state, inc, 1);
        state_out[i] = state;
    }
    return 0;
}
