/*
 * oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin extern "C" driver around the *unmodified reference headers* that live
 * under /root/reference/include (they are #included where they lie, never
 * copied).  Built by oracle/Makefile into oracle/_ref/libenoki_ref.so with the
 * pinned oracle flags:
 *
 *   -O2 -mavx2 -mfma -mf16c -mbmi -mbmi2 -mlzcnt -ffp-contract=off -fno-math-errno
 *
 * which selects the reference's AVX2 `Packet<float,8>` code path with
 * contraction disabled, i.e. only the explicit fmadd() calls fuse.
 *
 * Every entry point takes plain host pointers, runs the op through
 * enoki::DynamicArray<Packet<T,8>> (include/enoki/dynamic.h) or through
 * DiffArray<DynamicArray<Packet<float>>> + Tape (include/enoki/autodiff.h,
 * src/autodiff/autodiff.cpp, compiled into the same .so) and writes the
 * result back to a host pointer.  Used to (a) pin oracle/enoki_oracle.c,
 * (b) generate tests/golden/ fixtures, (c) serve as the "reference" CPU
 * baseline in bench.py.
 */
#include <enoki/array.h>
#include <enoki/dynamic.h>
#include <enoki/autodiff.h>
#include <enoki/random.h>
#include <enoki/matrix.h>
#include <enoki/special.h>
#include <enoki/complex.h>
#include <enoki/quaternion.h>
#include <enoki/transform.h>
#include <enoki/sh.h>

#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../examples/path_trace.h"
using namespace enoki;

namespace {

template <typename T> using Pk  = Packet<T, 8>;
template <typename T> using Dyn = DynamicArray<Pk<T>>;

using FloatX  = DynamicArray<Packet<float>>;    // default width: 8 under -mavx2
using UInt32X = DynamicArray<Packet<uint32_t>>;
using FloatD  = DiffArray<FloatX>;
using UInt32D = DiffArray<UInt32X>;

template <typename T> Dyn<T> load(const T *p, size_t n) {
    Dyn<T> r = Dyn<T>::copy(p, n);
    return r;
}

template <typename A> void store(const A &a, scalar_t<A> *out, size_t n) {
    if (a.size() == 1 && n != 1) {
        for (size_t i = 0; i < n; ++i) out[i] = a.coeff(0);
    } else {
        memcpy(out, a.data(), n * sizeof(scalar_t<A>));
    }
}

/* masks cross the boundary as u8 (0/1) */
template <typename T> mask_t<Dyn<T>> load_mask(const uint8_t *m, size_t n) {
    std::vector<uint32_t> tmp(n);
    for (size_t i = 0; i < n; ++i) tmp[i] = m[i] ? 1u : 0u;
    Dyn<uint32_t> v = Dyn<uint32_t>::copy(tmp.data(), n);
    return mask_t<Dyn<T>>(neq(v, 0u));
}

template <typename M> void store_mask(const M &m, uint8_t *out, size_t n) {
    Dyn<uint32_t> v = select(mask_t<Dyn<uint32_t>>(m), Dyn<uint32_t>(1u), Dyn<uint32_t>(0u));
    if (v.size() == 1 && n != 1) {
        for (size_t i = 0; i < n; ++i) out[i] = (uint8_t) v.coeff(0);
    } else {
        for (size_t i = 0; i < n; ++i) out[i] = (uint8_t) v.coeff(i);
    }
}

bool is(const char *a, const char *b) { return strcmp(a, b) == 0; }

template <typename T> int unary_float(const char *op, const T *a_, T *out, size_t n) {
    Dyn<T> a = load(a_, n), r;
    if      (is(op, "neg"))   r = -a;
    else if (is(op, "abs"))   r = abs(a);
    else if (is(op, "sqrt"))  r = sqrt(a);
    else if (is(op, "rcp"))   r = rcp(a);
    else if (is(op, "rsqrt")) r = rsqrt(a);
    else if (is(op, "floor")) r = floor(a);
    else if (is(op, "ceil"))  r = ceil(a);
    else if (is(op, "round")) r = round(a);
    else if (is(op, "trunc")) r = trunc(a);
    else if (is(op, "sin"))   r = sin(a);
    else if (is(op, "cos"))   r = cos(a);
    else if (is(op, "exp"))   r = exp(a);
    else if (is(op, "log"))   r = log(a);
    else if (is(op, "tan"))   r = tan(a);
    else if (is(op, "asin"))  r = asin(a);
    else if (is(op, "acos"))  r = acos(a);
    else if (is(op, "atan"))  r = atan(a);
    else if (is(op, "sinh"))  r = sinh(a);
    else if (is(op, "cosh"))  r = cosh(a);
    else if (is(op, "tanh"))  r = tanh(a);
    else if (is(op, "cot"))   r = cot(a);
    else if (is(op, "asinh")) r = asinh(a);
    else if (is(op, "acosh")) r = acosh(a);
    else if (is(op, "atanh")) r = atanh(a);
    else if (is(op, "cbrt"))  r = cbrt(a);
    else if (is(op, "erf"))    r = erf(a);
    else if (is(op, "erfc"))   r = erfc(a);
    else if (is(op, "erfinv")) r = erfinv(a);
    else if (is(op, "i0e"))    r = i0e(a);
    else if (is(op, "dawson")) r = dawson(a);
    else if (is(op, "erfi"))   r = erfi(a);
    else if (is(op, "lgamma")) r = lgamma(a);
    else if (is(op, "tgamma")) r = tgamma(a);
    else if (is(op, "sign"))  r = sign(a);
    else return -1;
    store(r, out, n);
    return 0;
}

template <typename T> int unary_int(const char *op, const T *a_, T *out, size_t n) {
    Dyn<T> a = load(a_, n), r;
    if      (is(op, "neg"))    r = -a;
    else if (is(op, "not"))    r = ~a;
    else if (is(op, "abs"))    r = abs(a);
    else if (is(op, "popcnt")) r = popcnt(a);
    else if (is(op, "lzcnt"))  r = lzcnt(a);
    else if (is(op, "tzcnt"))  r = tzcnt(a);
    else return -1;
    store(r, out, n);
    return 0;
}

template <typename T> int binary_float(const char *op, const T *a_, const T *b_, T *out, size_t n) {
    Dyn<T> a = load(a_, n), b = load(b_, n), r;
    if      (is(op, "add")) r = a + b;
    else if (is(op, "sub")) r = a - b;
    else if (is(op, "mul")) r = a * b;
    else if (is(op, "div")) r = a / b;
    else if (is(op, "min")) r = min(a, b);
    else if (is(op, "max")) r = max(a, b);
    else if (is(op, "atan2")) r = atan2(a, b);
    else if (is(op, "pow"))   r = pow(a, b);
    else if (is(op, "fmod"))  r = fmod(a, b);
    else if (is(op, "ldexp")) r = ldexp(a, b);
    else if (is(op, "safe_mul")) {
        /* restates the CPU branch of safe_mul, src/autodiff/autodiff.cpp:1191-1205 */
        Dyn<T> t = a * b, z = T(0);
        r = select(eq(a, z) || eq(b, z), z, t);
    }
    else return -1;
    store(r, out, n);
    return 0;
}

template <typename T> int binary_int(const char *op, const T *a_, const T *b_, T *out, size_t n) {
    Dyn<T> a = load(a_, n), b = load(b_, n), r;
    if      (is(op, "add"))   r = a + b;
    else if (is(op, "sub"))   r = a - b;
    else if (is(op, "mul"))   r = a * b;
    else if (is(op, "div"))   r = a / b;
    else if (is(op, "mod"))   r = a % b;
    else if (is(op, "min"))   r = min(a, b);
    else if (is(op, "max"))   r = max(a, b);
    else if (is(op, "mulhi")) r = mulhi(a, b);
    else if (is(op, "and"))   r = a & b;
    else if (is(op, "or"))    r = a | b;
    else if (is(op, "xor"))   r = a ^ b;
    else if (is(op, "sl"))    r = a << b;
    else if (is(op, "sr"))    r = a >> b;
    else return -1;
    store(r, out, n);
    return 0;
}

template <typename T> int ternary_any(const char *op, const T *a_, const T *b_, const T *c_, T *out, size_t n) {
    Dyn<T> a = load(a_, n), b = load(b_, n), c = load(c_, n), r;
    if      (is(op, "fmadd"))  r = fmadd(a, b, c);
    else if (is(op, "fmsub"))  r = fmsub(a, b, c);
    else if (is(op, "fnmadd")) r = fnmadd(a, b, c);
    else if (is(op, "fnmsub")) r = fnmsub(a, b, c);
    else if (is(op, "safe_fmadd")) {
        /* CPU branch of safe_fmadd, src/autodiff/autodiff.cpp:1207-1221 */
        Dyn<T> t = fmadd(a, b, c), z = T(0);
        r = select(eq(a, z) || eq(b, z), c, t);
    }
    else return -1;
    store(r, out, n);
    return 0;
}

template <typename T> int compare_any(const char *op, const T *a_, const T *b_, uint8_t *out, size_t n) {
    Dyn<T> a = load(a_, n), b = load(b_, n);
    mask_t<Dyn<T>> m;
    if      (is(op, "eq"))  m = eq(a, b);
    else if (is(op, "neq")) m = neq(a, b);
    else if (is(op, "lt"))  m = a < b;
    else if (is(op, "le"))  m = a <= b;
    else if (is(op, "gt"))  m = a > b;
    else if (is(op, "ge"))  m = a >= b;
    else return -1;
    store_mask(m, out, n);
    return 0;
}

template <typename T> int select_any(const uint8_t *m_, const T *t_, const T *f_, T *out, size_t n) {
    auto m = load_mask<T>(m_, n);
    Dyn<T> r = select(m, load(t_, n), load(f_, n));
    store(r, out, n);
    return 0;
}

template <typename S, typename D> int cast_any(const S *a_, D *out, size_t n) {
    Dyn<D> r = Dyn<D>(load(a_, n));
    store(r, out, n);
    return 0;
}

template <typename T, typename I>
int gather_any(const T *base, size_t /*src_size*/, const I *idx_, const uint8_t *mask_, T *out, size_t n) {
    Dyn<I> idx = load(idx_, n);
    auto m = load_mask<T>(mask_, n);
    Dyn<T> r = gather<Dyn<T>>((const void *) base, idx, m);
    store(r, out, n);
    return 0;
}

template <typename T, typename I>
int scatter_any(int add, T *base, const T *val_, const I *idx_, const uint8_t *mask_, size_t n) {
    Dyn<I> idx = load(idx_, n);
    Dyn<T> val = load(val_, n);
    auto m = load_mask<T>(mask_, n);
    if (add) scatter_add((void *) base, val, idx, m);
    else     scatter((void *) base, val, idx, m);
    return 0;
}

template <typename T> int reduce_any(const char *op, const T *a_, T *out, size_t n) {
    Dyn<T> a = n ? load(a_, n) : Dyn<T>();
    if      (is(op, "hsum"))  *out = hsum(a);
    else if (is(op, "hprod")) *out = hprod(a);
    else if (is(op, "hmax"))  *out = hmax(a);
    else if (is(op, "hmin"))  *out = hmin(a);
    else return -1;
    return 0;
}

double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace


extern "C" {

/* type codes shared with include/enoki_hip.h: 1=i32 2=u32 3=i64 4=u64 5=f32 6=f64 */

int ref_packet_width() { return (int) Packet<float>::Size; }

int ref_unary(int type, const char *op, const void *a, void *out, size_t n) {
    switch (type) {
        case 1: return unary_int<int32_t>(op, (const int32_t *) a, (int32_t *) out, n);
        case 2: return unary_int<uint32_t>(op, (const uint32_t *) a, (uint32_t *) out, n);
        case 3: return unary_int<int64_t>(op, (const int64_t *) a, (int64_t *) out, n);
        case 4: return unary_int<uint64_t>(op, (const uint64_t *) a, (uint64_t *) out, n);
        case 5: return unary_float<float>(op, (const float *) a, (float *) out, n);
        case 6: return unary_float<double>(op, (const double *) a, (double *) out, n);
    }
    return -2;
}

int ref_binary(int type, const char *op, const void *a, const void *b, void *out, size_t n) {
    switch (type) {
        case 1: return binary_int<int32_t>(op, (const int32_t *) a, (const int32_t *) b, (int32_t *) out, n);
        case 2: return binary_int<uint32_t>(op, (const uint32_t *) a, (const uint32_t *) b, (uint32_t *) out, n);
        case 3: return binary_int<int64_t>(op, (const int64_t *) a, (const int64_t *) b, (int64_t *) out, n);
        case 4: return binary_int<uint64_t>(op, (const uint64_t *) a, (const uint64_t *) b, (uint64_t *) out, n);
        case 5: return binary_float<float>(op, (const float *) a, (const float *) b, (float *) out, n);
        case 6: return binary_float<double>(op, (const double *) a, (const double *) b, (double *) out, n);
    }
    return -2;
}

int ref_ternary(int type, const char *op, const void *a, const void *b, const void *c, void *out, size_t n) {
    switch (type) {
        case 1: return ternary_any<int32_t>(op, (const int32_t *) a, (const int32_t *) b, (const int32_t *) c, (int32_t *) out, n);
        case 2: return ternary_any<uint32_t>(op, (const uint32_t *) a, (const uint32_t *) b, (const uint32_t *) c, (uint32_t *) out, n);
        case 5: return ternary_any<float>(op, (const float *) a, (const float *) b, (const float *) c, (float *) out, n);
        case 6: return ternary_any<double>(op, (const double *) a, (const double *) b, (const double *) c, (double *) out, n);
    }
    return -2;
}

int ref_sincos(int type, const void *a_, void *s_, void *c_, size_t n) {
    if (type == 5) {
        auto [s, c] = sincos(load((const float *) a_, n));
        store(s, (float *) s_, n); store(c, (float *) c_, n);
        return 0;
    } else if (type == 6) {
        auto [s, c] = sincos(load((const double *) a_, n));
        store(s, (double *) s_, n); store(c, (double *) c_, n);
        return 0;
    }
    return -2;
}

int ref_compare(int type, const char *op, const void *a, const void *b, uint8_t *out, size_t n) {
    switch (type) {
        case 1: return compare_any<int32_t>(op, (const int32_t *) a, (const int32_t *) b, out, n);
        case 2: return compare_any<uint32_t>(op, (const uint32_t *) a, (const uint32_t *) b, out, n);
        case 3: return compare_any<int64_t>(op, (const int64_t *) a, (const int64_t *) b, out, n);
        case 4: return compare_any<uint64_t>(op, (const uint64_t *) a, (const uint64_t *) b, out, n);
        case 5: return compare_any<float>(op, (const float *) a, (const float *) b, out, n);
        case 6: return compare_any<double>(op, (const double *) a, (const double *) b, out, n);
    }
    return -2;
}

int ref_select(int type, const uint8_t *m, const void *t, const void *f, void *out, size_t n) {
    switch (type) {
        case 1: return select_any<int32_t>(m, (const int32_t *) t, (const int32_t *) f, (int32_t *) out, n);
        case 2: return select_any<uint32_t>(m, (const uint32_t *) t, (const uint32_t *) f, (uint32_t *) out, n);
        case 3: return select_any<int64_t>(m, (const int64_t *) t, (const int64_t *) f, (int64_t *) out, n);
        case 4: return select_any<uint64_t>(m, (const uint64_t *) t, (const uint64_t *) f, (uint64_t *) out, n);
        case 5: return select_any<float>(m, (const float *) t, (const float *) f, (float *) out, n);
        case 6: return select_any<double>(m, (const double *) t, (const double *) f, (double *) out, n);
    }
    return -2;
}

int ref_cast(int src, int dst, const void *a, void *out, size_t n) {
#define CAST_ROW(S, st)                                                                   \
    if (src == S) {                                                                       \
        switch (dst) {                                                                    \
            case 1: return cast_any<st, int32_t>((const st *) a, (int32_t *) out, n);     \
            case 2: return cast_any<st, uint32_t>((const st *) a, (uint32_t *) out, n);   \
            case 3: return cast_any<st, int64_t>((const st *) a, (int64_t *) out, n);     \
            case 4: return cast_any<st, uint64_t>((const st *) a, (uint64_t *) out, n);   \
            case 5: return cast_any<st, float>((const st *) a, (float *) out, n);         \
            case 6: return cast_any<st, double>((const st *) a, (double *) out, n);       \
        }                                                                                 \
    }
    CAST_ROW(1, int32_t) CAST_ROW(2, uint32_t) CAST_ROW(3, int64_t)
    CAST_ROW(4, uint64_t) CAST_ROW(5, float) CAST_ROW(6, double)
#undef CAST_ROW
    return -2;
}

/* index type: 1=i32 2=u32 3=i64 4=u64; value type 1..6 (4- and 8-byte values) */
int ref_gather(int type, int itype, const void *base, size_t src_size, const void *idx,
               const uint8_t *mask, void *out, size_t n) {
#define G(T, I) return gather_any<T, I>((const T *) base, src_size, (const I *) idx, mask, (T *) out, n)
    if (type == 5 && itype == 2) G(float, uint32_t);
    if (type == 5 && itype == 1) G(float, int32_t);
    if (type == 1 && itype == 2) G(int32_t, uint32_t);
    if (type == 1 && itype == 1) G(int32_t, int32_t);
    if (type == 2 && itype == 2) G(uint32_t, uint32_t);
    if (type == 2 && itype == 1) G(uint32_t, int32_t);
    if (type == 6 && itype == 4) G(double, uint64_t);
    if (type == 6 && itype == 3) G(double, int64_t);
    if (type == 3 && itype == 3) G(int64_t, int64_t);
    if (type == 4 && itype == 4) G(uint64_t, uint64_t);
#undef G
    return -2;
}

int ref_scatter(int type, int itype, int add, void *base, const void *val, const void *idx,
                const uint8_t *mask, size_t n) {
#define S(T, I) return scatter_any<T, I>(add, (T *) base, (const T *) val, (const I *) idx, mask, n)
    if (type == 5 && itype == 2) S(float, uint32_t);
    if (type == 5 && itype == 1) S(float, int32_t);
    if (type == 1 && itype == 2) S(int32_t, uint32_t);
    if (type == 1 && itype == 1) S(int32_t, int32_t);
    if (type == 2 && itype == 2) S(uint32_t, uint32_t);
    if (type == 2 && itype == 1) S(uint32_t, int32_t);
    if (type == 6 && itype == 4) S(double, uint64_t);
    if (type == 6 && itype == 3) S(double, int64_t);
    if (type == 3 && itype == 3) S(int64_t, int64_t);
    if (type == 4 && itype == 4) S(uint64_t, uint64_t);
#undef S
    return -2;
}

int ref_reduce(int type, const char *op, const void *a, void *out, size_t n) {
    switch (type) {
        case 1: return reduce_any<int32_t>(op, (const int32_t *) a, (int32_t *) out, n);
        case 2: return reduce_any<uint32_t>(op, (const uint32_t *) a, (uint32_t *) out, n);
        case 3: return reduce_any<int64_t>(op, (const int64_t *) a, (int64_t *) out, n);
        case 4: return reduce_any<uint64_t>(op, (const uint64_t *) a, (uint64_t *) out, n);
        case 5: return reduce_any<float>(op, (const float *) a, (float *) out, n);
        case 6: return reduce_any<double>(op, (const double *) a, (double *) out, n);
    }
    return -2;
}

/* mask reductions: op = all | any | count ; result as uint64 */
int ref_mask_reduce(const char *op, const uint8_t *m_, uint64_t *out, size_t n) {
    if (n == 0) {
        mask_t<Dyn<float>> m;
        if      (is(op, "all"))   *out = all(m);
        else if (is(op, "any"))   *out = any(m);
        else if (is(op, "count")) *out = count(m);
        else return -1;
        return 0;
    }
    auto m = load_mask<float>(m_, n);
    if      (is(op, "all"))   *out = all(m);
    else if (is(op, "any"))   *out = any(m);
    else if (is(op, "count")) *out = count(m);
    else return -1;
    return 0;
}

int ref_arange_f32(float *out, size_t n) { store(arange<Dyn<float>>(n), out, n); return 0; }
int ref_arange_u32(uint32_t *out, size_t n) { store(arange<Dyn<uint32_t>>(n), out, n); return 0; }
int ref_linspace_f32(float lo, float hi, float *out, size_t n) { store(linspace<Dyn<float>>(lo, hi, n), out, n); return 0; }
int ref_reverse_f32(const float *a, float *out, size_t n) { store(reverse(load(a, n)), out, n); return 0; }
int ref_psum_f32(const float *a, float *out, size_t n) { store(psum(load(a, n)), out, n); return 0; }

/* ------------------------------------------------------------------ */
/* BASELINE.json configs, run through the reference's own types.       */
/* Each returns the elapsed seconds of the timed region in *seconds.   */
/* ------------------------------------------------------------------ */

/* cfg1: hsum(fmadd(a, x, b)) on DynamicArray<Packet<float,8>>  (tests/dynamic.cpp style) */
float ref_cfg1(const float *a_, const float *x_, const float *b_, size_t n, double *seconds) {
    Dyn<float> a = load(a_, n), x = load(x_, n), b = load(b_, n);
    double t0 = now();
    float y = hsum(fmadd(a, x, b));
    if (seconds) *seconds = now() - t0;
    return y;
}

/* cfg2: hsum(sin(exp(fmadd(a, x, b)))) */
float ref_cfg2(const float *a_, const float *x_, const float *b_, size_t n, double *seconds) {
    Dyn<float> a = load(a_, n), x = load(x_, n), b = load(b_, n);
    double t0 = now();
    float y = hsum(sin(exp(fmadd(a, x, b))));
    if (seconds) *seconds = now() - t0;
    return y;
}

/* cfg3a: y = hsum(sin(fmadd(a, x, b))); backward(y); a, b leaves of size n, x plain. */
float ref_cfg3a(const float *a_, const float *x_, const float *b_, size_t n,
                float *grad_a, float *grad_b, double *seconds) {
    FloatD::set_log_level_(0);
    FloatD a = FloatX::copy(a_, n), x = FloatX::copy(x_, n), b = FloatX::copy(b_, n);
    set_requires_gradient(a);
    set_requires_gradient(b);
    double t0 = now();
    FloatD y = hsum(sin(fmadd(a, x, b)));
    backward(y);
    if (seconds) *seconds = now() - t0;
    if (grad_a) store(gradient(a), grad_a, n);
    if (grad_b) store(gradient(b), grad_b, n);
    return y.value_().coeff(0);
}

/* cfg3b: a = gather(A, idx), b = gather(B, idx) with A, B leaves of size K;
          y = hsum(sin(fmadd(a, x, b))); backward(y) -> grads land in K-arrays via scatter_add. */
float ref_cfg3b(const float *A_, const float *B_, size_t k, const float *x_, const uint32_t *idx_,
                size_t n, float *grad_A, float *grad_B, double *seconds) {
    FloatD::set_log_level_(0);
    FloatD A = FloatX::copy(A_, k), B = FloatX::copy(B_, k), x = FloatX::copy(x_, n);
    UInt32D idx = UInt32X::copy(idx_, n);
    set_requires_gradient(A);
    set_requires_gradient(B);
    double t0 = now();
    FloatD a = gather<FloatD>(A, idx), b = gather<FloatD>(B, idx);
    FloatD y = hsum(sin(fmadd(a, x, b)));
    backward(y);
    if (seconds) *seconds = now() - t0;
    if (grad_A) store(gradient(A), grad_A, k);
    if (grad_B) store(gradient(B), grad_B, k);
    return y.value_().coeff(0);
}

/* The neighbours of cfg3b that bench.py times next to it (round 4): y = seed * hsum(f(fmadd(gather(A, idx, mask), x,
   gather(B, idx, mask)))) with f = sin (0) | cos (1) | exp (2) | log (3) | sqrt (4) | rcp (5) | rsqrt (6), a 32- or 64-bit index array and an optional
   mask; backward() of the scaled loss.  func + 16 * spelling: u written with operators instead of fmadd (see below). */
float ref_cfg3b_variant(const float *A_, const float *B_, size_t k, const float *x_, const void *idx_, int idx64,
                        const uint8_t *mask_, size_t n, int func, float seed, float *grad_A, float *grad_B, double *seconds) {
    FloatD::set_log_level_(0);
    FloatD A = FloatX::copy(A_, k), B = FloatX::copy(B_, k), x = FloatX::copy(x_, n);
    using UInt64X = DynamicArray<Packet<uint64_t, Packet<float>::Size>>;
    using UInt64D = DiffArray<UInt64X>;
    mask_t<FloatD> mask = true;
    if (mask_) {
        mask_t<FloatX> m;
        set_slices(m, n);
        for (size_t i = 0; i < n; ++i) m.coeff(i) = mask_[i] != 0;
        mask = mask_t<FloatD>(m);
    }
    set_requires_gradient(A);
    set_requires_gradient(B);
    double t0 = now();
    FloatD a, b;
    if (idx64) {
        UInt64D idx = UInt64X::copy(idx_, n);
        a = gather<FloatD>(A, idx, mask); b = gather<FloatD>(B, idx, mask);
    } else {
        UInt32D idx = UInt32X::copy(idx_, n);
        a = gather<FloatD>(A, idx, mask); b = gather<FloatD>(B, idx, mask);
    }
    // how u is written: 0 fmadd(a, x, b) | 1 a * x + b | 2 a * x - b | 3 b - a * x | 4 b + a * x  (operators: a product and a sum
    // with a rounding each -- separate packet operations are never contracted)
    const int spelling = func >> 4;
    func &= 15;
    //                         5 a * x  (the product of ONE gather with an array; B is not used, its gradient is reported as zero)
    FloatD u = spelling == 0 ? fmadd(a, x, b) : spelling == 1 ? FloatD(a * x + b) : spelling == 2 ? FloatD(a * x - b)
             : spelling == 3 ? FloatD(b - a * x) : spelling == 4 ? FloatD(b + a * x) : FloatD(a * x);
    FloatD y = hsum(func == 0 ? sin(u) : func == 1 ? cos(u) : func == 2 ? exp(u) : func == 3 ? log(u) : func == 4 ? sqrt(u) : func == 5 ? rcp(u) : rsqrt(u));
    FloatD z = seed == 1.f ? y : y * seed;
    backward(z);
    if (seconds) *seconds = now() - t0;
    if (grad_A) store(gradient(A), grad_A, k);
    if (grad_B) {
        if (spelling == 5) memset(grad_B, 0, k * sizeof(float));
        else store(gradient(B), grad_B, k);
    }
    return z.value_().coeff(0);
}

/* cfg5 (BASELINE configs[4], synthetic): the templated path tracer of examples/path_trace.h on the reference's own arrays.
   loss = hsum(radiance over n paths), backward() -> grad_tex (K = width * width texels). */
float ref_cfg5(const float *tex_, size_t k, size_t n, uint64_t seed, uint64_t first_lane, int bounces, uint32_t width,
               float *grad_tex, double *seconds) {
    FloatD::set_log_level_(0);
    using UInt64X = DynamicArray<Packet<uint64_t, Packet<float>::Size>>;
    FloatD tex = FloatX::copy(tex_, k);
    set_requires_gradient(tex);
    double t0 = now();
    PCG32<FloatX> rng(UInt64X(seed), arange<UInt64X>(n) + UInt64X(first_lane));
    auto lookup = [&](const UInt32X &texel) { return gather<FloatD>(tex, UInt32D(texel)); };
    FloatD radiance = bounces == 1 ? cfg5::path_trace<1, FloatD, FloatX>(rng, lookup, width)
                    : bounces == 2 ? cfg5::path_trace<2, FloatD, FloatX>(rng, lookup, width)
                                   : cfg5::path_trace<3, FloatD, FloatX>(rng, lookup, width);
    if (bounces < 1 || bounces > 3) return 0.f / 0.f;
    FloatD y = hsum(radiance);
    backward(y);
    if (seconds) *seconds = now() - t0;
    if (grad_tex) store(gradient(tex), grad_tex, k);
    return y.value_().coeff(0);
}

/* the texels that the paths of ref_cfg5 look up, bounce by bounce (n entries per bounce) */
int ref_cfg5_texels(size_t n, uint64_t seed, uint64_t first_lane, int bounces, uint32_t width, uint32_t *out) {
    using UInt64X = DynamicArray<Packet<uint64_t, Packet<float>::Size>>;
    PCG32<FloatX> rng(UInt64X(seed), arange<UInt64X>(n) + UInt64X(first_lane));
    int k = 0;
    auto lookup = [&](const UInt32X &texel) { store(texel, out + (size_t) (k++) * n, n); return FloatX(.5f); };
    FloatX radiance = bounces == 1 ? cfg5::path_trace<1, FloatX, FloatX>(rng, lookup, width)
                    : bounces == 2 ? cfg5::path_trace<2, FloatX, FloatX>(rng, lookup, width)
                                   : cfg5::path_trace<3, FloatX, FloatX>(rng, lookup, width);
    (void) radiance;
    return 0;
}

/* first-bounce quantities of ref_cfg5, one row of n floats each (debugging aid of tools/debug_cfg5.py): z r phi s c b cc t px py pz
   theta ph uu vv texel r1 r2 */
int ref_cfg5_trace(size_t n, uint64_t seed, uint64_t first_lane, uint32_t width, float *out) {
    using UInt64X = DynamicArray<Packet<uint64_t, Packet<float>::Size>>;
    using Vector3 = Array<FloatX, 3>;
    const float pi = 3.14159265358979323846f;
    PCG32<FloatX> rng(UInt64X(seed), arange<UInt64X>(n) + UInt64X(first_lane));
    FloatX z = 1.f - 2.f * rng.next_float32();
    FloatX r = sqrt(max(FloatX(0.f), 1.f - z * z));
    FloatX phi = (2.f * pi) * rng.next_float32();
    auto [s_, c_] = sincos(phi);
    Vector3 d(r * c_, r * s_, z), o(FloatX(.1f), FloatX(.2f), FloatX(-.1f));
    FloatX b = dot(o, d), c = dot(o, o) - 1.f;
    FloatX t = sqrt(max(FloatX(0.f), b * b - c)) - b;
    Vector3 v = o + d * t;
    FloatX l = sqrt(dot(v, v));
    Vector3 p(v.x() / l, v.y() / l, v.z() / l);
    FloatX theta = acos(min(max(p.z(), FloatX(-1.f)), FloatX(1.f)));
    FloatX ph = atan2(p.y(), p.x());
    FloatX uu = fmadd(ph, FloatX(.5f / pi), FloatX(.5f)), vv = theta * (1.f / pi);
    UInt32X ix = min(UInt32X(uu * float(width)), UInt32X(width - 1)), iy = min(UInt32X(vv * float(width)), UInt32X(width - 1));
    FloatX texel = FloatX(iy * width + ix);
    FloatX r1 = 2.f * rng.next_float32() - 1.f, r2 = 2.f * rng.next_float32() - 1.f;
    Vector3 nrm = p * FloatX(-1.f);
    auto swap = abs(r1) < abs(r2);
    FloatX rad = select(swap, r2, r1);
    FloatX ratio = select(swap, r1, r2) / select(eq(rad, FloatX(0.f)), FloatX(1.f), rad);
    FloatX ang = select(swap, (.5f * pi) - (.25f * pi) * ratio, (.25f * pi) * ratio);
    auto [sn, cs] = sincos(ang);
    FloatX dx = rad * cs, dy = rad * sn;
    FloatX dz = sqrt(max(FloatX(0.f), 1.f - dx * dx - dy * dy));
    FloatX sign = copysign(FloatX(1.f), nrm.z());
    FloatX a = FloatX(-1.f) / (sign + nrm.z());
    FloatX bb = nrm.x() * nrm.y() * a;
    Vector3 sx(1.f + sign * nrm.x() * nrm.x() * a, sign * bb, FloatX(-1.f) * sign * nrm.x());
    Vector3 ty(bb, sign + nrm.y() * nrm.y() * a, FloatX(-1.f) * nrm.y());
    Vector3 w = sx * dx + ty * dy + nrm * dz;
    FloatX wl = sqrt(dot(w, w));
    Vector3 d2(w.x() / wl, w.y() / wl, w.z() / wl);
    Vector3 o2 = p + nrm * FloatX(1e-3f);
    const FloatX *rows[] = { &z, &r, &phi, &s_, &c_, &b, &c, &t, &p.x(), &p.y(), &p.z(), &theta, &ph, &uu, &vv, &texel, &r1, &r2,
                             &rad, &ratio, &ang, &sn, &cs, &dx, &dy, &dz, &sign, &a, &bb, &sx.x(), &sx.y(), &sx.z(), &ty.x(), &ty.y(),
                             &ty.z(), &w.x(), &w.y(), &w.z(), &d2.x(), &d2.y(), &d2.z(), &o2.x(), &o2.y(), &o2.z() };
    for (size_t q = 0; q < sizeof(rows) / sizeof(rows[0]); ++q) store(*rows[q], out + q * n, n);
    return 0;
}

/* Generic little tape programs used by the tape parity tests: see tests/test_tape_parity.py.
   prog: a sequence of (opcode, arg0, arg1, arg2) int32 quadruples acting on a register file of
   FloatD values.  Registers [0, n_in) are preloaded from `inputs` (each of length sizes[i]) and
   flagged as leaves when leaf[i] != 0.  The last register written is the output; mode 0 =
   backward, 1 = forward (seeded from leaf `fwd_leaf`).  Gradients of all leaves (backward) or
   of the output (forward) are written to grads[i] (backward) / grads[0] (forward). */
enum {
    P_ADD = 0, P_SUB, P_MUL, P_DIV, P_FMADD, P_NEG, P_ABS, P_SQRT, P_RCP, P_RSQRT, P_SIN, P_COS,
    P_EXP, P_LOG, P_HSUM, P_HPROD, P_MIN, P_MAX, P_GATHER, P_SCATTER_ADD, P_SCATTER, P_SELECT_GT0,
    P_MULC, P_ADDC, P_TANH, P_TAN, P_ATAN2, P_FMSUB, P_FNMADD, P_FNMSUB, P_SINH, P_COSH, P_ASIN,
    P_ACOS, P_ATAN, P_PSUM, P_REVERSE, P_ASINH, P_ACOSH, P_ATANH, P_CBRT, P_POW, P_COT
};

int ref_tape_program(const int32_t *prog, size_t n_ops, const float *const *inputs,
                     const uint64_t *sizes, const uint8_t *leaf, size_t n_in,
                     const uint32_t *const *index_inputs, const uint64_t *index_sizes, size_t n_idx,
                     int mode, int fwd_leaf, int simplify,
                     float *out_value, uint64_t *out_size, float *const *grads) {
    FloatD::set_log_level_(0);
    std::vector<FloatD> reg(n_in + n_ops);
    std::vector<UInt32D> ireg(n_idx);
    for (size_t i = 0; i < n_in; ++i) {
        /* a size-1 DynamicArray must be built from a scalar: copy(ptr, 1) leaves lanes 1..7 of packet 0
           undefined, and broadcasting reads the whole packet (dynamic.h:303-320) */
        if (sizes[i] == 1) reg[i] = FloatX(inputs[i][0]);
        else               reg[i] = FloatX::copy(inputs[i], sizes[i]);
        if (leaf[i]) set_requires_gradient(reg[i]);
    }
    for (size_t i = 0; i < n_idx; ++i)
        ireg[i] = UInt32X::copy(index_inputs[i], index_sizes[i]);

    size_t last = n_in ? n_in - 1 : 0;
    for (size_t k = 0; k < n_ops; ++k) {
        const int32_t *p = prog + 4 * k;
        size_t d = n_in + k;
        auto R = [&](int32_t i) -> FloatD & { return reg[(size_t) i]; };
        float cst; memcpy(&cst, &p[2], sizeof(float));
        switch (p[0]) {
            case P_ADD:    reg[d] = R(p[1]) + R(p[2]); break;
            case P_SUB:    reg[d] = R(p[1]) - R(p[2]); break;
            case P_MUL:    reg[d] = R(p[1]) * R(p[2]); break;
            case P_DIV:    reg[d] = R(p[1]) / R(p[2]); break;
            case P_FMADD:  reg[d] = fmadd(R(p[1]), R(p[2]), R(p[3])); break;
            case P_FMSUB:  reg[d] = fmsub(R(p[1]), R(p[2]), R(p[3])); break;
            case P_FNMADD: reg[d] = fnmadd(R(p[1]), R(p[2]), R(p[3])); break;
            case P_FNMSUB: reg[d] = fnmsub(R(p[1]), R(p[2]), R(p[3])); break;
            case P_NEG:    reg[d] = -R(p[1]); break;
            case P_ABS:    reg[d] = abs(R(p[1])); break;
            case P_SQRT:   reg[d] = sqrt(R(p[1])); break;
            case P_RCP:    reg[d] = rcp(R(p[1])); break;
            case P_RSQRT:  reg[d] = rsqrt(R(p[1])); break;
            case P_SIN:    reg[d] = sin(R(p[1])); break;
            case P_COS:    reg[d] = cos(R(p[1])); break;
            case P_TAN:    reg[d] = tan(R(p[1])); break;
            case P_SINH:   reg[d] = sinh(R(p[1])); break;
            case P_COSH:   reg[d] = cosh(R(p[1])); break;
            case P_TANH:   reg[d] = tanh(R(p[1])); break;
            case P_ASIN:   reg[d] = asin(R(p[1])); break;
            case P_ACOS:   reg[d] = acos(R(p[1])); break;
            case P_ATAN:   reg[d] = atan(R(p[1])); break;
            case P_ATAN2:  reg[d] = atan2(R(p[1]), R(p[2])); break;
            case P_COT:    reg[d] = cot(R(p[1])); break;
            case P_ASINH:  reg[d] = asinh(R(p[1])); break;
            case P_ACOSH:  reg[d] = acosh(R(p[1])); break;
            case P_ATANH:  reg[d] = atanh(R(p[1])); break;
            case P_CBRT:   reg[d] = cbrt(R(p[1])); break;
            case P_POW:    reg[d] = pow(R(p[1]), R(p[2])); break;
            case P_EXP:    reg[d] = exp(R(p[1])); break;
            case P_LOG:    reg[d] = log(R(p[1])); break;
            case P_HSUM:   reg[d] = hsum(R(p[1])); break;
            case P_HPROD:  reg[d] = hprod(R(p[1])); break;
            case P_PSUM:   reg[d] = psum(R(p[1])); break;
            case P_REVERSE:reg[d] = reverse(R(p[1])); break;
            case P_MIN:    reg[d] = min(R(p[1]), R(p[2])); break;
            case P_MAX:    reg[d] = max(R(p[1]), R(p[2])); break;
            case P_MULC:   reg[d] = R(p[1]) * cst; break;
            case P_ADDC:   reg[d] = R(p[1]) + cst; break;
            case P_SELECT_GT0: reg[d] = select(R(p[1]) > 0.f, R(p[2]), R(p[3])); break;
            case P_GATHER: reg[d] = gather<FloatD>(R(p[1]), ireg[(size_t) p[2]]); break;
            case P_SCATTER_ADD: /* target p[1] (modified in place), value p[2], index p[3] */
                scatter_add(R(p[1]), R(p[2]), ireg[(size_t) p[3]]);
                reg[d] = R(p[1]);
                break;
            case P_SCATTER:
                scatter(R(p[1]), R(p[2]), ireg[(size_t) p[3]]);
                reg[d] = R(p[1]);
                break;
            default: return -1;
        }
        last = d;
    }

    FloatD &y = reg[last];
    *out_size = y.size();
    store(y.value_(), out_value, y.size());
    if (simplify) FloatD::simplify_graph_();

    if (mode == 0) {
        backward(y);
        for (size_t i = 0; i < n_in; ++i) {
            if (!leaf[i]) continue;
            const FloatX &g = gradient(reg[i]);
            if (g.size() == 0) {            /* leaf not reached: reference leaves grad empty */
                for (size_t j = 0; j < sizes[i]; ++j) grads[i][j] = 0.f;
            } else {
                store(g, grads[i], sizes[i]);
            }
        }
    } else {
        forward(reg[(size_t) fwd_leaf]);
        const FloatX &g = gradient(y);
        store(g, grads[0], y.size());
    }
    return 0;
}


/* ------------------------------------------------------------------ */
/* BASELINE config 4: masked gather/scatter ray-sphere intersection.    */
/* The three kernels are the user-level program of tests/sphere.cpp     */
/* (make_rays 58-64, intersect_rays 67-78, shade_hits 81-83) written    */
/* against the reference's own Array<FloatX, N> types; the surrounding  */
/* gather / scatter / count follow SURVEY.md 8d (cfg4).                 */
/* ------------------------------------------------------------------ */
} // extern "C" (templates below)

#if !defined(ENOKI_REF_TAPE_ONLY)   /* the AVX-512 flavour (make ref512) only serves ref_tape_program: the helpers below pin 8-wide packets */

namespace {
using Vector2fX = Array<FloatX, 2>;
using Vector3fX = Array<FloatX, 3>;
using MaskX = mask_t<FloatX>;

template <typename Vector3> struct RayT {
    Vector3 o, d;
    Vector3 operator()(const value_t<Vector3> &t) const { return o + t * d; }
};

template <typename Vector2> auto sphere_make_rays(const Vector2 &p) {
    using Vector3 = Array<value_t<Vector2>, 3>;
    return RayT<Vector3>{ Vector3(p.x(), p.y(), -1.f), Vector3(0.f, 0.f, 1.f) };
}

template <typename Ray, typename Mask> auto sphere_intersect(const Ray &r, Mask &hit) {
    auto a = dot(r.d, r.d);
    auto b = 2.f * dot(r.o, r.d);
    auto c = dot(r.o, r.o) - 1.f;
    auto discrim = b * b - 4.f * a * c;
    auto t = (-b + sqrt(discrim)) / (2.f * a);
    hit = discrim >= 0.f;
    return select(hit, r(t), 0.f);
}

template <typename Vector3> auto sphere_shade(const Vector3 &n) {
    return 0.2f + max(dot(n, Vector3(-1.f, -1.f, 2.f)), 0.f) * 90.f;
}
} // namespace

extern "C" {

int ref_cfg4(const float *gx, const float *gy, const uint32_t *perm_, const uint8_t *mask_, size_t n,
             float *image /* n, pre-initialised by the caller */, uint64_t *hit_count) {
    Vector2fX p(FloatX::copy(gx, n), FloatX::copy(gy, n));
    UInt32X perm = UInt32X::copy(perm_, n);
    MaskX mask = MaskX(load_mask<float>(mask_, n));
    Vector2fX pp = gather<Vector2fX>(p, perm, mask);
    MaskX hit;
    auto pos = sphere_intersect(sphere_make_rays(pp), hit);
    FloatX shade = sphere_shade(pos);
    hit = hit & mask;
    FloatX img = FloatX::copy(image, n);
    scatter(img, shade, perm, hit);
    store(img, image, n);
    *hit_count = count(hit);
    return 0;
}

/* PCG32<FloatX> (include/enoki/random.h): a fixed script of draws.  out_u32: steps x n (masked draws),
   then one draw each of float32 / uint64 / float64 / uint32_bounded(bound), advance(delta), one more uint32,
   and the final state.  n must be a multiple of 8 (packet width; masked u64 lanes in the padding are unused). */
int ref_pcg32(uint64_t initstate, const uint64_t *initseq, size_t n, int steps, const uint8_t *mask_,
              uint32_t *out_u32, float *out_f32, uint64_t *out_u64, double *out_f64, uint32_t bound,
              uint32_t *out_bounded, int64_t delta, uint32_t *out_after, uint64_t *state_out) {
    using RNG = PCG32<FloatX>;
    using UInt64X = RNG::UInt64;
    RNG rng(UInt64X(initstate), UInt64X::copy(initseq, n));
    auto mask = mask_t<UInt64X>(load_mask<uint64_t>(mask_, n));
    for (int s = 0; s < steps; ++s)
        store(rng.next_uint32(mask), out_u32 + (size_t) s * n, n);
    store(rng.next_float32(), out_f32, n);
    store(rng.next_uint64(), out_u64, n);
    store(rng.next_float64(), out_f64, n);
    store(rng.next_uint32_bounded(bound), out_bounded, n);
    rng.advance(RNG::Int64(delta));
    store(rng.next_uint32(), out_after, n);
    store(rng.state, state_out, n);
    return 0;
}

} // extern "C"

/* Matrix<FloatX, N> (include/enoki/matrix.h): entries are passed row-major, entry (i, j) = row i * N + j of an
   (N*N, n) array.  Outputs: a * b, a * v, trace(a), frob(a), det(a) and inverse(a). */
namespace {
template <size_t N> int ref_matrix_impl(const float *a_, const float *b_, const float *v_, size_t n, float *mm, float *mv,
                                        float *tr, float *fr, float *dt, float *inv) {
    using M = Matrix<FloatX, N>;
    using V = Array<FloatX, N>;
    M a, b; V v;
    for (size_t i = 0; i < N; ++i) {
        for (size_t j = 0; j < N; ++j) {
            a(i, j) = FloatX::copy(a_ + (i * N + j) * n, n);
            b(i, j) = FloatX::copy(b_ + (i * N + j) * n, n);
        }
        v[i] = FloatX::copy(v_ + i * n, n);
    }
    M c = a * b;
    V w = a * v;
    for (size_t i = 0; i < N; ++i) {
        for (size_t j = 0; j < N; ++j)
            store(FloatX(c(i, j)), mm + (i * N + j) * n, n);
        store(FloatX(w[i]), mv + i * n, n);
    }
    store(FloatX(trace(a)), tr, n);
    store(FloatX(frob(a)), fr, n);
    {
        store(FloatX(det(a)), dt, n);
        M ia = inverse(a);
        for (size_t i = 0; i < N; ++i)
            for (size_t j = 0; j < N; ++j)
                store(FloatX(ia(i, j)), inv + (i * N + j) * n, n);
    }
    return 0;
}
} // namespace

extern "C" int ref_matrix(int size, const float *a, const float *b, const float *v, size_t n, float *mm, float *mv,
                          float *tr, float *fr, float *dt, float *inv) {
    switch (size) {
        case 2: return ref_matrix_impl<2>(a, b, v, n, mm, mv, tr, fr, dt, inv);
        case 3: return ref_matrix_impl<3>(a, b, v, n, mm, mv, tr, fr, dt, inv);
        case 4: return ref_matrix_impl<4>(a, b, v, n, mm, mv, tr, fr, dt, inv);
    }
    return -1;
}

/* Complex<FloatX> (include/enoki/complex.h): a, b are (2, n) arrays {re, im}.  out is (10, 2, n):
   a*b, a/b, exp(a), log(a), sqrt(a), pow(a, b), sin(a), cos(a), rcp(a), {abs(a), arg(a)} */
extern "C" int ref_complex(const float *a_, const float *b_, size_t n, float *out) {
    using C = Complex<FloatX>;
    C a(FloatX::copy(a_, n), FloatX::copy(a_ + n, n)), b(FloatX::copy(b_, n), FloatX::copy(b_ + n, n));
    C r[9] = { a * b, a / b, exp(a), log(a), sqrt(a), pow(a, b), sin(a), cos(a), rcp(a) };
    for (int k = 0; k < 9; ++k) {
        store(FloatX(real(r[k])), out + ((size_t) k * 2 + 0) * n, n);
        store(FloatX(imag(r[k])), out + ((size_t) k * 2 + 1) * n, n);
    }
    store(FloatX(abs(a)), out + (size_t) 18 * n, n);
    store(FloatX(arg(a)), out + (size_t) 19 * n, n);
    return 0;
}

/* More of Complex<FloatX> (include/enoki/complex.h:196-267): out is (9, 2, n): sinh, cosh, tanh, asin, acos, atan, asinh,
   acosh, atanh of a */
extern "C" int ref_complex_more(const float *a_, size_t n, float *out) {
    using C = Complex<FloatX>;
    C a(FloatX::copy(a_, n), FloatX::copy(a_ + n, n));
    C r[9] = { sinh(a), cosh(a), tanh(a), asin(a), acos(a), atan(a), asinh(a), acosh(a), atanh(a) };
    for (int k = 0; k < 9; ++k) {
        store(FloatX(real(r[k])), out + ((size_t) k * 2 + 0) * n, n);
        store(FloatX(imag(r[k])), out + ((size_t) k * 2 + 1) * n, n);
    }
    return 0;
}

/* include/enoki/transform.h on Matrix<FloatX, 4> / Matrix<FloatX, 3>: v is (3, n) (a direction / offset), p is (6, n) =
   {angle, fov, near, far, aspect, unused}.  out is (8, 16, n), row-major entries of: translate(v), scale(v),
   rotate(normalize(v), angle), perspective(fov, near, far, aspect), frustum(-aspect, aspect, -1, 1, near, far),
   ortho(-aspect, aspect, -1, 1, near, far), look_at(v, v * 0.25 + 1, (0, 1, 0)), and the 3 x 3 rotate(angle) padded to 16. */
extern "C" int ref_transform(const float *v_, const float *p_, size_t n, float *out) {
    using M4 = Matrix<FloatX, 4>;
    using M3 = Matrix<FloatX, 3>;
    using V3 = Array<FloatX, 3>;
    V3 v(FloatX::copy(v_, n), FloatX::copy(v_ + n, n), FloatX::copy(v_ + 2 * n, n));
    FloatX angle = FloatX::copy(p_, n), fov = FloatX::copy(p_ + n, n), nr = FloatX::copy(p_ + 2 * n, n), fr = FloatX::copy(p_ + 3 * n, n),
           aspect = FloatX::copy(p_ + 4 * n, n), one = FloatX(1.f) + zero<FloatX>(n);
    M4 m[7] = { translate<M4>(v), scale<M4>(v), rotate<M4>(normalize(v), angle), perspective<M4>(fov, nr, fr, aspect),
                frustum<M4>(-aspect, aspect, -one, one, nr, fr), ortho<M4>(-aspect, aspect, -one, one, nr, fr),
                look_at<M4>(v, v * 0.25f + 1.f, V3(zero<FloatX>(n), one, zero<FloatX>(n))) };
    for (int k = 0; k < 7; ++k)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                FloatX e = m[k](i, j);
                if (e.size() == 1) e = e + zero<FloatX>(n);
                store(e, out + ((size_t) k * 16 + i * 4 + j) * n, n);
            }
    M3 r = rotate<M3>(angle);
    for (int i = 0; i < 16; ++i) {
        FloatX e = i < 9 ? FloatX(r(i / 3, i % 3)) : zero<FloatX>(n);
        if (e.size() == 1) e = e + zero<FloatX>(n);
        store(e, out + ((size_t) 7 * 16 + i) * n, n);
    }
    return 0;
}

/* sh_eval (include/enoki/sh.h, generated code for orders 0..9), element by element on Array<float, 3> (the reference's
   store() does not take DynamicArray): d is (3, n), out is ((order + 1)^2, n) */
extern "C" int ref_sh(const float *d_, size_t n, size_t order, float *out) {
    if (order > 9) return -1;
    const size_t count = (order + 1) * (order + 1);
    std::vector<float> coeffs(count);
    for (size_t i = 0; i < n; ++i) {
        sh_eval(Array<float, 3>(d_[i], d_[n + i], d_[2 * n + i]), order, coeffs.data());
        for (size_t k = 0; k < count; ++k) out[k * n + i] = coeffs[k];
    }
    return 0;
}

/* Quaternion<FloatX> (include/enoki/quaternion.h): a, b are (4, n) arrays {x, y, z, w}, t is (n).  out is (8, 4, n):
   a*b, a/b (= a*rcp(b)), exp(a), log(a), slerp(na, nb, t) for the normalised inputs, matrix_to_quat(quat_to_matrix(na)),
   sqrt(a), rcp(a); mat is (9, n): quat_to_matrix<Matrix3>(na) in row-major order. */
extern "C" int ref_quaternion(const float *a_, const float *b_, const float *t_, size_t n, float *out, float *mat) {
    using Q = Quaternion<FloatX>;
    auto load_q = [&](const float *p) { return Q(FloatX::copy(p, n), FloatX::copy(p + n, n), FloatX::copy(p + 2 * n, n), FloatX::copy(p + 3 * n, n)); };
    Q a = load_q(a_), b = load_q(b_);
    FloatX t = FloatX::copy(t_, n);
    Q na = normalize(a), nb = normalize(b);
    Matrix<FloatX, 3> m3 = quat_to_matrix<Matrix<FloatX, 3>>(na);
    Q r[8] = { a * b, a / b, exp(a), log(a), slerp(na, nb, t), matrix_to_quat(m3), sqrt(a), rcp(a) };
    for (int k = 0; k < 8; ++k)
        for (int c = 0; c < 4; ++c)
            store(FloatX(r[k].coeff(c)), out + ((size_t) k * 4 + c) * n, n);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            store(FloatX(m3(i, j)), mat + ((size_t) i * 3 + j) * n, n);
    return 0;
}

/* Elliptic integrals of the reference (include/enoki/special.h:314-672) on DynamicArray<Packet<T>>: out is (10, n):
   comp_ellint_1(k), comp_ellint_2(k), comp_ellint_3(k, nu), ellint_1(phi, k), ellint_2(phi, k), ellint_3(phi, k, nu),
   carlson_rf(x, y, z), carlson_rd(x, y, z), carlson_rc(x, y), carlson_rj(x, y, z, r) with x = phi^2, y = 1.5 - k^2,
   z = 1 + |nu|, r = 0.5 + |nu| (positive by construction). */
template <typename T> static int ellint_impl(const T *phi_, const T *k_, const T *nu_, size_t n, T *out) {
    using V = Dyn<T>;
    V phi = load(phi_, n), k = load(k_, n), nu = load(nu_, n);
    V x = phi * phi, y = T(1.5) - k * k, z = T(1) + abs(nu), r = T(0.5) + abs(nu);
    V res[10] = { comp_ellint_1(k), comp_ellint_2(k), comp_ellint_3(k, nu), ellint_1(phi, k), ellint_2(phi, k), ellint_3(phi, k, nu),
                  carlson_rf(Array<V, 3>(x, y, z)), carlson_rd(Array<V, 3>(x, y, z)), carlson_rc(Array<V, 2>(x, y)),
                  carlson_rj(Array<V, 4>(x, y, z, r)) };
    for (int i = 0; i < 10; ++i) store(res[i], out + (size_t) i * n, n);
    return 0;
}
extern "C" int ref_ellint_f32(const float *phi, const float *k, const float *nu, size_t n, float *out) { return ellint_impl<float>(phi, k, nu, n, out); }
extern "C" int ref_ellint_f64(const double *phi, const double *k, const double *nu, size_t n, double *out) { return ellint_impl<double>(phi, k, nu, n, out); }

#endif /* !ENOKI_REF_TAPE_ONLY */
