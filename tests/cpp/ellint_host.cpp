// Host evaluation of include/enoki/ellint.h on one-element packets (Array<T, 1>, the unit that vectorize() kernels are
// instantiated on): lets the CPU suite check the restated Carlson / Legendre integrals against the reference build's
// golden vectors without a GPU (tests/test_special.py).  g++ -O1 -ffp-contract=off -shared.
#include <enoki/special.h>

using namespace enoki;

template <typename T> static void run(const T *phi, const T *k, const T *nu, size_t n, T *out) {
    using V = Array<T, 1>;
    for (size_t i = 0; i < n; ++i) {
        V p(phi[i]), kk(k[i]), v(nu[i]);
        V x = p * p, y = V(T(1.5)) - kk * kk, z = V(T(1)) + abs(v), r = V(T(0.5)) + abs(v);
        V res[10] = { comp_ellint_1(kk), comp_ellint_2(kk), comp_ellint_3(kk, v), ellint_1(p, kk), ellint_2(p, kk), ellint_3(p, kk, v),
                      carlson_rf(x, y, z), carlson_rd(x, y, z), carlson_rc(x, y), carlson_rj(x, y, z, r) };
        for (int j = 0; j < 10; ++j) out[(size_t) j * n + i] = res[j].coeff(0);
    }
}

extern "C" void ellint_host_f32(const float *phi, const float *k, const float *nu, size_t n, float *out) { run<float>(phi, k, nu, n, out); }
extern "C" void ellint_host_f64(const double *phi, const double *k, const double *nu, size_t n, double *out) { run<double>(phi, k, nu, n, out); }
