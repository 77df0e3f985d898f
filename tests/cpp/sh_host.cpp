// include/enoki/sh.h on host scalars: CPU-suite check against scipy's complex harmonics and the reference build's generated
// code (tests/test_sh.py).  out is ((order + 1)^2, n).
#include <enoki/sh.h>

#include <vector>

using namespace enoki;

extern "C" void sh_host_f64(const double *d, size_t n, size_t order, double *out) {
    const size_t count = (order + 1) * (order + 1);
    std::vector<double> tmp(count);
    for (size_t i = 0; i < n; ++i) {
        sh_eval(Array<double, 3>(d[i], d[n + i], d[2 * n + i]), order, tmp.data());
        for (size_t k = 0; k < count; ++k) out[k * n + i] = tmp[k];
    }
}
extern "C" void sh_host_f32(const float *d, size_t n, size_t order, float *out) {
    const size_t count = (order + 1) * (order + 1);
    std::vector<float> tmp(count);
    for (size_t i = 0; i < n; ++i) {
        sh_eval(Array<float, 3>(d[i], d[n + i], d[2 * n + i]), order, tmp.data());
        for (size_t k = 0; k < count; ++k) out[k * n + i] = tmp[k];
    }
}
