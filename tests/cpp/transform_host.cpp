// Host evaluation of include/enoki/transform.h on scalar entries (Matrix<float, 4>): CPU-suite check against the reference
// build's golden matrices (tests/test_matrix.py).  Same parameterisation as oracle/ref_driver.cpp:ref_transform.
#include <enoki/transform.h>

using namespace enoki;

extern "C" void transform_host(const float *v_, const float *p_, size_t n, float *out) {
    using M4 = Matrix<float, 4>;
    using M3 = Matrix<float, 3>;
    using V3 = Array<float, 3>;
    for (size_t s = 0; s < n; ++s) {
        V3 v(v_[s], v_[n + s], v_[2 * n + s]);
        const float angle = p_[s], fov = p_[n + s], nr = p_[2 * n + s], fr = p_[3 * n + s], aspect = p_[4 * n + s];
        M4 m[7] = { translate<M4>(v), scale<M4>(v), rotate<M4>(normalize(v), angle), perspective<M4>(fov, nr, fr, aspect),
                    frustum<M4>(-aspect, aspect, -1.f, 1.f, nr, fr), ortho<M4>(-aspect, aspect, -1.f, 1.f, nr, fr),
                    look_at<M4>(v, v * 0.25f + 1.f, V3(0.f, 1.f, 0.f)) };
        for (int k = 0; k < 7; ++k)
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) out[((size_t) k * 16 + i * 4 + j) * n + s] = m[k](i, j);
        M3 r = rotate<M3>(angle);
        for (int i = 0; i < 16; ++i) out[((size_t) 7 * 16 + i) * n + s] = i < 9 ? r(i / 3, i % 3) : 0.f;
    }
}
