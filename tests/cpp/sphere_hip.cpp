// BASELINE config 4 on the product types: the user-level ray-sphere program (cf. the reference's
// tests/sphere.cpp:58-83, tests/ray.h) instantiated over Array<HIPArray<float>, N>, plus the masked
// gather / masked scatter / count wrapper of SURVEY.md 8d.  Compared against oracle/_ref's ref_cfg4 and the
// C oracle's orc_cfg4 by tests/test_sphere_gpu.py.
#include <enoki/hip.h>

using namespace enoki;
using FloatC = HIPArray<float>;
using UInt32C = HIPArray<uint32_t>;
using MaskC = HIPArray<bool>;
using Vector2fC = Array<FloatC, 2>;
using Vector3fC = Array<FloatC, 3>;

template <typename Vector_> struct Ray {
    using Vector = Vector_;
    using Value = value_t<Vector>;
    Vector o, d;
    Vector operator()(const Value &t) const { return o + t * d; }
    ENOKI_STRUCT(Ray, o, d)
};
ENOKI_STRUCT_SUPPORT(Ray, o, d)

template <typename Vector2> auto make_rays(const Vector2 &p) {
    using Vector3 = Array<value_t<Vector2>, 3>;
    return Ray<Vector3>(Vector3(p.x(), p.y(), -1.f), Vector3(0.f, 0.f, 1.f));
}

template <typename RayT, typename Mask> typename RayT::Vector intersect_rays(const RayT &r, Mask &hit) {
    auto a = dot(r.d, r.d);
    auto b = 2.f * dot(r.o, r.d);
    auto c = dot(r.o, r.o) - 1.f;
    auto discrim = b * b - 4.f * a * c;
    auto t = (-b + sqrt(discrim)) / (2.f * a);
    hit = discrim >= 0.f;
    return select(hit, r(t), 0.f);
}

template <typename Vector3> typename Vector3::Value shade_hits(const Vector3 &n) {
    return 0.2f + max(dot(n, Vector3(-1.f, -1.f, 2.f)), 0.f) * 90.f;
}

extern "C" __attribute__((visibility("default")))
int hip_cfg4(const float *gx, const float *gy, const uint32_t *perm_, const uint8_t *mask_, size_t n, float *image,
             uint64_t *hit_count) {
    try {
        Vector2fC p(FloatC::copy(gx, n), FloatC::copy(gy, n));
        UInt32C perm = UInt32C::copy(perm_, n);
        MaskC mask = MaskC::copy(mask_, n);
        Vector2fC pp = gather<Vector2fC>(p, perm, mask);
        MaskC hit;
        Vector3fC pos = intersect_rays(make_rays(pp), hit);
        FloatC shade = shade_hits(pos);
        hit = hit & mask;
        FloatC img = FloatC::copy(image, n);
        scatter(img, shade, perm, hit);
        auto host = img.to_host();
        memcpy(image, host.data(), n * sizeof(float));
        *hit_count = count(hit);
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "hip_cfg4: %s\n", e.what());
        return -3;
    }
}

/// grid = meshgrid(linspace(-1.2, 1.2, res))^2 built with the product's own initializers (sphere.cpp:130-131)
extern "C" __attribute__((visibility("default")))
int hip_sphere_grid(size_t res, float *gx, float *gy) {
    try {
        FloatC idx = linspace<FloatC>(-1.2f, 1.2f, res);
        Vector2fC grid = meshgrid(idx, idx);
        auto hx = grid.x().to_host(), hy = grid.y().to_host();
        memcpy(gx, hx.data(), hx.size() * sizeof(float));
        memcpy(gy, hy.data(), hy.size() * sizeof(float));
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "hip_sphere_grid: %s\n", e.what());
        return -3;
    }
}
