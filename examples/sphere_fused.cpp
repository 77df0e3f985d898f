// BASELINE config 4 (masked gather of the pixel grid through a random permutation -> make_rays ->
// intersect_rays -> shade_hits -> masked scatter, count(hit)) as ONE kernel through enoki::vectorize().
//
// The three kernels are the user-level templates of the reference's tests/sphere.cpp:58-83 over tests/ray.h's Ray
// (re-declared here the way a user of this library would write them); vectorize() instantiates them on one-element
// packets inside a single __global__ kernel, with the indirect accesses done by the raw-memory gather / scatter of
// array.h on device pointers captured by the functor.  Per ray the kernel touches perm 4 B + mask 1 B + two 4-byte
// lookups + one 4-byte scatter + the hit mask 1 B = 18 B, against ~318 B for the same program run op by op
// (tests/cpp/sphere_hip.cpp, bench.py --workload cfg4).  Results are bit-identical (tests/test_sphere_gpu.py).
//
// Build: hipcc --offload-arch=gfx950 -x hip -ffp-contract=off (enoki_amd/_build.py) -> examples/libsphere_fused.so
#include <enoki/vectorize.h>
#include <enoki/vectorize_indexed.h>

#include <cstdio>
#include <vector>

using namespace enoki;
using FloatC = HIPArray<float>;
using UInt32C = HIPArray<uint32_t>;
using MaskC = HIPArray<bool>;
using FloatP = Array<float>;

ENOKI_DEVICE_CODE_BEGIN

template <typename Vector_> struct Ray {
    using Vector = Vector_;
    using Value = value_t<Vector>;
    Vector o, d;
    Vector operator()(const Value &t) const { return o + t * d; }
    ENOKI_STRUCT(Ray, o, d)
};

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

ENOKI_DEVICE_CODE_END

ENOKI_STRUCT_SUPPORT(Ray, o, d)

/// All pointers are DEVICE pointers (bench.py keeps the inputs resident); `image` (n floats) is updated in place.
extern "C" __attribute__((visibility("default")))
int sphere_fused_device(const float *gx, const float *gy, const uint32_t *perm_, const uint8_t *mask_, size_t n, float *image,
                        uint64_t *hit_count) {
    try {
        UInt32C perm = UInt32C::map((void *) perm_, n);
        MaskC mask = MaskC::map((void *) mask_, n);
        vectorize_indirect_bytes(12);          // two 4-byte lookups + one 4-byte scatter per ray through gx / gy / image
        MaskC hit = vectorize(
            [gx, gy, image](auto &&perm, auto &&mask) {
                using Vector2fP = Array<FloatP, 2>;
                using MaskP = mask_t<FloatP>;
                Vector2fP p(gather<FloatP>(gx, perm, mask), gather<FloatP>(gy, perm, mask));
                MaskP hit;
                auto pos = intersect_rays(make_rays(p), hit);
                FloatP shade = shade_hits(pos);
                hit = hit & mask;
                scatter(image, shade, perm, hit);
                return hit;
            },
            (const UInt32C &) perm, (const MaskC &) mask);
        if (hit_count) *hit_count = count(hit);
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "sphere_fused_device: %s\n", e.what());
        return -3;
    }
}

/// The same program over a PACKED pixel grid: `gxy` holds {x, y} records side by side (2 n floats), so the lookup through
/// the permutation is ONE 8-byte request per ray instead of two 4-byte ones (array.h gather_packed; the reference's
/// gather<Vector2fP>(mem, index) has the same record layout, array_router.h:1097-1107).  Same image, bit for bit.
extern "C" __attribute__((visibility("default")))
int sphere_fused_packed_device(const float *gxy, const uint32_t *perm_, const uint8_t *mask_, size_t n, float *image,
                               uint64_t *hit_count) {
    try {
        UInt32C perm = UInt32C::map((void *) perm_, n);
        MaskC mask = MaskC::map((void *) mask_, n);
        vectorize_indirect_bytes(12);          // one 8-byte lookup + one 4-byte scatter per ray
        MaskC hit = vectorize(
            [gxy, image](auto &&perm, auto &&mask) {
                using Vector2fP = Array<FloatP, 2>;
                using MaskP = mask_t<FloatP>;
                Vector2fP p = gather<Vector2fP>(gxy, perm, mask);
                MaskP hit;
                auto pos = intersect_rays(make_rays(p), hit);
                FloatP shade = shade_hits(pos);
                hit = hit & mask;
                scatter(image, shade, perm, hit);
                return hit;
            },
            (const UInt32C &) perm, (const MaskC &) mask);
        if (hit_count) *hit_count = count(hit);
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "sphere_fused_packed_device: %s\n", e.what());
        return -3;
    }
}

/// Host-pointer wrapper of the packed version: interleaves the two planes on the host, otherwise like sphere_fused()
extern "C" __attribute__((visibility("default")))
int sphere_fused_packed(const float *gx, const float *gy, const uint32_t *perm_, const uint8_t *mask_, size_t n, float *image,
                        uint64_t *hit_count) {
    try {
        std::vector<float> xy(2 * n);
        for (size_t i = 0; i < n; ++i) { xy[2 * i] = gx[i]; xy[2 * i + 1] = gy[i]; }
        FloatC dxy = FloatC::copy(xy.data(), 2 * n), img = FloatC::copy(image, n);
        UInt32C perm = UInt32C::copy(perm_, n);
        MaskC mask = MaskC::copy(mask_, n);
        int rc = sphere_fused_packed_device(dxy.data(), perm.data(), (const uint8_t *) mask.data(), n, img.data(), hit_count);
        if (rc) return rc;
        auto host = img.to_host();
        memcpy(image, host.data(), n * sizeof(float));
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "sphere_fused_packed: %s\n", e.what());
        return -3;
    }
}

/// Host-pointer convenience wrapper with the signature of the checkers (oracle orc_cfg4 / tests/cpp hip_cfg4)
extern "C" __attribute__((visibility("default")))
int sphere_fused(const float *gx, const float *gy, const uint32_t *perm_, const uint8_t *mask_, size_t n, float *image,
                 uint64_t *hit_count) {
    try {
        FloatC dgx = FloatC::copy(gx, n), dgy = FloatC::copy(gy, n), img = FloatC::copy(image, n);
        UInt32C perm = UInt32C::copy(perm_, n);
        MaskC mask = MaskC::copy(mask_, n);
        int rc = sphere_fused_device(dgx.data(), dgy.data(), perm.data(), (const uint8_t *) mask.data(), n, img.data(), hit_count);
        if (rc) return rc;
        auto host = img.to_host();
        memcpy(image, host.data(), n * sizeof(float));
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "sphere_fused: %s\n", e.what());
        return -3;
    }
}

/// The same program executed per PIXEL instead of per ray (enoki/vectorize_indexed.h): gather and scatter go through the same
/// permutation and the three kernels depend on the gathered pixel position only, so the rays are grouped by pixel bucket once
/// (ek_hip_index_partition_*) and every bucket's slice of the grid is read, and of the image written, IN ORDER: ~28 B per ray of
/// streaming traffic instead of two (packed: one and a half) 64-byte random accesses.  Same image, same hit count, bit for bit.
static auto through_body() {
    return [](auto &&px, auto &&py) {
        using Vector2fP = Array<FloatP, 2>;
        using MaskP = mask_t<FloatP>;
        MaskP hit;
        auto pos = intersect_rays(make_rays(Vector2fP(px, py)), hit);
        FloatP shade = shade_hits(pos);
        return std::pair<FloatP, MaskP>(shade, hit);
    };
}

extern "C" __attribute__((visibility("default")))
int sphere_through_device(const float *gx, const float *gy, const uint32_t *perm_, const uint8_t *mask_, size_t n, float *image,
                          uint64_t *hit_count) {
    try {
        UInt32C perm = UInt32C::map((void *) perm_, n);
        MaskC mask = MaskC::map((void *) mask_, n);
        FloatC x = FloatC::map((void *) gx, n), y = FloatC::map((void *) gy, n), img = FloatC::map((void *) image, n);
        size_t hits = vectorize_through(through_body(), (const UInt32C &) perm, (const MaskC &) mask, img, (const FloatC &) x,
                                        (const FloatC &) y);
        if (hit_count) *hit_count = hits;
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "sphere_through_device: %s\n", e.what());
        return -3;
    }
}

/// `image = full(background); scatter(image, shade, perm, hit)` of sphere.cpp:66-83 in the same pass (vectorize_through_fill):
/// `image` need not be initialised, every pixel is written.
extern "C" __attribute__((visibility("default")))
int sphere_through_fill_device(const float *gx, const float *gy, const uint32_t *perm_, const uint8_t *mask_, size_t n,
                               float background, float *image, uint64_t *hit_count) {
    try {
        UInt32C perm = UInt32C::map((void *) perm_, n);
        MaskC mask = MaskC::map((void *) mask_, n);
        FloatC x = FloatC::map((void *) gx, n), y = FloatC::map((void *) gy, n), img = FloatC::map((void *) image, n);
        size_t hits = vectorize_through_fill(through_body(), (const UInt32C &) perm, (const MaskC &) mask, background, img,
                                             (const FloatC &) x, (const FloatC &) y);
        if (hit_count) *hit_count = hits;
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "sphere_through_fill_device: %s\n", e.what());
        return -3;
    }
}

/// Host-pointer wrapper with the signature of the checkers
extern "C" __attribute__((visibility("default")))
int sphere_through(const float *gx, const float *gy, const uint32_t *perm_, const uint8_t *mask_, size_t n, float *image,
                   uint64_t *hit_count) {
    try {
        FloatC dgx = FloatC::copy(gx, n), dgy = FloatC::copy(gy, n), img = FloatC::copy(image, n);
        UInt32C perm = UInt32C::copy(perm_, n);
        MaskC mask = MaskC::copy(mask_, n);
        int rc = sphere_through_device(dgx.data(), dgy.data(), perm.data(), (const uint8_t *) mask.data(), n, img.data(), hit_count);
        if (rc) return rc;
        auto host = img.to_host();
        memcpy(image, host.data(), n * sizeof(float));
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "sphere_through: %s\n", e.what());
        return -3;
    }
}

/// Host-pointer wrapper of the fill variant: `image` is an output only
extern "C" __attribute__((visibility("default")))
int sphere_through_fill(const float *gx, const float *gy, const uint32_t *perm_, const uint8_t *mask_, size_t n, float background,
                        float *image, uint64_t *hit_count) {
    try {
        FloatC dgx = FloatC::copy(gx, n), dgy = FloatC::copy(gy, n), img = empty<FloatC>(n);
        UInt32C perm = UInt32C::copy(perm_, n);
        MaskC mask = MaskC::copy(mask_, n);
        int rc = sphere_through_fill_device(dgx.data(), dgy.data(), perm.data(), (const uint8_t *) mask.data(), n, background, img.data(),
                                            hit_count);
        if (rc) return rc;
        auto host = img.to_host();
        memcpy(image, host.data(), n * sizeof(float));
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "sphere_through_fill: %s\n", e.what());
        return -3;
    }
}
