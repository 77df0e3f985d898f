// Vectorised virtual method calls on device pointer arrays (include/enoki/array_call.h), in the style of the
// reference's tests/call.cpp and tests/autodiff.cpp:564-607 (test35_call): a small class hierarchy whose
// methods take and return HIPArray / Array<HIPArray, 3> / DiffArray values.  Driven by tests/test_call_gpu.py,
// which computes the expected lanes with the CPU oracle's elementwise ops.
#include <enoki/array_call.h>
#include <enoki/autodiff.h>

using namespace enoki;
using FloatC = HIPArray<float>;
using UInt32C = HIPArray<uint32_t>;
using MaskC = HIPArray<bool>;
using Vector3fC = Array<FloatC, 3>;
using FloatD = DiffArray<FloatC>;

struct Shape {
    virtual ~Shape() = default;
    /// elementwise response; receives the lane mask as trailing argument
    virtual FloatC eval(const FloatC &x, const MaskC &active) const = 0;
    /// nested array argument and result, no mask parameter
    virtual Vector3fC offset(const Vector3fC &p, const FloatC &t) const = 0;
    /// returns a plain scalar: becomes one value per lane
    virtual float id() const = 0;
    /// void method with a side effect on the instance
    virtual void touch(const FloatC &x, const MaskC &active) = 0;
    /// differentiable argument / result
    virtual FloatD eval_d(const FloatD &x) const = 0;
    size_t lanes_seen = 0;
    size_t active_seen = 0;
    double tag_value = 0.0;
};

struct Sine : Shape {
    float amp;
    explicit Sine(float amp) : amp(amp) { }
    FloatC eval(const FloatC &x, const MaskC &active) const override { return select(active, sin(x) * amp, -1.f); }
    Vector3fC offset(const Vector3fC &p, const FloatC &t) const override { return p + Vector3fC(t, 0.f, amp); }
    float id() const override { return 1.f; }
    void touch(const FloatC &x, const MaskC &active) override { lanes_seen += x.size(); active_seen += count(active); }
    FloatD eval_d(const FloatD &x) const override { return sin(x) * amp; }
};

struct Poly : Shape {
    float c0, c1;
    Poly(float c0, float c1) : c0(c0), c1(c1) { }
    FloatC eval(const FloatC &x, const MaskC &active) const override { return select(active, fmadd(x, c1, c0), -2.f); }
    Vector3fC offset(const Vector3fC &p, const FloatC &t) const override { return p * t; }
    float id() const override { return 2.f; }
    void touch(const FloatC &x, const MaskC &active) override { lanes_seen += x.size(); active_seen += count(active); }
    FloatD eval_d(const FloatD &x) const override { return fmadd(x, c1, c0) * x; }
};

ENOKI_CALL_SUPPORT_BEGIN(Shape)
ENOKI_CALL_SUPPORT_METHOD(eval)
ENOKI_CALL_SUPPORT_METHOD(offset)
ENOKI_CALL_SUPPORT_METHOD(id)
ENOKI_CALL_SUPPORT_METHOD(touch)
ENOKI_CALL_SUPPORT_METHOD(eval_d)
ENOKI_CALL_SUPPORT_GETTER(lanes, lanes_seen)
ENOKI_CALL_SUPPORT_GETTER_TYPE(tag, tag_value, float)
ENOKI_CALL_SUPPORT_END(Shape)

using ShapePtrC = HIPArray<Shape *>;

static void to_host(const FloatC &a, float *dst, size_t n) {
    auto h = a.to_host();
    if (h.size() == 1 && n != 1) { for (size_t i = 0; i < n; ++i) dst[i] = h[0]; }
    else memcpy(dst, h.data(), n * sizeof(float));
}

/// which[i]: 0 -> Sine(1.5), 1 -> Poly(0.25, -2), 2 -> Sine(-0.5), 255 -> nullptr.  Outputs are n floats each;
/// part_* describe partition(): group count, per group the instance number and size, then all permutations.
extern "C" __attribute__((visibility("default")))
int hip_call_test(const uint8_t *which, const float *x_, const float *t_, const uint8_t *mask_, size_t n, float *out_eval,
                  float *out_eval_masked, float *out_off /* 3n */, float *out_id, uint64_t *touch_stats /* 3 x 2 */,
                  float *out_d, float *out_grad, uint32_t *part_groups /* 1 + 2*4 */, uint32_t *part_perm /* n */) {
    try {
        Sine s0(1.5f), s2(-0.5f);
        Poly p1(0.25f, -2.f);
        Shape *table[3] = { &s0, &p1, &s2 };
        std::vector<Shape *> host(n);
        for (size_t i = 0; i < n; ++i) host[i] = which[i] < 3 ? table[which[i]] : nullptr;
        ShapePtrC shapes = ShapePtrC::copy(host.data(), n);
        FloatC x = FloatC::copy(x_, n), t = FloatC::copy(t_, n);
        MaskC mask = MaskC::copy(mask_, n);

        to_host(shapes->eval(x), out_eval, n);
        to_host(shapes->eval(x, mask), out_eval_masked, n);
        Vector3fC off = shapes->offset(Vector3fC(x, t, 1.f), t);
        for (int k = 0; k < 3; ++k) to_host(off.coeff(k), out_off + k * n, n);
        to_host(shapes->id(), out_id, n);
        shapes->touch(x, mask);
        for (int k = 0; k < 3; ++k) { touch_stats[2 * k] = table[k]->lanes_seen; touch_stats[2 * k + 1] = table[k]->active_seen; }

        FloatD xd(x);
        set_requires_gradient(xd);
        FloatD yd = shapes->eval_d(xd);
        to_host(detach(yd), out_d, n);
        backward(hsum(yd));
        to_host(gradient(xd), out_grad, n);

        const auto &groups = partition(shapes);
        part_groups[0] = (uint32_t) groups.size();
        size_t pos = 0;
        for (size_t g = 0; g < groups.size() && g < 4; ++g) {
            Shape *p = groups[g].first;
            if (g > 0 && !((uintptr_t) p > (uintptr_t) groups[g - 1].first))
                return -4;                       // groups must come in ascending pointer order (horiz.cu:49-57)
            uint32_t number = 255;
            for (uint32_t k = 0; k < 3; ++k) if (p == table[k]) number = k;
            part_groups[1 + 2 * g] = number;
            part_groups[2 + 2 * g] = (uint32_t) groups[g].second.size();
            auto h = groups[g].second.to_host();
            memcpy(part_perm + pos, h.data(), h.size() * sizeof(uint32_t));
            pos += h.size();
        }
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "hip_call_test: %s\n", e.what());
        return -3;
    }
}

// ---- ENOKI_STRUCT types as arguments / results of vectorised calls, and struct-wise gather / scatter / select -------
template <typename Value_> struct Hit {
    using Value = Value_;
    using Vector3 = Array<Value_, 3>;
    Vector3 p;
    Value t;
    mask_t<Value_> valid;
    ENOKI_STRUCT(Hit, p, t, valid)
};
ENOKI_STRUCT_SUPPORT(Hit, p, t, valid)

using HitC = Hit<FloatC>;

struct Mover {
    virtual ~Mover() = default;
    virtual HitC advance(const HitC &h, const FloatC &dt) const = 0;
};
struct Forward : Mover {
    HitC advance(const HitC &h, const FloatC &dt) const override { return HitC(h.p + Vector3fC(dt, 0.f, 0.f), h.t + dt, h.valid); }
};
struct Flip : Mover {
    HitC advance(const HitC &h, const FloatC &dt) const override { return HitC(-h.p, h.t * dt, !h.valid); }
};

ENOKI_CALL_SUPPORT_BEGIN(Mover)
ENOKI_CALL_SUPPORT_METHOD(advance)
ENOKI_CALL_SUPPORT_END(Mover)

/// outputs: px, py, pz, t (n floats each), valid (n bytes); then struct-wise gather(reverse permutation) -> gx, gt
extern "C" __attribute__((visibility("default")))
int hip_struct_test(const uint8_t *which, const float *x_, const float *dt_, size_t n, float *px, float *py, float *pz,
                    float *t_out, uint8_t *valid_out, float *gx, float *gt, uint64_t *zero_slices) {
    try {
        Forward fwd; Flip flip;
        Mover *table[2] = { &fwd, &flip };
        std::vector<Mover *> host(n);
        for (size_t i = 0; i < n; ++i) host[i] = which[i] < 2 ? table[which[i]] : nullptr;
        HIPArray<Mover *> movers = HIPArray<Mover *>::copy(host.data(), n);
        FloatC x = FloatC::copy(x_, n), dt = FloatC::copy(dt_, n);
        HitC h(Vector3fC(x, x * 2.f, 1.f), x + 1.f, x > 0.f);
        HitC r = movers->advance(h, dt);
        to_host(r.p.x(), px, n); to_host(r.p.y(), py, n); to_host(r.p.z(), pz, n); to_host(r.t, t_out, n);
        auto vm = r.valid.to_host();
        for (size_t i = 0; i < n; ++i) valid_out[i] = vm.size() == 1 ? vm[0] : vm[i];

        UInt32C rev = UInt32C(uint32_t(n - 1)) - arange<UInt32C>(n);
        HitC g = gather<HitC>(r, rev);                        // struct-wise gather
        to_host(g.p.x(), gx, n); to_host(g.t, gt, n);
        HitC z = zero<HitC>(n);                               // struct-wise zero / slices / scatter / select
        zero_slices[0] = slices(z);
        scatter(z, g, rev);
        HitC sel = select(x > 0.f, z, h);
        zero_slices[1] = slices(sel);
        auto a = sel.t.to_host(), b = r.t.to_host(), c = h.t.to_host();
        auto xm = x.to_host();
        for (size_t i = 0; i < n; ++i)
            if (a[i] != (xm[i] > 0.f ? b[i] : c[i])) return -5;
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "hip_struct_test: %s\n", e.what());
        return -3;
    }
}

// ---- masked assignment: masked(x, m) = v, x[m] op= v on arrays, nested arrays, differentiable arrays and structs ----
extern "C" __attribute__((visibility("default")))
int hip_masked_test(const float *x_, size_t n, float *out_assign, float *out_add, float *out_vec_y, float *out_struct_t,
                    float *out_grad) {
    try {
        FloatC x = FloatC::copy(x_, n);
        MaskC m = x > 0.f;
        FloatC a = x;
        masked(a, m) = 5.f;                               // a = x > 0 ? 5 : x
        to_host(a, out_assign, n);
        FloatC b = x;
        b[m] += x * 2.f;                                  // b = x > 0 ? 3x : x
        b[!m] *= -1.f;                                    //     x <= 0: -x
        to_host(b, out_add, n);
        Vector3fC v(x, x + 1.f, 2.f);
        masked(v, m) = Vector3fC(0.f, 7.f, 0.f);
        to_host(v.y(), out_vec_y, n);                     // x > 0 ? 7 : x + 1
        HitC h(v, x, m);
        masked(h, !m) = HitC(v, x * 0.f - 3.f, m);        // x <= 0: t = -3
        to_host(h.t, out_struct_t, n);
        FloatD xd(x);
        set_requires_gradient(xd);
        FloatD yd = xd * xd;
        yd[DiffArray<MaskC>(m)] = xd * 3.f;               // y = x > 0 ? 3x : x^2
        backward(hsum(yd));
        to_host(gradient(xd), out_grad, n);               // x > 0 ? 3 : 2x
        float s = 1.f;
        masked(s, true) = 4.f;
        masked(s, false) = 9.f;
        return s == 4.f ? 0 : -6;
    } catch (const std::exception &e) {
        fprintf(stderr, "hip_masked_test: %s\n", e.what());
        return -3;
    }
}


/// getters: per-lane value of a data member (float view of a double field, and a size_t counter)
extern "C" __attribute__((visibility("default")))
int hip_getter_test(const uint8_t *which, const uint8_t *mask_, size_t n, float *out_tag, float *out_tag_masked, uint64_t *out_lanes) {
    try {
        Sine s0(1.5f), s2(-0.5f);
        Poly p1(0.25f, -2.f);
        s0.tag_value = 10.5; p1.tag_value = -3.25; s2.tag_value = 7.0;
        s0.lanes_seen = 11; p1.lanes_seen = 22; s2.lanes_seen = 33;
        Shape *table[3] = { &s0, &p1, &s2 };
        std::vector<Shape *> host(n);
        for (size_t i = 0; i < n; ++i) host[i] = which[i] < 3 ? table[which[i]] : nullptr;
        ShapePtrC shapes = ShapePtrC::copy(host.data(), n);
        MaskC mask = MaskC::copy(mask_, n);
        to_host(shapes->tag(), out_tag, n);
        to_host(shapes->tag(mask), out_tag_masked, n);
        auto lanes = shapes->lanes().to_host();
        for (size_t i = 0; i < n; ++i) out_lanes[i] = lanes.size() == 1 ? lanes[0] : lanes[i];
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "hip_getter_test: %s\n", e.what());
        return -3;
    }
}

// ---- getters on the device: instances in pinned host memory (ENOKI_PINNED_OPERATOR_NEW) ---------------------------------
struct Light {
    ENOKI_PINNED_OPERATOR_NEW(Light)
    virtual ~Light() = default;
    virtual float id() const = 0;
    float power = 0.f;
    double tag_value = 0.0;
    uint32_t samples = 0;
};
struct Spot : Light { float id() const override { return 1.f; } float cone = 0.5f; };
struct Area : Light { float id() const override { return 2.f; } double area = 2.0; char pad[24]; };

ENOKI_CALL_SUPPORT_BEGIN(Light)
ENOKI_CALL_SUPPORT_METHOD(id)
ENOKI_CALL_SUPPORT_GETTER(power, power)
ENOKI_CALL_SUPPORT_GETTER(samples, samples)
ENOKI_CALL_SUPPORT_GETTER_TYPE(tag, tag_value, float)
ENOKI_CALL_SUPPORT_END(Light)

/// which[i] selects one of `instances` heap-allocated lights (odd: Spot, even: Area; 0xFFFFFFFF: null).  Outputs per lane;
/// launches[0] = kernel launches of the three getters together (device path: one gather each, + one cast for the double
/// field -- whatever the number of instances).
extern "C" __attribute__((visibility("default")))
int hip_getter_device_test(const uint32_t *which, const uint8_t *mask_, size_t n, uint32_t instances, float *out_power,
                           float *out_tag_masked, uint32_t *out_samples, uint64_t *launches) {
    try {
        std::vector<std::unique_ptr<Light>> lights;
        for (uint32_t k = 0; k < instances; ++k) {
            if (k & 1) lights.emplace_back(new Spot()); else lights.emplace_back(new Area());
            lights.back()->power = 0.5f * (float) k + 1.f;
            lights.back()->tag_value = 1000.0 - (double) k;
            lights.back()->samples = 7u * k + 3u;
        }
        std::vector<Light *> host(n);
        for (size_t i = 0; i < n; ++i) host[i] = which[i] < instances ? lights[which[i]].get() : nullptr;
        using LightPtrC = HIPArray<Light *>;
        LightPtrC ptrs = LightPtrC::copy(host.data(), n);
        MaskC mask = MaskC::copy(mask_, n);
        const uint64_t l0 = ek_hip_launch_count();
        FloatC power = ptrs->power();
        FloatC tag = ptrs->tag(mask);
        HIPArray<uint32_t> samples = ptrs->samples();
        launches[0] = ek_hip_launch_count() - l0;
        to_host(power, out_power, n);
        to_host(tag, out_tag_masked, n);
        auto hs = samples.to_host();
        memcpy(out_samples, hs.data(), n * sizeof(uint32_t));
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "hip_getter_device_test: %s\n", e.what());
        return -3;
    }
}

// ---- partition() with many instances ---------------------------------------------------------------------------------
// `which[i]` selects one of `instances` objects (or none: which[i] == 0xFFFFFFFF -> null pointer).  Returns, per group in
// partition order, the instance number (0xFFFFFFFF for null) and the group size; `perm_out` receives the concatenated
// lane lists; `elapsed_ms` the wall time of partition() itself (cold: the array has no cached partition).
#include <chrono>
extern "C" __attribute__((visibility("default")))
int hip_partition_many(const uint32_t *which, size_t n, uint32_t instances, uint32_t *group_instance, uint32_t *group_size,
                       uint32_t *n_groups, uint32_t *perm_out, double *elapsed_ms, uint64_t *launches) {
    try {
        struct Dummy { virtual ~Dummy() = default; double pad[3]; };
        std::vector<Dummy> pool(instances);
        std::vector<Dummy *> host(n);
        for (size_t i = 0; i < n; ++i) host[i] = which[i] == 0xFFFFFFFFu ? nullptr : &pool[which[i]];
        HIPArray<Dummy *> ptrs = HIPArray<Dummy *>::copy(host.data(), n);
        hip_sync();
        const uint64_t before = ek_hip_launch_count();
        auto t0 = std::chrono::steady_clock::now();
        const auto &groups = partition(ptrs);
        hip_sync();
        *elapsed_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        *launches = ek_hip_launch_count() - before;
        *n_groups = (uint32_t) groups.size();
        size_t at = 0;
        for (size_t g = 0; g < groups.size(); ++g) {
            group_instance[g] = groups[g].first ? (uint32_t) (groups[g].first - pool.data()) : 0xFFFFFFFFu;
            auto lanes = groups[g].second.to_host();
            group_size[g] = (uint32_t) lanes.size();
            memcpy(perm_out + at, lanes.data(), lanes.size() * sizeof(uint32_t));
            at += lanes.size();
        }
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "hip_partition_many: %s\n", e.what());
        return -3;
    }
}
