// BASELINE config 5 (SURVEY 8d "cfg5", synthetic -- not part of the reference's tests): the inner loop of a 3-bounce path tracer
// with gradients w.r.t. an albedo texture, written ONCE as a template over the array type.  The same source is instantiated
//   * by oracle/ref_driver.cpp against the REFERENCE's headers on DiffArray<DynamicArray<Packet<float, 8>>>  (ref_cfg5: the
//     parity oracle),
//   * by examples/path_trace.cpp against this repository's headers on DiffArray<HIPArray<float>> (op by op, every operation one
//     kernel, the texture lookups on the tape) and on one-element packets inside ONE kernel (enoki::vectorize_gather_grad).
// The pieces are the reference's own: PCG32 streams (include/enoki/random.h:71-119), the sphere intersection of
// tests/sphere.cpp:67-78, the concentric disk mapping of tests/autodiff.cpp:468-491, the differentiable gather of
// include/enoki/autodiff.h:962-998.  Only parity class A operations (IEEE arithmetic, sqrt, division, sincos, acos, atan2,
// float -> uint truncation, select) decide WHERE a path goes -- no rcp / rsqrt / normalize() -- so every implementation visits
// the same texels and the results differ by the order of fp additions only (class D).
//
//   Float   the differentiable value type (the texture's entries, throughput, radiance)
//   FloatC  its plain counterpart (geometry and sampling do not depend on the texture)
//   Tex     how a texel is looked up:  Float tex(const UInt32C &texel)
//   Bounces compile time: the loop is unrolled, so a fused kernel can keep one derivative slot per lookup in registers
#pragma once
#include <cstddef>
#include <cstdint>

namespace cfg5 {

template <int Bounces, typename Float, typename FloatC, typename RNG, typename Tex>
Float path_trace(RNG &rng, Tex &&tex, uint32_t width) {
    using namespace enoki;
    using UInt32C = uint32_array_t<FloatC>;
    using Vector3 = Array<FloatC, 3>;
    const float pi = 3.14159265358979323846f;
    // (normalize() goes through rsqrt, vector / value through rcp -- array_router.h -- both class C: divide component by component)
    auto unit = [](const Vector3 &v) { FloatC l = sqrt(dot(v, v)); return Vector3(v.x() / l, v.y() / l, v.z() / l); };

    // primary directions: uniform on the sphere; origin: a fixed point inside
    FloatC z = 1.f - 2.f * rng.next_float32();
    FloatC r = sqrt(enoki::max(FloatC(0.f), 1.f - z * z));
    FloatC phi = (2.f * pi) * rng.next_float32();
    auto [s_, c_] = sincos(phi);
    Vector3 d(r * c_, r * s_, z), o(FloatC(.1f), FloatC(.2f), FloatC(-.1f));
    Float throughput(1.f), radiance(0.f);
#pragma unroll
    for (int k = 0; k < Bounces; ++k) {
        FloatC b = dot(o, d), c = dot(o, o) - 1.f;
        FloatC t = sqrt(enoki::max(FloatC(0.f), b * b - c)) - b;                      // far root: we are inside the sphere
        Vector3 p = unit(o + d * t);
        FloatC theta = acos(enoki::min(enoki::max(p.z(), FloatC(-1.f)), FloatC(1.f)));
        FloatC ph = atan2(p.y(), p.x());
        FloatC uu = fmadd(ph, FloatC(.5f / pi), FloatC(.5f)), vv = theta * (1.f / pi);
        UInt32C ix = enoki::min(UInt32C(uu * float(width)), UInt32C(width - 1)), iy = enoki::min(UInt32C(vv * float(width)), UInt32C(width - 1));
        Float albedo = tex(iy * width + ix);                                   // the only differentiable operation
        radiance = radiance + throughput * albedo * .1f;                       // the surface emits a little of its colour
        throughput = throughput * albedo;
        // cosine-weighted bounce around the inward normal (concentric disk mapping)
        Vector3 nrm = p * FloatC(-1.f);
        FloatC r1 = 2.f * rng.next_float32() - 1.f, r2 = 2.f * rng.next_float32() - 1.f;
        auto swap = abs(r1) < abs(r2);
        FloatC rad = select(swap, r2, r1);
        FloatC ratio = select(swap, r1, r2) / select(eq(rad, FloatC(0.f)), FloatC(1.f), rad);
        FloatC ang = select(swap, (.5f * pi) - (.25f * pi) * ratio, (.25f * pi) * ratio);
        auto [sn, cs] = sincos(ang);
        FloatC dx = rad * cs, dy = rad * sn;
        FloatC dz = sqrt(enoki::max(FloatC(0.f), 1.f - dx * dx - dy * dy));
        // orthonormal frame around nrm (Duff et al. 2017)
        FloatC sign = copysign(FloatC(1.f), nrm.z());
        FloatC a = FloatC(-1.f) / (sign + nrm.z());
        FloatC bb = nrm.x() * nrm.y() * a;
        Vector3 sx(1.f + sign * nrm.x() * nrm.x() * a, sign * bb, FloatC(-1.f) * sign * nrm.x());
        Vector3 ty(bb, sign + nrm.y() * nrm.y() * a, FloatC(-1.f) * nrm.y());
        d = unit(sx * dx + ty * dy + nrm * dz);
        o = p + nrm * FloatC(1e-3f);
    }
    return radiance + throughput;                                              // leftover energy reaches a white environment
}

} // namespace cfg5
