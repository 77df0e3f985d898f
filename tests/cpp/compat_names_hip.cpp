// A program written with the reference's names -- <enoki/cuda.h>, CUDAArray<float>, DiffArray<CUDAArray<float>>, cuda_eval(),
// <enoki/dynamic.h>, DynamicArray<Packet<float>> -- compiled against this repository's headers without an edit
// (compat/enoki/cuda.h, compat/enoki/dynamic.h: the opt-in include root, -I compat).  Run by tests/test_reference_sources_gpu.py.
#include <enoki/cuda.h>
#include <enoki/dynamic.h>
#include <enoki/autodiff.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <vector>

using namespace enoki;

using FloatC = CUDAArray<float>;
using FloatD = DiffArray<FloatC>;
using UInt32D = DiffArray<CUDAArray<uint32_t>>;
using FloatX = DynamicArray<Packet<float>>;

static_assert(is_dynamic_array_v<FloatC> && array_size_v<FloatC> == Dynamic && is_dynamic_array_v<FloatD> && !is_static_array_v<FloatD>);
static_assert(is_diff_array<FloatD>::value && !is_diff_array<FloatC>::value && is_static_array_v<Array<FloatC, 3>> && array_size_v<Array<FloatC, 3>> == 3);
static_assert(std::is_same_v<bool_array_t<FloatC>, mask_t<FloatC>> && std::is_same_v<float_array_t<UInt32D>, FloatD>);

int main() {
    const size_t n = 1 << 16, K = 1024;
    FloatD table = linspace<FloatD>(0.f, 1.f, K);
    set_requires_gradient(table);
    UInt32D idx = arange<UInt32D>(n) & UInt32D(uint32_t(K - 1));
    FloatD x = linspace<FloatD>(-1.f, 1.f, n);
    FloatD y = hsum(sin(gather<FloatD>(table, idx) * x));
    cuda_eval();                                    // nothing to flush: accepted and ignored
    backward(y);
    FloatC g = gradient(table);
    cuda_device_sync();
    // analytic gradient of bin k: sum over the elements that read it of cos(t_k x_i) x_i
    double worst = 0;
    for (size_t k = 0; k < K; k += 97) {
        double want = 0, tk = (double) k / (K - 1);
        for (size_t i = k; i < n; i += K) { double xi = -1.0 + 2.0 * (double) i / (n - 1); want += std::cos(tk * xi) * xi; }
        worst = std::fmax(worst, std::fabs(want - (double) g.coeff(k)));
    }
    // generic code of the reference's users: reductions with a device default, inner / nested reductions, small helpers
    using Vector3fC = Array<FloatC, 3>;
    FloatC t = linspace<FloatC>(1.f, 4.f, 4);
    Vector3fC v(t, t * 2.f, FloatC(-1.f));
    const bool skipped = any_or<true>(t > 100.f) && !all_or<false>(t > 0.f) && none_or<true>(t > 0.f);    // no evaluation, no sync
    auto inner = hsum_inner(v);                      // one sum per component
    const bool helpers = skipped && inner.x().coeff(0) == 10.f && inner.y().coeff(0) == 20.f && inner.z().coeff(0) == -1.f &&
                         hmean(t).coeff(0) == 2.5f && hmax_nested(v).coeff(0) == 8.f &&
                         abs_dot(v, Vector3fC(FloatC(-1.f), FloatC(0.f), FloatC(2.f))).coeff(3) == 6.f &&
                         std::fabs(rad_to_deg(FloatC(3.14159265f)).coeff(0) - 180.f) < 1e-3f &&
                         copysign_neg(t, t).coeff(1) == -2.f &&
                         next_float(t).coeff(0) == std::nextafter(1.f, 2.f) && prev_float(t).coeff(3) == std::nextafter(4.f, 0.f) &&
                         !any(isdenormal(t)) && count(isdenormal(t * 1e-39f)) == 4 &&
                         std::fabs(unit_angle_z(normalize(Vector3fC(t, FloatC(0.f), FloatC(1.f)))).coeff(0) - 0.78539816f) < 1e-6f &&
                         std::fabs(unit_angle(Vector3fC(FloatC(0.f), FloatC(0.f), FloatC(1.f)),
                                              Vector3fC(FloatC(0.f), FloatC(1.f), FloatC(0.f))).coeff(0) - 1.5707964f) < 1e-6f;
    // a per-lane binary search over a sorted device table (array_utils.h:130-171)
    using UInt32C = CUDAArray<uint32_t>;
    FloatC sorted = linspace<FloatC>(0.f, 99.f, 100);           // sorted[i] = i
    FloatC needles = FloatC::copy(std::vector<float>{ -5.f, 0.5f, 42.f, 98.5f, 1000.f }.data(), 5);
    UInt32C lower = binary_search(0u, 100u, [&](const UInt32C &i) { return gather<FloatC>(sorted, min(i, UInt32C(99u))) < needles; });
    const bool searched = lower.coeff(0) == 0 && lower.coeff(1) == 1 && lower.coeff(2) == 42 && lower.coeff(3) == 99 && lower.coeff(4) == 100 &&
                          log2i(UInt32C(1000u)).coeff(0) == 9 && scalar_cast(hsum(t)) == 10.f && sr<2>(UInt32C(64u)).coeff(0) == 16;
    // shape / ragged / set_shape of a structure of arrays, division by a run-time constant
    Vector3fC soa(t, t, FloatC(1.f));
    auto shp = shape(soa);
    bool shaped = shp[0] == 3 && shp[1] == 4 && ragged(soa);           // the broadcast component has length 1
    set_shape(soa, { 3, 4 });
    shaped = shaped && !ragged(soa) && soa.z().size() == 4 && soa.z().coeff(3) == 1.f &&
             (UInt32C(100u) / divisor<uint32_t>(7u)).coeff(0) == 14u;
    // gradient-free members on differentiable arrays: texel coordinates, bit rotations, first active entry
    FloatD coord = linspace<FloatD>(0.25f, 7.75f, 4);
    UInt32D texel = floor2int<UInt32D>(coord), upper = ceil2int<UInt32D>(coord);
    const bool casts = texel.coeff(1) == 2 && upper.coeff(1) == 3 && rol(texel, UInt32D(31u)).coeff(1) == 1 &&
                       extract(coord, coord > 3.f) == 5.25f && andnot(texel, UInt32D(2u)).coeff(1) == 0;
    // printing (array_base.h:190-237): one row per slice, long arrays abbreviated
    std::ostringstream small, large;
    small << Vector3fC(FloatC::copy(std::vector<float>{ 1.f, 2.f }.data(), 2), FloatC::copy(std::vector<float>{ 3.f, 4.f }.data(), 2),
                       FloatC::copy(std::vector<float>{ 5.f, 6.f }.data(), 2));
    large << arange<UInt32C>(100);
    const bool printed = small.str() == "[[1, 3, 5],\n [2, 4, 6]]" &&
                         large.str() == "[0, 1, 2, 3, 4, .. 90 skipped .., 95, 96, 97, 98, 99]";
    Array<float, 3> third = slice(Vector3fC(t, t * 2.f, FloatC(-1.f)), 2);                  // (x_2, y_2, z_2)
    const bool sliced = third.x() == 3.f && third.y() == 6.f && third.z() == -1.f && slice(t, 3) == 4.f;
    FloatX z = zero<FloatX>(8) + 1.f;
    char *w = cuda_whos();
    const bool ok = worst < 2e-3 && hsum(z).coeff(0) == 8.f && w != nullptr && helpers && searched && shaped && casts && printed && sliced;
    free(w);
    if (!ok)
        printf("  helpers %d searched %d shaped %d casts %d printed %d  [%s] [%s]\n", (int) helpers, (int) searched, (int) shaped,
               (int) casts, (int) (printed && sliced), small.str().c_str(), large.str().c_str());
    printf("compat names: max gradient error %.2e -> %s\n", worst, ok ? "ok" : "FAILED");
    return ok ? 0 : 1;
}
