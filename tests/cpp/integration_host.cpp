// integration/enoki/hip.h + integration/hip_hooks.cpp (the reference-side binding) on the host stand-in of the C ABI, under
// AddressSanitizer / UBSan, against the reference's own headers: the bookkeeping that lets the reference's autodiff layer
// run fast without an edit --
//   * the trace fragments of safe_mul / safe_fmadd (autodiff.cpp:1191-1221) become ONE fused call when the select's
//     operand is the product of exactly the two tested arrays, and run literally (compare / or / select) otherwise: other
//     operands, an operand written to in between, masks that are read by something else;
//   * a 64-bit index array hands the library the 32-bit array it was widened from while that is alive and unwritten.
// Built only where /root/reference exists (enoki_amd/_build.py); run by tests/test_host_sanitizers.py.
#include <enoki/hip.h>              // integration/enoki/hip.h (the include path puts integration/ first)

#include "host_abi_stub.h"
#include "../../integration/hip_hooks.cpp"

#include <cmath>
#include <vector>

using namespace enoki;
using FloatH = HIPArray<float>;
using MaskH = mask_t<FloatH>;
using UInt32H = HIPArray<uint32_t>;
using Int64H = HIPArray<int64_t>;

#define CHECK(expr) do { if (!(expr)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #expr); exit(1); } } while (0)

static FloatH make(std::initializer_list<float> v) { std::vector<float> h(v); return FloatH::copy(h.data(), h.size()); }

/// the reference's spelling (autodiff.cpp:1198-1202 / 1214-1218)
static FloatH spelled_safe_mul(const FloatH &v1, const FloatH &v2, const FloatH &tentative) {
    MaskH m1 = MaskH::from_index_(cuda_trace_append(EnokiType::Bool, "setp.eq.f32 $r1, $r2, 0.0", v1.index_())),
          m2 = MaskH::from_index_(cuda_trace_append(EnokiType::Bool, "setp.eq.or.f32 $r1, $r2, 0.0, $r3", v2.index_(), m1.index_()));
    return FloatH::from_index_(cuda_trace_append(FloatH::Type, "selp.$t1 $r1, 0.0, $r2, $r3", tentative.index_(), m2.index_()));
}
static FloatH spelled_safe_fmadd(const FloatH &v1, const FloatH &v2, const FloatH &v3, const FloatH &tentative) {
    MaskH m1 = MaskH::from_index_(cuda_trace_append(EnokiType::Bool, "setp.eq.f32 $r1, $r2, 0.0", v1.index_())),
          m2 = MaskH::from_index_(cuda_trace_append(EnokiType::Bool, "setp.eq.or.f32 $r1, $r2, 0.0, $r3", v2.index_(), m1.index_()));
    return FloatH::from_index_(cuda_trace_append(FloatH::Type, "selp.$t1 $r1, $r2, $r3, $r4", v3.index_(), tentative.index_(), m2.index_()));
}
static void expect(const FloatH &got, std::initializer_list<float> want) {
    CHECK(got.size() == want.size());
    size_t i = 0;
    for (float w : want) { float g = got.coeff(i++); CHECK((std::isnan(w) && std::isnan(g)) || g == w); }
}

int main() {
    const float inf = INFINITY;
    {   // the pattern: one fused call, the reference's result (0 * inf = 0, not NaN)
        FloatH a = make({ 0.f, 2.f, inf, -3.f }), b = make({ inf, 0.f, 0.f, 4.f });
        long before = g_safe_calls;
        FloatH r = spelled_safe_mul(a, b, a * b);
        CHECK(g_safe_calls == before + 4);
        expect(r, { 0.f, 0.f, 0.f, -12.f });
        FloatH c = make({ 1.f, 1.f, 1.f, 1.f });
        before = g_safe_calls;
        FloatH f = spelled_safe_fmadd(a, b, c, fmadd(a, b, c));
        CHECK(g_safe_calls == before + 4);
        expect(f, { 1.f, 1.f, 1.f, -11.f });
        before = g_safe_calls;
        FloatH z(0.f);
        expect(spelled_safe_mul(z, b, z * b), { 0.f, 0.f, 0.f, 0.f });                          // a broadcast factor
        CHECK(g_safe_calls == before + 4);
        before = g_safe_calls;
        expect(spelled_safe_mul(FloatH(0.f), b, FloatH(0.f) * b), { 0.f, 0.f, 0.f, 0.f });     // equal values, different arrays: literal
        CHECK(g_safe_calls == before);
    }
    {   // NOT the pattern: the select's operand is some other array -> literal evaluation, literal result
        FloatH a = make({ 0.f, 2.f, 5.f }), b = make({ 7.f, 0.f, 3.f }), other = make({ 9.f, 9.f, 9.f });
        long before = g_safe_calls;
        expect(spelled_safe_mul(a, b, other), { 0.f, 0.f, 9.f });
        expect(spelled_safe_mul(a, b, b * a + 0.f), { 0.f, 0.f, 15.f });                        // a sum, not the tagged product
        expect(spelled_safe_fmadd(a, b, other, fmadd(a, b, b)), { 9.f, 9.f, 18.f });            // tentative built from another addend
        CHECK(g_safe_calls == before);
    }
    {   // an operand is written to between the product and the fragments: the tag has expired
        FloatH a = make({ 1.f, 2.f }), b = make({ 3.f, 4.f });
        FloatH tentative = a * b;
        a.data()[0] = 0.f;                                        // mutable pointer: host memory under the stand-in
        long before = g_safe_calls;
        expect(spelled_safe_mul(a, b, tentative), { 0.f, 8.f });   // mask from the NEW a, values from the OLD product
        CHECK(g_safe_calls == before);
    }
    {   // masks that are read by something else are evaluated from their recipes
        FloatH a = make({ 0.f, 2.f, 0.f }), b = make({ 1.f, 0.f, 0.f });
        MaskH m1 = MaskH::from_index_(cuda_trace_append(EnokiType::Bool, "setp.eq.f32 $r1, $r2, 0.0", a.index_()));
        MaskH m2 = MaskH::from_index_(cuda_trace_append(EnokiType::Bool, "setp.eq.or.f32 $r1, $r2, 0.0, $r3", b.index_(), m1.index_()));
        CHECK(count(m2) == 3 && count(m1) == 2 && m1.coeff(1) == false && m2.coeff(1) == true);
        MaskH real = a > 1.f;                                     // an ordinary mask as the third operand: literal path
        MaskH m3 = MaskH::from_index_(cuda_trace_append(EnokiType::Bool, "setp.eq.or.f32 $r1, $r2, 0.0, $r3", b.index_(), real.index_()));
        CHECK(count(m3) == 2 && m3.coeff(0) == false);
    }
    {   // widened index arrays
        std::vector<uint32_t> host = { 3, 1, 2, 0 };
        UInt32H idx = UInt32H::copy(host.data(), host.size());
        Int64H wide(idx);
        int code = 0;
        ek_operand op = wide.index_operand(code);
        CHECK(code == EK_U32 && op.ptr == (const void *) ((const UInt32H &) idx).data() && wide.coeff(0) == 3);
        FloatH table = make({ 10.f, 11.f, 12.f, 13.f });
        expect(gather<FloatH>(table, wide), { 13.f, 11.f, 12.f, 10.f });          // the library sees the 32-bit original
        FloatH target = make({ 0.f, 0.f, 0.f, 0.f });
        scatter_add(target, make({ 1.f, 2.f, 3.f, 4.f }), wide);
        expect(target, { 4.f, 2.f, 3.f, 1.f });
        idx.data();                                               // a mutable pointer was handed out: the tag expires
        op = wide.index_operand(code);
        CHECK(code == EK_I64 && op.ptr == (const void *) ((const Int64H &) wide).data());
        {
            Int64H orphan;
            { UInt32H temp = UInt32H::copy(host.data(), host.size()); orphan = Int64H(temp); }   // the original is gone
            op = orphan.index_operand(code);
            CHECK(code == EK_I64 && orphan.coeff(3) == 0);
        }
    }
    for (auto &slot : hip_detail::Handles::get().slot) slot.reset();      // the ring of recently named buffers (bounded, by design)
    CHECK(g_live.empty());
    printf("integration_host: safe_mul / safe_fmadd fragments fused exactly when tagged, literal otherwise; widened index arrays "
           "traced to their 32-bit origin while valid; no block left allocated\n");
    return 0;
}
