// The hooks of the reference's JIT backend that its generic code calls on "CUDA" arrays, for an EAGER backend:
//   * cuda_eval / cuda_eval_var / cuda_sync / cuda_var_mark_dirty / cuda_set_scatter_gather_operand / cuda_var_set_label (cuda.h:33-63, 181;
//     callers: array_struct.h scatter / gather wrappers, array_router.h any_or, array_macro.h): nothing is pending, nothing
//     to mark -- no-ops;
//   * cuda_register_callback / cuda_unregister_callback (Tape's constructor, autodiff.cpp:218-219): there is no cuda_eval()
//     to call back from -- no-ops;
//   * cuda_trace_append for the FOUR trace fragments with which autodiff.cpp:1198-1218 spells safe_mul / safe_fmadd
//     ((w == 0 || g == 0) ? 0 : w * g and its fma form): recognised as a whole and executed as ONE fused kernel when the
//     operands match (see below), otherwise literally as compare / or / select kernels on the buffers that
//     HIPArray::index_() parked (integration/enoki/hip.h).  Any other fragment is an error.
// A maintainer would compile this file into the library that ships integration/enoki/hip.h; the reference tree itself is
// untouched.
#include <enoki/hip.h>

#include <cstring>

NAMESPACE_BEGIN(enoki)

void cuda_eval(bool) { }
void cuda_eval_var(uint32_t, bool) { }
void cuda_sync() { ek_hip_sync(); }
void cuda_var_mark_dirty(uint32_t) { }
void cuda_set_scatter_gather_operand(uint32_t index, bool) {
    // array_struct.h:25-37, 69-84, 103-118: announces the array behind the pointer of the next indexed operation (0: done)
    hip_detail::Operand &o = hip_detail::announced_operand();
    if (index == 0) { o = hip_detail::Operand(); return; }
    auto b = hip_detail::Handles::get().find(index);
    o.ptr = b ? b->get() : nullptr;
    o.size = b ? b->size : 0;
}
void cuda_var_set_label(uint32_t, const char *) { }          // set_label() on device arrays (cuda.h:956-964): no trace to label
void cuda_register_callback(void (*)(void *), void *) { }
void cuda_unregister_callback(void (*)(void *), void *) { }

namespace {
using hip_detail::Buffer;
using hip_detail::Handles;

std::shared_ptr<Buffer> make(size_t size, size_t elem) {
    auto b = std::make_shared<Buffer>();
    b->size = size;
    hip_detail::check(ek_hip_malloc((size ? size : 1) * elem, &b->ptr), "trace fragment");
    return b;
}
ek_operand op(const std::shared_ptr<Buffer> &b) { return ek_operand{ b->get(), 0, b->size }; }
size_t bsize(size_t a, size_t b) { return a == 1 ? b : a; }
/// The fragments below are the float32 spellings of safe_mul / safe_fmadd (autodiff.cpp:1191-1221) and are executed by float32
/// kernels on 4-byte buffers: any other element type is refused loudly instead of being reinterpreted.
void require_f32(EnokiType type, const char *fragment) {
    if (type != EnokiType::Float32 && type != EnokiType::Bool)
        throw std::runtime_error(std::string("enoki-hip integration: trace fragment \"") + fragment +
                                 "\" on a non-float32 array is not provided (Tape<double> is not wired through the integration layer)");
}

[[noreturn]] void unknown(const char *fragment) {
    throw std::runtime_error(std::string("integration/hip_hooks.cpp: trace fragment not provided by the eager backend: ") + fragment);
}
} // namespace

// The three fragments of safe_mul(value1, value2) arrive as: m1 = (value1 == 0); m2 = (value2 == 0) | m1;
// result = m2 ? 0 : tentative, with tentative = value1 * value2 computed by the router just before (autodiff.cpp:1192-1202;
// safe_fmadd: 1207-1218 with tentative = fmadd(value1, value2, value3) and result = m2 ? value3 : tentative).  The masks
// are kept as recipes (Buffer::lazy); when the final select finds that its `tentative` operand is tagged as the product of
// exactly the two arrays the mask tests (Buffer::prod, set by mul_ / fmadd_), ONE fused kernel (EK_SAFE_MUL / EK_SAFE_FMADD
// of the C ABI, the same arithmetic) replaces compare + compare + or + select.  Any other use of the fragments evaluates
// them literally, as before.
namespace {
std::shared_ptr<Buffer> recipe(int kind, size_t size, const std::shared_ptr<Buffer> &a, const std::shared_ptr<Buffer> &b) {
    auto m = std::make_shared<Buffer>();
    m->size = size;
    m->lazy_kind = kind;
    m->lazy[0] = a;
    m->lazy[1] = b;
    return m;
}
/// does `mask` test exactly the two factors that `tentative` was computed from?
bool tests_factors_of(const std::shared_ptr<Buffer> &mask, const std::shared_ptr<Buffer> &tentative, int kind) {
    if (mask->lazy_kind != 2 || tentative->prod_kind != kind) return false;
    auto f0 = tentative->prod[0].lock(), f1 = tentative->prod[1].lock();
    Buffer *p0 = f0.get(), *p1 = f1.get(), *l0 = mask->lazy[0].get(), *l1 = mask->lazy[1].get();
    if (!p0 || !p1 || p0->version != tentative->prod_version[0] || p1->version != tentative->prod_version[1])
        return false;                                                            // an operand was written to since
    return (p0 == l0 && p1 == l1) || (p0 == l1 && p1 == l0);
}
} // namespace

uint32_t cuda_trace_append(EnokiType type, const char *fragment, uint32_t i1) {
    require_f32(type, fragment);
    if (strcmp(fragment, "setp.eq.f32 $r1, $r2, 0.0") != 0) unknown(fragment);
    auto v = Handles::get().find(i1);
    return Handles::get().park(recipe(1, v->size, v, nullptr));                  // (v == 0), not evaluated yet
}

uint32_t cuda_trace_append(EnokiType type, const char *fragment, uint32_t i1, uint32_t i2) {
    require_f32(type, fragment);
    auto x = Handles::get().find(i1), y = Handles::get().find(i2);
    if (strcmp(fragment, "setp.eq.or.f32 $r1, $r2, 0.0, $r3") == 0) {          // (x == 0) | y
        const size_t n = bsize(x->size, y->size);
        if (y->lazy_kind == 1)                                                   // y = (v1 == 0): keep (v1 == 0) | (x == 0) as a recipe
            return Handles::get().park(recipe(2, n, y->lazy[0], x));
        auto e = make(x->size, 1), m = make(n, 1);
        ek_operand a = op(x), zero{ nullptr, 0, 1 };
        hip_detail::check(ek_hip_compare(EK_EQ, EK_F32, (uint8_t *) e->ptr, &a, &zero, x->size), "setp.eq.or");
        ek_operand oe = op(e), oy = op(y);
        hip_detail::check(ek_hip_binary(EK_OR, EK_BOOL, m->ptr, &oe, &oy, n), "setp.eq.or");
        return Handles::get().park(m);
    }
    if (strcmp(fragment, "selp.$t1 $r1, 0.0, $r2, $r3") == 0) {                // y ? 0 : x
        const size_t n = bsize(x->size, y->size);
        auto r = make(n, 4);
        if (tests_factors_of(y, x, 1)) {                                         // safe_mul: one fused kernel
            ek_operand oa = op(y->lazy[0]), ob = op(y->lazy[1]);
            hip_detail::check(ek_hip_binary(EK_SAFE_MUL, EK_F32, r->ptr, &oa, &ob, n), "safe_mul");
            return Handles::get().park(r);
        }
        ek_operand om = op(y), zero{ nullptr, 0, 1 }, ox = op(x);
        hip_detail::check(ek_hip_select(EK_F32, r->ptr, &om, &zero, &ox, n), "selp");
        return Handles::get().park(r);
    }
    unknown(fragment);
}

uint32_t cuda_trace_append(EnokiType type, const char *fragment, uint32_t i1, uint32_t i2, uint32_t i3) {
    require_f32(type, fragment);
    if (strcmp(fragment, "selp.$t1 $r1, $r2, $r3, $r4") != 0) unknown(fragment);   // m ? x : y
    auto x = Handles::get().find(i1), y = Handles::get().find(i2), m = Handles::get().find(i3);
    const size_t n = bsize(bsize(x->size, y->size), m->size);
    auto r = make(n, 4);
    if (tests_factors_of(m, y, 2) && y->prod[2].lock().get() == x.get() && x->version == y->prod_version[2]) {   // safe_fmadd
        auto f0 = y->prod[0].lock(), f1 = y->prod[1].lock();
        ek_operand oa = op(f0), ob = op(f1), oc = op(x);
        hip_detail::check(ek_hip_ternary(EK_SAFE_FMADD, EK_F32, r->ptr, &oa, &ob, &oc, n), "safe_fmadd");
        return Handles::get().park(r);
    }
    ek_operand om = op(m), ox = op(x), oy = op(y);
    hip_detail::check(ek_hip_select(EK_F32, r->ptr, &om, &ox, &oy, n), "selp");
    return Handles::get().park(r);
}

NAMESPACE_END(enoki)
