// Host stand-in for the C ABI of libenoki-hip.so (malloc'd "device" memory, scalar loops; float32 values, uint32 / int64
// indices, u8 masks): just the entry points the sanitizer checkers reach, so that the HOST logic of the binding (deferred
// nodes: asan_deferred.cpp; the tape on top of them: asan_tape.cpp) runs under ASan / LSan / UBSan without a GPU.  A
// checker, never linked into the product.  Include once per binary, after <enoki/hip.h>.
#pragma once

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#define STUB_UNSUPPORTED(what, a, b) (fprintf(stderr, "stand-in: %s not provided (%d, %d)\n", what, (int) (a), (int) (b)), abort())

// ------------------------------------------------------------------------------------------------------------------
//  Host stand-in for the C ABI (float32 values, uint32 indices, u8 masks)
// ------------------------------------------------------------------------------------------------------------------
static std::map<void *, size_t> g_live;
static long g_unary_calls = 0, g_sincos_calls = 0, g_gather_calls = 0, g_fused_calls = 0;
static const char *g_error = "";

static float op_f(const ek_operand *o, size_t i) {
    if (!o->ptr) { float v; uint32_t b = (uint32_t) o->imm; memcpy(&v, &b, 4); return v; }
    return ((const float *) o->ptr)[o->size == 1 ? 0 : i];
}
static uint32_t op_u(const ek_operand *o, size_t i) {
    if (!o->ptr) return (uint32_t) o->imm;
    return ((const uint32_t *) o->ptr)[o->size == 1 ? 0 : i];
}
static bool op_m(const ek_operand *o, size_t i) {
    if (!o->ptr) return (o->imm & 1) != 0;
    return ((const uint8_t *) o->ptr)[o->size == 1 ? 0 : i] != 0;
}
static float unary_f(int op, float x) {
    switch (op) {
        case EK_NEG: return -x;
        case EK_ABS: return std::fabs(x);
        case EK_SQRT: return std::sqrt(x);
        case EK_RCP: return 1.0f / x;
        case EK_RSQRT: return 1.0f / std::sqrt(x);
        case EK_RCP_SQR: { volatile float r = 1.0f / x; return r * r; }
        case EK_RSQRT_SQR: { volatile float r = 1.0f / std::sqrt(x); return r * r; }
        case EK_RSQRT_CUBE: { volatile float r = 1.0f / std::sqrt(x); volatile float r2 = r * r; return r * r2; }
        case EK_SIN: return std::sin(x);
        case EK_COS: return std::cos(x);
        case EK_EXP: return std::exp(x);
        case EK_LOG: return std::log(x);
        case EK_FLOOR: return std::floor(x);
        case EK_SIGN: return std::copysign(1.0f, x);
        case EK_COPY: return x;
        default: fprintf(stderr, "stand-in: unary op %d\n", op); abort();
    }
}

extern "C" {
const char *ek_hip_last_error(void) { return g_error; }
void **ek_hip_binding_slot(void) { static void *slot = nullptr; return &slot; }
uint32_t ek_hip_log_level(void) { return 0; }
int ek_hip_malloc(size_t bytes, void **out) {
    *out = malloc(bytes ? bytes : 1);
    g_live[*out] = bytes;
    return EK_OK;
}
int ek_hip_free(void *p) {
    if (!p) return EK_OK;
    if (!g_live.erase(p)) { fprintf(stderr, "stand-in: free of an unknown block\n"); abort(); }
    free(p);
    return EK_OK;
}
int ek_hip_sync(void) { return EK_OK; }
int ek_hip_memcpy_device(void *d, const void *s, size_t b) { memcpy(d, s, b); return EK_OK; }
int ek_hip_memcpy_to_host(void *d, const void *s, size_t b) { memcpy(d, s, b); return EK_OK; }
int ek_hip_memcpy_to_device(void *d, const void *s, size_t b) { memcpy(d, s, b); return EK_OK; }
int ek_hip_memset(void *d, int v, size_t b) { memset(d, v, b); return EK_OK; }
int ek_hip_fill(int type, void *out, uint64_t bits, size_t n) {
    const size_t w = type == EK_BOOL ? 1 : (type == EK_I64 || type == EK_U64 || type == EK_F64) ? 8 : 4;
    for (size_t i = 0; i < n; ++i) memcpy((char *) out + i * w, &bits, w);
    return EK_OK;
}
int ek_hip_arange(int type, void *out, int64_t start, int64_t step, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        if (type == EK_F32) ((float *) out)[i] = (float) (start + (int64_t) i * step);
        else ((uint32_t *) out)[i] = (uint32_t) (start + (int64_t) i * step);
    }
    return EK_OK;
}
int ek_hip_linspace(int, void *out, double lo, double hi, size_t n) {
    for (size_t i = 0; i < n; ++i) ((float *) out)[i] = (float) (lo + (hi - lo) * (double) i / (double) (n > 1 ? n - 1 : 1));
    return EK_OK;
}
int ek_hip_unary(int op, int type, void *out, const ek_operand *a, size_t n) {
    ++g_unary_calls;
    if (type == EK_F32) for (size_t i = 0; i < n; ++i) ((float *) out)[i] = unary_f(op, op_f(a, i));
    else if (op == EK_COPY && type == EK_U32) for (size_t i = 0; i < n; ++i) ((uint32_t *) out)[i] = op_u(a, i);
    else if (op == EK_COPY && type == EK_BOOL) for (size_t i = 0; i < n; ++i) ((uint8_t *) out)[i] = op_m(a, i);
    else if (op == EK_NOT && type == EK_BOOL) for (size_t i = 0; i < n; ++i) ((uint8_t *) out)[i] = !op_m(a, i);
    else STUB_UNSUPPORTED("unary", op, type);
    return EK_OK;
}
int ek_hip_sincos(int, void *s, void *c, const ek_operand *a, size_t n) {
    ++g_sincos_calls;
    for (size_t i = 0; i < n; ++i) { float x = op_f(a, i); ((float *) s)[i] = std::sin(x); ((float *) c)[i] = std::cos(x); }
    return EK_OK;
}
static long g_safe_calls = 0;               // elements that went through EK_SAFE_MUL / EK_SAFE_FMADD
static long g_array_launches = 0;           // ek_hip_unary / binary / ternary calls over more than one element
static float binary_f(int op, float a, float b) {
    switch (op) {
        case EK_ADD: return a + b;
        case EK_SUB: return a - b;
        case EK_MUL: return a * b;
        case EK_DIV: return a / b;
        case EK_MIN: return b < a ? b : a;
        case EK_MAX: return b > a ? b : a;
        case EK_SAFE_MUL: ++g_safe_calls; return (a == 0 || b == 0) ? 0.f : a * b;
        default: fprintf(stderr, "stand-in: binary op %d\n", op); abort();
    }
}
int ek_hip_binary(int op, int type, void *out, const ek_operand *a, const ek_operand *b, size_t n) {
    g_array_launches += n > 1;
    if (type == EK_BOOL) {
        for (size_t i = 0; i < n; ++i) {
            bool x = op_m(a, i), y = op_m(b, i);
            ((uint8_t *) out)[i] = op == EK_AND ? (x && y) : op == EK_OR ? (x || y) : op == EK_XOR ? (x != y) : (STUB_UNSUPPORTED("mask binary", op, 0), 0);
        }
        return EK_OK;
    }
    if (type == EK_U32) {
        for (size_t i = 0; i < n; ++i) {
            uint32_t x = op_u(a, i), y = op_u(b, i);
            ((uint32_t *) out)[i] = op == EK_AND ? (x & y) : op == EK_MUL ? x * y : op == EK_ADD ? x + y : op == EK_SR ? (y < 32 ? x >> y : 0u) : op == EK_SL ? (y < 32 ? x << y : 0u) :
                                    op == EK_MOD ? x % y : op == EK_SUB ? x - y : (STUB_UNSUPPORTED("u32 binary", op, 0), 0u);
        }
        return EK_OK;
    }
    for (size_t i = 0; i < n; ++i) ((float *) out)[i] = binary_f(op, op_f(a, i), op_f(b, i));
    return EK_OK;
}
int ek_hip_ternary(int op, int, void *out, const ek_operand *a, const ek_operand *b, const ek_operand *c, size_t n) {
    g_array_launches += n > 1;
    for (size_t i = 0; i < n; ++i) {
        const float x = op_f(a, i), y = op_f(b, i), z = op_f(c, i);
        float r;
        switch (op) {
            case EK_FMADD: r = std::fma(x, y, z); break;
            case EK_FMSUB: r = std::fma(x, y, -z); break;
            case EK_FNMADD: r = std::fma(-x, y, z); break;
            case EK_FNMSUB: r = std::fma(-x, y, -z); break;
            case EK_SAFE_FMADD: ++g_safe_calls; r = (x == 0 || y == 0) ? z : std::fma(x, y, z); break;
            case EK_MULADD: { volatile float p = x * y; r = p + z; break; }          // a product and a sum, a rounding each
            case EK_MULSUB: { volatile float p = x * y; r = p - z; break; }
            case EK_NMULADD: { volatile float p = x * y; r = z - p; break; }
            default: STUB_UNSUPPORTED("ternary", op, 0);
        }
        ((float *) out)[i] = r;
    }
    return EK_OK;
}
int ek_hip_hsum_safe_mul(int, void *out, const ek_operand *w, const ek_operand *g, size_t n) {
    float acc = 0.f;
    for (size_t i = 0; i < n; ++i) acc += binary_f(EK_SAFE_MUL, op_f(w, i), op_f(g, i));
    *(float *) out = acc;
    return EK_OK;
}
int ek_hip_psum(int, void *out, const void *in, size_t n) {
    float acc = 0.f;
    for (size_t i = 0; i < n; ++i) { acc += ((const float *) in)[i]; ((float *) out)[i] = acc; }
    return EK_OK;
}
int ek_hip_reverse(int, void *out, const void *in, size_t n) {
    for (size_t i = 0; i < n; ++i) ((float *) out)[i] = ((const float *) in)[n - 1 - i];
    return EK_OK;
}
int ek_hip_mask_reduce(int op, const uint8_t *mask, size_t n, uint64_t *result) {
    uint64_t count = 0;
    for (size_t i = 0; i < n; ++i) count += mask[i] ? 1 : 0;
    *result = op == EK_ALL ? (count == n) : op == EK_ANY ? (count != 0) : count;
    return EK_OK;
}
int ek_hip_cast(int src, int dst, void *out, const ek_operand *a, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        if (src == EK_U32 && dst == EK_F32) ((float *) out)[i] = (float) op_u(a, i);
        else if (src == EK_F32 && dst == EK_U32) ((uint32_t *) out)[i] = (uint32_t) op_f(a, i);
        else if (src == EK_U32 && dst == EK_I64) ((int64_t *) out)[i] = (int64_t) op_u(a, i);
        else if (src == EK_U32 && dst == EK_U64) ((uint64_t *) out)[i] = (uint64_t) op_u(a, i);
        else STUB_UNSUPPORTED("cast", src, dst);
    }
    return EK_OK;
}
int ek_hip_select(int, void *out, const ek_operand *m, const ek_operand *t, const ek_operand *f, size_t n) {
    for (size_t i = 0; i < n; ++i) ((float *) out)[i] = op_m(m, i) ? op_f(t, i) : op_f(f, i);
    return EK_OK;
}
int ek_hip_compare(int op, int type, uint8_t *out, const ek_operand *a, const ek_operand *b, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        bool r;
        if (type == EK_U32) {
            const uint32_t x = op_u(a, i), y = op_u(b, i);
            r = op == EK_EQ ? x == y : op == EK_NEQ ? x != y : op == EK_LT ? x < y : op == EK_LE ? x <= y : op == EK_GT ? x > y : x >= y;
        } else if (type == EK_F32) {
            const float x = op_f(a, i), y = op_f(b, i);
            r = op == EK_EQ ? x == y : op == EK_NEQ ? x != y : op == EK_LT ? x < y : op == EK_LE ? x <= y : op == EK_GT ? x > y : x >= y;
        } else STUB_UNSUPPORTED("compare", op, type);
        out[i] = r;
    }
    return EK_OK;
}
int ek_hip_gather(int, int, void *out, const void *base, const ek_operand *index, const ek_operand *mask, size_t n) {
    ++g_gather_calls;
    for (size_t i = 0; i < n; ++i) ((float *) out)[i] = op_m(mask, i) ? ((const float *) base)[op_u(index, i)] : 0.f;
    return EK_OK;
}
// struct gathers: the plan is "records" for every table of 64+ entries, so that the binding's record paths (HIPArray::
// gather_records_, DiffArray::gather_multi_) run on the host stand-in
static long g_record_gathers = 0;
int ek_hip_gather_multi(int, int, int count, void *const *outs, const void *const *bases, const ek_operand *index,
                        const ek_operand *mask, size_t n) {
    for (int c = 0; c < count; ++c)
        for (size_t i = 0; i < n; ++i) ((float *) outs[c])[i] = op_m(mask, i) ? ((const float *) bases[c])[op_u(index, i)] : 0.f;
    return EK_OK;
}
int ek_hip_gather_multi_plan(int, int, int, size_t base_size, size_t) { return base_size >= 64 ? EK_GATHER_RECORDS : EK_GATHER_ONE_LAUNCH; }
int ek_hip_gather_multi_sized(int type, int index_type, int count, void *const *outs, const void *const *bases, size_t base_size,
                              const ek_operand *index, const ek_operand *mask, size_t n) {
    if (base_size >= 64) ++g_record_gathers;
    return ek_hip_gather_multi(type, index_type, count, outs, bases, index, mask, n);
}
int ek_hip_map_gathered(int arity, int op, int, void *out, const ek_operand *const *o, const ek_gathered *const *g, size_t n) {
    ++g_fused_calls;
    for (size_t i = 0; i < n; ++i) {
        float x[3] = { 0, 0, 0 };
        for (int k = 0; k < arity; ++k)
            x[k] = g[k] ? (op_m(&g[k]->mask, i) ? ((const float *) g[k]->table)[op_u(&g[k]->index, i)] : 0.f) : op_f(o[k], i);
        if (arity == 3 && (op == EK_FNMADD || op == EK_FNMSUB || op == EK_NMULADD)) x[0] = -x[0];
        if (arity == 3 && (op == EK_FMSUB || op == EK_FNMSUB || op == EK_MULSUB)) x[2] = -x[2];
        volatile float prod = x[0] * x[1];
        ((float *) out)[i] = arity == 2 ? binary_f(op, x[0], x[1]) : op >= EK_MULADD ? prod + x[2] : std::fma(x[0], x[1], x[2]);
    }
    return EK_OK;
}
// bucket-ordered evaluation (ek_hip_bucketed_*): the stand-in keeps the operands and evaluates in ELEMENT order -- the host
// logic of the binding (which nodes stay unevaluated, who holds what, when the partition dies) is what the checkers exercise
struct ek_hip_bucketed {
    int op;
    const float *a, *c;
    size_t table_size, n;
    float *x;
    uint32_t *idx;
    float *u;
    uint8_t *mask = nullptr;       // null: every lane is active
};
static long g_bucketed_live = 0, g_bucketed_reduces = 0, g_bucketed_scatters = 0;
static float bucketed_u(const ek_hip_bucketed *b, size_t i) {
    if (b->mask && !b->mask[i]) return 0.f;                // masked-out lanes gather 0 (the device path drops them: u = 0)
    float a = b->a[b->idx[i]], c = b->c ? b->c[b->idx[i]] : -0.0f;        // no addend table: the product alone
    if (b->op == EK_FNMADD || b->op == EK_FNMSUB || b->op == EK_NMULADD) a = -a;
    if (b->op == EK_FMSUB || b->op == EK_FNMSUB || b->op == EK_MULSUB) c = -c;
    volatile float prod = a * b->x[i];
    return b->op >= EK_MULADD ? prod + c : std::fma(a, b->x[i], c);
}
int ek_hip_bucketed_applicable(int type, int index_type, size_t table_size, size_t n) {
    return type == EK_F32 && (index_type == EK_U32 || index_type == EK_I32) && table_size >= 8 && n >= 16;
}
int ek_hip_bucketed_pair_create(int type, int index_type, int op, const void *a, const void *c, size_t table_size, const void *x,
                                const void *index, size_t n, ek_hip_bucketed **out) {
    if (!ek_hip_bucketed_applicable(type, index_type, table_size, n)) return EK_ERR_UNSUPPORTED;
    ek_hip_bucketed *b = new ek_hip_bucketed{ op, (const float *) a, (const float *) c, table_size, n, nullptr, nullptr, nullptr };
    // like the device version: x and index are only read here, the tables by later calls
    b->x = (float *) malloc(n * sizeof(float)); memcpy(b->x, x, n * sizeof(float));
    b->idx = (uint32_t *) malloc(n * sizeof(uint32_t)); memcpy(b->idx, index, n * sizeof(uint32_t));
    ++g_bucketed_live;
    *out = b;
    return EK_OK;
}
int ek_hip_bucketed_pair_create_hinted(int type, int index_type, int op, const void *a, const void *c, size_t table_size, const void *x,
                                       const void *index, size_t n, unsigned /* hints: this stand-in works in element order */,
                                       ek_hip_bucketed **out) {
    return ek_hip_bucketed_pair_create(type, index_type, op, a, c, table_size, x, index, n, out);
}
int ek_hip_bucketed_pair_create_masked(int type, int index_type, int op, const void *a, const void *c, size_t table_size, const void *x,
                                       const void *index, const uint8_t *mask, size_t n, unsigned, ek_hip_bucketed **out) {
    int rc = ek_hip_bucketed_pair_create(type, index_type, op, a, c, table_size, x, index, n, out);
    if (rc == EK_OK && mask) { (*out)->mask = (uint8_t *) malloc(n); memcpy((*out)->mask, mask, n); }
    return rc;
}
int ek_hip_bucketed_reduce(ek_hip_bucketed *b, int op, int map, void *out, int keep, int /* keep_op: a hint, u is kept */) {
    ++g_bucketed_reduces;
    float *u = (float *) malloc(b->n * sizeof(float));
    for (size_t i = 0; i < b->n; ++i) u[i] = b->u ? b->u[i] : bucketed_u(b, i);
    float acc = op == EK_HSUM ? 0.f : op == EK_HPROD ? 1.f : unary_f(map, u[0]);
    for (size_t i = 0; i < b->n; ++i) {
        float v = unary_f(map, u[i]);
        acc = op == EK_HSUM ? acc + v : op == EK_HPROD ? acc * v : op == EK_HMIN ? std::fmin(acc, v) : std::fmax(acc, v);
    }
    *(float *) out = acc;
    if (keep && !b->u) b->u = u; else free(u);
    return EK_OK;
}
static long g_fresh_targets = 0;
int ek_hip_dist_unique_id(void *) { return EK_OK; }
int ek_hip_dist_init(int, int, const void *) { return EK_OK; }
int ek_hip_dist_finalize(void) { return EK_OK; }
int ek_hip_dist_shard_range(size_t n, int rank, int world, size_t *b, size_t *e) { *b = n * rank / world; *e = n * (rank + 1) / world; return EK_OK; }
int ek_hip_dist_all_reduce(int, int, void *, size_t) { return EK_OK; }
int ek_hip_bucketed_early_pair(int map_op, int keep_op) {
    return (map_op == EK_SIN && keep_op == EK_COS) || (map_op == EK_COS && keep_op == EK_SIN) || (map_op == EK_LOG && keep_op == EK_RCP) ||
           (map_op == EK_SQRT && keep_op == EK_RSQRT) || (map_op == EK_RCP && keep_op == EK_RCP_SQR) ||
           (map_op == EK_RSQRT && keep_op == EK_RSQRT_CUBE) || map_op == keep_op;
}
int ek_hip_bucketed_scatter_add_scaled(ek_hip_bucketed *b, int count, void *const *bases, const int *from_u, const int *ops,
                                       const uint64_t *imm, const int *weighted, const int *fresh, const uint64_t *scale) {
    ++g_bucketed_scatters;
    for (int c = 0; c < count; ++c) {
        if (fresh && fresh[c]) {           // the table holds no data yet (malloc'd garbage here): its sums are written
            ++g_fresh_targets;
            for (size_t k = 0; k < b->table_size; ++k) ((float *) bases[c])[k] = 0.f;
        }
        for (size_t i = 0; i < b->n; ++i) {
            if (b->mask && !b->mask[i]) continue;
            float v;
            if (from_u[c]) v = unary_f(ops ? ops[c] : (int) EK_COPY, b->u ? b->u[i] : bucketed_u(b, i));
            else { uint32_t bits = (uint32_t) imm[c]; memcpy(&v, &bits, 4); }
            if (scale) { uint32_t bits = (uint32_t) scale[c]; float f; memcpy(&f, &bits, 4); v = v * f; }
            if (weighted[c]) v = (b->x[i] == 0.f || v == 0.f) ? 0.f : b->x[i] * v;
            ((float *) bases[c])[b->idx[i]] += v;
        }
    }
    return EK_OK;
}
int ek_hip_bucketed_scatter_add(ek_hip_bucketed *b, int count, void *const *bases, const int *from_u, const int *ops,
                                const uint64_t *imm, const int *weighted, const int *fresh) {
    return ek_hip_bucketed_scatter_add_scaled(b, count, bases, from_u, ops, imm, weighted, fresh, nullptr);
}
int ek_hip_bucketed_destroy(ek_hip_bucketed *b) {
    if (b) { free(b->x); free(b->idx); free(b->u); free(b->mask); delete b; --g_bucketed_live; }
    return EK_OK;
}
int ek_hip_scatter(int, int, void *base, const ek_operand *v, const ek_operand *index, const ek_operand *mask, size_t n) {
    for (size_t i = 0; i < n; ++i) if (op_m(mask, i)) ((float *) base)[op_u(index, i)] = op_f(v, i);
    return EK_OK;
}
int ek_hip_scatter_add(int, int, void *base, size_t, const ek_operand *v, const ek_operand *index, const ek_operand *mask, size_t n, int) {
    for (size_t i = 0; i < n; ++i) if (op_m(mask, i)) ((float *) base)[op_u(index, i)] += op_f(v, i);
    return EK_OK;
}
int ek_hip_scatter_add_multi_map(int, int, int count, void *const *bases, size_t, const ek_operand *const *values, const int *ops,
                                 const ek_operand *const *weights, const ek_operand *index, const ek_operand *mask, size_t n, int) {
    for (int c = 0; c < count; ++c)
        for (size_t i = 0; i < n; ++i) {
            if (!op_m(mask, i)) continue;
            float v = op_f(values[c], i);
            if (ops && ops[c] != EK_COPY) v = unary_f(ops[c], v);
            if (weights && weights[c]) v = binary_f(EK_SAFE_MUL, op_f(weights[c], i), v);
            ((float *) bases[c])[op_u(index, i)] += v;
        }
    return EK_OK;
}
int ek_hip_scatter_add_multi(int t, int it, int count, void *const *bases, size_t bs, const ek_operand *const *values,
                             const ek_operand *const *weights, const ek_operand *index, const ek_operand *mask, size_t n, int mode) {
    return ek_hip_scatter_add_multi_map(t, it, count, bases, bs, values, nullptr, weights, index, mask, n, mode);
}
static float reduce_f(int op, int map, const float *in, size_t n) {
    float acc = op == EK_HSUM ? 0.f : op == EK_HPROD ? 1.f : unary_f(map, in[0]);
    for (size_t i = 0; i < n; ++i) {
        float v = unary_f(map, in[i]);
        acc = op == EK_HSUM ? acc + v : op == EK_HPROD ? acc * v : op == EK_HMIN ? std::fmin(acc, v) : std::fmax(acc, v);
    }
    return acc;
}
int ek_hip_reduce(int op, int, void *out, const void *in, size_t n) { *(float *) out = reduce_f(op, EK_COPY, (const float *) in, n); return EK_OK; }
// chains (ek_hip_reduce_chain / ek_hip_map_chain): base op over the sources, then the maps -- element by element
static long g_chain_calls = 0;
static float chain_value(const ek_chain *ch, size_t i) {
    float v[3] = { 0, 0, 0 };
    for (int k = 0; k < ch->arity; ++k) v[k] = op_f(&ch->src[k], i);
    float r = v[0];
    if (ch->arity == 2) r = binary_f(ch->base_op, v[0], v[1]);
    if (ch->arity == 3) {
        ek_operand a{ nullptr, 0, 1 }, b = a, c = a;
        memcpy(&a.imm, &v[0], 4); memcpy(&b.imm, &v[1], 4); memcpy(&c.imm, &v[2], 4);
        ek_hip_ternary(ch->base_op, EK_F32, &r, &a, &b, &c, 1);
    }
    for (int k = 0; k < ch->n_maps; ++k) r = unary_f(ch->map_ops[k], r);
    return r;
}
int ek_hip_map_chain(int, void *out, const ek_chain *ch, size_t n) {
    ++g_chain_calls;
    for (size_t i = 0; i < n; ++i) ((float *) out)[i] = chain_value(ch, i);
    return EK_OK;
}
static long g_chain_product_calls = 0;
int ek_hip_map_chain_product(int, void *out, void *out2, const ek_chain *ch, const ek_operand *scale, int op2, const ek_operand *w, size_t n) {
    ++g_chain_calls;
    if (out2) ++g_chain_product_calls;
    float factor = 1.f;
    if (scale) memcpy(&factor, &scale->imm, 4);
    for (size_t i = 0; i < n; ++i) {
        float v = chain_value(ch, i);
        if (scale) v = binary_f(EK_MUL, v, factor);
        if (out) ((float *) out)[i] = v;
        if (out2) ((float *) out2)[i] = binary_f(op2, op_f(w, i), v);
    }
    return EK_OK;
}
int ek_hip_reduce_chain(int op, int, void *out, const ek_chain *ch, size_t n) {
    ++g_chain_calls;
    std::vector<float> tmp(n);
    for (size_t i = 0; i < n; ++i) tmp[i] = chain_value(ch, i);
    *(float *) out = reduce_f(op, EK_COPY, tmp.data(), n);
    return EK_OK;
}
int ek_hip_reduce_map(int op, int map, int, void *out, const void *in, size_t n) {
    ++g_fused_calls;
    *(float *) out = reduce_f(op, map, (const float *) in, n);
    return EK_OK;
}
} // extern "C"
